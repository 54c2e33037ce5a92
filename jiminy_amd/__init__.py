"""jiminy_amd: MI355X-native batched rigid-body dynamics behind the jiminy_py /
gym_jiminy reset/step/observe surface (hot path only)."""
from .model import (CompiledModel, build_model_from_urdf, build_robot, load_builtin,
                    load_hardware_description_file, add_motor, add_sensor,
                    add_contact_points, add_frame, add_frame_constraint, add_joint_constraint,
                    add_sphere_constraint, add_wheel_constraint, add_distance_constraint)

__all__ = ["CompiledModel", "build_model_from_urdf", "build_robot", "load_builtin",
           "load_hardware_description_file", "add_motor", "add_sensor",
           "add_contact_points", "add_frame", "add_frame_constraint", "add_joint_constraint", "add_sphere_constraint",
           "add_wheel_constraint", "add_distance_constraint"]
