"""The C-ABI shared libraries load and export every symbol include/jiminy_hip.h declares;
the entry points that do not touch the device behave (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from jiminy_amd import _abi, _lib, codegen, load_builtin
from tests import robots

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "jiminy_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jm_[a-z_]+)\s*\(", text)))


def _prebuilt(model):
    path = codegen.lib_path(model)
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not built (run __graft_entry__.build())")
    return _lib.load_for(model, allow_build=False)


def test_header_and_loader_agree_on_the_symbol_list():
    assert _declared_symbols() == sorted(_lib.ABI_SYMBOLS)


@pytest.mark.parametrize("name", ["double_pendulum", "cartpole", "anymal", "atlas"])
def test_library_exports_every_declared_symbol(name):
    model = load_builtin(name)
    lib = _prebuilt(model)
    for sym in _declared_symbols():
        assert hasattr(lib.L, sym), sym
    assert lib.signature() == model.topology_signature()
    assert lib.L.jm_abi_version() == _abi.ABI_VERSION


def test_model_create_validates_the_topology():
    model = load_builtin("cartpole")
    lib = _prebuilt(model)
    desc, keep = _abi.make_model_desc(model)
    h = C.c_void_p()
    assert lib.L.jm_model_create(C.byref(desc), C.byref(h)) == _abi.JM_OK and h.value
    assert lib.L.jm_model_destroy(h) == _abi.JM_OK
    other, keep2 = _abi.make_model_desc(load_builtin("double_pendulum"))
    h2 = C.c_void_p()
    rc = lib.L.jm_model_create(C.byref(other), C.byref(h2))
    assert rc == _abi.JM_ETOPOLOGY
    with pytest.raises(_lib.TopologyMismatch, match="topology mismatch"):
        lib.check(rc)
    assert lib.L.jm_model_create(None, C.byref(h2)) == _abi.JM_EINVAL


def test_two_topology_libraries_coexist_in_one_process():
    a, b = load_builtin("cartpole"), load_builtin("anymal")
    la, lb = _prebuilt(a), _prebuilt(b)
    assert la.signature() == a.topology_signature() and lb.signature() == b.topology_signature()
    for lib, model in ((la, a), (lb, b)):
        desc, keep = _abi.make_model_desc(model)
        h = C.c_void_p()
        assert lib.L.jm_model_create(C.byref(desc), C.byref(h)) == _abi.JM_OK
        lib.L.jm_model_destroy(h)


def test_generated_topology_header():
    text = codegen.topology_header(load_builtin("anymal"))
    assert "static constexpr bool QUAD = true;" in text and "limb_joint[4][3]" in text
    assert "QUAD = false" in codegen.topology_header(robots.tree_arm(True))
    # Atlas: back chain + neck form the trunk tree, arms (7) and legs (6, padded) are the limbs
    q = codegen.quad_structure(load_builtin("atlas"))
    assert q is not None and q["n"] == 7 and q["limb_len"] == [7, 7, 6, 6]
    assert q["limb_attach"] == [3, 3, 0, 0] and q["limb_ncontact"] == [0, 0, 16, 16]
    assert len(q["trunk"]) == 5 and q["imu_trunk"] == [3]
    assert codegen.quad_structure(load_builtin("cartpole")) is None


def test_model_desc_packing():
    model = load_builtin("anymal")
    d, keep = _abi.make_model_desc(model)
    assert (d.njoints, d.nq, d.nv, d.nmotors, d.ncontacts) == (14, 19, 18, 12, 4)
    assert (d.nimu, d.nforce, d.nencoder, d.neffort) == (1, 4, 12, 12)
    assert [d.parents[i] for i in range(14)] == [int(x) for x in model.parents]
    assert d.motor_flags[0] == _abi.JM_MOTOR_EFFORT_LIMIT | _abi.JM_MOTOR_VELOCITY_LIMIT
    assert d.motor_params[1] == 80.0 and d.motor_params[2] == 7.5 and d.motor_params[3] == 0.02
    rows = _abi.field_rows(model)
    assert rows["imu"] == 6 and rows["force"] == 24 and rows["encoder"] == 24 and rows["effort"] == 12


def test_build_variants_bookkeeping(monkeypatch, tmp_path):
    """codegen.BUILD_VARIANTS / build_variants.json: the variant recorded for a topology selects the library file the
    loader opens; variant 0 (the default of every topology since round 4: no pin is left) builds with the basic SGPR
    allocator, variant 1 with the compiler's default flags, variant 2 at -O1."""
    import json
    model = robots.crane_walker()
    monkeypatch.delenv("JIMINY_AMD_BUILD_VARIANT", raising=False)
    monkeypatch.delenv("JIMINY_AMD_LIB_TAG", raising=False)
    assert codegen.BUILD_VARIANTS[0] == ("-mllvm", "-sgpr-regalloc=basic") and codegen.BUILD_VARIANTS[1] == ()
    recorded = json.load(open(os.path.join(codegen.CSRC, "build_variants.json")))
    # (keys with a leading underscore are notes: `_dropped` keeps the history of the pins that were removed)
    pins = {k: v for k, v in recorded.items() if not k.startswith("_")}
    assert all(0 <= int(v.get("variant", 0)) < len(codegen.BUILD_VARIANTS) for v in pins.values())
    for m in (model, robots.tree_arm(False), robots.tree_arm(True)):
        want = int(pins.get(m.topology_hash(), {"variant": 0}).get("variant", 0))
        assert codegen.preferred_variant(m) == want
        assert codegen.lib_path(m).endswith(f"libjm_{m.topology_hash()}" + (f"_v{want}.so" if want else ".so"))
    assert codegen.lib_path(model, 0).endswith(f"libjm_{model.topology_hash()}.so")
    assert codegen.lib_path(model, 2).endswith(f"libjm_{model.topology_hash()}_v2.so")
    monkeypatch.setenv("JIMINY_AMD_BUILD_VARIANT", "2")
    assert codegen.preferred_variant(model) == 2
    # a pin in the file is honoured (per-topology variant and per-unit flags)
    monkeypatch.delenv("JIMINY_AMD_BUILD_VARIANT")
    fake = tmp_path / "pins.json"
    fake.write_text(json.dumps({model.topology_hash(): {"variant": 1, "part_flags": {"5": ["-O1"]}}, "_note": "x"}))
    monkeypatch.setattr(codegen, "_VARIANT_FILE", str(fake))
    assert codegen.preferred_variant(model) == 1
    assert codegen.part_flags(model) == {**codegen.DEFAULT_PART_FLAGS, "5": ["-O1"]}
    monkeypatch.setenv("JIMINY_AMD_NO_PART_FLAGS", "1")
    assert codegen.part_flags(model) == codegen.DEFAULT_PART_FLAGS
    monkeypatch.setattr(codegen, "_VARIANT_FILE", str(tmp_path / "absent.json"))
    assert codegen.preferred_variant(model) == 0


def test_probe_state_of_the_library_self_test_is_valid():
    """engine._probe_state: inside the joint bounds, unit quaternion / unit (cos, sin) pairs."""
    from jiminy_amd import engine
    for model in (load_builtin("atlas"), robots.tree_arm(True), robots.crane_walker(), load_builtin("cartpole")):
        q, v, cmd = engine._probe_state(model, 16)
        assert q.shape == (model.nq, 16) and v.shape == (model.nv, 16) and cmd.shape == (model.nmotors, 16)
        m = model.bounded_position_mask()
        assert np.all(q[m] >= model.position_lower[m][:, None] - 1e-12)
        assert np.all(q[m] <= model.position_upper[m][:, None] + 1e-12)
        if model.has_freeflyer:
            assert np.allclose(np.linalg.norm(q[3:7], axis=0), 1.0)
        assert np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(cmd).all()
