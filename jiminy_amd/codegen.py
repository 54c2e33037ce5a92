"""Topology header generator + in-tree build of the per-topology HIP library.

The kernels (csrc/jm_kernels.h) are specialised at compile time on the robot topology; this
module writes the `Topo` traits header for a CompiledModel and drives `hipcc` for gfx950.
One shared object per topology: `jiminy_amd/csrc/build/libjm_<topology hash>.so`, exporting the
C ABI of include/jiminy_hip.h.  Built in-tree so that it travels with the repository snapshot.
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

from .model import CompiledModel

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
BUILD = os.path.join(CSRC, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
OFFLOAD_ARCH = "gfx950"


def _arr(name: str, vals, n_min: int = 1) -> str:
    vals = [int(v) for v in vals]
    if len(vals) < n_min:
        vals = vals + [0] * (n_min - len(vals))
    return f"    static constexpr int {name}[{len(vals)}] = {{{', '.join(str(v) for v in vals)}}};"


def quad_structure(model: CompiledModel):
    """Branch-parallel decomposition served by the 4-lanes-per-robot kernel (csrc/jm_quad.h):

    * the four longest leaf chains of bounded revolute joints ("limbs": lane k of a quad owns
      limb k; shorter limbs are padded at the tip with mass-less dummy joints),
    * everything else ("trunk tree": the free-flyer root plus the 1-dof joints the limbs hang
      from, e.g. Atlas' back chain and neck), evaluated redundantly by the four lanes.

    Requirements: free-flyer root at joint 1, contact points and force sensors only on limb tips,
    IMUs only on trunk-tree joints, one motor per movable joint with uniform option flags, sensor
    coverage all-or-nothing.  Returns a dict of index tables, or None when the tree does not fit
    (the one-robot-per-lane kernel is used then)."""
    from .model import (JT_FREEFLYER, JT_PU, JT_PX, JT_PY, JT_PZ, JT_RU, JT_RX, JT_RY, JT_RZ)
    REV = (JT_RX, JT_RY, JT_RZ, JT_RU)
    ONE_DOF = REV + (JT_PX, JT_PY, JT_PZ, JT_PU)
    nj = model.njoints
    parents = [int(x) for x in model.parents]
    if nj < 6 or int(model.jtypes[1]) != JT_FREEFLYER or parents[1] != 0:
        return None
    if model.constraint_frames or model.constraint_joints:     # user constraints on rows of their own: one-robot-per-lane constraint kernel (jm_constraint.h)
        return None
    if any(int(t) == 14 for t in model.jtypes):                # spherical (flexibility) joints: one-robot-per-lane kernels
        return None
    children = {j: [c for c in range(1, nj) if parents[c] == j] for j in range(nj)}
    if len(children[0]) != 1:
        return None
    # leaf chains: maximal paths of revolute joints ending at a leaf, every node with <= 1 child
    chains = []
    for leaf in [j for j in range(2, nj) if not children[j]]:
        chain, j = [], leaf
        while j > 1 and int(model.jtypes[j]) in REV and len(children[j]) <= 1:
            chain.append(j)
            j = parents[j]
        if chain:
            chains.append(chain[::-1])
    # Two or three leaf chains (a biped without arms, a tripod): the missing limbs are EMPTY -- padded with dummy joints
    # from the first one on, attached to the root, no contact point: their lanes idle, but the robot keeps the
    # register-resident branch-parallel kernel instead of the one-robot-per-lane fallback (kilobytes of scratch per lane)
    if len(chains) < 2:
        return None
    chains.sort(key=lambda c: (-len(c), c[0]))
    limbs = sorted(chains[:4], key=lambda c: c[0])
    limbs += [[] for _ in range(4 - len(limbs))]
    limb_set = {j for c in limbs for j in c}
    trunk = [j for j in range(1, nj) if j not in limb_set]          # topological (index) order
    if any(int(model.jtypes[j]) not in ONE_DOF for j in trunk[1:]):
        return None
    if any(parents[c[0]] not in trunk for c in limbs if c) or any(parents[j] not in trunk for j in trunk[1:]):
        return None
    n = max(len(c) for c in limbs)
    # the padded limbs must do most of the work, otherwise the redundancy does not pay
    if 4 * n + len(trunk) > 2 * (nj - 1):
        return None
    tindex = {j: i for i, j in enumerate(trunk)}
    movable = limb_set | set(trunk[1:])
    motor_of = {m.joint: i for i, m in enumerate(model.motors)}
    if len(motor_of) != len(model.motors) or set(motor_of) != movable:
        return None
    flags = {(m.enable_effort_limit, m.enable_velocity_limit, m.enable_friction) for m in model.motors}
    if len(flags) != 1:
        return None
    pad = lambda row: list(row) + [-1] * (n - len(row))  # noqa: E731
    # contacts: only on limb tips
    cj = [model.frames[c].parent_joint for c in model.contacts]
    limb_contacts = [[i for i, j in enumerate(cj) if c and j == c[-1]] for c in limbs]
    if sum(len(x) for x in limb_contacts) != len(cj):
        return None
    ncl = max([len(x) for x in limb_contacts] + [0])
    padc = lambda row: list(row) + [-1] * (max(ncl, 1) - len(row))  # noqa: E731
    s = model.sensors
    imu = [model.frames[x["frame"]].parent_joint for x in s.get("ImuSensor", [])]
    if any(j not in tindex for j in imu):
        return None
    fj = [model.frames[x["frame"]].parent_joint for x in s.get("ForceSensor", [])]
    limb_force = [[i for i, j in enumerate(fj) if c and j == c[-1]] for c in limbs]
    has_force = len(fj) > 0
    if any(len(x) > 1 for x in limb_force) or sum(len(x) for x in limb_force) != len(fj):
        return None
    cs = [model.contacts.index(x["frame"]) for x in s.get("ContactSensor", [])]
    has_cs = len(cs) > 0
    if has_cs and sorted(cs) != list(range(len(cj))):
        return None
    limb_cs = [[cs.index(c) if has_cs else -1 for c in lc] for lc in limb_contacts]
    enc = s.get("EncoderSensor", [])
    has_enc = len(enc) > 0
    sides = {bool(x["joint_side"]) for x in enc}
    enc_of = {x["joint"]: i for i, x in enumerate(enc)}
    if has_enc and (len(sides) != 1 or len(enc_of) != len(enc) or set(enc_of) != movable):
        return None
    eff = s.get("EffortSensor", [])
    has_eff = len(eff) > 0
    eff_of = {x["motor_index"]: i for i, x in enumerate(eff)}
    if has_eff and (len(eff_of) != len(eff) or set(eff_of) != set(range(len(model.motors)))):
        return None
    return {
        "n": n, "ncl": ncl, "limbs": limbs, "trunk": trunk,
        "trunk_parent": [-1] + [tindex[parents[j]] for j in trunk[1:]],
        "trunk_motor": [-1] + [motor_of[j] for j in trunk[1:]],
        "trunk_enc": [-1] + [enc_of.get(j, -1) for j in trunk[1:]],
        "trunk_eff": [-1] + [eff_of.get(motor_of[j], -1) for j in trunk[1:]],
        "limb_len": [len(c) for c in limbs],
        "limb_attach": [tindex[parents[c[0]]] if c else 0 for c in limbs],
        "limb_joint": [pad(c) for c in limbs],
        "motor": [pad([motor_of[j] for j in c]) for c in limbs],
        "limb_ncontact": [len(x) for x in limb_contacts],
        "contact": [padc(x) for x in limb_contacts],
        "force": [x[0] if x else -1 for x in limb_force],
        "cs": [padc(x) for x in limb_cs],
        "enc": [pad([enc_of.get(j, -1) for j in c]) for c in limbs],
        "eff": [pad([eff_of.get(motor_of[j], -1) for j in c]) for c in limbs],
        "imu_trunk": [tindex[j] for j in imu],
        "has_force": has_force, "has_cs": has_cs, "has_enc": has_enc, "has_eff": has_eff,
        "enc_side": 1 if (has_enc and True in sides) else 0,
    }


def _arr2(name: str, rows) -> str:
    rows = [list(r) if len(r) else [0] for r in rows]
    body = ", ".join("{" + ", ".join(str(int(v)) for v in r) + "}" for r in rows)
    return f"    static constexpr int {name}[{len(rows)}][{len(rows[0])}] = {{{body}}};"


def topology_header(model: CompiledModel) -> str:
    s = model.sensors
    nj = model.njoints
    parents = [int(x) for x in model.parents]
    nchildren = [0] * nj
    first_child = [0] * nj   # child processed first by the backward sweep = largest index
    for j in range(1, nj):
        p = parents[j]
        nchildren[p] += 1
        first_child[p] = max(first_child[p], j)
    imu = [model.frames[x["frame"]].parent_joint for x in s.get("ImuSensor", [])]
    frc = [model.frames[x["frame"]].parent_joint for x in s.get("ForceSensor", [])]
    css = [model.contacts.index(x["frame"]) for x in s.get("ContactSensor", [])]
    enc = s.get("EncoderSensor", [])
    eff = s.get("EffortSensor", [])
    mflags = [(1 if m.enable_effort_limit else 0) | (2 if m.enable_velocity_limit else 0)
              | (4 if m.enable_friction else 0) for m in model.motors]
    sig = model.topology_signature()
    lines = [
        "// GENERATED by jiminy_amd/codegen.py -- robot topology traits (no numeric parameters).",
        f"// model: {model.name}",
        "#pragma once",
        f"#define JM_TOPO_QUAD {1 if quad_structure(model) is not None else 0}",
        "// constraint contact model: solves of more than 32 rows step through pre | solve | post launches (jm_qcon.h)",
        f"#define JM_TOPO_QCON_SPLIT {1 if qcon_split(model) else 0}",
        "// The struct name carries the topology hash so that two topology libraries loaded in the",
        "// same process never share (STB_GNU_UNIQUE / weak) template instantiations.",
        f"struct Topo_{model.topology_hash()}",
        "{",
        f"    static constexpr int NJ = {nj};",
        f"    static constexpr int NQ = {model.nq};",
        f"    static constexpr int NV = {model.nv};",
        f"    static constexpr int NM = {model.nmotors};",
        f"    static constexpr int NC = {model.ncontacts};",
        f"    static constexpr int NIMU = {len(imu)};",
        f"    static constexpr int NFORCE = {len(frc)};",
        f"    static constexpr int NCS = {len(css)};",
        f"    static constexpr int NENC = {len(enc)};",
        f"    static constexpr int NEFF = {len(eff)};",
        _arr("parent", parents),
        _arr("jtype", model.jtypes),
        _arr("axis_signed", model.signed_axes()),   # +-(i + 1): the joint axis is exactly +-e_i; 0: general
        _arr("idx_q", model.idx_q),
        _arr("idx_v", model.idx_v),
        _arr("nchildren", nchildren),
        _arr("first_child", first_child),
        _arr("motor_joint", [m.joint for m in model.motors]),
        _arr("motor_flags", mflags),
        _arr("contact_joint", [model.frames[c].parent_joint for c in model.contacts]),
        f"    static constexpr int NX = {len(model.constraint_frames)};   // user constraint frames (FrameConstraint)",
        _arr("xframe_joint", [model.frames[x["frame"]].parent_joint for x in model.constraint_frames]),
        _arr("xframe_mask", [x["mask"] for x in model.constraint_frames]),
        _arr("xframe_kind", [{"frame": 0, "sphere": 1, "wheel": 2, "distance": 3}[x.get("kind", "frame")] for x in model.constraint_frames]),
        _arr("xframe_joint2", [model.frames[x["frame2"]].parent_joint if x.get("frame2") else 0 for x in model.constraint_frames]),
        f"    static constexpr int NXJ = {len(model.constraint_joints)};   // user constraint joints (JointConstraint rows of their own)",
        _arr("xjoint", [x["joint"] for x in model.constraint_joints]),
        _arr("imu_joint", imu),
        _arr("force_joint", frc),
        _arr("cs_contact", css),
        _arr("enc_joint", [x["joint"] for x in enc]),
        _arr("enc_side", [1 if x["joint_side"] else 0 for x in enc]),
        _arr("eff_motor", [x["motor_index"] for x in eff]),
        f'    static constexpr const char * signature = "{sig}";',
        *_quad_lines(model),
        "};",
        f"using Topo = Topo_{model.topology_hash()};",
        "",
    ]
    return "\n".join(lines)


def _quad_lines(model: CompiledModel):
    q = quad_structure(model)
    if q is None:
        return ["    static constexpr bool QUAD = false;"]
    b = lambda x: "true" if x else "false"  # noqa: E731
    return [
        "    // branch-parallel kernel tables (csrc/jm_quad.h): trunk tree (all lanes) + 4 limbs (one per lane)",
        "    static constexpr bool QUAD = true;",
        f"    static constexpr int QN = {q['n']};",
        f"    static constexpr int QCL = {q['ncl']};",
        f"    static constexpr int QT = {len(q['trunk'])};",
        f"    static constexpr bool QHAS_FORCE = {b(q['has_force'])};",
        f"    static constexpr bool QHAS_CS = {b(q['has_cs'])};",
        f"    static constexpr bool QHAS_ENC = {b(q['has_enc'])};",
        f"    static constexpr bool QHAS_EFF = {b(q['has_eff'])};",
        f"    static constexpr int QENC_SIDE = {q['enc_side']};",
        _arr("trunk_joint", q["trunk"]),
        _arr("trunk_parent", q["trunk_parent"]),
        _arr("trunk_motor", q["trunk_motor"]),
        _arr("trunk_enc", q["trunk_enc"]),
        _arr("trunk_eff", q["trunk_eff"]),
        _arr("imu_trunk", q["imu_trunk"]),
        _arr("limb_len", q["limb_len"]),
        _arr("limb_attach", q["limb_attach"]),
        _arr("limb_ncontact", q["limb_ncontact"]),
        _arr2("limb_joint", q["limb_joint"]),
        _arr2("limb_motor", q["motor"]),
        _arr2("limb_contact", q["contact"]),
        _arr2("limb_cs", q["cs"]),
        _arr("limb_force", q["force"]),
        _arr2("limb_enc", q["enc"]),
        _arr2("limb_eff", q["eff"]),
    ]


# Build variants.  hipcc 7.2 mis-compiles the evaluation loop of some register-bound topologies
# (DESIGN.md section 4.7): the engine verifies every library on first use (BatchedEngine self-test)
# and moves on to the next variant when a build fails the check.  `build_variants.json` (tracked)
# records the variant a topology is known to need, so that `__graft_entry__.build()` compiles it
# ahead of time.
# Variant 0 (round 4): every translation unit is built with the BASIC SGPR register allocator instead of the greedy one.
# It is the one backend switch that repairs all three mis-compiled kernels found at -O3 -- tree_arm's k_batch, the output pass of
# tree_arm_ff's k_batch, Atlas' persistent adaptive kernels -- bit for bit against their -O1 builds, where twelve other
# switches repair at most one; with it no library needs a pin.  Cost (MI355X, same box, two runs
# each): ANYmal step launch 0.1507 -> 0.1495 ms, ANYmal constraint launch 0.920 -> 0.938, Atlas step launch 0.348 -> 0.362,
# Atlas constraint launch unchanged.  Variant 1 is the compiler's default allocator, variant 2 -O1: the fall-back chain of
# the self-tests, and what `tests/test_gpu_parity.py` forces to exercise that chain.
BUILD_VARIANTS: Tuple[Tuple[str, ...], ...] = (
    ("-mllvm", "-sgpr-regalloc=basic"),
    (),
    ("-O1",),
)
_VARIANT_FILE = os.path.join(CSRC, "build_variants.json")


def preferred_variant(model: CompiledModel) -> int:
    env = os.environ.get("JIMINY_AMD_BUILD_VARIANT")
    if env is not None:
        return int(env)
    try:
        with open(_VARIANT_FILE) as f:
            return int(json.load(f).get(model.topology_hash(), {}).get("variant", 0))
    except (OSError, ValueError):
        return 0


def qcon_split(model: CompiledModel) -> bool:
    """`jm::qcon_split<Topo>()` (jm_qcon.h): branch-parallel topology whose constraint solves can exceed 32 rows
    (bounded joints + 4 per contact point, capped at JM_QCON_MAXM = 96)."""
    if quad_structure(model) is None:
        return False
    nb = sum(1 for t in model.jtypes[1:] if 1 <= int(t) <= 8)
    # ... or whose whole solve fits the fixed 16-row layout of the one-lane-per-robot solve (`jm::qcon_split_lane`, round 6)
    lane = 1 <= model.ncontacts and 3 * model.ncontacts <= 16
    return min(nb + 4 * model.ncontacts, 96) > qcon_split_min() or lane


def qcon_split_min() -> int:
    """Solves of more rows than this step through pre | solve | post launches (`JM_QCON_SPLIT_MIN` of jm_qcon.h, 32).
    JIMINY_AMD_QCON_SPLIT_MIN builds an experimental library with another threshold (use with JIMINY_AMD_LIB_TAG: round 5
    measured ANYmal -- 28 rows -- in the split form)."""
    return int(os.environ.get("JIMINY_AMD_QCON_SPLIT_MIN", "32"))


# Flags every topology compiles a unit with.  Unit 1 (the one-robot-per-lane constraint kernel): without the IR
# load-store vectorizer.  Its scalar-register pressure is far beyond the file; under the basic SGPR allocator (the default
# of every build) a spilled 16-dword tuple of model constants is re-loaded WHOLE (s_load_dwordx16 +
# wait) in front of every single use: narrow loads halve the instructions of the kernel (round 5: 0.29 -> 0.25 ms on the
# 7-joint arm, 0.22 -> 0.18 on `tree_arm`).
DEFAULT_PART_FLAGS: Dict[str, List[str]] = {"1": ["-mllvm", "-amdgpu-load-store-vectorizer=0"]}


def part_flags(model: CompiledModel) -> Dict[str, List[str]]:
    """Extra flags of single translation units of a topology (build_variants.json `part_flags`: {"5": ["-O1"]} compiles
    the persistent adaptive kernel of that topology at -O1), appended after the common flags and DEFAULT_PART_FLAGS.
    JIMINY_AMD_NO_PART_FLAGS=1 ignores the per-topology ones (re-testing whether a pinned unit still needs its flags)."""
    flags = {k: list(v) for k, v in DEFAULT_PART_FLAGS.items()}
    if os.environ.get("JIMINY_AMD_NO_PART_FLAGS") == "1":
        return flags
    try:
        with open(_VARIANT_FILE) as f:
            for k, v in json.load(f).get(model.topology_hash(), {}).get("part_flags", {}).items():
                flags[str(k)] = flags.get(str(k), []) + list(v)
    except (OSError, ValueError):
        pass
    return flags


def lib_path(model: CompiledModel, variant: Optional[int] = None) -> str:
    v = preferred_variant(model) if variant is None else variant
    # JIMINY_AMD_LIB_TAG selects an experimental build (tuning A/B runs only)
    tag = os.environ.get("JIMINY_AMD_LIB_TAG", "")
    return os.path.join(BUILD, f"libjm_{model.topology_hash()}{('_v%d' % v) if v else ''}"
                               f"{('_' + tag) if tag else ''}.so")


def header_path(model: CompiledModel) -> str:
    # (experimental builds that change what the header says keep a header of their own)
    tag = os.environ.get("JIMINY_AMD_LIB_TAG", "") if qcon_split_min() != 32 else ""
    return os.path.join(BUILD, f"topo_{model.topology_hash()}{('_' + tag) if tag else ''}.h")


def write_header(model: CompiledModel) -> str:
    os.makedirs(BUILD, exist_ok=True)
    path = header_path(model)
    text = topology_header(model)
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


def _sources() -> List[str]:
    return [os.path.join(CSRC, n) for n in ("jm_lib.cpp", "jm_kernels.h", "jm_math.h", "jm_quad.h",
                                            "jm_pack.h", "jm_adaptive.h", "jm_blocks.h", "jm_random.h",
                                            "jm_constraint.h", "jm_qcon.h", "jm_qtip.h", "jm_qdopri.h", "jm_lib_constraint.cpp")] + \
           [os.path.join(CSRC, "..", "..", "include", "jiminy_hip.h")]


def source_digest(model: CompiledModel, variant: Optional[int] = None, extra_flags: Optional[List[str]] = None) -> str:
    """Digest of everything a topology library is compiled from: the kernel sources, the generated topology header
    and the flags of its build variant.  Stored next to the library (`<lib>.src`) when it is built."""
    import hashlib
    v = preferred_variant(model) if variant is None else variant
    h = hashlib.sha256()
    for path in _sources():
        # (jm_qtip.h only reaches the code of topologies that step in the split form: the others are not rebuilt for it)
        if os.path.basename(path) == "jm_qtip.h" and not qcon_split(model):
            continue
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(topology_header(model).encode())
    h.update(" ".join(BUILD_VARIANTS[v]).encode() if v < len(BUILD_VARIANTS) else b"?")
    h.update(json.dumps(part_flags(model), sort_keys=True).encode())
    h.update(" ".join(extra_flags or []).encode())
    return h.hexdigest()


def is_stale(model: CompiledModel, variant: Optional[int] = None) -> bool:
    """A library is stale when it was built from other sources than the ones in the tree: content digest recorded at
    build time (file times do not survive a checkout or a copy to another machine); libraries without a record fall
    back to the file times."""
    lib = lib_path(model, variant)
    if not os.path.exists(lib):
        return True
    try:
        with open(lib + ".src") as f:
            recorded = f.read().split()
        if os.environ.get("JIMINY_AMD_LIB_TAG"):
            return False   # experimental builds carry their own flags: never rebuilt behind the experimenter's back
        return recorded[0] != source_digest(model, variant)
    except (OSError, IndexError):
        pass
    t = os.path.getmtime(lib)
    deps = _sources() + [header_path(model)]
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def build_library(model: CompiledModel, force: bool = False, verbose: bool = False,
                  extra_flags: Optional[List[str]] = None, variant: Optional[int] = None) -> str:
    """Compile the HIP library specialised for `model`'s topology (gfx950)."""
    hdr = write_header(model)
    v = preferred_variant(model) if variant is None else variant
    lib = lib_path(model, v)
    if not force and not is_stale(model, v):
        return lib
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        raise RuntimeError(
            f"hipcc not found ({HIPCC}); cannot build the HIP library for topology "
            f"{model.topology_hash()} and no prebuilt {lib} exists")
    # up to five translation units compiled in parallel (the constraint-model kernels are the longest single compiles
    # of a large topology), then linked into one shared library
    common = [f"--offload-arch={OFFLOAD_ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
              f"-DJM_TOPO_HEADER=\"{hdr}\"", "-Wno-unused-value", "-ffp-contract=fast"]
    common += list(BUILD_VARIANTS[v])
    common += extra_flags or []
    if qcon_split_min() != 32:
        common.append(f"-DJM_QCON_SPLIT_MIN={qcon_split_min()}")
    # (15, 16: the one-robot-per-lane kernels in the form that reads applied wrenches -- topologies without branch-parallel kernels)
    parts = [1, 2, 3, 4, 5, 6, 11, 12] if quad_structure(model) is not None else [1, 15, 16]
    if qcon_split(model):
        parts += [7, 8, 9, 10, 13, 14]
    objs = [lib + ".main.o"] + [lib + f".part{p}.o" for p in parts]
    cmds = [[HIPCC] + common + ["-DJM_SPLIT_CONSTRAINT", "-c", os.path.join(CSRC, "jm_lib.cpp"), "-o", objs[0]]]
    pf = part_flags(model)
    cmds += [[HIPCC] + common + pf.get(str(p), []) + [f"-DJM_CON_PART={p}", "-c", os.path.join(CSRC, "jm_lib_constraint.cpp"), "-o", o]
             for p, o in zip(parts, objs[1:])]
    if verbose:
        for c in cmds:
            print(" ".join(c))
    procs = [subprocess.Popen(c) for c in cmds]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(next(r for r in rcs if r), cmds[[bool(r) for r in rcs].index(True)])
    link = [HIPCC, f"--offload-arch={OFFLOAD_ARCH}", "-fPIC", "-shared", *objs, "-o", lib + ".tmp"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    os.replace(lib + ".tmp", lib)
    with open(lib + ".src", "w") as f:
        f.write(source_digest(model, v, extra_flags) + "\n")
    return lib
