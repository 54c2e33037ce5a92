"""Topology header generator + in-tree build of the per-topology HIP library.

The kernels (csrc/jm_kernels.h) are specialised at compile time on the robot topology; this
module writes the `Topo` traits header for a CompiledModel and drives `hipcc` for gfx950.
One shared object per topology: `jiminy_amd/csrc/build/libjm_<topology hash>.so`, exporting the
C ABI of include/jiminy_hip.h.  Built in-tree so that it travels with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from typing import List, Optional

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
    """Detect the 'free-flyer trunk + 4 identical revolute chains' structure served by the
    limb-parallel kernel (csrc/jm_quad.h). Returns a dict of index tables or None."""
    from .model import JT_FREEFLYER, JT_RU, JT_RX, JT_RY, JT_RZ
    nj = model.njoints
    parents = [int(x) for x in model.parents]
    if nj < 6 or int(model.jtypes[1]) != JT_FREEFLYER or parents[1] != 0:
        return None
    children = {j: [c for c in range(1, nj) if parents[c] == j] for j in range(nj)}
    if len(children[0]) != 1 or len(children[1]) != 4:
        return None
    limbs = []
    for root in children[1]:
        chain, j = [], root
        while True:
            if int(model.jtypes[j]) not in (JT_RX, JT_RY, JT_RZ, JT_RU):
                return None
            chain.append(j)
            if len(children[j]) == 0:
                break
            if len(children[j]) != 1:
                return None
            j = children[j][0]
        limbs.append(chain)
    n = len(limbs[0])
    if any(len(c) != n for c in limbs) or 1 + 4 * n != nj - 1:
        return None
    motor_of = {m.joint: i for i, m in enumerate(model.motors)}
    if len(motor_of) != len(model.motors) or set(motor_of) != {j for c in limbs for j in c}:
        return None
    flags = {(m.enable_effort_limit, m.enable_velocity_limit, m.enable_friction) for m in model.motors}
    if len(flags) != 1:
        return None
    # contacts: the same number per limb, all on the last chain joint
    cj = [model.frames[c].parent_joint for c in model.contacts]
    limb_contacts = [[i for i, j in enumerate(cj) if j == c[-1]] for c in limbs]
    ncl = len(limb_contacts[0])
    if any(len(x) != ncl for x in limb_contacts) or 4 * ncl != len(cj):
        return None
    s = model.sensors
    if any(model.frames[x["frame"]].parent_joint != 1 for x in s.get("ImuSensor", [])):
        return None
    fj = [model.frames[x["frame"]].parent_joint for x in s.get("ForceSensor", [])]
    limb_force = [[i for i, j in enumerate(fj) if j == c[-1]] for c in limbs]
    has_force = len(fj) > 0
    if has_force and (any(len(x) != 1 for x in limb_force) or len(fj) != 4):
        return None
    cs = [model.contacts.index(x["frame"]) for x in s.get("ContactSensor", [])]
    has_cs = len(cs) > 0
    limb_cs = []
    for lc in limb_contacts:
        row = []
        for c in lc:
            idx = [i for i, cc in enumerate(cs) if cc == c]
            if has_cs and len(idx) != 1:
                return None
            row.append(idx[0] if idx else -1)
        limb_cs.append(row)
    if has_cs and len(cs) != 4 * ncl:
        return None
    enc = s.get("EncoderSensor", [])
    has_enc = len(enc) > 0
    sides = {bool(x["joint_side"]) for x in enc}
    if has_enc and (len(sides) != 1 or len(enc) != 4 * n):
        return None
    enc_of = {x["joint"]: i for i, x in enumerate(enc)}
    if has_enc and set(enc_of) != {j for c in limbs for j in c}:
        return None
    eff = s.get("EffortSensor", [])
    has_eff = len(eff) > 0
    eff_of = {x["motor_index"]: i for i, x in enumerate(eff)}
    if has_eff and (len(eff) != 4 * n or set(eff_of) != set(range(len(model.motors)))):
        return None
    return {
        "n": n, "ncl": ncl, "limbs": limbs,
        "motor": [[motor_of[j] for j in c] for c in limbs],
        "contact": limb_contacts,
        "force": [x[0] if has_force else -1 for x in limb_force],
        "cs": limb_cs,
        "enc": [[enc_of.get(j, -1) for j in c] for c in limbs],
        "eff": [[eff_of.get(motor_of[j], -1) for j in c] for c in limbs],
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
        _arr("idx_q", model.idx_q),
        _arr("idx_v", model.idx_v),
        _arr("nchildren", nchildren),
        _arr("first_child", first_child),
        _arr("motor_joint", [m.joint for m in model.motors]),
        _arr("motor_flags", mflags),
        _arr("contact_joint", [model.frames[c].parent_joint for c in model.contacts]),
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
    return [
        "    // limb-parallel kernel tables (csrc/jm_quad.h): 4 chains hanging off the free-flyer trunk",
        "    static constexpr bool QUAD = true;",
        f"    static constexpr int QN = {q['n']};",
        f"    static constexpr int QCL = {q['ncl']};",
        f"    static constexpr bool QHAS_FORCE = {'true' if q['has_force'] else 'false'};",
        f"    static constexpr bool QHAS_CS = {'true' if q['has_cs'] else 'false'};",
        f"    static constexpr bool QHAS_ENC = {'true' if q['has_enc'] else 'false'};",
        f"    static constexpr bool QHAS_EFF = {'true' if q['has_eff'] else 'false'};",
        f"    static constexpr int QENC_SIDE = {q['enc_side']};",
        _arr2("limb_joint", q["limbs"]),
        _arr2("limb_motor", q["motor"]),
        _arr2("limb_contact", q["contact"]),
        _arr2("limb_cs", q["cs"]),
        _arr("limb_force", q["force"]),
        _arr2("limb_enc", q["enc"]),
        _arr2("limb_eff", q["eff"]),
    ]


def lib_path(model: CompiledModel) -> str:
    # JIMINY_AMD_LIB_TAG selects an experimental build variant (tuning A/B runs only)
    tag = os.environ.get("JIMINY_AMD_LIB_TAG", "")
    return os.path.join(BUILD, f"libjm_{model.topology_hash()}{('_' + tag) if tag else ''}.so")


def header_path(model: CompiledModel) -> str:
    return os.path.join(BUILD, f"topo_{model.topology_hash()}.h")


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
                                            "jm_pack.h")] + \
           [os.path.join(CSRC, "..", "..", "include", "jiminy_hip.h")]


def is_stale(model: CompiledModel) -> bool:
    lib = lib_path(model)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = _sources() + [header_path(model)]
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def build_library(model: CompiledModel, force: bool = False, verbose: bool = False,
                  extra_flags: Optional[List[str]] = None) -> str:
    """Compile the HIP library specialised for `model`'s topology (gfx950)."""
    hdr = write_header(model)
    lib = lib_path(model)
    if not force and not is_stale(model):
        return lib
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        raise RuntimeError(
            f"hipcc not found ({HIPCC}); cannot build the HIP library for topology "
            f"{model.topology_hash()} and no prebuilt {lib} exists")
    cmd = [HIPCC, f"--offload-arch={OFFLOAD_ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-x", "hip", os.path.join(CSRC, "jm_lib.cpp"),
           f"-DJM_TOPO_HEADER=\"{hdr}\"", "-o", lib + ".tmp",
           "-Wno-unused-value", "-ffp-contract=fast"]
    cmd += extra_flags or []
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    return lib
