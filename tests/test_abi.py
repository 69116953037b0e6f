"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without
a GPU and exports every symbol include/hironaka_hip.h declares; the python mirror of the header
agrees with it; the product never imports the oracle."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from hironaka_amd import _abi as A
from hironaka_amd import _lib


def _header():
    with open(os.path.join(ROOT, "include", "hironaka_hip.h")) as f:
        return f.read()


def test_library_exports_every_declared_symbol():
    _lib.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = set(re.findall(r"^(?:int|uint64_t|const char\*)\s+(hk_\w+)\(", _header(), flags=re.M))
    assert declared == set(A.PROTOTYPES), declared ^ set(A.PROTOTYPES)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().hk_abi_version() == A.HK_ABI_VERSION
    assert _lib.lib().hk_strerror(A.HK_ERR_SHAPE).decode().startswith("batch")


def test_python_constants_match_header():
    text = _header()
    for name, value in re.findall(r"#define\s+(HK_\w+)\s+\(?(-?\d+)u?\)?\s", text):
        assert getattr(A, name) == int(value), name


def test_descriptor_layout_matches_c():
    """sizeof/offsetof as the C compiler sees them (gcc on the header) == ctypes."""
    import subprocess, tempfile
    fields = {"hk_step_desc": [f[0] for f in A.hk_step_desc._fields_],
              "hk_rollout_desc": [f[0] for f in A.hk_rollout_desc._fields_],
              "hk_search_tree": [f[0] for f in A.hk_search_tree._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "hironaka_hip.h"', 'int main(){']
    for st, fl in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fl:
            src.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src.append('return 0;}')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "layout.c")
        open(c, "w").write("\n".join(src))
        exe = os.path.join(td, "layout")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(line.split() for line in out if line)
    for st, cls in (("hk_step_desc", A.hk_step_desc), ("hk_rollout_desc", A.hk_rollout_desc),
                    ("hk_search_tree", A.hk_search_tree)):
        assert int(got[st]) == ctypes.sizeof(cls)
        for f in fields[st]:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, (st, f)


def test_argument_validation_without_gpu():
    """status codes for bad arguments are decided on the host, before any launch"""
    L = _lib.lib()
    s = A.hk_step_desc()
    s.batch, s.max_points, s.dim, s.dtype = 4, 0, 3, A.HK_F32
    assert L.hk_step(ctypes.byref(s), None) == A.HK_ERR_SHAPE
    s.max_points = 5
    assert L.hk_step(ctypes.byref(s), None) == A.HK_ERR_NULL
    s.dtype = A.HK_I32
    assert L.hk_step(ctypes.byref(s), None) == A.HK_ERR_UNSUPPORTED
    assert L.hk_step(None, None) == A.HK_ERR_NULL
    assert L.hk_generate_points(None, 0, 5, 3, A.HK_F32, 10, 1, 0, 0, -1.0, 0, None) == A.HK_OK  # empty batch
    assert L.hk_generate_points(None, 8, 5, 3, A.HK_F32, 10, 1, 0, 0, -1.0, 0, None) == A.HK_ERR_NULL
    assert L.hk_has_fast_path(20, 3, A.HK_F32) == 1 and L.hk_has_fast_path(21, 3, A.HK_F32) == 0


def test_rollout_and_search_validation_without_gpu():
    """the rollout / deferred-count / search entry points refuse bad descriptors on the host"""
    L = _lib.lib()
    r = A.hk_rollout_desc()
    r.batch, r.max_points, r.dim, r.dtype, r.steps = 64, 20, 3, A.HK_F32, 5
    assert L.hk_rollout(ctypes.byref(r), None) == A.HK_ERR_NULL          # no state
    assert L.hk_rollout_reduce_counts(ctypes.byref(r), None) == A.HK_ERR_NULL   # no workspace / done_count
    assert L.hk_rollout_reduce_counts(None, None) == A.HK_ERR_NULL
    r.steps = -1
    assert L.hk_rollout(ctypes.byref(r), None) == A.HK_ERR_SHAPE
    r.steps, r.batch = 5, 0
    assert L.hk_rollout(ctypes.byref(r), None) == A.HK_OK                # empty batch: nothing to do
    r.batch = 64
    assert L.hk_rollout_workspace_bytes(ctypes.byref(r)) == 0            # invalid descriptor (no state)
    buf = (ctypes.c_uint8 * 64)()
    r.points = ctypes.addressof(buf)
    need = L.hk_rollout_workspace_bytes(ctypes.byref(r))
    assert need == 4 * 6 * 4   # (steps + 1) rows of counters, one slot per 16 games: the finest kernel variant's grid
    r.flags = A.HK_FLAG_DEFER_COUNTS
    assert L.hk_rollout(ctypes.byref(r), None) == A.HK_ERR_NULL          # deferred counts need the workspace
    t = A.hk_search_tree()
    t.batch, t.num_nodes, t.num_actions = 8, 9, 4
    assert L.hk_search_policy(ctypes.byref(t), None, None, None, None, None) == A.HK_ERR_NULL
    t.num_actions = 33
    assert L.hk_search_policy(ctypes.byref(t), None, None, None, None, None) == A.HK_ERR_SHAPE
    t.num_actions, t.batch = 4, 0
    assert L.hk_search_select(ctypes.byref(t), None, None, None, 4, 8, 8, 1, None, None, None, None) == A.HK_OK
    assert L.hk_search_backup(None, None, None, None, None, None, None, None, None) == A.HK_ERR_NULL


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "hironaka_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn
                assert "libhironaka_oracle" not in text and "np_oracle" not in text, fn


def test_ops_refuse_cpu_tensors():
    import torch
    from hironaka_amd import ops
    with pytest.raises(TypeError):
        ops.get_newton_polytope(torch.zeros(2, 4, 3))
    with pytest.raises(TypeError):
        ops.get_dones(torch.zeros(2, 4, 3))


def test_graft_entry_build_runs():
    """__graft_entry__.build() -- the driver's "does it build" check -- compiles (make: up to date here), loads the
    library, checks its ABI version against the python binding and builds the oracle"""
    import __graft_entry__ as G
    G.build()
    from hironaka_amd import _abi, _lib
    assert _lib.lib().hk_abi_version() == _abi.HK_ABI_VERSION


def test_inline_asm_keeps_the_scalar_register_hazard_distance(tmp_path):
    """gfx950 wants two wait states between a vector instruction that writes a scalar register pair and a vector
    instruction that reads it; the compiler's hazard recogniser does not look into inline asm, and the borrow-chain
    blocks (hk_fast_rows.h: kb_rank3 / kb_rank2 / zeil_pair2, the single chain's s_nop) provide the distance
    themselves.  A device listing of the kernels that use them is scanned for the smallest distance."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_sgpr_hazards
    src = os.path.join(ROOT, "hironaka_amd", "csrc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-mllvm",
             "-amdgpu-kernarg-preload-count=8", "-S", "--cuda-device-only"]
    jobs = []
    for tu, m, d in (("hk_quad_spec.hip", 20, 3), ("hk_quadroll_spec.hip", 20, 4)):
        out = str(tmp_path / f"{tu}_{m}_{d}.s")
        jobs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, f"-DHK_SPEC_M={m}", f"-DHK_SPEC_D={d}",
                                            os.path.join(src, tu), "-o", out], stderr=subprocess.DEVNULL)))
    for out, proc in jobs:
        assert proc.wait() == 0
        violations, distances = check_sgpr_hazards.scan(out)
        assert violations == 0, distances
        assert distances.get("v_subb_co_u32_e64", 2) >= 2 and distances.get("v_addc_co_u32_e64", 2) >= 2
