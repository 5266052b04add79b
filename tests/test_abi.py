"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/*.h declares,
and argument validation fails loudly without touching a GPU."""
import ctypes as C

import numpy as np
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(dgppo_[a-z0-9_]+)\s*\(", src):
            names.add(m.group(1))
    return sorted(names)


def test_library_loads_and_exports_all_declared_symbols():
    from dgppo_amd import _native as N
    lib = N.lib()
    syms = _declared_symbols()
    assert len(syms) >= 8
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert lib.dgppo_abi_version() == N.ABI_VERSION
    # and the other direction: nothing is exported that the header does not declare (debug hooks of -DDGPPO_STAMPS builds aside)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T dgppo_" in ln}
    undeclared = sorted(s for s in exported - set(syms) if not s.startswith("dgppo_debug_"))
    assert not undeclared, f"exported but not declared in include/: {undeclared}"


def test_cfg_struct_matches_header_and_sizes():
    from dgppo_amd import _native as N
    assert C.sizeof(N.EnvCfg) == 8 * 4 + 12 * 4 + 8 * 4          # ABI 3: + the eight task-variant fields
    cfg = N.make_env_cfg(0, 8, 3)
    lib = N.lib()
    lib.dgppo_env_num_nodes.restype = C.c_int32
    lib.dgppo_env_num_edges.restype = C.c_int32
    assert lib.dgppo_env_num_nodes(C.byref(cfg)) == 81 == cfg.num_nodes
    assert lib.dgppo_env_num_edges(C.byref(cfg)) == 192 == cfg.num_edges
    cfg5 = N.make_env_cfg(2, 16, 8)
    assert lib.dgppo_env_num_nodes(C.byref(cfg5)) == 161 and lib.dgppo_env_num_edges(C.byref(cfg5)) == 400
    cfg2 = N.make_env_cfg(3, 3, 3)
    assert lib.dgppo_env_num_nodes(C.byref(cfg2)) == 10 and lib.dgppo_env_num_edges(C.byref(cfg2)) == 27
    # task variants: 2 landmark nodes (Line), 1 (Formation), fixed obstacle counts (Corridor 2, ConnectSpread 1)
    for kind, n, n_obs, nodes, edges in ((5, 4, 3, 4 + 2 + 32 + 1, 4 * (4 + 2 + 8)), (6, 5, 2, 5 + 2 + 2 + 1, 5 * (5 + 2 + 2)),
                                         (7, 4, 3, 4 + 1 + 3 + 1, 4 * (4 + 1 + 3)), (8, 4, 5, 4 + 4 + 2 + 1, 4 * (4 + 4 + 2)),
                                         (9, 4, 0, 4 + 4 + 1 + 1, 4 * (4 + 4 + 1))):
        cv = N.make_env_cfg(kind, n, n_obs)
        assert lib.dgppo_env_num_nodes(C.byref(cv)) == nodes == cv.num_nodes, (kind, lib.dgppo_last_error())
        assert lib.dgppo_env_num_edges(C.byref(cv)) == edges == cv.num_edges
    assert N.make_env_cfg(9, 4, 0).n_cost == 3 and N.make_env_cfg(8, 4, 0).obs_radius == np.float32((1.0 - 0.2) / 4)
    bad = N.make_env_cfg(6, 5, 2)
    bad.n_goals = 5                                               # MPELine must carry its 2 landmarks
    assert lib.dgppo_env_num_nodes(C.byref(bad)) == -1
    cfg1 = N.make_env_cfg(4, 3, 0)
    assert lib.dgppo_env_num_nodes(C.byref(cfg1)) == 7 and lib.dgppo_env_num_edges(C.byref(cfg1)) == 12


def test_bad_arguments_are_rejected_before_launch():
    from dgppo_amd import _native as N
    lib = N.lib()
    cfg = N.make_env_cfg(0, 8, 3)
    cfg.kind = 17
    lib.dgppo_env_num_nodes.restype = C.c_int32
    assert lib.dgppo_env_num_nodes(C.byref(cfg)) == -1
    assert b"unknown env kind" in lib.dgppo_last_error()
    cfg = N.make_env_cfg(0, 8, 3)
    rc = lib.dgppo_env_step(C.byref(cfg), None, None, None, None, None, None, None, None, None, None, None, None,
                            C.c_int32(4), None)
    assert rc == -1 and b"NULL" in lib.dgppo_last_error()


def test_product_path_refuses_cpu_tensors():
    import torch
    from dgppo_amd import _native as N
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        N.ptr(torch.zeros(3))


def test_comm_entry_points_validate_without_a_gpu():
    """C1 (include/dgppo_hip.h): bad handles / NULL arguments are refused on the host, before RCCL is even looked up"""
    from dgppo_amd import _native as N
    lib = N.lib()
    assert lib.dgppo_comm_allreduce_sum_f32(None, None, C.c_int64(4), None) == -1
    assert b"not a communicator" in lib.dgppo_last_error()
    bogus = (C.c_uint8 * 64)()
    assert lib.dgppo_comm_allreduce_sum_f32(bogus, None, C.c_int64(4), None) == -1
    assert lib.dgppo_comm_destroy(None) == 0
    assert lib.dgppo_comm_destroy(bogus) == -1
    assert lib.dgppo_comm_unique_id(None) == -1
    handle = C.c_void_p()
    assert lib.dgppo_comm_init(None, 0, 1, C.byref(handle)) == -1
    idb = (C.c_uint8 * 128)()
    assert lib.dgppo_comm_init(idb, 3, 2, C.byref(handle)) == -1 and b"outside world" in lib.dgppo_last_error()
