"""TEST INFRASTRUCTURE ONLY.  Child process of tests/test_sanitizer_cpu.py, started with LD_PRELOAD=libasan.so: drives the
sanitizer build of the thread-independent kernels (dgppo_amd/csrc/libdgppo_cpu_asan.so: g++ -fsanitize=address,undefined over
csrc/env_reset.hip + capi.hip with the serial host stand-in for the HIP launch syntax) on host buffers.  numpy + ctypes only.

    python run_asan_child.py <lib> <job.npz> <out.npz>
job: cfg_bytes [n_cfg, sizeof(dgppo_env_cfg)] uint8, shapes [n_cfg, 5] = (n_agents, n_goals, n_obs, obst_stride, state_dim),
     seeds [B] uint64; noise: (seed, rows, row_len, global_row_len, col_offset)."""
import ctypes as C
import sys

import numpy as np

lib = C.CDLL(sys.argv[1])
job = np.load(sys.argv[2])
lib.dgppo_last_error.restype = C.c_char_p
P = lambda a: a.ctypes.data_as(C.c_void_p)
out = {}
seeds = np.ascontiguousarray(job["seeds"], dtype=np.uint64)
B = len(seeds)
for k, (cfgb, shp) in enumerate(zip(job["cfg_bytes"], job["shapes"])):
    n, ng, no, ostride, sd = (int(x) for x in shp)
    cfg = (C.c_uint8 * len(cfgb)).from_buffer_copy(bytes(cfgb))
    # exactly-sized buffers: AddressSanitizer flags the first byte written past a row
    agent = np.full((B, n, sd), np.nan, np.float32); goal = np.full((B, ng, sd), np.nan, np.float32)
    obst = np.full((B, max(no, 1), ostride), np.nan, np.float32)[:, :no].copy()
    nfail = np.zeros(1, np.int32)
    rc = lib.dgppo_env_reset_checked(C.byref(cfg), P(seeds), P(agent), P(goal), P(obst) if no > 0 else None, P(nfail), C.c_int32(B), None)
    out[f"rc{k}"] = np.int32(rc)
    out[f"err{k}"] = np.frombuffer(lib.dgppo_last_error(), dtype=np.uint8) if rc else np.zeros(0, np.uint8)
    out[f"agent{k}"], out[f"goal{k}"], out[f"obst{k}"], out[f"nfail{k}"] = agent, goal, obst, nfail
seed, rows, row_len, glen, off = (int(x) for x in job["noise"])
flat = np.full(rows * glen, np.nan, np.float32)
assert lib.dgppo_randn(C.c_uint64(seed), C.c_uint64(0), P(flat), C.c_int64(flat.size), None) == 0
win = np.full((rows, row_len), np.nan, np.float32)
assert lib.dgppo_randn_rows(C.c_uint64(seed), P(win), C.c_int64(rows), C.c_int64(row_len), C.c_int64(glen), C.c_int64(off), None) == 0
odd = np.full(7, np.nan, np.float32)                    # n_elem not a multiple of 4: the last Philox block is cut
assert lib.dgppo_randn(C.c_uint64(seed), C.c_uint64(3), P(odd), C.c_int64(7), None) == 0
out["flat"], out["win"], out["odd"] = flat, win, odd
np.savez(sys.argv[3], **out)
print("asan child ok")
