import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, ops_nn as K
dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
G = 16384; Rg = G * 8
for (F, Kp) in ((8, 48), (32, 144)):
    qt = torch.randn(Rg, 3 * F, device=dev); Xa = torch.randn(Rg, F, device=dev); Xo = torch.randn(G * 72, F, device=dev)
    ef = torch.randn(Rg, 24, 4, device=dev); em = (torch.rand(Rg, 24, device=dev) > 0.3).float(); em[:, 8:16] = 1.0
    z = torch.empty(Rg, Kp, device=dev); at = torch.empty(Rg, 24, 3, device=dev)
    for _ in range(3):
        K.attn_fwd(cfg, F, 3, Kp, qt, Xa, Xo, ef, em, z, at, G)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    N.lib().dgppo_debug_stamps_attn(out)
    st = np.array(out[:7], dtype=np.int64)
    print("F", F, "fwd phases (wave kernel: [issue frag loads, L mfma+write, bz issue+copies, softmax, Z mfma+stores, -])", (st[1:] - st[:-1]).tolist(), "total", int(st[6] - st[0]))
