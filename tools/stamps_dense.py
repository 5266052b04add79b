import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, ops_nn as K
dev = torch.device("cuda:0")
for (M, Kd, Nd) in ((131072, 144, 64), (131072, 64, 64), (64, 144, 64), (131072, 64, 192)):
    X = torch.randn(M, Kd, device=dev); W = torch.randn(Kd, Nd, device=dev); b = torch.randn(Nd, device=dev); Y = torch.empty(M, Nd, device=dev)
    for _ in range(3):
        K.dense_fwd(X, W, b, Y)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    N.lib().dgppo_debug_stamps_dense(out)
    st = np.array(out[:4], dtype=np.int64)
    print("M", M, "K", Kd, "N", Nd, "phases [W+tile0, all tiles]:", (st[1:3] - st[0:2]).tolist())
