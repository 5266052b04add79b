"""Developer tool: per-phase s_memtime stamps of attn_fwd_bd_kernel (wave 0 of workgroup 0, under a full 16 384-graph launch).
Build: hipcc ... -DDGPPO_STAMPS -c nn_graph.hip, link as libdgppo_hip_attnstamps.so; run with DGPPO_HIP_LIB pointing at it."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, ops_nn as K
dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
F, H, Kp = 32, 3, 144
names = ["start", "loads issued", "LDS images written", "logits done", "softmax done", "Zx done + stores", "end"]
for G in (16384, 4096):
    R, Ro = G * 8, G * 72
    qt = torch.randn(R, 96, device=dev); Xa = torch.randn(R, 32, device=dev); Xo = torch.randn(Ro, 32, device=dev)
    raw = torch.randn(Ro, 8, device=dev); Wo = torch.randn(8, 32, device=dev); bo = torch.randn(32, device=dev)
    ef = torch.randn(R, 24, 4, device=dev); em = torch.ones(R, 24, device=dev)
    z = torch.empty(R, Kp, device=dev); at = torch.empty(R, 24, 3, device=dev)
    for label, fn in (("materialised", lambda: K.attn_fwd(cfg, F, H, Kp, qt, Xa, Xo, ef, em, z, at, G)),
                      ("recomputed", lambda: K.attn_fwd_xo(cfg, F, H, Kp, qt, Xa, raw, Wo, bo, ef, em, z, at, G))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        out = (C.c_ulonglong * 32)()
        N.lib().dgppo_debug_stamps_attn(out)
        t = [out[i] for i in range(7)]
        print(f"G={G} {label}: " + ", ".join(f"{names[i + 1]} +{t[i + 1] - t[i]}" for i in range(6)) + f"  | total {t[6] - t[0]} shader cycles (s_memtime)")
