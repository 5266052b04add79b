"""Developer tool: GPU time of the network building blocks at training sizes (minibatch of 16384 LidarSpread n=8 graphs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, ops_nn as K

dev = torch.device("cuda:0")
R = 131072          # agent rows of a minibatch (16384 graphs x 8)
Ro = 16384 * 72


def timeit(name, fn, iters=20, flops=None, bytes_=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    extra = ""
    if flops:
        extra += f"  {flops / us / 1e6:7.1f} TFLOP/s"
    if bytes_:
        extra += f"  {bytes_ / us / 1e3:7.0f} GB/s"
    print(f"{name:44s} {us:9.1f} us{extra}", flush=True)


def dense_case(M, Kd, Nd, trans=False, act=0):
    X = torch.randn(M, Kd, device=dev)
    W = torch.randn(Nd, Kd, device=dev) if trans else torch.randn(Kd, Nd, device=dev)
    b = torch.randn(Nd, device=dev)
    Y = torch.empty(M, Nd, device=dev)
    timeit(f"dense_fwd M={M} K={Kd} N={Nd} trans={int(trans)}", lambda: K.dense_fwd(X, W, b, Y, act=act, trans_w=trans),
           flops=2.0 * M * Kd * Nd, bytes_=4.0 * M * (Kd + Nd))
    dW = torch.zeros(Kd, Nd, device=dev); db = torch.zeros(Nd, device=dev)
    if not trans:
        timeit(f"dense_bwd_w M={M} K={Kd} N={Nd}", lambda: K.dense_bwd_w(X, Y, dW, db), flops=2.0 * M * Kd * Nd,
               bytes_=4.0 * M * (Kd + Nd))


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "dense"):
    for (M, Kd, Nd, tr) in [(R, 144, 64, False), (R, 64, 64, False), (R, 64, 192, False), (R, 48, 32, False), (R, 32, 96, False),
                            (R, 8, 24, False), (Ro, 8, 32, False), (R, 64, 144, True), (R, 192, 64, True), (R, 64, 64, True),
                            (R, 64, 4, False), (R, 4, 64, True), (R, 1, 64, True), (32768, 8, 24, False), (32768, 144, 64, False), (32768, 64, 64, False), (32768, 64, 192, False)]:
        dense_case(M, Kd, Nd, tr)
if which in ("all", "dense", "densemask"):
    # the ReLU-masked input gradients of the update: dX = (mask > 0) ? dY @ W^T : 0
    for (M, Kd, Nd) in [(R, 64, 64), (R, 192, 64), (R, 64, 144), (Ro, 32, 32)]:
        X = torch.randn(M, Kd, device=dev); W = torch.randn(Nd, Kd, device=dev); Y = torch.empty(M, Nd, device=dev)
        mk = torch.randn(M, Nd, device=dev)
        timeit(f"dense_fwd masked M={M} K={Kd} N={Nd} trans=1", lambda: K.dense_fwd(X, W, None, Y, trans_w=True, relu_mask=mk),
               flops=2.0 * M * Kd * Nd, bytes_=4.0 * M * (Kd + 2 * Nd))
if which in ("all", "elem"):
    x = torch.randn(R, 64, device=dev); g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev)
    y = torch.empty_like(x); st = torch.empty(R, 2, device=dev); dx = torch.empty_like(x)
    dg = torch.zeros(64, device=dev); dbb = torch.zeros(64, device=dev)
    timeit("ln_relu_fwd R x 64", lambda: K.ln_relu_fwd(x, g, b, y, st), bytes_=4.0 * R * 128)
    timeit("ln_relu_bwd R x 64", lambda: K.ln_relu_bwd(x, y, st, g, x, dx, dg, dbb), bytes_=4.0 * R * 256)
    timeit("relu_bwd R x 64", lambda: K.relu_bwd(dx, y), bytes_=4.0 * R * 128)
    gi = torch.randn(R, 192, device=dev); Wh = torch.randn(64, 192, device=dev) * 0.1; bh = torch.zeros(64, device=dev)
    hs = torch.empty(R, 64, device=dev); hp = torch.empty(R, 64, device=dev); gt = torch.empty(R, 256, device=dev)
    timeit("gru_fwd n_seq=8192 T=16", lambda: K.gru_fwd(gi, Wh, bh, None, hs, hp, gt, R // 16, 16, 8))
    dgi = torch.empty(R, 192, device=dev); dgh = torch.empty(R, 192, device=dev)
    timeit("gru_bwd n_seq=8192 T=16", lambda: K.gru_bwd(hs, Wh, hp, gt, dgi, dgh, R // 16, 16, 8))
    h0 = torch.randn(32768, 64, device=dev); gi1 = torch.randn(32768, 192, device=dev); hs1 = torch.empty(32768, 64, device=dev)
    timeit("gru_fwd rollout n_seq=32768 T=1", lambda: K.gru_fwd(gi1, Wh, bh, h0, hs1, None, None, 32768, 1, 8))
if which in ("all", "attn"):
    cfg = N.make_env_cfg(0, 8, 3)
    for G in (16384, 4096):
        Rg = G * 8
        for (F, Kp) in ((8, 48), (32, 144)):
            qt = torch.randn(Rg, 3 * F, device=dev); Xa = torch.randn(Rg, F, device=dev); Xo = torch.randn(G * 72, F, device=dev)
            ef = torch.randn(Rg, 24, 4, device=dev); em = (torch.rand(Rg, 24, device=dev) > 0.3).float()
            em[:, 8:16] = 1.0
            z = torch.empty(Rg, Kp, device=dev); at = torch.empty(Rg, 24, 3, device=dev)
            timeit(f"attn_fwd G={G} F={F}", lambda: K.attn_fwd(cfg, F, 3, Kp, qt, Xa, Xo, ef, em, z, at, G))
            dq = torch.empty_like(qt); dXa = torch.empty_like(Xa); dXo = torch.empty_like(Xo)
            timeit(f"attn_bwd G={G} F={F}", lambda: K.attn_bwd(cfg, F, 3, Kp, z, at, qt, Xa, Xo, ef, dq, dXa, dXo, G))
            if F == 32:   # how much of the wide backward is the input gradient (dXs: 160 of its 320 MFMAs, 10 KB of stores per graph)
                timeit(f"attn_bwd G={G} F={F} (dqt only)", lambda: K.attn_bwd(cfg, F, 3, Kp, z, at, qt, Xa, Xo, ef, dq, None, None, G))
            if F == 32 and K.attn_xo_supported(cfg, F, 3, Kp):   # other-node rows recomputed from the raw features in the kernel
                raw = torch.randn(Xo.shape[0], 8, device=dev); Wo = torch.randn(8, 32, device=dev) * 0.5; bo = torch.randn(32, device=dev) * 0.3
                timeit(f"attn_fwd_xo G={G} F={F}", lambda: K.attn_fwd_xo(cfg, F, 3, Kp, qt, Xa, raw, Wo, bo, ef, em, z, at, G))
                timeit(f"attn_fwd_xo G={G} F={F} (inference)", lambda: K.attn_fwd_xo(cfg, F, 3, Kp, qt, Xa, raw, Wo, bo, ef, em, z, None, G))
                timeit(f"attn_bwd_xo G={G} F={F}", lambda: K.attn_bwd_xo(cfg, F, 3, Kp, z, at, qt, Xa, raw, Wo, bo, ef, dq, dXa, dXo, G, relu_xo=True))
                dWo = torch.zeros(8, 32, device=dev); dbo = torch.zeros(32, device=dev); ws = torch.empty(K.attn_xo_workspace_floats(G), device=dev)
                timeit(f"attn_bwd_xo_dw G={G} F={F} (+ slab reduce)", lambda: K.attn_bwd_xo_dw(cfg, F, 3, Kp, z, at, qt, Xa, raw, Wo, bo, ef, dq, dXa, dWo, dbo, ws, G))
            if F == 8:    # the first layer needs no input gradient
                timeit(f"attn_bwd G={G} F={F} (dqt only)", lambda: K.attn_bwd(cfg, F, 3, Kp, z, at, qt, Xa, Xo, ef, dq, None, None, G))
if which in ("all", "fused"):
    for M, train in ((32768, False), (131072, False), (131072, True)):
        X = torch.randn(M, 64, device=dev)
        P = [torch.randn(*s, device=dev) * 0.1 for s in ((64, 64), (64,), (64,), (64,), (64, 64), (64,), (64,), (64,), (64, 192), (192,))]
        gi = torch.empty(M, 192, device=dev)
        sv = tuple(torch.empty(M, w, device=dev) for w in (64, 64, 2, 64, 64, 2)) if train else None
        timeit(f"mlp_gi_fwd M={M} train={train}", lambda: K.mlp_gi_fwd(X, *P, gi, sv))
if which in ("all", "fusedbwd"):
    for M in (131072, 16384):
        P = [torch.randn(*s_, device=dev) * 0.1 for s_ in ((64, 192), (64, 64), (64, 64), (64,), (64,))]
        dgi = torch.randn(M, 192, device=dev)
        A = [torch.randn(M, w, device=dev) for w in (64, 64, 2, 64, 64, 2)]
        mask = torch.randn(M, 64, device=dev)
        o = [torch.empty(M, 64, device=dev) for _ in range(3)]
        pg = [torch.zeros(64, device=dev) for _ in range(4)]
        timeit(f"mlp_gi_bwd fused M={M}", lambda: K.mlp_gi_bwd(dgi, P[0], P[1], P[2], P[3], P[4], A[3], A[4], A[5], A[0], A[1], A[2], mask,
                                                               o[0], o[1], o[2], *pg))
        dy = torch.empty(M, 64, device=dev)

        def unfused():
            K.dense_fwd(dgi, P[0], None, dy, trans_w=True)
            K.ln_relu_bwd(A[3], A[4], A[5], P[3], dy, o[0], pg[0], pg[1])
            K.dense_fwd(o[0], P[1], None, dy, trans_w=True)
            K.ln_relu_bwd(A[0], A[1], A[2], P[4], dy, o[1], pg[2], pg[3])
            K.dense_fwd(o[1], P[2], None, o[2], trans_w=True, relu_mask=mask)
        timeit(f"mlp bwd chain unfused (5 launches) M={M}", unfused)
if which in ("all", "tail"):
    for M, two, train in ((32768, True, False), (131072, False, True), (524288, False, False)):
        gi = torch.randn(M, 192, device=dev); h0 = torch.randn(M, 64, device=dev)
        Wh = torch.randn(64, 192, device=dev) * 0.1; bhn = torch.zeros(64, device=dev)
        n_out = 4 if two else 2
        W1 = torch.randn(64, 64 if two else n_out, device=dev) * 0.1; b1 = torch.zeros(64 if two else n_out, device=dev)
        W2 = torch.randn(64, n_out, device=dev) * 0.1 if two else None; b2 = torch.zeros(n_out, device=dev) if two else None
        hs = torch.empty(M, 64, device=dev); out = torch.empty(M, n_out, device=dev)
        hp = torch.empty(M, 64, device=dev) if train else None; gt = torch.empty(M, 256, device=dev) if train else None
        u = torch.empty(M, 64, device=dev)
        timeit(f"gru1_head_fwd fused M={M} two={two} train={train}",
               lambda: K.gru1_head_fwd(gi, Wh, bhn, h0, W1, b1, W2, b2, hs, hp, gt, u if (two and train) else None, out))

        def unfused():
            K.gru_fwd(gi, Wh, bhn, h0, hs, hp, gt, M, 1, 8)
            if two:
                K.dense_fwd(hs, W1, b1, u); K.dense_fwd(u, W2, b2, out)
            else:
                K.dense_fwd(hs, W1, b1, out)
        timeit(f"gru_fwd + dense unfused M={M} two={two} train={train}", unfused)
