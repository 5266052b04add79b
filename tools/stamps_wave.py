"""Developer tool: per-phase s_memtime stamps of the wave-per-env LiDAR kernel (csrc/env_wave.hip).
    make -C dgppo_amd/csrc stamps && DGPPO_HIP_LIB=dgppo_amd/csrc/libdgppo_hip_stamps.so python tools/stamps_wave.py
Stamps are taken by wave 0 of workgroup 0 on its first environment, at launch sizes 1 (alone on the chip), 4096 and 16384."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgppo_amd import _native as N, ops_env as OE
dev = torch.device("cuda:0")
cfg = N.make_env_cfg(0, 8, 3)
names = {0: "P0 stage", 1: "P1a dyn/seg/circ", 2: "P1b far/as", 3: "P1c cost terms", 4: "P1d reward + early outputs",
         5: "P2a candidate list", 6: "P2b dense segment tests", 10: "P2c bad-ray check", 11: "P3 top-k (all 4 steps)",
         7: "P4 late compact", 8: "P5 late graph", 9: "env end"}
order = [0, 1, 2, 3, 4, 5, 6, 10, 11, 7, 8, 9]
for B in (1, 4096, 16384):
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=dev) * 7919
    agent = torch.empty(B, 8, 4, device=dev); goal = torch.empty(B, 8, 4, device=dev); obst = torch.empty(B, 3, 16, device=dev)
    OE.env_reset(cfg, seeds, agent, goal, obst)
    rc, rs = OE.ray_tables(32, dev)
    hits = torch.empty(B, 8, 8, 2, device=dev)
    OE.env_step(cfg, agent, None, goal, obst, None, rc, rs, None, hits, None, None, None)
    act = torch.empty(B, 8, 2, device=dev).uniform_(-1, 1)
    nx = torch.empty_like(agent); nh = torch.empty_like(hits); rew = torch.empty(B, device=dev); cost = torch.empty(B, 8, 2, device=dev)
    g = OE.alloc_graph(cfg, B, dev)
    for _ in range(3):
        OE.env_step(cfg, agent, act, goal, obst, hits, rc, rs, nx, nh, rew, cost, g)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 64)()
    N.lib().dgppo_debug_wave_stamps(out)
    st = np.array(out[:], dtype=np.int64)
    rt = max(int(st[41] - st[40]), 1)
    print(f"B={B}: total {int(st[9] - st[0])} ticks in {rt * 10} ns of s_memrealtime -> shader clock ~ {int(st[9] - st[0]) / (rt * 10e-9) / 1e9:.2f} GHz")
    for a_, b_ in zip(order[:-1], order[1:]):
        print(f"   {names[a_]:28s} {int(st[b_] - st[a_]):7d}")
    for it in range(4):
        b0 = 12 + it * 3
        prev = st[11] if it == 0 else st[12 + (it - 1) * 3 + 2]
        print(f"   step {it}: alpha read + key {int(st[b0 + 1] - prev):6d}  top-k + hit write {int(st[b0 + 2] - st[b0 + 1]):6d}")
    if B >= 4096:
        sp = (C.c_ulonglong * (3 * 8192))()
        N.lib().dgppo_debug_wave_spans(sp)
        w = np.array(sp[:], dtype=np.int64).reshape(8192, 3)[:min(B, 8192)]
        w = w[w[:, 2] > 0]
        t0 = w[:, 0].min()
        ent, st_, en = w[:, 0] - t0, w[:, 1] - t0, w[:, 2] - t0
        q = lambda x: [int(v) for v in np.percentile(x, [0, 10, 50, 90, 99, 100])]
        print(f"   {len(w)} waves; percentiles [0,10,50,90,99,100] in ticks relative to the first wave's entry:")
        print("     kernel entry      ", q(ent))
        print("     first env start   ", q(st_), " (prologue = start - entry:", q(st_ - ent), ")")
        print("     last env end      ", q(en))
        print("     wave lifetime     ", q(en - ent), " per-wave env time (end - start):", q(en - st_))
