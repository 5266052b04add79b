#!/usr/bin/env python
"""Benchmark of the DGPPO hot path on MI355X.

One "step" = one full DGPPO training iteration on the workload of BASELINE.json (LidarSpread, n=8 agents, 3 obstacles,
4096 envs per GPU, T=128): stochastic rollout (collect) + deterministic rollout + value pre-passes + 2x GAE + advantage +
all PPO minibatch updates (Vl, Vh, policy; clip + Adam), i.e. exactly what `algo.collect` + `algo.update` do per
iteration (dgppo/trainer/trainer.py:131-137).  Counted work = B*T env-steps of the stochastic rollout per iteration
(BASELINE.md §3).  Synthetic random scenes, random-init networks.

    python bench.py --gpus N --steps K --warmup W
        N>1: one rank per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ...` the ranks come from the
        environment; a bare `python bench.py --gpus N` starts the N ranks itself (before touching the GPU).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (raycast+graph kernel vs the HBM roofline, timed with HIP events
inside this process), "cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample), "phases".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks this file prices against (MI355X_MICROARCH.md "Chip-level parameters"); printed in the JSON line as "peaks"
HBM_PEAK_GBS = 8000.0          # HBM3E spec; ~6300 GB/s is the measured float4-copy ceiling on this part
FP32_PEAK_TFLOPS = 157.3       # fp32 vector = fp32-input MFMA (v_mfma_f32_16x16x4_f32) dense peak


def log(msg):
    """progress to stderr (and gpurun_out/ when present) so a long run never looks hung."""
    line = f"[bench {time.strftime('%H:%M:%S')}] {msg}"
    print(line, file=sys.stderr, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "bench_progress.log"), "a") as f:
            f.write(line + "\n")


def host_threads():
    """CPU threads this process may really use: the scheduler affinity, cut down to the cgroup CPU quota when there is one
    (a 1-GPU box of the pool exposes all 256 host CPUs but grants a 16-CPU share)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    return max(1, min(aff, quota) if quota else aff), aff, quota


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--env", type=str, default="LidarSpread")
    p.add_argument("-n", "--num-agents", type=int, default=8)
    p.add_argument("--obs", type=int, default=3)
    p.add_argument("--n-env", type=int, default=4096, help="envs PER GPU (weak scaling)")
    p.add_argument("--batch-size", type=int, default=16384)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--stall-seconds", type=float, default=420.0,
                   help="self-launched ranks (--gpus N without torchrun): stop the job when no rank has written anything for this long")
    return p.parse_args()


def roofline_env_kernel(cfg, device, B, iters=50):
    """Raycast+graph kernel (dgppo_env_step with the materialised GraphsTuple) alone, timed with HIP events on the stream it
    is launched on.  Algorithmic bytes per env-step: SURVEY §8(d) B_api (+ the pre-step hit points the cost reads)."""
    from dgppo_amd import ops_env as OE
    n, sd, k = cfg.n_agents, cfg.state_dim, cfg.top_k
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=device) * 2654435761
    agent = torch.empty(B, n, sd, device=device); goal = torch.empty(B, n, sd, device=device)
    obst = torch.empty(B, cfg.n_obs, cfg.obst_stride, device=device)
    OE.env_reset(cfg, seeds, agent, goal, obst)
    rc, rs = OE.ray_tables(cfg.n_rays, device)
    hits = torch.empty(B, n, k, 2, device=device)
    OE.env_step(cfg, agent, None, goal, obst, None, rc, rs, None, hits, None, None, None)
    action = torch.empty(B, n, 2, device=device).uniform_(-1, 1)
    nx = torch.empty_like(agent); nh = torch.empty_like(hits)
    rew = torch.empty(B, device=device); cost = torch.empty(B, n, 2, device=device)
    g = OE.alloc_graph(cfg, B, device)
    N_, E_ = cfg.num_nodes, cfg.num_edges
    obs_bytes = 4 * 13 * cfg.n_obs
    R_ = 4 * (n * sd + 2 * n + n * sd) + obs_bytes
    W_api = 4 * (N_ * cfg.node_dim + 4 * E_ + N_ * sd + 2 * E_ + N_) + 8 + 4 + 4 * n * 2
    B_api = R_ + W_api                                                    # 9048 for LidarSpread n=8
    W_min = 4 * (n * sd + 2 * n * k + 1 + n * 2) + (n * (n + k) + 7) // 8
    B_min = R_ + W_min
    out = {}
    stream = torch.cuda.Stream(device)
    for name, graph, bytes_per in (("api", g, B_api), ("compact", None, B_min)):
        # The launches are captured into ONE HIP graph and replayed: with ~15-20 us of Python per ctypes call an eager loop
        # would time the host, not a ~20 us kernel.  HIP events on the stream the graph runs on bracket the replay.
        with torch.cuda.stream(stream):
            for _ in range(3):
                OE.env_step(cfg, agent, action, goal, obst, hits, rc, rs, nx, nh, rew, cost, graph)
            stream.synchronize()
            hg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hg, stream=stream, capture_error_mode="thread_local"):
                for _ in range(iters):
                    OE.env_step(cfg, agent, action, goal, obst, hits, rc, rs, nx, nh, rew, cost, graph)
            hg.replay()                                   # first replay uploads the graph
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            hg.replay()
            e1.record(stream)
            stream.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out[name] = dict(us_per_launch=us, bytes_per_env_step=bytes_per, gbs=bytes_per * B / us / 1e3,
                         env_steps_per_s=B / us * 1e6, launches_timed=iters)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget: float):
    """CPU restatement (JAX not installable offline): the oracle's full DGPPO iteration on a bounded sample of
    the workload — B = 256 envs with a 16-step horizon so that one iteration is ~30 s of CPU work: same
    topology (LidarSpread n=8 obs=3), same 128 envs per minibatch, rnn_step 16, two minibatches; every env-step costs
    what it costs at T = 128 except the O(T^2) GAE, which is negligible here.  Measured twice: all host threads and one
    thread.  No extrapolation: value = counted env-steps / wall time."""
    from oracle import train_ref
    B, T = 256, 16
    bs = 128 * T
    out = {}
    n_thr, affinity, quota = host_threads()
    for label, thr in (("all", n_thr), ("one", 1)):
        torch.set_num_threads(thr)
        t0 = time.time()
        train_ref.iteration("LidarSpread", 8, 3, B=B, T=T, batch_size=bs, seed=0)
        dt = time.time() - t0
        log(f"cpu_baseline[{label}]: one oracle iteration (B={B}, T={T}) took {dt:.1f} s on {thr} thread(s)")
        out[label] = (B * T / dt, thr, dt)
        if label == "all" and dt > 3.0 * seconds_budget:
            log("cpu_baseline: skipping the 1-thread run (the all-thread run already exceeded 3x the budget)")
            break
    v, thr, dt = out["all"]
    res = {"value": v, "unit": "env-steps/s", "cores": thr, "kind": "port", "cpu_model": cpu_model(),
           "host_cpus_visible": os.cpu_count(), "sched_affinity": affinity, "cgroup_cpu_quota": quota, "seconds": dt,
           "sample": f"1 full DGPPO iteration of the CPU oracle (numpy-fp32 env + torch-CPU per-edge networks + autograd), "
                     f"LidarSpread n=8 obs=3, {B} envs x {T} steps, batch {bs} (128 envs/minibatch), rnn_step 16; {dt:.1f} s; "
                     f"CPU restatement (JAX not installable offline)"}
    if "one" in out:
        res["one_thread"] = {"value": out["one"][0], "cores": 1, "seconds": out["one"][2]}
    return res


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no torchrun environment: the parent starts the N ranks itself — BEFORE it makes any GPU
    call (it never initialises HIP) — supervises them (dgppo_amd/launch.py: per-rank logs, stop everything on the first
    failure or on a stall) and relays rank 0's JSON line."""
    from dgppo_amd import launch
    return launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, launch.default_log_dir(ROOT),
                              stall_seconds=args.stall_seconds)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                  # nothing above touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    n_dev = torch.cuda.device_count()                # counting devices does not initialise HIP
    backend = os.environ.get("DGPPO_DIST_BACKEND", "rccl")
    if world > n_dev and backend != "gloo":
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s) visible.  RCCL needs one device per rank; "
                         f"DGPPO_DIST_BACKEND=gloo rehearses several ranks on one GPU.")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev_index = local_rank % max(n_dev, 1)           # ranks > devices only in gloo rehearsals
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    from dgppo_amd import _native as N, engine as EN, init, dist as D, ops_nn as K
    if world > 1 and rank == 0:
        log(f"multi-GPU diagnostics: world {world}, {n_dev} device(s) visible, data plane '{backend}', RCCL version "
            f"{D.rccl_version()}, NCCL_DEBUG={os.environ.get('NCCL_DEBUG')}, HSA_ENABLE_IPC_MODE_LEGACY="
            f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, rendezvous "
            f"{'file store' if os.environ.get('DGPPO_RDZV_FILE') else os.environ.get('MASTER_ADDR', '?') + ':' + os.environ.get('MASTER_PORT', '?')}")
    D.init_control_plane()
    if world > 1:
        log(f"rank {rank}: control plane (gloo) up")
    # data plane: ONE dgppo_comm_allreduce_sum_f32 (RCCL, C ABI) of the flat gradient buffer per minibatch step
    allreduce, close_comm = D.make_allreduce(world, backend)
    if allreduce is not None:                        # first collective (connection setup) outside any timed region
        log(f"rank {rank}: communicator up, first collective")
        allreduce(torch.zeros(1 << 16, device=device))
        torch.cuda.synchronize()
        # known-answer all-reduce: rank r contributes r + 1, every entry must come back as world (world + 1) / 2
        D.selfcheck_allreduce(allreduce, rank, world, device)
        log(f"rank {rank}: all-reduce self-check passed")
    fault_rank = os.environ.get("DGPPO_BENCH_FAULT_RANK")       # test hook: this rank dies after the communicator is up
    if fault_rank is not None and int(fault_rank) == rank:
        log(f"rank {rank}: DGPPO_BENCH_FAULT_RANK set — exiting with code 3 (supervision test)")
        os._exit(3)

    cfg = N.make_env_cfg(N.ENV_KINDS[args.env], args.num_agents, args.obs)
    T = 128
    hp = EN.Hyper(batch_size=args.batch_size, train_steps=1000)
    eng = EN.Engine(cfg, hp, device, T=T, allreduce=allreduce, world=world, rank=rank, use_graphs=True, multi_stream=True)
    eng.policy.load_tree(init.init_policy(0, cfg.node_dim, 2, hp.actor_gnn_layers))
    eng.Vl.load_tree(init.init_value(0, cfg.node_dim, 1, hp.Vl_gnn_layers, 2))
    eng.Vh.load_tree(init.init_value(0, cfg.node_dim, 2, hp.Vh_gnn_layers, 3))
    eng.set_entropy_noise(12345)
    B = args.n_env
    rng = np.random.default_rng(1000)                # the same permutation on every rank (shared seed)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    phases = {"collect+det_rollout": 0.0, "update": 0.0}

    def iteration(it: int, timed: bool):
        seeds = torch.from_numpy(D.shard_seeds(rank, B, it)).to(device)
        e = [ev() for _ in range(3)]
        e[0].record()
        # algo.collect and the det_rollout_fn inside algo.update: same parameters, independent -> two HIP streams
        ro, det = eng.rollout_pair(seeds, seeds ^ 0x5DEECE66D, noise_seed=it * 2 + 1)
        e[1].record()
        if it == 0:
            torch.cuda.synchronize(); log("first rollouts done")
        info = eng.update(ro, det, it, rng.permutation(B))                    # rest of algo.update (+ info sync)
        e[2].record()
        if timed:
            torch.cuda.synchronize()
            phases["collect+det_rollout"] += e[0].elapsed_time(e[1])
            phases["update"] += e[1].elapsed_time(e[2])
        return info

    log(f"rank {rank}/{world}: engine ready on {torch.cuda.get_device_name(device)} (backend {backend}), B={B} envs, "
        f"starting {args.warmup} warm-up iteration(s)")
    for w in range(args.warmup):
        iteration(w, False)
        torch.cuda.synchronize()
        log(f"warm-up {w} done")
    torch.cuda.synchronize()
    D.barrier(world)
    torch.cuda.synchronize()
    flops0 = K.FLOPS[0]
    t0 = time.perf_counter()
    info = None
    step_ms = []
    for k in range(args.steps):
        tk = time.perf_counter()
        info = iteration(args.warmup + k, True)      # ends with the host sync of the logged scalars
        step_ms.append((time.perf_counter() - tk) * 1e3)
        log(f"timed iteration {k} done ({time.perf_counter() - t0:.2f} s since start)")
    torch.cuda.synchronize()
    D.barrier(world)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops_per_step = (K.FLOPS[0] - flops0) / args.steps
    dt = D.max_over_ranks(dt, world)
    if rank != 0:
        close_comm()
        D.shutdown(world)
        return
    ms_per_step = dt * 1e3 / args.steps
    value = world * B * T * args.steps / dt
    log(f"training timed: {ms_per_step:.1f} ms/iteration -> {value:.0f} env-steps/s; timing the raycast+graph kernel")
    # the stochastic rollout alone (collect), timed after the training loop: env-steps/s of the rollout path by itself
    seeds = torch.from_numpy(D.shard_seeds(rank, B, 10_000)).to(device)
    eng.rollout(seeds, True, noise_seed=1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for r in range(3):
        eng.rollout(seeds, True, noise_seed=2 + r)
    torch.cuda.synchronize()
    rollout_only = B * T * 3 / (time.perf_counter() - t1)
    rl = roofline_env_kernel(cfg, device, B) if cfg.is_lidar and cfg.n_obs > 0 else None
    rl_big = roofline_env_kernel(cfg, device, 4 * B, iters=30) if rl is not None else None
    log(f"roofline kernel: {rl}")
    sm = np.sort(np.asarray(step_ms))
    out = {
        "metric": "env-steps/sec whole node, LidarSpread n=8 4096 envs, 1/2/4/8 GPUs",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.env} n={args.num_agents} obs={args.obs}, {B} envs/GPU x T=128, full DGPPO iteration "
                               f"(collect + det rollout + value pre-passes + GAE + {B // (args.batch_size // T)} minibatches "
                               f"of batch_size {args.batch_size} per GPU, rnn_step 16)",
                   "envs_per_gpu": B, "global_envs": world * B, "parallelism": f"dp{world}",
                   "collective": None if world == 1 else f"{backend}: 1 all-reduce(sum) of {eng.n_reduced} fp32 per minibatch"},
        "ms_per_step_stats": {"median": float(np.median(sm)), "min": float(sm[0]), "max": float(sm[-1]), "n": len(sm),
                              "note": "rank-0 wall time per iteration (each ends with the host sync of the logged scalars)"},
        "phases_ms_per_step": {k: v / args.steps for k, v in phases.items()},
        "update_host_ms_last_step": getattr(eng, "host_ms", None),
        "rollout_only_env_steps_per_s_per_gpu": rollout_only,
        "mfma": {"flops_executed_per_step": flops_per_step, "unit": "flop (fp32, 2 per multiply-add, unpadded operand shapes)",
                 "tflops": flops_per_step / (ms_per_step * 1e-3) / 1e12, "peak_tflops": FP32_PEAK_TFLOPS,
                 "util": flops_per_step / (ms_per_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                 "note": "network multiply-add work of one rank's iteration / its wall time (wall time includes rollouts, GAE and all "
                         "non-MFMA kernels; < 2 % of the counted flops run on the VALU: the first GNN layer's slot-sparse attention "
                         "and the K <= 16 dense kernels)"},
        "peaks_note": "fp32 VALU issue rates measured on this part: profiles/r03_valu_issue_rates.json",
        "peaks": {"hbm_GBps": HBM_PEAK_GBS, "fp32_TFLOPs": FP32_PEAK_TFLOPS, "source": "MI355X_MICROARCH.md (spec values)",
                  "device": torch.cuda.get_device_name(device)},
        "last_info": {k: info[k] for k in ("policy/loss", "Vl/loss", "Vh/loss_Vh", "eval/safe_data")},
    }
    nnc = os.path.join(ROOT, "profiles", "r03_nn_counters.json")
    if os.path.exists(nnc):
        # matrix-core busy fraction of the update phase from the committed counter passes (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
        # GRBM_GUI_ACTIVE ..., tools/collect_profiles.sh): measured offline, like roofline.traffic
        try:
            out["mfma"]["mfma_busy"] = json.load(open(nnc))["update_mfma_busy_frac"]
            out["mfma"]["mfma_busy_source"] = "profiles/r03_nn_counters.json: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) over the update phase (offline pass)"
        except Exception:
            pass
    if world > 1:
        out["multi_gpu"] = {"rccl_version": D.rccl_version(), "devices_visible": n_dev, "backend": backend,
                            "allreduce_selfcheck": "passed"}
    if rl is not None:
        traffic, tsrc = None, None
        for name in ("r03_env_step_traffic.json", "r02_env_step_traffic.json", "r01_env_step_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath) and B == 4096 and args.env == "LidarSpread" and args.num_agents == 8 and args.obs == 3:
                # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE), see profiles/README.md
                traffic, tsrc = json.load(open(tpath))["hbm_bytes_per_launch_api_fetch_x2"], f"profiles/{name} (rocprofv3 --pmc, offline)"
                break
        comp = dict(rl["compact"])
        flop_ray = 30.0 * cfg.n_agents * cfg.n_rays * cfg.n_obs * 4          # SURVEY §8d F_ray
        comp["valu_tflops"] = flop_ray * comp["env_steps_per_s"] / 1e12
        comp["valu_frac_of_fp32_peak"] = comp["valu_tflops"] / FP32_PEAK_TFLOPS
        out["roofline"] = {"bound": "hbm", "kernel": "lidar_wave_kernel (dynamics + raycast + top-k + reward/cost + GraphsTuple emit)",
                           "achieved": rl["api"]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rl["api"]["gbs"] / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": tsrc,
                           "algorithmic_bytes_per_launch": rl["api"]["bytes_per_env_step"] * B, "bytes_per_env_step": rl["api"]["bytes_per_env_step"],
                           "us_per_launch": rl["api"]["us_per_launch"], "kernel_env_steps_per_s": rl["api"]["env_steps_per_s"],
                           "at_4x_envs": {"envs": 4 * B, "us_per_launch": rl_big["api"]["us_per_launch"], "achieved": rl_big["api"]["gbs"],
                                          "frac": rl_big["api"]["gbs"] / HBM_PEAK_GBS},
                           "compact": comp}
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        except Exception as ex:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(ex)}
    print(json.dumps(out), flush=True)
    close_comm()
    D.shutdown(world)


if __name__ == "__main__":
    main()
