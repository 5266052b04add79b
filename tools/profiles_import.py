"""Developer tool: turn the raw rocprofv3 output of `tools/collect_profiles.sh <tag>` (gpurun_out/profiles_<tag>/) into the
small, tracked summaries under profiles/:

    python tools/profiles_import.py r02

  <tag>_env_step_kernel_stats.csv            rocprofv3 --stats table of tools/bench_env.py (SIZES=4096,16384)
  <tag>_env_step_kernel_trace_summary.json   per (kernel instantiation, grid) launch statistics from the kernel trace
  <tag>_env_step_pmc_{FETCH,WRITE}_SIZE.csv  the raycast+graph kernel's rows of the two HBM counter passes
  <tag>_env_step_traffic.json                HBM bytes per launch (units / gfx950 FETCH_SIZE x2 correction: MI355X_MICROARCH.md)
  <tag>_env_step_insts.json                  SQ instruction counters per wave / per env
  <tag>_bench_kernel_stats.csv, <tag>_update_kernel_stats.csv, <tag>_bench_line.json, <tag>_env_wave_stamps.txt

The materialised-GraphsTuple ("api") and compact launches are different template instantiations of lidar_wave_kernel
(last template argument), so they are told apart by kernel name; launch sizes by the dispatch's grid."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = f"gpurun_out/profiles_{TAG}"
DST = "profiles"
KERNEL = "lidar_wave_kernel"


def find(sub, suffix):
    # (gpurun merges a call's output into the local directory without deleting older runs: take the newest file)
    hits = sorted(glob.glob(f"{SRC}/{sub}/**/*{suffix}", recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


def is_graph(name):
    """lidar_wave_kernel<SD, SPREAD, NA, NO, WPB, MODE, GRAPH>: GRAPH is the last template argument."""
    args = name[name.index("<") + 1:name.rindex(">")].replace(" ", "").split(",")
    return args[-1] in ("true", "1")


def mode_of(name):
    args = name[name.index("<") + 1:name.rindex(">")].replace(" ", "").split(",")
    return args[-2]


def trace_summary():
    path = find("env_kt", "kernel_trace.csv")
    if not path:
        return None
    rows = [r for r in csv.DictReader(open(path)) if KERNEL in r["Kernel_Name"]]
    gs = "Grid_Size_X" if "Grid_Size_X" in rows[0] else "Grid_Size"
    ws = "Workgroup_Size_X" if "Workgroup_Size_X" in rows[0] else "Workgroup_Size"
    groups = {}
    for r in rows:
        key = (r["Kernel_Name"], int(r[gs]) // int(r[ws]))
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = []
    for (name, blocks), d in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        if len(d) < 50:
            continue                                   # parity / warm-up launches
        d = d[-100:] if len(d) >= 100 else d
        out.append({"kernel": name, "workgroups": blocks, "graph_materialised": is_graph(name), "mode": mode_of(name),
                    "launches": len(d), "avg_us": statistics.mean(d), "median_us": statistics.median(d),
                    "min_us": min(d), "max_us": max(d)})
    res = {"command": "SIZES=4096,16384 rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_env.py",
           "note": "persistent grid: workgroups = min(ceil(envs / waves-per-block), resident blocks); per size bench_env "
                   "launches 3 + 100 (capture) + 2 x 100 (replays) per variant, the last 100 are summarised",
           "groups": out}
    json.dump(res, open(f"{DST}/{TAG}_env_step_kernel_trace_summary.json", "w"), indent=1)
    st = find("env_kt", "kernel_stats.csv")
    if st:
        shutil.copy(st, f"{DST}/{TAG}_env_step_kernel_stats.csv")
    return res


def pmc_rows(sub):
    path = find(sub, "counter_collection.csv")
    if not path:
        return []
    return [r for r in csv.DictReader(open(path)) if KERNEL in r["Kernel_Name"]]


def write_rows(rows, path):
    if not rows:
        return
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def big_launches(rows, counter):
    """values of `counter` for the 4096-env launches (the largest grid in the file), split api / compact by kernel name."""
    rows = [r for r in rows if r["Counter_Name"] == counter]
    if not rows:
        return [], []
    g = max(int(r["Grid_Size"]) for r in rows)
    rows = [r for r in rows if int(r["Grid_Size"]) == g]
    api = [float(r["Counter_Value"]) for r in rows if is_graph(r["Kernel_Name"]) and mode_of(r["Kernel_Name"]) == "0"]
    cmp_ = [float(r["Counter_Value"]) for r in rows if not is_graph(r["Kernel_Name"]) and mode_of(r["Kernel_Name"]) == "0"]
    return api, cmp_


def traffic():
    fr, wr = pmc_rows("env_fetch"), pmc_rows("env_write")
    if not fr or not wr:
        return None
    write_rows(fr, f"{DST}/{TAG}_env_step_pmc_FETCH_SIZE.csv")
    write_rows(wr, f"{DST}/{TAG}_env_step_pmc_WRITE_SIZE.csv")
    f_api, f_cmp = big_launches(fr, "FETCH_SIZE")
    w_api, w_cmp = big_launches(wr, "WRITE_SIZE")
    med = lambda v: statistics.median(v) if v else None
    envs = 4096
    d = {"FETCH_SIZE_KB_api": med(f_api), "FETCH_SIZE_KB_compact": med(f_cmp),
         "WRITE_SIZE_KB_api": med(w_api), "WRITE_SIZE_KB_compact": med(w_cmp),
         "launches_api": len(f_api), "launches_compact": len(f_cmp), "envs_per_launch": envs,
         "hbm_bytes_per_launch_api_raw": (med(f_api) + med(w_api)) * 1024,
         "hbm_bytes_per_launch_api_fetch_x2": (2 * med(f_api) + med(w_api)) * 1024,
         "hbm_bytes_per_launch_compact_fetch_x2": (2 * med(f_cmp) + med(w_cmp)) * 1024 if f_cmp and w_cmp else None,
         "algorithmic_bytes_per_launch_api": 9048 * envs,
         "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over SIZES=4096 tools/bench_env.py "
                 "(kernel lidar_wave_kernel<4,true,8,3,...>, medians over the 4096-env launches; KB units). WRITE_SIZE is "
                 "exact for streaming stores, FETCH_SIZE under-reports coalesced reads by 2x on gfx950 (guide) and is "
                 "uncalibrated for this kernel's one-float4-per-lane loads."}
    json.dump(d, open(f"{DST}/{TAG}_env_step_traffic.json", "w"), indent=1)
    return d


def insts():
    """SQ instruction / wait counters of the raycast+graph kernel per launch size (SIZES=4096,16384: the two largest grids
    that were launched >= 50 times, in that order), for the materialised and the compact variant"""
    res = {}
    for sub in ("env_insts", "env_insts2"):
        rows = pmc_rows(sub)
        if not rows:
            continue
        for variant, want in (("api", True), ("compact", False)):
            sel = [r for r in rows if is_graph(r["Kernel_Name"]) == want and mode_of(r["Kernel_Name"]) == "0"]
            by_grid = {}
            for r in sel:
                by_grid.setdefault(int(r["Grid_Size"]), []).append(r)
            n_ctr = len({r["Counter_Name"] for r in sel}) or 1
            grids = sorted(g for g, v in by_grid.items() if len(v) >= 50 * n_ctr)[-2:]
            for g, envs in zip(grids, (4096, 16384)[-len(grids):]):
                c = {}
                for r in by_grid[g]:
                    c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                c = {k: statistics.median(v) for k, v in c.items()}
                d = res.setdefault(variant, {}).setdefault(str(envs), {"envs_per_launch": envs, "grid_threads": g, "counters_per_launch": {}})
                d["counters_per_launch"].update(c)
    for variant in res:
        for d in res[variant].values():
            c, envs = d["counters_per_launch"], d["envs_per_launch"]
            d["waves_per_launch"] = c.get("SQ_WAVES")
            d["per_env"] = {k: v / envs for k, v in c.items() if k.startswith("SQ_INSTS")}
            if "SQ_WAVE_CYCLES" in c:        # quad-cycles -> cycles of wave lifetime per env, and where they go
                d["wave_cycles_per_env"] = 4.0 * c["SQ_WAVE_CYCLES"] / envs
                d["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                d["wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                d["active_valu_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
    if not res:
        return None
    res["note"] = ("two rocprofv3 --pmc passes (8 SQ counters each) over SIZES=4096,16384 tools/bench_env.py; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_* count quad-cycles; persistent waves walk several envs each at 16 384 envs, so the per-env figures are "
                   "the comparable ones (round 2: 1 698 VALU + 525 SALU + 170 LDS per env; round 1: 3 322 + 1 782 + 300)")
    json.dump(res, open(f"{DST}/{TAG}_env_step_insts.json", "w"), indent=1)
    return res


def _short(name):
    """kernel name without the argument list"""
    return name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").strip()


def nn_counters():
    """Per update-phase kernel (tools/prof_update_only.py, MS=0 REPS=1: value pre-passes + 32 minibatches, single stream):
    calls and average duration from the kernel trace, HBM bytes per call from the FETCH_SIZE / WRITE_SIZE passes (KB units;
    FETCH_SIZE doubled: gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "HBM"), matrix-core busy cycles and
    MFMA instruction counts, LDS bank conflicts — every counter from its own `--pmc`-only pass."""
    kt = find("update_kt", "kernel_trace.csv")
    if not kt:
        return None
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur.setdefault(_short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ctr = {}
    for sub in ("nn_fetch", "nn_write", "nn_mfma", "nn_lds"):
        path = find(sub, "counter_collection.csv")
        if not path:
            continue
        for r in csv.DictReader(open(path)):
            ctr.setdefault(_short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    total_us = sum(sum(v) for v in dur.values())
    out = []
    for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if sum(d) < 0.004 * total_us:
            continue
        c = {k: sum(v) / len(v) for k, v in ctr.get(name, {}).items()}          # per call
        row = {"kernel": name, "calls": len(d), "avg_us": statistics.mean(d), "total_ms": sum(d) / 1e3,
               "share_of_kernel_time": sum(d) / total_us}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd, wr = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
            row.update({"hbm_read_MB_per_call_fetch_x2": rd / 1e6, "hbm_write_MB_per_call": wr / 1e6,
                        "hbm_GBps_at_avg_duration": (rd + wr) / statistics.mean(d) / 1e3,
                        "hbm_frac_of_8TBps": (rd + wr) / statistics.mean(d) / 1e3 / 8000.0})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # busy cycles summed over the SIMDs the counter samples; GRBM_GUI_ACTIVE is summed over the 8 XCDs (guide: DVFS)
            act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            row.update({"mfma_busy_cycles_per_call": c["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_f32_insts_per_call": c.get("SQ_INSTS_VALU_MFMA_F32"),
                        "mfma_mops_f32_per_call": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"), "sq_busy_cycles_per_call": c.get("SQ_BUSY_CYCLES"),
                        "gpu_active_cycles_per_call": act or None,
                        "mfma_busy_frac": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * act)) if act else None,
                        "valu_insts_per_call": c.get("SQ_INSTS_VALU"), "waves_per_call": c.get("SQ_WAVES")})
        if "SQ_LDS_BANK_CONFLICT" in c:
            row.update({"lds_bank_conflict_cycles_per_call": c["SQ_LDS_BANK_CONFLICT"], "lds_idx_active_per_call": c.get("SQ_LDS_IDX_ACTIVE"),
                        "lds_conflict_frac": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
                        "wait_any_frac": (c["SQ_WAIT_ANY"] / (c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]))
                        if all(k in c for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")) else None,
                        "wait_inst_frac": (c["SQ_WAIT_INST_ANY"] / (c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]))
                        if all(k in c for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")) else None})
        out.append(row)
    mfma_busy = sum(r.get("mfma_busy_cycles_per_call", 0.0) * r["calls"] for r in out)
    active = sum((r.get("gpu_active_cycles_per_call") or 0.0) * r["calls"] for r in out)
    res = {"command": "MS=0 REPS=1 rocprofv3 {--kernel-trace --stats | --pmc <one group>} -- python3 tools/prof_update_only.py",
           "workload": "LidarSpread n=8 obs=3, 4096 envs x T=128: value pre-passes + 2 GAE + 32 minibatches (Vl, Vh, policy), one stream",
           "total_kernel_ms": total_us / 1e3,
           "update_mfma_busy_frac": (mfma_busy / (1024.0 * active)) if active else None,
           "units": "FETCH_SIZE / WRITE_SIZE in KB; *_cycles summed over the SIMDs sampled; mfma_busy_frac = busy cycles / (1024 SIMDs x "
                    "GRBM_GUI_ACTIVE / 8); kernels below 0.4 % of the kernel time are omitted",
           "kernels": out}
    json.dump(res, open(f"{DST}/{TAG}_nn_counters.json", "w"), indent=1)
    return res


def main():
    os.makedirs(DST, exist_ok=True)
    nn = nn_counters()
    if nn:
        print("nn counters:", len(nn["kernels"]), "kernels; update mfma busy frac", nn["update_mfma_busy_frac"])
    print(json.dumps(trace_summary(), indent=1))
    print(json.dumps(traffic(), indent=1))
    print(json.dumps(insts(), indent=1))
    for sub, name in (("bench_kt", "bench"), ("update_kt", "update")):
        st = find(sub, "kernel_stats.csv")
        if st:
            shutil.copy(st, f"{DST}/{TAG}_{name}_kernel_stats.csv")
    log = f"{SRC}/bench_plain.log"
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{") and '"metric"' in l]
        if lines:
            json.dump(json.loads(lines[-1]), open(f"{DST}/{TAG}_bench_line.json", "w"), indent=1)
    if os.path.exists(f"{SRC}/stamps.log"):
        shutil.copy(f"{SRC}/stamps.log", f"{DST}/{TAG}_env_wave_stamps.txt")
    if os.path.exists(f"{SRC}/valu_rate.log"):
        txt = open(f"{SRC}/valu_rate.log").read()
        if txt.strip().startswith("{"):
            open(f"{DST}/{TAG}_valu_issue_rates.json", "w").write(txt)


if __name__ == "__main__":
    main()
