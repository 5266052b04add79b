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

TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
    rows = pmc_rows("env_insts")
    if not rows:
        return None
    g = max(int(r["Grid_Size"]) for r in rows)
    res = {}
    for variant, want in (("api", True), ("compact", False)):
        sel = [r for r in rows if int(r["Grid_Size"]) == g and is_graph(r["Kernel_Name"]) == want and mode_of(r["Kernel_Name"]) == "0"]
        c = {}
        for r in sel:
            c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        c = {k: statistics.median(v) for k, v in c.items()}
        if not c:
            continue
        waves = c.get("SQ_WAVES", 0) or 1
        res[variant] = {"counters_per_launch": c, "waves_per_launch": waves, "envs_per_launch": 4096,
                        "per_wave": {k: v / waves for k, v in c.items() if k.startswith("SQ_INSTS")},
                        "per_env": {k: v / 4096 for k, v in c.items() if k.startswith("SQ_INSTS")}}
    res["note"] = ("rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                   "SQ_ACTIVE_INST_VALU over SIZES=4096 tools/bench_env.py; persistent waves walk several envs each, so the "
                   "per-env figure (not the per-wave one) is what compares with round 1's 2 waves x 1 661 VALU = 3 322 per env")
    json.dump(res, open(f"{DST}/{TAG}_env_step_insts.json", "w"), indent=1)
    return res


def main():
    os.makedirs(DST, exist_ok=True)
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


if __name__ == "__main__":
    main()
