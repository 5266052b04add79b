"""Developer tool: per-mode average duration of the raycast+graph kernel from a `rocprofv3 --kernel-trace` run of
`SIZES=4096 python3 tools/bench_env.py` (bench.roofline_env_kernel: warm-up + 100 launches with the GraphsTuple
materialised, then warm-up + 100 compact launches).  Writes profiles/r01_env_step_kernel_trace_summary.json."""
import csv, json, statistics, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lidar_step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gs = "Grid_Size_X" if "Grid_Size_X" in rows[0] else "Grid_Size"
big = [r for r in rows if int(r[gs]) == 4096 * 128]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in big]
# the last 2 x (warm-up + 100) launches belong to the roofline loop: api first, compact second
tail = dur[-(2 * 110):] if len(dur) >= 220 else dur
half = len(tail) // 2
api, cmp_ = tail[:half][-100:], tail[half:][-100:]
out = {
    "command": "rocprofv3 --kernel-trace --output-format csv -- python3 tools/bench_env.py  (SIZES=4096)",
    "kernel": big[0]["Kernel_Name"] if big else None,
    "envs_per_launch": 4096,
    "api_launches": len(api), "api_avg_us": statistics.mean(api), "api_median_us": statistics.median(api),
    "compact_launches": len(cmp_), "compact_avg_us": statistics.mean(cmp_), "compact_median_us": statistics.median(cmp_),
    "api_GBps_of_B_api": 9048 * 4096 / (statistics.mean(api) * 1e-6) / 1e9,
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
