"""Developer tool: derive HBM bytes per launch of the raycast+graph kernel from two rocprofv3 PMC passes over
tools/bench_env.py (separate runs: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with `--output-format csv`):

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> profiles/r01_env_step_traffic.json

Dispatches of 4096 workgroups: the first half of the timed launches materialises the GraphsTuple ("api"), the second
half is the compact variant.  Units and the gfx950 FETCH_SIZE x2 correction follow /opt/skills/guides/MI355X_MICROARCH.md."""
import csv, json, statistics, sys


def per_mode(path, counter, envs=4096, wg=128):
    vals = []
    for r in csv.DictReader(open(path)):
        if "lidar_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) == envs * wg:
            vals.append(float(r["Counter_Value"]))
    # bench_env.py: warm-up + api launches, then warm-up + compact launches; api writes ~13x more than compact
    hi = [v for v in vals if v > 0.5 * max(vals)] if counter == "WRITE_SIZE" else vals[:len(vals) // 2]
    lo = [v for v in vals if v <= 0.5 * max(vals)] if counter == "WRITE_SIZE" else vals[len(vals) // 2:]
    return statistics.median(hi), statistics.median(lo)


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    f_api, f_cmp = per_mode(fetch_csv, "FETCH_SIZE")
    w_api, w_cmp = per_mode(write_csv, "WRITE_SIZE")
    envs = 4096
    d = {
        "FETCH_SIZE_KB_api": f_api, "FETCH_SIZE_KB_compact": f_cmp, "WRITE_SIZE_KB_api": w_api, "WRITE_SIZE_KB_compact": w_cmp,
        "envs_per_launch": envs,
        "hbm_bytes_per_launch_api_raw": (f_api + w_api) * 1024,
        "hbm_bytes_per_launch_api_fetch_x2": (2 * f_api + w_api) * 1024,
        "algorithmic_bytes_per_launch_api": 9048 * envs,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/bench_env.py (kernel "
                "lidar_step_kernel<4,true,128>, 4096 workgroups; medians over the timed launches); WRITE_SIZE is exact for "
                "streaming stores, FETCH_SIZE under-reports coalesced reads by 2x on gfx950 and is uncalibrated for this "
                "kernel's narrow loads.",
    }
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
