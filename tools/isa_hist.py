#!/usr/bin/env python
"""Instruction-class histogram of one kernel from hipcc's assembly, per phase, priced with the cycle table of
MI355X_MICROARCH.md ("Per-instruction cycle constants") — VERDICT r2 item 3: where do the issue slots of
lidar_wave_kernel<4,true,8,3,4,0,true> go?

    python tools/isa_hist.py [--kernel SUBSTR] [--out profiles/r03_env_wave_isa_hist.json]

Compiles dgppo_amd/csrc/env_wave.hip twice with the production flags (-O3 -ffp-contract=off, gfx950): once as shipped (the
whole-kernel histogram) and once with -DDGPPO_PHASE_MARKS, which turns the PHASE() markers into assembler comments (no
"memory" clobber: they do not fence the scheduler) so that the stream can be split by phase.  Counts are STATIC; the body
is fully unrolled straight-line code, so they equal the dynamic count of one environment except where noted:
  * regions behind a wave-uniform skip (the obstacle cull, the "some lane hits" division block) run less often;
  * the slow paths (det == 0 / NaN literal arithmetic, top-k tie-break) are out of line and practically never run;
  * the top-k slot loop runs ceil(#hits / 4) times.
Needs no GPU."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-pass-failed", "-ffp-contract=off",
         "-S", "--cuda-device-only"]

# issue cost in cycles of a wave64 instruction on its SIMD with several waves resident (MI355X_MICROARCH.md: v_fma_f32 2 cyc
# SIMD-32 throughput; transcendentals 8 = 4x the one-wave 4-cycle base -> 2x here is not documented, the table's ratio is
# kept: 2 -> 4).  64-bit integer / f64 ops and v_div_fixup/fmas/scale are full-rate encodings on CDNA (VOP3, 2 passes).
PRICE = {"valu_full": 2, "valu_trans": 4, "valu_dpp": 2, "valu_xlane": 2, "valu_cmp": 2, "valu_div": 2, "lds": 2, "vmem": 2,
         "salu": 1, "smem": 1, "branch": 1, "waitcnt": 0, "nop": 1}

TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def classify(mn: str, ops: str) -> str:
    if mn.startswith("s_waitcnt"):
        return "waitcnt"
    if mn.startswith(("s_nop", "s_sleep")):
        return "nop"
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_getpc")):
        return "branch"
    if mn.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute")):
        return "valu_xlane"
    if "dpp" in ops or "row_" in ops or "quad_perm" in ops or mn.endswith("_dpp"):
        return "valu_dpp"
    if mn.startswith(TRANS):
        return "valu_trans"
    if mn.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return "valu_div"
    if mn.startswith("v_cmp") or mn.startswith("v_cndmask"):
        return "valu_cmp"
    if mn.startswith("v_"):
        return "valu_full"
    return "other"


def kernel_body(asm: str, needle: str):
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        if ln.endswith(":") or ":" in ln.split(";")[0]:
            lab = ln.split(":")[0].strip()
            if needle in lab and lab.startswith("_Z") and start is None:
                start = i
        if start is not None and ln.strip().startswith(".Lfunc_end"):
            return lines[start + 1:i]
    raise SystemExit(f"kernel containing '{needle}' not found")


INST = re.compile(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)\s*(;.*)?$")


def histogram(body, split_phases: bool):
    phases = collections.OrderedDict()
    cur = "prologue"
    mn_counts = collections.Counter()
    for ln in body:
        if "; PHASE" in ln:
            cur = ln.split("; PHASE", 1)[1].strip() if split_phases else cur
            continue
        st = ln.strip()
        if not st or st.startswith((".", ";")) or st.endswith(":"):
            continue
        m = INST.match(ln)
        if not m:
            continue
        mn, ops = m.group(1), m.group(2)
        cls = classify(mn, ops)
        phases.setdefault(cur, collections.Counter())[cls] += 1
        mn_counts[mn] += 1
    return phases, mn_counts


def compile_asm(src: str, extra):
    out = subprocess.run([HIPCC] + FLAGS + extra + ["-o", "-", src], capture_output=True, text=True, cwd=os.path.dirname(src))
    if out.returncode != 0:
        sys.stderr.write(out.stderr[-3000:])
        raise SystemExit("hipcc failed")
    return out.stdout


def priced(counter):
    return sum(PRICE.get(k, 2) * v for k, v in counter.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="lidar_wave_kernelILi4ELb1ELi8ELi3ELi4ELi0ELb1")
    ap.add_argument("--src", default=os.path.join(ROOT, "dgppo_amd", "csrc", "env_wave.hip"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--measured-cycles-per-env", type=float, default=None,
                    help="SIMD cycles per env from the launch-time slope (us per extra 1024 envs x clock)")
    a = ap.parse_args()
    prod = kernel_body(compile_asm(a.src, []), a.kernel)
    marks = kernel_body(compile_asm(a.src, ["-DDGPPO_PHASE_MARKS"]), a.kernel)
    whole, mn_counts = histogram(prod, False)
    total = collections.Counter()
    for c in whole.values():
        total.update(c)
    per_phase, _ = histogram(marks, True)
    # merge the four unrolled (agent pair) iterations of the ray / top-k phases
    merged = collections.OrderedDict()
    for name, c in per_phase.items():
        merged.setdefault(name, collections.Counter()).update(c)
    valu = sum(v for k, v in total.items() if k.startswith("valu"))
    res = {
        "kernel": a.kernel, "flags": " ".join(FLAGS[:-2]),
        "static_counts": dict(total), "static_valu": valu,
        "priced_cycles_static": priced(total),
        "price_table_cycles_per_wave64_instruction": PRICE,
        "per_phase": {k: {"counts": dict(v), "valu": sum(x for kk, x in v.items() if kk.startswith("valu")), "priced_cycles": priced(v)}
                      for k, v in merged.items()},
        "top_mnemonics": mn_counts.most_common(45),
    }
    if a.measured_cycles_per_env:
        res["measured_cycles_per_env"] = a.measured_cycles_per_env
        res["priced_over_measured"] = priced(total) / a.measured_cycles_per_env
    txt = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(f"kernel {a.kernel}: static {sum(total.values())} instructions, {valu} VALU; priced {priced(total)} cycles")
    print(f"{'phase':24s} " + " ".join(f"{k[:9]:>9s}" for k in PRICE) + "   priced")
    for name, c in merged.items():
        print(f"{name:24s} " + " ".join(f"{c.get(k, 0):9d}" for k in PRICE) + f"   {priced(c):6d}")
    print(f"{'TOTAL (production build)':24s} " + " ".join(f"{total.get(k, 0):9d}" for k in PRICE) + f"   {priced(total):6d}")
    print("top mnemonics:", ", ".join(f"{m}:{n}" for m, n in mn_counts.most_common(30)))


if __name__ == "__main__":
    main()
