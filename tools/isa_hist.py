#!/usr/bin/env python
"""Instruction-class histogram of one kernel from hipcc's assembly, per phase, priced with issue rates MEASURED on MI355X
(tools/micro/valu_rate.hip -> profiles/r03_valu_issue_rates.json; MI355X_MICROARCH.md's table gives 2 cycles for v_fma_f32 only) — VERDICT r2 item 3: where do the issue slots of
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

# Cycles a wave64 instruction occupies its pipe, per SIMD, with the pipe saturated (>= 4 waves per SIMD) — MEASURED on MI355X
# with tools/micro/valu_rate.hip (profiles/r03_valu_issue_rates.json), not taken from the guide's table:
#   * "full": v_add / v_sub / v_mul / v_fma / v_fmac _f32, v_mov_b32, v_and / v_xor / v_or _b32, v_add_u32 ... 2.3 cycles
#   * "half": v_min / v_max / v_med3 _f32, every v_cmp*, v_cndmask, shifts, v_bfe, v_mad_u32_u24, v_div_scale / fmas / fixup,
#     every DPP form, v_readlane, v_pk_mul_f32 (two multiplies) ... 4.2 cycles — HALF rate
#   * "trans": v_rcp / v_sqrt ... 8.2 cycles
#   * SALU: 4.2 cycles per SIMD (one scalar pipe per CU, shared by its four SIMDs); overlaps with the VALU of other waves
#   * ds_read_b128: ~23 cycles per SIMD (LDS return path: 1 KiB per wave-instruction), same for a broadcast address
#   * a single wave issues at most one instruction per ~4.6 cycles, whatever the class
# (mnemonics that were not measured are put with their nearest measured relative and marked in MEASURED below)
PRICE = {"valu_full": 2.3, "valu_half": 4.2, "valu_trans": 8.2, "lds": 16.0, "vmem": 4.0, "salu": 4.2, "smem": 4.2, "branch": 4.2,
         "waitcnt": 0.0, "nop": 1.2}
FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32")
MEASURED = {"full": ["v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_and_b32", "v_xor_b32", "v_add_u32"],
            "half": ["v_min_f32", "v_med3_f32", "v_bfe_u32", "v_lshlrev_b32", "v_mad_u32_u24", "v_cndmask_b32", "v_cmp_lt_f32",
                     "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_mov_b32_dpp", "v_min_f32_dpp", "v_readlane_b32",
                     "v_pk_mul_f32"],
            "trans": ["v_rcp_f32", "v_sqrt_f32"], "salu": ["s_and_b64"], "lds": ["ds_read_b128"], "nop": ["s_nop 0"]}

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
    if mn.startswith(TRANS):
        return "valu_trans"
    if "dpp" in ops or "row_" in ops or "quad_perm" in ops or mn.endswith("_dpp"):
        return "valu_half"
    base = mn[:-4] if mn.endswith(("_e32", "_e64")) else mn
    if base in FULL:
        return "valu_full"
    if mn.startswith("v_"):
        return "valu_half"
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
    """cycles of the busiest pipe's work if nothing overlapped within a class; VALU, SALU and LDS are separate pipes, so the
    VALU sum is the number to hold against the measured cycles per env and SIMD"""
    return round(sum(PRICE.get(k, 2.3) * v for k, v in counter.items()), 1)


def priced_valu(counter):
    return round(sum(PRICE[k] * v for k, v in counter.items() if k.startswith("valu")), 1)


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
        "price_table_cycles_per_wave64_instruction_per_simd": PRICE, "price_table_source": "profiles/r03_valu_issue_rates.json (measured)",
        "measured_mnemonics": MEASURED, "priced_valu_cycles_static": priced_valu(total),
        "per_phase": {k: {"counts": dict(v), "valu": sum(x for kk, x in v.items() if kk.startswith("valu")), "priced_valu_cycles": priced_valu(v),
                          "priced_cycles": priced(v)}
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
        print(f"{name:24s} " + " ".join(f"{c.get(k, 0):9d}" for k in PRICE) + f"   {priced(c):8.0f}")
    print(f"{'TOTAL (production build)':24s} " + " ".join(f"{total.get(k, 0):9d}" for k in PRICE) + f"   {priced(total):8.0f}   (VALU pipe alone: {priced_valu(total):.0f})")
    print("top mnemonics:", ", ".join(f"{m}:{n}" for m, n in mn_counts.most_common(30)))


if __name__ == "__main__":
    main()
