#!/usr/bin/env python3
"""Copies the judged evidence of one tools/prof_round.sh run from gpurun_out/prof_<tag>/ into profiles/ (tracked):
kernel-stats CSVs, the bench lines printed under the profiler, the per-kernel PMC averages, and
profiles/pmc_traffic.json -- per kernel: measured HBM bytes per frame (bench.py reads it for roofline.traffic) and, where
an SQ pass exists, vector instructions per frame and the share of SIMD time the vector ALU is issuing (roofline.issue_frac).
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies a 128-B request of a wide streaming read at
64 B, MI355X_MICROARCH.md "HBM").  A kernel listed in KEYS that the counter files do not hold is a HARD ERROR (a kernel
renamed in the sources must not silently drop out of the evidence); shapes that were not profiled (prof_r03.sh quick) are
skipped as a whole and say so.  usage: tools/prof_commit.py <tag> [git-commit] [subdirectory of profiles/ for the copied files]
(pmc_traffic.json stays at profiles/pmc_traffic.json: bench.py reads it there)"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
sub = sys.argv[3] if len(sys.argv) > 3 else ""
dst = os.path.join(root, "profiles", sub) if sub else os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
rel = ("profiles/" + sub + "/" if sub else "profiles/")
commit = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
summ = json.load(open(os.path.join(src, "summary.json")))

for t in glob.glob(os.path.join(src, "trace_*")):
    if os.path.isdir(t):
        name = os.path.basename(t).replace("trace_", "")
        for f in glob.glob(os.path.join(t, "*", "*kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
        bl = os.path.join(src, os.path.basename(t) + ".bench_line.json")
        if os.path.exists(bl) and os.path.getsize(bl):
            shutil.copy(bl, os.path.join(dst, f"{tag}_{name}_bench_line.json"))
json.dump(summ["pmc"], open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w"), indent=1, sort_keys=True)


def bench_line(run):
    try:
        return json.loads(open(os.path.join(src, run + ".bench_line.json")).readline())
    except (OSError, ValueError):
        return None


# bench.py profile name -> (workload suffix of the pmc_* runs, kernel-name PREFIXES in the counter files, unit).  Several
# prefixes = one logical kernel made of several launches (the chunked tracker scan): their per-step totals are summed.
KEYS = {
    "analyze": ("pipeline", ["void analyze_kernel<true, true, true, 0,"], "frame"),
    "pitch": ("config3", ["void analyze_kernel<false, false, true, 0,"], "frame"),
    "analyze_2048": ("pipeline_2048", ["void analyze_pow2_kernel<2, true, true, true, 0,"], "frame"),     # --frame-len 2048 --hop 1024
    "pitch_2048": ("config3_2048", ["void analyze_pow2_kernel<2, false, false, true, 0,"], "frame"),
    "pitch_1024": ("config3_1024", ["void analyze_pow2_kernel<1, false, false, true, 0,"], "frame"),
    # --frame-len 4096 --hop 2048: two wavefronts per frame; since the end of round 5 the transforms, the peak scan, the refinement and the far frames are four kernels
    "analyze_4096": ("pipeline_4096", ["void analyze_pow2_kernel<2, true, true, true, 5, 2", "vbx::scan_curve_kernel", "vbx::refine_list_kernel", "vbx::refine_far_kernel"], "frame"),
    "pitch_4096": ("config3_4096", ["void analyze_pow2_kernel<2, false, false, true, 5, 2", "vbx::scan_curve_kernel", "vbx::refine_list_kernel", "vbx::refine_far_kernel"], "frame"),
    "analyze_3000": ("pipeline_3000", ["void analyze_pow2_kernel<2, true, true, false, 6, 2", "vbx::scan_curve_kernel", "vbx::refine_list_kernel", "vbx::refine_far_kernel"], "frame"),   # --frame-len 3000 --hop 1200: the curves through HBM
    "analyze_1103": ("pipeline_1103", ["void analyze_kernel<true, true, false, 4,"], "frame"),             # --frame-len 1103 --hop 441: MFCC by interpolated bins
    # Burg = the one-pass form (k_burg_fast.hip): lag sums, recursion, and the direct recursion on the frames its guard sent on
    "burg_lags_512": ("config4", ["void burg_lags_kernel<8, 12, double"], "frame"),
    "burg_lags": ("pipeline", ["void burg_lags_kernel<20, 12, double"], "frame"),
    "burg_recursion_512": ("config4", ["void burg_recursion_kernel<12>"], "frame"),
    "burg_recursion": ("pipeline", ["void burg_recursion_kernel<12>"], "frame"),
    "burg_direct_list_512": ("config4", ["void burg_kernel<16, 32, double"], "frame"),
    "burg_direct_list": ("pipeline", ["void burg_kernel<64, 20, double"], "frame"),
    "formant_resonances_512": ("config4", ["void formant_resonances_fast_kernel<12>"], "frame"),     # k_roots_fast.hip
    "formant_resonances": ("pipeline", ["void formant_resonances_fast_kernel<12>"], "frame"),
    "tracker_chunked_512": ("config4", ["void tracker_spec_kernel<4>", "void tracker_check_kernel<4>", "void tracker_repair_kernel<4>",
                                        "void tracker_sweep_kernel<4>"], "frame"),
    "tracker_chunked": ("pipeline", ["void tracker_spec_kernel<4>", "void tracker_check_kernel<4>", "void tracker_repair_kernel<4>",
                                     "void tracker_sweep_kernel<4>"], "frame"),
    # round 6: LPC::lpc of the fused call as its own lane-per-row kernel (+ the conditioning probe), and the double-double redo of the rows it lists
    "lpc_rows": ("pipeline", ["void levinson_rows_kernel_t<12, true>"], "frame"),          # + the deferred MFCC tail of the same record
    "lpc_exact_list": ("pipeline", ["lpc_exact_list_kernel", "vbx::lpc_exact_list_kernel"], "frame"),
    "autocorr_lpc_512": ("config2", ["void autocorr_fewlags_kernel<8, 13, double"], "frame"),
    "pcm16": ("frontend", ["void pcm16_kernel", "pcm16_kernel"], "sample"),
}
N_SIMD = 256 * 4


def find(run, prefixes, counter):
    """[(avg, n)] of `counter` for every kernel of `run` whose name starts with one of the prefixes."""
    hits = []
    for pre in prefixes:
        got = [(v[counter]["avg"], v[counter]["n"]) for k, v in sorted(summ["pmc"].get(run, {}).items()) if k.startswith(pre) and counter in v]
        hits.extend(got[:1])
    return hits


traffic, skipped = {}, []
for key, (wl, prefixes, unit) in KEYS.items():
    if "pmc_fetch_" + wl not in summ["pmc"]:
        skipped.append(key)                                   # the whole shape was not profiled this time
        continue
    line = bench_line("pmc_fetch_" + wl)
    if line is None:
        sys.exit(f"prof_commit: no bench line for pmc_fetch_{wl}")
    units = line["config"]["samples"] if unit == "sample" else line["config"]["frames_per_gpu"]
    steps_total = line["steps"] + line["warmup"]               # the profiler sees the warm-up launches too
    fe, wr = find("pmc_fetch_" + wl, prefixes, "FETCH_SIZE"), find("pmc_write_" + wl, prefixes, "WRITE_SIZE")
    if not fe or not wr:
        have = sorted(summ["pmc"].get("pmc_fetch_" + wl, {}))
        sys.exit(f"prof_commit: kernel of '{key}' ({prefixes}) is not in pmc_fetch_{wl} / pmc_write_{wl}; kernels there: {have}")
    # bytes per STEP of the logical kernel = sum over its launches (avg per launch x launches per step)
    fetch_b = sum(a * n for a, n in fe) / steps_total * 1024.0 * 2.0
    write_b = sum(a * n for a, n in wr) / steps_total * 1024.0
    e = {"bytes_per_frame": (fetch_b + write_b) / units, "fetch_bytes_per_frame": fetch_b / units,
         "write_bytes_per_frame": write_b / units, "per": unit, "units_per_step": units, "kernels": prefixes,
         "source": f"{rel}{tag}_pmc_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                   "FETCH_SIZE x1024 x2, WRITE_SIZE x1024; launches of one step summed)", "commit": commit}
    sq = "pmc_sq_" + wl
    if sq in summ["pmc"]:
        lsq = bench_line(sq)
        st_sq = (lsq["steps"] + lsq["warmup"]) if lsq else steps_total
        tot = lambda c: sum(a * n for a, n in find(sq, prefixes, c)) / st_sq
        insts, act, gui, wcyc, waves = tot("SQ_INSTS_VALU"), tot("SQ_ACTIVE_INST_VALU"), tot("GRBM_GUI_ACTIVE"), tot("SQ_WAVE_CYCLES"), tot("SQ_WAVES")
        if insts and gui:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over every SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            simd_quads = gui / 8.0 / 4.0 * N_SIMD
            e.update({"valu_insts_per_frame": insts / units,
                      "valu_busy": act / simd_quads, "wave_quad_cycles_per_wave": wcyc / max(waves, 1),
                      "valu_active_quad_cycles_per_wave": act / max(waves, 1),
                      "sq_source": f"{rel}{tag}_pmc_counters.json {sq}: SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 / 4 x {N_SIMD} SIMDs)"})
    traffic[key] = e
json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for k, v in traffic.items():
    print("%-24s %10.1f B/%s (fetch %.1f + write %.1f)%s" % (k, v["bytes_per_frame"], v["per"], v["fetch_bytes_per_frame"], v["write_bytes_per_frame"],
                                                         "  VALU busy %.2f, %.0f VALU insts/%s" % (v["valu_busy"], v["valu_insts_per_frame"], v["per"]) if "valu_busy" in v else ""))
if skipped:
    print("not profiled in this run (kept out of pmc_traffic.json):", ", ".join(skipped))
