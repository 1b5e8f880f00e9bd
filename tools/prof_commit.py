#!/usr/bin/env python3
"""Copies the judged evidence of one tools/prof_r02.sh run from gpurun_out/prof_<tag>/ into profiles/ (tracked):
kernel-stats CSVs, the bench lines printed under the profiler, the per-kernel PMC averages, and
profiles/pmc_traffic.json -- measured HBM bytes per frame of each kernel, which bench.py reads for roofline.traffic.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies a 128-B request of a wide streaming read at
64 B, MI355X_MICROARCH.md "HBM").  usage: tools/prof_commit.py <tag> [git-commit]"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
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

# frames per launch of each PMC run, from the bench line printed in that run
def frames_of(run):
    try:
        line = json.loads(open(os.path.join(src, run + ".bench_line.json")).readline())
        return line["config"]["frames_per_gpu"]
    except (OSError, ValueError, KeyError):
        return None

KEYS = {   # bench.py profile name -> (workload, kernel-name prefix in the counter files)
    "analyze": ("pipeline", "void analyze_kernel<true, true, true, 0>"),
    "pitch": ("config3", "void analyze_kernel<false, false, true, 0>"),
    "analyze_2048": ("pipeline_2048", "void analyze_pow2_kernel<2, true, true, true, 0>"),     # bench.py --frame-len 2048 --hop 1024
    "pitch_2048": ("config3_2048", "void analyze_pow2_kernel<2, false, false, true, 0>"),
    "pitch_1024": ("config3_1024", "void analyze_pow2_kernel<1, false, false, true, 0>"),
    "burg_512": ("config4", "void burg_kernel<16, 32, double>"),
    "burg": ("pipeline", "void burg_kernel<64, 20, double>"),
    "formant_resonances_512": ("config4", "formant_resonances_kernel"),
    "formant_resonances": ("pipeline", "formant_resonances_kernel"),
    "tracker_512": ("config4", "void tracker_kernel<4>"),                 # VBX_TRACKER_CHUNKED=0 runs (time-sliced scan)
    "tracker": ("pipeline", "void tracker_kernel<4>"),
    "tracker_chunked_512": ("config4", "void tracker_spec_kernel<4>"),    # the scan's first and longest kernel
    "tracker_chunked": ("pipeline", "void tracker_spec_kernel<4>"),
    "autocorr_lpc_512": ("config2", "void autocorr_fewlags_kernel<8, 13>"),
}
traffic = {}
for key, (wl, kern) in KEYS.items():
    fe = summ["pmc"].get("pmc_fetch_" + wl, {}).get(kern, {}).get("FETCH_SIZE")
    wr = summ["pmc"].get("pmc_write_" + wl, {}).get(kern, {}).get("WRITE_SIZE")
    F = frames_of("pmc_fetch_" + wl)
    if not fe or not wr or not F:
        continue
    # sliced kernels run several launches per step: the counter averages are per launch, the frames too
    # the time-sliced find_formants (VBX_TRACKER_CHUNKED=0) runs its kernels VBX_FF_SLICES (6) times per step
    SLICES = int(os.environ.get("VBX_FF_SLICES", "6")) if os.environ.get("VBX_TRACKER_CHUNKED") == "0" else 1
    per_step = SLICES if key.split("_")[0] in ("burg", "formant", "tracker") else 1
    F = F / per_step
    fetch_b, write_b = fe["avg"] * 1024.0 * 2.0, wr["avg"] * 1024.0
    traffic[key] = {"bytes_per_frame": (fetch_b + write_b) / F, "fetch_bytes_per_frame": fetch_b / F,
                    "write_bytes_per_frame": write_b / F, "frames_per_launch": F, "kernel": kern,
                    "FETCH_SIZE_KB": fe["avg"], "WRITE_SIZE_KB": wr["avg"],
                    "source": f"profiles/{tag}_pmc_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                              "FETCH_SIZE x1024 x2, WRITE_SIZE x1024)", "commit": commit}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for k, v in traffic.items():
    print("%-20s %10.0f B/frame (fetch %.0f + write %.0f)" % (k, v["bytes_per_frame"], v["fetch_bytes_per_frame"], v["write_bytes_per_frame"]))
