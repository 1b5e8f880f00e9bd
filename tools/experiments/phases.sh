#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
VBX_LIB_PATH=$R/vox_box.rs_amd/lib/libvoxbox_hip_phases.so python3 tools/experiments/phases.py ${1:-0.5} ${2:-1200} ${3:-480} ${4:-48000} 2> gpurun_out/phases.err > /dev/null
python3 - <<'PY'
import os,re
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
names=["load","fft_fwd","split","mfcc_rest","fft_inv","normalise","peak_scan","filter","bounds","refine","mfcc_sums","mfcc_log","mfcc_dct","mfcc_products(interp: split+Z)","interp_loop","interp_after_tail"]
lab=None
for l in open(R+"/gpurun_out/phases.err"):
    if l.startswith("VBX_PHASES_LABEL"): lab=l.split()[1:]; continue
    if l.startswith("VBX_PHASES "):
        t=l.split(); frames=int(t[2]); cyc=[int(x) for x in t[4:]]
        tot=sum(cyc)
        print(lab, "frames", frames, "cycles/frame", round(tot/max(frames,1)))
        for n,c in zip(names,cyc): print("   %-34s %8.0f cycles/frame  %5.1f %%" % (n, c/max(frames,1), 100.0*c/max(tot,1)))
PY
