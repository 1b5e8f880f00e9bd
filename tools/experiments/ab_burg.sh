#!/bin/bash
# A/B of the one-pass Burg (k_burg_fast.hip) against the direct recursion (VBX_BURG_DIRECT=1): pipeline and config 4 bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ab_burg
for d in 0 1; do
  for rep in 1 2; do
    VBX_BURG_DIRECT=$d python bench.py --no-cpu --no-sub --steps 3 --warmup 1 | grep '^{"metric' > gpurun_out/ab_burg/pipeline_direct${d}_$rep.json
    VBX_BURG_DIRECT=$d python bench.py --workload config4 --no-cpu --no-sub --steps 10 --warmup 2 | grep '^{"metric' > gpurun_out/ab_burg/config4_direct${d}_$rep.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_burg/*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], "%.4g" % d["value"], d["unit"], "ms/step %.3f" % d["ms_per_step"], "kernels", {k: round(v, 2) for k, v in (d.get("kernel_ms") or {}).items()} if "kernel_ms" in d else "")
PY
