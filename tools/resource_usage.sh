#!/bin/bash
# Register / scratch / occupancy report of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/resource_usage.sh vox_box.rs_amd/csrc/k_spectral.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas "$@" -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
def flush():
    if cur: print("%-64s vgpr %4s agpr %3s sgpr %4s scratch %5s occ %2s vspill %4s sspill %4s" % (
        cur.get("Function Name", "?")[:64], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize [bytes/lane]"),
        cur.get("Occupancy [waves/SIMD]"), cur.get("VGPRs Spill"), cur.get("SGPRs Spill")))
for line in sys.stdin:
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        flush(); cur = {}
    cur[k] = v
flush()
'
