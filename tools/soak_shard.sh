#!/bin/bash
# EVERY frame of a stretch of the bench's recording against the CPU oracle, in pieces (host memory stays small):
#   tools/soak_shard.sh <total_frames> <piece_frames> [frame_len=1200] [hop=480]      (GPU box; ~1,600 frames/s on 16 host threads)
# writes gpurun_out/soak_shard/piece_<k>.json and total.json (the sum of the pieces' disagreement counters)
TOTAL=$1; PIECE=$2; N=${3:-1200}; H=${4:-480}
O=gpurun_out/soak_shard; mkdir -p $O
k=0
for ((f = 0; f < TOTAL; f += PIECE)); do
  n=$PIECE; if ((f + n > TOTAL)); then n=$((TOTAL - f)); fi
  python3 tools/soak_parity.py $n $O/piece_$k.json $N $H 1 $f 2>&1 | tail -1 | cut -c1-300
  k=$((k + 1))
done
python3 - <<PY
import glob, json
tot = {}; frames = 0
for p in sorted(glob.glob("$O/piece_*.json")):
    d = json.load(open(p)); frames += d["frames"]
    for k, v in d["disagreements"].items(): tot[k] = tot.get(k, 0) + v
out = {"frame_len": $N, "hop": $H, "frames": frames, "every_frame_of": "frames 0 .. %d of the bench recording (sample offset 5 s)" % frames, "disagreements": tot}
json.dump(out, open("$O/total.json", "w"), indent=1); print(json.dumps(out))
PY
