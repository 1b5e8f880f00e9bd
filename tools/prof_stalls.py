#!/usr/bin/env python3
"""tools/prof_stalls.sh <dir> <tag>: the five SQ passes of one bench command -> <dir>/<tag>_analyze_stalls.json: for every
kernel of the run, every counter's per-launch average and -- for the dominant kernel -- the split of its wavefront cycles
(SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY, quad-cycles, MI355X_MICROARCH.md "rocprofv3 PMC
slots"), the busy cycles per instruction class, the instruction mix per frame and the LDS conflict share."""
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
s = json.load(open(os.path.join(d, "summary.json")))["pmc"]
kern = {}
for p, ks in s.items():
    for k, cs in ks.items():
        kern.setdefault(k, {}).update({c: v["avg"] for c, v in cs.items()})
out = {"source": f"tools/prof_stalls.sh {tag} (rocprofv3 --pmc, five SQ passes of one bench command; per-launch averages)", "kernels": kern}
want = sys.argv[3] if len(sys.argv) > 3 else "analyze"
cands = [k for k in kern if "SQ_WAVE_CYCLES" in kern[k] and want in k] or [k for k in kern if "SQ_WAVE_CYCLES" in kern[k]]
dom = max(cands, key=lambda k: kern[k]["SQ_WAVE_CYCLES"])
c = kern[dom]
wc = c["SQ_WAVE_CYCLES"]
g = lambda n: c.get(n, 0.0)
frames = None
for line in open(os.path.join(d, "pmc_p1.log")):
    if line.startswith('{"metric'):
        frames = json.loads(line)["config"]["frames_per_gpu"]
acct = {
    "kernel": dom, "frames_per_launch": frames,
    "wave_cycles_split": {"active_inst_any": g("SQ_ACTIVE_INST_ANY") / wc, "wait_inst_any (issue stall: dependency / pipe)": g("SQ_WAIT_INST_ANY") / wc,
                          "of which wait_inst_lds": g("SQ_WAIT_INST_LDS") / wc, "wait_any (s_waitcnt / barrier)": g("SQ_WAIT_ANY") / wc,
                          "sum": (g("SQ_ACTIVE_INST_ANY") + g("SQ_WAIT_INST_ANY") + g("SQ_WAIT_ANY")) / wc},
    "active_cycles_by_class_over_wave_cycles": {n: g("SQ_ACTIVE_INST_" + n) / wc for n in ("VALU", "LDS", "SCA", "VMEM", "MISC", "FLAT")},
    "busy_simd_share": {"valu_active_over_busy": g("SQ_ACTIVE_INST_VALU") / max(g("SQ_BUSY_CYCLES"), 1.0),
                        "note": "SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) / SQ_BUSY_CYCLES"},
}
if frames:
    acct["instructions_per_frame"] = {n: g("SQ_INSTS_" + n) / frames for n in
                                      ("VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "SMEM", "BRANCH", "VALU_FMA_F64", "VALU_ADD_F64",
                                       "VALU_MUL_F64", "VALU_TRANS_F64", "VALU_INT32", "VALU_INT64", "VALU_CVT", "LDS_LOAD", "LDS_STORE")}
    v = acct["instructions_per_frame"]
    v["VALU_other (moves, DPP, selects, compares, f64 min/max ...)"] = v["VALU"] - sum(v[k] for k in ("VALU_FMA_F64", "VALU_ADD_F64", "VALU_MUL_F64", "VALU_TRANS_F64", "VALU_INT32", "VALU_INT64", "VALU_CVT"))
acct["lds"] = {"bank_conflict_cycles_over_lds_active": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
               "bank_conflict_cycles_over_wave_cycles_x4": 4.0 * g("SQ_LDS_BANK_CONFLICT") / wc if wc else None,
               "addr_conflict": g("SQ_LDS_ADDR_CONFLICT"), "unaligned_stall": g("SQ_LDS_UNALIGNED_STALL"),
               "data_fifo_full": g("SQ_LDS_DATA_FIFO_FULL"), "cmd_fifo_full": g("SQ_LDS_CMD_FIFO_FULL")}
out["dominant"] = acct
json.dump(out, open(os.path.join(d, f"{tag}_analyze_stalls.json"), "w"), indent=1)
print(json.dumps(acct, indent=1))
