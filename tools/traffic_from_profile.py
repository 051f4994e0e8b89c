"""Reads the FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ passes that tools/profile.sh left under gpurun_out/<tag>/ and
writes the per-launch HBM-side traffic of every k_* kernel as JSON (the figure bench.py reports as roofline.traffic).

Units and corrections: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE = read requests x 64 B
while every request moves one 128-byte line (checked here with tools/calib/random_lines.hip: one request per random
128-byte line whether 64 or 128 bytes of it are used, and with a plain streamed read of a known size), so read bytes =
2 x FETCH_SIZE.  WRITE_SIZE matches streamed 16-byte stores and float atomics as it stands.  Infinity-Cache hits are
included in both: this is traffic on the memory side of the L2, an upper bound of what reaches HBM.

Also recorded per kernel, from the SQ / TCC / GRBM passes of the same script (ratios of sums over all dispatches):
  valu_busy                      SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs): rocprof's VALUBusy formula
  valu_lane_utilization          SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): lanes doing work per vector instruction
  valu_issue_share_of_wave_time  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
  wait_share_of_wave_time        SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt / barriers)
  l2_hit_rate                    TCC_HIT / (TCC_HIT + TCC_MISS)"""
import csv, glob, json, os, sys
from collections import defaultdict

tag_dir, out_path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
rows = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(tag_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
                                 "SQ_WAIT_ANY", "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mi355rt::", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            rows[k][r["Counter_Name"]] += 1
res = {}
for k in acc:
    if not k.startswith("k_"):
        continue
    n = rows[k]["FETCH_SIZE"]
    res[k] = {"dispatches": n,
              "read_bytes_per_launch": 2.0 * acc[k]["FETCH_SIZE"] * 1024 / n,
              "write_bytes_per_launch": acc[k]["WRITE_SIZE"] * 1024 / max(rows[k]["WRITE_SIZE"], 1),
              "read_requests_per_launch": acc[k]["TCC_EA0_RDREQ_sum"] / max(rows[k]["TCC_EA0_RDREQ_sum"], 1)}
    res[k]["bytes_per_launch"] = res[k]["read_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    a = acc[k]
    if a["SQ_ACTIVE_INST_VALU"] > 0 and a["SQ_WAVE_CYCLES"] > 0:
        res[k]["valu_lane_utilization"] = a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"])
        res[k]["valu_issue_share_of_wave_time"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]
        res[k]["wait_share_of_wave_time"] = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]
        if a["GRBM_GUI_ACTIVE"] > 0:
            res[k]["valu_busy"] = a["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * a["GRBM_GUI_ACTIVE"] / 8.0)
        res[k]["valu_instructions_per_launch"] = a["SQ_INSTS_VALU"] / max(rows[k]["SQ_INSTS_VALU"], 1)
    if a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
        res[k]["l2_hit_rate"] = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh, read bytes = 2 x FETCH_SIZE (gfx950)",
           "command": " ".join(sys.argv[3:]), "kernels": res}, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
