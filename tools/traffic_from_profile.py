"""Reads the FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ passes that tools/profile.sh left under gpurun_out/<tag>/ and
writes the per-launch HBM-side traffic of every k_* kernel as JSON (the figure bench.py reports as roofline.traffic).

Units and corrections: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE = read requests x 64 B
while every request moves one 128-byte line (checked here with tools/calib/random_lines.hip: one request per random
128-byte line whether 64 or 128 bytes of it are used, and with a plain streamed read of a known size), so read bytes =
2 x FETCH_SIZE.  WRITE_SIZE matches streamed 16-byte stores and float atomics as it stands.  Infinity-Cache hits are
included in both: this is traffic on the memory side of the L2, an upper bound of what reaches HBM.

Also recorded per kernel, from the SQ / TCC passes of the same script (ratios of sums over all dispatches):
  valu_issue_frac                wave-level vector instructions per second over the sustained issue rate of this chip for the kernel's
                                 own instruction mix, measured by tools/calib/valu_issue_rate (profiles/r03_valu_issue_rate.txt, the
                                 4-waves-per-SIMD column -- the occupancy these kernels run at): fp64 instructions (counted by the
                                 SQ_INSTS_VALU_*_F64 pass) priced at the "f64" rate, the rest at the "node" rate.  The time is the
                                 kernel's own (kernel trace of the same script).  This replaces rocprof's VALUBusy formula
                                 (SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE), which assumes four cycles per instruction
                                 and read 1.05 on a kernel of this library.
  valu_lane_utilization          SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): lanes doing work per vector instruction
  valu_issue_share_of_wave_time  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
  wait_share_of_wave_time        SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt / barriers)
  wait_inst_share_of_wave_time   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting for an instruction to come back: memory latency)
  l2_hit_rate                    TCC_HIT / (TCC_HIT + TCC_MISS)"""
import csv, glob, json, os, sys
from collections import defaultdict

tag_dir, out_path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
rows = defaultdict(lambda: defaultdict(int))
def newest_per_directory(paths):
    """one file per directory, the newest: results of an earlier run of the same passes may still lie beside the new ones"""
    best = {}
    for f in paths:
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest_per_directory(glob.glob(os.path.join(tag_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
                                 "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD",
                                 "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mi355rt::", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            rows[k][r["Counter_Name"]] += 1
# the kernels' own durations, from the kernel trace of the same script
dur = defaultdict(lambda: [0.0, 0])
for f in newest_per_directory(glob.glob(os.path.join(tag_dir, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mi355rt::", "").split("(")[0]
        dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        dur[k][1] += 1


def calibrated_rates():
    """wave-level vector instructions per second the chip sustains at 4 waves per SIMD, per mix (tools/calib/valu_issue_rate)"""
    rates = {}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_valu_issue_rate.txt")
    try:
        for line in open(path):
            w = line.split()
            if w and w[0] in ("fma", "node", "f64", "cndmask") and len(w) >= 9:
                rates[w[0]] = float(w[5]) * 1e9  # name, then (rate, [cycles]) per column: 1, 2, 4, 8 waves per SIMD
    except OSError:
        pass
    return rates


RATES = calibrated_rates()
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
        if a["SQ_WAIT_INST_ANY"] > 0:
            res[k]["wait_inst_share_of_wave_time"] = a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]
        res[k]["valu_instructions_per_launch"] = a["SQ_INSTS_VALU"] / max(rows[k]["SQ_INSTS_VALU"], 1)
        f64 = sum(a[c] / max(rows[k][c], 1) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        res[k]["valu_f64_instructions_per_launch"] = f64
        if dur[k][1] and "node" in RATES and "f64" in RATES:
            seconds = dur[k][0] / dur[k][1]
            res[k]["seconds_per_launch_in_trace"] = seconds
            n_all = res[k]["valu_instructions_per_launch"]
            res[k]["valu_instr_per_s"] = n_all / seconds
            # time the instruction stream alone would take at the sustained rates over the time the kernel took
            res[k]["valu_issue_frac"] = ((n_all - f64) / RATES["node"] + f64 / RATES["f64"]) / seconds
            res[k]["valu_issue_rates"] = {"node": RATES["node"], "f64": RATES["f64"], "source": "profiles/r03_valu_issue_rate.txt, 4 waves per SIMD"}
    if a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
        res[k]["l2_hit_rate"] = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh, read bytes = 2 x FETCH_SIZE (gfx950)",
           "command": " ".join(sys.argv[3:]), "kernels": res}, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
