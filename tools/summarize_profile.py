"""Condenses the rocprofv3 CSV output of tools/profile.sh into one text summary:
kernel-trace statistics plus per-kernel sums of every PMC counter collected (per dispatch average)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mi355rt::", "")
    return name.split("(")[0][:70]


print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for r in rows:
        print("%-72s %8s %14s %12.0f %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
# per-dispatch durations from the trace itself (min/median/max per kernel)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = defaultdict(list)
    regs = {}
    for r in csv.DictReader(open(f)):
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        regs[short(r["Kernel_Name"])] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    print("\n== per-dispatch duration (ns): n, min, median, max; VGPR, AGPR, SGPR, LDS, scratch ==")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print("%-72s %6d %10d %10d %10d   %s" % (k, len(v), v[0], v[len(v) // 2], v[-1], regs[k]))

print("\n== PMC counters: sum over dispatches / number of dispatches, per kernel ==")
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if k.startswith("at::"):
        continue  # torch's own helper kernels (film set-up and checks in bench.py)
    print(k)
    for c in sorted(acc[k]):
        # rocprofv3 emits one row per dispatch (and per dimension instance for some counters)
        print("    %-40s total %18.0f   rows %6d" % (c, acc[k][c], cnt[k][c]))
for f in glob.glob(os.path.join(out, "status.txt")):
    print("\n== pass status ==")
    print(open(f).read())
