"""diagnostic: per-launch timeline of one render step from a rocprofv3 kernel trace CSV"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mi355rt::", "")
    return n.split("(")[0][:28]
# last k_generate marks the start of the last step
gen = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]]
start = gen[-1]
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:start + 120]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %9.1f us  dur %8.1f us  gap %7.1f us  grid %s" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size")))
    prev_end = e
