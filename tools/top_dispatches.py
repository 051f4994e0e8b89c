"""lists the longest dispatches of a rocprofv3 --kernel-trace run: python tools/top_dispatches.py <dir> [n]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60], r["Grid_Size_X"], int(r["Start_Timestamp"])))
t0 = min(r[3] for r in rows)
rows.sort(key=lambda r: r[3])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
big = sorted(rows, key=lambda r: -r[0])[:n]
for d, k, g, s in sorted(big, key=lambda r: r[3]):
    print("%10.3f ms  +%9.3f ms  grid %9s  %s" % (d / 1e6, (s - t0) / 1e6, g, k))
