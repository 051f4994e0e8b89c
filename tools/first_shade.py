"""per-kernel dispatch durations of a tools/kernel_trace.sh run: count, longest, mean of the 16 longest (the first bounce of every batch)
usage: python tools/first_shade.py gpurun_out/<tag>"""
import csv, glob, re, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        if m:
            d[m.group(0)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print("%-40s n %4d  max %9d ns  mean of 16 longest %9.0f ns  total %.2f ms" % (k, len(v), v[-1], sum(v[-16:]) / min(16, len(v)), sum(v) / 1e6))
