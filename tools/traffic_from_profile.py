"""Reads the FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ passes that tools/profile.sh left under gpurun_out/<tag>/ and
writes the per-launch HBM-side traffic of every k_* kernel as JSON (the figure bench.py reports as roofline.traffic).

Units and corrections: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE = read requests x 64 B
while every request moves one 128-byte line (checked here with tools/calib/random_lines.hip: one request per random
128-byte line whether 64 or 128 bytes of it are used, and with a plain streamed read of a known size), so read bytes =
2 x FETCH_SIZE.  WRITE_SIZE matches streamed 16-byte stores and float atomics as it stands.  Infinity-Cache hits are
included in both: this is traffic on the memory side of the L2, an upper bound of what reaches HBM."""
import csv, glob, json, os, sys
from collections import defaultdict

tag_dir, out_path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
rows = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(tag_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
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
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh, read bytes = 2 x FETCH_SIZE (gfx950)",
           "command": " ".join(sys.argv[3:]), "kernels": res}, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
