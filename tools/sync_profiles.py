"""copies the results that tools/collect_round_numbers.sh left under gpurun_out/final/ into profiles/ under this round's names
usage: python tools/sync_profiles.py r02"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out", "final")
dst = os.path.join(root, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


bench = json.load(open(os.path.join(src, "bench.json")))
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
algs = {}
for a in ("naive", "nee", "mis", "bdpt"):
    d = json.load(open(os.path.join(src, "alg_%s.json" % a)))
    algs[a] = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "rays_per_sample", "kernels_ms", "kernel_items") if k in d}
json.dump(algs, open(os.path.join(dst, tag + "_algorithms.json"), "w"), indent=1)
for name, out in (("ply_glass", "ply_scene_glass"), ("ply_gold", "ply_scene_gold"), ("hall", "hall_4k"), ("hall_long", "hall_4k_long_polygons"),
                  ("hall_long_whole", "hall_4k_long_polygons_uncut")):
    json.dump(last_json(os.path.join(src, name + ".json")), open(os.path.join(dst, "%s_%s.json" % (tag, out)), "w"), indent=1)
for line in open(os.path.join(src, "rehearsal.out")):
    if line.startswith("{"):
        json.dump(json.loads(line), open(os.path.join(dst, tag + "_bench_rehearsal_2ranks_one_gpu.json"), "w"), indent=1)
print("bench %.0f spp/s; " % bench["value"] + ", ".join("%s %.0f" % (a, algs[a]["value"]) for a in algs))
print("soups %.0f / %.0f Mrays/s" % (bench["aux"]["closest_hit_soup"]["mrays_per_s"], bench["aux"]["closest_hit_soup_10m"]["mrays_per_s"]))
for n in ("ply_scene_glass", "ply_scene_gold", "hall_4k", "hall_4k_long_polygons", "hall_4k_long_polygons_uncut"):
    d = json.load(open(os.path.join(dst, "%s_%s.json" % (tag, n))))
    print(n, "%.1f spp/s, build %.1f ms, %.2f Grays/s" % (d["spp_per_s"], d["bvh_build_s"] * 1e3, d["closest_grays_per_s_in_kernel"]))
