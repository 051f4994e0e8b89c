import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3),
      {k: round(v, 1) for k, v in d["kernels_ms"].items()}, "rays", d["roofline"]["rays"], "frac", d["roofline"]["frac"], "alg", (d["roofline"].get("algorithmic") or {}).get("bvh8", {}).get("rate_gbs"),
      "soup", (d["aux"]["closest_hit_soup"] or {}).get("mrays_per_s"), "soup10m", (d["aux"].get("closest_hit_soup_10m") or {}).get("mrays_per_s"))
