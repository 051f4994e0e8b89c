"""Bilateral filter timing at 1080p, image resident in HBM: python tools/bench_filter.py [window]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ray_tracer_amd import capi
window = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w, h = 1920, 1080
ctx = capi.Context(0)
img = torch.rand((h, w, 4), dtype=torch.float64, device="cuda")
out = torch.empty_like(img)
torch.cuda.synchronize()
for _ in range(2):
    ctx.filter_bilateral_device(img.data_ptr(), w, h, window, 3.0, 0.2, out.data_ptr())
ctx.render_sync()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    ctx.filter_bilateral_device(img.data_ptr(), w, h, window, 3.0, 0.2, out.data_ptr())
ctx.render_sync()
dt = (time.perf_counter() - t0) / reps
taps = w * h * (2 * window + 1) ** 2 * 3
res = {"window": window, "ms": dt * 1e3, "mpixels_per_s": w * h / dt / 1e6, "g_exp_per_s": taps / dt / 1e9,
       "streamed_gb_per_s": 2 * w * h * 32 / dt / 1e9}
# CPU side by side: the oracle's restatement of the reference loop on a crop (single thread, as the reference runs it)
import oracle
crop = img[:135, :240].cpu().numpy()
t0 = time.perf_counter()
ref = oracle.bilateral(crop, window, 3.0, 0.2)
cdt = time.perf_counter() - t0
res["cpu_port_mpixels_per_s"] = crop.shape[0] * crop.shape[1] / cdt / 1e6
inner = slice(window, 135 - window), slice(window, 240 - window)
res["max_rel_err_vs_oracle_on_crop_interior"] = float(np.abs(out[:135, :240].cpu().numpy()[inner][..., :3] / ref[inner][..., :3] - 1).max())
print(json.dumps(res))
