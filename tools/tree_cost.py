"""diagnostic: surface-area cost of the exported 8-wide tree (node visits and triangle tests expected for a random ray through the root box)
and the root's children, for the hall under either builder with and without cutting"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_amd import capi, scene_io, synth
paths = synth.atrium(os.path.join(tempfile.mkdtemp(), "hall"), long_polygons="--long-polygons" in sys.argv)
mesh = scene_io.Obj(); mesh.Load(paths["obj"]); mesh = mesh.to_mesh()
ctx = capi.Context(0)
for builder, bname in ((capi.BUILDER_DEVICE_LBVH, "device"), (capi.BUILDER_HOST_SAH, "host")):
    for budget in (0.0, 0.25):
        ctx.set_builder(builder); ctx.set_reference_cutting(budget); ctx.upload_mesh(mesh)
        wn, wt = ctx.wide_bvh()
        cell = np.ldexp(1.0, wn["e"].astype(np.int32) - 127)                      # [n, 3]
        lo = wn["org"][:, :, None] + wn["qlo"].astype(np.float64) * cell[:, :, None]  # [n, 3, 8]
        hi = wn["org"][:, :, None] + wn["qhi"].astype(np.float64) * cell[:, :, None]
        valid = (wn["qlo"] <= wn["qhi"]).all(1)                                    # [n, 8]
        d = np.where(valid[:, None, :], hi - lo, 0.0)
        area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]           # [n, 8] half areas of the children
        interior = ((wn["imask"][:, None] >> np.arange(8)) & 1).astype(bool) & valid
        count = (wn["meta"] >> 5).astype(np.float64)
        root = area[0].sum()  # not the root's own area, but a common scale
        rlo, rhi = lo[0][:, valid[0]].min(1), hi[0][:, valid[0]].max(1)
        rd = rhi - rlo; root = rd[0] * rd[1] + rd[1] * rd[2] + rd[2] * rd[0]
        print("%s %.2f: nodes %d refs %d | expected node visits %.2f, triangle tests %.2f | children per node %.2f, leaf children %.2f of them, triangles per leaf child %.2f"
              % (bname, budget, len(wn), len(wt), 1 + (area * interior).sum() / root, (area * count * (~interior & valid)).sum() / root,
                 valid.sum() / len(wn), (valid & ~interior).sum() / len(wn), count[valid & ~interior].mean()))
        for i in range(8):
            if valid[0][i]:
                print("      slot %d %s lo %s hi %s" % (i, "i" if interior[0][i] else "l", np.round(lo[0][:, i], 3), np.round(hi[0][:, i], 3)))
        k = np.argsort(-area[0])
        print("   root children (half area / root, interior): " + " ".join("%.3f%s" % (area[0][i] / root, "i" if interior[0][i] else "l") for i in k if valid[0][i]))
