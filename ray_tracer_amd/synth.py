"""Synthetic benchmark inputs: random-triangle soups and incoherent rays.

Both come from a counter-based generator, u(k) = (splitmix64(seed * 0x9E3779B97F4A7C15 + k) >> 11) * 2^-53,
so every process (and the CPU checker) regenerates identical inputs without files.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(seed, start, count):
    """u(k) for k = start .. start+count-1, float64 in [0, 1)."""
    with np.errstate(over="ignore"):
        k = np.arange(start, start + count, dtype=np.uint64)
        z = _mix(np.uint64(seed) * _GOLD + k)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def soup(n, sigma, seed=12345, chunk=1 << 20):
    """Soup(N, sigma, seed): triangle i has centre c_i ~ U[-1,1]^3 and vertices c_i + sigma*N(0, I3).

    Sixteen counters per triangle: 3 for the centre, 12 for six Box-Muller pairs (nine normals used).
    Returns the rt_mesh tables: one object, one face per triangle, a flat normal per face."""
    v = np.empty((n, 3, 3), dtype=np.float64)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        u = uniform(seed, s * 16, m * 16).reshape(m, 16)
        c = 2.0 * u[:, 0:3] - 1.0
        u1 = np.maximum(u[:, 3:9], 1e-300)
        u2 = u[:, 9:15]
        r = np.sqrt(-2.0 * np.log(u1))
        g = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], axis=1)[:, :9]
        v[s:s + m] = c[:, None, :] + sigma * g.reshape(m, 3, 3)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 1]
    nrm = np.cross(e1, e2)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    idx = np.arange(3 * n, dtype=np.int64)
    corners = np.stack([idx, np.full(3 * n, -1, dtype=np.int64), idx // 3], axis=1)
    return {
        "v": v.reshape(-1, 3), "vt": np.zeros((0, 3)), "vn": nrm, "corners": corners,
        "face_begin": np.arange(0, 3 * n + 1, 3, dtype=np.uint64),
        "object_begin": np.array([0, n], dtype=np.uint64), "names": ["soup"],
    }


def incoherent_rays(m, seed=777, t_min=1e-3, t_max=np.inf, start=0, chunk=1 << 22):
    """Rays(M, seed): origin U[-1,1]^3, direction uniform on the sphere (z = 1-2u1, phi = 2 pi u2).
    Six counters per ray (one spare).  float32 [M, 8] = ox oy oz tmin dx dy dz tmax."""
    out = np.empty((m, 8), dtype=np.float32)
    for s in range(0, m, chunk):
        k = min(chunk, m - s)
        u = uniform(seed, (start + s) * 6, k * 6).reshape(k, 6)
        z = 1.0 - 2.0 * u[:, 3]
        r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        phi = 2.0 * np.pi * u[:, 4]
        out[s:s + k, 0:3] = 2.0 * u[:, 0:3] - 1.0
        out[s:s + k, 3] = t_min
        out[s:s + k, 4] = r * np.cos(phi)
        out[s:s + k, 5] = r * np.sin(phi)
        out[s:s + k, 6] = z
        out[s:s + k, 7] = t_max
    return out


# ----------------------------------------------------------------------------------------------
# Procedural stand-ins for the scanned / modelled scenes of the benchmark configurations (no model
# files are available offline): a displaced sphere of scan-like triangle count, and a multi-object
# OBJ project with quads and triangles, several lights and every BSDF node type.
# ----------------------------------------------------------------------------------------------
def blob(n_lat, n_lon, centre, radius):
    """displaced sphere: (vertices [n, 3], triangles [m, 3]); 660 x 660 gives 869 880 triangles"""
    th = np.linspace(0.01, np.pi - 0.01, n_lat)
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = radius * (1 + 0.10 * np.sin(9 * T) * np.cos(7 * P) + 0.03 * np.sin(40 * T + 3 * P))
    v = np.stack([r * np.sin(T) * np.cos(P), r * np.cos(T), r * np.sin(T) * np.sin(P)], -1).reshape(-1, 3) + np.asarray(centre)
    idx = np.arange(n_lat * n_lon).reshape(n_lat, n_lon)
    a, b = idx[:-1, :], idx[1:, :]
    a2, b2 = np.roll(a, -1, axis=1), np.roll(b, -1, axis=1)
    return v, np.concatenate([np.stack([a, b, b2], -1).reshape(-1, 3), np.stack([a, b2, a2], -1).reshape(-1, 3)])


def _grid_quads(nu, nv, wrap_u):
    """quad corner indices [m, 4] of an nu x nv vertex grid (u fastest); wrap_u closes the grid in u"""
    i = np.arange(nv - 1)[:, None] * nu
    j = np.arange(nu if wrap_u else nu - 1)[None, :]
    j1 = (j + 1) % nu
    return np.stack([i + j, i + j1, i + nu + j1, i + nu + j], -1).reshape(-1, 4)


def _material_json(name, node_type, light=False, **arrays):
    slots = {"Lambertian": ["albedo", "normal"], "DiffuseLight": ["color"], "Dielectric": ["surface color"], "Conductor": [], "ColoredMetal": ["albedo"]}
    node = {"type": node_type, "name": node_type, "pos": {"x": 165.0, "y": 200.0},
            "input_slots": [{"name": s, "type": "SlotSpectrum"} for s in slots[node_type]], "output_slots": [{"name": "BSDF", "type": "SlotBSDF"}]}
    for k, a in arrays.items():
        node[k] = [float(x) for x in np.broadcast_to(np.asarray(a, dtype=np.float64), (30,))]
    return {"name": name, "light_flag": bool(light),
            "nodes": [{"type": "UV", "name": "", "pos": {"x": 86.0, "y": 37.0}, "input_slots": [], "output_slots": [{"name": "Out->", "type": "SlotVec3"}]},
                      {"type": "Output", "name": "Output", "pos": {"x": 407.0, "y": 70.0}, "input_slots": [{"name": "BSDF", "type": "SlotBSDF"}], "output_slots": []},
                      node],
            "connections": [{"input_node": 1, "input_slot": "BSDF", "output_node": 2, "output_slot": "BSDF"}]}


def atrium(directory, detail=1.0, seed=7, long_polygons=False):
    """Writes a hall-like project (model.obj, materials.json, objmatmap.json) into `directory` and returns their paths and the
    polygon count.  At detail 1: 31 objects, about 320 k polygons (170 k quads, of which 58 k are not planar, 150 k triangles; 490 k fan
    triangles): a displaced floor and side walls (quads that are not planar), flat walls and ceiling (quads), two rows of fluted columns (quads), displaced spheres (triangles)
    of glass, metal and paint, and six ceiling light panels of two quads each.  Smooth vertex normals everywhere (every corner carries a vn index).
    long_polygons: the flat sheets (ceiling, back and front wall) are strips that run the sheet's whole length instead of grid cells, and 96
    thin braces (quads 3 cm wide, 3 to 9 m long) cross the hall diagonally between the columns: the long thin polygons of a modelled
    building, whose boxes cover far more space than they do."""
    import json
    import os
    rng = np.random.default_rng(seed)
    lam = 405.0 + 10.0 * np.arange(30)
    objects = []  # (name, material, vertices, normals, faces as [m, k] index array into this object's vertices)

    def add(name, material, v, n, faces):
        n = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
        objects.append((name, material, np.asarray(v, dtype=np.float64), n, np.asarray(faces, dtype=np.int64)))

    def sheet(name, material, origin, eu, ev, nu, nv, bump):
        """nu x nv grid spanned by eu, ev from origin, displaced along the normal"""
        u, v = np.meshgrid(np.linspace(0, 1, nu), np.linspace(0, 1, nv), indexing="xy")
        nrm = np.cross(eu, ev)
        nrm = nrm / np.linalg.norm(nrm)
        h = bump * (np.sin(17 * u + 3 * v) * np.cos(13 * v) + 0.3 * np.sin(71 * u * v))
        p = origin + u[..., None] * eu + v[..., None] * ev + h[..., None] * nrm
        du, dv = np.gradient(p, axis=1), np.gradient(p, axis=0)
        # quads also where the sheet is displaced: such a cell is not planar, and the reference intersects both of its fan triangles
        # with the plane of its first three vertices (object.cpp:244), as a modelled scene's slightly warped quads are
        if long_polygons and bump == 0.0:
            # one strip per column of cells: the first and the last row of the grid
            top = (nv - 1) * nu
            faces = np.stack([np.arange(nu - 1), np.arange(nu - 1) + 1, top + np.arange(nu - 1) + 1, top + np.arange(nu - 1)], -1)
        else:
            faces = _grid_quads(nu, nv, False)
        add(name, material, p.reshape(-1, 3), np.cross(du, dv).reshape(-1, 3), faces)

    g = max(8, int(140 * np.sqrt(detail)))
    X, Y, Z = 4.0, 2.5, 8.0  # half width, height, half depth of the hall
    # spanning vectors are ordered so that eu x ev points into the hall: the reference shades one side only (no normal flipping)
    sheet("floor", "stone", np.array([-X, 0.0, Z]), np.array([2 * X, 0, 0]), np.array([0, 0, -2 * Z]), g, 2 * g, 0.01)
    sheet("ceiling", "plaster", np.array([-X, Y, -Z]), np.array([2 * X, 0, 0]), np.array([0, 0, 2 * Z]), g // 2, g, 0.0)
    sheet("wall_left", "red_paint", np.array([-X, 0.0, Z]), np.array([0, 0, -2 * Z]), np.array([0, Y, 0]), g, g // 2, 0.004)
    sheet("wall_right", "green_paint", np.array([X, 0.0, -Z]), np.array([0, 0, 2 * Z]), np.array([0, Y, 0]), g, g // 2, 0.004)
    sheet("wall_back", "plaster", np.array([-X, 0.0, -Z]), np.array([2 * X, 0, 0]), np.array([0, Y, 0]), g // 2, g // 2, 0.0)
    sheet("wall_front", "plaster", np.array([X, 0.0, Z]), np.array([-2 * X, 0, 0]), np.array([0, Y, 0]), g // 2, g // 2, 0.0)
    # columns: closed grids of quads, fluted
    seg, rings = max(12, int(96 * np.sqrt(detail))), max(8, int(80 * np.sqrt(detail)))
    for k in range(12):
        cx, cz = (-2.4 if k % 2 == 0 else 2.4), -6.5 + (k // 2) * 2.6
        a = np.linspace(0, 2 * np.pi, seg, endpoint=False)
        y = np.linspace(0.0, Y, rings)
        A, H = np.meshgrid(a, y, indexing="xy")
        r = 0.22 * (1 + 0.06 * np.cos(12 * A)) * (1 + 0.25 * np.exp(-30 * H) + 0.25 * np.exp(-30 * (Y - H)))
        p = np.stack([cx + r * np.cos(A), H, cz + r * np.sin(A)], -1)
        nrm = np.stack([np.cos(A), 0 * A, np.sin(A)], -1)
        add("column.%03d" % k, "marble" if k % 3 else "bronze", p.reshape(-1, 3), nrm.reshape(-1, 3), _grid_quads(seg, rings, True)[:, ::-1])
    # displaced spheres (triangles)
    sph = max(12, int(105 * np.sqrt(detail)))
    for k, mat in enumerate(["glass", "gold", "blue_paint", "glass", "copper", "stone", "gold"]):
        c = np.array([rng.uniform(-1.6, 1.6), 0.45 + 0.25 * (k % 3), -5.5 + 1.7 * k])
        v, tris = blob(sph, sph, c, 0.38 + 0.05 * (k % 2))
        add("orb.%03d" % k, mat, v, v - c, tris[:, ::-1])
    if long_polygons:
        # braces: thin quads between points on the two column rows, at all heights and slopes
        ends = rng.uniform([-2.4, 0.1, -6.5], [-2.4, Y - 0.1, 6.5], (96, 3))
        other = rng.uniform([2.4, 0.1, -6.5], [2.4, Y - 0.1, 6.5], (96, 3))
        other[:, 2] = np.clip(ends[:, 2] + rng.uniform(-6.0, 6.0, 96), -6.5, 6.5)
        side = np.array([0.0, 0.03, 0.0])
        v = np.stack([ends, other, other + side, ends + side], 1).reshape(-1, 3)
        nrm = np.cross(other - ends, side)
        add("braces", "bronze", v, np.repeat(nrm, 4, axis=0), np.arange(4 * 96).reshape(96, 4))
    # light panels just below the ceiling, facing down: two quads each
    for k in range(6):
        cx, cz = (-1.5 if k % 2 == 0 else 1.5), -5.0 + (k // 2) * 5.0
        p = np.array([[cx - 0.5, Y - 0.02, cz - 0.8], [cx, Y - 0.02, cz - 0.8], [cx + 0.5, Y - 0.02, cz - 0.8],
                      [cx - 0.5, Y - 0.02, cz + 0.8], [cx, Y - 0.02, cz + 0.8], [cx + 0.5, Y - 0.02, cz + 0.8]])
        add("panel.%03d" % k, "lamp", p, np.tile([0.0, -1.0, 0.0], (6, 1)), np.array([[0, 1, 4, 3], [1, 2, 5, 4]]))

    os.makedirs(directory, exist_ok=True)
    paths = {k: os.path.join(directory, f) for k, f in (("obj", "model.obj"), ("materials", "materials.json"), ("objmatmap", "objmatmap.json"))}
    n_polygons, base = 0, 0
    with open(paths["obj"], "w") as f:
        f.write("# procedural hall: quads and triangles, one vn per vertex\n")
        for name, material, v, n, faces in objects:
            f.write("o %s\n" % name)
            f.write("".join("v %.9g %.9g %.9g\n" % tuple(x) for x in v))
            f.write("".join("vn %.9g %.9g %.9g\n" % tuple(x) for x in n))
            f.write("usemtl %s\ns 1\n" % material)
            idx = faces + base + 1
            f.write("\n".join("f " + " ".join("%d//%d" % (i, i) for i in row) for row in idx) + "\n")
            base += len(v)
            n_polygons += len(faces)
    gold_n, gold_k = np.interp(lam, [400, 500, 550, 700], [1.45, 0.95, 0.35, 0.13]), np.interp(lam, [400, 500, 550, 700], [1.95, 1.85, 2.60, 3.90])
    mats = [
        _material_json("stone", "Lambertian", albedo=0.55), _material_json("plaster", "Lambertian", albedo=0.8),
        _material_json("marble", "Lambertian", albedo=0.9),
        _material_json("red_paint", "Lambertian", albedo=np.where(lam > 590, 0.75, 0.05)),
        _material_json("green_paint", "Lambertian", albedo=np.where((lam > 490) & (lam < 580), 0.7, 0.06)),
        _material_json("blue_paint", "Lambertian", albedo=np.where(lam < 490, 0.7, 0.08)),
        _material_json("glass", "Dielectric", n=1.5 + 0.02 * (550.0 - lam) / 150.0, surface_color=0.97),
        _material_json("gold", "Conductor", n=gold_n, k=gold_k),
        _material_json("bronze", "ColoredMetal", albedo=np.interp(lam, [400, 550, 700], [0.25, 0.55, 0.8])),
        _material_json("copper", "ColoredMetal", albedo=np.interp(lam, [400, 580, 700], [0.3, 0.5, 0.95])),
        _material_json("lamp", "DiffuseLight", light=True, color=12.0),
    ]
    with open(paths["materials"], "w") as f:
        json.dump(mats, f)
    with open(paths["objmatmap"], "w") as f:
        json.dump({name: material for name, material, _, _, _ in objects}, f)
    paths["polygons"] = n_polygons
    paths["objects"] = len(objects)
    return paths
