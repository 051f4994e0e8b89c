// Per-polygon data in the reference's exact fp64 operation order, so that the plane equation the
// device evaluates (face normal, v[0]) carries the same bits as the reference's ConvexPolygon.
// Compiled with -ffp-contract=off.
#include "host_scene.h"
#include "parallel_for.h"
#include <cmath>
#include <mutex>

namespace mi355rt {

namespace {
struct d3 { double x, y, z; };
inline d3 ld(const double *p) { return {p[0], p[1], p[2]}; }
inline d3 sub(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       // vec3.h:120-124
inline d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x}; } // vec3.h:126-133
inline double len(d3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }                   // vec3.h:44-47
inline d3 unit(d3 a) { double l = len(a); return {a.x / l, a.y / l, a.z / l}; }                    // vec3.h:186-190
}

bool HostScene::build(const rt_mesh &m, std::string &err)
{
	if (!m.face_begin || !m.object_begin || (m.n_faces && !m.corners)) { err = "rt_mesh: null table"; return false; }
	n_faces = m.n_faces;
	n_objects = m.n_objects;
	n_corners = m.face_begin[n_faces];
	face_begin.assign(m.face_begin, m.face_begin + n_faces + 1);
	object_begin.assign(m.object_begin, m.object_begin + n_objects + 1);
	if (object_begin[0] != 0 || object_begin[n_objects] != n_faces) { err = "rt_mesh: object_begin does not cover the faces"; return false; }
	tri_begin.resize(n_faces + 1);
	uint64_t nt = 0;
	for (uint64_t f = 0; f < n_faces; f++) {
		if (face_begin[f + 1] < face_begin[f] + 3) { err = "rt_mesh: face " + std::to_string(f) + " has fewer than 3 corners"; return false; }
		tri_begin[f] = nt;
		nt += face_begin[f + 1] - face_begin[f] - 2;
	}
	tri_begin[n_faces] = nt;
	n_tris = nt;
	if (nt >= (1ull << 27)) { err = "rt_mesh: more than 2^27 fan triangles"; return false; }
	pv.resize(3 * n_corners); pvt.resize(3 * n_corners); pn.resize(3 * n_corners);
	face_normal.resize(3 * n_faces); polygon_area.resize(n_faces); face_object.resize(n_faces);
	tri_cum.resize(nt); tri_face.resize(nt);
	for (uint32_t o = 0; o < n_objects; o++) {
		if (object_begin[o + 1] < object_begin[o]) { err = "rt_mesh: object_begin not monotone"; return false; }
		for (uint64_t f = object_begin[o]; f < object_begin[o + 1]; f++) face_object[f] = o;
	}
	// polygons are independent of each other: the loop is split among host threads; the lowest bad corner is the one reported
	std::mutex bad_lock;
	uint64_t bad_corner = UINT64_MAX;
	parallel_for(n_faces, 1 << 14, [&](uint64_t f_begin, uint64_t f_end) {
		for (uint64_t f = f_begin; f < f_end; f++) {
			uint64_t b = face_begin[f], e = face_begin[f + 1];
			for (uint64_t k = b; k < e; k++) { // object.cpp:111-116
				int64_t iv = m.corners[3 * k], ivt = m.corners[3 * k + 1], ivn = m.corners[3 * k + 2];
				if (iv < 0 || (uint64_t)iv >= m.n_v || ivn < 0 || (uint64_t)ivn >= m.n_vn || (ivt >= 0 && (uint64_t)ivt >= m.n_vt)) {
					std::lock_guard<std::mutex> hold(bad_lock);
					bad_corner = std::min(bad_corner, k);
					return;
				}
				for (int c = 0; c < 3; c++) pv[3 * k + c] = m.v[3 * iv + c];
				for (int c = 0; c < 3; c++) pvt[3 * k + c] = ivt >= 0 ? m.vt[3 * ivt + c] : 0.0;
				d3 n = unit(ld(m.vn + 3 * ivn));
				pn[3 * k] = n.x; pn[3 * k + 1] = n.y; pn[3 * k + 2] = n.z;
			}
			d3 v0 = ld(&pv[3 * b]), v1 = ld(&pv[3 * (b + 1)]), v2 = ld(&pv[3 * (b + 2)]);
			d3 fn = unit(cross(sub(v1, v0), sub(v2, v1))); // object.cpp:117-120
			for (uint64_t k = b; k < e; k++)               // object.cpp:121-126: flip once per disagreeing vertex normal
				if (dot(fn, ld(&pn[3 * k])) < 0.0) { fn.x *= -1.0; fn.y *= -1.0; fn.z *= -1.0; }
			face_normal[3 * f] = fn.x; face_normal[3 * f + 1] = fn.y; face_normal[3 * f + 2] = fn.z;
			// CalcTriangleAreas, object.cpp:375-395
			double area = 0.0;
			uint64_t t0 = tri_begin[f], ntf = e - b - 2;
			for (uint64_t ti = 0; ti < ntf; ti++) {
				d3 a = sub(ld(&pv[3 * (b + ti + 1)]), v0), bb = sub(ld(&pv[3 * (b + ti + 2)]), v0);
				double tri_area = len(cross(a, bb)) / 2.0;
				tri_cum[t0 + ti] = ti == 0 ? tri_area : tri_cum[t0 + ti - 1] + tri_area;
				tri_face[t0 + ti] = (uint32_t)f;
				area += tri_area;
			}
			for (uint64_t ti = 0; ti < ntf; ti++) tri_cum[t0 + ti] /= area;
			tri_cum[t0 + ntf - 1] = 1.0;
			polygon_area[f] = area;
		}
	});
	if (bad_corner != UINT64_MAX) {
		err = "rt_mesh: corner " + std::to_string(bad_corner) + " indexes outside v/vt/vn (every corner needs a vn)";
		return false;
	}
	return true;
}

} // namespace mi355rt
