#include "presplit.h"
#include "accept_region.h"
#include "parallel_for.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

namespace mi355rt {
namespace {

constexpr double kMissesPerHit = 2.0; // a piece is worth its reference while this many rays miss it per ray that hits it
constexpr int kGridBits = 21; // the spatial-median planes are those of the device builder's Morton grid (21 bits per axis)

inline float round_down(double x)
{
	float f = (float)x;
	if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
	return f;
}
inline float round_up(double x)
{
	float f = (float)x;
	if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
	return f;
}

struct Grid {
	double lo[3], ext[3];
	// cell of coordinate x on axis a, 0 .. 2^21 - 1
	uint32_t cell(int a, double x) const
	{
		if (!(ext[a] > 0.0)) return 0u;
		const double u = (x - lo[a]) / ext[a] * (double)(1u << kGridBits);
		if (!(u > 0.0)) return 0u;
		return u >= (double)((1u << kGridBits) - 1u) ? (1u << kGridBits) - 1u : (uint32_t)u;
	}
	double plane(int a, uint32_t q) const { return lo[a] + (double)q / (double)(1u << kGridBits) * ext[a]; }
};

// The most important spatial-median plane strictly inside [lo, hi] on any axis: level 1 halves the scene, level 2 quarters it, ...
// Returns the level (kGridBits + 1 when no plane of the grid crosses the box) and the axis and position of the plane.
inline int important_plane(const Grid &g, const double *lo, const double *hi, int &axis, double &pos)
{
	int best = kGridBits + 1;
	double best_ext = -1.0;
	for (int a = 0; a < 3; a++) {
		// neighbouring pieces overlap by two float steps around the plane between them: a plane that close to the end of the box is
		// the one the box was cut at
		const double margin = 4.0 * 1.1920928955078125e-7 * std::max(std::fabs(lo[a]), std::fabs(hi[a])) + 1e-37;
		const double l = lo[a] + margin, h = hi[a] - margin;
		if (!(l < h)) continue;
		const uint32_t q0 = g.cell(a, l), q1 = g.cell(a, h);
		const uint32_t x = q0 ^ q1;
		if (x == 0u) continue;
		const int top = 31 - __builtin_clz(x);            // highest bit in which the two cells differ
		const uint32_t q = (q1 >> top) << top;            // the first cell of the upper half at that level
		const double p = g.plane(a, q);
		if (!(p > l && p < h)) continue;                  // the rounding of the cell arithmetic can put the plane on the boundary
		const int level = kGridBits - top;
		const double ext = hi[a] - lo[a];
		if (level < best || (level == best && ext > best_ext)) { best = level; best_ext = ext; axis = a; pos = p; }
	}
	return best;
}

// bounds of (triangle clipped against the box [rlo, rhi]) joined into [lo, hi]; false when nothing is left
bool clip_bounds(const double tri[3][3], const double *rlo, const double *rhi, double *lo, double *hi)
{
	double poly[2][12][3];
	int n = 3, cur = 0;
	for (int v = 0; v < 3; v++)
		for (int k = 0; k < 3; k++) poly[0][v][k] = tri[v][k];
	for (int a = 0; a < 3 && n > 0; a++)
		for (int side = 0; side < 2 && n > 0; side++) {
			const double bound = side ? rhi[a] : rlo[a];
			int m = 0;
			for (int i = 0; i < n; i++) {
				const double *p = poly[cur][i], *q = poly[cur][(i + 1) % n];
				const bool pin = side ? p[a] <= bound : p[a] >= bound, qin = side ? q[a] <= bound : q[a] >= bound;
				if (pin) { for (int k = 0; k < 3; k++) poly[cur ^ 1][m][k] = p[k]; m++; }
				if (pin != qin) {
					const double t = (bound - p[a]) / (q[a] - p[a]);
					for (int k = 0; k < 3; k++) poly[cur ^ 1][m][k] = k == a ? bound : p[k] + t * (q[k] - p[k]);
					m++;
				}
			}
			n = m;
			cur ^= 1;
		}
	if (n == 0) return false;
	for (int i = 0; i < n; i++)
		for (int k = 0; k < 3; k++) {
			lo[k] = std::min(lo[k], poly[cur][i][k]);
			hi[k] = std::max(hi[k], poly[cur][i][k]);
		}
	return true;
}

struct Shape {
	double tris[3][3][3]; // the triangle, the region's corners in its plane, the corners lifted onto the polygon's plane
	int n_tris;
	double pad;           // against the rounding of the clipping
};

// tight bounds of the shape inside the region, padded, and no larger than the region; false when the shape does not reach into it
bool piece_bounds(const Shape &s, const double *rlo, const double *rhi, double *lo, double *hi)
{
	const double inf = std::numeric_limits<double>::infinity();
	for (int k = 0; k < 3; k++) { lo[k] = inf; hi[k] = -inf; }
	bool any = false;
	for (int i = 0; i < s.n_tris; i++) any = clip_bounds(s.tris[i], rlo, rhi, lo, hi) || any;
	if (!any) return false;
	for (int k = 0; k < 3; k++) {
		lo[k] = std::max(lo[k] - s.pad, rlo[k]);
		hi[k] = std::min(hi[k] + s.pad, rhi[k]);
	}
	return true;
}

struct Item {
	double rlo[3], rhi[3]; // the part of space this piece answers for
	double lo[3], hi[3];   // bounds of the shape inside it
	uint32_t splits;
};

// cuts triangle t into at most splits + 1 pieces; out[0 .. return value) receives them
uint32_t cut_triangle(const HostScene &scene, const Grid &grid, uint32_t t, uint32_t splits, SplitRef *out)
{
	const double *v0, *v1, *v2;
	scene.tri_vertices(t, v0, v1, v2);
	const double *nf = &scene.face_normal[3 * (size_t)scene.tri_face[t]];
	Shape sh;
	double flat[3][3], lifted[3][3];
	const int kind = accept_region_corners(v0, v1, v2, nf, flat, lifted);
	if (kind < 0) return 0; // the region is only known as a box
	for (int k = 0; k < 3; k++) { sh.tris[0][0][k] = v0[k]; sh.tris[0][1][k] = v1[k]; sh.tris[0][2][k] = v2[k]; }
	sh.n_tris = 1;
	if (kind > 0) {
		for (int v = 0; v < 3; v++)
			for (int k = 0; k < 3; k++) { sh.tris[1][v][k] = flat[v][k]; sh.tris[2][v][k] = lifted[v][k]; }
		sh.n_tris = 3;
	}
	double big = 0.0;
	for (int i = 0; i < sh.n_tris; i++)
		for (int v = 0; v < 3; v++)
			for (int k = 0; k < 3; k++) {
				if (!(std::fabs(sh.tris[i][v][k]) < 1e300)) return 0; // not finite
				big = std::max(big, std::fabs(sh.tris[i][v][k]));
			}
	sh.pad = big * 1e-13; // a clipped vertex is within a few units of the last place of its inputs; 2^-43 of the largest coordinate is ample
	// the whole reference: the box the builders give the uncut triangle
	double blo[3], bhi[3];
	accept_region_box(v0, v1, v2, nf, blo, bhi);
	Item stack[64];
	int sp = 0;
	Item first;
	for (int k = 0; k < 3; k++) { first.rlo[k] = first.lo[k] = (double)round_down(blo[k]); first.rhi[k] = first.hi[k] = (double)round_up(bhi[k]); }
	first.splits = splits;
	stack[sp++] = first;
	uint32_t n_out = 0;
	auto emit = [&](const Item &it) {
		SplitRef r;
		r.tri = t;
		for (int k = 0; k < 3; k++) { r.lo[k] = round_down(it.lo[k]); r.hi[k] = round_up(it.hi[k]); }
		out[n_out++] = r;
	};
	while (sp > 0) {
		const Item it = stack[--sp];
		int axis = 0;
		double pos = 0.0;
		if (it.splits == 0 || sp + 2 > 64 || important_plane(grid, it.lo, it.hi, axis, pos) > kGridBits) { emit(it); continue; }
		// the two parts of space overlap by two float steps around the plane
		const float inf = std::numeric_limits<float>::infinity();
		const double up = (double)std::nextafterf(round_up(pos), inf), down = (double)std::nextafterf(round_down(pos), -inf);
		Item l = it, r = it;
		l.rhi[axis] = std::min(it.rhi[axis], up);
		r.rlo[axis] = std::max(it.rlo[axis], down);
		if (!piece_bounds(sh, l.rlo, l.rhi, l.lo, l.hi) || !piece_bounds(sh, r.rlo, r.rhi, r.lo, r.hi)) { emit(it); continue; }
		// a cut that does not shorten either part (a plane within two float steps of the end of the box) would not end
		if (!(l.hi[axis] - l.lo[axis] < it.hi[axis] - it.lo[axis]) || !(r.hi[axis] - r.lo[axis] < it.hi[axis] - it.lo[axis])) { emit(it); continue; }
		double wl = 0.0, wr = 0.0;
		for (int k = 0; k < 3; k++) { wl = std::max(wl, l.hi[k] - l.lo[k]); wr = std::max(wr, r.hi[k] - r.lo[k]); }
		const uint32_t rest = it.splits - 1u;
		uint32_t sl = wl + wr > 0.0 ? (uint32_t)std::floor((double)rest * wl / (wl + wr) + 0.5) : rest / 2u;
		if (sl > rest) sl = rest;
		l.splits = sl;
		r.splits = rest - sl;
		stack[sp++] = r;
		stack[sp++] = l;
	}
	return n_out;
}

} // namespace

void presplit(const HostScene &scene, const PresplitOptions &opt, Presplit &out)
{
	const auto t_start = std::chrono::steady_clock::now();
	out = Presplit();
	const uint64_t n = scene.n_tris;
	if (n < opt.min_tris || !(opt.budget > 0.0f) || opt.max_pieces < 2) return;
	// ---- the grid: the box of the triangles' box centres, as the device builder's Morton keys use it
	const double inf = std::numeric_limits<double>::infinity();
	const unsigned kParts = 64;
	std::vector<double> part_lo(3 * kParts, inf), part_hi(3 * kParts, -inf);
	parallel_for(kParts, 1, [&](uint64_t p_begin, uint64_t p_end) {
		for (uint64_t p = p_begin; p < p_end; p++)
			for (uint64_t t = n * p / kParts; t < n * (p + 1) / kParts; t++) {
				const double *v0, *v1, *v2;
				scene.tri_vertices(t, v0, v1, v2);
				for (int k = 0; k < 3; k++) {
					const double c = 0.5 * (std::min(v0[k], std::min(v1[k], v2[k])) + std::max(v0[k], std::max(v1[k], v2[k])));
					if (c == c) { part_lo[3 * p + k] = std::min(part_lo[3 * p + k], c); part_hi[3 * p + k] = std::max(part_hi[3 * p + k], c); }
				}
			}
	});
	Grid grid;
	for (int k = 0; k < 3; k++) {
		double lo = inf, hi = -inf;
		for (unsigned p = 0; p < kParts; p++) { lo = std::min(lo, part_lo[3 * p + k]); hi = std::max(hi, part_hi[3 * p + k]); }
		if (!(lo <= hi) || !(hi - lo < 1e300)) return; // no finite triangle
		grid.lo[k] = lo;
		grid.ext[k] = hi - lo;
	}
	// ---- how much space each triangle's box wastes: the box's surface less the smallest surface a box around the triangle could
	// have if boxes could turn (twice its projections).  The two fan triangles of an axis-aligned rectangle waste half their
	// boxes each, but a ray through the wasted half of one hits the other: what counts there is the waste of the polygon's box,
	// which is nothing.  A triangle's waste is therefore the smaller of its own and its polygon's.
	double misses_per_hit = kMissesPerHit;
	if (const char *e = std::getenv("MI355RT_CUT_RATIO")) misses_per_hit = std::atof(e); // experiments
	std::vector<float> waste(n), most(n); // most: the largest number of pieces that pays (below)
	parallel_for(scene.n_faces, 1 << 14, [&](uint64_t f_begin, uint64_t f_end) {
		for (uint64_t f = f_begin; f < f_end; f++) {
			const uint64_t t0 = scene.tri_begin[f], t1 = scene.tri_begin[f + 1];
			double plo[3] = {inf, inf, inf}, phi[3] = {-inf, -inf, -inf}, p_ideal = 0.0;
			for (uint64_t t = t0; t < t1; t++) {
				const double *v0, *v1, *v2;
				scene.tri_vertices(t, v0, v1, v2);
				double lo[3], hi[3], e0[3], e1[3];
				for (int k = 0; k < 3; k++) {
					lo[k] = std::min(v0[k], std::min(v1[k], v2[k]));
					hi[k] = std::max(v0[k], std::max(v1[k], v2[k]));
					plo[k] = std::min(plo[k], lo[k]);
					phi[k] = std::max(phi[k], hi[k]);
					e0[k] = v1[k] - v0[k];
					e1[k] = v2[k] - v0[k];
				}
				const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
				const double ideal = std::fabs(e0[1] * e1[2] - e0[2] * e1[1]) + std::fabs(e0[2] * e1[0] - e0[0] * e1[2]) + std::fabs(e0[0] * e1[1] - e0[1] * e1[0]);
				const double w = 2.0 * (dx * dy + dy * dz + dz * dx) - ideal;
				p_ideal += ideal;
				waste[t] = w > 0.0 && w < 1e300 ? (float)w : 0.0f;
				most[t] = !(w > 0.0) ? 0.0f : (ideal > 0.0 ? (float)std::min(1e9, w / (misses_per_hit * ideal)) : 1e9f);
			}
			const double dx = phi[0] - plo[0], dy = phi[1] - plo[1], dz = phi[2] - plo[2];
			const double pw = 2.0 * (dx * dy + dy * dz + dz * dx) - p_ideal;
			const float cap = pw > 0.0 && pw < 1e300 ? (float)pw : 0.0f;
			for (uint64_t t = t0; t < t1; t++) waste[t] = std::min(waste[t], cap);
		}
	});
	// (`ideal` is also the measure of the rays that hit the triangle, `waste` of those that enter its box and miss.  A hit on a cut
	// triangle pays for the deeper tree, a miss is what cutting saves: a piece's misses, waste / k^2, must stay above kMissesPerHit
	// times its hits, ideal / k, so k <= waste / (kMissesPerHit * ideal).  The two triangles of a wall, however large, have as many
	// hits as misses and are never cut -- cutting the walls of a Cornell box around a scanned object cost that scene 19 % --
	// while a brace thirty times as long as wide is cut into pieces four times as long as wide.)
	// ---- pieces per triangle.  Cut into k pieces, a long triangle's boxes waste about 1 / k of what the whole box wastes, so the k-th
	// piece saves waste / k^2: pieces are added while that is more than kWorth times what a typical triangle of the scene wastes,
	// k = sqrt(waste / (kWorth * typical)).  "Typical" is the cube of the mean cube root: a few huge triangles do not move it.
	// A scene whose triangles are all of a size gets no cuts at all (k < 2 up to 16 times the typical waste: four times the size).
	// sums over fixed parts of the triangles, added up in order: the same result whatever the number of threads
	auto sum_over = [&](auto term) {
		std::vector<double> part(kParts, 0.0);
		parallel_for(kParts, 1, [&](uint64_t p_begin, uint64_t p_end) {
			for (uint64_t p = p_begin; p < p_end; p++) {
				double acc = 0.0;
				for (uint64_t t = n * p / kParts; t < n * (p + 1) / kParts; t++) acc += term(t);
				part[p] = acc;
			}
		});
		double total = 0.0;
		for (unsigned p = 0; p < kParts; p++) total += part[p];
		return total;
	};
	const double sum_cbrt = sum_over([&](uint64_t t) { return std::cbrt((double)waste[t]); });
	const double mean_cbrt = sum_cbrt / (double)n, typical = mean_cbrt * mean_cbrt * mean_cbrt;
	if (!(typical > 0.0)) return;
	double kWorth = 4.0;
	if (const char *e = std::getenv("MI355RT_CUT_WORTH")) kWorth = std::atof(e); // experiments
	auto pieces_of = [&](uint64_t t, double shrink) {
		const double k = std::floor(std::min(std::sqrt((double)waste[t] / (kWorth * typical)), (double)most[t]) * shrink);
		return k < 2.0 ? 1u : (uint32_t)std::min<double>(k, (double)opt.max_pieces);
	};
	// the budget: if the scene asks for more than it allows, every triangle's share shrinks by the same factor
	const double allowed = (double)opt.budget * (double)n;
	double shrink = 1.0;
	for (int round = 0; round < 40; round++) {
		const float least = (float)(4.0 * kWorth * typical / (shrink * shrink) * 0.999); // below this waste a triangle stays whole
		const double total = sum_over([&](uint64_t t) { return waste[t] < least ? 0.0 : (double)(pieces_of(t, shrink) - 1u); });
		if (total <= allowed) break;
		shrink *= std::max(0.5, std::min(0.95, allowed / total));
	}
	if (std::getenv("MI355RT_DEBUG")) { float wmax = 0; for (uint64_t t = 0; t < n; t++) wmax = std::max(wmax, waste[t]); std::fprintf(stderr, "[mi355rt] presplit: typical waste %g, largest %g, shrink %g\n", typical, (double)wmax, shrink); }
	std::vector<uint32_t> chosen, splits;
	std::vector<uint64_t> offset;
	uint64_t room = 0;
	const float least = (float)(4.0 * kWorth * typical / (shrink * shrink) * 0.999);
	for (uint64_t t = 0; t < n; t++) {
		if (waste[t] < least) continue;
		const uint32_t k = pieces_of(t, shrink);
		if (k >= 2u) {
			chosen.push_back((uint32_t)t);
			splits.push_back(k - 1u);
			offset.push_back(room);
			room += k;
		}
	}
	if (chosen.empty()) return;
	std::vector<SplitRef> pieces(room);
	std::vector<uint32_t> made(chosen.size());
	parallel_for(chosen.size(), 64, [&](uint64_t j_begin, uint64_t j_end) {
		for (uint64_t j = j_begin; j < j_end; j++) made[j] = cut_triangle(scene, grid, chosen[j], splits[j], &pieces[offset[j]]);
	});
	for (size_t j = 0; j < chosen.size(); j++) {
		if (made[j] < 2) continue; // left whole
		out.first.push_back(pieces[offset[j]]);
		for (uint32_t k = 1; k < made[j]; k++) out.extra.push_back(pieces[offset[j] + k]);
	}
	out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
}

} // namespace mi355rt
