// Where can the reference's fan-triangle test accept a hit point?  (host and device, fp64)
//
// ConvexPolygon::HitTriangle (reference object.cpp:238-253) tests every fan triangle (a, b, c) = (v[0], v[k+1], v[k+2]) of a polygon
// at the point p where the ray meets the plane P0 through v[0] with the polygon's face normal -- the normal of the polygon's FIRST
// three vertices (object.cpp:117-120, 244) -- and accepts when
//     r0 . r1 > 0  and  r1 . r2 > 0,     r0 = (b - a) x (p - b),  r1 = (c - b) x (p - c),  r2 = (a - c) x (p - a).
// For a polygon whose vertices are not coplanar, p does not lie in the plane of the later fan triangles, and the accepted points are
// not the triangle: a tree whose leaf box is the box of the triangle's vertices may or may not find such a hit, depending on what
// it visits and prunes.  This header gives a box that contains every acceptable p, so that the closest hit found through the tree
// is the closest hit a loop over all polygons finds.
//
// Derivation.  Write p = x + h n, x the projection of p onto the triangle's plane, n the triangle's unit normal, h the height.  Then
// r0 = alpha n + h (b - a) x n, and likewise r1 (beta), r2 (gamma), where alpha, beta, gamma are twice the signed areas of
// (a, b, x), (b, c, x), (c, a, x): with the triangle's doubled area 2A they are 2A times the barycentric coordinates lc, la, lb of x.
// The in-plane parts are the edge vectors turned by 90 degrees, so their dot products are those of the edges:
//     r0 . r1 = 4A^2 la lc + h^2 (b - a).(c - b),      r1 . r2 = 4A^2 la lb + h^2 (c - b).(a - c).
// With (b - a).(c - b) = -2A cot B, (c - b).(a - c) = -2A cot C (B, C the angles at b and c) and u = h^2 / 2A >= 0 the test reads
//     la lc > u cot B     and     la lb > u cot C.                                                         (*)
// B + C < 180 degrees, so at most one of the two cotangents is negative, and cot C > -cot B, cot B > -cot C.
//   * cot B >= 0 and cot C >= 0: la, lb, lc have one sign and sum to 1: x lies inside the triangle.
//   * cot B < 0 (obtuse at b): la and lb have one sign.  Both negative is impossible (it needs |lb| > (cot C / |cot B|) lc > lc > |lb|).
//     Both positive, lc = -s < 0: (*) gives la s < u |cot B| and lb > u cot C / la > s cot C / |cot B|, while lb < 1 + s; hence
//     s < d := |cot B| / (cot C - |cot B|).  The accepted x lie in the triangle {la >= 0, lb >= 0, lc >= -d}, whose corners are
//     c, a + d (a - c), b + d (b - c) -- a thin spike along the extension of cb beyond b, whatever the size of h.
//   * cot C < 0 (obtuse at c): the same with b and c exchanged: corners b, a + d (a - b), c + d (c - b), d = |cot C| / (cot B - |cot C|).
// Every fan triangle has a = v[0] on P0, so the height of P0 over the triangle's plane is linear in (lb, lc) and zero at a: the
// accepted p lie in the triangle spanned by the three corners lifted onto P0 along n.  When P0 is steep against the triangle's
// plane the lift is useless; then (*) itself bounds the height: the cotangent that is positive gives u < (1 + d)^2 / (4 cot), i.e.
// |h| < sqrt(2A u), and the box of the corners is grown by that much along n instead.
// The spike is u |cot| / s wide at s beyond the corner.  Where that is below 2^-55 no double p can tell it from the line, and the
// reference's own arithmetic answers rounding noise there (as it does for a ray through the edge of a planar polygon): the spike
// is boxed up to s = min(d, 2^55 u |cot|).  For a polygon that is planar up to the rounding of its coordinates (u ~ 1e-18 for
// nine-digit text) that is a per cent of an edge beyond an obtuse corner; for an exactly planar one, nothing.  The lifted corners
// are always included: p lies on P0, not on the triangle, and next to a coordinate that is exactly zero no outward rounding
// covers the difference.
#pragma once
#include <cmath>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define MI355RT_HD __host__ __device__
#else
#define MI355RT_HD
#endif

namespace mi355rt {

// The region's triangle for fan triangle (a, b, c): barycentric coordinates q[v] of its three corners, their heights h[v] over the
// triangle's plane when lifted onto P0 (along the triangle's unit normal n), or, when P0 is steep against that plane, the bound
// h_all on any accepted height instead.
struct AcceptRegion {
	double q[3][3], h[3], h_all, n[3];
	bool steep;
};

// false: nothing beyond the box of the vertices can be accepted (no area, no plane, not finite)
MI355RT_HD inline bool accept_region_solve(const double *a, const double *b, const double *c, const double *nf, AcceptRegion &r)
{
	const double e0[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e1[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]}, e2[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
	const double nx = e0[1] * e1[2] - e0[2] * e1[1], ny = e0[2] * e1[0] - e0[0] * e1[2], nz = e0[0] * e1[1] - e0[1] * e1[0];
	const double nn = nx * nx + ny * ny + nz * nz;
	if (!(nn > 0.0) || !(nn < 1e300)) return false; // no area (or not finite): alpha, beta, gamma vanish with it, nothing beyond rounding is accepted
	const double two_a = sqrt(nn);
	r.n[0] = nx / two_a; r.n[1] = ny / two_a; r.n[2] = nz / two_a;
	const double cos_t = r.n[0] * nf[0] + r.n[1] * nf[1] + r.n[2] * nf[2];
	if (!(cos_t == cos_t)) return false; // no plane (first three vertices collinear): every plane distance is NaN, nothing is accepted
	const double hb = e0[0] * nf[0] + e0[1] * nf[1] + e0[2] * nf[2];    // height of b over P0
	const double hc = -(e2[0] * nf[0] + e2[1] * nf[1] + e2[2] * nf[2]); // height of c over P0
	const double cot_b = -(e0[0] * e1[0] + e0[1] * e1[1] + e0[2] * e1[2]) / two_a;
	const double cot_c = -(e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2]) / two_a;
	// corners of the region's triangle as barycentric coordinates (la, lb, lc)
	double d = 0.0;
	int obtuse = 0; // 1: at b, 2: at c
	if (cot_b < 0.0) { obtuse = 1; d = -cot_b / (cot_c + cot_b); }
	else if (cot_c < 0.0) { obtuse = 2; d = -cot_c / (cot_b + cot_c); }
	if (obtuse && !(d >= 0.0 && d < 1e6)) d = 1e6; // a sliver: the sum of the cotangents has lost its digits; a box this large holds whatever is left of the spike
	d *= 1.0 + 1e-9;
	r.steep = fabs(cos_t) < 0.5;
	for (int pass = 0; pass < 2; pass++) {
		for (int v = 0; v < 3; v++)
			for (int k = 0; k < 3; k++) r.q[v][k] = v == k ? 1.0 : 0.0;
		if (obtuse == 1) { r.q[0][0] = 1.0 + d; r.q[0][2] = -d; r.q[1][1] = 1.0 + d; r.q[1][2] = -d; }
		else if (obtuse == 2) { r.q[0][0] = 1.0 + d; r.q[0][1] = -d; r.q[2][2] = 1.0 + d; r.q[2][1] = -d; }
		// heights of P0 over the triangle's plane at the corners (linear in lb, lc; zero at a), or the bound (*) leaves for them
		r.h_all = 0.0;
		if (!r.steep) {
			for (int v = 0; v < 3; v++) { r.h[v] = -(r.q[v][1] * hb + r.q[v][2] * hc) / cos_t; r.h_all = fmax(r.h_all, fabs(r.h[v])); }
		} else {
			const double cot_pos = fmax(cot_b, cot_c); // > 0: the two angles sum to less than 180 degrees
			r.h_all = sqrt(two_a * (1.0 + d) * (1.0 + d) / (4.0 * cot_pos)) * (1.0 + 1e-9);
			r.h[0] = r.h[1] = r.h[2] = 0.0;
		}
		if (pass == 0 && obtuse) {
			// At s beyond the corner the spike is u |cot| / s wide (in barycentric units): where that is below 2^-55 no double p can
			// tell it from the line, and what the reference's arithmetic answers there is rounding noise.  With u at its largest
			// over the whole spike: s < 2^55 u |cot|.  (A planar polygon has u of the order 2^-100: its spike ends at once.)
			const double u_max = r.h_all * r.h_all / two_a, reach = 3.6028797018963968e16 * u_max * (obtuse == 1 ? -cot_b : -cot_c);
			if (reach < d) {
				d = reach;
				continue; // once more, with the shorter spike
			}
		}
		break;
	}
	return true;
}

// lo/hi: box of every point of P0 = {p : (p - a).nf = 0} that the reference's test of fan triangle (a, b, c) can accept, joined with
// the box of the vertices themselves.  nf: the polygon's unit face normal (either orientation).
MI355RT_HD inline void accept_region_box(const double *a, const double *b, const double *c, const double *nf, double *lo, double *hi)
{
	for (int k = 0; k < 3; k++) {
		lo[k] = fmin(a[k], fmin(b[k], c[k]));
		hi[k] = fmax(a[k], fmax(b[k], c[k]));
	}
	AcceptRegion r;
	if (!accept_region_solve(a, b, c, nf, r)) return;
	for (int v = 0; v < 3; v++)
		for (int k = 0; k < 3; k++) {
			const double x = r.q[v][0] * a[k] + r.q[v][1] * b[k] + r.q[v][2] * c[k];
			if (!r.steep) {
				const double p = x + r.h[v] * r.n[k];
				lo[k] = fmin(lo[k], fmin(x, p)); // the corner itself as well: a margin of the size of the lift against its rounding
				hi[k] = fmax(hi[k], fmax(x, p));
			} else {
				lo[k] = fmin(lo[k], x - r.h_all * fabs(r.n[k]));
				hi[k] = fmax(hi[k], x + r.h_all * fabs(r.n[k]));
			}
		}
}

// The same region as triangles, for callers that cut it into pieces (a builder that splits a long triangle's reference): every
// acceptable point lies in the triangle of the lifted corners `lifted`; `flat` is the triangle of the corners in the fan triangle's
// own plane (the margin accept_region_box adds against the rounding of the lift).  The box of a, b, c, `flat` and `lifted` is what
// accept_region_box returns.  Returns 0 when nothing beyond the vertices is acceptable (flat = lifted = the vertices), 1 when the
// corners are valid, and -1 when P0 is steep against the triangle's plane: then only the box bounds the region.
MI355RT_HD inline int accept_region_corners(const double *a, const double *b, const double *c, const double *nf, double flat[3][3], double lifted[3][3])
{
	for (int k = 0; k < 3; k++) { flat[0][k] = lifted[0][k] = a[k]; flat[1][k] = lifted[1][k] = b[k]; flat[2][k] = lifted[2][k] = c[k]; }
	AcceptRegion r;
	if (!accept_region_solve(a, b, c, nf, r)) return 0;
	if (r.steep) return -1;
	for (int v = 0; v < 3; v++)
		for (int k = 0; k < 3; k++) {
			flat[v][k] = r.q[v][0] * a[k] + r.q[v][1] * b[k] + r.q[v][2] * c[k];
			lifted[v][k] = flat[v][k] + r.h[v] * r.n[k];
		}
	return 1;
}

} // namespace mi355rt
