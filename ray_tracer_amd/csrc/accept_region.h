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

// lo/hi: box of every point of P0 = {p : (p - a).nf = 0} that the reference's test of fan triangle (a, b, c) can accept, joined with
// the box of the vertices themselves.  nf: the polygon's unit face normal (either orientation).
MI355RT_HD inline void accept_region_box(const double *a, const double *b, const double *c, const double *nf, double *lo, double *hi)
{
	for (int k = 0; k < 3; k++) {
		lo[k] = fmin(a[k], fmin(b[k], c[k]));
		hi[k] = fmax(a[k], fmax(b[k], c[k]));
	}
	const double e0[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e1[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]}, e2[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
	const double nx = e0[1] * e1[2] - e0[2] * e1[1], ny = e0[2] * e1[0] - e0[0] * e1[2], nz = e0[0] * e1[1] - e0[1] * e1[0];
	const double nn = nx * nx + ny * ny + nz * nz;
	if (!(nn > 0.0) || !(nn < 1e300)) return; // no area (or not finite): alpha, beta, gamma vanish with it, nothing beyond rounding is accepted
	const double two_a = sqrt(nn);
	const double n[3] = {nx / two_a, ny / two_a, nz / two_a};
	const double cos_t = n[0] * nf[0] + n[1] * nf[1] + n[2] * nf[2];
	if (!(cos_t == cos_t)) return; // no plane (first three vertices collinear): every plane distance is NaN, nothing is accepted
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
	const bool steep = fabs(cos_t) < 0.5;
	for (int pass = 0; pass < 2; pass++) {
		double q[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
		if (obtuse == 1) { q[0][0] = 1.0 + d; q[0][2] = -d; q[1][1] = 1.0 + d; q[1][2] = -d; }
		else if (obtuse == 2) { q[0][0] = 1.0 + d; q[0][1] = -d; q[2][2] = 1.0 + d; q[2][1] = -d; }
		// heights of P0 over the triangle's plane at the corners (linear in lb, lc; zero at a), or the bound (*) leaves for them
		double h[3], h_all = 0.0;
		if (!steep) {
			for (int v = 0; v < 3; v++) { h[v] = -(q[v][1] * hb + q[v][2] * hc) / cos_t; h_all = fmax(h_all, fabs(h[v])); }
		} else {
			const double cot_pos = fmax(cot_b, cot_c); // > 0: the two angles sum to less than 180 degrees
			h_all = sqrt(two_a * (1.0 + d) * (1.0 + d) / (4.0 * cot_pos)) * (1.0 + 1e-9);
			h[0] = h[1] = h[2] = 0.0;
		}
		if (pass == 0 && obtuse) {
			// At s beyond the corner the spike is u |cot| / s wide (in barycentric units): where that is below 2^-55 no double p can
			// tell it from the line, and what the reference's arithmetic answers there is rounding noise.  With u at its largest
			// over the whole spike: s < 2^55 u |cot|.  (A planar polygon has u of the order 2^-100: its spike ends at once.)
			const double u_max = h_all * h_all / two_a, reach = 3.6028797018963968e16 * u_max * (obtuse == 1 ? -cot_b : -cot_c);
			if (reach < d) {
				d = reach;
				continue; // once more, with the shorter spike
			}
		}
		for (int v = 0; v < 3; v++)
			for (int k = 0; k < 3; k++) {
				const double x = q[v][0] * a[k] + q[v][1] * b[k] + q[v][2] * c[k];
				if (!steep) {
					const double p = x + h[v] * n[k];
					lo[k] = fmin(lo[k], fmin(x, p)); // the corner itself as well: a margin of the size of the lift against its rounding
					hi[k] = fmax(hi[k], fmax(x, p));
				} else {
					lo[k] = fmin(lo[k], x - h_all * fabs(n[k]));
					hi[k] = fmax(hi[k], x + h_all * fabs(n[k]));
				}
			}
		break;
	}
}

} // namespace mi355rt
