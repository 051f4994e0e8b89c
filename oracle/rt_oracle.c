/*
 * rt_oracle.c -- CPU restatement (fp64, plain C) of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY: see rt_oracle.h for who may use it and for the parity pin status
 * of every part ("PARITY UNPINNED" parts are listed there).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * All arithmetic keeps the reference's operation order; dot() and cross() follow vec3.h:120-133.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double e[3]; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {{x, y, z}}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
static inline v3 vsub(v3 a, v3 b) { return V(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
static inline v3 vscale(double t, v3 a) { return V(t * a.e[0], t * a.e[1], t * a.e[2]); }   /* vec3.h:103-112 */
static inline v3 vdiv(v3 a, double t) { return V(a.e[0] / t, a.e[1] / t, a.e[2] / t); }     /* vec3.h:113-117 */
static inline v3 vneg(v3 a) { return V(-a.e[0], -a.e[1], -a.e[2]); }
static inline double vdot(v3 a, v3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; } /* :120-124 */
static inline v3 vcross(v3 a, v3 b)                                                            /* :126-133 */
{
	return V(a.e[1] * b.e[2] - a.e[2] * b.e[1], -(a.e[0] * b.e[2] - a.e[2] * b.e[0]), a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
static inline double vlen(v3 a) { return sqrt(a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]); } /* :44-47 */
static inline double vlen2(v3 a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }      /* :49-52 */
static inline v3 vunit(v3 a) { return vdiv(a, vlen(a)); }                                             /* :186-190 */
static inline v3 vload(const double *p) { return V(p[0], p[1], p[2]); }
static inline void vstore(double *p, v3 a) { p[0] = a.e[0]; p[1] = a.e[1]; p[2] = a.e[2]; }

/* rodrigues(v, axis, theta), vec3.h:248-256 */
static inline v3 vrodrigues(v3 v, v3 ax, double theta)
{
	double c = cos(theta), sn = sin(theta);
	double nx = ax.e[0], ny = ax.e[1], nz = ax.e[2];
	v3 r0 = V(c + nx * nx * (1.0 - c), nx * ny * (1.0 - c) - nz * sn, nx * nz * (1.0 - c) + ny * sn);
	v3 r1 = V(ny * nx * (1.0 - c) + nz * sn, c + ny * ny * (1.0 - c), ny * nz * (1.0 - c) - nx * sn);
	v3 r2 = V(nz * nx * (1.0 - c) - ny * sn, nz * ny * (1.0 - c) + nx * sn, c + nz * nz * (1.0 - c));
	return V(vdot(r0, v), vdot(r1, v), vdot(r2, v));
}

/* The vector helpers every function of this file is built from, exported so that they can be checked against vec3.h compiled
 * unmodified: dot, length(a), squared_length(a), then cross, unit(a), a*b, a/b, t*a, a/t, a+b, a-b, -a, rodrigues(a, unit(b), t). */
void orc_vec3_ops(const double *a_, const double *b_, double t, double *out)
{
	const v3 a = vload(a_), b = vload(b_);
	out[0] = vdot(a, b);
	out[1] = vlen(a);
	out[2] = vlen2(a);
	vstore(out + 3, vcross(a, b));
	vstore(out + 6, vunit(a));
	vstore(out + 9, V(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2])); /* vec3.h:92-96 */
	vstore(out + 12, V(a.e[0] / b.e[0], a.e[1] / b.e[1], a.e[2] / b.e[2])); /* vec3.h:97-101 */
	vstore(out + 15, vscale(t, a));
	vstore(out + 18, vdiv(a, t));
	vstore(out + 21, vadd(a, b));
	vstore(out + 24, vsub(a, b));
	vstore(out + 27, vneg(a));
	vstore(out + 30, vrodrigues(a, vunit(b), t));
}
/* ray::point_at_parameter, ray.h:25-27 */
void orc_ray_point(const double *o, const double *d, double t, double *out) { vstore(out, vadd(vload(o), vscale(t, vload(d)))); }

int orc_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* BiliteralFilter::FilterImage (filter.cpp:13-51), statement for statement.  img, result: width*height*4 doubles. */
void orc_bilateral(const double *img, int img_width, int img_height, int window, double sigma_d, double sigma_r, double *result)
{
	for (int x = 0; x < img_width; x++) {
		for (int y = 0; y < img_height; y++) {
			double sum[3] = {0.0, 0.0, 0.0};
			double W[3] = {0.0, 0.0, 0.0};
			for (int i = -window; i <= window; i++) {
				for (int j = -window; j <= window; j++) {
					for (int c = 0; c < 3; c++) {
						int wx = x + i;
						int wy = y + j;
						if (wx < 0) wx = 0;
						if (wx >= img_width) wx = img_width - 1;
						if (wy < 0) wy = 0;
						if (wy >= img_height) wy = img_height - 1;
						double orig_value = img[((img_height - y - 1) * img_width + x) * 4 + c];
						double orig_wvalue = img[((img_height - wy - 1) * img_width + wx) * 4 + c];
						double power_index = -(double)((wx - x) * (wx - x) + (wy - y) * (wy - y)) / (2.0 * sigma_d * sigma_d);
						power_index += -(pow(orig_value - orig_wvalue, 2.0) / (2.0 * sigma_r * sigma_r));
						double w = exp(power_index);
						sum[c] += orig_wvalue * w;
						W[c] += w;
					}
				}
			}
			for (int c = 0; c < 3; c++) result[((img_height - y - 1) * img_width + x) * 4 + c] = sum[c] / W[c];
			result[((img_height - y - 1) * img_width + x) * 4 + 3] = 1.0;
		}
	}
}

/* ======================================================================================= */
/* AABB (aabb.cpp)                                                                          */
/* ======================================================================================= */
static inline double ffmin(double a, double b) { return a < b ? a : b; } /* aabb.cpp:7-10 */
static inline double ffmax(double a, double b) { return a > b ? a : b; } /* aabb.cpp:12-15 */

typedef struct { v3 minp, maxp, center; } aabb;

static inline aabb aabb_make(v3 a, v3 b) /* aabb.h:13-18 */
{
	aabb r;
	r.minp = a;
	r.maxp = b;
	r.center = vdiv(vadd(a, b), 2.0);
	return r;
}

static inline int aabb_hit(const aabb *b, v3 o, v3 d, double t_min, double t_max) /* aabb.cpp:17-34 */
{
	for (int i = 0; i < 3; i++) {
		double t0 = ffmin((b->minp.e[i] - o.e[i]) / d.e[i], (b->maxp.e[i] - o.e[i]) / d.e[i]);
		double t1 = ffmax((b->minp.e[i] - o.e[i]) / d.e[i], (b->maxp.e[i] - o.e[i]) / d.e[i]);
		t_min = ffmax(t0, t_min);
		t_max = ffmin(t1, t_max);
		if (t_max <= t_min)
			return 0;
	}
	return 1;
}

static inline double aabb_area(const aabb *b) /* aabb.cpp:37-41 */
{
	v3 v = vsub(b->maxp, b->minp);
	return 2.0 * (v.e[0] * v.e[1] + v.e[1] * v.e[2] + v.e[2] * v.e[0]);
}

static inline aabb aabb_union(const aabb *a, const aabb *b) /* aabb.cpp:45-54 */
{
	v3 mn = V(fmin(a->minp.e[0], b->minp.e[0]), fmin(a->minp.e[1], b->minp.e[1]), fmin(a->minp.e[2], b->minp.e[2]));
	v3 mx = V(fmax(a->maxp.e[0], b->maxp.e[0]), fmax(a->maxp.e[1], b->maxp.e[1]), fmax(a->maxp.e[2], b->maxp.e[2]));
	return aabb_make(mn, mx);
}

int orc_aabb_hit(const double minp[3], const double maxp[3], const double o[3], const double d[3], double t_min, double t_max)
{
	aabb b = aabb_make(vload(minp), vload(maxp));
	return aabb_hit(&b, vload(o), vload(d), t_min, t_max);
}
double orc_aabb_surface_area(const double minp[3], const double maxp[3])
{
	aabb b = aabb_make(vload(minp), vload(maxp));
	return aabb_area(&b);
}
void orc_surrounding_box(const double min0[3], const double max0[3], const double min1[3], const double max1[3],
                         double omin[3], double omax[3])
{
	aabb a = aabb_make(vload(min0), vload(max0)), b = aabb_make(vload(min1), vload(max1));
	aabb u = aabb_union(&a, &b);
	vstore(omin, u.minp);
	vstore(omax, u.maxp);
}

/* ======================================================================================= */
/* ONB (onb.cpp)                                                                            */
/* ======================================================================================= */
typedef struct { v3 axis[3]; } onb;

static inline onb onb_from_w(v3 n) /* onb.cpp:31-42 */
{
	onb r;
	r.axis[2] = vunit(n);
	v3 a;
	if (fabs(r.axis[2].e[0]) > 0.9)
		a = V(0, 1, 0);
	else
		a = V(1, 0, 0);
	r.axis[1] = vunit(vcross(r.axis[2], a));
	r.axis[0] = vcross(r.axis[2], r.axis[1]);
	return r;
}
static inline v3 onb_l2w(const onb *b, v3 a) /* onb.cpp:21-24 */
{
	return vadd(vadd(vscale(a.e[0], b->axis[0]), vscale(a.e[1], b->axis[1])), vscale(a.e[2], b->axis[2]));
}
static inline v3 onb_w2l(const onb *b, v3 a) /* onb.cpp:26-29 */
{
	return V(vdot(b->axis[0], a), vdot(b->axis[1], a), vdot(b->axis[2], a));
}
void orc_onb_from_w(const double n[3], double axis[9])
{
	onb b = onb_from_w(vload(n));
	for (int i = 0; i < 3; i++) vstore(axis + 3 * i, b.axis[i]);
}
void orc_onb_local_to_world(const double axis[9], const double a[3], double out[3])
{
	onb b;
	for (int i = 0; i < 3; i++) b.axis[i] = vload(axis + 3 * i);
	vstore(out, onb_l2w(&b, vload(a)));
}
void orc_onb_world_to_local(const double axis[9], const double a[3], double out[3])
{
	onb b;
	for (int i = 0; i < 3; i++) b.axis[i] = vload(axis + 3 * i);
	vstore(out, onb_w2l(&b, vload(a)));
}

/* ======================================================================================= */
/* Spectrum (spectrum.cpp)                                                                  */
/* ======================================================================================= */
#define SAMPLE_SIZE 10 /* spectrum.h:9  (int) RANGE/N_SAMPLE */
#define RANGE_NM 300   /* spectrum.h:7 */

/* CIE 1931 colour matching functions at 405..695 nm (published CVRL data; spectrum.cpp:5-102) */
static const double cie_x[ORC_NBINS] = {
	4.742986E-02, 1.446214E-01, 2.488523E-01, 3.227087E-01, 3.418483E-01, 2.826646E-01, 2.219781E-01, 1.29192E-01,
	4.600865E-02, 7.097731E-03, 3.649178E-03, 4.315171E-02, 1.268468E-01, 2.405015E-01, 3.804244E-01, 5.280233E-01,
	7.016774E-01, 8.853376E-01, 1.051011E+00, 1.14362E+00,  1.134757E+00, 1.007344E+00, 8.135565E-01, 5.75541E-01,
	3.844986E-01, 2.277792E-01, 1.263808E-01, 6.63996E-02,  3.292138E-02, 1.575417E-02};
static const double cie_y[ORC_NBINS] = {
	4.971717E-03, 1.429377E-02, 2.612106E-02, 4.15794E-02,  5.743393E-02, 7.238339E-02, 1.060145E-01, 1.535066E-01,
	2.064828E-01, 2.85068E-01,  4.277595E-01, 6.206256E-01, 7.946448E-01, 9.071347E-01, 9.814106E-01, 9.994608E-01,
	9.902549E-01, 9.424569E-01, 8.587203E-01, 7.544785E-01, 6.270066E-01, 4.89595E-01,  3.609245E-01, 2.416902E-01,
	1.547397E-01, 8.979594E-02, 4.899699E-02, 2.554223E-02, 1.261573E-02, 6.027677E-03};
static const double cie_z[ORC_NBINS] = {
	2.369246E-01, 7.378822E-01, 1.305008E+00, 1.74828E+00,  1.918437E+00, 1.664439E+00, 1.42844E+00,  9.991789E-01,
	5.617313E-01, 3.105939E-01, 1.720018E-01, 8.283548E-02, 3.751912E-02, 1.566174E-02, 6.131421E-03, 2.327186E-03,
	8.822531E-04, 3.386739E-04, 1.335031E-04, 5.460706E-05, 2.334738E-05, 1.048387E-05, 0, 0, 0, 0, 0, 0, 0, 0};

double orc_spectrum_integrate(const double data[ORC_NBINS], double min_wl, double max_wl) /* spectrum.cpp:126-146 */
{
	double sum = 0.0;
	size_t min_i = (size_t)ceil((min_wl - 400) / SAMPLE_SIZE);
	double mod_min = fmod(min_wl - 400, (double)(RANGE_NM / ORC_NBINS)); /* int/size_t quotient = 10 */
	size_t max_i = (size_t)floor((max_wl - 400) / SAMPLE_SIZE);
	double mod_max = fmod(max_wl - 400, (double)(RANGE_NM / ORC_NBINS));
	if (min_i >= 1)
		sum += data[min_i - 1] * mod_min;
	if (max_i <= (size_t)(ORC_NBINS - 2))
		sum += data[max_i + 1] * mod_max;
	for (size_t i = min_i; i <= max_i; i++)
		sum += data[i] * SAMPLE_SIZE;
	return sum;
}

void orc_spectrum_add(double data[ORC_NBINS], double value, double min_wl, double max_wl) /* spectrum.cpp:148-155 */
{
	size_t min_i = (size_t)ceil((min_wl - 400) / SAMPLE_SIZE);
	size_t max_i = (size_t)floor((max_wl - 400) / SAMPLE_SIZE);
	for (size_t i = min_i; i <= max_i; i++)
		data[i] += value;
}

double orc_spectrum_get(const double data[ORC_NBINS], double wavelength) /* spectrum.cpp:158-162 */
{
	size_t i = (size_t)floor((wavelength - 400) / SAMPLE_SIZE);
	return data[i];
}

void orc_xyz(const double data[ORC_NBINS], double out[3]) /* spectrum.cpp:164-171 */
{
	double sx[ORC_NBINS], sy[ORC_NBINS], sz[ORC_NBINS];
	for (int i = 0; i < ORC_NBINS; i++) {
		sx[i] = data[i] * cie_x[i];
		sy[i] = data[i] * cie_y[i];
		sz[i] = data[i] * cie_z[i];
	}
	out[0] = orc_spectrum_integrate(sx, 400, 699.999);
	out[1] = orc_spectrum_integrate(sy, 400, 699.999);
	out[2] = orc_spectrum_integrate(sz, 400, 699.999);
}

void orc_rgb(const double data[ORC_NBINS], double out[3]) /* spectrum.cpp:184-199 */
{
	double c[3];
	orc_xyz(data, c);
	out[0] = 3.240479 * c[0] + (-1.537150) * c[1] + (-0.498535) * c[2];
	out[1] = (-0.969256) * c[0] + 1.875991 * c[1] + 0.041556 * c[2];
	out[2] = 0.055648 * c[0] + (-0.204043) * c[1] + 1.057311 * c[2];
}

void orc_rgb_to_spectrum(const double rgb[3], double out[ORC_NBINS]) /* spectrum.cpp:219-232 */
{
	for (int i = 0; i < ORC_NBINS; i++) out[i] = 0.0;
	for (int i = 0; i < ORC_NBINS / 3.0; i++) out[i] = rgb[2];
	for (int i = ORC_NBINS / 3.0; i < ORC_NBINS / 3.0 * 2.0; i++) out[i] = rgb[1];
	for (int i = ORC_NBINS / 3.0 * 2.0; i < ORC_NBINS; i++) out[i] = rgb[0];
}

void orc_tonemap(const double data[ORC_NBINS], double scale, double out[3]) /* renderer.cpp:131-139 */
{
	double s[ORC_NBINS];
	for (int i = 0; i < ORC_NBINS; i++) s[i] = data[i] * scale;
	orc_rgb(s, out);
	for (int c = 0; c < 3; c++) {
		if (out[c] >= 0.0) {
			out[c] = pow(out[c], 1.0 / 2.2);
			out[c] = out[c] < 1.0 ? out[c] : 1.0;
		} else
			out[c] = 0.0;
	}
}

/* ======================================================================================= */
/* Fresnel, heuristics, sampling (material.cpp, pdf.cpp)                                    */
/* ======================================================================================= */
double orc_rfresnel(double cos_theta, double n) /* material.cpp:14-25 */
{
	const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
	const double sin_t = sin_theta / n;
	if (sin_t > 1.0)
		return 1.0;
	const double q = sqrt(n * n - sin_theta * sin_theta);
	const double rte = (cos_theta - q) / (cos_theta + q);
	const double rtm = (n * n * cos_theta - q) / (n * n * cos_theta + q);
	return (rte * rte + rtm * rtm) / 2.0;
}

/* complex helpers following libstdc++'s std::complex<double> operator semantics closely enough
 * for a unit-tolerance check (principal square root, textbook division) */
typedef struct { double re, im; } cplx;
static cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static cplx cdivc(cplx a, cplx b)
{
	double den = b.re * b.re + b.im * b.im;
	cplx r = {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
	return r;
}
static cplx csqrtc(cplx z)
{
	double m = hypot(z.re, z.im);
	cplx r;
	if (m == 0.0) { r.re = 0; r.im = 0; return r; }
	if (z.re >= 0.0) {
		r.re = sqrt(0.5 * (m + z.re));
		r.im = z.im / (2.0 * r.re);
	} else {
		double t = sqrt(0.5 * (m - z.re));
		r.re = fabs(z.im) / (2.0 * t);
		r.im = z.im < 0 ? -t : t;
	}
	return r;
}

double orc_cfresnel(double cos_theta, double n_re, double n_im) /* material.cpp:26-34 */
{
	const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
	const double sin_t = sin_theta / n_re;
	if (sin_t > 1.0)
		return 1.0;
	cplx n = {n_re, n_im};
	cplx nn = cmul(n, n);
	cplx q = {nn.re - sin_theta * sin_theta, nn.im};
	cplx s = csqrtc(q);
	cplx num = {cos_theta - s.re, -s.im};
	cplx den = {cos_theta + s.re, s.im};
	cplx f = cdivc(num, den);
	const double r = hypot(f.re, f.im);
	return r * r;
}

double orc_power_heuristic(double pdf1, double pdf2, double beta) /* pdf.cpp:139-146 */
{
	if (isinf(pdf1))
		return 1.0;
	return pow(pdf1, beta) / (pow(pdf1, beta) + pow(pdf2, beta));
}

/* MixturePdf (pdf.cpp:116-135; constructed nowhere in the reference, restated because the hot-path description names it).
 * Generate(): one uniform number r picks component i = the first with r < (i+1)/l, the last one otherwise.
 * PdfVal(): the plain average of the components' values.  Parity unpinned: pdf.cpp does not compile here (pdf.h -> object.h ->
 * obj.h needs Boost) and the reference holds no fixture for it. */
uint64_t orc_mixture_pick(double r, uint64_t l)
{
	for (uint64_t i = 0; i < l; i++)
		if (i == (l - 1) || r < ((double)(i + 1) / (double)l))
			return i;
	return l - 1;
}
double orc_mixture_pdf(const double *component_pdf, uint64_t l)
{
	double sum = 0.0;
	for (uint64_t i = 0; i < l; i++) sum += component_pdf[i];
	return sum / (double)l;
}
/* CosinePdf::PdfVal (pdf.cpp:76-84): dot(direction, w) / pi (the direction is not normalised there), 0 below the horizon */
double orc_cosine_pdf(const double *direction, const double *w)
{
	double cos_theta = vdot(vload(direction), vload(w));
	return cos_theta >= 0.0 ? cos_theta / M_PI : 0.0;
}

void orc_cosine_hemisphere(double r1, double r2, double out[3]) /* pdf.cpp:36-46 (z is 1-r2 as there) */
{
	double phi = 2 * M_PI * r1;
	double sin_theta = sqrt(r2);
	out[0] = sin_theta * cos(phi);
	out[1] = sin_theta * sin(phi);
	out[2] = 1 - r2;
}

/* ======================================================================================= */
/* Random stream shared with the device integrator                                          */
/* ======================================================================================= */
static inline uint64_t mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
uint64_t orc_rng_seed(uint64_t seed, uint64_t pixel, uint64_t sample)
{
	return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL * (sample + 1)) ^ (0xD1B54A32D192ED03ULL * (pixel + 1)));
}
/* One draw: PCG-XSH-RR 64/32 (O'Neill 2014) -- a 64-bit linear congruential step, output = the old state's high bits xor-shifted
 * and rotated by its top five bits.  Every (pixel, sample) starts at its own point of the sequence (orc_rng_seed, two splitmix64
 * rounds).  One 64-bit multiplication per draw, where a splitmix64 stream needs two: the device's generate kernel is bound by its
 * vector instruction rate, and 64-bit multiplications are its most expensive instructions. */
uint64_t orc_rng_next(uint64_t *state)
{
	const uint64_t old = *state;
	*state = old * 6364136223846793005ULL + 1442695040888963407ULL;
	const uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
	return (uint64_t)((xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u)));
}
double orc_rng_f64(uint64_t *state) { return (double)orc_rng_next(state) * (1.0 / 4294967296.0); }
static inline uint32_t rng_below(uint64_t *state, uint32_t n) { return (uint32_t)(orc_rng_next(state) % n); }

/* ======================================================================================= */
/* Camera (camera.h:101-209)                                                                */
/* ======================================================================================= */
void orc_camera_default(orc_camera *c) /* camera.h:189-193 */
{
	memset(c, 0, sizeof(*c));
	c->film_sensitivity = 1.0;
	c->aperture = 0.5;
	c->film_width = 0.024;
	c->film_height = 0.024;
	for (int i = 0; i < ORC_NBINS; i++) c->n[i] = 2.0;
}

void orc_camera_set_lens(orc_camera *c, double vfov, double focal_length) /* camera.h:108-125 */
{
	double theta = vfov * M_PI / 180;
	c->d = (0.5 * c->film_height) / tan(theta / 2.0);
	c->lens_maker_coefficient = 1.0 / (focal_length * (c->n[ORC_NBINS / 2] - 1));
}

void orc_camera_set_camera(orc_camera *c, const double from[3], const double at[3], const double up[3]) /* :128-145 */
{
	v3 origin = vload(from);
	v3 w = vunit(vsub(vload(from), vload(at)));
	v3 u = vunit(vcross(vload(up), w));
	v3 v = vcross(w, u);
	v3 llc = vadd(vsub(vsub(origin, vscale(0.5 * c->film_width, u)), vscale(0.5 * c->film_height, v)), vscale(c->d, w));
	vstore(c->origin, origin);
	vstore(c->w, w);
	vstore(c->u, u);
	vstore(c->v, v);
	vstore(c->film_lower_left_corner, llc);
	vstore(c->horizontal, vscale(c->film_width, u));
	vstore(c->vertical, vscale(c->film_height, v));
}

void orc_camera_get_ray(const orc_camera *c, double u, double v, double wavelength, const double lens_xy[2],
                        double o[3], double d[3], double *p_image, double *p_lens, double *cos_theta) /* :159-187 */
{
	double focal_length = 1.0 / ((orc_spectrum_get(c->n, wavelength) - 1) * c->lens_maker_coefficient); /* :197-200 */
	double a = 1.0 / (1.0 / focal_length - 1.0 / c->d);
	v3 origin = vload(c->origin), horizontal = vload(c->horizontal), vertical = vload(c->vertical);
	v3 image_sensor_pos = vadd(vadd(vload(c->film_lower_left_corner), vscale(u, horizontal)), vscale(v, vertical));
	v3 target = vadd(origin, vscale(-a / c->d, vsub(image_sensor_pos, origin)));
	v3 tmp = vscale(c->aperture / 2.0, V(lens_xy[0], lens_xy[1], 0.0)); /* :148-157 */
	v3 random_point_on_lens = vadd(vadd(origin, vscale(tmp.e[0], horizontal)), vscale(tmp.e[1], vertical));
	*cos_theta = vdot(vunit(vsub(random_point_on_lens, image_sensor_pos)), vneg(vload(c->w)));
	*p_image = 1.0 / (c->film_width * c->film_height);
	*p_lens = 1.0 / (M_PI * c->aperture * c->aperture / 4.0);
	v3 dir = vunit(vsub(target, random_point_on_lens));
	if (!(a > 0.0))
		dir = vneg(dir);
	vstore(o, random_point_on_lens);
	vstore(d, dir);
}

/* ======================================================================================= */
/* Scene: polygons (object.cpp:94-140, 286-395)                                             */
/* ======================================================================================= */
typedef struct {
	int32_t left, right; /* >= 0: node index; < 0: ~polygon index (per-object trees) or ~(object root) never */
	aabb box;
} bnode;

struct orc_scene {
	uint64_t n_polys, n_corners, n_tris;
	uint32_t n_objects;
	uint64_t *poly_begin;  /* n_polys+1 -> corner arrays */
	uint64_t *tri_begin;   /* n_polys+1 -> fan triangle ids */
	uint64_t *object_begin;/* n_objects+1 -> polygons */
	v3 *pv, *pvt, *pn;     /* per corner */
	v3 *face_normal;       /* per polygon */
	uint32_t *object_id;   /* per polygon */
	double *polygon_area;  /* per polygon */
	double *tri_cum;       /* per fan triangle: normalised cumulative area (object.cpp:375-395) */
	uint32_t *tri_poly;    /* per fan triangle */
	aabb *poly_box;        /* per polygon (object.cpp:309-328) */
	/* BVH */
	bnode *nodes;
	uint64_t n_nodes, cap_nodes;
	int32_t root;
	uint32_t depth;
	/* materials */
	orc_material *mats; /* per object */
	uint64_t *lights;   /* polygon ids */
	uint64_t n_lights;
	/* per-hit material graphs: program of slot a (albedo / color) per object, shared tables */
	uint32_t *prog_begin, *prog_len; /* per object; len 0: the constant record */
	uint32_t *code; uint32_t n_code; /* (op, arg) pairs */
	double *g_spectra, *g_vectors, *g_scalars;
	uint32_t n_g_spectra, n_g_vectors, n_g_scalars, n_g_textures;
	uint8_t **g_tex; int *g_tex_w, *g_tex_h, *g_tex_bpp;
	int watertight;     /* 0: the reference's triangle test; 1: the optional watertight test (watertight_t) */
	/* environment texture (globals of scene.h:20-22) */
	uint8_t *env;
	int env_w, env_h, env_bpp;
	double env_brightness;
};

static aabb polygon_bbox(const orc_scene *s, uint64_t p) /* object.cpp:309-328 */
{
	uint64_t b = s->poly_begin[p], e = s->poly_begin[p + 1];
	v3 mn = s->pv[b], mx = s->pv[b];
	for (uint64_t i = b + 1; i < e; i++)
		for (int j = 0; j < 3; j++) {
			mn.e[j] = fmin(mn.e[j], s->pv[i].e[j]);
			mx.e[j] = fmax(mx.e[j], s->pv[i].e[j]);
		}
	for (int i = 0; i < 3; i++)
		if (mn.e[i] == mx.e[i]) {
			mn.e[i] -= 0.0001;
			mx.e[i] += 0.0001;
		}
	return aabb_make(mn, mx);
}

orc_scene *orc_scene_create(const double *v, uint64_t n_v, const double *vt, uint64_t n_vt, const double *vn,
                            uint64_t n_vn, const int64_t *corners, const uint64_t *face_begin, uint64_t n_faces,
                            const uint64_t *object_begin, uint32_t n_objects)
{
	orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
	uint64_t nc = face_begin[n_faces];
	s->n_polys = n_faces;
	s->n_corners = nc;
	s->n_objects = n_objects;
	s->poly_begin = (uint64_t *)malloc((n_faces + 1) * sizeof(uint64_t));
	s->tri_begin = (uint64_t *)malloc((n_faces + 1) * sizeof(uint64_t));
	s->object_begin = (uint64_t *)malloc((n_objects + 1) * sizeof(uint64_t));
	memcpy(s->poly_begin, face_begin, (n_faces + 1) * sizeof(uint64_t));
	memcpy(s->object_begin, object_begin, (n_objects + 1) * sizeof(uint64_t));
	s->pv = (v3 *)malloc(nc * sizeof(v3));
	s->pvt = (v3 *)malloc(nc * sizeof(v3));
	s->pn = (v3 *)malloc(nc * sizeof(v3));
	s->face_normal = (v3 *)malloc(n_faces * sizeof(v3));
	s->object_id = (uint32_t *)malloc(n_faces * sizeof(uint32_t));
	s->polygon_area = (double *)malloc(n_faces * sizeof(double));
	s->poly_box = (aabb *)malloc(n_faces * sizeof(aabb));
	uint64_t nt = 0;
	for (uint64_t f = 0; f < n_faces; f++) {
		s->tri_begin[f] = nt;
		nt += face_begin[f + 1] - face_begin[f] - 2;
	}
	s->tri_begin[n_faces] = nt;
	s->n_tris = nt;
	s->tri_cum = (double *)malloc(nt * sizeof(double));
	s->tri_poly = (uint32_t *)malloc(nt * sizeof(uint32_t));
	for (uint32_t o = 0; o < n_objects; o++)
		for (uint64_t f = object_begin[o]; f < object_begin[o + 1]; f++)
			s->object_id[f] = o;
	for (uint64_t f = 0; f < n_faces; f++) {
		uint64_t b = face_begin[f], e = face_begin[f + 1];
		for (uint64_t k = b; k < e; k++) { /* object.cpp:111-116 */
			int64_t iv = corners[3 * k], ivt = corners[3 * k + 1], ivn = corners[3 * k + 2];
			if (iv < 0 || (uint64_t)iv >= n_v || ivn < 0 || (uint64_t)ivn >= n_vn || (ivt >= 0 && (uint64_t)ivt >= n_vt)) {
				fprintf(stderr, "orc_scene_create: corner %llu index out of range\n", (unsigned long long)k);
				orc_scene_free(s);
				return NULL;
			}
			s->pv[k] = vload(v + 3 * iv);
			s->pvt[k] = ivt >= 0 ? vload(vt + 3 * ivt) : V(0, 0, 0);
			s->pn[k] = vunit(vload(vn + 3 * ivn));
		}
		v3 fn = vunit(vcross(vsub(s->pv[b + 1], s->pv[b]), vsub(s->pv[b + 2], s->pv[b + 1]))); /* :117-120 */
		for (uint64_t k = b; k < e; k++) /* :121-126 */
			if (vdot(fn, s->pn[k]) < 0.0)
				fn = vscale(-1.0, fn);
		s->face_normal[f] = fn;
		/* CalcTriangleAreas, object.cpp:375-395 */
		double area = 0.0;
		uint64_t t0 = s->tri_begin[f];
		for (uint64_t vi = 1; vi + 1 < e - b; vi++) {
			v3 a = vsub(s->pv[b + vi], s->pv[b]);
			v3 bb = vsub(s->pv[b + vi + 1], s->pv[b]);
			double tri_area = vlen(vcross(a, bb)) / 2.0;
			uint64_t ti = vi - 1;
			s->tri_cum[t0 + ti] = ti == 0 ? tri_area : s->tri_cum[t0 + ti - 1] + tri_area;
			s->tri_poly[t0 + ti] = (uint32_t)f;
			area += tri_area;
		}
		uint64_t ntf = e - b - 2;
		for (uint64_t ti = 0; ti < ntf; ti++) s->tri_cum[t0 + ti] /= area;
		s->tri_cum[t0 + ntf - 1] = 1.0;
		s->polygon_area[f] = area;
		s->poly_box[f] = polygon_bbox(s, f);
	}
	s->root = -1;
	return s;
}

static void free_graphs(orc_scene *s);
void orc_scene_free(orc_scene *s)
{
	if (!s) return;
	free(s->poly_begin); free(s->tri_begin); free(s->object_begin);
	free(s->pv); free(s->pvt); free(s->pn); free(s->face_normal); free(s->object_id);
	free(s->polygon_area); free(s->tri_cum); free(s->tri_poly); free(s->poly_box);
	free(s->nodes); free(s->mats); free(s->lights); free(s->env);
	free_graphs(s);
	free(s);
}
uint64_t orc_scene_n_polygons(const orc_scene *s) { return s->n_polys; }
uint64_t orc_scene_n_triangles(const orc_scene *s) { return s->n_tris; }
void orc_scene_polygon_bbox(const orc_scene *s, uint64_t poly, double minp[3], double maxp[3])
{
	vstore(minp, s->poly_box[poly].minp);
	vstore(maxp, s->poly_box[poly].maxp);
}
void orc_scene_set_triangle_test(orc_scene *s, int watertight) { s->watertight = watertight ? 1 : 0; }

void orc_scene_get_triangles(const orc_scene *s, double *v9, double *nf3, uint32_t *poly)
{
	for (uint64_t f = 0; f < s->n_polys; f++) {
		uint64_t b = s->poly_begin[f];
		for (uint64_t t = s->tri_begin[f]; t < s->tri_begin[f + 1]; t++) {
			uint64_t i = t - s->tri_begin[f] + 1;
			vstore(v9 + 9 * t, s->pv[b]);
			vstore(v9 + 9 * t + 3, s->pv[b + i]);
			vstore(v9 + 9 * t + 6, s->pv[b + i + 1]);
			vstore(nf3 + 3 * t, s->face_normal[f]);
			poly[t] = (uint32_t)f;
		}
	}
}

/* --------------------------------------------------------------------------------------- */
/* HitRecord subset and ConvexPolygon::Hit / HitTriangle (object.cpp:238-305)               */
/* --------------------------------------------------------------------------------------- */
typedef struct {
	double t;
	v3 p, normal;
	v3 vt;        /* rec.vt, object.cpp:258 */
	int64_t poly; /* rec.hit_object */
	int64_t tri;  /* which fan triangle accepted the hit */
} hitrec;

/* The optional watertight triangle test (no counterpart in the reference, whose own test -- hit_triangle below -- leaks rays
 * through shared edges and vertices: strict `> 0` on products of cross products).  Woop, Benthin, Wald, "Watertight Ray/Triangle
 * Intersection" (JCGT 2013), in double precision without FMA contraction: translate the vertices to the ray origin, permute the
 * axes so that the direction's largest component is z (swapping x and y when it is negative keeps the winding), shear the
 * direction onto +z, and evaluate the three 2D edge functions.  An edge shared by two triangles yields the same products with
 * the operands exchanged, i.e. exactly opposite values, so a ray cannot miss both; edges and vertices count as hits; no
 * back-face culling (the reference has none).  t is the sheared z of the hit divided by the sum of the edge functions.
 * The library's RT_TRIANGLE_WATERTIGHT mode computes exactly this, operation for operation. */
static inline int watertight_t(v3 o, v3 d, double t_min, double t_max, v3 v0, v3 v1, v3 v2, double *t_out)
{
	const double ax = fabs(d.e[0]), ay = fabs(d.e[1]), az = fabs(d.e[2]);
	int kz = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
	int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
	if (d.e[kz] < 0.0) { int tmp = kx; kx = ky; ky = tmp; }
	const double sx = d.e[kx] / d.e[kz], sy = d.e[ky] / d.e[kz], sz = 1.0 / d.e[kz];
	const v3 A = vsub(v0, o), B = vsub(v1, o), C = vsub(v2, o);
	const double Ax = A.e[kx] - sx * A.e[kz], Ay = A.e[ky] - sy * A.e[kz];
	const double Bx = B.e[kx] - sx * B.e[kz], By = B.e[ky] - sy * B.e[kz];
	const double Cx = C.e[kx] - sx * C.e[kz], Cy = C.e[ky] - sy * C.e[kz];
	const double U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
	if ((U < 0.0 || V < 0.0 || W < 0.0) && (U > 0.0 || V > 0.0 || W > 0.0))
		return 0;
	const double det = U + V + W;
	if (det == 0.0)
		return 0;
	const double Az = sz * A.e[kz], Bz = sz * B.e[kz], Cz = sz * C.e[kz];
	const double t = (U * Az + V * Bz + W * Cz) / det;
	if (!(t >= t_min && t <= t_max))
		return 0;
	*t_out = t;
	return 1;
}

static inline int hit_triangle(v3 o, v3 d, double t_min, double t_max, v3 v0, v3 v1, v3 v2, v3 n0, v3 n1, v3 n2,
                               v3 vt0, v3 vt1, v3 vt2, v3 face_normal, hitrec *rec, int watertight) /* object.cpp:238-284 */
{
	double t;
	if (watertight) {
		if (!watertight_t(o, d, t_min, t_max, v0, v1, v2, &t))
			return 0;
	} else {
		t = vdot(vsub(v0, o), face_normal) / vdot(d, face_normal);
		if (!(t >= t_min && t <= t_max))
			return 0;
	}
	v3 p = vadd(o, vscale(t, d)); /* ray.h:26-28 */
	double tri_area = vlen(vcross(vsub(v1, v0), vsub(v2, v1)));
	v3 result0 = vcross(vsub(v1, v0), vsub(p, v1));
	v3 result1 = vcross(vsub(v2, v1), vsub(p, v2));
	v3 result2 = vcross(vsub(v0, v2), vsub(p, v0));
	if (watertight || (vdot(result0, result1) > 0.0 && vdot(result1, result2) > 0.0)) {
		rec->t = t;
		rec->p = p;
		rec->normal = vdiv(vadd(vadd(vscale(vlen(result1), n0), vscale(vlen(result2), n1)), vscale(vlen(result0), n2)), tri_area);
		rec->vt = vdiv(vadd(vadd(vscale(vlen(result1), vt0), vscale(vlen(result2), vt1)), vscale(vlen(result0), vt2)), tri_area); /* :258 */
		return 1;
	}
	return 0;
}

static inline int polygon_hit(const orc_scene *s, uint64_t f, v3 o, v3 d, double t_min, double t_max, hitrec *rec) /* :286-305 */
{
	uint64_t b = s->poly_begin[f], n = s->poly_begin[f + 1] - b;
	int found = 0;
	for (uint64_t i = 1; i + 1 < n; i++) {
		hitrec cand;
		if (hit_triangle(o, d, t_min, t_max, s->pv[b], s->pv[b + i], s->pv[b + i + 1], s->pn[b], s->pn[b + i],
		                 s->pn[b + i + 1], s->pvt[b], s->pvt[b + i], s->pvt[b + i + 1], s->face_normal[f], &cand, s->watertight)) {
			cand.poly = (int64_t)f;
			cand.tri = (int64_t)(s->tri_begin[f] + i - 1);
			if (!s->watertight) { /* the reference: the first fan triangle that accepts wins (all share one plane distance) */
				*rec = cand;
				return 1;
			}
			/* watertight mode: every fan triangle has its own plane, a warped polygon can be met twice: the nearest counts */
			if (!found || cand.t < rec->t) *rec = cand;
			found = 1;
		}
	}
	return found;
}

int orc_polygon_hit(const orc_scene *s, uint64_t poly, const double o[3], const double d[3], double t_min, double t_max,
                    double *t, double p[3], double normal[3], int64_t *tri)
{
	hitrec rec;
	if (!polygon_hit(s, poly, vload(o), vload(d), t_min, t_max, &rec))
		return 0;
	*t = rec.t;
	vstore(p, rec.p);
	vstore(normal, rec.normal);
	*tri = rec.tri;
	return 1;
}

/* --------------------------------------------------------------------------------------- */
/* SAHBVHNode (bvh.cpp:134-232) over cached polygon boxes                                   */
/* --------------------------------------------------------------------------------------- */
typedef struct { int32_t id; aabb box; } bitem; /* id: ~polygon for polygons, node index for object roots */

static int g_axis; /* comparator axis; builds are single-threaded */
static int cmp_center(const void *a, const void *b)
{
	double ca = ((const bitem *)a)->box.center.e[g_axis], cb = ((const bitem *)b)->box.center.e[g_axis];
	return (ca < cb) ? -1 : (cb < ca) ? 1 : 0;
}

static int32_t new_node(orc_scene *s)
{
	if (s->n_nodes == s->cap_nodes) {
		s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 1024;
		s->nodes = (bnode *)realloc(s->nodes, s->cap_nodes * sizeof(bnode));
	}
	return (int32_t)s->n_nodes++;
}

static aabb item_box(const orc_scene *s, int32_t id) { return id < 0 ? s->poly_box[~id] : s->nodes[id].box; }

static int32_t build_sah(orc_scene *s, bitem *list, size_t n, uint32_t depth)
{
	if (depth > s->depth) s->depth = depth;
	int32_t me = new_node(s);
	if (n == 1) { /* bvh.cpp:137-141 */
		s->nodes[me].left = list[0].id;
		s->nodes[me].right = list[0].id;
		s->nodes[me].box = list[0].box;
		return me;
	} else if (n == 2) { /* :142-150 */
		s->nodes[me].left = list[0].id;
		s->nodes[me].right = list[1].id;
		s->nodes[me].box = aabb_union(&list[0].box, &list[1].box);
		return me;
	}
	double best_cost = DBL_MAX;
	int best_axis = -1;
	long best_split_index = -1;
	double *sa0 = (double *)malloc((n - 1) * sizeof(double)), *sa1 = (double *)malloc((n - 1) * sizeof(double));
	for (int axis = 0; axis < 3; axis++) { /* :157-211 */
		g_axis = axis;
		qsort(list, n, sizeof(bitem), cmp_center);
		aabb box = list[0].box;
		sa0[0] = aabb_area(&box);
		for (size_t i = 1; i < n - 1; i++) {
			box = aabb_union(&box, &list[i].box);
			sa0[i] = aabb_area(&box);
		}
		box = list[n - 1].box;
		sa1[n - 2] = aabb_area(&box);
		for (long i = (long)n - 3; i >= 0; i--) { /* unions list[i], not list[i+1], as the reference does (:185-191) */
			box = aabb_union(&box, &list[i].box);
			sa1[i] = aabb_area(&box);
		}
		box = aabb_union(&box, &list[0].box); /* :192-197 */
		double all = aabb_area(&box);
		for (size_t i = 0; i < n - 1; i++) {
			int c0 = (int)i + 1, c1 = (int)(n - 1 - i);
			double cost = 1.0 + sa0[i] / all * c0 * 1.0 + sa1[i] / all * c1 * 1.0;
			if (cost < best_cost) {
				best_cost = cost;
				best_axis = axis;
				best_split_index = (long)i;
			}
		}
	}
	free(sa0);
	free(sa1);
	if (best_axis < 0) { /* all costs NaN/inf: reference asserts; fall back to a median split */
		best_axis = 0;
		best_split_index = (long)n / 2 - 1;
	}
	g_axis = best_axis;
	qsort(list, n, sizeof(bitem), cmp_center);
	size_t nl = (size_t)best_split_index + 1;
	int32_t l = build_sah(s, list, nl, depth + 1);
	int32_t r = build_sah(s, list + nl, n - nl, depth + 1);
	s->nodes[me].left = l;
	s->nodes[me].right = r;
	s->nodes[me].box = aabb_union(&s->nodes[l].box, &s->nodes[r].box); /* :228-231 */
	return me;
}

int orc_scene_build_bvh(orc_scene *s) /* object.cpp:97-139 */
{
	free(s->nodes);
	s->nodes = NULL;
	s->n_nodes = s->cap_nodes = 0;
	s->depth = 0;
	bitem *roots = (bitem *)malloc(s->n_objects * sizeof(bitem));
	uint32_t obj_depth = 0;
	for (uint32_t o = 0; o < s->n_objects; o++) {
		size_t n = s->object_begin[o + 1] - s->object_begin[o];
		if (n == 0) { free(roots); return 0; }
		bitem *list = (bitem *)malloc(n * sizeof(bitem));
		for (size_t i = 0; i < n; i++) {
			uint64_t f = s->object_begin[o] + i;
			list[i].id = ~(int32_t)f;
			list[i].box = s->poly_box[f];
		}
		s->depth = 0;
		int32_t r = build_sah(s, list, n, 1);
		if (s->depth > obj_depth) obj_depth = s->depth;
		free(list);
		roots[o].id = r;
		roots[o].box = s->nodes[r].box;
	}
	s->depth = 0;
	s->root = build_sah(s, roots, s->n_objects, 1);
	s->depth += obj_depth;
	free(roots);
	(void)item_box;
	return 1;
}
uint64_t orc_scene_bvh_nodes(const orc_scene *s) { return s->n_nodes; }
uint32_t orc_scene_bvh_depth(const orc_scene *s) { return s->depth; }

/* BVHNode::Hit (bvh.cpp:76-102): box test, both children with the same interval, smaller t,
 * right child on ties. Children that are polygons go to ConvexPolygon::Hit. */
static int bvh_hit(const orc_scene *s, int32_t id, v3 o, v3 d, double t_min, double t_max, hitrec *rec)
{
	if (id < 0)
		return polygon_hit(s, (uint64_t)~id, o, d, t_min, t_max, rec);
	const bnode *n = &s->nodes[id];
	if (aabb_hit(&n->box, o, d, t_min, t_max)) {
		hitrec lrec, rrec;
		int hl = bvh_hit(s, n->left, o, d, t_min, t_max, &lrec);
		int hr = bvh_hit(s, n->right, o, d, t_min, t_max, &rrec);
		if (hl && hr) {
			*rec = lrec.t < rrec.t ? lrec : rrec;
			return 1;
		} else if (hl) {
			*rec = lrec;
			return 1;
		} else if (hr) {
			*rec = rrec;
			return 1;
		}
		return 0;
	}
	return 0;
}

/* HittableList::Hit (hittablelist.cpp:3-17) over every polygon in scene order */
static int list_hit(const orc_scene *s, v3 o, v3 d, double t_min, double t_max, hitrec *rec, int lowest_wins)
{
	hitrec tmp;
	int hit_anything = 0;
	double closest = t_max;
	for (uint64_t f = 0; f < s->n_polys; f++) {
		if (polygon_hit(s, f, o, d, t_min, closest, &tmp)) {
			if (lowest_wins && hit_anything && tmp.t == closest)
				continue;
			hit_anything = 1;
			closest = tmp.t;
			*rec = tmp;
		}
	}
	return hit_anything;
}

static int world_hit(const orc_scene *s, int mode, v3 o, v3 d, double t_min, double t_max, hitrec *rec)
{
	if (mode == ORC_MODE_BVH_REFERENCE)
		return bvh_hit(s, s->root, o, d, t_min, t_max, rec); /* object.cpp:142-145 */
	return list_hit(s, o, d, t_min, t_max, rec, mode == ORC_MODE_LIST_LOWEST);
}

static void set_threads(int nthreads)
{
#ifdef _OPENMP
	if (nthreads > 0) omp_set_num_threads(nthreads);
	else omp_set_num_threads(omp_get_num_procs());
#else
	(void)nthreads;
#endif
}

uint64_t orc_trace_closest(const orc_scene *s, const float *rays, uint64_t n, int mode, int32_t *out_poly,
                           int32_t *out_tri, double *out_t, int nthreads)
{
	uint64_t hits = 0;
	set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : hits)
	for (int64_t i = 0; i < (int64_t)n; i++) {
		const float *r = rays + 8 * i;
		v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
		double t_max = isinf(r[7]) ? DBL_MAX : (double)r[7]; /* renderer.cpp:173 passes numeric_limits::max() */
		hitrec rec;
		int h = world_hit(s, mode, o, d, (double)r[3], t_max, &rec);
		if (out_poly) out_poly[i] = h ? (int32_t)rec.poly : -1;
		if (out_tri) out_tri[i] = h ? (int32_t)rec.tri : -1;
		if (out_t) out_t[i] = h ? rec.t : INFINITY;
		hits += h;
	}
	return hits;
}

uint64_t orc_trace_any(const orc_scene *s, const float *rays, uint64_t n, int mode, uint8_t *out_hit, int nthreads)
{
	uint64_t hits = 0;
	set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : hits)
	for (int64_t i = 0; i < (int64_t)n; i++) {
		const float *r = rays + 8 * i;
		v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
		double t_max = isinf(r[7]) ? DBL_MAX : (double)r[7];
		hitrec rec; /* Hittable::Occluded default = full closest hit, hittable.h:34-40 */
		int h = world_hit(s, mode, o, d, (double)r[3], t_max, &rec);
		out_hit[i] = (uint8_t)h;
		hits += h;
	}
	return hits;
}

/* --------------------------------------------------------------------------------------- */
/* Ordered, pruned traversal of an external flat BVH2 (counts the roofline's work units)    */
/* --------------------------------------------------------------------------------------- */
typedef struct {
	float lo0[3], hi0[3], lo1[3], hi1[3];
	int32_t child0, child1;
	uint32_t pad[2];
} flat_node; /* 64 bytes */

static inline int slab(const float lo[3], const float hi[3], v3 o, v3 inv, double t_min, double t_max, double *tn)
{
	double a = t_min, b = t_max;
	for (int i = 0; i < 3; i++) {
		double t0 = ((double)lo[i] - o.e[i]) * inv.e[i], t1 = ((double)hi[i] - o.e[i]) * inv.e[i];
		if (t0 > t1) { double x = t0; t0 = t1; t1 = x; }
		if (t0 > a) a = t0;
		if (t1 < b) b = t1;
	}
	*tn = a;
	/* conservative: tiny relative slack so that box rounding never culls what the triangle test accepts */
	return a <= b * (1.0 + 1e-12) + 1e-300;
}

uint64_t orc_trace_bvh2(const orc_scene *s, const void *nodes_v, uint64_t n_nodes, const uint32_t *tri_index,
                        const float *rays, uint64_t n, int32_t *out_poly, int32_t *out_tri, double *out_t,
                        uint64_t *node_visits, uint64_t *tri_tests, int nthreads)
{
	const flat_node *nodes = (const flat_node *)nodes_v;
	uint64_t hits = 0, nv = 0, ntt = 0;
	(void)n_nodes;
	set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : hits, nv, ntt)
	for (int64_t i = 0; i < (int64_t)n; i++) {
		const float *r = rays + 8 * i;
		v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
		v3 inv = V(1.0 / d.e[0], 1.0 / d.e[1], 1.0 / d.e[2]);
		double t_min = r[3], t_best = isinf(r[7]) ? DBL_MAX : (double)r[7];
		int64_t best_tri = -1;
		hitrec best;
		int32_t stack[256];
		int sp = 0;
		int32_t cur = 0;
		for (;;) {
			if (cur >= 0) {
				const flat_node *nd = &nodes[cur];
				nv++;
				double tn0, tn1;
				int h0 = slab(nd->lo0, nd->hi0, o, inv, t_min, t_best, &tn0);
				int h1 = slab(nd->lo1, nd->hi1, o, inv, t_min, t_best, &tn1);
				if (h0 && h1) {
					int32_t nearc = nd->child0, farc = nd->child1;
					if (tn1 < tn0) { nearc = nd->child1; farc = nd->child0; }
					stack[sp++] = farc;
					cur = nearc;
					continue;
				} else if (h0) { cur = nd->child0; continue; }
				else if (h1) { cur = nd->child1; continue; }
			} else {
				uint32_t c = (uint32_t)~cur;
				uint32_t first = c & 0x07FFFFFFu, count = (c >> 27) + 1;
				for (uint32_t k = 0; k < count; k++) {
					uint64_t t = tri_index[first + k];
					uint64_t f = s->tri_poly[t];
					uint64_t b = s->poly_begin[f], fi = t - s->tri_begin[f] + 1;
					hitrec rec;
					ntt++;
					if (hit_triangle(o, d, t_min, t_best, s->pv[b], s->pv[b + fi], s->pv[b + fi + 1], s->pn[b],
					                 s->pn[b + fi], s->pn[b + fi + 1], s->pvt[b], s->pvt[b + fi], s->pvt[b + fi + 1], s->face_normal[f], &rec, s->watertight)) {
						if (best_tri < 0 || rec.t < t_best || (int64_t)t < best_tri) { /* tie -> lowest fan triangle */
							best = rec;
							best.poly = (int64_t)f;
							best_tri = (int64_t)t;
							t_best = rec.t;
						}
					}
				}
			}
			if (sp == 0) break;
			cur = stack[--sp];
		}
		int h = best_tri >= 0;
		if (out_poly) out_poly[i] = h ? (int32_t)best.poly : -1;
		if (out_tri) out_tri[i] = h ? (int32_t)best_tri : -1;
		if (out_t) out_t[i] = h ? best.t : INFINITY;
		hits += h;
	}
	if (node_visits) *node_visits = nv;
	if (tri_tests) *tri_tests = ntt;
	return hits;
}

/* The same over the product's 8-wide tree (128-byte nodes, layout in include/mi355rt.h: rt_wide_node): child boxes are
 * dequantised exactly in double, children are visited front to back by entry distance, boxes beyond the closest hit so far
 * are skipped.  Counts the nodes fetched and the triangle records tested: the work the traversal kernel's memory traffic is
 * priced with.  The triangle test is hit_triangle (object.cpp:238-284) as everywhere in this file. */
typedef struct {
	float org[3];
	uint8_t e[3], imask;
	uint32_t child_base, tri_base;
	uint8_t meta[8];
	uint8_t qlo[3][8], qhi[3][8];
	uint8_t pad[48];
} wide_node;

uint64_t orc_trace_wide(const orc_scene *s, const void *nodes_v, uint64_t n_nodes, const uint32_t *tri_index,
                        const float *rays, uint64_t n, int32_t *out_poly, int32_t *out_tri, double *out_t,
                        uint64_t *node_visits, uint64_t *tri_tests, int nthreads)
{
	const wide_node *nodes = (const wide_node *)nodes_v;
	uint64_t hits = 0, nv = 0, ntt = 0;
	(void)n_nodes;
	set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : hits, nv, ntt)
	for (int64_t i = 0; i < (int64_t)n; i++) {
		const float *r = rays + 8 * i;
		v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
		v3 inv = V(1.0 / d.e[0], 1.0 / d.e[1], 1.0 / d.e[2]);
		double t_min = r[3], t_best = isinf(r[7]) ? DBL_MAX : (double)r[7];
		int64_t best_tri = -1;
		hitrec best;
		struct { uint32_t node; double tn; } stack[8 * 96];
		int sp = 0;
		stack[sp].node = 0; stack[sp].tn = -DBL_MAX; sp++;
		while (sp > 0) {
			sp--;
			if (stack[sp].tn > t_best) continue; /* found something closer since this node was pushed */
			const wide_node *nd = &nodes[stack[sp].node];
			nv++;
			double cell[3];
			for (int a = 0; a < 3; a++) cell[a] = ldexp(1.0, (int)nd->e[a] - 127);
			struct { double tn; int slot; } order[8];
			int nh = 0;
			for (int c = 0; c < 8; c++) {
				if (nd->qlo[0][c] > nd->qhi[0][c]) continue; /* empty slot */
				float lo[3], hi[3]; /* exact: org + q * 2^k with q < 256 fits a double; slab() takes floats rounded outward */
				double tn;
				for (int a = 0; a < 3; a++) {
					double l = (double)nd->org[a] + nd->qlo[a][c] * cell[a], h = (double)nd->org[a] + nd->qhi[a][c] * cell[a];
					lo[a] = (float)l; if ((double)lo[a] > l) lo[a] = nextafterf(lo[a], -INFINITY);
					hi[a] = (float)h; if ((double)hi[a] < h) hi[a] = nextafterf(hi[a], INFINITY);
				}
				if (!slab(lo, hi, o, inv, t_min, t_best, &tn)) continue;
				int k = nh++;
				while (k > 0 && order[k - 1].tn > tn) { order[k] = order[k - 1]; k--; }
				order[k].tn = tn; order[k].slot = c;
			}
			/* leaf children first (their triangles may shorten the ray before anything is pushed), nearest first */
			for (int k = 0; k < nh; k++) {
				int c = order[k].slot;
				if ((nd->imask >> c) & 1) continue;
				if (order[k].tn > t_best) continue;
				uint32_t first = nd->tri_base + (nd->meta[c] & 31u), count = nd->meta[c] >> 5;
				for (uint32_t q = 0; q < count; q++) {
					uint64_t t = tri_index[first + q];
					uint64_t f = s->tri_poly[t];
					uint64_t b = s->poly_begin[f], fi = t - s->tri_begin[f] + 1;
					hitrec rec;
					ntt++;
					if (hit_triangle(o, d, t_min, t_best, s->pv[b], s->pv[b + fi], s->pv[b + fi + 1], s->pn[b],
					                 s->pn[b + fi], s->pn[b + fi + 1], s->pvt[b], s->pvt[b + fi], s->pvt[b + fi + 1], s->face_normal[f], &rec, s->watertight)) {
						if (best_tri < 0 || rec.t < t_best || (int64_t)t < best_tri) { /* tie -> lowest fan triangle */
							best = rec;
							best.poly = (int64_t)f;
							best_tri = (int64_t)t;
							t_best = rec.t;
						}
					}
				}
			}
			for (int k = nh - 1; k >= 0; k--) { /* farthest first onto the stack: the nearest is popped next */
				int c = order[k].slot;
				if (!((nd->imask >> c) & 1)) continue;
				uint32_t rel = 0;
				for (int j = 0; j < c; j++) rel += (nd->imask >> j) & 1;
				stack[sp].node = nd->child_base + rel; stack[sp].tn = order[k].tn; sp++;
			}
		}
		int h = best_tri >= 0;
		if (out_poly) out_poly[i] = h ? (int32_t)best.poly : -1;
		if (out_tri) out_tri[i] = h ? (int32_t)best_tri : -1;
		if (out_t) out_t[i] = h ? best.t : INFINITY;
		hits += h;
	}
	if (node_visits) *node_visits = nv;
	if (tri_tests) *tri_tests = ntt;
	return hits;
}

/* ======================================================================================= */
/* Materials (materialnode.cpp) and Renderer::LoadMaterials (renderer.cpp:24-44)            */
/* ======================================================================================= */
static void free_graphs(orc_scene *s)
{
	free(s->prog_begin); free(s->prog_len); free(s->code); free(s->g_spectra); free(s->g_vectors); free(s->g_scalars);
	for (uint32_t i = 0; i < s->n_g_textures; i++) free(s->g_tex[i]);
	free(s->g_tex); free(s->g_tex_w); free(s->g_tex_h); free(s->g_tex_bpp);
	s->prog_begin = s->prog_len = s->code = NULL;
	s->g_spectra = s->g_vectors = s->g_scalars = NULL;
	s->g_tex = NULL; s->g_tex_w = s->g_tex_h = s->g_tex_bpp = NULL;
	s->n_code = s->n_g_spectra = s->n_g_vectors = s->n_g_scalars = s->n_g_textures = 0;
}

/* per-hit graphs for slot a of some objects (see graph_eval); programs: (object, begin, len) triples; textures given as parallel arrays.
 * Parity unpinned: materialnode.cpp does not compile here. */
int orc_scene_set_material_graphs(orc_scene *s, const uint32_t *programs, uint32_t n_programs, const uint32_t *code, uint32_t n_code,
                                  const double *spectra, uint32_t n_spectra, const double *vectors, uint32_t n_vectors,
                                  const double *scalars, uint32_t n_scalars, const uint8_t *const *tex, const int *tex_w,
                                  const int *tex_h, const int *tex_bpp, uint32_t n_textures)
{
	free_graphs(s);
	if (n_programs == 0) return 0;
	s->prog_begin = (uint32_t *)calloc(s->n_objects, sizeof(uint32_t));
	s->prog_len = (uint32_t *)calloc(s->n_objects, sizeof(uint32_t));
	for (uint32_t i = 0; i < n_programs; i++) {
		if (programs[3 * i] >= s->n_objects || programs[3 * i + 1] + programs[3 * i + 2] > n_code) return -1;
		s->prog_begin[programs[3 * i]] = programs[3 * i + 1];
		s->prog_len[programs[3 * i]] = programs[3 * i + 2];
	}
	s->code = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (n_code ? n_code : 1));
	memcpy(s->code, code, sizeof(uint32_t) * 2 * n_code);
	s->n_code = n_code;
	s->g_spectra = (double *)malloc(sizeof(double) * ORC_NBINS * (n_spectra ? n_spectra : 1));
	memcpy(s->g_spectra, spectra, sizeof(double) * ORC_NBINS * n_spectra);
	s->g_vectors = (double *)malloc(sizeof(double) * 3 * (n_vectors ? n_vectors : 1));
	memcpy(s->g_vectors, vectors, sizeof(double) * 3 * n_vectors);
	s->g_scalars = (double *)malloc(sizeof(double) * (n_scalars ? n_scalars : 1));
	memcpy(s->g_scalars, scalars, sizeof(double) * n_scalars);
	s->n_g_spectra = n_spectra; s->n_g_vectors = n_vectors; s->n_g_scalars = n_scalars; s->n_g_textures = n_textures;
	s->g_tex = (uint8_t **)calloc(n_textures ? n_textures : 1, sizeof(uint8_t *));
	s->g_tex_w = (int *)calloc(n_textures ? n_textures : 1, sizeof(int));
	s->g_tex_h = (int *)calloc(n_textures ? n_textures : 1, sizeof(int));
	s->g_tex_bpp = (int *)calloc(n_textures ? n_textures : 1, sizeof(int));
	for (uint32_t i = 0; i < n_textures; i++) {
		size_t bytes = (size_t)tex_w[i] * tex_h[i] * tex_bpp[i];
		s->g_tex[i] = (uint8_t *)malloc(bytes);
		memcpy(s->g_tex[i], tex[i], bytes);
		s->g_tex_w[i] = tex_w[i]; s->g_tex_h[i] = tex_h[i]; s->g_tex_bpp[i] = tex_bpp[i];
	}
	return 0;
}

/* stands for stbi_load into the globals env_mapping_texture/width/height/bpp and env_brightness (scene.cpp:245, scene.h:20-22);
 * tex == NULL removes the environment */
int orc_scene_set_environment(orc_scene *s, const uint8_t *tex, int w, int h, int bpp, double brightness)
{
	free(s->env);
	s->env = NULL;
	if (tex == NULL) return 0;
	if (w <= 0 || h <= 0 || bpp < 3) return -1;
	s->env = (uint8_t *)malloc((size_t)w * h * bpp);
	if (!s->env) return -1;
	memcpy(s->env, tex, (size_t)w * h * bpp);
	s->env_w = w; s->env_h = h; s->env_bpp = bpp; s->env_brightness = brightness;
	return 0;
}

int orc_scene_set_materials(orc_scene *s, const orc_material *per_object, uint32_t n_objects)
{
	if (n_objects != s->n_objects) return 0;
	free(s->mats);
	free(s->lights);
	s->mats = (orc_material *)malloc(n_objects * sizeof(orc_material));
	memcpy(s->mats, per_object, n_objects * sizeof(orc_material));
	s->lights = (uint64_t *)malloc((s->n_polys ? s->n_polys : 1) * sizeof(uint64_t));
	s->n_lights = 0;
	for (uint32_t o = 0; o < n_objects; o++)
		if (per_object[o].light_flag)
			for (uint64_t f = s->object_begin[o]; f < s->object_begin[o + 1]; f++)
				s->lights[s->n_lights++] = f;
	return s->n_lights != 0;
}
uint64_t orc_scene_n_lights(const orc_scene *s) { return s->n_lights; }

typedef struct { double min_wl, max_wl, central_wl; } band;

static double mat_emitted(const orc_material *m, const band *b) /* materialnode.cpp:215-218, 739-749 */
{
	if (m->type == ORC_MAT_DIFFUSE_LIGHT)
		return orc_spectrum_integrate(m->a, b->min_wl, b->max_wl);
	return 0.0;
}

/* GGXReflectionNode made well defined -- parity unpinned and a deviation: in the reference `alpha` is read uninitialised,
 * BSDF() and PDF() return nothing, and dot(micro_normal, rec.normal) mixes a local with a world vector (materialnode.cpp:647-695).
 * What is kept is SampleBSDF as written there with everything expressed in the shading frame (z = rec.normal): the micro-normal
 * distribution d = alpha^2 x / (pi cos^4 (alpha^2 + tan^2)^2) with x = max(m.z, 0) (:669-675), pdf = d |m.z| / (4 |m.z|) (:676-677),
 * BSDF = d / (4 |vi.z vo.z|) with fresnel = g = 1 (:665-666,682), ratio = |vo.m| / |vo.z vi.z m.z| (:686).  BSDF() and PDF() are
 * defined as the same expressions for the half vector m = unit(vi + vo), and as 0 unless vi and vo lie on the same side of the
 * surface (a reflection); a sampled direction that ends up on the other side carries no energy.  alpha comes from the material record. */
static double ggx_d(double alpha, double cos_theta)
{
	if (!(cos_theta > 0.0)) return 0.0; /* x = 0 there: the numerator vanishes (and the denominator with it at the horizon) */
	const double x = cos_theta;
	const double cos2 = cos_theta * cos_theta, sin_theta = sqrt(1.0 - cos2 < 0.0 ? 0.0 : 1.0 - cos2), tan_theta = sin_theta / cos_theta;
	return alpha * alpha * x / (M_PI * cos2 * cos2 * pow(alpha * alpha + tan_theta * tan_theta, 2));
}
static double ggx_half_z(v3 vi, v3 vo) { return fabs(vunit(vadd(vi, vo)).e[2]); } /* the half vector on the side of the normal */

static double mat_bsdf(const orc_material *m, v3 vi, double wli, v3 vo)
{
	switch (m->type) {
	case ORC_MAT_GGX:
		if (!(vi.e[2] * vo.e[2] > 0.0)) return 0.0;
		return ggx_d(m->roughness, ggx_half_z(vi, vo)) / (4.0 * fabs(vi.e[2] * vo.e[2]));
	case ORC_MAT_LAMBERTIAN: /* materialnode.cpp:278-288 */
		if (vi.e[2] < 0.0) return 0.0;
		return orc_spectrum_get(m->a, wli) / M_PI;
	case ORC_MAT_CONDUCTOR:      /* :503-508 */
	case ORC_MAT_COLORED_METAL:  /* :585-590 */
		if (-vi.e[0] == vo.e[0] && -vi.e[1] == vo.e[1] && vi.e[2] == vo.e[2]) return INFINITY;
		return 0.0;
	default: /* Dielectric :408-413, DiffuseLight/base :207-210 */
		return 0.0;
	}
}

static double mat_pdf(const orc_material *m, v3 vi, v3 vo)
{
	switch (m->type) {
	case ORC_MAT_GGX:
		if (!(vi.e[2] * vo.e[2] > 0.0)) return 0.0;
		return ggx_d(m->roughness, ggx_half_z(vi, vo)) / 4.0;
	case ORC_MAT_LAMBERTIAN: /* :289-295 */
		if (vi.e[2] < 0.0) return 0.0;
		return vi.e[2] / M_PI;
	case ORC_MAT_CONDUCTOR:     /* :509-514 */
	case ORC_MAT_COLORED_METAL: /* :591-596 */
		if (-vi.e[0] == vo.e[0] && -vi.e[1] == vo.e[1] && vi.e[2] == vo.e[2]) return INFINITY;
		return 0.0;
	default:
		return 0.0;
	}
}

/* light_type: RayType of the call (material.h): 0 = ImportanceType (paths from the eye), 1 = LightType (the light subpath of
 * NaiveBDPT, renderer.cpp:589); only the refraction branch of the dielectric tells them apart (materialnode.cpp:399-403) */
static int mat_sample_t(const orc_material *m, v3 normal, const onb *uvw, v3 vo, double wlo, uint64_t *rng, v3 *vi,
                        double *ratio, double *bsdf, double *pdfval, int light_type)
{
	switch (m->type) {
	case ORC_MAT_LAMBERTIAN: { /* :262-276 with CosinePdf pdf.cpp:70-84 */
		onb pb = onb_from_w(normal);
		double r1 = orc_rng_f64(rng), r2 = orc_rng_f64(rng);
		double c[3];
		orc_cosine_hemisphere(r1, r2, c);
		v3 gen = onb_l2w(&pb, V(c[0], c[1], c[2]));
		double cos_theta = vdot(gen, pb.axis[2]);
		*pdfval = cos_theta >= 0.0 ? cos_theta / M_PI : 0.0;
		*vi = onb_w2l(uvw, gen);
		*bsdf = mat_bsdf(m, *vi, wlo, vo);
		*ratio = *bsdf / *pdfval;
		return 1;
	}
	case ORC_MAT_DIELECTRIC: { /* :354-407 */
		double ref_idx = orc_spectrum_get(m->a, wlo);
		double surface_scale = orc_spectrum_get(m->b, wlo);
		double cos_o = vo.e[2];
		v3 nrm = V(0, 0, 1);
		double n1, n2;
		if (cos_o >= 0.0) { n1 = 1.0; n2 = ref_idx; }
		else { nrm = V(0, 0, -1); n1 = ref_idx; n2 = 1.0; cos_o = fabs(cos_o); }
		double rel = n2 / n1;
		const double fresnel = orc_rfresnel(cos_o, rel);
		double r = orc_rng_f64(rng);
		*pdfval = INFINITY;
		*bsdf = INFINITY;
		if (r <= fresnel) {
			*vi = V(-vo.e[0], -vo.e[1], vo.e[2]);
			*ratio = surface_scale / cos_o;
		} else {
			double sin_o = sqrt(1.0 - cos_o * cos_o);
			double sin_t = sin_o / rel;
			double cos_t = sqrt(1.0 - sin_t * sin_t);
			*vi = vsub(vscale(-cos_t, nrm), vscale(sin_t, vunit(V(vo.e[0], vo.e[1], 0))));
			*ratio = light_type ? surface_scale * pow(rel, 3) / cos_t : surface_scale / pow(rel, 3) / cos_t;
		}
		return 1;
	}
	case ORC_MAT_CONDUCTOR: { /* :473-502 */
		double ref_idx = orc_spectrum_get(m->a, wlo), kv = orc_spectrum_get(m->b, wlo);
		double cos_o = vo.e[2];
		cplx n1, n2;
		if (cos_o >= 0.0) { n1.re = 1.0; n1.im = 0; n2.re = ref_idx; n2.im = kv; }
		else { n1.re = ref_idx; n1.im = kv; n2.re = 1.0; n2.im = 0; cos_o = fabs(cos_o); }
		cplx rel = cdivc(n2, n1);
		const double fresnel = orc_cfresnel(cos_o, rel.re, rel.im);
		*vi = V(-vo.e[0], -vo.e[1], vo.e[2]);
		*pdfval = INFINITY;
		*bsdf = INFINITY;
		*ratio = fresnel / cos_o;
		return 1;
	}
	case ORC_MAT_GGX: { /* :647-689, see ggx_d */
		const double alpha = m->roughness;
		double phi = 2 * M_PI * orc_rng_f64(rng);
		double u = orc_rng_f64(rng);
		double cos_theta = sqrt((1.0 - u) / ((alpha * alpha - 1.0) * u + 1));
		double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
		v3 mn = V(sin_theta * cos(phi), sin_theta * sin(phi), cos_theta);
		double d = ggx_d(alpha, cos_theta);
		double pm = d * fabs(mn.e[2]);
		*vi = vadd(vneg(vo), vscale(2.0 * vdot(vo, mn), mn)); /* reflect2, material.cpp:98-101 */
		*pdfval = pm / (4.0 * fabs(mn.e[2]));
		*bsdf = d / (4.0 * fabs(vi->e[2] * vo.e[2]));
		*ratio = fabs(vdot(vo, mn)) / fabs(vo.e[2] * vi->e[2] * mn.e[2]);
		if (!(vi->e[2] * vo.e[2] > 0.0)) { *bsdf = 0.0; *ratio = 0.0; } /* reflected below the surface: no energy */
		return 1;
	}
	case ORC_MAT_COLORED_METAL: /* :564-584 */
		*vi = V(-vo.e[0], -vo.e[1], vo.e[2]);
		*bsdf = INFINITY;
		*pdfval = INFINITY;
		*ratio = orc_spectrum_get(m->a, wlo) / fabs(vo.e[2]);
		return 1;
	default: /* DiffuseLight and unconnected outputs: base class, :203-206 */
		return 0;
	}
}

static int mat_sample(const orc_material *m, v3 normal, const onb *uvw, v3 vo, double wlo, uint64_t *rng, v3 *vi,
                      double *ratio, double *bsdf, double *pdfval)
{
	return mat_sample_t(m, normal, uvw, vo, wlo, rng, vi, ratio, bsdf, pdfval, 0);
}

/* ConvexPolygon::GetRandomPointOnPolygon (object.cpp:340-372) */
static void random_point_on_polygon(const orc_scene *s, uint64_t f, uint64_t *rng, v3 *p, double *area, v3 *normal)
{
	double r = orc_rng_f64(rng);
	uint64_t t0 = s->tri_begin[f], nt = s->tri_begin[f + 1] - t0, b = s->poly_begin[f];
	for (uint64_t ti = 0; ti < nt; ti++) {
		int a = 0;
		if (ti == 0) { if (r < s->tri_cum[t0]) a = 1; }
		else if (r >= s->tri_cum[t0 + ti - 1] && r < s->tri_cum[t0 + ti]) a = 1;
		if (a) {
			uint64_t vi = ti + 1;
			v3 va = vsub(s->pv[b + vi], s->pv[b]), vb = vsub(s->pv[b + vi + 1], s->pv[b]);
			double x = orc_rng_f64(rng), y = orc_rng_f64(rng);
			if (x + y >= 1.0) { x = 1.0 - x; y = 1.0 - y; }
			*p = vadd(vadd(s->pv[b], vscale(x, va)), vscale(y, vb));
			*area = s->polygon_area[f];
			*normal = s->face_normal[f];
			return;
		}
	}
	/* unreachable: the last cumulative sum is forced to 1.0 and r < 1 */
	*p = s->pv[b];
	*area = s->polygon_area[f];
	*normal = s->face_normal[f];
}

/* Sphere::Hit (object.cpp:43-75) for the dome of renderer.cpp:169-171: centre 0, radius 100.  Returns t only. */
int orc_sphere_hit(const double o[3], const double d[3], double t_min, double t_max, double *t)
{
	const double radius = 100.0;
	double blen = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
	double a = pow(blen, 2.0);
	double b_prime = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
	double c = pow(sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 2.0) - pow(radius, 2.0);
	double disc = pow(b_prime, 2.0) - a * c;
	if (disc < 0.0) return 0;
	double t1 = (-b_prime - sqrt(disc)) / a;
	if (t1 >= t_min && t1 <= t_max) { *t = t1; return 1; }
	double t2 = (-b_prime + sqrt(disc)) / a;
	if (t2 >= t_min && t2 <= t_max) { *t = t2; return 1; }
	return 0;
}

/* renderer.cpp:180-191: direction of the dome point -> texel -> RGBtoSpectrum(rgb * brightness).get(central_wl).
 * The reference's x can equal the width (it then reads the first texel of the next row) and, on the last row or for
 * theta = pi, an address past the texture (undefined there); here such an address reads the last texel. */
double orc_environment_lookup(const uint8_t *tex, int w, int h, int bpp, double brightness, const double o[3], const double d[3],
                              double t, double central_wl)
{
	double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
	double l = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
	p[0] /= l; p[1] /= l; p[2] /= l;
	double theta = acos(p[1]);
	double phi = atan2(p[0] / sin(theta), p[2] / sin(theta));
	phi += M_PI;
	int x = w - (int)(phi / (2.0 * M_PI) * w);
	int y = (int)(theta / M_PI * h);
	long idx = (long)x + (long)y * w;
	if (idx < 0) idx = 0;
	if (idx > (long)w * h - 1) idx = (long)w * h - 1;
	double rgb[3], spec[ORC_NBINS];
	for (int c = 0; c < 3; c++) rgb[c] = (double)tex[(long)bpp * idx + c] / 255.0 * brightness;
	orc_rgb_to_spectrum(rgb, spec);
	return orc_spectrum_get(spec, central_wl);
}

static double geometry_term(const hitrec *rec0, const hitrec *rec1) /* renderer.cpp:784-789 */
{
	v3 v = vsub(rec1->p, rec0->p);
	v3 dir = vunit(vsub(rec1->p, rec0->p));
	return fabs(vdot(dir, rec0->normal) * vdot(dir, rec1->normal) / vlen2(v));
}

/* ---- per-hit material graphs: the Compute() chains of materialnode.cpp that read Argument::vt, as a postfix program.
 * Ops (arg): 0 SPEC k  push spectrum k | 1 VEC k  push vector k | 2 UV  push rec.vt (UVNode, :893-896) | 3 ZERO_S | 4 ZERO_V |
 * 5 IMAGE k  uv -> texel rgb (ImageTextureNode, :1001-1018) | 6 CHECKER k  uv -> 0/1 with size scalars[k] (:1046-1063) |
 * 7 RGB2SPEC (spectrum.cpp:219-232) | 8 ADD_S | 9 MUL_S | 10 ADD_V | 11 MUL_V (elementwise, vec3.h:93) |
 * 12 ACCESS i  vector -> its component i (AccessVec3ComponentNode, :1293-1301) | 13 COMBINE  three scalars -> vector (:1333-1347) |
 * 14 NOISE1 k  scalar -> ValueNoiseNode (:1366-1384), scalars[k] = size, then size + 1 table values |
 * 15 NOISE2 k  u, v -> ValueNoise2DNode (:1447-1478), scalars[k] = u_size, v_size, then (u_size + 1)(v_size + 1) values |
 * 16 ROTATE  v, n, theta -> RodriguesRotationNode (:1536-1573) | 17 SCALAR k  push scalars[k].
 * A scalar sits in s[0] (ADD_S / MUL_S serve scalars too). */
typedef struct { int is_vec; double s[ORC_NBINS]; v3 v; } gval;
static void graph_eval(const orc_scene *s, uint32_t begin, uint32_t len, v3 vt, double out[ORC_NBINS])
{
	gval st[16];
	int sp = 0;
	for (uint32_t pc = begin; pc < begin + len; pc++) {
		uint32_t op = s->code[2 * pc], arg = s->code[2 * pc + 1];
		switch (op) {
		case 0: st[sp].is_vec = 0; memcpy(st[sp].s, s->g_spectra + (size_t)arg * ORC_NBINS, sizeof(double) * ORC_NBINS); sp++; break;
		case 1: st[sp].is_vec = 1; st[sp].v = vload(s->g_vectors + 3 * (size_t)arg); sp++; break;
		case 2: st[sp].is_vec = 1; st[sp].v = vt; sp++; break;
		case 3: st[sp].is_vec = 0; memset(st[sp].s, 0, sizeof(st[sp].s)); sp++; break;
		case 4: st[sp].is_vec = 1; st[sp].v = V(0, 0, 0); sp++; break;
		case 5: { /* the reference reads past the texture for uv outside [0, 1) (undefined); such an address reads the nearest end */
			v3 uv = st[sp - 1].v;
			int w = s->g_tex_w[arg], h = s->g_tex_h[arg], bpp = s->g_tex_bpp[arg];
			int x = (int)((double)w * uv.e[0]);
			int y = (int)((double)h * (1.0 - uv.e[1]));
			long idx = (long)x + (long)y * w;
			if (idx < 0) idx = 0;
			if (idx > (long)w * h - 1) idx = (long)w * h - 1;
			const uint8_t *t = s->g_tex[arg] + (long)bpp * idx;
			st[sp - 1].v = V((double)t[0] / 255.0, (double)t[1] / 255.0, (double)t[2] / 255.0);
			break;
		}
		case 6: {
			v3 uv = st[sp - 1].v;
			double size = s->g_scalars[arg];
			int x = (int)((uv.e[0] - fmod(uv.e[0], size)) / size);
			int y = (int)((uv.e[1] - fmod(uv.e[1], size)) / size);
			st[sp - 1].v = (x % 2 == y % 2) ? V(1.0, 1.0, 1.0) : V(0.0, 0.0, 0.0);
			break;
		}
		case 7: { double rgb[3]; vstore(rgb, st[sp - 1].v); st[sp - 1].is_vec = 0; orc_rgb_to_spectrum(rgb, st[sp - 1].s); break; }
		case 8: for (int i = 0; i < ORC_NBINS; i++) st[sp - 2].s[i] = st[sp - 2].s[i] + st[sp - 1].s[i]; sp--; break;
		case 9: for (int i = 0; i < ORC_NBINS; i++) st[sp - 2].s[i] = st[sp - 2].s[i] * st[sp - 1].s[i]; sp--; break;
		case 10: st[sp - 2].v = vadd(st[sp - 2].v, st[sp - 1].v); sp--; break;
		case 11: for (int i = 0; i < 3; i++) st[sp - 2].v.e[i] *= st[sp - 1].v.e[i]; sp--; break;
		case 12: { double c = st[sp - 1].v.e[arg]; st[sp - 1].is_vec = 0; memset(st[sp - 1].s, 0, sizeof(st[sp - 1].s)); st[sp - 1].s[0] = c; break; }
		case 13: st[sp - 3].v = V(st[sp - 3].s[0], st[sp - 2].s[0], st[sp - 1].s[0]); st[sp - 3].is_vec = 1; sp -= 2; break;
		case 14: {
			const double *r = s->g_scalars + arg + 1;
			int n = (int)s->g_scalars[arg]; /* r has n + 1 entries */
			double u = fmod(st[sp - 1].s[0], 1.0);
			if (u < 0.0) u += 1.0;
			double us = 1.0 / n, rem = fmod(u, us);
			int x0 = (int)((u - rem) / us), x1 = x0 + 1;
			if (x1 > n) { x1 = n; if (x0 > n) x0 = n; } /* only reachable through rounding at u -> 1; the reference asserts */
			double t = rem / us;
			st[sp - 1].s[0] = (1.0 - t) * r[x0] + t * r[x1];
			break;
		}
		case 15: {
			unsigned nu = (unsigned)s->g_scalars[arg], nv = (unsigned)s->g_scalars[arg + 1];
			const double *r = s->g_scalars + arg + 2;
			double u = fmod(st[sp - 2].s[0], 1.0), v = fmod(st[sp - 1].s[0], 1.0);
			if (u < 0.0) u += 1.0;
			if (v < 0.0) v += 1.0;
			double us = 1.0 / nu, vs = 1.0 / nv, urem = fmod(u, us), vrem = fmod(v, vs);
			unsigned u0 = (unsigned)((u - urem) / us), v0 = (unsigned)((v - vrem) / vs);
			if (u0 > nu - 1) u0 = nu - 1; /* as above: the reference would read past its table */
			if (v0 > nv - 1) v0 = nv - 1;
			unsigned u1 = u0 + 1, v1 = v0 + 1;
			double ut = urem / us, vt2 = vrem / vs;
			ut = ut * ut * (3.0 - 2.0 * ut);
			vt2 = vt2 * vt2 * (3.0 - 2.0 * vt2);
			double a = (1.0 - ut) * r[u0 * (nv + 1) + v0] + ut * r[u1 * (nv + 1) + v0];
			double b = (1.0 - ut) * r[u0 * (nv + 1) + v1] + ut * r[u1 * (nv + 1) + v1];
			st[sp - 2].s[0] = (1.0 - vt2) * a + vt2 * b;
			sp--;
			break;
		}
		case 16:
			st[sp - 3].v = vrodrigues(st[sp - 3].v, vunit(st[sp - 2].v), st[sp - 1].s[0]);
			sp -= 2;
			break;
		case 17: st[sp].is_vec = 0; memset(st[sp].s, 0, sizeof(st[sp].s)); st[sp].s[0] = s->g_scalars[arg]; sp++; break;
		default: break;
		}
	}
	memcpy(out, st[0].s, sizeof(double) * ORC_NBINS);
}

/* slot a of `object` at texture coordinate vt, for tests of the host's graph compiler; 0 when the object has no program */
int orc_graph_eval(const orc_scene *s, uint32_t object, const double vt[3], double out[ORC_NBINS])
{
	if (!s->prog_len || s->prog_len[object] == 0) return 0;
	graph_eval(s, s->prog_begin[object], s->prog_len[object], V(vt[0], vt[1], vt[2]), out);
	return 1;
}

/* the material of a hit; when the object's slot a is a per-hit graph, a copy with that slot evaluated at rec->vt (the copies rotate
 * through a small thread-local ring: at most a handful of them are alive at once inside one integrator step) */
static const orc_material *mat_at(const orc_scene *s, uint32_t object, v3 vt)
{
	static _Thread_local orc_material ring[16];
	static _Thread_local int next = 0;
	if (!s->prog_len || s->prog_len[object] == 0) return &s->mats[object];
	orc_material *m = &ring[next];
	next = (next + 1) & 15;
	*m = s->mats[object];
	graph_eval(s, s->prog_begin[object], s->prog_len[object], vt, m->a);
	return m;
}
static inline const orc_material *rec_mat(const orc_scene *s, const hitrec *rec) { return mat_at(s, s->object_id[rec->poly], rec->vt); }

#define WORLD_MODE ORC_MODE_BVH_REFERENCE

static double naive_path_tracing(const orc_scene *s, v3 o, v3 d, const band *b, uint64_t *rng, orc_render_stats *st) /* renderer.cpp:163-224 */
{
	hitrec rec;
	double radiance = 0.0, beta = 1.0;
	for (;;) {
		st->closest_rays++;
		if (!world_hit(s, WORLD_MODE, o, d, 0.001, DBL_MAX, &rec)) {
			if (s->env == NULL) /* :176-177 */
				break;
			double t;
			if (orc_sphere_hit(o.e, d.e, 0.001, DBL_MAX, &t)) /* :178 */
				radiance += beta * orc_environment_lookup(s->env, s->env_w, s->env_h, s->env_bpp, s->env_brightness, o.e, d.e, t, b->central_wl);
			break;
		}
		const orc_material *m = rec_mat(s, &rec);
		radiance += beta * mat_emitted(m, b);
		onb uvw = onb_from_w(rec.normal);
		v3 vi;
		double ratio, bsdf, pdf;
		if (!mat_sample(m, rec.normal, &uvw, onb_w2l(&uvw, vneg(d)), b->central_wl, rng, &vi, &ratio, &bsdf, &pdf))
			break;
		beta *= (ratio * fabs(vi.e[2]));
		double prr = 0.5;
		if (orc_rng_f64(rng) < prr)
			break;
		beta /= (1 - prr);
		o = rec.p;
		d = onb_l2w(&uvw, vi); /* not re-normalised, :217 */
	}
	return radiance;
}

static double nee_path_tracing(const orc_scene *s, v3 o, v3 d, const band *b, uint64_t *rng, orc_render_stats *st) /* :227-325 */
{
	hitrec rec;
	int bounce = 0;
	double radiance = 0.0, beta = 1.0;
	for (;;) {
		st->closest_rays++;
		if (!world_hit(s, WORLD_MODE, o, d, 0.0001, DBL_MAX, &rec))
			break;
		const orc_material *m = rec_mat(s, &rec);
		if (m->light_flag) {
			if (bounce != 0)
				break;
			radiance += mat_emitted(m, b);
		}
		{
			uint64_t sel = rng_below(rng, (uint32_t)s->n_lights);
			uint64_t lf = s->lights[sel];
			v3 p, light_normal;
			double area;
			random_point_on_polygon(s, lf, rng, &p, &area, &light_normal);
			v3 sd = vunit(vsub(p, rec.p));
			hitrec tmp;
			st->shadow_rays++;
			if (!world_hit(s, WORLD_MODE, rec.p, sd, 0.0001, vlen(vsub(p, rec.p)) - 0.0001, &tmp)) { /* :263 */
				onb uvw_ = onb_from_w(rec.normal);
				const v3 vi = onb_w2l(&uvw_, vunit(vsub(p, rec.p)));
				const v3 vo = onb_w2l(&uvw_, vneg(d));
				const double BSDF = mat_bsdf(m, vi, b->central_wl, vo);
				const double G = fabs(vdot(sd, s->face_normal[lf]) * vdot(sd, rec.normal) / vlen2(vsub(p, rec.p)));
				const double pdfarea = 1.0 / area / (double)s->n_lights;
				hitrec light_rec;
				if (polygon_hit(s, lf, rec.p, sd, 0.0001, vlen(vsub(p, rec.p)) + 0.0001, &light_rec)) {
					const double light = mat_emitted(rec_mat(s, &light_rec), b);
					radiance += BSDF * beta * light * G / pdfarea;
				}
			}
		}
		v3 gvi;
		double ratio, bsdf, pdf;
		onb uvw = onb_from_w(rec.normal);
		int respawn = mat_sample(m, rec.normal, &uvw, onb_w2l(&uvw, vneg(d)), b->central_wl, rng, &gvi, &ratio, &bsdf, &pdf);
		if (respawn)
			beta *= ratio * fabs(gvi.e[2]);
		double prr = 0.5;
		if (bounce > 6)
			prr = 0.9;
		if (orc_rng_f64(rng) < prr)
			break;
		beta /= (1.0 - prr);
		if (!respawn)
			break;
		o = rec.p;
		d = vunit(onb_l2w(&uvw, gvi));
		bounce++;
	}
	return radiance;
}

static double mis_path_tracing(const orc_scene *s, v3 o, v3 d, const band *b, uint64_t *rng, orc_render_stats *st) /* :329-475 */
{
	hitrec rec;
	int bounce = 0;
	double radiance = 0.0, beta = 1.0;
	for (;;) {
		st->closest_rays++;
		if (!world_hit(s, WORLD_MODE, o, d, 0.001, DBL_MAX, &rec))
			break;
		const orc_material *m = rec_mat(s, &rec);
		if (m->light_flag) {
			if (bounce != 0)
				break;
			radiance += mat_emitted(m, b);
		}
		uint64_t sel = rng_below(rng, (uint32_t)s->n_lights);
		uint64_t lf = s->lights[sel];
		double addition = 0.0;
		{ /* sample light, :361-401 */
			v3 pl, light_normal;
			double area;
			random_point_on_polygon(s, lf, rng, &pl, &area, &light_normal);
			v3 sd = vunit(vsub(pl, rec.p));
			hitrec light_rec;
			st->closest_rays++;
			int light_hit = world_hit(s, WORLD_MODE, rec.p, sd, 0.001, vlen(vsub(pl, rec.p)) + 0.001, &light_rec);
			if (light_hit && (uint64_t)light_rec.poly == lf) {
				onb uvw_ = onb_from_w(rec.normal);
				const v3 vi = onb_w2l(&uvw_, vunit(vsub(pl, rec.p)));
				const v3 vo = onb_w2l(&uvw_, vneg(d));
				const double BSDF = mat_bsdf(m, vi, b->central_wl, vo);
				const double G = geometry_term(&light_rec, &rec);
				const double light_pdf_area = 1.0 / area / (double)s->n_lights;
				const double light = mat_emitted(rec_mat(s, &light_rec), b);
				const double light_scale = beta * BSDF * G / light_pdf_area;
				const double spdf = mat_pdf(m, vi, vo);
				const double spdf_area = spdf * G / fabs(vdot(sd, rec.normal));
				double w = orc_power_heuristic(light_pdf_area, spdf_area, 2.0);
				addition += w * light_scale * light;
			}
		}
		{ /* sample bsdf, :404-433 */
			double ratio, bsdf, spdf;
			onb uvw = onb_from_w(rec.normal);
			v3 gvi;
			int respawn = mat_sample(m, rec.normal, &uvw, onb_w2l(&uvw, vneg(d)), b->central_wl, rng, &gvi, &ratio, &bsdf, &spdf);
			if (respawn) {
				v3 sd = onb_l2w(&uvw, gvi);
				hitrec light_rec;
				st->closest_rays++;
				if (world_hit(s, WORLD_MODE, rec.p, sd, 0.001, DBL_MAX, &light_rec) && (uint64_t)light_rec.poly == lf) {
					const double light = mat_emitted(rec_mat(s, &light_rec), b);
					const double G = geometry_term(&light_rec, &rec);
					const double light_pdf_area = 1.0 / s->polygon_area[lf] / (double)s->n_lights;
					const double bsdf_scale = beta * ratio * fabs(gvi.e[2]);
					const double bspa = G * spdf / fabs(gvi.e[2]);
					const double w = orc_power_heuristic(bspa, light_pdf_area, 2.0);
					addition += w * bsdf_scale * light;
				}
			}
		}
		radiance += addition;
		v3 gvi;
		double ratio, bsdf, pdf;
		onb uvw = onb_from_w(rec.normal);
		int respawn = mat_sample(m, rec.normal, &uvw, onb_w2l(&uvw, vneg(d)), b->central_wl, rng, &gvi, &ratio, &bsdf, &pdf);
		if (respawn)
			beta *= ratio * fabs(gvi.e[2]);
		double prr = 0.5;
		if (bounce > 6)
			prr = 0.9;
		if (orc_rng_f64(rng) < prr)
			break;
		beta /= (1.0 - prr);
		if (!respawn)
			break;
		o = rec.p;
		d = vunit(onb_l2w(&uvw, gvi));
		bounce++;
	}
	return radiance;
}

/* NaiveBDPT (renderer.cpp:478-771), statement order kept.  Subpaths are std::vectors in the reference; here they are capped at
 * ORC_BDPT_MAX vertices each (survival beyond vertex k has probability 0.25 * 0.1^(k-2): 2.5e-7 at 8), the same cap the device uses. */
#define ORC_BDPT_MAX 8
/* debugging aid: with ORC_TRACE_PIXEL set, orc_render writes the vertices and terms of that pixel's NaiveBDPT samples to stderr */
static __thread int trace_on = 0;
#define TR(...) do { if (trace_on) fprintf(stderr, __VA_ARGS__); } while (0)
static double naive_bdpt(const orc_scene *s, v3 o, v3 d, const band *b, uint64_t *rng, orc_render_stats *st)
{
	const v3 cam_o = o, cam_d = d;
	/* ---- eye subpath, :482-529 */
	hitrec erec[ORC_BDPT_MAX];
	v3 eray_d[ORC_BDPT_MAX];
	double alphas[ORC_BDPT_MAX];
	size_t K = 0;
	double alpha = 1.0, alpha_ = 1.0;
	int last_eye_ray_hit = 0; /* what world->Hit(_ray, ...) at :558 returns: _ray is still the eye subpath's last ray there */
	v3 last_eye_vt = V(0, 0, 0);
	for (;;) {
		hitrec rec;
		st->closest_rays++;
		last_eye_ray_hit = world_hit(s, WORLD_MODE, o, d, 0.001, DBL_MAX, &rec);
		if (!last_eye_ray_hit) break;
		last_eye_vt = rec.vt;
		const orc_material *m = rec_mat(s, &rec);
		if (m->light_flag) break;
		alpha *= alpha_;
		onb uvw = onb_from_w(rec.normal);
		v3 gvi;
		double ratio, bsdf, pdf;
		if (!mat_sample(m, rec.normal, &uvw, onb_w2l(&uvw, vneg(d)), b->central_wl, rng, &gvi, &ratio, &bsdf, &pdf)) break;
		alpha_ = ratio * fabs(gvi.e[2]);
		if (K == ORC_BDPT_MAX) break; /* cap (not in the reference) */
		alphas[K] = alpha; erec[K] = rec; eray_d[K] = d;
		TR("eye %zu: p %.17g %.17g %.17g n %.6g %.6g %.6g poly %ld alpha %.9g alpha_ %.9g\n", K, rec.p.e[0], rec.p.e[1], rec.p.e[2], rec.normal.e[0], rec.normal.e[1], rec.normal.e[2], (long)rec.poly, alpha, alpha_);
		double prr = 0.5;
		K++;
		if (K > 2) prr = 0.9;
		if (orc_rng_f64(rng) < prr) break;
		alpha_ /= (1.0 - prr);
		o = rec.p;
		d = vunit(onb_l2w(&uvw, gvi));
	}
	/* ---- light subpath, :538-613 */
	uint64_t sel = rng_below(rng, (uint32_t)s->n_lights);
	uint64_t lf = s->lights[sel];
	v3 pol, light_normal;
	double light_area;
	random_point_on_polygon(s, lf, rng, &pol, &light_area, &light_normal);
	double light_pdf_area = 1.0 / light_area / (double)s->n_lights;
	onb lb = onb_from_w(light_normal); /* CosinePdf pdf(light_normal), pdf.cpp:70-84 */
	double c3[3];
	{
		double r1 = orc_rng_f64(rng), r2 = orc_rng_f64(rng);
		orc_cosine_hemisphere(r1, r2, c3);
	}
	v3 dir = onb_l2w(&lb, V(c3[0], c3[1], c3[2]));
	double cos_l = vdot(dir, lb.axis[2]);
	double light_dir_pdf_solidangle = cos_l >= 0.0 ? cos_l / M_PI : 0.0;
	double light_dir_pdf_projected = light_dir_pdf_solidangle / vdot(light_normal, dir);
	double light_value = 0.0;
	st->closest_rays++; /* :558 traces _ray (the eye subpath's last ray), not shadow_ray: same result as that ray's own trace */
	if (last_eye_ray_hit) light_value = mat_emitted(mat_at(s, s->object_id[lf], last_eye_vt), b); /* Emitted(shadow_ray, sampled_light_rec, sampled_light_rec.vt) */
	TR("pol %.17g %.17g %.17g dir %.17g %.17g %.17g\n", pol.e[0], pol.e[1], pol.e[2], dir.e[0], dir.e[1], dir.e[2]);
	TR("K %zu last_hit %d light_value %.9g pol %.9g %.9g %.9g pdf_area %.9g dir_pdf_proj %.9g\n", K, last_eye_ray_hit, light_value, pol.e[0], pol.e[1], pol.e[2], light_pdf_area, light_dir_pdf_projected);
	hitrec lrec[ORC_BDPT_MAX];
	v3 lray_d[ORC_BDPT_MAX];
	double betas[ORC_BDPT_MAX];
	size_t L = 0;
	double beta = 1.0, beta_ = 1.0;
	o = pol;
	d = dir;
	for (;;) {
		hitrec rec;
		st->closest_rays++;
		if (!world_hit(s, WORLD_MODE, o, d, 0.001, DBL_MAX, &rec)) break;
		const orc_material *m = rec_mat(s, &rec);
		if (m->light_flag) break;
		beta *= beta_;
		onb uvw = onb_from_w(rec.normal);
		v3 gvi;
		double ratio, bsdf, pdf;
		if (!mat_sample_t(m, rec.normal, &uvw, onb_w2l(&uvw, vneg(d)), b->central_wl, rng, &gvi, &ratio, &bsdf, &pdf, 1)) break;
		beta_ = ratio * fabs(gvi.e[2]);
		if (L == ORC_BDPT_MAX) break; /* cap (not in the reference) */
		betas[L] = beta; lrec[L] = rec; lray_d[L] = d;
		TR("light %zu: p %.17g %.17g %.17g n %.6g %.6g %.6g poly %ld beta %.9g beta_ %.9g\n", L, rec.p.e[0], rec.p.e[1], rec.p.e[2], rec.normal.e[0], rec.normal.e[1], rec.normal.e[2], (long)rec.poly, beta, beta_);
		double prr = 0.5;
		L++;
		if (L > 2) prr = 0.9;
		if (orc_rng_f64(rng) < prr) break;
		beta_ /= (1.0 - prr);
		o = rec.p;
		d = vunit(onb_l2w(&uvw, gvi));
	}
	/* ---- contributions by path length, :619-769 */
	double radiances[2 * ORC_BDPT_MAX + 1];
	size_t path_numbers[2 * ORC_BDPT_MAX + 1];
	const size_t n_len = K + (L + 1);
	for (size_t i = 0; i < n_len; i++) { radiances[i] = 0.0; path_numbers[i] = 0; }
	{ /* s = 0, t = 0, :632-642 */
		path_numbers[0]++;
		hitrec rec;
		st->closest_rays++;
		if (world_hit(s, WORLD_MODE, cam_o, cam_d, 0.001, DBL_MAX, &rec)) {
			const orc_material *m = rec_mat(s, &rec);
			if (m->light_flag) radiances[0] += mat_emitted(m, b) / (0 + 0 + 1);
		}
	}
	for (size_t e = 0; e < K; e++) { /* s >= 1, t = 0, :654-686 */
		size_t total_length = e + 2;
		path_numbers[total_length - 1]++;
		onb uvw = onb_from_w(erec[e].normal);
		v3 gvi;
		double ratio, bsdf, pdf;
		if (!mat_sample(rec_mat(s, &erec[e]), erec[e].normal, &uvw, onb_w2l(&uvw, vneg(eray_d[e])), b->central_wl, rng, &gvi, &ratio, &bsdf, &pdf))
			continue;
		hitrec light_rec;
		st->closest_rays++;
		if (!world_hit(s, WORLD_MODE, erec[e].p, onb_l2w(&uvw, gvi), 0.001, DBL_MAX, &light_rec)) continue;
		const orc_material *lm = rec_mat(s, &light_rec);
		if (!lm->light_flag) continue;
		const double light = mat_emitted(lm, b);
		radiances[total_length - 1] += alphas[e] * ratio * fabs(gvi.e[2]) * light;
		TR("t0 e %zu: %.9g\n", e, alphas[e] * ratio * fabs(gvi.e[2]) * light);
	}
	for (size_t e = 0; e < K; e++) { /* s >= 1, t = 1, :692-716 */
		size_t total_length = e + 2;
		path_numbers[total_length - 1]++;
		hitrec light_rec;
		st->closest_rays++;
		if (!world_hit(s, WORLD_MODE, erec[e].p, vunit(vsub(pol, erec[e].p)), 0.001, DBL_MAX, &light_rec)) continue;
		if ((uint64_t)light_rec.poly != lf) continue;
		onb uvw = onb_from_w(erec[e].normal);
		v3 vi = onb_w2l(&uvw, vunit(vsub(pol, erec[e].p)));
		v3 vo = onb_w2l(&uvw, vneg(eray_d[e]));
		double bsdf = mat_bsdf(rec_mat(s, &erec[e]), vi, b->central_wl, vo);
		radiances[total_length - 1] += alphas[e] * bsdf * geometry_term(&erec[e], &light_rec) * mat_emitted(rec_mat(s, &light_rec), b) / light_pdf_area;
		TR("t1 e %zu: bsdf %.9g G %.9g -> %.9g\n", e, bsdf, geometry_term(&erec[e], &light_rec), alphas[e] * bsdf * geometry_term(&erec[e], &light_rec) * mat_emitted(rec_mat(s, &light_rec), b) / light_pdf_area);
	}
	for (size_t e = 0; e < K; e++) { /* s >= 1, t > 1, :722-762 */
		for (size_t l = 0; l < L; l++) {
			size_t total_length = (e + 1) + (l + 2);
			path_numbers[total_length - 1]++;
			v3 to = vsub(lrec[l].p, erec[e].p);
			hitrec tmp;
			st->shadow_rays++;
			TR("conn e %zu l %zu: len %.17g\n", e, l, vlen(to));
			if (world_hit(s, WORLD_MODE, erec[e].p, vunit(to), 0.001, vlen(to) - 0.001, &tmp)) { TR("  occluded at t %.17g poly %ld\n", tmp.t, (long)tmp.poly); continue; } /* Occluded = closest hit, hittable.h:34-40 */
			double eye_bsdf, light_bsdf;
			{
				onb uvw = onb_from_w(erec[e].normal);
				v3 vi = onb_w2l(&uvw, vunit(to));
				v3 vo = onb_w2l(&uvw, vneg(eray_d[e]));
				eye_bsdf = mat_bsdf(rec_mat(s, &erec[e]), vi, b->central_wl, vo);
			}
			{
				onb uvw = onb_from_w(lrec[l].normal);
				v3 vi = onb_w2l(&uvw, vneg(lray_d[l]));
				v3 vo = onb_w2l(&uvw, vunit(vsub(erec[e].p, lrec[l].p)));
				light_bsdf = mat_bsdf(rec_mat(s, &lrec[l]), vi, b->central_wl, vo);
			}
			radiances[total_length - 1] += alphas[e] * betas[l] * eye_bsdf * light_bsdf * geometry_term(&erec[e], &lrec[l]) * light_value /
			                               (light_pdf_area * light_dir_pdf_projected);
			TR("  eye_bsdf %.9g light_bsdf %.9g G %.9g -> %.9g\n", eye_bsdf, light_bsdf, geometry_term(&erec[e], &lrec[l]), alphas[e] * betas[l] * eye_bsdf * light_bsdf * geometry_term(&erec[e], &lrec[l]) * light_value / (light_pdf_area * light_dir_pdf_projected));
		}
	}
	double radiance = 0.0;
	for (size_t i = 0; i < n_len; i++) {
		if (path_numbers[i] == 0) continue;
		radiance += radiances[i] / (double)path_numbers[i];
	}
	return radiance;
}

static double radiance_dispatch(const orc_scene *s, int algorithm, v3 o, v3 d, const band *b, uint64_t *rng, orc_render_stats *st)
{
	switch (algorithm) { /* renderer.cpp:96-112 */
	case ORC_ALG_NAIVE: return naive_path_tracing(s, o, d, b, rng, st);
	case ORC_ALG_NEE: return nee_path_tracing(s, o, d, b, rng, st);
	case ORC_ALG_NAIVE_BDPT: return naive_bdpt(s, o, d, b, rng, st);
	case ORC_ALG_MIS_BDPT: return 0.0; /* MISBDPT, renderer.cpp:775-778 */
	default: return mis_path_tracing(s, o, d, b, rng, st);
	}
}

double orc_radiance(const orc_scene *s, int algorithm, const double o[3], const double d[3], double min_wl,
                    double max_wl, uint64_t *rng_state, orc_render_stats *stats)
{
	orc_render_stats local = {0, 0, 0};
	band b = {min_wl, max_wl, (min_wl + max_wl) / 2.0};
	double r = radiance_dispatch(s, algorithm, vload(o), vload(d), &b, rng_state, stats ? stats : &local);
	return r;
}

void orc_render(const orc_scene *s, const orc_camera *cam, int nx, int ny, int ns, int first_sample, int n_samples,
                int spectral_samples, int algorithm, uint64_t seed, double *film, orc_render_stats *stats, int nthreads)
{
	uint64_t closest = 0, shadow = 0, samples = 0;
	set_threads(nthreads);
	const char *trace_env = getenv("ORC_TRACE_PIXEL"); /* film index i * ny + j whose NaiveBDPT samples are written to stderr */
	const long trace_pixel = trace_env ? atol(trace_env) : -1;
	for (int smp = first_sample; smp < first_sample + n_samples; smp++) { /* renderer.cpp:69 */
#pragma omp parallel for schedule(dynamic) reduction(+ : closest, shadow, samples)
		for (int i = 0; i < nx; i++) {
			for (int j = 0; j < ny; j++) {
				uint64_t rng = orc_rng_seed(seed, (uint64_t)i * (uint64_t)ny + (uint64_t)j, (uint64_t)smp);
				double u = (i + orc_rng_f64(&rng)) / nx; /* :81-82 */
				double v = (j + orc_rng_f64(&rng)) / ny;
				int k = (int)rng_below(&rng, (uint32_t)spectral_samples); /* :85 */
				band b;
				b.min_wl = 400.0 + 300.0 / spectral_samples * k;
				b.max_wl = b.min_wl + 300.0 / spectral_samples - 0.00001;
				b.central_wl = (b.min_wl + b.max_wl) / 2.0;
				double lens[2];
				do { /* camera.h:148-157 */
					lens[0] = 2 * orc_rng_f64(&rng) - 1.0;
					lens[1] = 2 * orc_rng_f64(&rng) - 1.0;
				} while (sqrt(lens[0] * lens[0] + lens[1] * lens[1] + 0.0 * 0.0) > 1.0);
				double o[3], d[3], p_image, p_lens, cos_theta;
				orc_camera_get_ray(cam, u, v, b.central_wl, lens, o, d, &p_image, &p_lens, &cos_theta);
				orc_render_stats st = {0, 0, 0};
				trace_on = trace_pixel == (long)i * ny + j;
				double rad = radiance_dispatch(s, algorithm, vload(o), vload(d), &b, &rng, &st);
				double geometry_factor = pow(cos_theta, 4.0) / (cam->d * cam->d); /* :113-114 */
				TR("radiance %.9g\n", rad);
				rad *= cam->film_sensitivity * geometry_factor / (p_image * p_lens);
				TR("film value %.9g\n", rad);
				if (!isnan(rad))
					orc_spectrum_add(film + ((size_t)i * ny + j) * ORC_NBINS, rad / ns, b.min_wl, b.max_wl);
				closest += st.closest_rays;
				shadow += st.shadow_rays;
				samples++;
			}
		}
	}
	if (stats) {
		stats->closest_rays += closest;
		stats->shadow_rays += shadow;
		stats->samples += samples;
	}
}
