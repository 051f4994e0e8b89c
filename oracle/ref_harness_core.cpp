/*
 * ref_harness_core.cpp -- extern "C" wrappers around reference translation units that compile
 * unmodified in the build container: aabb.cpp, onb.cpp, spectrum.cpp and the header-only camera.h, vec3.h, ray.h.
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/rt_oracle.c and to generate tests/golden vectors.
 * It is built only where the reference checkout exists; nothing here is reference code.
 */
#include <cstdlib>
#include <cstring>
#include "aabb.h"
#include "onb.h"
#include "spectrum.h"
#include "camera.h"

static dvec3 L(const double *p) { return dvec3(p[0], p[1], p[2]); }
static void S(double *p, const dvec3 &v) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; }

extern "C" {

/* vec3.h: dot, cross, length, squared_length, unit_vector, the component-wise and scalar operators, rodrigues (axis = unit(b)).
 * out: dot, length(a), squared_length(a), then 3 each of cross, unit(a), a*b, a/b, t*a, a/t, a+b, a-b, -a, rodrigues = 33 doubles */
void ref_vec3_ops(const double *a_, const double *b_, double t, double *out)
{
	const dvec3 a = L(a_), b = L(b_);
	out[0] = dot(a, b);
	out[1] = a.length();
	out[2] = a.squared_length();
	S(out + 3, cross(a, b));
	S(out + 6, unit_vector(a));
	S(out + 9, a * b);
	S(out + 12, a / b);
	S(out + 15, t * a);
	S(out + 18, a / t);
	S(out + 21, a + b);
	S(out + 24, a - b);
	S(out + 27, -a);
	S(out + 30, rodrigues(a, unit_vector(b), t));
}
/* ray.h: ray(a, b).point_at_parameter(t) */
void ref_ray_point(const double *o, const double *d, double t, double *out) { S(out, ray(L(o), L(d)).point_at_parameter(t)); }

int ref_aabb_hit(const double *minp, const double *maxp, const double *o, const double *d, double t_min, double t_max)
{
	AABB b(L(minp), L(maxp));
	ray r(L(o), L(d));
	return b.Hit(r, t_min, t_max) ? 1 : 0;
}
double ref_aabb_surface_area(const double *minp, const double *maxp) { return AABB(L(minp), L(maxp)).SurfaceArea(); }
void ref_surrounding_box(const double *min0, const double *max0, const double *min1, const double *max1, double *omin, double *omax)
{
	AABB u = SurroundingBox(AABB(L(min0), L(max0)), AABB(L(min1), L(max1)));
	S(omin, u.minp);
	S(omax, u.maxp);
}
void ref_onb_from_w(const double *n, double *axis)
{
	ONB b;
	b.BuildFromW(L(n));
	for (int i = 0; i < 3; i++) S(axis + 3 * i, b.axis[i]);
}
void ref_onb_local_to_world(const double *axis, const double *a, double *out)
{
	ONB b;
	for (int i = 0; i < 3; i++) b.axis[i] = L(axis + 3 * i);
	S(out, b.LocalToWorld(L(a)));
}
void ref_onb_world_to_local(const double *axis, const double *a, double *out)
{
	ONB b;
	for (int i = 0; i < 3; i++) b.axis[i] = L(axis + 3 * i);
	S(out, b.WorldToLocal(L(a)));
}
double ref_spectrum_get(const double *data, double wl) { return Spectrum(data).get(wl); }
double ref_spectrum_integrate(const double *data, double a, double b) { return Spectrum(data).integrate(a, b); }
void ref_spectrum_add(double *data, double value, double a, double b)
{
	Spectrum s(data);
	s.add(value, a, b);
	for (int i = 0; i < N_SAMPLE; i++) data[i] = s.data[i];
}
void ref_xyz(const double *data, double *out) { S(out, xyz(Spectrum(data))); }
void ref_rgb(const double *data, double *out) { S(out, rgb(Spectrum(data))); }
void ref_rgb_to_spectrum(const double *c, double *out)
{
	Spectrum s = RGBtoSpectrum(L(c));
	for (int i = 0; i < N_SAMPLE; i++) out[i] = s.data[i];
}

/* LensCamera: SetLens + set_Camera + get_ray with drand48 seeded by the caller's seed.
 * n30: lens refractive index spectrum; out: o[3] d[3] p_image p_lens cos_theta d a_over = 11 doubles */
void ref_camera_get_ray(double film_width, double film_height, double aperture, const double *n30, double vfov,
                        double focal_length, const double *from, const double *at, const double *up, double u,
                        double v, double wavelength, long seed, double *out)
{
	LensCamera cam;
	cam.film_width = film_width;
	cam.film_height = film_height;
	cam.aperture = aperture;
	cam.n = Spectrum(n30);
	cam.SetLens(vfov, focal_length);
	cam.set_Camera(L(from), L(at), L(up));
	srand48(seed);
	double p_image, p_lens, cos_theta;
	ray r = cam.get_ray(u, v, wavelength, p_image, p_lens, cos_theta);
	S(out, r.A);
	S(out + 3, r.B);
	out[6] = p_image;
	out[7] = p_lens;
	out[8] = cos_theta;
	out[9] = cam.d;
	out[10] = cam.LensMakersFormula(wavelength);
}
}
