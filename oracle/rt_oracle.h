/*
 * rt_oracle.h -- CPU restatement (fp64, plain C) of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ray_tracer_amd/, include/) may include,
 * link, load or call this code.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to the reference checkout) whose
 * arithmetic it follows, operation for operation, in IEEE double without FMA contraction
 * (build with -ffp-contract=off; the reference Makefile builds with plain -O3 on x86-64).
 *
 * PARITY PIN STATUS
 *   pinned against reference translation units compiled unmodified in the build container
 *   (oracle/ref_harness.cpp -> oracle/_ref/):
 *       AABB::Hit / SurfaceArea / SurroundingBox (aabb.cpp), ONB (onb.cpp),
 *       Spectrum::{get,integrate,add}, xyz, rgb, RGBtoSpectrum (spectrum.cpp),
 *       LensCamera::{SetLens,set_Camera,get_ray} (camera.h),
 *       BVHNode::Hit / SAHBVHNode build / HittableList::Hit (bvh.cpp, hittablelist.cpp).
 *   PARITY UNPINNED (restated from source text; the translation units need Boost/ImGui, which the
 *   build container lacks, and the reference's own tests hold no vectors for them):
 *       ConvexPolygon::{Hit,HitTriangle,BoundingBox,GetRandomPointOnPolygon} (object.cpp),
 *       the material nodes (materialnode.cpp), rfresnel/cfresnel (material.cpp),
 *       CosinePdf/PowerHeuristic (pdf.cpp) and the integrators (renderer.cpp).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NBINS 30 /* spectrum.h:8  N_SAMPLE */

/* ---------------- small value-type functions ---------------- */
int    orc_aabb_hit(const double minp[3], const double maxp[3], const double o[3], const double d[3],
                    double t_min, double t_max);                                 /* aabb.cpp:17-34 */
double orc_aabb_surface_area(const double minp[3], const double maxp[3]);       /* aabb.cpp:37-41 */
void   orc_surrounding_box(const double min0[3], const double max0[3], const double min1[3],
                           const double max1[3], double omin[3], double omax[3]); /* aabb.cpp:45-54 */
void   orc_onb_from_w(const double n[3], double axis[9]);                        /* onb.cpp:31-42 */
void   orc_onb_local_to_world(const double axis[9], const double a[3], double out[3]); /* onb.cpp:21-24 */
void   orc_onb_world_to_local(const double axis[9], const double a[3], double out[3]); /* onb.cpp:26-29 */

double orc_spectrum_get(const double data[ORC_NBINS], double wavelength);        /* spectrum.cpp:158-162 */
double orc_spectrum_integrate(const double data[ORC_NBINS], double min_wl, double max_wl); /* :126-146 */
void   orc_spectrum_add(double data[ORC_NBINS], double value, double min_wl, double max_wl); /* :148-155 */
void   orc_xyz(const double data[ORC_NBINS], double out[3]);                     /* spectrum.cpp:164-171 */
void   orc_rgb(const double data[ORC_NBINS], double out[3]);                     /* spectrum.cpp:184-199 */
void   orc_rgb_to_spectrum(const double rgb[3], double out[ORC_NBINS]);          /* spectrum.cpp:219-232 */
void   orc_tonemap(const double data[ORC_NBINS], double scale, double out[3]);   /* renderer.cpp:131-139 */

double orc_rfresnel(double cos_theta, double n);                                 /* material.cpp:14-25 */
double orc_cfresnel(double cos_theta, double n_re, double n_im);                 /* material.cpp:26-34 */
double orc_power_heuristic(double pdf1, double pdf2, double beta);               /* pdf.cpp:139-146 */
void   orc_cosine_hemisphere(double r1, double r2, double out[3]);               /* pdf.cpp:36-46 */
double orc_cosine_pdf(const double *direction, const double *w);                   /* pdf.cpp:76-84 */
uint64_t orc_mixture_pick(double r, uint64_t l);                                   /* MixturePdf::Generate, pdf.cpp:116-126 */
double orc_mixture_pdf(const double *component_pdf, uint64_t l);                   /* MixturePdf::PdfVal, pdf.cpp:128-135 */
void   orc_vec3_ops(const double *a, const double *b, double t, double *out33);    /* vec3.h operators, dot, cross, unit_vector, rodrigues */
void   orc_ray_point(const double *o, const double *d, double t, double *out);     /* ray.h:25-27 */

/* ---------------- camera (camera.h:101-209) ---------------- */
typedef struct orc_camera {
	double film_sensitivity, aperture, film_width, film_height; /* camera.h:189-192 */
	double n[ORC_NBINS];                                          /* camera.h:193 */
	double d;                                                     /* camera.h:194 */
	double lens_maker_coefficient;
	double origin[3], film_lower_left_corner[3], horizontal[3], vertical[3], w[3], u[3], v[3];
} orc_camera;
void orc_camera_default(orc_camera *c);
void orc_camera_set_lens(orc_camera *c, double vfov, double focal_length);       /* camera.h:108-125 */
void orc_camera_set_camera(orc_camera *c, const double from[3], const double at[3], const double up[3]); /* :128-145 */
/* lens_xy = the accepted rejection-sampled point of get_random_on_lens before the aperture/2 scale */
void orc_camera_get_ray(const orc_camera *c, double u, double v, double wavelength, const double lens_xy[2],
                        double o[3], double d[3], double *p_image, double *p_lens, double *cos_theta); /* :159-187 */

/* ---------------- counter-based random stream shared with the device code ---------------- */
uint64_t orc_rng_seed(uint64_t seed, uint64_t pixel, uint64_t sample);
uint64_t orc_rng_next(uint64_t *state);
double   orc_rng_f64(uint64_t *state); /* next * 2^-32: 32-bit draws (PCG-XSH-RR 64/32) */

/* ---------------- scene ---------------- */
typedef struct orc_scene orc_scene;

/* The mesh is the reference's `obj` container flattened (obj.h:10-31): v/vt/vn tables, per-face
 * corner index triples (0-based, -1 = absent vt), faces grouped per object. */
orc_scene *orc_scene_create(const double *v, uint64_t n_v, const double *vt, uint64_t n_vt,
                            const double *vn, uint64_t n_vn, const int64_t *corners,
                            const uint64_t *face_begin, uint64_t n_faces,
                            const uint64_t *object_begin, uint32_t n_objects);   /* object.cpp:94-140 */
void     orc_scene_free(orc_scene *s);
uint64_t orc_scene_n_polygons(const orc_scene *s);
uint64_t orc_scene_n_triangles(const orc_scene *s); /* fan triangles */
/* copies fan-triangle geometry: v[9*n], face normal[3*n], polygon id[n] */
void     orc_scene_get_triangles(const orc_scene *s, double *v9, double *nf3, uint32_t *poly);
void     orc_scene_polygon_bbox(const orc_scene *s, uint64_t poly, double minp[3], double maxp[3]); /* object.cpp:309-328 */
/* ConvexPolygon::Hit for one polygon (object.cpp:286-305); returns 1 and fills t, p, normal, fan triangle id */
int      orc_polygon_hit(const orc_scene *s, uint64_t poly, const double o[3], const double d[3], double t_min,
                         double t_max, double *t, double p[3], double normal[3], int64_t *tri);

/* two-level sweep-SAH BVH: one per object, then one over the objects (object.cpp:134-139, bvh.cpp:134-232) */
int      orc_scene_build_bvh(orc_scene *s);
uint64_t orc_scene_bvh_nodes(const orc_scene *s);
uint32_t orc_scene_bvh_depth(const orc_scene *s);

enum { ORC_MODE_BVH_REFERENCE = 0, /* bvh.cpp:76-102: both children, same t_max, tie -> right */
       ORC_MODE_LIST = 1,          /* hittablelist.cpp:3-17: linear, shrinking t_max, tie -> later */
       ORC_MODE_LIST_LOWEST = 2    /* linear, tie -> lowest fan-triangle index (the device tie rule) */ };

/* rays: 8 floats each {ox,oy,oz,t_min, dx,dy,dz,t_max}, widened to double before any arithmetic.
 * out_poly/out_tri: -1 on miss.  nthreads<=0: all cores. Returns number of hits. */
uint64_t orc_trace_closest(const orc_scene *s, const float *rays, uint64_t n, int mode,
                           int32_t *out_poly, int32_t *out_tri, double *out_t, int nthreads);
uint64_t orc_trace_any(const orc_scene *s, const float *rays, uint64_t n, int mode,
                       uint8_t *out_hit, int nthreads);                          /* hittable.h:34-40 */

/* Ordered, t-pruned traversal of an externally supplied flat BVH2 (the product's host-built tree):
 * node i = {float lo0[3],hi0[3],lo1[3],hi1[3]; int32 child0, child1}. child >= 0: interior node;
 * child < 0: leaf, first = (~child) & 0x0FFFFFFF... see rt_oracle.c. Uses the reference triangle
 * test on this scene's fan triangles through tri_index (leaf order -> fan triangle id).
 * Accumulates interior-node visits and triangle tests (the figures the roofline's algorithmic
 * bytes per ray are built from). */
uint64_t orc_trace_bvh2(const orc_scene *s, const void *nodes, uint64_t n_nodes, const uint32_t *tri_index,
                        const float *rays, uint64_t n, int32_t *out_poly, int32_t *out_tri, double *out_t,
                        uint64_t *node_visits, uint64_t *tri_tests, int nthreads);

/* The same walk over the product's 8-wide tree (128-byte nodes: rt_wide_node of include/mi355rt.h; from rt_get_wide_bvh):
 * counts the nodes fetched and the triangle records tested per ray, which is what the traversal kernel's traffic is priced with. */
uint64_t orc_trace_wide(const orc_scene *s, const void *nodes, uint64_t n_nodes, const uint32_t *tri_index,
                        const float *rays, uint64_t n, int32_t *out_poly, int32_t *out_tri, double *out_t,
                        uint64_t *node_visits, uint64_t *tri_tests, int nthreads);

/* 0 (default): ConvexPolygon::HitTriangle as the reference writes it; 1: the optional watertight test of the library's
 * RT_TRIANGLE_WATERTIGHT mode (no reference counterpart), used by every traversal and integrator of this file from then on */
void orc_scene_set_triangle_test(orc_scene *s, int watertight);

/* ---------------- materials (flat table, one record per object) ---------------- */
enum { ORC_MAT_NONE = 0, ORC_MAT_LAMBERTIAN = 1, ORC_MAT_DIFFUSE_LIGHT = 2, ORC_MAT_DIELECTRIC = 3,
       ORC_MAT_CONDUCTOR = 4, ORC_MAT_COLORED_METAL = 5, ORC_MAT_GGX = 6 };
typedef struct orc_material {
	int32_t type;
	int32_t light_flag;                 /* NodeMaterial::light_flag, materialnode.cpp:1613 */
	double a[ORC_NBINS];                /* albedo | color | n */
	double b[ORC_NBINS];                /* k | surface_color */
	double roughness;                   /* ORC_MAT_GGX: alpha (the reference never initialises its alpha, materialnode.cpp:651) */
} orc_material;
/* returns 0 when no light polygons exist (renderer.cpp:41-42) */
/* BiliteralFilter::FilterImage (filter.cpp:13-51); pinned by the compiled reference (oracle/_ref/libref_filter.so) */
void orc_bilateral(const double *img, int img_width, int img_height, int window, double sigma_d, double sigma_r, double *result);
int orc_scene_set_materials(orc_scene *s, const orc_material *per_object, uint32_t n_objects); /* renderer.cpp:24-44 */
/* per-hit material graphs (the Compute() chains of materialnode.cpp that read the hit's texture coordinate), as postfix programs for
 * slot a (albedo / color) of some objects; see graph_eval in rt_oracle.c for the ops; parity unpinned */
int orc_scene_set_material_graphs(orc_scene *s, const uint32_t *programs, uint32_t n_programs, const uint32_t *code, uint32_t n_code,
                                  const double *spectra, uint32_t n_spectra, const double *vectors, uint32_t n_vectors,
                                  const double *scalars, uint32_t n_scalars, const uint8_t *const *tex, const int *tex_w,
                                  const int *tex_h, const int *tex_bpp, uint32_t n_textures);
/* slot a of `object` evaluated at texture coordinate vt (returns 0 when the object has no program) */
int orc_graph_eval(const orc_scene *s, uint32_t object, const double vt[3], double out[ORC_NBINS]);
/* environment texture of NaivePathTracing (renderer.cpp:175-193; globals of scene.h:20-22); parity unpinned */
int orc_scene_set_environment(orc_scene *s, const uint8_t *tex, int w, int h, int bpp, double brightness);
int orc_sphere_hit(const double o[3], const double d[3], double t_min, double t_max, double *t); /* object.cpp:43-75, dome of renderer.cpp:169-171 */
double orc_environment_lookup(const uint8_t *tex, int w, int h, int bpp, double brightness, const double o[3], const double d[3],
                              double t, double central_wl); /* renderer.cpp:180-191 */
uint64_t orc_scene_n_lights(const orc_scene *s);

enum { ORC_ALG_NAIVE = 0, ORC_ALG_NEE = 1, ORC_ALG_MIS = 2, ORC_ALG_NAIVE_BDPT = 3, ORC_ALG_MIS_BDPT = 4 }; /* renderer.h:12-18 */

typedef struct orc_render_stats {
	uint64_t closest_rays, shadow_rays, samples;
} orc_render_stats;

/* Renderer::RenderImage restated (renderer.cpp:54-160) for sample indices [first_sample,
 * first_sample+n_samples) of an ns-sample render.  film: nx*ny*30 doubles, index (i*ny+j)*30+bin,
 * accumulated into (not cleared). */
void orc_render(const orc_scene *s, const orc_camera *cam, int nx, int ny, int ns, int first_sample,
                int n_samples, int spectral_samples, int algorithm, uint64_t seed, double *film,
                orc_render_stats *stats, int nthreads);

/* one path, for unit tests: returns the radiance of the integrator for the given primary ray */
double orc_radiance(const orc_scene *s, int algorithm, const double o[3], const double d[3],
                    double min_wl, double max_wl, uint64_t *rng_state, orc_render_stats *stats);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
