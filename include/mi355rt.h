/*
 * mi355rt.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The reference renderer has no FFI layer: its hot path is reached through the C++ class
 * `Renderer` (reference renderer.h:21-50).  Each entry point below replaces the body of one
 * `Renderer` member (or one call it makes into hittable/bvh/object code); the reference line it
 * stands in for is cited on every declaration.  All arguments are plain pointers and sizes,
 * caller-owned.  Every function returns RT_OK (0) or a negative rt_status; no exception crosses
 * the boundary.  A context may be used from one host thread at a time (any thread); each call
 * makes the context's device current.  The library never falls back to the CPU: without a usable
 * HIP device rt_create fails.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_NBINS 30 /* reference spectrum.h:8, N_SAMPLE: 30 bins of 10 nm from 400 nm */

typedef enum rt_status {
	RT_OK = 0,
	RT_ERR_INVALID = -1,    /* bad argument / call order */
	RT_ERR_HIP = -2,        /* a HIP runtime call failed; text in rt_last_error */
	RT_ERR_NO_DEVICE = -3,  /* no HIP device: there is no CPU fallback */
	RT_ERR_NO_LIGHTS = -4,  /* Renderer::LoadMaterials returns false when no object is a light (renderer.cpp:41-42) */
	RT_ERR_NO_MATERIAL = -5,/* Renderer::LoadMaterials returns false when an object has no material (renderer.cpp:29-32) */
	RT_ERR_MESH = -6,       /* malformed mesh (index out of range, face with < 3 corners, missing vn) */
	RT_ERR_HOST = -7        /* the host side ran out of memory or could not start a thread; text in rt_last_error */
} rt_status;

typedef struct rt_context rt_context;

/* One ray, 32 bytes.  Origin and direction are widened to double on the device before the
 * reference's fp64 triangle arithmetic, so a float ray means the same ray on both sides.
 * tmax = +inf stands for the reference's numeric_limits<double>::max() (renderer.cpp:173). */
typedef struct rt_ray {
	float ox, oy, oz, tmin;
	float dx, dy, dz, tmax;
} rt_ray;

#define RT_MISS 0xFFFFFFFFu

/* Closest-hit result, 16 bytes.  prim = global face (polygon) index in obj order, i.e. the
 * reference's rec.hit_object mapped through ObjModel::polygon_models[i][j] (object.cpp:131-133)
 * as object_begin[i]+j; tri = fan-triangle index (triangle k of face f is tri_begin[f]+k);
 * t = the reference's rec.t rounded once from double to float. */
typedef struct rt_hit {
	float t;
	uint32_t prim;
	uint32_t tri;
	uint32_t object;
} rt_hit;

/* The reference `obj` container flattened (obj.h:10-31): what Renderer::Load hands to
 * ObjModel::ObjModel (renderer.cpp:18-22, object.cpp:94-140).  Indices are 0-based; a corner's vt
 * index is -1 when the face has no texture coordinate (obj.cpp:116-119); every corner needs a vn
 * index (object.cpp:115 dereferences it). */
typedef struct rt_mesh {
	const double *v;  uint64_t n_v;   /* xyz triples, obj::v */
	const double *vt; uint64_t n_vt;  /* uv0 triples, obj::vt (third component 0, obj.cpp:92) */
	const double *vn; uint64_t n_vn;  /* xyz triples, obj::vn */
	const int64_t *corners;           /* per corner: v, vt, vn index */
	const uint64_t *face_begin;       /* n_faces+1 offsets into corners */
	uint64_t n_faces;
	const uint64_t *object_begin;     /* n_objects+1 offsets into faces, objobject::faces */
	uint32_t n_objects;
} rt_mesh;

/* BSDF node wired to a NodeMaterial's Output (materialnode.cpp:1758-1795), constant-folded to
 * 30-bin spectra by the host.  a/b: Lambertian a=albedo; DiffuseLight a=color; Dielectric a=n,
 * b=surface_color; Conductor a=n, b=k; ColoredMetal a=albedo; GGX a=n, b=k (carried, not used: the node sets fresnel = 1). */
typedef enum rt_material_type {
	RT_MAT_NONE = 0,          /* nothing wired to "BSDF": SampleBSDF false, BSDF/PDF/Emitted 0 */
	RT_MAT_LAMBERTIAN = 1,    /* materialnode.cpp:262-295 */
	RT_MAT_DIFFUSE_LIGHT = 2, /* materialnode.cpp:739-749 */
	RT_MAT_DIELECTRIC = 3,    /* materialnode.cpp:354-419 */
	RT_MAT_CONDUCTOR = 4,     /* materialnode.cpp:473-514 */
	RT_MAT_COLORED_METAL = 5, /* materialnode.cpp:564-596 */
	RT_MAT_GGX = 6            /* GGXReflectionNode (materialnode.cpp:647-695) made well defined, a deviation: the reference reads an
	                           * uninitialised alpha and returns nothing from BSDF() / PDF().  SampleBSDF as written there, in the
	                           * shading frame, with alpha = `roughness`; BSDF() and PDF() are the same expressions for the half vector */
} rt_material_type;

typedef struct rt_material {
	int32_t type;       /* rt_material_type */
	int32_t light_flag; /* NodeMaterial::light_flag (materialnode.cpp:1613) */
	double a[RT_NBINS];
	double b[RT_NBINS];
	double roughness;   /* RT_MAT_GGX: alpha, the node's "roughness" (0.5 when the file has none); ignored by the other types */
} rt_material;

/* LensCamera state after SetLens + set_Camera (camera.h:101-209), copied field by field. */
typedef struct rt_camera {
	double film_sensitivity, aperture, film_width, film_height; /* camera.h:189-192 */
	double n[RT_NBINS];                                          /* camera.h:193 lens refractive index */
	double d;                                                    /* camera.h:117 sensor-lens distance */
	double lens_maker_coefficient;                               /* camera.h:122 */
	double origin[3], film_lower_left_corner[3], horizontal[3], vertical[3], w[3]; /* camera.h:137-144 */
} rt_camera;

typedef enum rt_algorithm { /* enum RenderingAlgorithm, renderer.h:12-18 */
	RT_ALG_NAIVE = 0,  /* Renderer::NaivePathTracing, renderer.cpp:163-224 */
	RT_ALG_NEE = 1,    /* Renderer::NEEPathTracingWithoutSpecular, renderer.cpp:227-325 */
	RT_ALG_MIS = 2,    /* Renderer::NEEMISPathTracing, renderer.cpp:329-475 */
	RT_ALG_NAIVE_BDPT = 3, /* Renderer::NaiveBDPT, renderer.cpp:478-771 (the default of renderer.h:33) */
	RT_ALG_MIS_BDPT = 4    /* Renderer::MISBDPT, renderer.cpp:775-778: returns 0, the film stays empty */
} rt_algorithm;

typedef struct rt_stats {
	uint64_t n_faces, n_triangles, n_objects, n_lights;
	uint64_t bvh_nodes, bvh_depth, bvh_leaves;
	double   bvh_build_seconds;
	uint64_t closest_rays, any_rays, camera_samples; /* traced by render passes since rt_render_begin */
	uint64_t device_bytes;                           /* geometry + BVH resident in HBM */
} rt_stats;

/* flat BVH2 node as built on the host, 64 bytes: two child boxes (float, rounded outward) and two
 * child links.  link >= 0: interior node index; link < 0: leaf, c = ~link, first = c & 0x07FFFFFF
 * (index into the leaf-ordered triangle arrays), count = (c >> 27) + 1. */
typedef struct rt_bvh2_node {
	float lo0[3], hi0[3], lo1[3], hi1[3];
	int32_t child0, child1;
	uint32_t pad[2];
} rt_bvh2_node;

/* ---- lifetime ---- */
/* replaces: Renderer::Renderer (renderer.cpp:12-16).  device = HIP device ordinal. */
int rt_create(int device, rt_context **out);
void rt_destroy(rt_context *ctx);
/* text of the last failure on this context (or of the last failed rt_create when ctx is NULL) */
const char *rt_last_error(const rt_context *ctx);

/* ---- scene ---- */
/* replaces: ObjModel::ObjModel inside Renderer::Load (renderer.cpp:21, object.cpp:94-140) and the
 * SAHBVHNode construction it triggers (bvh.cpp:134-232).  Builds per-polygon data exactly as the
 * reference does (unit vertex normals, face normal with the flip rule, fan-triangle areas), builds
 * a BVH on the host, flattens it and uploads nodes + triangles to HBM. */
int rt_upload_mesh(rt_context *ctx, const rt_mesh *mesh);
/* which builder the next rt_upload_mesh uses for the tree that replaces SAHBVHNode::SAHBVHNode (bvh.cpp:134-232,
 * 68 s per million triangles in the reference).  RT_BUILDER_HOST_SAH: binned SAH on one host thread.
 * RT_BUILDER_DEVICE_LBVH: Morton-ordered radix tree built on the GPU (870 k triangles: 4 ms against 350 ms, 10 M: 12 ms against
 * 5 s; traversal within 2 % of the host tree's on meshes, 4.5 % slower on a triangle soup).  RT_BUILDER_AUTO (default):
 * the device builder above 2^15 fan triangles, else the host builder.  Hits do not depend on the choice (ties in t aside).  The
 * environment variable MI355RT_BUILDER=host|device overrides it. */
typedef enum rt_builder { RT_BUILDER_HOST_SAH = 0, RT_BUILDER_DEVICE_LBVH = 1, RT_BUILDER_AUTO = 2 } rt_builder;
int rt_set_builder(rt_context *ctx, int builder);
/* No counterpart in the reference (its tree, bvh.cpp:134-232, stores every polygon once under the box of the whole polygon).
 * Before either builder runs, the next rt_upload_mesh cuts the references of triangles whose boxes waste much space -- long thin
 * triangles lying diagonally: braces, cables, roof edges -- into several references with the boxes of the pieces (Karras and Aila,
 * HPG 2013, section 4).  `budget` is the number of additional references the whole scene may get, as a fraction of its fan
 * triangles (default 0.25; 0 switches the cutting off; scenes below 4096 triangles are never cut).  Hits, ids and t do not depend
 * on it.  rt_get_reference_cutting reports what the last upload did: triangles that were cut, references added. */
int rt_set_reference_cutting(rt_context *ctx, double budget);
int rt_get_reference_cutting(rt_context *ctx, uint64_t *cut_triangles, uint64_t *extra_references);
/* Which ray-triangle test the traversal kernels (and the direct light-polygon test of the NEE integrator) run; takes effect at once.
 * RT_TRIANGLE_REFERENCE (default): ConvexPolygon::HitTriangle as the reference writes it (object.cpp:238-253) -- distance to the
 *   plane of the polygon's first three vertices, three cross products, strict `> 0`: ids and t identical to the reference's
 *   arithmetic, and like it not watertight (a ray through a shared edge or vertex can miss both neighbours).
 * RT_TRIANGLE_WATERTIGHT: no counterpart in the reference.  Woop-Benthin-Wald edge functions in fp64 on every fan triangle's own
 *   plane: no ray slips between triangles that share an edge, edges and vertices count as hits, and a hit always lies inside its
 *   triangle's box, so results do not depend on the tree even for polygons that are not planar. */
typedef enum rt_triangle_test { RT_TRIANGLE_REFERENCE = 0, RT_TRIANGLE_WATERTIGHT = 1 } rt_triangle_test;
int rt_set_triangle_test(rt_context *ctx, int mode);
/* Per-hit material graphs.  replaces: the Compute() chains that NodeMaterial::{SampleBSDF,BSDF,PDF,Emitted} run with
 * Argument{rec.vt} on every call (materialnode.cpp:1758-1795) when the albedo / color slot of a Lambertian, ColoredMetal or
 * DiffuseLight node is fed by nodes that read the hit's texture coordinate.  The host compiles such a graph to a postfix
 * program of (op, arg) pairs; values are vectors, spectra or scalars, at most six deep:
 *   0 SPEC k     push spectra[k]                      (SpectrumNode and folded constant sub-graphs)
 *   1 VEC k      push vectors[k]                      (RGBColorNode, :931-936)
 *   2 UV         push rec.vt                          (UVNode, :893-896)
 *   3 ZERO_S, 4 ZERO_V  push zeros                    (an unconnected input)
 *   5 IMAGE k    uv -> rgb of textures[k]             (ImageTextureNode, :1001-1018)
 *   6 CHECKER k  uv -> (1,1,1) or (0,0,0), size scalars[k]  (CheckerboardNode, :1046-1063)
 *   7 RGB2SPEC   rgb -> spectrum                      (RGBtoSpectrumNode, :949-957)
 *   8 ADD_S, 9 MUL_S, 10 ADD_V, 11 MUL_V              (AdditionNode :1083-1109, MultiplicationNode :1179-1203; 8 and 9 also on two scalars)
 *   12 ACCESS i  vector -> its component i, a scalar  (AccessVec3ComponentNode, :1293-1301)
 *   13 COMBINE   three scalars -> vector              (CombineVec3ComponentNode, :1333-1347)
 *   14 NOISE1 k  scalar -> scalar, scalars[k] = size followed by the size + 1 table values  (ValueNoiseNode, :1366-1384)
 *   15 NOISE2 k  u, v -> scalar, scalars[k] = u_size, v_size followed by (u_size + 1)(v_size + 1) values, u outermost
 *                                                     (ValueNoise2DNode, :1447-1478; the tables are drand48 after srand48(323048))
 *   16 ROTATE    v, n, theta -> v rotated about n     (RodriguesRotationNode, :1536-1573)
 *   17 SCALAR k  push scalars[k]
 * A program must leave exactly one spectrum.  Call after rt_upload_materials (which drops earlier programs); the record's
 * `a` is ignored for objects that have a program.  graphs == NULL or n_programs == 0 removes all programs. */
typedef struct rt_texture { const uint8_t *texels; int32_t width, height, bpp; } rt_texture; /* height rows of width texels, bpp >= 3 bytes each */
typedef struct rt_graph_program { uint32_t object, code_begin, code_len; } rt_graph_program; /* in pairs */
typedef struct rt_material_graphs {
	const rt_graph_program *programs; uint32_t n_programs;
	const uint32_t *code; uint32_t n_code;        /* n_code (op, arg) pairs */
	const double *spectra; uint32_t n_spectra;    /* [n][RT_NBINS] */
	const double *vectors; uint32_t n_vectors;    /* [n][3] */
	const double *scalars; uint32_t n_scalars;
	const rt_texture *textures; uint32_t n_textures;
} rt_material_graphs;
int rt_upload_material_graphs(rt_context *ctx, const rt_material_graphs *graphs);
/* replaces: the globals env_mapping_texture / env_mapping_width / env_mapping_height / env_mapping_bpp / env_brightness
 * (scene.h:20-22, filled by stbi_load at scene.cpp:245) that NaivePathTracing reads when a ray leaves the scene
 * (renderer.cpp:175-193).  texels: height rows of width texels, bpp (3 or 4) bytes each, copied; NULL removes the
 * environment.  Only the naive integrator uses it, as in the reference. */
int rt_set_environment(rt_context *ctx, const uint8_t *texels, int width, int height, int bpp, double brightness);
/* replaces: Renderer::LoadMaterials (renderer.cpp:24-44): one record per obj object, in object
 * order; collects the light polygons; RT_ERR_NO_LIGHTS when there is none. */
int rt_upload_materials(rt_context *ctx, const rt_material *per_object, uint32_t n_objects);
/* replaces: reads of Renderer::cam inside RenderImage (renderer.cpp:91,113-114) */
int rt_set_camera(rt_context *ctx, const rt_camera *cam);
/* replaces: Renderer::Clear (renderer.cpp:47-51) plus dropping the uploaded scene */
int rt_clear(rt_context *ctx);

/* ---- closest hit / any hit ---- */
/* replaces: world->Hit(r, t_min, t_max, rec) = ObjModel::Hit -> BVHNode::Hit (object.cpp:142-145,
 * bvh.cpp:76-102) -> AABB::Hit (aabb.cpp:17-34) + ConvexPolygon::Hit (object.cpp:286-305).
 * Host buffers; t64 (optional, may be NULL) receives rec.t in double, +inf on a miss.
 * Ties in t between different polygons resolve to the lowest fan-triangle index. */
int rt_trace_closest(rt_context *ctx, const rt_ray *rays, uint64_t n, rt_hit *hits, double *t64);
/* rt_trace_closest for callers that may not allocate (tight serving loops, harnesses that forbid hipMalloc / hipFree while
 * they run).  No counterpart in the reference.  Synchronous like rt_trace_closest: it copies from and to the caller's host
 * buffers on the context's own stream and waits for it, so it is not something to call while a stream is being captured into a
 * HIP graph (such a caller wants rt_trace_closest_device on resident buffers, which enqueues only and, below the sizes at which
 * batches are ordered, allocates nothing either).  Same arguments and results as rt_trace_closest, but the device-side staging of
 * rays, hits and t64 lives in a file-scope device array of the library (56 MiB per GPU, held from library load to process end)
 * instead of buffers grown with hipMalloc: the call allocates and frees nothing.  n > RT_STATIC_MAX_RAYS is refused with
 * RT_ERR_INVALID.  All contexts of one GPU share that GPU's array: calls that overlap on one GPU race (serialise them). */
#define RT_STATIC_MAX_RAYS (1ull << 20)
int rt_trace_closest_static(rt_context *ctx, const rt_ray *rays, uint64_t n, rt_hit *hits, double *t64);
/* replaces: world->Occluded(r, t_min, t_max) (hittable.h:34-40, called at renderer.cpp:263). out[i] = 0/1 */
int rt_trace_any(rt_context *ctx, const rt_ray *rays, uint64_t n, uint8_t *out);
/* same, on buffers already resident in HBM, enqueued on `stream` (a hipStream_t, NULL = the
 * context's own stream); no host synchronisation. d_t64 may be NULL.
 * A batch of at least 2^22 rays in a scene of at least 2M triangles is traced in the order of the cells its origins lie in (the
 * answers come back in the caller's order and are the same ray by ray; the closest-hit call keeps 60 bytes per ray of device
 * memory for this, grown on demand, so the first such call allocates).  MI355RT_RAY_BINS=0 in the environment switches that off,
 * =1 on for every batch.  rt_trace_any_device does the same; rt_trace_closest_static never orders.
 * Those buffers belong to the context, not to the batch: ordered batches of one context run one after the other even when they
 * are enqueued on different streams (the later stream waits on the device for the earlier batch's results to be out; the host
 * never waits).  Batches that are not ordered keep no state and overlap freely.  One context serves one host thread at a time. */
int rt_trace_closest_device(rt_context *ctx, const void *d_rays, uint64_t n, void *d_hits, void *d_t64, void *stream);
int rt_trace_any_device(rt_context *ctx, const void *d_rays, uint64_t n, void *d_out, void *stream);

/* ---- rendering ---- */
/* replaces: the set-up of Renderer::RenderImage (renderer.cpp:56-58): sizes and clears the film
 * (nx*ny*30 floats, element (i*ny+j)*30+bin, the reference's column-major pixel order).
 * d_film: optional caller-owned device buffer of nx*ny*30 floats (e.g. memory an RCCL all-reduce
 * will run on); NULL = the context allocates it.
 * Rays and samples that cannot add to the film are dropped where that is certain (camera rays that miss the scene's bounding
 * box, BSDF-sampled light rays that miss the chosen light's box, paths with zero throughput); the film is the same without that.
 * The environment variable MI355RT_TRACE_ALL_CONNECTIONS=1, read here, switches it off (tests, ray statistics). */
int rt_render_begin(rt_context *ctx, int nx, int ny, int ns, int spectral_samples, int algorithm, uint64_t seed, void *d_film);
/* replaces: passes s = first_sample .. first_sample+n_samples-1 of the loop at renderer.cpp:69-123
 * (camera ray, integrator, film weight, Spectrum::add of rad/ns).  Work is queued on the context's stream.
 * A pass may leave the last few of its paths (the deep, nearly empty bounces) for the next pass to carry along;
 * rt_render_sync, the downloads, rt_get_stats and rt_profile_stop finish them first, so whatever those calls
 * return accounts for every sample of every pass.  Read the buffer of rt_film_device_ptr after rt_render_sync.
 * A pass is rendered in batches of up to 2^27 paths (64 samples per pixel at 1920x1080), and the wavefront queues take about 380
 * bytes per path of the largest batch so far (51 GB for a full one); a pass of fewer samples allocates proportionally less, and the
 * environment variable MI355RT_BATCH_LOG2 (20..28, read by rt_render_begin) caps the batch for hosts that must leave HBM to others.
 * RT_ALG_NAIVE_BDPT also stores its subpath vertices (about 1.65 KB per sample in all): its batches are sized by the device memory
 * that is free at its first pass (three quarters of it, at most 2^27 samples); MI355RT_BDPT_BATCH_LOG2 (20..27) fixes the size. */
int rt_render_pass(rt_context *ctx, int first_sample, int n_samples);
/* finishes pending paths and waits for the context's stream */
int rt_render_sync(rt_context *ctx);
/* replaces: reads of Renderer::spectrum_img (scene.cpp:929): nx*ny*30 doubles, same layout */
int rt_download_film(rt_context *ctx, double *spectrum_img);
/* replaces: the tonemap block renderer.cpp:124-143 for `samples_done` finished passes:
 * rgb(spectrum*ns/samples_done), negative -> 0, pow 1/2.2, min 1; nx*ny*3 doubles */
int rt_download_preview(rt_context *ctx, int samples_done, double *preview_img);
int rt_film_device_ptr(rt_context *ctx, void **d_film, size_t *bytes);

/* ---- several GPUs in one process ---- */
/* replaces: `#pragma omp parallel for` over image columns (renderer.cpp:74-77), the reference's only parallel construct.
 * A group is one context per entry of `devices` (an ordinal may repeat: several contexts on one GPU), each with a full copy of the
 * scene.  A render pass is split by sample index -- member i of n renders the i-th block of the pass's samples of every pixel into
 * its own film, on a host thread of its own -- and random streams are keyed by (pixel, sample), so the image does not depend on n.
 * The downloads gather the member films onto the first member's GPU (peer copies over xGMI) and sum them there; nothing else is
 * exchanged.  The scene calls forward to every member and mean what their rt_* namesakes mean; all return the first failure,
 * whose text rt_group_last_error holds.  One host thread at a time per group. */
typedef struct rt_group rt_group;
int rt_group_create(const int *devices, int n, rt_group **out);
void rt_group_destroy(rt_group *g);
const char *rt_group_last_error(const rt_group *g);
int rt_group_size(const rt_group *g);
rt_context *rt_group_context(rt_group *g, int i); /* member i, for rt_get_stats, rt_set_builder, rt_profile_* */
int rt_group_upload_mesh(rt_group *g, const rt_mesh *mesh);
int rt_group_upload_materials(rt_group *g, const rt_material *per_object, uint32_t n_objects);
int rt_group_upload_material_graphs(rt_group *g, const rt_material_graphs *graphs);
int rt_group_set_environment(rt_group *g, const uint8_t *texels, int width, int height, int bpp, double brightness);
int rt_group_set_camera(rt_group *g, const rt_camera *cam);
int rt_group_set_triangle_test(rt_group *g, int mode);
int rt_group_clear(rt_group *g);
int rt_group_render_begin(rt_group *g, int nx, int ny, int ns, int spectral_samples, int algorithm, uint64_t seed);
int rt_group_render_pass(rt_group *g, int first_sample, int n_samples);
int rt_group_render_sync(rt_group *g);
int rt_group_download_film(rt_group *g, double *spectrum_img);                       /* the sum of the member films */
int rt_group_download_preview(rt_group *g, int samples_done, double *preview_img);   /* tonemap of that sum */
/* milliseconds the last gather-and-sum of the member films took on the first member's GPU (peer copies of all members at once,
 * each over its own xGMI link, and one summing kernel); 0 before the first download */
double rt_group_last_sum_ms(const rt_group *g);
/* the traversal benchmark on a group (replaces the loop of renderer.cpp:74-77 for rays instead of pixels): member i answers the
 * i-th of n contiguous ranges of `rays` on its own replica of the scene; arguments and results as rt_trace_closest */
int rt_group_trace_closest(rt_group *g, const rt_ray *rays, uint64_t n, rt_hit *hits, double *t64);

/* ---- introspection (tests, roofline accounting) ---- */
/* Per-kernel device time measured with HIP events recorded on the stream each kernel is launched on.
 * items = rays for the two traversal kernels, paths / connections for the others. */
typedef struct rt_kernel_time {
	double ms;
	uint64_t launches;
	uint64_t items;
} rt_kernel_time;
typedef struct rt_profile {
	rt_kernel_time closest, any, generate, shade, connect;
} rt_profile;
/* clears the accumulated times and starts recording an event pair around every kernel launch */
int rt_profile_start(rt_context *ctx);
/* waits for the recorded events, sums them per kernel and stops recording */
int rt_profile_stop(rt_context *ctx, rt_profile *out);
/* Diagnostic: while set, every ray a render pass traces (path and connection rays alike) is also copied
 * to `buf` (caller-owned host memory, `capacity` rays) until it is full; *n_captured is updated as the
 * pass runs.  NULL switches it off.  For sampling the traced ray distribution of a small render. */
int rt_render_capture_rays(rt_context *ctx, rt_ray *buf, uint64_t capacity, uint64_t *n_captured);
int rt_get_stats(rt_context *ctx, rt_stats *out);
/* copies the host-built BVH2: nodes (n_nodes records) and tri_index (leaf order -> fan triangle).
 * Pass NULL pointers to query the counts only. */
int rt_get_bvh2(rt_context *ctx, rt_bvh2_node *nodes, uint64_t *n_nodes, uint32_t *tri_index, uint64_t *n_tris);
/* One node of the 8-wide tree the traversal kernels walk, as it lies in HBM (128 bytes = one cache line; the structure that
 * replaces the reference's BVHNode objects, bvh.h:15-34).  Child i occupies [org + qlo[a][i] * cell_a, org + qhi[a][i] * cell_a] on
 * axis a with cell_a = 2^(e[a] - 127); an empty slot has qlo 255 > qhi 0.  Bit i of imask: child i is an interior node, and the
 * interior children are the nodes child_base, child_base + 1, ... in slot order.  A leaf child i holds meta[i] >> 5 triangles
 * starting at leaf position tri_base + (meta[i] & 31).  Children sit in the slot whose three bits say on which side of the node's
 * centre they lie. */
typedef struct rt_wide_node {
	float org[3];
	uint8_t e[3], imask;
	uint32_t child_base, tri_base;
	uint8_t meta[8];
	uint8_t qlo[3][8], qhi[3][8];
	uint8_t pad[48];
} rt_wide_node;
/* copies the wide tree (n_nodes records, node 0 = root) and its leaf order (leaf position -> fan triangle), whichever builder made
 * it.  Pass NULL pointers to query the counts only.  For work counting and tests. */
int rt_get_wide_bvh(rt_context *ctx, rt_wide_node *nodes, uint64_t *n_nodes, uint32_t *tri_index, uint64_t *n_tris);
/* The box that the tree builders give fan triangle (a, b, c) of a polygon with unit face normal nf (host arithmetic, no device
 * needed; for tests).  No counterpart in the reference, which tests every fan triangle on the plane of the polygon's first three
 * vertices (object.cpp:244) and therefore, for a polygon that is not planar, accepts hit points outside the triangle and outside
 * its vertices' box: this box holds every point that test can accept, so that the closest hit found through the tree is the one
 * a loop over all polygons (HittableList::Hit, hittablelist.cpp:3-17) finds. */
void rt_accept_region_box(const double a[3], const double b[3], const double c[3], const double nf[3], double lo[3], double hi[3]);
/* What rt_upload_mesh's cutting of long triangles (rt_set_reference_cutting) does to a mesh, on the host alone (no device needed;
 * for tests).  A triangle that is cut keeps its own reference with the box of its first piece (`first`, ordered by triangle) and
 * gets one additional reference per further piece (`extra`).  Pass NULL arrays to query the counts; min_tris: scenes with fewer
 * fan triangles are left alone (rt_upload_mesh uses 4096).  No counterpart in the reference. */
typedef struct rt_reference {
	uint32_t tri;         /* fan triangle */
	float lo[3], hi[3];   /* box of the piece */
} rt_reference;
int rt_cut_references(const rt_mesh *mesh, double budget, uint32_t min_tris, rt_reference *first, uint64_t *n_first, rt_reference *extra, uint64_t *n_extra);
/* replaces: BiliteralFilter::FilterImage (filter.cpp:13-51; window / sigma_d / sigma_r are the public fields of filter.h:17-27).
 * rgba, out: width * height * 4 doubles, row major, as Filter::orig_img / Filter::result.  Host buffers. */
int rt_filter_bilateral(rt_context *ctx, const double *rgba, int width, int height, int window, double sigma_d, double sigma_r, double *out);
/* the same on buffers already in HBM (they may not alias); runs on the context's stream, rt_render_sync waits for it */
int rt_filter_bilateral_device(rt_context *ctx, const void *d_rgba, int width, int height, int window, double sigma_d, double sigma_r, void *d_out);
/* the context's stream as a hipStream_t, for callers that time with HIP events */
void *rt_stream(rt_context *ctx);

#ifdef __cplusplus
}
#endif
#endif
