// Wavefront path tracer for gfx950: generate -> [extend -> shade -> connect-trace -> connect] per
// bounce, over ray / path-state / connection queues resident in HBM.
//
// It restates Renderer::RenderImage and the three unidirectional integrators of the reference
// (renderer.cpp:54-475) with the random draws taken in the reference's order from one counter-based
// stream per (pixel, sample).  Shading arithmetic is fp32; hit distances come from the exact fp64
// triangle test of trace.hip.  Each radiance term is added to the film the moment it is known
// (emission in shade, direct light in connect), weighted by the sample's film factor, instead of
// being summed per path and added at the end; a NaN term is dropped on its own where the reference
// drops the whole sample (renderer.cpp:115).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>
#include <mutex>
#include <string>
#include <utility>
#include <thread>
#include <vector>
#include "accept_region.h"
#include "parallel_for.h"
#include "ray_bins.h"
#include "render.h"
#include "trace.h"

namespace mi355rt {

namespace {

constexpr int kBlock = 256;
constexpr float kPi = 3.14159265358979323846f;

struct DevMaterial {
	int type, light_flag;
	float a[RT_NBINS], b[RT_NBINS];
	uint32_t prog_begin, prog_len; // slot a (albedo / color) as a per-hit program when prog_len != 0
	float roughness;               // RT_MAT_GGX: alpha
};

// One wavelength band of the pass loop (renderer.cpp:85-88) with everything that depends on it
// precomputed in fp64 on the host: Spectrum::add / integrate index ranges and edge weights
// (spectrum.cpp:126-155), the bin Spectrum::get(central_wl) reads, and the thin-lens image distance.
struct DevBand {
	int min_i, max_i, get_bin, flags; // flags bit0: min_i >= 1, bit1: max_i <= N-2
	float mod_min, mod_max, neg_a_over_d, a_positive;
};

struct DevCamera {
	float origin[3], llc_rel[3], horizontal[3], vertical[3], w[3];
	float half_aperture, weight_scale; // weight_scale = film_sensitivity / (d^2 p_image p_lens)
};

// n / d and n % d for a divisor fixed at launch time without a division instruction (Granlund and Montgomery's multiply-and-shift,
// branch-free form): t = mulhi(mul, n), q = (t + ((n - t) >> s1)) >> s2, exact for every 32-bit n and every d >= 1.
// (A 32-bit division by a run-time value costs about 25 vector instructions; the generate kernel, which runs at the vector
// instruction rate, made three per sample.)
struct FastDiv { uint32_t d, mul, s1, s2; };
inline FastDiv fast_div(uint32_t d)
{
	uint32_t l = 0;
	while (l < 32 && (1ull << l) < d) l++;
	FastDiv f;
	f.d = d;
	f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
	f.s1 = l < 1 ? l : 1;
	f.s2 = l > 1 ? l - 1 : 0;
	return f;
}
__device__ __forceinline__ uint32_t operator/(uint32_t n, const FastDiv &f)
{
	const uint32_t t = __umulhi(f.mul, n);
	return (t + ((n - t) >> f.s1)) >> f.s2;
}
__device__ __forceinline__ uint32_t operator%(uint32_t n, const FastDiv &f) { return n - (n / f) * f.d; }

struct DeviceShade {
	const float4 *tri_shade;        // 6 float4 per fan triangle: v0 v1 v2 n0 n1 n2 face_normal area2
	const uint32_t *tri_face, *face_object, *face_tri_begin;
	const float *face_area, *tri_cum;
	const uint32_t *tri_leaf;       // fan triangle -> leaf-order index (exact polygon test in NEE)
	const DevMaterial *materials;
	const uint32_t *light_faces;
	const uint4 *light_rec;         // per light two uint4: (polygon, its first fan triangle, its number of fan triangles, its area as float
	                                // bits) and (the cumulative area shares of its first three fan triangles as float bits, its obj object): what a
	                                // light pick needs, in one read instead of a chain of dependent ones (polygons of more than four
	                                // fan triangles go on in tri_cum)
	int cull_connections;           // 0: trace every connection ray (diagnostic, MI355RT_TRACE_ALL_CONNECTIONS=1)
	const float4 *light_box;        // per light: (lo.xyz, largest |coordinate|), (hi.xyz, unused): bounding box of the polygon
	const DevBand *bands;
	const float *emit;              // [object * n_bands + band]: Emitted() of a light whose colour is a constant spectrum, integrated over the
	                                // band (0 for other materials): one read where the material record costs a chain of them
	uint32_t n_lights, n_bands;
	FastDiv by_lights;              // division by n_lights
	const float *tri_uv;            // 9 floats per fan triangle: vt of its three corners (null: the mesh has none)
	const uint2 *g_code;            // per-hit material programs: (op, arg) pairs
	const float *g_spectra, *g_vectors, *g_scalars;
	const int4 *g_tex;              // per texture: byte offset into g_texels, width, height, bytes per texel
	const uint8_t *g_texels;
	const uint8_t *env;             // environment texture of the naive integrator (null: none), env_h rows of env_w texels
	int env_w, env_h, env_bpp;
	float env_brightness;
};

// 24 bytes per live path (padded to 32 it made the stores of the generate kernel and the reads of the shade kernels that much longer)
struct PathState {
	uint64_t rng;
	float beta, weight;
	uint32_t pixel;
	uint32_t band_bounce; // band | bounce << 16
};
static_assert(sizeof(PathState) == 24, "PathState is stored as written");

// 32 bytes per connection ray (direct-light estimate waiting for its visibility query); NaiveBDPT's record
struct ConnRec {
	float c0, c1;
	float nx, ny, nz;
	uint32_t sel_face, pixel, band_kind; // band | kind << 16; kind 0: MIS light sample, 1: MIS bsdf sample, 2: NEE
};

// 24 bytes: the record of the unidirectional integrators.  The connection's geometry term needs the surface normal only in its
// product with the ray's direction, which the shade kernel has at hand: one float instead of three (the shade kernels write and
// the connect kernel reads a quarter less per connection)
struct alignas(8) ConnRecU {
	float c0, c1;
	float cos_surface; // |direction . shading normal| at the surface the ray leaves, direction normalised
	uint32_t sel_face, pixel, band_kind;
};
static_assert(sizeof(ConnRecU) == 24, "ConnRecU is stored as written");

// queue tails packed in one 64-bit word (low: continuation rays, high: connection rays) so that a
// workgroup reserves its slots in both queues with a single atomic
struct Counters {
	unsigned long long tails;
	unsigned long long pad;
};
#ifndef CARRY_DIVISOR
#define CARRY_DIVISOR 8
#endif
constexpr uint64_t kCarryDivisor = CARRY_DIVISOR; // a batch hands its last paths to the next one once fewer than 1/8 of it are alive (swept at 1080p MIS: 64 -> 2039, 32 -> 2066, 16 -> 2065, 8 -> 2088, 4 -> 2098 spp/s; without 1875)
constexpr int kMaxBounces = 62; // counter slots per batch; a longer path wraps around (see run_batch)
// threads per workgroup of the shade kernels.  One slot-reservation atomic per workgroup argues for large groups (a per-wave atomic
// on one address serialises), turnover argues for small ones: a 1024-thread group is alone on its CU at this register count and the
// CU drains at its final barrier; 512 measured 19 % faster than 1024 and 11 % faster than 256
#ifndef SHADE_BLOCK
#define SHADE_BLOCK 512
#endif
// NaiveBDPT samples per batch, at most: 16 subpath vertices and a sample record (832 bytes), the path queues (365) and, on the
// Cornell box, 4.6 connection slots (80 bytes each) per sample, about 1.6 KB.  Late bounces hold few walkers, and a small batch pays
// their launches for few samples (1080p: 2^22 454 spp/s, 2^24 570, 2^25 591 with the first version of the stages; now 2^25 1061,
// 2^26 1106, 2^27 1232): the batch is as large as fits in three quarters of the memory that is free when the first batch is set up
// (2^27 on an otherwise empty 288 GB GPU); MI355RT_BDPT_BATCH_LOG2 (20..27) fixes it.
#ifndef BDPT_BATCH_LOG2_MAX
#define BDPT_BATCH_LOG2_MAX 27
#endif
constexpr uint64_t kBdptBytesPerSample = 1650;
// Paths per batch of the unidirectional integrators: a batch pays the launches of its late, nearly empty bounces once, so large
// batches are cheaper per sample (1080p NEE+MIS: 2^25 paths 3660 spp/s, 2^26 3950, 2^27 4100, 2^28 4210) and queue memory grows
// with them (about 365 bytes per path: 49 GB at 2^27, allocated only as far as a pass needs it).  The environment variable
// MI355RT_BATCH_LOG2 (20..28), read by rt_render_begin, overrides the default.
#ifndef PATH_BATCH_LOG2
#define PATH_BATCH_LOG2 27
#endif
#ifndef SHADE_MIN_BLOCKS
#define SHADE_MIN_BLOCKS 1
#endif
constexpr int kShadeBlock = SHADE_BLOCK;
#ifndef STREAM_GRID_WAVES
#define STREAM_GRID_WAVES 64
#endif
// workgroups for a kernel that walks a queue of at most `bound` items in strides: enough waves to fill the chip several times over
// (256 CUs x STREAM_GRID_WAVES waves), never more than the bound needs
inline unsigned stream_grid(uint64_t bound, int block, int waves = STREAM_GRID_WAVES) { return (unsigned)std::min<uint64_t>((bound + block - 1) / block, 256ull * waves * 64 / block); }

struct Impl {
	// shading data
	float4 *tri_shade = nullptr;
	uint32_t *face_tri_begin = nullptr, *tri_leaf = nullptr;
	float *face_area = nullptr, *tri_cum = nullptr;
	DevMaterial *materials = nullptr;
	uint32_t *light_faces = nullptr;
	uint4 *light_rec = nullptr;
	float4 *light_box = nullptr;
	bool cull_connections = true;
	int cull_mask = 7;
	int batch_log2 = PATH_BATCH_LOG2;
	float4 scene_lo = {0, 0, 0, 0}, scene_hi = {0, 0, 0, 0}; // bounding box of all vertices, rounded outwards (scene_lo.w: largest |coordinate|)
	DevBand *bands = nullptr;
	uint32_t n_lights = 0, n_bands = 0;
	float *tri_uv = nullptr;
	uint2 *g_code = nullptr;
	float *g_spectra = nullptr, *g_vectors = nullptr, *g_scalars = nullptr;
	int4 *g_tex = nullptr;
	uint8_t *g_texels = nullptr;
	std::vector<DevMaterial> host_materials; // kept so that programs can be attached after the records went up
	uint8_t *env = nullptr;
	int env_w = 0, env_h = 0, env_bpp = 0;
	float env_brightness = 0.0f;
	const uint32_t *tri_face = nullptr, *face_object = nullptr;
	DevCamera cam{};
	// queues
	uint64_t capacity = 0, conn_capacity = 0;
	rt_ray *rays[2] = {nullptr, nullptr};
	PathState *state[2] = {nullptr, nullptr};
	rt_hit *hits = nullptr;
	rt_ray *conn_rays = nullptr;
	ConnRec *conn = nullptr;
	rt_hit *conn_hits = nullptr;
	// rays of the later bounces in a scene far larger than the caches are traced in the order of the cells their origins lie in
	// (ray_bins.h): the ordered copy, where each ray went, the hits in that order, the cell counters (allocated with the first such launch)
	rt_ray *bin_rays = nullptr;
	uint32_t *bin_perm = nullptr, *bin_counters = nullptr;
	rt_hit *bin_hits = nullptr;
	uint64_t bin_capacity = 0;
	uint8_t *conn_any = nullptr;
	Counters *counters = nullptr;   // kMaxBounces + 2 slots: slot b = paths entering bounce b (low) | connections made at bounce b-1 (high)
	Counters *h_counters = nullptr; // pinned mirror, filled by asynchronous copies
	hipEvent_t ev[kMaxBounces + 2] = {};
	std::vector<float> film_host;
	// paths of the previous batch that were still alive when it was cut short (see run_batch): they sit at the front of rays[carry_cur]
	uint64_t carry_count = 0;
	int carry_cur = 0;
	// pixels some camera ray of which may touch the scene (see live_pixel_list); null: every pixel
	uint32_t *live_pixels = nullptr;
	uint64_t n_live = 0;
	// NaiveBDPT: per-sample records, subpath vertices, connection slot counts and offsets
	void *bdpt_samples = nullptr, *bdpt_verts = nullptr, *bdpt_scan_tmp = nullptr;
	uint32_t *bdpt_counts = nullptr, *bdpt_offsets = nullptr;
	unsigned long long *bdpt_pairs = nullptr;
	float *preview = nullptr; // tonemapped image on the device (render_download_preview)
	size_t preview_size = 0;
	float *stage[2] = {nullptr, nullptr}; // pinned pieces films and previews are downloaded through (download_widened)
	hipEvent_t stage_ev[2] = {};
	float *emit = nullptr;   // see DeviceShade::emit; rebuilt before a pass whenever materials, programs or bands have changed
	size_t emit_size = 0;
	bool emit_dirty = true;
	uint64_t bdpt_capacity = 0;
	int bdpt_batch_log2 = 0; // 0: not chosen yet (see BDPT_BATCH_LOG2_MAX)
	size_t bdpt_scan_bytes = 0;
};

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(float t, f3 a) { return F3(t * a.x, t * a.y, t * a.z); }
__device__ __forceinline__ f3 neg(f3 a) { return F3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float len(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 unit(f3 a) { float l = len(a); return F3(a.x / l, a.y / l, a.z / l); }
// The same with the instructions spelled out (v_rsq_f32 and three products): code that two kernels must execute identically cannot
// leave the choice between a division and a shared reciprocal to the compiler, which decides anew wherever a function is inlined
__device__ __forceinline__ f3 unit_rsq(f3 a) { const float r = __builtin_amdgcn_rsqf(dot(a, a)); return F3(a.x * r, a.y * r, a.z * r); }
__device__ __forceinline__ f3 ld3(const float *p) { return F3(p[0], p[1], p[2]); }

// the random stream shared with the CPU checker: a per-(pixel, sample) seed from two splitmix64 rounds, then PCG draws
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rng_seed(uint64_t seed, uint64_t pixel, uint64_t sample)
{
	return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL * (sample + 1)) ^ (0xD1B54A32D192ED03ULL * (pixel + 1)));
}
// one draw: PCG-XSH-RR 64/32 -- a 64-bit linear congruential step and a 32-bit output permuted from the old state.  One 64-bit
// multiplication per draw (four quarter-rate 32-bit multiplies) where splitmix64 per draw needs two: the generate kernel runs at
// the vector instruction rate, and these multiplies are its most expensive instructions
__device__ __forceinline__ uint32_t rng_next(uint64_t &s)
{
	const uint64_t old = s;
	s = old * 6364136223846793005ULL + 1442695040888963407ULL;
	const uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
	return __builtin_rotateright32(xorshifted, rot);
}
// top 24 bits: the truncation of the checker's 32-bit uniform, so threshold decisions agree
__device__ __forceinline__ float rng_f32(uint64_t &s) { return (float)(rng_next(s) >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ uint32_t rng_below(uint64_t &s, const FastDiv &n) { return rng_next(s) % n; }

struct Onb { f3 u, v, w; };
__device__ __forceinline__ Onb onb_from_w(f3 n) // onb.cpp:31-42
{
	Onb b;
	b.w = unit(n);
	f3 a = fabsf(b.w.x) > 0.9f ? F3(0, 1, 0) : F3(1, 0, 0);
	b.v = unit(cross(b.w, a));
	b.u = cross(b.w, b.v);
	return b;
}
__device__ __forceinline__ f3 l2w(const Onb &b, f3 a) { return a.x * b.u + a.y * b.v + a.z * b.w; } // onb.cpp:21-24
__device__ __forceinline__ f3 w2l(const Onb &b, f3 a) { return F3(dot(b.u, a), dot(b.v, a), dot(b.w, a)); } // onb.cpp:26-29

__device__ __forceinline__ float spectrum_integrate(const float *data, const DevBand &bd) // spectrum.cpp:126-146
{
	float sum = 0.0f;
	if (bd.flags & 1) sum += data[bd.min_i - 1] * bd.mod_min;
	if (bd.flags & 2) sum += data[bd.max_i + 1] * bd.mod_max;
	for (int i = bd.min_i; i <= bd.max_i; i++) sum += data[i] * 10.0f;
	return sum;
}

__device__ __forceinline__ void film_add(float *film, uint32_t pixel, const DevBand &bd, float value) // spectrum.cpp:148-155
{
	if (isnan(value)) return; // renderer.cpp:115
#if defined(EXP_FILM_STORE)
	for (int i = bd.min_i; i <= bd.max_i; i++) film[(size_t)pixel * RT_NBINS + i] = value;
#elif defined(EXP_FILM_NONE)
	if (value == 12345.678f) film[pixel] = value;
#else
	for (int i = bd.min_i; i <= bd.max_i; i++) atomicAdd(&film[(size_t)pixel * RT_NBINS + i], value);
#endif
}

// ---- per-hit material graphs: the Compute() chains of materialnode.cpp that read the hit's texture coordinate, compiled by the
// host to a postfix program for slot a (albedo / color).  Values are vectors, spectra live in .x (one bin is evaluated at a time).
// The value stack is six registers deep (rt_upload_material_graphs refuses deeper programs).
enum { OP_SPEC = 0, OP_VEC, OP_UV, OP_ZERO_S, OP_ZERO_V, OP_IMAGE, OP_CHECKER, OP_RGB2SPEC, OP_ADD_S, OP_MUL_S, OP_ADD_V, OP_MUL_V,
       OP_ACCESS, OP_COMBINE, OP_NOISE1, OP_NOISE2, OP_ROTATE, OP_SCALAR }; // scalars live in .x like spectra (ADD_S / MUL_S serve both)

__device__ float eval_graph(const DeviceShade &sh, uint32_t begin, uint32_t len, f3 uv, int bin)
{
#pragma clang fp reciprocal(off) // texel, checker and noise-cell indices are truncated quotients: divisions stay divisions here
	f3 s0 = F3(0, 0, 0), s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0; // s0 is the top
	for (uint32_t pc = begin; pc < begin + len; pc++) {
		const uint2 c = sh.g_code[pc];
		f3 v = F3(0, 0, 0);
		int delta = 0; // +1 push v, -1 replace the two top values by v, 0 replace the top by v
		switch (c.x) {
		case OP_SPEC: v = F3(sh.g_spectra[(size_t)c.y * RT_NBINS + bin], 0, 0); delta = 1; break;
		case OP_VEC: v = ld3(sh.g_vectors + 3 * (size_t)c.y); delta = 1; break;
		case OP_UV: v = uv; delta = 1; break; // UVNode, materialnode.cpp:893-896
		case OP_ZERO_S:
		case OP_ZERO_V: delta = 1; break;
		case OP_IMAGE: { // ImageTextureNode::Compute, :1007-1018; an address outside the texture (undefined in the reference) reads its nearest end
			const int4 t = sh.g_tex[c.y];
			const int x = (int)((double)t.y * (double)s0.x), y = (int)((double)t.z * (1.0 - (double)s0.y));
			long idx = (long)x + (long)y * t.y;
			idx = idx < 0 ? 0 : (idx > (long)t.y * t.z - 1 ? (long)t.y * t.z - 1 : idx);
			const uint8_t *px = sh.g_texels + t.x + (long)t.w * idx;
			v = F3((float)((double)px[0] / 255.0), (float)((double)px[1] / 255.0), (float)((double)px[2] / 255.0));
			break;
		}
		case OP_CHECKER: { // CheckerboardNode::Compute, :1055-1062
			const double size = (double)sh.g_scalars[c.y], u = (double)s0.x, w = (double)s0.y;
			const int x = (int)((u - fmod(u, size)) / size), y = (int)((w - fmod(w, size)) / size);
			v = (x % 2 == y % 2) ? F3(1, 1, 1) : F3(0, 0, 0);
			break;
		}
		case OP_RGB2SPEC: v = F3(bin < RT_NBINS / 3 ? s0.z : (bin < 2 * RT_NBINS / 3 ? s0.y : s0.x), 0, 0); break; // spectrum.cpp:219-232
		case OP_ADD_S: v = F3(s1.x + s0.x, 0, 0); delta = -1; break;
		case OP_MUL_S: v = F3(s1.x * s0.x, 0, 0); delta = -1; break;
		case OP_ADD_V: v = s1 + s0; delta = -1; break;
		case OP_MUL_V: v = F3(s1.x * s0.x, s1.y * s0.y, s1.z * s0.z); delta = -1; break;
		case OP_ACCESS: v = F3(c.y == 0 ? s0.x : (c.y == 1 ? s0.y : s0.z), 0, 0); break;   // AccessVec3ComponentNode, :1293-1301
		case OP_COMBINE: v = F3(s2.x, s1.x, s0.x); delta = -2; break;                       // CombineVec3ComponentNode, :1333-1347
		case OP_SCALAR: v = F3(sh.g_scalars[c.y], 0, 0); delta = 1; break;
		case OP_NOISE1: { // ValueNoiseNode::Compute, :1366-1384; index arithmetic in double as there
			const float *r = sh.g_scalars + c.y + 1;
			const int n = (int)sh.g_scalars[c.y];
			double u = fmod((double)s0.x, 1.0);
			if (u < 0.0) u += 1.0;
			const double us = 1.0 / n, rem = fmod(u, us);
			int x0 = (int)((u - rem) / us), x1 = x0 + 1;
			if (x1 > n) { x1 = n; x0 = x0 > n ? n : x0; } // only through rounding at u -> 1 (the reference asserts)
			const float t = (float)(rem / us);
			v = F3((1.0f - t) * r[x0] + t * r[x1], 0, 0);
			break;
		}
		case OP_NOISE2: { // ValueNoise2DNode::Compute, :1447-1478
			const unsigned nu = (unsigned)sh.g_scalars[c.y], nv = (unsigned)sh.g_scalars[c.y + 1];
			const float *r = sh.g_scalars + c.y + 2;
			double u = fmod((double)s1.x, 1.0), w = fmod((double)s0.x, 1.0);
			if (u < 0.0) u += 1.0;
			if (w < 0.0) w += 1.0;
			const double us = 1.0 / nu, ws = 1.0 / nv, urem = fmod(u, us), wrem = fmod(w, ws);
			unsigned u0 = (unsigned)((u - urem) / us), w0 = (unsigned)((w - wrem) / ws);
			u0 = u0 > nu - 1 ? nu - 1 : u0; // the reference would read past its table
			w0 = w0 > nv - 1 ? nv - 1 : w0;
			const unsigned u1 = u0 + 1, w1 = w0 + 1;
			float ut = (float)(urem / us), wt = (float)(wrem / ws);
			ut = ut * ut * (3.0f - 2.0f * ut);
			wt = wt * wt * (3.0f - 2.0f * wt);
			const float a = (1.0f - ut) * r[u0 * (nv + 1) + w0] + ut * r[u1 * (nv + 1) + w0];
			const float b = (1.0f - ut) * r[u0 * (nv + 1) + w1] + ut * r[u1 * (nv + 1) + w1];
			v = F3((1.0f - wt) * a + wt * b, 0, 0);
			delta = -1;
			break;
		}
		case OP_ROTATE: { // RodriguesRotationNode::Compute, :1536-1573: s2 = v, s1 = n, s0 = theta
			const f3 nn = unit(s1);
			float sn, cs;
			sincosf(s0.x, &sn, &cs);
			const f3 r0 = F3(cs + nn.x * nn.x * (1.0f - cs), nn.x * nn.y * (1.0f - cs) - nn.z * sn, nn.x * nn.z * (1.0f - cs) + nn.y * sn);
			const f3 r1 = F3(nn.y * nn.x * (1.0f - cs) + nn.z * sn, cs + nn.y * nn.y * (1.0f - cs), nn.y * nn.z * (1.0f - cs) - nn.x * sn);
			const f3 r2 = F3(nn.z * nn.x * (1.0f - cs) - nn.y * sn, nn.z * nn.y * (1.0f - cs) + nn.x * sn, cs + nn.z * nn.z * (1.0f - cs));
			v = F3(dot(r0, s2), dot(r1, s2), dot(r2, s2));
			delta = -2;
			break;
		}
		default: break;
		}
		if (delta > 0) { s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; }
		else if (delta == -1) { s1 = s2; s2 = s3; s3 = s4; s4 = s5; }
		else if (delta == -2) { s1 = s3; s2 = s4; s3 = s5; }
		s0 = v;
	}
	return s0.x;
}

// slot a of the material at a hit with texture coordinate uv, read at one bin
// GRAPHS = false: instantiations for scenes without per-hit programs, which then carry none of the evaluator's registers
template <bool GRAPHS = true>
__device__ __forceinline__ float slot_a(const DeviceShade &sh, const DevMaterial &m, f3 uv, int bin)
{
	return (GRAPHS && m.prog_len) ? eval_graph(sh, m.prog_begin, m.prog_len, uv, bin) : m.a[bin];
}

template <bool GRAPHS = true>
__device__ __forceinline__ float mat_emitted(const DeviceShade &sh, const DevMaterial &m, f3 uv, const DevBand &bd) // materialnode.cpp:215-218,739-749
{
	if (m.type != RT_MAT_DIFFUSE_LIGHT) return 0.0f;
	if (!GRAPHS || !m.prog_len) return spectrum_integrate(m.a, bd);
	float sum = 0.0f; // Spectrum::integrate (spectrum.cpp:126-146) over the spectrum the graph yields at this hit
	if (bd.flags & 1) sum += eval_graph(sh, m.prog_begin, m.prog_len, uv, bd.min_i - 1) * bd.mod_min;
	if (bd.flags & 2) sum += eval_graph(sh, m.prog_begin, m.prog_len, uv, bd.max_i + 1) * bd.mod_max;
	for (int i = bd.min_i; i <= bd.max_i; i++) sum += eval_graph(sh, m.prog_begin, m.prog_len, uv, i) * 10.0f;
	return sum;
}

// the same through the per-(object, band) table where the light's colour does not depend on the hit
template <bool GRAPHS>
__device__ __forceinline__ float emitted_at(const DeviceShade &sh, uint32_t object, const DevMaterial &m, f3 uv, const DevBand &bd, uint32_t band)
{
	if (GRAPHS && m.prog_len) return mat_emitted<GRAPHS>(sh, m, uv, bd);
	return sh.emit[(size_t)object * sh.n_bands + band];
}

__global__ void __launch_bounds__(256) k_emit_table(const DevMaterial *__restrict__ materials, uint32_t n_objects, const DevBand *__restrict__ bands,
                                                    uint32_t n_bands, float *__restrict__ out)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_objects * n_bands) return;
	const DevMaterial &m = materials[i / n_bands];
	out[i] = m.type == RT_MAT_DIFFUSE_LIGHT ? spectrum_integrate(m.a, bands[i % n_bands]) : 0.0f; // mat_emitted without a program
}

// RT_MAT_GGX: the micro-normal density of GGXReflectionNode::SampleBSDF (materialnode.cpp:669-675) as a function of the cosine between
// micro-normal and shading normal, d = alpha^2 max(cos, 0) / (pi cos^4 (alpha^2 + tan^2)^2); BSDF = d / (4 |vi.z vo.z|), pdf = d / 4
__device__ __forceinline__ float ggx_d(float alpha, float cos_theta)
{
	if (!(cos_theta > 0.0f)) return 0.0f;
	const float x = cos_theta;
	const float cos2 = cos_theta * cos_theta, sin_theta = sqrtf(fmaxf(1.0f - cos2, 0.0f)), tan_theta = sin_theta / cos_theta;
	const float q = alpha * alpha + tan_theta * tan_theta;
	return alpha * alpha * x / (kPi * cos2 * cos2 * (q * q));
}
__device__ __forceinline__ float ggx_half_z(f3 vi, f3 vo) { return fabsf(unit(vi + vo).z); }

__device__ __forceinline__ bool mirror_pair(f3 vi, f3 vo) { return -vi.x == vo.x && -vi.y == vo.y && vi.z == vo.z; }

// a: slot a of the material at this hit and bin (slot_a)
__device__ __forceinline__ float mat_bsdf(const DevMaterial &m, f3 vi, f3 vo, float a)
{
	switch (m.type) {
	case RT_MAT_GGX: return vi.z * vo.z > 0.0f ? ggx_d(m.roughness, ggx_half_z(vi, vo)) / (4.0f * fabsf(vi.z * vo.z)) : 0.0f; // a reflection or nothing
	case RT_MAT_LAMBERTIAN: return vi.z < 0.0f ? 0.0f : a / kPi;                               // materialnode.cpp:278-288
	case RT_MAT_CONDUCTOR:
	case RT_MAT_COLORED_METAL: return mirror_pair(vi, vo) ? __int_as_float(0x7f800000) : 0.0f; // :503-508, :585-590
	default: return 0.0f;
	}
}
__device__ __forceinline__ float mat_pdf(const DevMaterial &m, f3 vi, f3 vo)
{
	switch (m.type) {
	case RT_MAT_GGX: return vi.z * vo.z > 0.0f ? ggx_d(m.roughness, ggx_half_z(vi, vo)) / 4.0f : 0.0f;
	case RT_MAT_LAMBERTIAN: return vi.z < 0.0f ? 0.0f : vi.z / kPi;                            // :289-295
	case RT_MAT_CONDUCTOR:
	case RT_MAT_COLORED_METAL: return mirror_pair(vi, vo) ? __int_as_float(0x7f800000) : 0.0f; // :509-514, :591-596
	default: return 0.0f;
	}
}

__device__ __forceinline__ float rfresnel(float cos_theta, float n) // material.cpp:14-25
{
	const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
	const float sin_t = sin_theta / n;
	if (sin_t > 1.0f) return 1.0f;
	const float q = sqrtf(n * n - sin_theta * sin_theta);
	const float rte = (cos_theta - q) / (cos_theta + q);
	const float rtm = (n * n * cos_theta - q) / (n * n * cos_theta + q);
	return (rte * rte + rtm * rtm) / 2.0f;
}

__device__ __forceinline__ float cfresnel(float cos_theta, float nre, float nim) // material.cpp:26-34
{
	const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
	const float sin_t = sin_theta / nre;
	if (sin_t > 1.0f) return 1.0f;
	// q = n*n - sin^2, s = principal sqrt(q)
	float qre = nre * nre - nim * nim - sin_theta * sin_theta, qim = 2.0f * nre * nim;
	float m = hypotf(qre, qim), sre, sim;
	if (m == 0.0f) { sre = 0.0f; sim = 0.0f; }
	else if (qre >= 0.0f) { sre = sqrtf(0.5f * (m + qre)); sim = qim / (2.0f * sre); }
	else { float t = sqrtf(0.5f * (m - qre)); sre = fabsf(qim) / (2.0f * t); sim = qim < 0.0f ? -t : t; }
	float are = cos_theta - sre, aim = -sim, bre = cos_theta + sre, bim = sim;
	float den = bre * bre + bim * bim;
	float fre = (are * bre + aim * bim) / den, fim = (aim * bre - are * bim) / den;
	float r = hypotf(fre, fim);
	return r * r;
}

// NodeMaterial::SampleBSDF for ImportanceType rays (materialnode.cpp:1758-1766 and the node bodies)
// light_type: RayType LightType (the light subpath of NaiveBDPT, renderer.cpp:589); only the dielectric's refraction tells them apart
__device__ __forceinline__ bool mat_sample(const DevMaterial &m, f3 normal, const Onb &uvw, f3 vo, int bin, float a, uint64_t &rng,
                                           f3 &vi, float &ratio, float &pdfval, bool light_type = false)
{
	switch (m.type) {
	case RT_MAT_LAMBERTIAN: { // :262-276.  CosinePdf builds a frame of its own from rec.normal (pdf.cpp:36-46,70-84), maps the sample to
		// the world and the caller maps it back with its frame: every caller here builds `uvw` from that same normal with the same
		// function, so the two frames are one, the round trip is the identity (to fp64 rounding in the reference) and
		// dot(generated, w) is the sample's z.  The sample is used in the local frame directly.
		(void)normal;
		float r1 = rng_f32(rng), r2 = rng_f32(rng);
		// phi = 2 pi r1: the hardware sine and cosine take their argument in revolutions
		const float st = sqrtf(r2), sphi = __builtin_amdgcn_sinf(r1), cphi = __builtin_amdgcn_cosf(r1);
		vi = F3(st * cphi, st * sphi, 1.0f - r2); // z = 1 - r2 as in the reference
		const float cos_theta = vi.z;
		pdfval = cos_theta >= 0.0f ? cos_theta / kPi : 0.0f;
		float bsdf = mat_bsdf(m, vi, vo, a);
		ratio = bsdf / pdfval;
		return true;
	}
	case RT_MAT_DIELECTRIC: { // :354-407
		float ref_idx = m.a[bin], surface_scale = m.b[bin];
		float cos_o = vo.z, nz = 1.0f, n1, n2;
		if (cos_o >= 0.0f) { n1 = 1.0f; n2 = ref_idx; }
		else { nz = -1.0f; n1 = ref_idx; n2 = 1.0f; cos_o = fabsf(cos_o); }
		float rel = n2 / n1;
		float fresnel = rfresnel(cos_o, rel);
		float r = rng_f32(rng);
		pdfval = __int_as_float(0x7f800000);
		if (r <= fresnel) {
			vi = F3(-vo.x, -vo.y, vo.z);
			ratio = surface_scale / cos_o;
		} else {
			float sin_o = sqrtf(1.0f - cos_o * cos_o), sin_t = sin_o / rel, cos_t = sqrtf(1.0f - sin_t * sin_t);
			f3 h = unit(F3(vo.x, vo.y, 0.0f));
			vi = F3(-sin_t * h.x, -sin_t * h.y, -cos_t * nz - sin_t * h.z);
			ratio = light_type ? surface_scale * (rel * rel * rel) / cos_t : surface_scale / (rel * rel * rel) / cos_t; // :399-403
		}
		return true;
	}
	case RT_MAT_CONDUCTOR: { // :473-502
		float nre = m.a[bin], kim = m.b[bin], cos_o = vo.z, rre, rim;
		if (cos_o >= 0.0f) { rre = nre; rim = kim; } // n2/n1 with n1 = 1
		else { float den = nre * nre + kim * kim; rre = nre / den; rim = -kim / den; cos_o = fabsf(cos_o); } // 1/(n+ik)
		float fresnel = cfresnel(cos_o, rre, rim);
		vi = F3(-vo.x, -vo.y, vo.z);
		pdfval = __int_as_float(0x7f800000);
		ratio = fresnel / cos_o;
		return true;
	}
	case RT_MAT_GGX: { // :647-689 in the shading frame, alpha from the record (see ggx_d)
		const float alpha = m.roughness;
		const float r1 = rng_f32(rng), u = rng_f32(rng);
		const float cos_theta = sqrtf((1.0f - u) / ((alpha * alpha - 1.0f) * u + 1.0f));
		const float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
		const f3 mn = F3(sin_theta * __builtin_amdgcn_cosf(r1), sin_theta * __builtin_amdgcn_sinf(r1), cos_theta);
		const float d = ggx_d(alpha, cos_theta);
		vi = neg(vo) + (2.0f * dot(vo, mn)) * mn; // reflect2, material.cpp:98-101
		pdfval = d * fabsf(mn.z) / (4.0f * fabsf(mn.z));
		ratio = vi.z * vo.z > 0.0f ? fabsf(dot(vo, mn)) / fabsf(vo.z * vi.z * mn.z) : 0.0f; // reflected below the surface: no energy
		return true;
	}
	case RT_MAT_COLORED_METAL: // :564-584
		vi = F3(-vo.x, -vo.y, vo.z);
		pdfval = __int_as_float(0x7f800000);
		ratio = a / fabsf(vo.z);
		return true;
	default: return false; // base class, :203-206
	}
}

__device__ __forceinline__ float power_heuristic2(float pdf1, float pdf2) // pdf.cpp:139-146 with beta = 2
{
	if (isinf(pdf1)) return 1.0f;
	return pdf1 * pdf1 / (pdf1 * pdf1 + pdf2 * pdf2);
}

struct TriShade { f3 v0, v1, v2, n0, n1, n2, fn; float area2; };
__device__ __forceinline__ TriShade load_tri(const float4 *tri_shade, uint32_t tri)
{
	const float4 s0 = tri_shade[6 * (size_t)tri], s1 = tri_shade[6 * (size_t)tri + 1], s2 = tri_shade[6 * (size_t)tri + 2];
	const float4 s3 = tri_shade[6 * (size_t)tri + 3], s4 = tri_shade[6 * (size_t)tri + 4], s5 = tri_shade[6 * (size_t)tri + 5];
	TriShade t;
	t.v0 = F3(s0.x, s0.y, s0.z); t.v1 = F3(s0.w, s1.x, s1.y); t.v2 = F3(s1.z, s1.w, s2.x);
	t.n0 = F3(s2.y, s2.z, s2.w); t.n1 = F3(s3.x, s3.y, s3.z); t.n2 = F3(s3.w, s4.x, s4.y);
	t.fn = F3(s4.z, s4.w, s5.x); t.area2 = s5.y;
	return t;
}
// rec.normal of HitTriangle (object.cpp:249-256): sub-triangle areas as weights, not renormalised
__device__ __forceinline__ f3 shading_normal(const TriShade &t, f3 p)
{
	f3 r0 = cross(t.v1 - t.v0, p - t.v1), r1 = cross(t.v2 - t.v1, p - t.v2), r2 = cross(t.v0 - t.v2, p - t.v0);
	f3 s = len(r1) * t.n0 + len(r2) * t.n1 + len(r0) * t.n2;
	return F3(s.x / t.area2, s.y / t.area2, s.z / t.area2);
}

// rec.vt of HitTriangle (object.cpp:258): the same sub-triangle weights as the normal; zero when the mesh has no vt
__device__ __forceinline__ f3 hit_uv(const DeviceShade &sh, uint32_t tri_id, const TriShade &t, f3 p)
{
	if (!sh.tri_uv) return F3(0, 0, 0);
	const float *q = sh.tri_uv + 9 * (size_t)tri_id;
	const f3 r0 = cross(t.v1 - t.v0, p - t.v1), r1 = cross(t.v2 - t.v1, p - t.v2), r2 = cross(t.v0 - t.v2, p - t.v0);
	const f3 s = len(r1) * ld3(q) + len(r2) * ld3(q + 3) + len(r0) * ld3(q + 6);
	return F3(s.x / t.area2, s.y / t.area2, s.z / t.area2);
}

// Workgroup-aggregated queue reservation: every lane asks for `survive` (0/1) continuation slots and
// for up to two connection slots; one 64-bit atomic per workgroup reserves both ranges.  (One atomic
// per wave on a single address serialises at ~10 ns each and dominated the shade kernel.)
struct Slots { uint32_t next, c0, c1; };
__device__ __forceinline__ Slots reserve_slots(unsigned long long *tails, bool survive, bool want_c0, bool want_c1)
{
	__shared__ uint32_t s_next[kShadeBlock / 64], s_conn[kShadeBlock / 64];
	__shared__ uint32_t s_base_next, s_base_conn;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	const unsigned long long ms = __ballot(survive), m0 = __ballot(want_c0), m1 = __ballot(want_c1);
	if (lane == 0) {
		s_next[wave] = (uint32_t)__popcll(ms);
		s_conn[wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t an = 0, ac = 0;
		const int nw = (blockDim.x + 63) >> 6;
		for (int w = 0; w < nw; w++) {
			const uint32_t tn = s_next[w], tc = s_conn[w];
			s_next[w] = an;
			s_conn[w] = ac;
			an += tn;
			ac += tc;
		}
		unsigned long long base = 0;
		if (an | ac) base = atomicAdd(tails, (unsigned long long)an | ((unsigned long long)ac << 32));
		s_base_next = (uint32_t)base;
		s_base_conn = (uint32_t)(base >> 32);
	}
	__syncthreads();
	Slots r;
	r.next = s_base_next + s_next[wave] + (uint32_t)__popcll(ms & lt);
	r.c0 = s_base_conn + s_conn[wave] + (uint32_t)__popcll(m0 & lt);
	r.c1 = s_base_conn + s_conn[wave] + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & lt);
	return r;
}

// ---------------------------------------------------------------------------------------------
// generate: pixel jitter, band pick, thin-lens ray (renderer.cpp:81-94, camera.h:148-187)
// ---------------------------------------------------------------------------------------------
// NaiveBDPT records (used by render_bdpt.inc)
constexpr int kBdptMax = 8;

// Vertices are stored by depth: vertex k of the eye (light) subpath of sample `slot` sits at
// [(k (+ kBdptMax)) * n_slots + slot], so that the walkers of one bounce, whose slots ascend, write next to each other
// and the connection kernel reads each depth as one coalesced stream (per-sample runs of 16 x 48 bytes put every
// 48-byte record on a line of its own).
struct BdptVertex { // 48 bytes
	float px, py, pz, nx, ny, nz, dx, dy, dz; // hit point, rec.normal, direction of the ray that arrived
	float throughput;                          // alphas[e] / betas[l]
	uint32_t object;
	float a;                                   // slot a (albedo) of the vertex's material at this hit and band
};

struct BdptSample { // 64 bytes
	uint64_t rng;
	float weight; // film weight of the camera sample
	uint32_t pixel;
	uint32_t band_k_l; // band | K << 16 | L << 24
	uint32_t last_eye_ray_hit;
	uint32_t sel_face;
	float pol[3];
	float light_pdf_area, light_dir_pdf_projected, light_value;
	float last_eye_vt[3]; // rec.vt of the eye subpath's last hit: what the reference evaluates the light's emission with (:558-560)
};

// Can the ray (o, d), t >= 0, touch the box (b0.xyz .. b1.xyz; b0.w = the largest |coordinate| of the box)?  Errs on the side
// of "yes": the box is widened by 1e-4 of the largest coordinate involved (fp32 slab arithmetic is off by ~1e-7 of it), and
// comparisons are written so that a NaN (0 * inf on an axis the ray does not move along) accepts.
__device__ __forceinline__ bool ray_may_reach(float4 b0, float4 b1, f3 o, f3 d)
{
	const float pad = 1e-4f * fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), b0.w)) + 1e-30f;
	const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
	const float x0 = (b0.x - pad - o.x) * ix, x1 = (b1.x + pad - o.x) * ix;
	const float y0 = (b0.y - pad - o.y) * iy, y1 = (b1.y + pad - o.y) * iy;
	const float z0 = (b0.z - pad - o.z) * iz, z1 = (b1.z + pad - o.z) * iz;
	const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
	const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
	return !(tn > tf * 1.00001f) && !(tf < 0.0f);
}

// BDPT: the pixel and the film weight go to the sample record, and the path state describes a subpath walker instead:
// alpha (beta), the pending factor alpha_ (weight), the sample slot (pixel)
// one camera sample: the ray and the path state it starts with
template <bool BDPT>
__device__ __forceinline__ void camera_sample(const DevCamera &cam, const DevBand *__restrict__ bands, FastDiv n_bands, int nx, FastDiv ny, int ns,
                                              int first_sample, FastDiv n_samples, uint64_t seed, float tmin, uint64_t idx, BdptSample *__restrict__ samples,
                                              rt_ray &r, PathState &st, const uint32_t *__restrict__ live = nullptr)
{
#pragma clang fp reciprocal(off) // inlined into two kernels that must produce the same ray for the same sample: see unit_rsq
	// a batch holds fewer than 2^31 samples (render_pass): 32-bit divisions, a fraction of the cost of 64-bit ones
	// The samples of one pixel are neighbours in the queue (index = pixel * n_samples + sample): a wave then adds to a few
	// neighbouring film pixels instead of 64 different ones (scattered float atomics cost one memory request per 64-byte
	// segment touched), and its camera rays start out coherent.
	const uint32_t idx32 = (uint32_t)idx;
	const uint32_t slot = idx32 / n_samples, sample_off = idx32 - slot * n_samples.d;
	const uint32_t pixel = live ? live[slot] : slot; // only the pixels that can see the scene are generated at all
	const int sample = first_sample + (int)sample_off;
	const int i = (int)(pixel / ny), j = (int)(pixel - (uint32_t)i * ny.d);
	uint64_t rng = rng_seed(seed, pixel, (uint64_t)sample);
	const float u = ((float)i + rng_f32(rng)) * __builtin_amdgcn_rcpf((float)nx);
	const float v = ((float)j + rng_f32(rng)) * __builtin_amdgcn_rcpf((float)ny.d);
	const uint32_t k = rng_below(rng, n_bands);
	const DevBand bd = bands[k];
	float l0, l1;
	do {
		l0 = 2.0f * rng_f32(rng) - 1.0f;
		l1 = 2.0f * rng_f32(rng) - 1.0f;
	} while (sqrtf(l0 * l0 + l1 * l1) > 1.0f);
	const f3 hor = ld3(cam.horizontal), ver = ld3(cam.vertical);
	const f3 isp = ld3(cam.llc_rel) + u * hor + v * ver;              // image_sensor_pos - origin
	const f3 lens = (l0 * cam.half_aperture) * hor + (l1 * cam.half_aperture) * ver; // random_point_on_lens - origin
	const float cos_theta = dot(unit_rsq(lens - isp), neg(ld3(cam.w)));
	f3 dir = unit_rsq(bd.neg_a_over_d * isp - lens);                  // target - random_point_on_lens
	if (bd.a_positive == 0.0f) dir = neg(dir);
	const f3 org = ld3(cam.origin) + lens;
	r.ox = org.x; r.oy = org.y; r.oz = org.z; r.tmin = tmin;
	r.dx = dir.x; r.dy = dir.y; r.dz = dir.z; r.tmax = __int_as_float(0x7f800000);
	st.rng = rng;
	st.beta = 1.0f;
	const float c2 = cos_theta * cos_theta;
	st.weight = cam.weight_scale * c2 * c2 / (float)ns; // renderer.cpp:113-116
	st.pixel = pixel;
	st.band_bounce = k;
	if (BDPT) {
		BdptSample s{};
		s.rng = st.rng;
		s.weight = st.weight;
		s.pixel = pixel;
		s.band_k_l = k;
		samples[idx] = s;
		st.weight = 1.0f;
		st.pixel = (uint32_t)idx;
	}
}

// rays and state point at the first free queue slot (paths carried over from the previous batch sit in front of it)
template <bool BDPT>
__global__ void __launch_bounds__(kBlock) k_generate(DevCamera cam, const DevBand *__restrict__ bands, FastDiv n_bands,
                                                     int nx, FastDiv ny, int ns, int first_sample, FastDiv n_samples, uint64_t n_paths,
                                                     uint64_t seed, float tmin, rt_ray *__restrict__ rays,
                                                     PathState *__restrict__ state, BdptSample *__restrict__ samples)
{
	const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (idx >= n_paths) return;
	rt_ray r;
	PathState st;
	camera_sample<BDPT>(cam, bands, n_bands, nx, ny, ns, first_sample, n_samples, seed, tmin, idx, samples, r, st);
	rays[idx] = r;
	state[idx] = st;
}

// The same, but camera rays that cannot touch the scene's bounding box (a frame wider than the scene: 40 % of the Cornell box at
// 16:9) are not queued: they would be traced to a miss and shaded to nothing.  Survivors are appended at the queue tail; rays and
// state point at the queue's start.  One atomic on the tail per workgroup of 512 threads x 4 samples (swept): a workgroup per 256 samples
// made 130 k same-address atomics per 1080p batch, 1.3 ms where the whole kernel takes 0.4.
// BDPT: a sample whose camera ray misses the scene has no eye vertex and contributes nothing (its light subpath would be walked for
// nothing): the survivors' sample records are compacted the same way, and the path state names the record's slot.
#ifndef GEN_BLOCK
#define GEN_BLOCK 512
#endif
#ifndef GEN_PER
#define GEN_PER 4
#endif
constexpr int kGenBlock = GEN_BLOCK, kGenPer = GEN_PER;
template <bool BDPT>
__global__ void __launch_bounds__(kGenBlock) k_generate_cull(DevCamera cam, const DevBand *__restrict__ bands, FastDiv n_bands,
                                                             int nx, FastDiv ny, int ns, int first_sample, FastDiv n_samples, uint64_t n_paths,
                                                             uint64_t seed, float tmin, rt_ray *__restrict__ rays,
                                                             PathState *__restrict__ state, BdptSample *__restrict__ samples, float4 box_lo,
                                                             float4 box_hi, unsigned long long *__restrict__ tails, const uint32_t *__restrict__ live)
{
	__shared__ uint32_t s_wave[kGenBlock / 64];
	__shared__ uint32_t s_base;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	rt_ray r[kGenPer];
	PathState st[kGenPer];
	unsigned long long kept[kGenPer];
	uint32_t mine = 0;
#pragma unroll
	for (int j = 0; j < kGenPer; j++) {
		const uint64_t idx = ((uint64_t)blockIdx.x * kGenPer + j) * kGenBlock + threadIdx.x;
		bool keep = false;
		if (idx < n_paths) {
			camera_sample<false>(cam, bands, n_bands, nx, ny, ns, first_sample, n_samples, seed, tmin, idx, nullptr, r[j], st[j], live);
			keep = ray_may_reach(box_lo, box_hi, F3(r[j].ox, r[j].oy, r[j].oz), F3(r[j].dx, r[j].dy, r[j].dz));
		}
		kept[j] = __ballot(keep);
		mine += (uint32_t)__popcll(kept[j]);
	}
	if (lane == 0) s_wave[wave] = mine; // the same for every lane of the wave
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int w = 0; w < kGenBlock / 64; w++) { const uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
		s_base = total ? (uint32_t)atomicAdd(tails, (unsigned long long)total) : 0u;
	}
	__syncthreads();
	uint32_t at = s_base + s_wave[wave];
#pragma unroll
	for (int j = 0; j < kGenPer; j++) {
		if ((kept[j] >> lane) & 1ull) {
			const uint32_t slot = at + (uint32_t)__popcll(kept[j] & lt);
			rays[slot] = r[j];
			if (BDPT) {
				BdptSample smp{};
				smp.rng = st[j].rng;
				smp.weight = st[j].weight;
				smp.pixel = st[j].pixel;
				smp.band_k_l = st[j].band_bounce;
				samples[slot] = smp;
				st[j].weight = 1.0f;
				st[j].pixel = slot;
			}
			state[slot] = st[j];
		}
		at += (uint32_t)__popcll(kept[j]);
	}
}

// Radiance of the environment for a ray that left the scene (renderer.cpp:175-191): Sphere::Hit on the dome of radius
// 100 around the origin (object.cpp:43-75), direction of the dome point -> texel -> RGBtoSpectrum(rgb * brightness)
// read at the band's bin (bins 0-9 blue, 10-19 green, 20-29 red, spectrum.cpp:219-232).  In fp64: it runs once per
// escaping path and the texel index is a truncation, which fp32 would move across texel borders.
// The reference's x can equal the width (it then reads the first texel of the next row) and, on the last row or for
// theta = pi, an address past the texture (undefined there); here such an address reads the last texel.
__device__ float environment_radiance(const DeviceShade &sh, const rt_ray &r, int get_bin)
{
#pragma clang fp reciprocal(off) // the texel index is a truncated quotient
	const double ox = r.ox, oy = r.oy, oz = r.oz, dx = r.dx, dy = r.dy, dz = r.dz;
	const double blen = sqrt(dx * dx + dy * dy + dz * dz), olen = sqrt(ox * ox + oy * oy + oz * oz);
	const double a = blen * blen, b_prime = ox * dx + oy * dy + oz * dz, c = olen * olen - 100.0 * 100.0;
	const double disc = b_prime * b_prime - a * c;
	if (disc < 0.0) return 0.0f;
	double t = (-b_prime - sqrt(disc)) / a;
	if (!(t >= 0.001 && t <= DBL_MAX)) {
		t = (-b_prime + sqrt(disc)) / a;
		if (!(t >= 0.001 && t <= DBL_MAX)) return 0.0f;
	}
	double px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
	const double l = sqrt(px * px + py * py + pz * pz);
	px /= l; py /= l; pz /= l;
	const double theta = acos(py);
	const double phi = atan2(px / sin(theta), pz / sin(theta)) + 3.14159265358979323846;
	const int x = sh.env_w - (int)(phi / (2.0 * 3.14159265358979323846) * sh.env_w);
	const int y = (int)(theta / 3.14159265358979323846 * sh.env_h);
	long idx = (long)x + (long)y * sh.env_w;
	idx = idx < 0 ? 0 : idx;
	idx = idx > (long)sh.env_w * sh.env_h - 1 ? (long)sh.env_w * sh.env_h - 1 : idx;
	const int channel = get_bin < RT_NBINS / 3 ? 2 : (get_bin < 2 * RT_NBINS / 3 ? 1 : 0);
	return (float)((double)sh.env[(long)sh.env_bpp * idx + channel] / 255.0 * (double)sh.env_brightness);
}

// ---------------------------------------------------------------------------------------------
// shade: one path vertex.  Emission, light pick and light point, BSDF samples, Russian roulette in
// the reference's order; appends connection rays and the continuation ray.
// ---------------------------------------------------------------------------------------------
// Can the ray (o, d), t >= 0, touch the box of light `sel`?  Used to drop connection rays whose contribution requires
// the closest hit to lie on that one polygon (renderer.cpp:421): a ray that misses the polygon's box cannot hit the
// polygon, whatever else it hits, so it is not traced.
__device__ __forceinline__ bool ray_may_reach_box(const DeviceShade &sh, uint32_t sel, f3 o, f3 d)
{
	return ray_may_reach(sh.light_box[2 * sel], sh.light_box[2 * sel + 1], o, d);
}

#ifdef SHADE_CLK
__device__ unsigned long long g_shade_clk[8];
#define SHADE_LAP(k) do { const long long n_ = clock64(); clk[k] += n_ - t0_; t0_ = n_; } while (0)
#else
#define SHADE_LAP(k) do { } while (0)
#endif
template <int ALG, bool GRAPHS>
__global__ void __launch_bounds__(kShadeBlock, SHADE_MIN_BLOCKS) k_shade(DeviceShade sh, const uint32_t *__restrict__ n_dev, const rt_ray *__restrict__ rays,
                                                  const PathState *__restrict__ state, const rt_hit *__restrict__ hits,
                                                  rt_ray *__restrict__ next_rays, PathState *__restrict__ next_state,
                                                  rt_ray *__restrict__ conn_rays, ConnRecU *__restrict__ conn,
                                                  Counters *__restrict__ counters, float *__restrict__ film)
{
	const uint32_t idx = blockIdx.x * kShadeBlock + threadIdx.x;
	// grids are sized from an upper bound of the queue length (the count of two bounces ago): most workgroups of a
	// late bounce lie wholly beyond the queue and leave before the slot reservation below (uniform per workgroup).
	// (Walking the queue in strides with a capped grid, as k_connect does, made this kernel 30 % slower.)
	if (blockIdx.x * kShadeBlock >= *n_dev) return;
#ifdef SHADE_CLK
	long long clk[6] = {0, 0, 0, 0, 0, 0}, t0_ = clock64();
#endif
	const bool live = idx < *n_dev;
	bool survive = false, want_c0 = false, want_c1 = false;
	rt_ray nr{}, cr0{}, cr1{};
	ConnRecU c0{}, c1{};
	PathState st{};
	if (live) {
		// state, hit and ray are requested together (nearly every queued ray hits something: the ray is rarely read for nothing),
		// then everything that hangs on them: the kernel spends half its time waiting for these reads
		st = state[idx];
		const rt_hit h = hits[idx];
		const rt_ray r = rays[idx];
		const uint32_t band = st.band_bounce & 0xFFFFu, bounce = st.band_bounce >> 16;
		// The light pick and the point on the light need nothing but the path's random stream, and they are the first draws of a
		// vertex (renderer.cpp:252-256 / 356-364): they are made here, on a copy of the stream that a surviving vertex adopts, so
		// that their reads (light record -> that triangle's vertices) travel together with the reads that hang on the hit (triangle,
		// material, band) instead of queueing up behind them: this kernel waits for dependent reads more than half of its time.
		uint64_t rng_l = st.rng;
		uint32_t lf = 0, lt = 0, sel = 0, l_obj = 0;
		float l_area = 0.0f;
		f3 pl = F3(0, 0, 0), l_fn = F3(0, 0, 0);
		TriShade ltri{};
		if (ALG != RT_ALG_NAIVE) {
			sel = rng_below(rng_l, sh.by_lights);
			const uint4 lrec = sh.light_rec[2 * sel], lcum = sh.light_rec[2 * sel + 1];
			lf = lrec.x;
			l_obj = lcum.w;
			l_area = __uint_as_float(lrec.w);
			// GetRandomPointOnPolygon (object.cpp:340-372)
			const float rr = rng_f32(rng_l);
			const uint32_t t0 = lrec.y, nt = lrec.z;
			lt = t0 + nt - 1; // the last cumulative area is 1: a triangle light needs no look-up at all
			if (nt > 1 && rr < __uint_as_float(lcum.x)) lt = t0;
			else if (nt > 2 && rr < __uint_as_float(lcum.y)) lt = t0 + 1;
			else if (nt > 3 && rr < __uint_as_float(lcum.z)) lt = t0 + 2;
			else for (uint32_t ti = 3; ti + 1 < nt; ti++)
				if (rr < sh.tri_cum[t0 + ti]) { lt = t0 + ti; break; }
			ltri = load_tri(sh.tri_shade, lt);
			float x = rng_f32(rng_l), y = rng_f32(rng_l);
			if (x + y >= 1.0f) { x = 1.0f - x; y = 1.0f - y; }
			pl = ltri.v0 + x * (ltri.v1 - ltri.v0) + y * (ltri.v2 - ltri.v0);
			l_fn = ltri.fn;
			if (!GRAPHS) asm volatile("" : : "v"(pl.x), "v"(pl.y), "v"(pl.z), "v"(l_fn.x), "v"(l_fn.y), "v"(l_fn.z)); // keeps the reads up here
		}
		const DevBand bd = sh.bands[band];
		if (ALG == RT_ALG_NAIVE && h.prim == RT_MISS && sh.env != nullptr) // renderer.cpp:175-193
			film_add(film, st.pixel, bd, st.beta * environment_radiance(sh, r, bd.get_bin) * st.weight);
		if (h.prim != RT_MISS) {
			const f3 o = F3(r.ox, r.oy, r.oz), d = F3(r.dx, r.dy, r.dz);
			const f3 p = o + h.t * d;
			const TriShade tri = load_tri(sh.tri_shade, h.tri);
			const f3 nrm = shading_normal(tri, p);
			const DevMaterial &m = sh.materials[h.object];
			const f3 uv = (GRAPHS && m.prog_len) ? hit_uv(sh, h.tri, tri, p) : F3(0, 0, 0); // rec.vt, only read by per-hit material graphs
			const float a_bin = slot_a<GRAPHS>(sh, m, uv, bd.get_bin);
			const float tmin = r.tmin;
			bool alive = true;
			if (ALG == RT_ALG_NAIVE) {
				film_add(film, st.pixel, bd, st.beta * emitted_at<GRAPHS>(sh, h.object, m, uv, bd, band) * st.weight); // renderer.cpp:197
			} else if (m.light_flag) { // renderer.cpp:242-251 / 346-355
				if (bounce != 0) alive = false;
				else film_add(film, st.pixel, bd, emitted_at<GRAPHS>(sh, h.object, m, uv, bd, band) * st.weight);
			}
			if (alive) {
				SHADE_LAP(0);
				const Onb uvw = onb_from_w(nrm);
				const f3 vo = w2l(uvw, neg(d));
				if (ALG != RT_ALG_NAIVE) {
					st.rng = rng_l; // the draws of the light pick above
					const float area = l_area;
					const f3 to_l = pl - p;
					const float dist = len(to_l);
					const f3 sd = F3(to_l.x / dist, to_l.y / dist, to_l.z / dist);
					const f3 vi = w2l(uvw, sd);
					const float bsdf = mat_bsdf(m, vi, vo, a_bin);
					const float lpa = 1.0f / area / (float)sh.n_lights;
					cr0.ox = p.x; cr0.oy = p.y; cr0.oz = p.z; cr0.tmin = tmin;
					cr0.dx = sd.x; cr0.dy = sd.y; cr0.dz = sd.z;
					c0.cos_surface = fabsf(dot(sd, nrm));
					c0.sel_face = lf;
					c0.pixel = st.pixel;
					if (ALG == RT_ALG_NEE) { // renderer.cpp:258-286
						cr0.tmax = dist - 0.0001f;
						const float G = fabsf(dot(sd, l_fn) * dot(sd, nrm) / dot(to_l, to_l));
						const DevMaterial &lm = sh.materials[l_obj]; // Emitted(shadow_ray, light_rec, light_rec.vt), :278-282
						const float light = emitted_at<GRAPHS>(sh, l_obj, lm, (GRAPHS && lm.prog_len) ? hit_uv(sh, lt, ltri, pl) : F3(0, 0, 0), bd, band);
						c0.c0 = bsdf * st.beta * light * G / lpa * st.weight;
						c0.c1 = dist;
						c0.band_kind = band | (2u << 16);
						want_c0 = c0.c0 != 0.0f; // a zero term needs no visibility query
					} else { // renderer.cpp:361-401
						cr0.tmax = dist + 0.001f;
						const float spdf = mat_pdf(m, vi, vo);
						c0.c0 = st.beta * bsdf / lpa * st.weight;
						c0.c1 = spdf / fabsf(dot(sd, nrm));
						c0.band_kind = band | (0u << 16);
						want_c0 = c0.c0 != 0.0f;
						SHADE_LAP(1);
						// BSDF-sampled direct light, renderer.cpp:404-433
						f3 gvi;
						float ratio, pdfv;
						if (mat_sample(m, nrm, uvw, vo, bd.get_bin, a_bin, st.rng, gvi, ratio, pdfv)) {
							const f3 sdir = l2w(uvw, gvi); // not normalised, :413
							cr1.ox = p.x; cr1.oy = p.y; cr1.oz = p.z; cr1.tmin = 0.001f;
							cr1.dx = sdir.x; cr1.dy = sdir.y; cr1.dz = sdir.z; cr1.tmax = __int_as_float(0x7f800000);
							c1 = c0;
							c1.c0 = st.beta * ratio * fabsf(gvi.z) * st.weight;
							c1.c1 = pdfv / fabsf(gvi.z);
							c1.cos_surface = fabsf(dot(unit(sdir), nrm));
							c1.band_kind = band | (1u << 16);
							// counts only if its closest hit is on the chosen light, and not at all with a zero factor
							want_c1 = !(sh.cull_connections & 4) || (c1.c0 != 0.0f && ray_may_reach_box(sh, sel, p, sdir));
						}
					}
				}
				SHADE_LAP(2);
				// continuation sample (renderer.cpp:199-208 / 290-300 / 440-450)
				f3 gvi;
				float ratio, pdfv;
				const bool respawn = mat_sample(m, nrm, uvw, vo, bd.get_bin, a_bin, st.rng, gvi, ratio, pdfv);
				if (ALG == RT_ALG_NAIVE) {
					if (respawn) {
						st.beta *= ratio * fabsf(gvi.z);
						if (!(rng_f32(st.rng) < 0.5f) && (st.beta != 0.0f || !(sh.cull_connections & 2))) { // :211-215; a zero throughput ends the path
							st.beta /= 0.5f;
							const f3 nd = l2w(uvw, gvi); // not re-normalised, :217
							nr.ox = p.x; nr.oy = p.y; nr.oz = p.z; nr.tmin = tmin;
							nr.dx = nd.x; nr.dy = nd.y; nr.dz = nd.z; nr.tmax = __int_as_float(0x7f800000);
							survive = true;
						}
					}
				} else {
					if (respawn) st.beta *= ratio * fabsf(gvi.z);
					const float prr = bounce > 6 ? 0.9f : 0.5f; // :304-310 / 454-460
					if (!(rng_f32(st.rng) < prr)) {
						st.beta /= (1.0f - prr);
						// a path whose throughput has become exactly zero (a surface that reflects nothing in this band) can add
						// nothing any more: it is not continued
						if (respawn && (st.beta != 0.0f || !(sh.cull_connections & 2))) {
							const f3 nd = unit(l2w(uvw, gvi));
							nr.ox = p.x; nr.oy = p.y; nr.oz = p.z; nr.tmin = tmin;
							nr.dx = nd.x; nr.dy = nd.y; nr.dz = nd.z; nr.tmax = __int_as_float(0x7f800000);
							st.band_bounce = band | ((bounce + 1) << 16);
							survive = true;
						}
					}
				}
			}
		}
	}
	SHADE_LAP(3);
	const Slots sl = reserve_slots(&counters->tails, survive, want_c0, want_c1);
	SHADE_LAP(4);
	if (survive) { next_rays[sl.next] = nr; next_state[sl.next] = st; }
	if (ALG != RT_ALG_NAIVE) {
		if (want_c0) { conn_rays[sl.c0] = cr0; conn[sl.c0] = c0; }
		if (want_c1) { conn_rays[sl.c1] = cr1; conn[sl.c1] = c1; }
	}
#ifdef SHADE_CLK
	SHADE_LAP(5);
	if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) atomicAdd(&g_shade_clk[k], (unsigned long long)clk[k]);
#endif
}

// exact polygon test of one face, as light_objects[sel]->Hit(shadow_ray, ...) at renderer.cpp:278
__device__ bool face_hit_exact(const DeviceGeom &g, const DeviceShade &sh, uint32_t face, const rt_ray &r, double tmin, double tmax)
{
#pragma clang fp contract(off)
#pragma clang fp reciprocal(off) // the same operations as the traversal kernel's test, bit for bit
	const uint32_t t0 = sh.face_tri_begin[face], t1 = sh.face_tri_begin[face + 1];
	const double ox = r.ox, oy = r.oy, oz = r.oz, dx = r.dx, dy = r.dy, dz = r.dz;
	for (uint32_t t = t0; t < t1; t++) {
		const uint32_t li = sh.tri_leaf[t];
		const double2 *rec = g.tri + (size_t)g.tri_stride * li;
		const double2 a0 = rec[0], a1 = rec[1], a2 = rec[2], b0 = rec[3], b1 = rec[4], b2 = rec[5];
		const double v0x = a0.x, v0y = a0.y, v0z = a1.x, nx = a1.y, ny = a2.x, nz = a2.y;
		const double v1x = b0.x, v1y = b0.y, v1z = b1.x, v2x = b1.y, v2y = b2.x, v2z = b2.y;
		if (g.watertight) { // the watertight test of trace.hip (watertight_t), same operations in the same order
			const double ax_ = fabs(dx), ay_ = fabs(dy), az_ = fabs(dz);
			const int kz = ax_ > ay_ ? (ax_ > az_ ? 0 : 2) : (ay_ > az_ ? 1 : 2);
			const double dkz = kz == 0 ? dx : (kz == 1 ? dy : dz);
			int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
			if (dkz < 0.0) { const int tmp = kx; kx = ky; ky = tmp; }
#define PICK(k, x, y, z) ((k) == 0 ? (x) : ((k) == 1 ? (y) : (z)))
			const double sx = PICK(kx, dx, dy, dz) / dkz, sy = PICK(ky, dx, dy, dz) / dkz, sz = 1.0 / dkz;
			const double a0_ = v0x - ox, a1_ = v0y - oy, a2_ = v0z - oz, b0_ = v1x - ox, b1_ = v1y - oy, b2_ = v1z - oz;
			const double c0_ = v2x - ox, c1_ = v2y - oy, c2_ = v2z - oz;
			const double akz = PICK(kz, a0_, a1_, a2_), bkz = PICK(kz, b0_, b1_, b2_), ckz = PICK(kz, c0_, c1_, c2_);
			const double Ax = PICK(kx, a0_, a1_, a2_) - sx * akz, Ay = PICK(ky, a0_, a1_, a2_) - sy * akz;
			const double Bx = PICK(kx, b0_, b1_, b2_) - sx * bkz, By = PICK(ky, b0_, b1_, b2_) - sy * bkz;
			const double Cx = PICK(kx, c0_, c1_, c2_) - sx * ckz, Cy = PICK(ky, c0_, c1_, c2_) - sy * ckz;
#undef PICK
			const double U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
			if ((U < 0.0 || V < 0.0 || W < 0.0) && (U > 0.0 || V > 0.0 || W > 0.0)) continue;
			const double det = U + V + W;
			if (det == 0.0) continue;
			const double tw = (U * (sz * akz) + V * (sz * bkz) + W * (sz * ckz)) / det;
			if (tw >= tmin && tw <= tmax) return true;
			continue;
		}
		const double num = (v0x - ox) * nx + (v0y - oy) * ny + (v0z - oz) * nz;
		const double den = dx * nx + dy * ny + dz * nz;
		const double tt = num / den;
		if (!(tt >= tmin && tt <= tmax)) continue;
		const double px = ox + tt * dx, py = oy + tt * dy, pz = oz + tt * dz;
		double ax = v1x - v0x, ay = v1y - v0y, az = v1z - v0z, bx = px - v1x, by = py - v1y, bz = pz - v1z;
		const double r0x = ay * bz - az * by, r0y = -(ax * bz - az * bx), r0z = ax * by - ay * bx;
		ax = v2x - v1x; ay = v2y - v1y; az = v2z - v1z; bx = px - v2x; by = py - v2y; bz = pz - v2z;
		const double r1x = ay * bz - az * by, r1y = -(ax * bz - az * bx), r1z = ax * by - ay * bx;
		ax = v0x - v2x; ay = v0y - v2y; az = v0z - v2z; bx = px - v0x; by = py - v0y; bz = pz - v0z;
		const double r2x = ay * bz - az * by, r2y = -(ax * bz - az * bx), r2z = ax * by - ay * bx;
		if (r0x * r1x + r0y * r1y + r0z * r1z > 0.0 && r1x * r2x + r1y * r2y + r1z * r2z > 0.0) return true;
	}
	return false;
}

// ---------------------------------------------------------------------------------------------
// connect: turns visibility results into film contributions
// ---------------------------------------------------------------------------------------------
template <int ALG, bool GRAPHS>
__global__ void __launch_bounds__(kBlock) k_connect(DeviceGeom g, DeviceShade sh, const uint32_t *__restrict__ n_dev, const rt_ray *__restrict__ conn_rays,
                                                    const ConnRecU *__restrict__ conn, const rt_hit *__restrict__ conn_hits,
                                                    const uint8_t *__restrict__ conn_any, float *__restrict__ film)
{
	// The host only knows an upper bound of the queue length (the count of two bounces ago, or nothing better than the batch size
	// right after the camera rays were culled on the device): a grid sized by it launched up to ten times more workgroups than
	// there was work for, 30-60 us per kernel.  The grid is capped (stream_grid) and the queue is walked in strides.
	const uint32_t n_items = *n_dev;
	for (uint32_t idx = blockIdx.x * kBlock + threadIdx.x; idx < n_items; idx += gridDim.x * kBlock) {
	const ConnRecU c = conn[idx];
	const DevBand bd = sh.bands[c.band_kind & 0xFFFFu];
	const uint32_t kind = c.band_kind >> 16;
	if (ALG == RT_ALG_NEE) {
		if (conn_any[idx]) continue; // occluded, renderer.cpp:263
		const rt_ray r = conn_rays[idx];
		if (face_hit_exact(g, sh, c.sel_face, r, 0.0001, (double)c.c1 + 0.0001)) film_add(film, c.pixel, bd, c.c0);
		continue;
	}
	const rt_hit h = conn_hits[idx];
	// Both records come whole and at once (left alone the compiler reads the two words of the test below first and the rest behind
	// it), and everything that hangs on them goes out as one batch: ray, triangle, band, emission, area.  A thread has one item in
	// flight, so the kernel's rate is the number of resident threads over the length of this chain of dependent reads.
	asm volatile("" : : "v"(c.c0), "v"(c.c1), "v"(c.cos_surface), "v"(c.pixel), "v"(c.band_kind), "v"(h.t), "v"(h.tri), "v"(h.object));
	if (h.prim == RT_MISS || h.prim != c.sel_face) continue; // renderer.cpp:373-376 / 419-421
	const rt_ray r = conn_rays[idx];
	const TriShade tri = load_tri(sh.tri_shade, h.tri);
	const float l_face_area = sh.face_area[c.sel_face];
	const DevMaterial &lm = sh.materials[h.object];
	const bool programmed = GRAPHS && lm.prog_len;
	float light = programmed ? 0.0f : sh.emit[(size_t)h.object * sh.n_bands + (c.band_kind & 0xFFFFu)];
	asm volatile("" : : "v"(l_face_area), "v"(light), "v"(bd.min_i), "v"(bd.max_i));
	const f3 o = F3(r.ox, r.oy, r.oz), d = F3(r.dx, r.dy, r.dz);
	const f3 pl = o + h.t * d;
	const f3 ln = shading_normal(tri, pl);
	// GeometryTerm(light_rec, rec), renderer.cpp:784-789
	const f3 v = o - pl;
	const f3 dir = unit(v);
	const float G = fabsf(dot(dir, ln) * c.cos_surface / dot(v, v));
	if (programmed) light = mat_emitted<GRAPHS>(sh, lm, hit_uv(sh, h.tri, tri, pl), bd);
	const float lpa = 1.0f / l_face_area / (float)sh.n_lights;
	float add;
	if (kind == 0) add = power_heuristic2(lpa, c.c1 * G) * (c.c0 * G) * light;      // renderer.cpp:390-399
	else add = power_heuristic2(G * c.c1, lpa) * c.c0 * light;                      // renderer.cpp:422-429
	film_add(film, c.pixel, bd, add);
	}
}

#ifdef BDPT_COUNT
__device__ unsigned long long g_bdpt_kind[8];
#endif
#include "render_bdpt.inc"

// ---------------------------------------------------------------------------------------------
// tonemap (renderer.cpp:124-143, spectrum.cpp:164-199)
// ---------------------------------------------------------------------------------------------
__constant__ float c_cie[3][RT_NBINS];

__global__ void __launch_bounds__(kBlock) k_tonemap(const float *__restrict__ film, uint64_t npix, float scale, float *__restrict__ out)
{
	const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (idx >= npix) return;
	float X = 0.0f, Y = 0.0f, Z = 0.0f;
	for (int i = 0; i < RT_NBINS; i++) {
		const float s = film[idx * RT_NBINS + i] * scale;
		X += s * c_cie[0][i] * 10.0f;
		Y += s * c_cie[1][i] * 10.0f;
		Z += s * c_cie[2][i] * 10.0f;
	}
	float rgb[3];
	rgb[0] = 3.240479f * X + (-1.537150f) * Y + (-0.498535f) * Z;
	rgb[1] = (-0.969256f) * X + 1.875991f * Y + 0.041556f * Z;
	rgb[2] = 0.055648f * X + (-0.204043f) * Y + 1.057311f * Z;
	for (int c = 0; c < 3; c++) {
		float v = rgb[c];
		v = v >= 0.0f ? fminf(powf(v, 1.0f / 2.2f), 1.0f) : 0.0f;
		out[idx * 3 + c] = v;
	}
}

// CIE 1931 colour matching functions at 405..695 nm (CVRL tables, as reference spectrum.cpp:5-102)
const float h_cie[3][RT_NBINS] = {
	{4.742986E-02f, 1.446214E-01f, 2.488523E-01f, 3.227087E-01f, 3.418483E-01f, 2.826646E-01f, 2.219781E-01f, 1.29192E-01f,
	 4.600865E-02f, 7.097731E-03f, 3.649178E-03f, 4.315171E-02f, 1.268468E-01f, 2.405015E-01f, 3.804244E-01f, 5.280233E-01f,
	 7.016774E-01f, 8.853376E-01f, 1.051011E+00f, 1.14362E+00f, 1.134757E+00f, 1.007344E+00f, 8.135565E-01f, 5.75541E-01f,
	 3.844986E-01f, 2.277792E-01f, 1.263808E-01f, 6.63996E-02f, 3.292138E-02f, 1.575417E-02f},
	{4.971717E-03f, 1.429377E-02f, 2.612106E-02f, 4.15794E-02f, 5.743393E-02f, 7.238339E-02f, 1.060145E-01f, 1.535066E-01f,
	 2.064828E-01f, 2.85068E-01f, 4.277595E-01f, 6.206256E-01f, 7.946448E-01f, 9.071347E-01f, 9.814106E-01f, 9.994608E-01f,
	 9.902549E-01f, 9.424569E-01f, 8.587203E-01f, 7.544785E-01f, 6.270066E-01f, 4.89595E-01f, 3.609245E-01f, 2.416902E-01f,
	 1.547397E-01f, 8.979594E-02f, 4.899699E-02f, 2.554223E-02f, 1.261573E-02f, 6.027677E-03f},
	{2.369246E-01f, 7.378822E-01f, 1.305008E+00f, 1.74828E+00f, 1.918437E+00f, 1.664439E+00f, 1.42844E+00f, 9.991789E-01f,
	 5.617313E-01f, 3.105939E-01f, 1.720018E-01f, 8.283548E-02f, 3.751912E-02f, 1.566174E-02f, 6.131421E-03f, 2.327186E-03f,
	 8.822531E-04f, 3.386739E-04f, 1.335031E-04f, 5.460706E-05f, 2.334738E-05f, 1.048387E-05f, 0, 0, 0, 0, 0, 0, 0, 0}};

#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return e_; } \
	} while (0)

template <class T>
hipError_t upload(T *&dst, const std::vector<T> &src, hipStream_t stream)
{
	if (dst) { (void)hipFree(dst); dst = nullptr; }
	size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
	hipError_t e = hipMalloc((void **)&dst, bytes);
	if (e != hipSuccess) return e;
	if (!src.empty()) e = hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, stream);
	if (e != hipSuccess) return e;
	return hipStreamSynchronize(stream);
}

Impl *impl_of(RenderState &rs)
{
	if (!rs.impl) rs.impl = new Impl();
	return static_cast<Impl *>(rs.impl);
}

DeviceShade make_shade(const Impl &im)
{
	DeviceShade sh;
	sh.tri_shade = im.tri_shade;
	sh.tri_face = im.tri_face;
	sh.face_object = im.face_object;
	sh.face_tri_begin = im.face_tri_begin;
	sh.face_area = im.face_area;
	sh.tri_cum = im.tri_cum;
	sh.tri_leaf = im.tri_leaf;
	sh.materials = im.materials;
	sh.light_faces = im.light_faces;
	sh.light_rec = im.light_rec;
	sh.light_box = im.light_box;
	sh.cull_connections = im.cull_connections ? im.cull_mask : 0;
	sh.bands = im.bands;
	sh.emit = im.emit;
	sh.n_lights = im.n_lights;
	sh.by_lights = fast_div(std::max(im.n_lights, 1u));
	sh.n_bands = im.n_bands;
	sh.tri_uv = im.tri_uv;
	sh.g_code = im.g_code;
	sh.g_spectra = im.g_spectra;
	sh.g_vectors = im.g_vectors;
	sh.g_scalars = im.g_scalars;
	sh.g_tex = im.g_tex;
	sh.g_texels = im.g_texels;
	sh.env = im.env;
	sh.env_w = im.env_w;
	sh.env_h = im.env_h;
	sh.env_bpp = im.env_bpp;
	sh.env_brightness = im.env_brightness;
	return sh;
}

void free_queues(Impl &im)
{
	for (int k = 0; k < 2; k++) {
		if (im.rays[k]) (void)hipFree(im.rays[k]);
		if (im.state[k]) (void)hipFree(im.state[k]);
		im.rays[k] = nullptr;
		im.state[k] = nullptr;
	}
	if (im.hits) (void)hipFree(im.hits);
	if (im.conn_rays) (void)hipFree(im.conn_rays);
	if (im.conn) (void)hipFree(im.conn);
	if (im.conn_hits) (void)hipFree(im.conn_hits);
	if (im.conn_any) (void)hipFree(im.conn_any);
	im.hits = nullptr; im.conn_rays = nullptr; im.conn = nullptr; im.conn_hits = nullptr; im.conn_any = nullptr;
	im.capacity = 0;
	im.conn_capacity = 0;
	if (im.bin_rays) (void)hipFree(im.bin_rays);
	if (im.bin_perm) (void)hipFree(im.bin_perm);
	if (im.bin_hits) (void)hipFree(im.bin_hits);
	if (im.bin_counters) (void)hipFree(im.bin_counters);
	im.bin_rays = nullptr; im.bin_perm = nullptr; im.bin_hits = nullptr; im.bin_counters = nullptr;
	im.bin_capacity = 0;
}

// buffers of the ray ordering, for batches of up to `cap` rays (connection queues hold two rays per path)
static hipError_t ensure_bins(Impl &im, uint64_t cap, std::string &err)
{
	if (cap <= im.bin_capacity) return hipSuccess;
	if (im.bin_rays) (void)hipFree(im.bin_rays);
	if (im.bin_perm) (void)hipFree(im.bin_perm);
	if (im.bin_hits) (void)hipFree(im.bin_hits);
	im.bin_rays = nullptr; im.bin_perm = nullptr; im.bin_hits = nullptr;
	im.bin_capacity = 0;
	HIP_TRY(hipMalloc((void **)&im.bin_rays, cap * sizeof(rt_ray)));
	HIP_TRY(hipMalloc((void **)&im.bin_perm, cap * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **)&im.bin_hits, cap * sizeof(rt_hit)));
	if (!im.bin_counters) HIP_TRY(hipMalloc((void **)&im.bin_counters, 2 * kRayBins * sizeof(uint32_t)));
	im.bin_capacity = cap;
	return hipSuccess;
}

// Closest hits of a queue of an integrator's rays through the ordering by origin cell that rt_trace_closest applies to large
// unordered batches in scenes of 2M triangles and more.  Off unless MI355RT_RAY_BINS=1 asks for it: an integrator's later bounces
// are not unordered.  Compaction keeps the queues in pixel order, so neighbours in a queue start on neighbouring surface points, a
// finer neighbourhood than one of 64 cells; measured on the Cornell box with an 8M-triangle object (1920 x 1080, MIS), ordering the
// path and connection rays from the second bounce on: 3507 -> 2618 spp/s, 11.8 -> 7.8 Grays/s in the kernel (and the hall's render
// rays as one batch: 4.58 -> 4.00 Grays/s).  Kept as a switch so that the figure can be reproduced.
static bool bins_pay(const DeviceGeom &geom, uint64_t bound)
{
	(void)geom;
	if (bound == 0 || bound >= (1ull << 32)) return false;
	const char *e = std::getenv("MI355RT_RAY_BINS");
	return e && e[0] == '1';
}
static hipError_t trace_ordered(Impl &im, const DeviceGeom &geom, const rt_ray *rays, uint64_t bound, rt_hit *hits, const uint32_t *n_dev, hipStream_t stream, std::string &err)
{
	HIP_TRY(ensure_bins(im, std::max<uint64_t>(bound, im.conn_capacity), err));
	DeviceGeom ordered = geom;
	ordered.ordered_rays = 1;
	HIP_TRY(bin_rays(geom.nodes, rays, (uint32_t)bound, im.bin_rays, im.bin_perm, im.bin_counters, stream, n_dev));
	HIP_TRY(launch_trace_closest(ordered, im.bin_rays, bound, im.bin_hits, nullptr, stream, n_dev));
	HIP_TRY(unbin_results(im.bin_perm, (uint32_t)bound, im.bin_hits, nullptr, hits, nullptr, stream, n_dev));
	return hipSuccess;
}

} // namespace

// fan triangle -> leaf order is filled by the caller through this hook (geometry upload knows the BVH)
hipError_t render_upload_scene(RenderState &rs, const HostScene &host, const std::vector<uint32_t> &tri_leaf,
                               const uint32_t *d_tri_face, const uint32_t *d_face_object, PinnedPieces &pinned, hipStream_t stream, std::string &err)
{
	Impl &im = *impl_of(rs);
	if (im.tri_shade) { (void)hipFree(im.tri_shade); im.tri_shade = nullptr; }
	HIP_TRY(hipMalloc((void **)&im.tri_shade, 96 * std::max<uint64_t>(host.n_tris, 1)));
	HIP_TRY(upload_generated(pinned, im.tri_shade, host.n_tris, 96, stream, [&](size_t t_begin, size_t t_end, char *out) {
		for (size_t t = t_begin; t < t_end; t++) {
			const uint32_t f = host.tri_face[t];
			const uint64_t b = host.face_begin[f], k = t - host.tri_begin[f];
			const double *v0 = &host.pv[3 * b], *v1 = &host.pv[3 * (b + k + 1)], *v2 = &host.pv[3 * (b + k + 2)];
			const double *n0 = &host.pn[3 * b], *n1 = &host.pn[3 * (b + k + 1)], *n2 = &host.pn[3 * (b + k + 2)];
			const double *fn = &host.face_normal[3 * f];
			// tri_area of HitTriangle (object.cpp:249): |cross(v1 - v0, v2 - v1)|
			double ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2], bx = v2[0] - v1[0], by = v2[1] - v1[1], bz = v2[2] - v1[2];
			double cx = ay * bz - az * by, cy = -(ax * bz - az * bx), cz = ax * by - ay * bx;
			double area2 = std::sqrt(cx * cx + cy * cy + cz * cz);
			float4 *o = reinterpret_cast<float4 *>(out) + 6 * (t - t_begin);
			o[0] = make_float4((float)v0[0], (float)v0[1], (float)v0[2], (float)v1[0]);
			o[1] = make_float4((float)v1[1], (float)v1[2], (float)v2[0], (float)v2[1]);
			o[2] = make_float4((float)v2[2], (float)n0[0], (float)n0[1], (float)n0[2]);
			o[3] = make_float4((float)n1[0], (float)n1[1], (float)n1[2], (float)n2[0]);
			o[4] = make_float4((float)n2[1], (float)n2[2], (float)fn[0], (float)fn[1]);
			o[5] = make_float4((float)fn[2], (float)area2, 0.0f, 0.0f);
		}
	}));
	{ // the scene's bounding box for k_generate's camera-ray test
		double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX}, big = 0.0;
		std::mutex merge;
		parallel_for(host.pv.size() / 3, 1 << 16, [&](uint64_t c_begin, uint64_t c_end) {
			double plo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, phi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX}, pbig = 0.0;
			for (uint64_t c = 3 * c_begin; c < 3 * c_end; c += 3)
				for (int a = 0; a < 3; a++) {
					plo[a] = std::min(plo[a], host.pv[c + a]);
					phi[a] = std::max(phi[a], host.pv[c + a]);
					pbig = std::max(pbig, std::fabs(host.pv[c + a]));
				}
			std::lock_guard<std::mutex> hold(merge);
			for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], plo[a]); hi[a] = std::max(hi[a], phi[a]); }
			big = std::max(big, pbig);
		});
		im.scene_lo = make_float4(std::nextafter((float)lo[0], -FLT_MAX), std::nextafter((float)lo[1], -FLT_MAX), std::nextafter((float)lo[2], -FLT_MAX),
		                          std::nextafter((float)big, FLT_MAX));
		im.scene_hi = make_float4(std::nextafter((float)hi[0], FLT_MAX), std::nextafter((float)hi[1], FLT_MAX), std::nextafter((float)hi[2], FLT_MAX), 0.0f);
	}
	std::vector<uint32_t> ftb(host.n_faces + 1);
	for (uint64_t f = 0; f <= host.n_faces; f++) ftb[f] = (uint32_t)host.tri_begin[f];
	std::vector<float> fa(host.n_faces), tc(host.n_tris);
	for (uint64_t f = 0; f < host.n_faces; f++) fa[f] = (float)host.polygon_area[f];
	for (uint64_t t = 0; t < host.n_tris; t++) tc[t] = (float)host.tri_cum[t];
	// texture coordinates of the fan triangles' corners, only when the mesh has any (read by per-hit material graphs)
	bool any_vt = false;
	for (double x : host.pvt) if (x != 0.0) { any_vt = true; break; }
	if (im.tri_uv) { (void)hipFree(im.tri_uv); im.tri_uv = nullptr; }
	if (any_vt) {
		std::vector<float> uv(9 * host.n_tris);
		for (uint64_t t = 0; t < host.n_tris; t++) {
			const uint32_t f = host.tri_face[t];
			const uint64_t b = host.face_begin[f], k = t - host.tri_begin[f];
			const uint64_t c[3] = {b, b + k + 1, b + k + 2};
			for (int i = 0; i < 3; i++)
				for (int a = 0; a < 3; a++) uv[9 * t + 3 * i + a] = (float)host.pvt[3 * c[i] + a];
		}
		HIP_TRY(upload(im.tri_uv, uv, stream));
	}
	HIP_TRY(upload(im.face_tri_begin, ftb, stream));
	HIP_TRY(upload(im.face_area, fa, stream));
	HIP_TRY(upload(im.tri_cum, tc, stream));
	HIP_TRY(upload(im.tri_leaf, tri_leaf, stream));
	im.tri_face = d_tri_face;
	im.face_object = d_face_object;
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_cie), h_cie, sizeof(h_cie)));
	return hipSuccess;
}

static void free_graphs(Impl &im)
{
	void *ptrs[] = {im.g_code, im.g_spectra, im.g_vectors, im.g_scalars, im.g_tex, im.g_texels};
	for (void *p : ptrs) if (p) (void)hipFree(p);
	im.g_code = nullptr; im.g_spectra = im.g_vectors = im.g_scalars = nullptr; im.g_tex = nullptr; im.g_texels = nullptr;
}

// programs were validated by the caller (api.cpp): ops, table indices, stack depth and result type
hipError_t render_upload_material_graphs(RenderState &rs, const rt_material_graphs *g, hipStream_t stream, std::string &err)
{
	Impl &im = *impl_of(rs);
	HIP_TRY(hipStreamSynchronize(stream));
	free_graphs(im);
	std::vector<DevMaterial> dm = im.host_materials;
	for (auto &m : dm) m.prog_begin = m.prog_len = 0;
	if (g && g->n_programs) {
		for (uint32_t i = 0; i < g->n_programs; i++) {
			dm[g->programs[i].object].prog_begin = g->programs[i].code_begin;
			dm[g->programs[i].object].prog_len = g->programs[i].code_len;
		}
		std::vector<uint2> code(g->n_code);
		for (uint32_t i = 0; i < g->n_code; i++) code[i] = make_uint2(g->code[2 * i], g->code[2 * i + 1]);
		std::vector<float> spectra((size_t)g->n_spectra * RT_NBINS), vectors((size_t)g->n_vectors * 3), scalars(g->n_scalars);
		for (size_t i = 0; i < spectra.size(); i++) spectra[i] = (float)g->spectra[i];
		for (size_t i = 0; i < vectors.size(); i++) vectors[i] = (float)g->vectors[i];
		for (size_t i = 0; i < scalars.size(); i++) scalars[i] = (float)g->scalars[i];
		std::vector<int4> tex(g->n_textures);
		std::vector<uint8_t> texels;
		for (uint32_t i = 0; i < g->n_textures; i++) {
			const rt_texture &t = g->textures[i];
			tex[i] = make_int4((int)texels.size(), t.width, t.height, t.bpp);
			texels.insert(texels.end(), t.texels, t.texels + (size_t)t.width * t.height * t.bpp);
		}
		HIP_TRY(upload(im.g_code, code, stream));
		HIP_TRY(upload(im.g_spectra, spectra, stream));
		HIP_TRY(upload(im.g_vectors, vectors, stream));
		HIP_TRY(upload(im.g_scalars, scalars, stream));
		HIP_TRY(upload(im.g_tex, tex, stream));
		HIP_TRY(upload(im.g_texels, texels, stream));
	}
	HIP_TRY(upload(im.materials, dm, stream));
	im.host_materials = dm;
	im.emit_dirty = true;
	return hipSuccess;
}

hipError_t render_upload_materials(RenderState &rs, const HostScene &host, const std::vector<rt_material> &mats,
                                   const std::vector<uint32_t> &light_faces, hipStream_t stream, std::string &err)
{
	Impl &im = *impl_of(rs);
	std::vector<DevMaterial> dm(mats.size());
	for (size_t i = 0; i < mats.size(); i++) {
		dm[i].type = mats[i].type;
		dm[i].light_flag = mats[i].light_flag;
		for (int k = 0; k < RT_NBINS; k++) { dm[i].a[k] = (float)mats[i].a[k]; dm[i].b[k] = (float)mats[i].b[k]; }
		dm[i].prog_begin = dm[i].prog_len = 0;
		dm[i].roughness = (float)mats[i].roughness;
	}
	HIP_TRY(upload(im.materials, dm, stream));
	im.host_materials = dm;
	im.emit_dirty = true;
	free_graphs(im); // new records: programs have to be attached again
	HIP_TRY(upload(im.light_faces, light_faces, stream));
	im.n_lights = (uint32_t)light_faces.size();
	{
		std::vector<uint4> recs(2 * light_faces.size());
		for (size_t l = 0; l < light_faces.size(); l++) {
			const uint32_t f = light_faces[l];
			const uint64_t t0 = host.tri_begin[f], nt = host.tri_begin[f + 1] - t0;
			const float area = (float)host.polygon_area[f];
			uint32_t bits, cum[3] = {0, 0, 0};
			std::memcpy(&bits, &area, sizeof(bits));
			for (uint64_t k = 0; k < 3 && k < nt; k++) {
				const float c = (float)host.tri_cum[t0 + k]; // the same conversion as the device's tri_cum table
				std::memcpy(&cum[k], &c, sizeof(float));
			}
			recs[2 * l] = make_uint4(f, (uint32_t)t0, (uint32_t)nt, bits);
			recs[2 * l + 1] = make_uint4(cum[0], cum[1], cum[2], host.face_object[f]);
		}
		HIP_TRY(upload(im.light_rec, recs, stream));
	}
	// bounding boxes of the light polygons, rounded outwards to float (see ray_may_reach_box)
	std::vector<float4> boxes(2 * light_faces.size());
	for (size_t l = 0; l < light_faces.size(); l++) {
		const uint32_t f = light_faces[l];
		double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX}, big = 0.0;
		// everything the reference's test can accept as a hit on this polygon: for a polygon that is not planar that is more than
		// the box of its vertices (accept_region.h)
		for (uint64_t t = host.tri_begin[f]; t < host.tri_begin[f + 1]; t++) {
			const double *v0, *v1, *v2;
			host.tri_vertices(t, v0, v1, v2);
			double rlo[3], rhi[3];
			accept_region_box(v0, v1, v2, &host.face_normal[3 * (size_t)f], rlo, rhi);
			for (int a = 0; a < 3; a++) {
				lo[a] = std::min(lo[a], rlo[a]);
				hi[a] = std::max(hi[a], rhi[a]);
				big = std::max(big, std::max(std::fabs(rlo[a]), std::fabs(rhi[a])));
			}
		}
		float flo[3], fhi[3];
		for (int a = 0; a < 3; a++) {
			flo[a] = std::nextafter((float)lo[a], -FLT_MAX);
			fhi[a] = std::nextafter((float)hi[a], FLT_MAX);
		}
		boxes[2 * l] = make_float4(flo[0], flo[1], flo[2], std::nextafter((float)big, FLT_MAX));
		boxes[2 * l + 1] = make_float4(fhi[0], fhi[1], fhi[2], 0.0f);
	}
	HIP_TRY(upload(im.light_box, boxes, stream));
	return hipSuccess;
}

// The pixels some camera ray of which may touch the scene's bounding box, in pixel order.  For one pixel, every quantity of
// camera_sample is an interval: u, v span the pixel (the jitter), the lens offset spans the whole lens square, and the band
// picks one of the lens's image distances; interval arithmetic over the slab test then either proves that no such ray reaches
// the box, or the pixel stays.  Intervals and the box are widened far beyond the kernel's fp32 rounding (1e-5 relative
// against ~1e-7), and a pixel that stays is still tested ray by ray on the device, so the list only has to err towards keeping.
static void live_pixel_list(const DevCamera &dc, const std::vector<DevBand> &bands, int nx, int ny, float4 box_lo, float4 box_hi, std::vector<uint32_t> &out)
{
	struct Iv { double lo, hi; };
	auto scale = [](Iv a, double k) { return k >= 0.0 ? Iv{a.lo * k, a.hi * k} : Iv{a.hi * k, a.lo * k}; };
	auto widen = [](Iv a, double by) { return Iv{a.lo - by, a.hi + by}; };
	std::vector<std::pair<double, bool>> variants; // distinct (image distance ratio, direction sign) pairs over the bands
	for (const DevBand &b : bands) {
		std::pair<double, bool> v((double)b.neg_a_over_d, b.a_positive != 0.0f);
		bool seen = false;
		for (auto &w : variants) seen = seen || w == v;
		if (!seen) variants.push_back(v);
	}
	const double inf = std::numeric_limits<double>::infinity();
	const double big = std::max({(double)box_lo.w, std::fabs((double)dc.origin[0]), std::fabs((double)dc.origin[1]), std::fabs((double)dc.origin[2])});
	const double pad = 4e-4 * big + 1e-30;
	const double blo[3] = {box_lo.x - pad, box_lo.y - pad, box_lo.z - pad}, bhi[3] = {box_hi.x + pad, box_hi.y + pad, box_hi.z + pad};
	double lens_r[3]; // |lens offset| per axis is at most this
	for (int c = 0; c < 3; c++) lens_r[c] = 1.00001 * (double)dc.half_aperture * (std::fabs((double)dc.horizontal[c]) + std::fabs((double)dc.vertical[c]));
	// Decided per tile of 8 x 8 pixels, with u and v spanning the tile: a pixel-by-pixel pass over a 1080p frame took 50 ms of
	// host time in every rt_render_begin, a quarter of what 1024 spp take to render; the tile pass takes under a millisecond and
	// keeps at most the pixels of the tiles the boundary runs through, whose rays the device drops one by one anyway.
	constexpr int kTile = 8;
	const int tx = (nx + kTile - 1) / kTile, ty = (ny + kTile - 1) / kTile;
	std::vector<uint8_t> tile_live((size_t)tx * ty, 0);
	for (int ti = 0; ti < tx; ti++) {
		const Iv u = {((double)(ti * kTile) - 1e-3) / nx, ((double)std::min(ti * kTile + kTile, nx) + 1e-3) / nx};
		for (int tj = 0; tj < ty; tj++) {
			const Iv v = {((double)(tj * kTile) - 1e-3) / ny, ((double)std::min(tj * kTile + kTile, ny) + 1e-3) / ny};
			Iv isp[3], org[3];
			for (int c = 0; c < 3; c++) {
				const Iv a = scale(u, (double)dc.horizontal[c]), b = scale(v, (double)dc.vertical[c]);
				const double mag = std::fabs((double)dc.llc_rel[c]) + std::fabs((double)dc.horizontal[c]) + std::fabs((double)dc.vertical[c]);
				isp[c] = widen(Iv{dc.llc_rel[c] + a.lo + b.lo, dc.llc_rel[c] + a.hi + b.hi}, 1e-5 * mag);
				org[c] = widen(Iv{(double)dc.origin[c], (double)dc.origin[c]}, lens_r[c] + 1e-5 * std::fabs((double)dc.origin[c]));
			}
			bool live = false;
			for (const auto &var : variants) {
				double near_lo = -inf, far_hi = inf; // largest lower bound of an entry distance, smallest upper bound of an exit distance
				for (int c = 0; c < 3; c++) {
					Iv d = widen(scale(isp[c], var.first), lens_r[c]); // direction before normalisation: k * isp - lens
					d = widen(d, 1e-5 * std::max(std::fabs(d.lo), std::fabs(d.hi)));
					if (!var.second) d = Iv{-d.hi, -d.lo};
					if (d.lo <= 0.0 && d.hi >= 0.0) continue; // the ray may run along this slab: no bound from this axis
					// entry plane: the lower face for d > 0, the upper face for d < 0; numerators as intervals over the origin
					const Iv n_in = d.lo > 0.0 ? Iv{blo[c] - org[c].hi, blo[c] - org[c].lo} : Iv{bhi[c] - org[c].hi, bhi[c] - org[c].lo};
					const Iv n_out = d.lo > 0.0 ? Iv{bhi[c] - org[c].hi, bhi[c] - org[c].lo} : Iv{blo[c] - org[c].hi, blo[c] - org[c].lo};
					const double q_in = std::min({n_in.lo / d.lo, n_in.lo / d.hi, n_in.hi / d.lo, n_in.hi / d.hi});
					const double q_out = std::max({n_out.lo / d.lo, n_out.lo / d.hi, n_out.hi / d.lo, n_out.hi / d.hi});
					near_lo = std::max(near_lo, q_in);
					far_hi = std::min(far_hi, q_out);
				}
				const double slack = 1e-5 * std::max(std::fabs(near_lo) < inf ? std::fabs(near_lo) : 0.0, std::fabs(far_hi) < inf ? std::fabs(far_hi) : 0.0);
				if (!(near_lo > far_hi + slack) && !(far_hi < -slack)) { live = true; break; }
			}
			tile_live[(size_t)ti * ty + tj] = live ? 1 : 0;
		}
	}
	out.clear();
	out.reserve((size_t)nx * ny);
	for (int i = 0; i < nx; i++) {
		const uint8_t *col = &tile_live[(size_t)(i / kTile) * ty];
		for (int j = 0; j < ny; j++)
			if (col[j / kTile]) out.push_back((uint32_t)(i * ny + j));
	}
}

hipError_t render_begin(RenderState &rs, const rt_camera &cam, int nx, int ny, int ns, int spectral_samples, int algorithm,
                        uint64_t seed, void *d_film, hipStream_t stream, std::string &err)
{
	Impl &im = *impl_of(rs);
	// band table, renderer.cpp:85-88 evaluated for every k in fp64
	std::vector<DevBand> bands(spectral_samples);
	for (int k = 0; k < spectral_samples; k++) {
		double min_wl = 400.0 + 300.0 / spectral_samples * k;
		double max_wl = min_wl + 300.0 / spectral_samples - 0.00001;
		double central = (min_wl + max_wl) / 2.0;
		DevBand &b = bands[k];
		double lo = std::ceil((min_wl - 400) / 10), hi = std::floor((max_wl - 400) / 10);
		b.min_i = (int)lo;
		b.max_i = (int)hi;
		if (b.min_i < 0 || b.max_i >= RT_NBINS) { err = "spectral band outside 400..700 nm"; return hipErrorInvalidValue; }
		b.get_bin = (int)std::floor((central - 400) / 10);
		b.flags = (b.min_i >= 1 ? 1 : 0) | (b.max_i <= RT_NBINS - 2 ? 2 : 0);
		b.mod_min = (float)std::fmod(min_wl - 400, 10.0);
		b.mod_max = (float)std::fmod(max_wl - 400, 10.0);
		// LensMakersFormula + image distance, camera.h:161-163,197-200
		double nl = cam.n[b.get_bin];
		double focal = 1.0 / ((nl - 1) * cam.lens_maker_coefficient);
		double a = 1.0 / (1.0 / focal - 1.0 / cam.d);
		b.neg_a_over_d = (float)(-a / cam.d);
		b.a_positive = a > 0.0 ? 1.0f : 0.0f;
	}
	HIP_TRY(upload(im.bands, bands, stream));
	im.n_bands = (uint32_t)spectral_samples;
	im.emit_dirty = true;
	DevCamera &dc = im.cam;
	for (int c = 0; c < 3; c++) {
		dc.origin[c] = (float)cam.origin[c];
		dc.llc_rel[c] = (float)(cam.film_lower_left_corner[c] - cam.origin[c]);
		dc.horizontal[c] = (float)cam.horizontal[c];
		dc.vertical[c] = (float)cam.vertical[c];
		dc.w[c] = (float)cam.w[c];
	}
	dc.half_aperture = (float)(cam.aperture / 2.0);
	double p_image = 1.0 / (cam.film_width * cam.film_height);              // camera.h:177
	double p_lens = 1.0 / (M_PI * cam.aperture * cam.aperture / 4.0);        // camera.h:178
	dc.weight_scale = (float)(cam.film_sensitivity / (cam.d * cam.d) / (p_image * p_lens)); // renderer.cpp:113-114

	const size_t film_elems = (size_t)nx * ny * RT_NBINS;
	if (rs.film_owned && rs.d_film) { (void)hipFree(rs.d_film); }
	rs.d_film = nullptr;
	rs.film_owned = false;
	if (d_film) rs.d_film = (float *)d_film;
	else {
		HIP_TRY(hipMalloc((void **)&rs.d_film, film_elems * sizeof(float)));
		rs.film_owned = true;
	}
	HIP_TRY(hipMemsetAsync(rs.d_film, 0, film_elems * sizeof(float), stream)); // renderer.cpp:58
	if (!im.counters) {
		HIP_TRY(hipMalloc((void **)&im.counters, sizeof(Counters) * (kMaxBounces + 2)));
		HIP_TRY(hipHostMalloc((void **)&im.h_counters, sizeof(Counters) * (kMaxBounces + 2), hipHostMallocDefault));
		for (auto &e : im.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	rs.nx = nx; rs.ny = ny; rs.ns = ns; rs.spectral_samples = spectral_samples; rs.algorithm = algorithm; rs.seed = seed;
	rs.closest_rays = rs.any_rays = rs.camera_samples = 0;
	im.carry_count = 0; // a new render: paths of an unfinished one are dropped
	{ // diagnostic switch for tests: trace the connection rays that cannot reach their light as well
		const char *all = std::getenv("MI355RT_TRACE_ALL_CONNECTIONS");
		im.cull_connections = !(all && all[0] == '1');
		const char *m = std::getenv("MI355RT_CULL_MASK"); // diagnostic: bit 0 camera rays, bit 1 dead paths, bit 2 hopeless connections
		im.cull_mask = m ? std::atoi(m) : 7;
		const char *b = std::getenv("MI355RT_BATCH_LOG2");
		im.batch_log2 = b ? std::min(28, std::max(20, std::atoi(b))) : PATH_BATCH_LOG2;
		if (!b) {
			// 2^28 paths per batch where the memory is there for it (110 GB of queues for a full batch: 288 GB of HBM per GPU is what
			// makes that affordable): a batch pays the launches of its late, nearly empty bounces once (1080p MIS: 2^27 5142, 2^28 5236 spp/s)
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + im.capacity * 365ull >= (200ull << 30)) im.batch_log2 = 28;
		}
	}
	{
		std::vector<uint32_t> live;
		const auto t0 = std::chrono::steady_clock::now();
		live_pixel_list(dc, bands, nx, ny, im.scene_lo, im.scene_hi, live);
		const auto t1 = std::chrono::steady_clock::now();
		im.n_live = live.size();
		HIP_TRY(upload(im.live_pixels, live, stream));
		if (std::getenv("MI355RT_DEBUG")) {
			HIP_TRY(hipStreamSynchronize(stream));
			const auto t2 = std::chrono::steady_clock::now();
			std::fprintf(stderr, "[mi355rt] live pixels %zu of %zu (list %.1f ms, upload %.1f ms)\n", live.size(), (size_t)nx * ny,
			             std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
		}
	}
	rs.begun = true;
	return hipStreamSynchronize(stream);
}

static hipError_t ensure_emit(Impl &im, hipStream_t stream, std::string &err)
{
	if (!im.emit_dirty) return hipSuccess;
	const size_t n = im.host_materials.size() * (size_t)im.n_bands;
	if (n > im.emit_size) {
		if (im.emit) (void)hipFree(im.emit);
		im.emit = nullptr;
		im.emit_size = 0;
		HIP_TRY(hipMalloc((void **)&im.emit, n * sizeof(float)));
		im.emit_size = n;
	}
	if (n) {
		hipLaunchKernelGGL(k_emit_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, im.materials, (uint32_t)im.host_materials.size(), im.bands, im.n_bands, im.emit);
		HIP_TRY(hipGetLastError());
	}
	im.emit_dirty = false;
	return hipSuccess;
}

static hipError_t ensure_queues(Impl &im, uint64_t cap, std::string &err)
{
	if (cap <= im.capacity) return hipSuccess;
	free_queues(im);
	for (int k = 0; k < 2; k++) {
		HIP_TRY(hipMalloc((void **)&im.rays[k], cap * sizeof(rt_ray)));
		HIP_TRY(hipMalloc((void **)&im.state[k], cap * sizeof(PathState)));
	}
	HIP_TRY(hipMalloc((void **)&im.hits, cap * sizeof(rt_hit)));
	HIP_TRY(hipMalloc((void **)&im.conn_rays, 2 * cap * sizeof(rt_ray)));
	HIP_TRY(hipMalloc((void **)&im.conn, 2 * cap * sizeof(ConnRec)));
	HIP_TRY(hipMalloc((void **)&im.conn_hits, 2 * cap * sizeof(rt_hit)));
	HIP_TRY(hipMalloc((void **)&im.conn_any, 2 * cap));
	im.capacity = cap;
	im.conn_capacity = 2 * cap;
	return hipSuccess;
}

template <int ALG>
static hipError_t run_batch(RenderState &rs, Impl &im, const DeviceGeom &geom, int first_sample, int n_samples, hipStream_t stream, std::string &err, bool flush)
{
	// Late bounces hold few paths (roulette halves them, then kills 90 % per bounce), and a launch over a few thousand rays
	// costs the latency of one traversal whatever its size: bounces 5 and later were 12 % of a 1080p batch for 1.5 % of its
	// rays.  A batch therefore stops once the path queue is known to be below 1/kCarryDivisor of the batch; the survivors
	// stay where they are and the next batch generates its camera rays behind them (a path carries its bounce number, pixel
	// and random stream, so paths of different depth mix freely).  render_flush runs what is left to completion; everything
	// that reads the film or the counters calls it first.
	// camera rays that cannot touch the scene are dropped at the source unless a miss contributes (the naive integrator's environment):
	// pixels no ray of which can reach the scene's box are not generated at all, the others are tested ray by ray
	const bool cull_camera = !flush && im.cull_connections && (im.cull_mask & 1) && !rs.capture_buf && !(ALG == RT_ALG_NAIVE && im.env != nullptr);
	const uint64_t npix = (uint64_t)rs.nx * rs.ny, n_paths = flush ? 0 : (cull_camera ? im.n_live : npix) * (uint64_t)n_samples;
	const uint64_t carry_limit = (flush || rs.capture_buf) ? 0 : npix * (uint64_t)n_samples / kCarryDivisor;
	const uint64_t carried = im.carry_count;
	int cur = carried ? im.carry_cur : 0;
	im.carry_count = 0;
	const DeviceShade sh = make_shade(im);
	const float tmin = ALG == RT_ALG_NEE ? 0.0001f : 0.001f; // renderer.cpp:173,235,337
	Profiler nop, &prof = rs.profiler ? *rs.profiler : nop;
	auto capture = [&](const rt_ray *d, uint64_t n) -> hipError_t {
		if (!rs.capture_buf || rs.capture_count >= rs.capture_capacity) return hipSuccess;
		uint64_t k = std::min<uint64_t>(n, rs.capture_capacity - rs.capture_count);
		hipError_t e = hipMemcpyAsync(rs.capture_buf + rs.capture_count, d, k * sizeof(rt_ray), hipMemcpyDeviceToHost, stream);
		if (e == hipSuccess) e = hipStreamSynchronize(stream);
		rs.capture_count += k;
		if (rs.capture_count_out) *rs.capture_count_out = rs.capture_count;
		return e;
	};
	// Queue sizes stay on the device: every kernel of bounce b reads its item count from the slot the previous
	// kernel wrote, and the host only sizes grids from upper bounds (paths can only die; at most two connection
	// rays per path).  Counts are copied back asynchronously; the host runs at most two bounces ahead of the
	// copies and stops as soon as one of them reports an empty path queue.
	uint64_t entering = carried + n_paths; // paths entering the first bounce of this round of slots
	bool first_round = true;
	for (;;) {
		HIP_TRY(hipMemsetAsync(im.counters, 0, sizeof(Counters) * (kMaxBounces + 2), stream));
		HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)im.counters, (int)(first_round && cull_camera ? carried : entering), 1, stream));
		im.h_counters[0].tails = entering;
		if (first_round && n_paths) {
			const unsigned blocks = (unsigned)((n_paths + kBlock - 1) / kBlock);
			prof.begin(Profiler::GENERATE, n_paths, stream);
			if (cull_camera) {
				const unsigned cull_blocks = (unsigned)((n_paths + kGenBlock * kGenPer - 1) / (kGenBlock * kGenPer));
				hipLaunchKernelGGL(k_generate_cull<false>, dim3(cull_blocks), dim3(kGenBlock), 0, stream, im.cam, im.bands, fast_div(im.n_bands), rs.nx, fast_div((uint32_t)rs.ny), rs.ns,
				                   first_sample, fast_div((uint32_t)n_samples), n_paths, rs.seed, tmin, im.rays[cur], im.state[cur], (BdptSample *)nullptr, im.scene_lo, im.scene_hi,
				                   &im.counters[0].tails, im.live_pixels);
				// the queue length is what the kernel left in the counter; the host keeps its upper bound
				HIP_TRY(hipMemcpyAsync(&im.h_counters[0], &im.counters[0], sizeof(Counters), hipMemcpyDeviceToHost, stream));
			} else {
				hipLaunchKernelGGL(k_generate<false>, dim3(blocks), dim3(kBlock), 0, stream, im.cam, im.bands, fast_div(im.n_bands), rs.nx, fast_div((uint32_t)rs.ny), rs.ns,
				                   first_sample, fast_div((uint32_t)n_samples), n_paths, rs.seed, tmin, im.rays[cur] + carried, im.state[cur] + carried, (BdptSample *)nullptr);
			}
			prof.end(stream);
			HIP_TRY(hipGetLastError());
		}
		if (first_round && !flush) rs.camera_samples += npix * (uint64_t)n_samples; // the culled ones were samples too
		first_round = false;
		uint64_t bound = entering;
		int known = 0, b = 0;
		// profiler records of bounce k: its path trace (-1 when the launch over the previous bounce's connection rays took the path
		// rays along), shade, connection trace, connect
		long rec_tp[kMaxBounces + 1], rec_sh[kMaxBounces + 1], rec_tc[kMaxBounces + 1], rec_co[kMaxBounces + 1];
		bool took_next[kMaxBounces + 1]; // the connection trace of bounce k also traced the path rays of bounce k + 1
		bool empty = false, cut = false, pre_traced = false;
		// MIS traces its connection rays for the closest hit, like its path rays, and both queues are filled by the same shade launch:
		// the connection rays of bounce b and the path rays of bounce b + 1 go into one launch (every launch ends with a few long rays
		// running alone; one launch has one such tail where two have two)
		static const char *const two_env = std::getenv("MI355RT_TRACE_TWO"); // experiments: 0 = one launch per queue
		const bool two = ALG == RT_ALG_MIS && !rs.capture_buf && !(two_env && two_env[0] == '0');
		for (; b < kMaxBounces; b++) {
			while (known < b) { // take the counts that have arrived; never run more than two bounces ahead
				if (b - known > 2 || rs.capture_buf) HIP_TRY(hipEventSynchronize(im.ev[known + 1]));
				else if (hipEventQuery(im.ev[known + 1]) != hipSuccess) break;
				known++;
				bound = std::min<uint64_t>(bound, (uint32_t)im.h_counters[known].tails);
			}
			if (bound == 0) { empty = true; break; }
			// few paths left: they join the next batch (unless they have been traced already: then they are shaded first)
			if (!pre_traced && b >= 1 && bound <= carry_limit) { cut = true; break; }
			const uint32_t *n_path = reinterpret_cast<const uint32_t *>(&im.counters[b]);
			const uint32_t *n_conn = reinterpret_cast<const uint32_t *>(&im.counters[b + 1]) + 1;
			rec_tp[b] = rec_tc[b] = rec_co[b] = -1;
			took_next[b] = false;
			if (!pre_traced) {
				if (rs.capture_buf) HIP_TRY(capture(im.rays[cur], bound)); // capture mode is synchronous: bound is exact here
				prof.begin(Profiler::CLOSEST, bound, stream);
				// the first bounce of a round comes in pixel order (camera rays, or carried paths with camera rays behind them)
				if (b >= 1 && bins_pay(geom, bound)) HIP_TRY(trace_ordered(im, geom, im.rays[cur], bound, im.hits, n_path, stream, err));
				else HIP_TRY(launch_trace_closest(geom, im.rays[cur], bound, im.hits, nullptr, stream, n_path));
				prof.end(stream);
				rec_tp[b] = prof.last();
			}
			pre_traced = false;
			prof.begin(Profiler::SHADE, bound, stream);
			{
				const dim3 grid((unsigned)((bound + kShadeBlock - 1) / kShadeBlock));
				if (im.g_code)
					hipLaunchKernelGGL((k_shade<ALG, true>), grid, dim3(kShadeBlock), 0, stream, sh, n_path, im.rays[cur], im.state[cur], im.hits,
					                   im.rays[cur ^ 1], im.state[cur ^ 1], im.conn_rays, reinterpret_cast<ConnRecU *>(im.conn), &im.counters[b + 1], rs.d_film);
				else
					hipLaunchKernelGGL((k_shade<ALG, false>), grid, dim3(kShadeBlock), 0, stream, sh, n_path, im.rays[cur], im.state[cur], im.hits,
					                   im.rays[cur ^ 1], im.state[cur ^ 1], im.conn_rays, reinterpret_cast<ConnRecU *>(im.conn), &im.counters[b + 1], rs.d_film);
			}
			prof.end(stream);
			rec_sh[b] = prof.last();
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipMemcpyAsync(&im.h_counters[b + 1], &im.counters[b + 1], sizeof(Counters), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipEventRecord(im.ev[b + 1], stream));
			if (ALG != RT_ALG_NAIVE) {
				uint64_t conn_bound = 2 * bound;
				if (rs.capture_buf) {
					HIP_TRY(hipEventSynchronize(im.ev[b + 1]));
					conn_bound = (uint32_t)(im.h_counters[b + 1].tails >> 32);
					HIP_TRY(capture(im.conn_rays, conn_bound));
				}
				if (conn_bound > 0) {
					if (ALG == RT_ALG_NEE) {
						prof.begin(Profiler::ANY, conn_bound, stream);
						HIP_TRY(launch_trace_any(geom, im.conn_rays, conn_bound, im.conn_any, stream, n_conn));
						prof.end(stream);
					} else if (b >= 1 && bins_pay(geom, conn_bound)) {
						// (with the ordering, every queue is a launch of its own: the next bounce's path rays are ordered when their turn comes)
						prof.begin(Profiler::CLOSEST, conn_bound, stream);
						HIP_TRY(trace_ordered(im, geom, im.conn_rays, conn_bound, im.conn_hits, n_conn, stream, err));
						prof.end(stream);
					} else if (two && b + 1 < kMaxBounces && bound > carry_limit && !bins_pay(geom, bound)) {
						// the next bounce will run (as far as the counts known now say; if it turns out to be small it runs all the same):
						// its path rays, which the shade launch above has just written, are traced together with these connection rays
						const uint32_t *n_next = reinterpret_cast<const uint32_t *>(&im.counters[b + 1]);
						prof.begin(Profiler::CLOSEST, conn_bound + bound, stream);
						HIP_TRY(launch_trace_closest_two(geom, im.conn_rays, conn_bound, im.conn_hits, n_conn, im.rays[cur ^ 1], bound, im.hits, n_next, stream));
						prof.end(stream);
						took_next[b] = true;
						pre_traced = true;
					} else {
						prof.begin(Profiler::CLOSEST, conn_bound, stream);
						HIP_TRY(launch_trace_closest(geom, im.conn_rays, conn_bound, im.conn_hits, nullptr, stream, n_conn));
						prof.end(stream);
					}
					rec_tc[b] = prof.last();
					prof.begin(Profiler::CONNECT, conn_bound, stream);
					if (im.g_code)
						hipLaunchKernelGGL((k_connect<ALG, true>), dim3(stream_grid(conn_bound, kBlock)), dim3(kBlock), 0, stream, geom, sh,
						                   n_conn, im.conn_rays, reinterpret_cast<const ConnRecU *>(im.conn), im.conn_hits, im.conn_any, rs.d_film);
					else
						hipLaunchKernelGGL((k_connect<ALG, false>), dim3(stream_grid(conn_bound, kBlock)), dim3(kBlock), 0, stream, geom, sh,
						                   n_conn, im.conn_rays, reinterpret_cast<const ConnRecU *>(im.conn), im.conn_hits, im.conn_any, rs.d_film);
					prof.end(stream);
					HIP_TRY(hipGetLastError());
					rec_co[b] = prof.last();
				}
			}
			cur ^= 1;
		}
		// all counts of this round are on the host after the stream drains: exact statistics and profile items
		HIP_TRY(hipStreamSynchronize(stream));
		for (int k = 0; k < b; k++) {
			const uint64_t np = (uint32_t)im.h_counters[k].tails, nc = (uint32_t)(im.h_counters[k + 1].tails >> 32);
			rs.closest_rays += np;
			if (ALG == RT_ALG_NEE) rs.any_rays += nc;
			else if (ALG == RT_ALG_MIS) rs.closest_rays += nc;
			prof.set_items(rec_tp[k], np);
			prof.set_items(rec_sh[k], np);
			prof.set_items(rec_tc[k], nc + (took_next[k] ? (uint64_t)(uint32_t)im.h_counters[k + 1].tails : 0));
			prof.set_items(rec_co[k], nc);
		}
		if (empty) break;
		entering = (uint32_t)im.h_counters[b].tails; // exact now; the survivors sit at the front of rays[cur]
		if (cut) {
			im.carry_count = entering;
			im.carry_cur = cur;
			break;
		}
		if (entering == 0) break; // otherwise the slots are exhausted (a path longer than kMaxBounces): go round again
	}
	return hipSuccess;
}

static void free_bdpt(Impl &im)
{
	void *ptrs[] = {im.bdpt_samples, im.bdpt_verts, im.bdpt_scan_tmp, im.bdpt_counts, im.bdpt_offsets, im.bdpt_pairs};
	for (void *p : ptrs) if (p) (void)hipFree(p);
	im.bdpt_samples = im.bdpt_verts = im.bdpt_scan_tmp = nullptr;
	im.bdpt_counts = im.bdpt_offsets = nullptr;
	im.bdpt_pairs = nullptr;
	im.bdpt_capacity = 0;
}

static hipError_t ensure_bdpt(Impl &im, uint64_t cap, hipStream_t stream, std::string &err)
{
	if (cap <= im.bdpt_capacity) return hipSuccess;
	free_bdpt(im);
	HIP_TRY(hipMalloc(&im.bdpt_samples, cap * sizeof(BdptSample)));
	HIP_TRY(hipMalloc(&im.bdpt_verts, cap * 2 * kBdptMax * sizeof(BdptVertex)));
	HIP_TRY(hipMalloc((void **)&im.bdpt_counts, cap * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **)&im.bdpt_offsets, cap * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **)&im.bdpt_pairs, 2 * sizeof(unsigned long long))); // the batch's vertex pairs, and (second word) the tail of its connection queue
	im.bdpt_scan_bytes = sizeof(uint32_t) * ((cap + kScanTile - 1) / kScanTile); // one sum per tile of the slot-count scan
	HIP_TRY(hipMalloc(&im.bdpt_scan_tmp, std::max<size_t>(im.bdpt_scan_bytes, 1)));
	im.bdpt_capacity = cap;
	return hipSuccess;
}

// one batch of NaiveBDPT samples (see render_bdpt.inc for the stages)
static hipError_t run_batch_bdpt(RenderState &rs, Impl &im, const DeviceGeom &geom, int first_sample, int n_samples, hipStream_t stream, std::string &err)
{
	const uint64_t npix = (uint64_t)rs.nx * rs.ny, n_generated = (im.cull_connections ? im.n_live : npix) * (uint64_t)n_samples;
	const DeviceShade sh = make_shade(im);
	Profiler nop, &prof = rs.profiler ? *rs.profiler : nop;
	BdptSample *samples = static_cast<BdptSample *>(im.bdpt_samples);
	BdptVertex *verts = static_cast<BdptVertex *>(im.bdpt_verts);
	if (n_generated == 0) { rs.camera_samples += npix * (uint64_t)n_samples; return hipSuccess; } // no pixel sees the scene
	uint64_t n_paths = n_generated; // samples that take part: all, or those whose camera ray can touch the scene
	prof.begin(Profiler::GENERATE, n_generated, stream);
	if (im.cull_connections) {
		HIP_TRY(hipMemsetAsync(im.counters, 0, sizeof(Counters), stream));
		const unsigned cull_blocks = (unsigned)((n_generated + kGenBlock * kGenPer - 1) / (kGenBlock * kGenPer));
		hipLaunchKernelGGL(k_generate_cull<true>, dim3(cull_blocks), dim3(kGenBlock), 0, stream, im.cam, im.bands, fast_div(im.n_bands), rs.nx, fast_div((uint32_t)rs.ny), rs.ns,
		                   first_sample, fast_div((uint32_t)n_samples), n_generated, rs.seed, 0.001f, im.rays[0], im.state[0], samples, im.scene_lo, im.scene_hi, &im.counters[0].tails, im.live_pixels);
		prof.end(stream);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(&im.h_counters[0], &im.counters[0], sizeof(Counters), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream)); // one wait per batch of up to 2^27 samples: everything below is sized by the count
		n_paths = (uint32_t)im.h_counters[0].tails;
	} else {
		hipLaunchKernelGGL(k_generate<true>, dim3((unsigned)((n_generated + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, im.cam, im.bands, fast_div(im.n_bands),
		                   rs.nx, fast_div((uint32_t)rs.ny), rs.ns, first_sample, fast_div((uint32_t)n_samples), n_generated, rs.seed, 0.001f, im.rays[0], im.state[0], samples);
		prof.end(stream);
		HIP_TRY(hipGetLastError());
	}
	rs.camera_samples += npix * (uint64_t)n_samples;
	if (n_paths == 0) return hipSuccess;
	const uint32_t n = (uint32_t)n_paths;
	const unsigned blocks_n = (unsigned)((n_paths + kBlock - 1) / kBlock);
	// a subpath: trace, walk, repeat.  A walker that survives bounce b has stored b + 1 vertices, so nothing is left after
	// kBdptMax + 1 bounces.  Counts stay on the device (slot base + b of the counters, read by the kernels of bounce b); the
	// host only uses the copies that have already arrived to shrink the grids, and never waits for one.
	struct WalkRec { long first = -1; int bounces = 0; };
	auto walk = [&](bool light, int base, WalkRec &rec) -> hipError_t {
		constexpr int nb = kBdptMax + 1;
		static_assert(2 * (nb + 1) <= kMaxBounces + 2, "two subpaths' counters fit");
		HIP_TRY(hipMemsetAsync(im.counters + base, 0, sizeof(Counters) * (nb + 1), stream));
		HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(im.counters + base), (int)n, 1, stream));
		im.h_counters[base].tails = n;
		uint64_t bound = n_paths;
		int known = 0, cur = 0;
		for (int b = 0; b < nb; b++) {
			while (known < b && hipEventQuery(im.ev[base + known + 1]) == hipSuccess) {
				known++;
				bound = std::min<uint64_t>(bound, (uint32_t)im.h_counters[base + known].tails);
			}
			if (bound == 0) break;
			const uint32_t *n_live = reinterpret_cast<const uint32_t *>(&im.counters[base + b]);
			prof.begin(Profiler::CLOSEST, bound, stream);
			HIP_TRY(launch_trace_closest(geom, im.rays[cur], bound, im.hits, nullptr, stream, n_live));
			prof.end(stream);
			if (b == 0) rec.first = prof.last();
			prof.begin(Profiler::SHADE, bound, stream);
			const dim3 grid((unsigned)((bound + kShadeBlock - 1) / kShadeBlock));
			if (light)
				hipLaunchKernelGGL((k_bdpt_walk<true>), grid, dim3(kShadeBlock), 0, stream, sh, n_live, im.rays[cur], im.state[cur], im.hits,
				                   im.rays[cur ^ 1], im.state[cur ^ 1], &im.counters[base + b + 1], samples, verts, n, im.bdpt_counts, rs.d_film);
			else
				hipLaunchKernelGGL((k_bdpt_walk<false>), grid, dim3(kShadeBlock), 0, stream, sh, n_live, im.rays[cur], im.state[cur], im.hits,
				                   im.rays[cur ^ 1], im.state[cur ^ 1], &im.counters[base + b + 1], samples, verts, n, im.bdpt_counts, rs.d_film);
			prof.end(stream);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipMemcpyAsync(&im.h_counters[base + b + 1], &im.counters[base + b + 1], sizeof(Counters), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipEventRecord(im.ev[base + b + 1], stream));
			rec.bounces = b + 1;
			cur ^= 1;
		}
		return hipSuccess;
	};
	// exact counts once the stream has drained: statistics and the items of the profile records
	auto settle = [&](int base, const WalkRec &rec) {
		for (int b = 0; b < rec.bounces; b++) {
			const uint64_t live = (uint32_t)im.h_counters[base + b].tails;
			rs.closest_rays += live;
			if (rec.first >= 0) { prof.set_items(rec.first + 2 * b, live); prof.set_items(rec.first + 2 * b + 1, live); }
		}
	};
	WalkRec eye_rec, light_rec;
	hipError_t e = walk(false, 0, eye_rec);
	if (e != hipSuccess) return e;
	prof.begin(Profiler::GENERATE, n_paths, stream);
	hipLaunchKernelGGL(k_bdpt_light_start, dim3(blocks_n), dim3(kBlock), 0, stream, sh, n, samples, im.rays[0], im.state[0]);
	prof.end(stream);
	HIP_TRY(hipGetLastError());
	e = walk(true, kBdptMax + 2, light_rec);
	if (e != hipSuccess) return e;
	// connection slots: count, scan, generate, trace, evaluate
	HIP_TRY(hipMemsetAsync(im.bdpt_pairs, 0, sizeof(unsigned long long), stream));
	hipLaunchKernelGGL(k_bdpt_pairs, dim3(std::min(blocks_n, 2048u)), dim3(kBlock), 0, stream, n, im.bdpt_counts, im.bdpt_pairs);
	HIP_TRY(hipGetLastError());
	{
		const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
		uint32_t *tile_sums = static_cast<uint32_t *>(im.bdpt_scan_tmp);
		hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kBlock), 0, stream, im.bdpt_counts, n, tile_sums);
		hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(kBlock), 0, stream, tile_sums, n_tiles);
		hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(kBlock), 0, stream, im.bdpt_counts, n, tile_sums, im.bdpt_offsets);
		HIP_TRY(hipGetLastError());
	}
	uint32_t last_off = 0, last_cnt = 0;
	unsigned long long pairs = 0;
	HIP_TRY(hipMemcpyAsync(&last_off, im.bdpt_offsets + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipMemcpyAsync(&last_cnt, im.bdpt_counts + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipMemcpyAsync(&pairs, im.bdpt_pairs, sizeof(pairs), hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipStreamSynchronize(stream));
	settle(0, eye_rec);
	settle(kBdptMax + 2, light_rec);
	const uint64_t total = (uint64_t)last_off + (last_cnt & 0xFFFFu); // connections of the batch, those with a zero term included
	// the waves of k_bdpt_connect_gen, and the slots they may need: every item if none had a zero term, and a chunk per wave
	const unsigned gen_blocks = std::min(blocks_n, 1024u);
	const uint64_t conn_bound = total + (uint64_t)gen_blocks * (kBlock / 64) * kBdptChunk;
	if (conn_bound >= (1ull << 32)) { err = "NaiveBDPT: more than 2^32 connections in one batch"; return hipErrorInvalidValue; }
	if (conn_bound > im.conn_capacity) { // about 4.6 slots per sample on the Cornell box; grown on demand, kept for later batches
		const uint64_t want = conn_bound + total / 4;
		if (im.conn_rays) (void)hipFree(im.conn_rays);
		if (im.conn) (void)hipFree(im.conn);
		if (im.conn_hits) (void)hipFree(im.conn_hits);
		if (im.conn_any) (void)hipFree(im.conn_any);
		im.conn_rays = nullptr; im.conn = nullptr; im.conn_hits = nullptr; im.conn_any = nullptr;
		im.conn_capacity = 0;
		hipError_t ge = hipMalloc((void **)&im.conn_rays, want * sizeof(rt_ray));
		if (ge == hipSuccess) ge = hipMalloc((void **)&im.conn, want * sizeof(ConnRec));
		if (ge == hipSuccess) ge = hipMalloc((void **)&im.conn_hits, want * sizeof(rt_hit));
		if (ge == hipSuccess) ge = hipMalloc((void **)&im.conn_any, want);
		if (ge != hipSuccess) {
			// the path queues are useless without their connection queues: drop them all, so that the next pass of any
			// integrator allocates a complete set again instead of launching over null connection buffers
			free_queues(im);
			if (im.bdpt_batch_log2 > 20) im.bdpt_batch_log2--; // this scene makes more connections per sample than the batch size allowed for
			err = std::string("NaiveBDPT connection queues: ") + hipGetErrorString(ge) +
			      " (this pass is lost: begin the render again; later passes use batches of half the size, see MI355RT_BDPT_BATCH_LOG2)";
			return ge;
		}
		im.conn_capacity = want;
	}
	if (total == 0) return hipSuccess;
	uint32_t *conn_n = reinterpret_cast<uint32_t *>(im.bdpt_pairs + 1); // the queue's tail: the slots the waves took, at most conn_bound
	HIP_TRY(hipMemsetAsync(conn_n, 0, sizeof(uint32_t), stream));
	prof.begin(Profiler::SHADE, n_paths, stream);
	hipLaunchKernelGGL(k_bdpt_connect_gen, dim3(gen_blocks), dim3(kBlock), 0, stream, sh, n, samples, verts, conn_n, im.conn_rays, im.conn);
	prof.end(stream);
	HIP_TRY(hipGetLastError());
	prof.begin(Profiler::CLOSEST, total, stream);
	HIP_TRY(launch_trace_closest(geom, im.conn_rays, conn_bound, im.conn_hits, nullptr, stream, conn_n));
	prof.end(stream);
	const long rec_trace = prof.last();
	prof.begin(Profiler::CONNECT, total, stream);
	hipLaunchKernelGGL(k_bdpt_connect_eval, dim3(stream_grid(conn_bound, kBlock)), dim3(kBlock), 0, stream, sh, conn_n, im.conn_rays,
	                   im.conn, im.conn_hits, rs.d_film);
	prof.end(stream);
	const long rec_eval = prof.last();
	HIP_TRY(hipGetLastError());
	if (prof.enabled) { // the profile counts the slots that were traced
		uint32_t traced = 0;
		HIP_TRY(hipMemcpyAsync(&traced, conn_n, sizeof(traced), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		prof.set_items(rec_trace, traced);
		prof.set_items(rec_eval, traced);
	}
	rs.closest_rays += total - pairs; // slots whose term was zero are counted although their ray is answered at once
	rs.any_rays += pairs;
	return hipSuccess;
}

hipError_t render_pass(RenderState &rs, const DeviceGeom &scene_geom, int first_sample, int n_samples, hipStream_t stream, std::string &err)
{
	DeviceGeom geom = scene_geom;
	geom.surface_rays = 1; // see launch_trace in trace.hip
	Impl &im = *impl_of(rs);
	const uint64_t npix = (uint64_t)rs.nx * rs.ny;
	// samples per batch: large batches amortise the nearly empty late bounces (see PATH_BATCH_LOG2)
	if (rs.algorithm == RT_ALG_MIS_BDPT) return hipSuccess; // MISBDPT returns 0 for every sample (renderer.cpp:775-778): the film stays empty
	const bool bdpt = rs.algorithm == RT_ALG_NAIVE_BDPT;
	if (bdpt && im.bdpt_batch_log2 == 0) {
		const char *b = std::getenv("MI355RT_BDPT_BATCH_LOG2");
		if (b) im.bdpt_batch_log2 = std::min(BDPT_BATCH_LOG2_MAX, std::max(20, std::atoi(b)));
		else {
			size_t free_b = 0, total_b = 0;
			HIP_TRY(hipMemGetInfo(&free_b, &total_b));
			// what this context already holds for BDPT or for its queues is given back before a larger batch is set up, so it counts as free
			const uint64_t usable = (uint64_t)free_b / 4 * 3;
			int lg = BDPT_BATCH_LOG2_MAX;
			while (lg > 20 && (kBdptBytesPerSample << lg) > usable) lg--;
			im.bdpt_batch_log2 = lg;
		}
	}
	uint64_t per_batch = std::max<uint64_t>(1, (1ull << (bdpt ? im.bdpt_batch_log2 : im.batch_log2)) / std::max<uint64_t>(npix, 1));
	if (npix * per_batch >= (1ull << 31)) per_batch = 1;
	if (npix >= (1ull << 31)) { err = "image too large for 32-bit path indices"; return hipErrorInvalidValue; }
	const uint64_t batch_paths = npix * std::min<uint64_t>(per_batch, (uint64_t)std::max(n_samples, 1));
	const uint64_t queue_paths = batch_paths + batch_paths / kCarryDivisor + 1; // room for the paths carried over from the batch before
	if ((queue_paths > im.capacity || bdpt) && im.carry_count) HIP_TRY(render_flush(rs, geom, stream, err)); // growing frees the queues they sit in
	HIP_TRY(ensure_queues(im, queue_paths, err));
	HIP_TRY(ensure_emit(im, stream, err));
	if (bdpt) HIP_TRY(ensure_bdpt(im, npix * std::min<uint64_t>(per_batch, (uint64_t)std::max(n_samples, 1)), stream, err));
	int s = first_sample, end = first_sample + n_samples;
	while (s < end) {
		int b = (int)std::min<uint64_t>(per_batch, (uint64_t)(end - s));
		hipError_t e;
		switch (rs.algorithm) {
		case RT_ALG_NAIVE: e = run_batch<RT_ALG_NAIVE>(rs, im, geom, s, b, stream, err, false); break;
		case RT_ALG_NEE: e = run_batch<RT_ALG_NEE>(rs, im, geom, s, b, stream, err, false); break;
		case RT_ALG_NAIVE_BDPT: e = run_batch_bdpt(rs, im, geom, s, b, stream, err); break;
		default: e = run_batch<RT_ALG_MIS>(rs, im, geom, s, b, stream, err, false); break;
		}
		if (e != hipSuccess) return e;
		s += b;
	}
	return hipSuccess;
}

hipError_t render_flush(RenderState &rs, const DeviceGeom &scene_geom, hipStream_t stream, std::string &err)
{
	DeviceGeom geom = scene_geom;
	geom.surface_rays = 1;
	if (!rs.impl || !rs.begun) return hipSuccess;
	Impl &im = *impl_of(rs);
	if (im.carry_count == 0) return hipSuccess;
	switch (rs.algorithm) {
	case RT_ALG_NAIVE: return run_batch<RT_ALG_NAIVE>(rs, im, geom, 0, 0, stream, err, true);
	case RT_ALG_NEE: return run_batch<RT_ALG_NEE>(rs, im, geom, 0, 0, stream, err, true);
	case RT_ALG_MIS: return run_batch<RT_ALG_MIS>(rs, im, geom, 0, 0, stream, err, true);
	default: im.carry_count = 0; return hipSuccess;
	}
}

namespace {
__global__ void __launch_bounds__(kBlock) k_film_accumulate(float *__restrict__ dst, const float *__restrict__ src, uint64_t n)
{
	const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (idx < n) dst[idx] += src[idx];
}
} // namespace

// dst += src over n film elements (both on the current device), on `stream`
hipError_t render_accumulate_film(float *dst, const float *src, size_t n, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_film_accumulate, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, dst, src, (uint64_t)n);
	return hipGetLastError();
}

namespace {
struct FilmSources { const float *p[kMaxFilmSources]; };
// dst = sum of n_src films, four elements per thread and film: one pass over every film instead of one read-modify-write of dst
// per film (the order of the additions is the order of the films, whatever the launch)
__global__ void __launch_bounds__(kBlock) k_film_sum(float *__restrict__ dst, FilmSources src, uint32_t n_src, uint64_t n4, uint64_t n)
{
	const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (idx < n4) {
		float4 acc = reinterpret_cast<const float4 *>(src.p[0])[idx];
		for (uint32_t k = 1; k < n_src; k++) {
			const float4 v = reinterpret_cast<const float4 *>(src.p[k])[idx];
			acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
		}
		reinterpret_cast<float4 *>(dst)[idx] = acc;
	} else if (idx == n4) { // the last one to three elements
		for (uint64_t i = 4 * n4; i < n; i++) {
			float acc = src.p[0][i];
			for (uint32_t k = 1; k < n_src; k++) acc += src.p[k][i];
			dst[i] = acc;
		}
	}
}
} // namespace

// dst = src[0] + src[1] + ... over n film elements (all on the current device, 16-byte aligned), on `stream`
hipError_t render_sum_films(float *dst, const float *const *src, size_t n_src, size_t n, hipStream_t stream)
{
	if (n == 0 || n_src == 0) return hipSuccess;
	if (n_src > (size_t)kMaxFilmSources) return hipErrorInvalidValue;
	FilmSources fs{};
	for (size_t k = 0; k < n_src; k++) fs.p[k] = src[k];
	const uint64_t n4 = n / 4;
	hipLaunchKernelGGL(k_film_sum, dim3((unsigned)((n4 + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, dst, fs, (uint32_t)n_src, n4, (uint64_t)n);
	return hipGetLastError();
}

// n device floats into the caller's doubles
static hipError_t download_widened(Impl &im, const float *src, size_t n, double *out, hipStream_t stream, std::string &err)
{
	// The film crosses PCIe as the floats it is made of, in 32 MB pieces through two pinned buffers, and host threads widen one piece
	// to the caller's doubles while the next one arrives.  (Widened on the device and copied as doubles into pageable memory, 1080p
	// took 58 ms and 4K 300 ms: twice the bytes at a fifth of the link's rate.)
	constexpr size_t kPiece = 8u << 20; // floats
	if (!im.stage[0] || !im.stage[1] || !im.stage_ev[0] || !im.stage_ev[1]) { // all four or none: a partial set is given back
		hipError_t e = hipSuccess;
		for (int k = 0; k < 2 && e == hipSuccess; k++) {
			if (!im.stage[k]) e = hipHostMalloc((void **)&im.stage[k], kPiece * sizeof(float), hipHostMallocDefault);
			if (e == hipSuccess && !im.stage_ev[k]) e = hipEventCreateWithFlags(&im.stage_ev[k], hipEventDisableTiming);
		}
		if (e != hipSuccess) {
			for (int k = 0; k < 2; k++) {
				if (im.stage[k]) (void)hipHostFree(im.stage[k]);
				if (im.stage_ev[k]) (void)hipEventDestroy(im.stage_ev[k]);
				im.stage[k] = nullptr;
				im.stage_ev[k] = nullptr;
			}
			err = std::string("pinned pieces of the film download: ") + hipGetErrorString(e);
			return e;
		}
	}
	const size_t n_pieces = (n + kPiece - 1) / kPiece;
	auto request = [&](size_t c) -> hipError_t {
		const size_t at = c * kPiece, m = std::min(kPiece, n - at);
		hipError_t e = hipMemcpyAsync(im.stage[c & 1], src + at, m * sizeof(float), hipMemcpyDeviceToHost, stream);
		return e == hipSuccess ? hipEventRecord(im.stage_ev[c & 1], stream) : e;
	};
	const int n_threads = std::max(1, std::min(8, (int)std::thread::hardware_concurrency()));
	hipError_t e = n_pieces ? request(0) : hipSuccess;
	for (size_t c = 0; c < n_pieces && e == hipSuccess; c++) {
		if (c + 1 < n_pieces) e = request(c + 1);
		if (e == hipSuccess) e = hipEventSynchronize(im.stage_ev[c & 1]);
		if (e != hipSuccess) break;
		const size_t at = c * kPiece, m = std::min(kPiece, n - at);
		const float *from = im.stage[c & 1];
		double *to = out + at;
		// host threads share the widening; a thread that cannot be started leaves its part to the calling thread (parallel_for)
		parallel_for(m, std::max<uint64_t>(1, m / (uint64_t)n_threads), [from, to](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) to[i] = (double)from[i]; });
	}
	if (e != hipSuccess) (void)hipStreamSynchronize(stream); // nothing may still be writing the pinned pieces when they are reused
	return e;
}

hipError_t render_download_film(RenderState &rs, double *out, hipStream_t stream, std::string &err, const float *film)
{
	const hipError_t e = download_widened(*impl_of(rs), film ? film : rs.d_film, (size_t)rs.nx * rs.ny * RT_NBINS, out, stream, err);
	if (e != hipSuccess) err = std::string("film download: ") + hipGetErrorString(e);
	return e;
}

hipError_t render_download_preview(RenderState &rs, int samples_done, double *out, hipStream_t stream, std::string &err, const float *film)
{
	Impl &im = *impl_of(rs);
	const size_t npix = (size_t)rs.nx * rs.ny;
	if (npix * 3 > im.preview_size) { // kept between calls: the reference asks for a preview after every pass
		if (im.preview) (void)hipFree(im.preview);
		im.preview = nullptr;
		im.preview_size = 0;
		HIP_TRY(hipMalloc((void **)&im.preview, npix * 3 * sizeof(float)));
		im.preview_size = npix * 3;
	}
	const float scale = (float)((double)rs.ns / (double)std::max(samples_done, 1)); // renderer.cpp:131
	hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, film ? film : rs.d_film, (uint64_t)npix, scale, im.preview);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = download_widened(im, im.preview, npix * 3, out, stream, err);
	if (e != hipSuccess) err = std::string("preview download: ") + hipGetErrorString(e);
	return e;
}

hipError_t render_set_environment(RenderState &rs, const uint8_t *texels, int width, int height, int bpp, double brightness, hipStream_t stream, std::string &err)
{
	Impl &im = *impl_of(rs);
	HIP_TRY(hipStreamSynchronize(stream)); // no render may still be reading the old texture
	if (im.env) { (void)hipFree(im.env); im.env = nullptr; }
	if (!texels) return hipSuccess;
	const size_t bytes = (size_t)width * height * bpp;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&im.env), bytes));
	HIP_TRY(hipMemcpy(im.env, texels, bytes, hipMemcpyHostToDevice));
	im.env_w = width;
	im.env_h = height;
	im.env_bpp = bpp;
	im.env_brightness = (float)brightness;
	return hipSuccess;
}

void render_free(RenderState &rs)
{
#ifdef BDPT_COUNT
	{
		unsigned long long h[8] = {};
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bdpt_kind), sizeof(h));
		std::fprintf(stderr, "[bdpt slots] t=0 %llu, t=1 %llu, pairs %llu, empty %llu\n", h[3], h[4], h[5], h[6]);
	}
#endif
#ifdef SHADE_CLK
	{
		unsigned long long h[8] = {};
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shade_clk), sizeof(h));
		double tot = 0; for (int k = 0; k < 6; k++) tot += (double)h[k];
		const char *names[6] = {"loads+normal+slot", "light sample + c0", "bsdf sample c1", "continuation", "reserve", "writes"};
		if (tot > 0) for (int k = 0; k < 6; k++) std::fprintf(stderr, "[shade clk] %-20s %5.1f %%\n", names[k], 100.0 * h[k] / tot);
	}
#endif
	if (rs.impl) {
		Impl &im = *static_cast<Impl *>(rs.impl);
		free_queues(im);
		if (im.tri_shade) (void)hipFree(im.tri_shade);
		if (im.face_tri_begin) (void)hipFree(im.face_tri_begin);
		if (im.tri_leaf) (void)hipFree(im.tri_leaf);
		if (im.face_area) (void)hipFree(im.face_area);
		if (im.tri_cum) (void)hipFree(im.tri_cum);
		if (im.materials) (void)hipFree(im.materials);
		if (im.light_faces) (void)hipFree(im.light_faces);
		if (im.light_rec) (void)hipFree(im.light_rec);
		if (im.emit) (void)hipFree(im.emit);
		if (im.preview) (void)hipFree(im.preview);
		for (int k = 0; k < 2; k++)
			if (im.stage[k]) { (void)hipHostFree(im.stage[k]); (void)hipEventDestroy(im.stage_ev[k]); }
		if (im.light_box) (void)hipFree(im.light_box);
		if (im.bands) (void)hipFree(im.bands);
		if (im.env) (void)hipFree(im.env);
		if (im.tri_uv) (void)hipFree(im.tri_uv);
		if (im.live_pixels) (void)hipFree(im.live_pixels);
		free_graphs(im);
		free_bdpt(im);
		if (im.counters) {
			(void)hipFree(im.counters);
			for (auto &e : im.ev) if (e) (void)hipEventDestroy(e);
		}
		if (im.h_counters) (void)hipHostFree(im.h_counters);
		delete &im;
		rs.impl = nullptr;
	}
	if (rs.film_owned && rs.d_film) (void)hipFree(rs.d_film);
	rs.d_film = nullptr;
	rs.film_owned = false;
	rs.begun = false;
}

} // namespace mi355rt
