// extern "C" layer: argument checks, HIP error mapping, scene upload.  No kernels here.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include "context.h"
#include "device_build.h"
#include "filter.h"
#include "accept_region.h"
#include "parallel_for.h"
#include "presplit.h"
#include "ray_bins.h"
#include "trace.h"

using namespace mi355rt;

namespace {
thread_local std::string g_create_error;

int set_err(rt_context *ctx, int code, const std::string &msg)
{
	if (ctx) ctx->error = msg;
	else g_create_error = msg;
	return code;
}

// No exception leaves the library (the callers are C, ctypes, cgo-style bindings): what the host side can throw -- std::bad_alloc
// from a table of a large scene, std::system_error from a thread that could not be started -- ends the call with RT_ERR_HOST and
// its text in rt_last_error.  Every exported function with a body worth the name is a function-try-block ending in one of these.
int caught(rt_context *ctx, const char *what) noexcept
{
	try { return set_err(ctx, RT_ERR_HOST, std::string("host side: ") + what); } catch (...) { return RT_ERR_HOST; }
}
int caught_group(rt_group *g, const char *what) noexcept;
#define RT_CATCH(ctx) catch (const std::exception &e_) { return caught(ctx, e_.what()); } catch (...) { return caught(ctx, "unknown exception"); }
#define RT_CATCH_GROUP(g) catch (const std::exception &e_) { return caught_group(g, e_.what()); } catch (...) { return caught_group(g, "unknown exception"); }
#define RT_CATCH_CREATE catch (const std::exception &e_) { return caught(nullptr, e_.what()); } catch (...) { return caught(nullptr, "unknown exception"); }

template <class T>
bool dev_upload(rt_context *ctx, void *&dst, const std::vector<T> &src)
{
	size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
	if (ctx->fail_hip(hipMalloc(&dst, bytes), "hipMalloc")) return false;
	ctx->device_bytes += bytes;
	if (!src.empty() && ctx->fail_hip(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D")) return false;
	return true;
}

bool use_device(rt_context *ctx) { return !ctx->fail_hip(hipSetDevice(ctx->device), "hipSetDevice"); }

void free_bins(rt_context *ctx)
{
	for (void **p : {&ctx->d_bin_rays, &ctx->d_bin_perm, &ctx->d_bin_hits, &ctx->d_bin_t64, &ctx->d_bin_counters}) {
		if (*p) (void)hipFree(*p);
		*p = nullptr;
	}
	ctx->bin_capacity = 0;
}

// Enqueue-only ordering of the shared bin buffers between batches on different streams: bins_acquire makes `s` wait for the batch
// that used them last, bins_release marks the end of this batch's use.
bool bins_acquire(rt_context *ctx, hipStream_t s)
{
	if (!ctx->bin_done && ctx->fail_hip(hipEventCreateWithFlags(&ctx->bin_done, hipEventDisableTiming), "hipEventCreate (ray ordering)")) return false;
	if (ctx->bin_in_use && ctx->fail_hip(hipStreamWaitEvent(s, ctx->bin_done, 0), "hipStreamWaitEvent (ray ordering)")) return false;
	return true;
}
void bins_release(rt_context *ctx, hipStream_t s)
{
	if (hipEventRecord(ctx->bin_done, s) == hipSuccess) ctx->bin_in_use = true;
	else (void)hipStreamSynchronize(s); // cannot mark the end of this batch: let it finish before anyone else may touch the buffers
}

bool ensure_bins(rt_context *ctx, uint64_t n)
{
	if (n <= ctx->bin_capacity) return true;
	free_bins(ctx);
	if (ctx->fail_hip(hipMalloc(&ctx->d_bin_rays, n * sizeof(rt_ray)), "hipMalloc ordered rays") ||
	    ctx->fail_hip(hipMalloc(&ctx->d_bin_perm, n * sizeof(uint32_t)), "hipMalloc ray permutation") ||
	    ctx->fail_hip(hipMalloc(&ctx->d_bin_hits, n * sizeof(rt_hit)), "hipMalloc ordered hits") ||
	    ctx->fail_hip(hipMalloc(&ctx->d_bin_t64, n * sizeof(double)), "hipMalloc ordered t64") ||
	    ctx->fail_hip(hipMalloc(&ctx->d_bin_counters, 2 * kRayBins * sizeof(uint32_t)), "hipMalloc cell counters")) {
		free_bins(ctx);
		return false;
	}
	ctx->bin_capacity = n;
	return true;
}

// Ordering pays where the tree and the triangle records are far larger than the caches and the batch is large enough to fill the
// chip many times over.  Measured on 16 M unordered rays, ordering and the return of the results to the caller's order included:
// soups of 10M / 4M / 2.5M / 1M triangles +9.5 % / +5.4 % / +3.0 % / -4.3 % (the kernel alone gains 13 % on the largest).  An
// integrator's rays come in pixel order and go through launch_trace_closest directly.  MI355RT_RAY_BINS=0 switches the ordering
// off, =1 on for every batch.
constexpr uint64_t kBinMinRays = 1ull << 22, kBinMinTris = 2000000ull;
bool bins_wanted(const rt_context *ctx, uint64_t n)
{
	if (n >= (1ull << 32)) return false; // the permutation holds 32-bit indices
	if (const char *s = std::getenv("MI355RT_RAY_BINS")) {
		if (s[0] == '0') return false;
		if (s[0] == '1') return n > 0;
	}
	return n >= kBinMinRays && ctx->geom.n_tris >= kBinMinTris;
}

bool ensure_stage(rt_context *ctx, uint64_t n)
{
	if (n <= ctx->stage_capacity) return true;
	if (ctx->d_stage_rays) (void)hipFree(ctx->d_stage_rays);
	if (ctx->d_stage_out) (void)hipFree(ctx->d_stage_out);
	if (ctx->d_stage_t64) (void)hipFree(ctx->d_stage_t64);
	ctx->d_stage_rays = ctx->d_stage_out = ctx->d_stage_t64 = nullptr;
	ctx->stage_capacity = 0;
	if (ctx->fail_hip(hipMalloc(&ctx->d_stage_rays, n * sizeof(rt_ray)), "hipMalloc rays")) return false;
	if (ctx->fail_hip(hipMalloc(&ctx->d_stage_out, n * sizeof(rt_hit)), "hipMalloc hits")) return false;
	if (ctx->fail_hip(hipMalloc(&ctx->d_stage_t64, n * sizeof(double)), "hipMalloc t64")) return false;
	ctx->stage_capacity = n;
	return true;
}

// the caller's rays into the device staging buffer; true on failure (text in ctx->error)
// rays per part of a large host-buffer batch (rt_trace_closest): 2^22 rays are 128 MB of rays, four pinned pieces
constexpr uint64_t kHostPart = 1ull << 22;

bool stage_rays_in(rt_context *ctx, const rt_ray *rays, uint64_t n)
{
	const char *from = reinterpret_cast<const char *>(rays);
	return ctx->fail_hip(upload_generated(ctx->pinned, ctx->d_stage_rays, n, sizeof(rt_ray), ctx->stream,
	                                      [from](size_t b, size_t e, char *out) { std::memcpy(out, from + b * sizeof(rt_ray), (e - b) * sizeof(rt_ray)); }),
	                     "H2D rays");
}
} // namespace

bool rt_context::fail_hip(hipError_t e, const char *what)
{
	if (e == hipSuccess) return false;
	error = std::string(what) + ": " + hipGetErrorString(e);
	return true;
}

void rt_context::free_geometry()
{
	void **ptrs[] = {&d_nodes, &d_tri, &d_leaf_tri, &d_tri_face, &d_face_object};
	for (void **p : ptrs) {
		if (*p) (void)hipFree(*p);
		*p = nullptr;
	}
	geom = DeviceGeom{};
	device_bytes = 0;
	has_mesh = false;
}

extern "C" {

int rt_create(int device, rt_context **out)
try {
	if (!out) return set_err(nullptr, RT_ERR_INVALID, "rt_create: out is NULL");
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return set_err(nullptr, RT_ERR_NO_DEVICE, std::string("rt_create: no HIP device (") + hipGetErrorString(e) + "); this backend has no CPU path");
	if (device < 0 || device >= count) return set_err(nullptr, RT_ERR_INVALID, "rt_create: device ordinal out of range");
	rt_context *ctx = new (std::nothrow) rt_context();
	if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "rt_create: out of host memory");
	ctx->device = device;
	ctx->render.profiler = &ctx->profiler;
	if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
		std::string msg = std::string("rt_create: ") + hipGetErrorString(e);
		delete ctx;
		return set_err(nullptr, RT_ERR_HIP, msg);
	}
	// the allocation-free entry point's staging: address of this device's copy of the static array (device is current here)
	StaticStage stage;
	if ((e = static_stage(device, stage)) != hipSuccess) {
		std::string msg = std::string("rt_create: hipGetSymbolAddress(static staging): ") + hipGetErrorString(e);
		(void)hipStreamDestroy(ctx->stream);
		delete ctx;
		return set_err(nullptr, RT_ERR_HIP, msg);
	}
	if ((e = ctx->trace_work.init()) != hipSuccess) {
		std::string msg = std::string("rt_create: work counters of the traversal kernels: ") + hipGetErrorString(e);
		(void)hipStreamDestroy(ctx->stream);
		delete ctx;
		return set_err(nullptr, RT_ERR_HIP, msg);
	}
	*out = ctx;
	return RT_OK;
} RT_CATCH_CREATE

void rt_destroy(rt_context *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	render_free(ctx->render);
	ctx->profiler.destroy();
	ctx->trace_work.destroy();
	ctx->free_geometry();
	if (ctx->d_stage_rays) (void)hipFree(ctx->d_stage_rays);
	if (ctx->d_stage_out) (void)hipFree(ctx->d_stage_out);
	if (ctx->d_stage_t64) (void)hipFree(ctx->d_stage_t64);
	free_bins(ctx);
	if (ctx->bin_done) (void)hipEventDestroy(ctx->bin_done);
	ctx->pinned.release();
	ctx->pinned_down.release();
	if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
	if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

const char *rt_last_error(const rt_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

// render_pass may leave the last few paths of a batch for the next one; whatever reads the film or the counters, or changes
// what those paths would see, finishes them first
static int finish_pending_paths(rt_context *ctx)
{
	if (!ctx->render.begun) return RT_OK;
	std::string err;
	if (render_flush(ctx->render, ctx->geom, ctx->stream, err) != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	return RT_OK;
}

int rt_upload_mesh(rt_context *ctx, const rt_mesh *mesh)
try {
	if (!ctx || !mesh) return set_err(ctx, RT_ERR_INVALID, "rt_upload_mesh: NULL argument");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	if (ctx->fail_hip(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) return RT_ERR_HIP;
	// whatever happens below, the old scene is gone: a render in progress has nothing left to continue with
	ctx->free_geometry();
	ctx->has_materials = false;
	ctx->render.begun = false;
	std::string err;
	// MI355RT_DEBUG: where the time of an upload goes (host tables, tree, shading tables), on stderr
	const bool timing = std::getenv("MI355RT_DEBUG") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "[mi355rt] rt_upload_mesh: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	if (!ctx->host.build(*mesh, err)) return set_err(ctx, RT_ERR_MESH, err);
	lap("polygon tables");
	const HostScene &hs = ctx->host;
	const uint64_t nt = hs.n_tris;
	int builder = ctx->builder;
	if (const char *s = std::getenv("MI355RT_BUILDER")) builder = std::strcmp(s, "device") == 0 ? RT_BUILDER_DEVICE_LBVH : RT_BUILDER_HOST_SAH;
	const bool auto_chosen = builder == RT_BUILDER_AUTO;
	// device above 2^15 fan triangles (where the triangle records stop being packed): with the cost-minimising cut its trees trace within
	// 2 % of the host builder's on meshes (35 k and 65 k triangles: +1 %, 125 k: -1.7 %, 870 k: -2 %; 4.5 % on a triangle soup) and are
	// built in 1-10 ms where the host takes 30-500 ms, more than a 1024-spp render of such a scene loses to the slower tree
	if (auto_chosen) builder = nt > (1u << 15) ? RT_BUILDER_DEVICE_LBVH : RT_BUILDER_HOST_SAH;
	if (nt < 2) builder = RT_BUILDER_HOST_SAH; // the radix tree needs two leaves
	// fp64 triangle data: v0 + face normal, v1 + v2.  The host builder writes the 128-byte leaf-ordered records of device_scene.h
	// itself; the device builder takes the two halves in fan order and gathers the records on the GPU.
	auto fill_halves = [&](double *a, double *b, uint32_t t) {
		const double *v0, *v1, *v2;
		hs.tri_vertices(t, v0, v1, v2);
		const double *fn = &hs.face_normal[3 * (size_t)hs.tri_face[t]];
		a[0] = v0[0]; a[1] = v0[1]; a[2] = v0[2]; a[3] = fn[0]; a[4] = fn[1]; a[5] = fn[2];
		b[0] = v1[0]; b[1] = v1[1]; b[2] = v1[2]; b[3] = v2[0]; b[4] = v2[1]; b[5] = v2[2];
	};
	std::vector<uint32_t> tri_leaf(nt);
	// references of long triangles are cut into pieces first (presplit.h): both builders build over the references
	Presplit cut;
	{
		PresplitOptions po;
		po.budget = (float)ctx->cut_budget;
		if (const char *s = std::getenv("MI355RT_PRESPLIT")) po.budget = (float)std::atof(s); // experiments
		presplit(hs, po, cut);
	}
	ctx->cut_triangles = cut.first.size();
	ctx->extra_references = cut.extra.size();
	lap("long triangles cut");
	if (builder == RT_BUILDER_DEVICE_LBVH) {
		// records go up in fan order; the device builds the tree and reorders them
		void *d_a = nullptr, *d_b = nullptr;
		DeviceBvh built;
		bool ok = dev_upload(ctx, ctx->d_tri_face, hs.tri_face) && dev_upload(ctx, ctx->d_face_object, hs.face_object);
		const uint64_t bytes_before = ctx->device_bytes;
		ok = ok && !ctx->fail_hip(hipMalloc(&d_a, 48 * nt), "hipMalloc") && !ctx->fail_hip(hipMalloc(&d_b, 48 * nt), "hipMalloc");
		for (int half = 0; half < 2 && ok; half++)
			ok = !ctx->fail_hip(upload_generated(ctx->pinned, half ? d_b : d_a, nt, 48, ctx->stream, [&](size_t t_begin, size_t t_end, char *out) {
				double other[6];
				for (size_t t = t_begin; t < t_end; t++) {
					double *mine = reinterpret_cast<double *>(out) + 6 * (t - t_begin);
					if (half) fill_halves(other, mine, (uint32_t)t);
					else fill_halves(mine, other, (uint32_t)t);
				}
			}), "upload of the triangle halves");
		lap("triangle halves");
		if (ok && build_bvh_device((const double *)d_a, (const double *)d_b, (const uint32_t *)ctx->d_tri_face, (const uint32_t *)ctx->d_face_object,
		                           (uint32_t)nt, &cut, ctx->stream, built, err) != hipSuccess) {
			ctx->error = err;
			ok = false;
		}
		lap("copy to the device and tree build");
		if (d_a) (void)hipFree(d_a);
		if (d_b) (void)hipFree(d_b);
		ctx->device_bytes = bytes_before; // the fan-order copies are gone again
		if (ok) {
			ctx->d_nodes = built.nodes;
			ctx->d_tri = built.tri;
			ctx->geom.tri_stride = built.tri_stride;
			ctx->d_leaf_tri = built.leaf_tri;
			ctx->device_bytes += (uint64_t)built.n_nodes * sizeof(WideNode) + (uint64_t)built.n_refs * (16 * (uint64_t)built.tri_stride + 4);
			ctx->bvh = Bvh2();
			ctx->wide = BvhWide();
			ctx->wide.depth = built.depth;
			ctx->wide.tri_index.resize(built.n_refs);
			ctx->bvh.build_seconds = built.seconds;
			ctx->bvh.depth = built.depth;
			ctx->built_on_device = true;
			ctx->device_nodes = built.n_nodes;
			ok = !ctx->fail_hip(hipMemcpy(ctx->wide.tri_index.data(), built.leaf_tri, (size_t)built.n_refs * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy(leaf order)") &&
			     !ctx->fail_hip(hipMemcpy(tri_leaf.data(), built.tri_leaf, nt * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy(triangle positions)");
			(void)hipFree(built.tri_leaf);
		}
		if (!ok) {
			std::string keep = ctx->error;
			ctx->free_geometry();
			// a tree the Morton builder cannot shape (too deep for the traversal stack, too many coincident centroids in one node)
			// is not the caller's problem when the caller did not ask for that builder: the host builder takes over
			if (!auto_chosen) return set_err(ctx, RT_ERR_HIP, keep);
			builder = RT_BUILDER_HOST_SAH;
		}
	}
	if (builder != RT_BUILDER_DEVICE_LBVH) {
		BvhBuildOptions opt;
		if (const char *s = std::getenv("MI355RT_MAX_LEAF")) opt.max_leaf = std::atoi(s);
		// the cost-minimising cut pays on scenes that wait for memory or walk many nodes (1M soup +2.3 %, 870 k-triangle mesh +1.6 %, 490 k-triangle
		// hall +3.2 %); on a scene of a thousand triangles the greedy collapse traces 1 % faster
		opt.optimal_cut = triangle_record_stride(nt) == 8 ? 1 : 0;
		if (const char *s = std::getenv("MI355RT_WIDE_CUT")) opt.optimal_cut = std::atoi(s);        // experiments
		if (const char *s = std::getenv("MI355RT_WIDE_COST_TRI")) opt.wide_cost_tri = (float)std::atof(s);
		if (const char *s = std::getenv("MI355RT_SLOTS")) opt.slot_assignment = std::atoi(s);
		build_bvh(ctx->host, opt, &cut, ctx->bvh, ctx->wide);
		ctx->built_on_device = false;
		ctx->device_nodes = ctx->wide.nodes.size();
		if (ctx->wide.depth > kTraceMaxDepth) {
			ctx->free_geometry();
			return set_err(ctx, RT_ERR_MESH, "rt_upload_mesh: the tree is deeper than the traversal stack (" + std::to_string(ctx->wide.depth) + " levels, limit " + std::to_string(kTraceMaxDepth) + ")");
		}
		const uint64_t n_refs = ctx->wide.tri_index.size(); // the triangles, plus the additional pieces of those that were cut
		const uint32_t stride = triangle_record_stride(n_refs);
		ctx->geom.tri_stride = stride;
		const size_t rec_bytes = 16 * (size_t)stride;
		bool recs_ok = !ctx->fail_hip(hipMalloc(&ctx->d_tri, rec_bytes * std::max<uint64_t>(n_refs, 1)), "hipMalloc");
		if (recs_ok) ctx->device_bytes += rec_bytes * std::max<uint64_t>(n_refs, 1);
		for (uint64_t li = n_refs; li-- > 0;) tri_leaf[ctx->wide.tri_index[li]] = (uint32_t)li; // a cut triangle: the first of its leaves
		recs_ok = recs_ok && !ctx->fail_hip(upload_generated(ctx->pinned, ctx->d_tri, n_refs, rec_bytes, ctx->stream, [&](size_t li_begin, size_t li_end, char *out) {
			for (size_t li = li_begin; li < li_end; li++) { // leaf-ordered records
				const uint32_t t = ctx->wide.tri_index[li];
				double *rec = reinterpret_cast<double *>(out + rec_bytes * (li - li_begin));
				fill_halves(rec, rec + 6, t);
				const uint32_t meta[4] = {t, hs.tri_face[t], hs.face_object[hs.tri_face[t]], 0u};
				std::memcpy(rec + 12, meta, sizeof(meta));
				if (stride > 7) rec[14] = rec[15] = 0.0;
			}
		}), "upload of the triangle records");
		if (!recs_ok || !dev_upload(ctx, ctx->d_nodes, ctx->wide.nodes) ||
		    !dev_upload(ctx, ctx->d_leaf_tri, ctx->wide.tri_index) || !dev_upload(ctx, ctx->d_tri_face, hs.tri_face) ||
		    !dev_upload(ctx, ctx->d_face_object, hs.face_object)) {
			std::string keep = ctx->error;
			ctx->free_geometry();
			return set_err(ctx, RT_ERR_HIP, keep);
		}
	}
	lap("tree and records in place");
	ctx->geom.work = &ctx->trace_work;
	ctx->geom.nodes = (const float4 *)ctx->d_nodes;
	ctx->geom.tri = (const double2 *)ctx->d_tri;
	ctx->geom.leaf_tri = (const uint32_t *)ctx->d_leaf_tri;
	ctx->geom.tri_face = (const uint32_t *)ctx->d_tri_face;
	ctx->geom.face_object = (const uint32_t *)ctx->d_face_object;
	ctx->geom.n_tris = (uint32_t)nt;
	ctx->geom.depth = ctx->wide.depth;
	ctx->geom.watertight = ctx->triangle_test == RT_TRIANGLE_WATERTIGHT ? 1u : 0u;
	hipError_t e = render_upload_scene(ctx->render, hs, tri_leaf, ctx->geom.tri_face, ctx->geom.face_object, ctx->pinned, ctx->stream, err);
	if (e != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	lap("shading tables");
	ctx->has_mesh = true;
	return RT_OK;
} RT_CATCH(ctx)

int rt_set_builder(rt_context *ctx, int builder)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (builder != RT_BUILDER_HOST_SAH && builder != RT_BUILDER_DEVICE_LBVH && builder != RT_BUILDER_AUTO) return set_err(ctx, RT_ERR_INVALID, "rt_set_builder: unknown builder");
	ctx->builder = builder;
	return RT_OK;
} RT_CATCH(ctx)

int rt_set_reference_cutting(rt_context *ctx, double budget)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!(budget >= 0.0 && budget <= 4.0)) return set_err(ctx, RT_ERR_INVALID, "rt_set_reference_cutting: the budget is a fraction of the scene's triangles between 0 and 4");
	ctx->cut_budget = budget;
	return RT_OK;
} RT_CATCH(ctx)

int rt_get_reference_cutting(rt_context *ctx, uint64_t *cut_triangles, uint64_t *extra_references)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (cut_triangles) *cut_triangles = ctx->cut_triangles;
	if (extra_references) *extra_references = ctx->extra_references;
	return RT_OK;
} RT_CATCH(ctx)

int rt_set_triangle_test(rt_context *ctx, int mode)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (mode != RT_TRIANGLE_REFERENCE && mode != RT_TRIANGLE_WATERTIGHT) return set_err(ctx, RT_ERR_INVALID, "rt_set_triangle_test: unknown mode");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc; // paths in flight finish under the test they started with
	ctx->triangle_test = mode;
	ctx->geom.watertight = mode == RT_TRIANGLE_WATERTIGHT ? 1u : 0u;
	return RT_OK;
} RT_CATCH(ctx)

int rt_upload_materials(rt_context *ctx, const rt_material *per_object, uint32_t n_objects)
try {
	if (!ctx || !per_object) return set_err(ctx, RT_ERR_INVALID, "rt_upload_materials: NULL argument");
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_upload_materials: no mesh uploaded");
	if (n_objects != ctx->host.n_objects) return set_err(ctx, RT_ERR_NO_MATERIAL, "rt_upload_materials: one record per obj object is required");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	for (uint32_t o = 0; o < n_objects; o++)
		if (per_object[o].type < RT_MAT_NONE || per_object[o].type > RT_MAT_GGX)
			return set_err(ctx, RT_ERR_INVALID, "rt_upload_materials: unknown material type");
	for (uint32_t o = 0; o < n_objects; o++)
		if (per_object[o].type == RT_MAT_GGX && !(per_object[o].roughness > 0.0 && per_object[o].roughness <= 1.0e3))
			return set_err(ctx, RT_ERR_INVALID, "rt_upload_materials: a GGX record needs a roughness in (0, 1000]");
	ctx->materials.assign(per_object, per_object + n_objects);
	ctx->light_faces.clear();
	for (uint32_t o = 0; o < n_objects; o++) // renderer.cpp:35-39
		if (per_object[o].light_flag)
			for (uint64_t f = ctx->host.object_begin[o]; f < ctx->host.object_begin[o + 1]; f++) ctx->light_faces.push_back((uint32_t)f);
	ctx->has_materials = false;
	if (ctx->light_faces.empty()) return set_err(ctx, RT_ERR_NO_LIGHTS, "rt_upload_materials: no object has light_flag set");
	std::string err;
	if (render_upload_materials(ctx->render, ctx->host, ctx->materials, ctx->light_faces, ctx->stream, err) != hipSuccess)
		return set_err(ctx, RT_ERR_HIP, err);
	ctx->has_materials = true;
	return RT_OK;
} RT_CATCH(ctx)

int rt_upload_material_graphs(rt_context *ctx, const rt_material_graphs *g)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!ctx->has_materials) return set_err(ctx, RT_ERR_INVALID, "rt_upload_material_graphs: upload the material records first");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	if (g && g->n_programs) {
		for (uint32_t t = 0; t < g->n_textures; t++)
			if (!g->textures[t].texels || g->textures[t].width <= 0 || g->textures[t].height <= 0 || g->textures[t].bpp < 3)
				return set_err(ctx, RT_ERR_INVALID, "rt_upload_material_graphs: a texture needs texels, width, height > 0 and at least 3 bytes per texel");
		for (uint32_t i = 0; i < g->n_programs; i++) {
			const rt_graph_program &p = g->programs[i];
			if (p.object >= ctx->host.n_objects || p.code_len == 0 || (uint64_t)p.code_begin + p.code_len > g->n_code)
				return set_err(ctx, RT_ERR_INVALID, "rt_upload_material_graphs: program outside the object or code table");
			const int type = ctx->materials[p.object].type;
			if (type != RT_MAT_LAMBERTIAN && type != RT_MAT_DIFFUSE_LIGHT && type != RT_MAT_COLORED_METAL)
				return set_err(ctx, RT_ERR_INVALID, "rt_upload_material_graphs: only Lambertian, DiffuseLight and ColoredMetal have a slot fed by a graph");
			// type-check the program on an abstract stack: 'S' spectrum, 'V' vector, 'D' scalar
			char st[6];
			int sp = 0;
			auto bad = [&](const char *why) { return set_err(ctx, RT_ERR_INVALID, std::string("rt_upload_material_graphs: ") + why); };
			for (uint32_t pc = p.code_begin; pc < p.code_begin + p.code_len; pc++) {
				const uint32_t op = g->code[2 * pc], arg = g->code[2 * pc + 1];
				char push = 0;
				switch (op) {
				case 0: if (arg >= g->n_spectra) return bad("spectrum index out of range"); push = 'S'; break;
				case 1: if (arg >= g->n_vectors) return bad("vector index out of range"); push = 'V'; break;
				case 2: push = 'V'; break;
				case 3: push = 'S'; break;
				case 4: push = 'V'; break;
				case 5: if (arg >= g->n_textures) return bad("texture index out of range");
					if (sp < 1 || st[sp - 1] != 'V') return bad("IMAGE needs a vector"); break;
				case 6: if (arg >= g->n_scalars) return bad("scalar index out of range");
					if (sp < 1 || st[sp - 1] != 'V') return bad("CHECKER needs a vector"); break;
				case 7: if (sp < 1 || st[sp - 1] != 'V') return bad("RGB2SPEC needs a vector"); st[sp - 1] = 'S'; break;
				case 8: case 9: if (sp < 2 || st[sp - 1] == 'V' || st[sp - 1] != st[sp - 2]) return bad("this arithmetic needs two spectra or two scalars"); sp--; break;
				case 10: case 11: if (sp < 2 || st[sp - 1] != 'V' || st[sp - 2] != 'V') return bad("vector arithmetic needs two vectors"); sp--; break;
				case 12: if (arg > 2) return bad("ACCESS index must be 0, 1 or 2");
					if (sp < 1 || st[sp - 1] != 'V') return bad("ACCESS needs a vector"); st[sp - 1] = 'D'; break;
				case 13: if (sp < 3 || st[sp - 1] != 'D' || st[sp - 2] != 'D' || st[sp - 3] != 'D') return bad("COMBINE needs three scalars"); sp -= 2; st[sp - 1] = 'V'; break;
				case 14: { // scalars[arg] = size, then size + 1 table values
					if (arg >= g->n_scalars) return bad("scalar index out of range");
					const double n = g->scalars[arg];
					if (!(n >= 1.0 && n <= 1048576.0) || n != std::floor(n) || (uint64_t)arg + 1 + (uint64_t)n + 1 > g->n_scalars) return bad("NOISE1 table outside the scalar table");
					if (sp < 1 || st[sp - 1] != 'D') return bad("NOISE1 needs a scalar"); break;
				}
				case 15: { // scalars[arg] = u_size, v_size, then (u_size + 1)(v_size + 1) table values
					if ((uint64_t)arg + 1 >= g->n_scalars) return bad("scalar index out of range");
					const double nu = g->scalars[arg], nv = g->scalars[arg + 1];
					if (!(nu >= 1.0 && nu <= 65536.0 && nv >= 1.0 && nv <= 65536.0) || nu != std::floor(nu) || nv != std::floor(nv) ||
					    (uint64_t)arg + 2 + ((uint64_t)nu + 1) * ((uint64_t)nv + 1) > g->n_scalars) return bad("NOISE2 table outside the scalar table");
					if (sp < 2 || st[sp - 1] != 'D' || st[sp - 2] != 'D') return bad("NOISE2 needs two scalars"); sp--; break;
				}
				case 16: if (sp < 3 || st[sp - 1] != 'D' || st[sp - 2] != 'V' || st[sp - 3] != 'V') return bad("ROTATE needs vector, vector, scalar"); sp -= 2; break;
				case 17: if (arg >= g->n_scalars) return bad("scalar index out of range"); push = 'D'; break;
				default: return bad("unknown op");
				}
				if (push) {
					if (sp == 6) return bad("program needs more than six stack entries");
					st[sp++] = push;
				}
			}
			if (sp != 1 || st[0] != 'S') return bad("a program must leave exactly one spectrum");
		}
	}
	std::string err;
	if (render_upload_material_graphs(ctx->render, g, ctx->stream, err) != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH(ctx)

int rt_set_environment(rt_context *ctx, const uint8_t *texels, int width, int height, int bpp, double brightness)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (texels && (width <= 0 || height <= 0 || bpp < 3 || bpp > 4)) return set_err(ctx, RT_ERR_INVALID, "rt_set_environment: needs width, height > 0 and 3 or 4 bytes per texel");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	std::string err;
	if (render_set_environment(ctx->render, texels, width, height, bpp, brightness, ctx->stream, err) != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH(ctx)

int rt_set_camera(rt_context *ctx, const rt_camera *cam)
try {
	if (!ctx || !cam) return set_err(ctx, RT_ERR_INVALID, "rt_set_camera: NULL argument");
	ctx->camera = *cam;
	ctx->has_camera = true;
	return RT_OK;
} RT_CATCH(ctx)

int rt_clear(rt_context *ctx)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	(void)hipStreamSynchronize(ctx->stream);
	ctx->free_geometry();
	ctx->has_materials = false;
	ctx->render.begun = false; // a render in progress ends with the scene it was rendering
	ctx->host = HostScene();
	ctx->bvh = Bvh2();
	ctx->wide = BvhWide();
	return RT_OK;
} RT_CATCH(ctx)

// may_allocate: false for the entry point that promises to allocate nothing (it then never orders the rays)
static int trace_closest_device(rt_context *ctx, const void *d_rays, uint64_t n, void *d_hits, void *d_t64, void *stream, bool may_allocate)
{
	if (!ctx || (n && (!d_rays || !d_hits))) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest_device: NULL argument");
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest_device: no mesh uploaded");
	if (n >= (1ull << 32) * 256ull) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest_device: too many rays for one launch");
	if (!use_device(ctx)) return RT_ERR_HIP;
	hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
	const bool bins = may_allocate && bins_wanted(ctx, n);
	if (bins && (!ensure_bins(ctx, n) || !bins_acquire(ctx, s))) return RT_ERR_HIP;
	// the ordering and the way back are part of what the caller waits for: they are timed with the kernel
	ctx->profiler.begin(Profiler::CLOSEST, n, s);
	hipError_t e;
	if (bins) {
		e = bin_rays(ctx->geom.nodes, (const rt_ray *)d_rays, (uint32_t)n, (rt_ray *)ctx->d_bin_rays, (uint32_t *)ctx->d_bin_perm, (uint32_t *)ctx->d_bin_counters, s);
		if (e == hipSuccess) e = launch_trace_closest(ctx->geom, (const rt_ray *)ctx->d_bin_rays, n, (rt_hit *)ctx->d_bin_hits, d_t64 ? (double *)ctx->d_bin_t64 : nullptr, s);
		if (e == hipSuccess) e = unbin_results((const uint32_t *)ctx->d_bin_perm, (uint32_t)n, (const rt_hit *)ctx->d_bin_hits, d_t64 ? (const double *)ctx->d_bin_t64 : nullptr,
		                                       (rt_hit *)d_hits, (double *)d_t64, s);
		bins_release(ctx, s);
	} else {
		e = launch_trace_closest(ctx->geom, (const rt_ray *)d_rays, n, (rt_hit *)d_hits, (double *)d_t64, s);
	}
	ctx->profiler.end(s);
	if (ctx->fail_hip(e, "closest-hit launch")) return RT_ERR_HIP;
	return RT_OK;
}

void rt_accept_region_box(const double a[3], const double b[3], const double c[3], const double nf[3], double lo[3], double hi[3])
{
	accept_region_box(a, b, c, nf, lo, hi);
}

int rt_cut_references(const rt_mesh *mesh, double budget, uint32_t min_tris, rt_reference *first, uint64_t *n_first, rt_reference *extra, uint64_t *n_extra)
try {
	static_assert(sizeof(rt_reference) == sizeof(SplitRef), "the same record");
	if (!mesh || !n_first || !n_extra || !(budget >= 0.0)) return RT_ERR_INVALID;
	HostScene hs;
	std::string err;
	if (!hs.build(*mesh, err)) return RT_ERR_MESH;
	PresplitOptions po;
	po.budget = (float)budget;
	po.min_tris = min_tris;
	Presplit cut;
	presplit(hs, po, cut);
	if (first && *n_first >= cut.first.size()) std::memcpy(first, cut.first.data(), cut.first.size() * sizeof(SplitRef));
	if (extra && *n_extra >= cut.extra.size()) std::memcpy(extra, cut.extra.data(), cut.extra.size() * sizeof(SplitRef));
	*n_first = cut.first.size();
	*n_extra = cut.extra.size();
	return RT_OK;
} catch (...) { return RT_ERR_INVALID; }

int rt_trace_closest_device(rt_context *ctx, const void *d_rays, uint64_t n, void *d_hits, void *d_t64, void *stream)
try {
	return trace_closest_device(ctx, d_rays, n, d_hits, d_t64, stream, true);
} RT_CATCH(ctx)

int rt_trace_any_device(rt_context *ctx, const void *d_rays, uint64_t n, void *d_out, void *stream)
try {
	if (!ctx || (n && (!d_rays || !d_out))) return set_err(ctx, RT_ERR_INVALID, "rt_trace_any_device: NULL argument");
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_trace_any_device: no mesh uploaded");
	if (!use_device(ctx)) return RT_ERR_HIP;
	hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
	const bool bins = bins_wanted(ctx, n); // as for the closest hit; the ordered answers (a byte per ray) use the buffer of the ordered hits
	if (bins && (!ensure_bins(ctx, n) || !bins_acquire(ctx, s))) return RT_ERR_HIP;
	ctx->profiler.begin(Profiler::ANY, n, s);
	hipError_t e;
	if (bins) {
		e = bin_rays(ctx->geom.nodes, (const rt_ray *)d_rays, (uint32_t)n, (rt_ray *)ctx->d_bin_rays, (uint32_t *)ctx->d_bin_perm, (uint32_t *)ctx->d_bin_counters, s);
		if (e == hipSuccess) e = launch_trace_any(ctx->geom, (const rt_ray *)ctx->d_bin_rays, n, (uint8_t *)ctx->d_bin_hits, s);
		if (e == hipSuccess) e = unbin_bytes((const uint32_t *)ctx->d_bin_perm, (uint32_t)n, (const uint8_t *)ctx->d_bin_hits, (uint8_t *)d_out, s);
		bins_release(ctx, s);
	} else {
		e = launch_trace_any(ctx->geom, (const rt_ray *)d_rays, n, (uint8_t *)d_out, s);
	}
	ctx->profiler.end(s);
	if (ctx->fail_hip(e, "any-hit launch")) return RT_ERR_HIP;
	return RT_OK;
} RT_CATCH(ctx)

int rt_trace_closest(rt_context *ctx, const rt_ray *rays, uint64_t n, rt_hit *hits, double *t64)
try {
	if (!ctx || (n && (!rays || !hits))) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest: NULL argument");
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest: no mesh uploaded");
	if (n == 0) return RT_OK;
	if (!use_device(ctx) || !ensure_stage(ctx, n)) return RT_ERR_HIP;
	// the caller's buffers are pageable memory: rays and results cross the link through the context's pinned pieces, with host
	// threads copying one piece while the other is in flight
	if (n <= kHostPart) { // one part: nothing to overlap
		if (stage_rays_in(ctx, rays, n)) return RT_ERR_HIP;
		int rc = rt_trace_closest_device(ctx, ctx->d_stage_rays, n, ctx->d_stage_out, t64 ? ctx->d_stage_t64 : nullptr, nullptr);
		if (rc != RT_OK) return rc;
		if (ctx->fail_hip(download_staged(ctx->pinned, ctx->d_stage_out, n * sizeof(rt_hit), hits, ctx->stream), "closest-hit kernel / D2H hits")) return RT_ERR_HIP;
		if (t64 && ctx->fail_hip(download_staged(ctx->pinned, ctx->d_stage_t64, n * sizeof(double), t64, ctx->stream), "D2H t64")) return RT_ERR_HIP;
		return RT_OK;
	}
	// A large batch goes in parts of kHostPart rays.  This thread brings the rays of part k in (copy_stream) and launches its kernel
	// (the context's stream); a helper thread waits for each part's kernel and brings its results back (down_stream, pinned pieces
	// of its own).  Both directions of the link and the kernel then work at the same time, and the call takes about as long as
	// its longest stage, the rays on their way in.
	if (!ctx->copy_stream && ctx->fail_hip(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking), "hipStreamCreate")) return RT_ERR_HIP;
	if (!ctx->down_stream && ctx->fail_hip(hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking), "hipStreamCreate")) return RT_ERR_HIP;
	const uint64_t n_parts = (n + kHostPart - 1) / kHostPart;
	std::vector<hipEvent_t> traced(n_parts, nullptr);
	auto drop_events = [&] { for (auto ev : traced) if (ev) (void)hipEventDestroy(ev); };
	for (auto &ev : traced)
		if (ctx->fail_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) { drop_events(); return RT_ERR_HIP; }
	auto part_len = [&](uint64_t k) { return std::min<uint64_t>(kHostPart, n - k * kHostPart); };
	auto fetch = [&](uint64_t k) -> hipError_t { // results of part k, once its kernel is through
		const uint64_t at = k * kHostPart, m = part_len(k);
		hipError_t e = hipEventSynchronize(traced[k]);
		if (e == hipSuccess) e = download_staged(ctx->pinned_down, static_cast<rt_hit *>(ctx->d_stage_out) + at, m * sizeof(rt_hit), hits + at, ctx->down_stream);
		if (e == hipSuccess && t64) e = download_staged(ctx->pinned_down, static_cast<double *>(ctx->d_stage_t64) + at, m * sizeof(double), t64 + at, ctx->down_stream);
		return e;
	};
	std::atomic<uint64_t> recorded{0}; // parts whose kernel has been launched and marked
	std::atomic<bool> give_up{false};
	hipError_t down_error = hipSuccess;
	const int device = ctx->device;
	auto bring_back = [&] {
		hipError_t e = hipSetDevice(device);
		for (uint64_t k = 0; k < n_parts && e == hipSuccess; k++) {
			while (recorded.load(std::memory_order_acquire) <= k && !give_up.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
			if (recorded.load(std::memory_order_acquire) <= k) break; // the launching side gave up
			e = fetch(k);
		}
		down_error = e;
	};
	std::thread helper;
	bool threaded = true;
	try { helper = std::thread(bring_back); } catch (const std::system_error &) { threaded = false; } // no thread to be had: this thread fetches between the parts
	struct JoinOnExit { // an exception on this side (out of memory) must not leave the helper running with references into this frame
		std::thread &t; std::atomic<bool> &stop;
		~JoinOnExit() { if (t.joinable()) { stop.store(true, std::memory_order_release); t.join(); } }
	} join_on_exit{helper, give_up};
	int rc = RT_OK;
	for (uint64_t k = 0; k < n_parts && rc == RT_OK; k++) {
		const uint64_t at = k * kHostPart, m = part_len(k);
		const char *from = reinterpret_cast<const char *>(rays + at);
		if (ctx->fail_hip(upload_generated(ctx->pinned, static_cast<rt_ray *>(ctx->d_stage_rays) + at, m, sizeof(rt_ray), ctx->copy_stream,
		                                   [from](size_t b, size_t e, char *out) { std::memcpy(out, from + b * sizeof(rt_ray), (e - b) * sizeof(rt_ray)); }), "H2D rays")) { rc = RT_ERR_HIP; break; }
		rc = rt_trace_closest_device(ctx, static_cast<rt_ray *>(ctx->d_stage_rays) + at, m, static_cast<rt_hit *>(ctx->d_stage_out) + at,
		                             t64 ? static_cast<double *>(ctx->d_stage_t64) + at : nullptr, nullptr);
		if (rc == RT_OK && ctx->fail_hip(hipEventRecord(traced[k], ctx->stream), "hipEventRecord")) rc = RT_ERR_HIP;
		if (rc != RT_OK) break;
		recorded.store(k + 1, std::memory_order_release);
		if (!threaded && k >= 1 && ctx->fail_hip(fetch(k - 1), "closest-hit kernel / D2H")) rc = RT_ERR_HIP;
	}
	if (rc != RT_OK) give_up.store(true, std::memory_order_release);
	if (threaded) {
		helper.join();
		if (rc == RT_OK && ctx->fail_hip(down_error, "closest-hit kernel / D2H")) rc = RT_ERR_HIP;
	} else if (rc == RT_OK && ctx->fail_hip(fetch(n_parts - 1), "closest-hit kernel / D2H")) {
		rc = RT_ERR_HIP;
	}
	(void)hipStreamSynchronize(ctx->stream); // also after an error: nothing of this call may still run when the events go
	(void)hipStreamSynchronize(ctx->copy_stream);
	(void)hipStreamSynchronize(ctx->down_stream);
	drop_events();
	return rc;
} RT_CATCH(ctx)

int rt_trace_closest_static(rt_context *ctx, const rt_ray *rays, uint64_t n, rt_hit *hits, double *t64)
try {
	if (!ctx || (n && (!rays || !hits))) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest_static: NULL argument");
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest_static: no mesh uploaded");
	if (n > kStaticRays)
		return set_err(ctx, RT_ERR_INVALID, "rt_trace_closest_static: " + std::to_string(n) + " rays exceed the static staging area of " +
		                                     std::to_string(kStaticRays) + " rays (RT_STATIC_MAX_RAYS); split the batch or use rt_trace_closest");
	if (n == 0) return RT_OK;
	if (!use_device(ctx)) return RT_ERR_HIP;
	StaticStage st;
	if (ctx->fail_hip(static_stage(ctx->device, st), "static staging")) return RT_ERR_HIP;
	// nothing in the staging area has to start from zero: the kernel writes hits[i] and t64[i] for every i < n, and only those are copied back
	if (ctx->fail_hip(hipMemcpyAsync(st.rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream), "H2D rays")) return RT_ERR_HIP;
	int rc = trace_closest_device(ctx, st.rays, n, st.hits, t64 ? st.t64 : nullptr, nullptr, false);
	if (rc != RT_OK) return rc;
	if (ctx->fail_hip(hipMemcpyAsync(hits, st.hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream), "D2H hits")) return RT_ERR_HIP;
	if (t64 && ctx->fail_hip(hipMemcpyAsync(t64, st.t64, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "D2H t64")) return RT_ERR_HIP;
	if (ctx->fail_hip(hipStreamSynchronize(ctx->stream), "closest-hit kernel")) return RT_ERR_HIP;
	return RT_OK;
} RT_CATCH(ctx)

int rt_trace_any(rt_context *ctx, const rt_ray *rays, uint64_t n, uint8_t *out)
try {
	if (!ctx || (n && (!rays || !out))) return set_err(ctx, RT_ERR_INVALID, "rt_trace_any: NULL argument");
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_trace_any: no mesh uploaded");
	if (n == 0) return RT_OK;
	if (!use_device(ctx) || !ensure_stage(ctx, n)) return RT_ERR_HIP;
	if (stage_rays_in(ctx, rays, n)) return RT_ERR_HIP;
	int rc = rt_trace_any_device(ctx, ctx->d_stage_rays, n, ctx->d_stage_out, nullptr);
	if (rc != RT_OK) return rc;
	if (ctx->fail_hip(download_staged(ctx->pinned, ctx->d_stage_out, n, out, ctx->stream), "any-hit kernel / D2H")) return RT_ERR_HIP;
	return RT_OK;
} RT_CATCH(ctx)

int rt_render_begin(rt_context *ctx, int nx, int ny, int ns, int spectral_samples, int algorithm, uint64_t seed, void *d_film)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!ctx->has_mesh || !ctx->has_materials || !ctx->has_camera)
		return set_err(ctx, RT_ERR_INVALID, "rt_render_begin: mesh, materials and camera must be set first");
	if (nx <= 0 || ny <= 0 || ns <= 0 || spectral_samples <= 0 || spectral_samples > 65535)
		return set_err(ctx, RT_ERR_INVALID, "rt_render_begin: nx, ny, ns must be positive and spectral_samples in 1..65535");
	if (algorithm < RT_ALG_NAIVE || algorithm > RT_ALG_MIS_BDPT)
		return set_err(ctx, RT_ERR_INVALID, "rt_render_begin: unknown algorithm");
	if (!use_device(ctx)) return RT_ERR_HIP;
	std::string err;
	hipError_t e = render_begin(ctx->render, ctx->camera, nx, ny, ns, spectral_samples, algorithm, seed, d_film, ctx->stream, err);
	if (e != hipSuccess) return set_err(ctx, e == hipErrorInvalidValue ? RT_ERR_INVALID : RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH(ctx)

int rt_render_pass(rt_context *ctx, int first_sample, int n_samples)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!ctx->render.begun) return set_err(ctx, RT_ERR_INVALID, "rt_render_pass: call rt_render_begin first");
	// the tables the kernels index (tree, triangles, materials, lights) must be the ones rt_render_begin saw
	if (!ctx->has_mesh || !ctx->has_materials) return set_err(ctx, RT_ERR_INVALID, "rt_render_pass: the scene was cleared or its materials were refused after rt_render_begin");
	if (first_sample < 0 || n_samples < 0) return set_err(ctx, RT_ERR_INVALID, "rt_render_pass: negative sample range");
	if (n_samples == 0) return RT_OK;
	if (!use_device(ctx)) return RT_ERR_HIP;
	std::string err;
	if (render_pass(ctx->render, ctx->geom, first_sample, n_samples, ctx->stream, err) != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH(ctx)

int rt_render_sync(rt_context *ctx)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	if (ctx->fail_hip(hipStreamSynchronize(ctx->stream), "render kernels")) return RT_ERR_HIP;
	return RT_OK;
} RT_CATCH(ctx)

int rt_download_film(rt_context *ctx, double *spectrum_img)
try {
	if (!ctx || !spectrum_img) return set_err(ctx, RT_ERR_INVALID, "rt_download_film: NULL argument");
	if (!ctx->render.begun) return set_err(ctx, RT_ERR_INVALID, "rt_download_film: nothing rendered");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	std::string err;
	if (render_download_film(ctx->render, spectrum_img, ctx->stream, err) != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH(ctx)

int rt_download_preview(rt_context *ctx, int samples_done, double *preview_img)
try {
	if (!ctx || !preview_img) return set_err(ctx, RT_ERR_INVALID, "rt_download_preview: NULL argument");
	if (!ctx->render.begun) return set_err(ctx, RT_ERR_INVALID, "rt_download_preview: nothing rendered");
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	std::string err;
	if (render_download_preview(ctx->render, samples_done, preview_img, ctx->stream, err) != hipSuccess) return set_err(ctx, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH(ctx)

int rt_film_device_ptr(rt_context *ctx, void **d_film, size_t *bytes)
try {
	if (!ctx || !d_film) return RT_ERR_INVALID;
	if (!ctx->render.begun) return set_err(ctx, RT_ERR_INVALID, "rt_film_device_ptr: call rt_render_begin first");
	*d_film = ctx->render.d_film;
	if (bytes) *bytes = (size_t)ctx->render.nx * ctx->render.ny * RT_NBINS * sizeof(float);
	return RT_OK;
} RT_CATCH(ctx)

int rt_get_stats(rt_context *ctx, rt_stats *out)
try {
	if (!ctx || !out) return RT_ERR_INVALID;
	if (ctx->render.begun && use_device(ctx)) { if (int rc = finish_pending_paths(ctx)) return rc; }
	std::memset(out, 0, sizeof(*out));
	out->n_faces = ctx->host.n_faces;
	out->n_triangles = ctx->host.n_tris;
	out->n_objects = ctx->host.n_objects;
	out->n_lights = ctx->light_faces.size();
	out->bvh_nodes = ctx->built_on_device ? ctx->device_nodes : ctx->bvh.nodes.size();
	out->bvh_depth = ctx->bvh.depth;
	out->bvh_leaves = ctx->bvh.n_leaves;
	out->bvh_build_seconds = ctx->bvh.build_seconds;
	out->closest_rays = ctx->render.closest_rays;
	out->any_rays = ctx->render.any_rays;
	out->camera_samples = ctx->render.camera_samples;
	out->device_bytes = ctx->device_bytes;
	return RT_OK;
} RT_CATCH(ctx)

int rt_get_bvh2(rt_context *ctx, rt_bvh2_node *nodes, uint64_t *n_nodes, uint32_t *tri_index, uint64_t *n_tris)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_get_bvh2: no mesh uploaded");
	if (ctx->built_on_device) return set_err(ctx, RT_ERR_INVALID, "rt_get_bvh2: the tree was built on the device; only the host builder keeps a BVH2");
	if (n_nodes) *n_nodes = ctx->bvh.nodes.size();
	if (n_tris) *n_tris = ctx->bvh.tri_index.size();
	if (nodes) std::memcpy(nodes, ctx->bvh.nodes.data(), ctx->bvh.nodes.size() * sizeof(rt_bvh2_node));
	if (tri_index) std::memcpy(tri_index, ctx->bvh.tri_index.data(), ctx->bvh.tri_index.size() * sizeof(uint32_t));
	return RT_OK;
} RT_CATCH(ctx)

int rt_get_wide_bvh(rt_context *ctx, rt_wide_node *nodes, uint64_t *n_nodes, uint32_t *tri_index, uint64_t *n_tris)
try {
	static_assert(sizeof(rt_wide_node) == sizeof(WideNode), "the public record is the device record");
	if (!ctx) return RT_ERR_INVALID;
	if (!ctx->has_mesh) return set_err(ctx, RT_ERR_INVALID, "rt_get_wide_bvh: no mesh uploaded");
	if (n_nodes) *n_nodes = ctx->device_nodes;
	if (n_tris) *n_tris = ctx->wide.tri_index.size();
	if (nodes) { // from HBM: the device builder leaves no host copy
		if (!use_device(ctx)) return RT_ERR_HIP;
		if (ctx->fail_hip(hipMemcpy(nodes, ctx->d_nodes, ctx->device_nodes * sizeof(WideNode), hipMemcpyDeviceToHost), "hipMemcpy(wide nodes)")) return RT_ERR_HIP;
	}
	if (tri_index) std::memcpy(tri_index, ctx->wide.tri_index.data(), ctx->wide.tri_index.size() * sizeof(uint32_t));
	return RT_OK;
} RT_CATCH(ctx)

int rt_filter_bilateral_device(rt_context *ctx, const void *d_rgba, int width, int height, int window, double sigma_d, double sigma_r, void *d_out)
try {
	if (!ctx || !d_rgba || !d_out) return set_err(ctx, RT_ERR_INVALID, "rt_filter_bilateral: NULL argument");
	if (width <= 0 || height <= 0 || window < 0 || window > 64) return set_err(ctx, RT_ERR_INVALID, "rt_filter_bilateral: needs width, height > 0 and 0 <= window <= 64");
	if (!use_device(ctx)) return RT_ERR_HIP;
	hipError_t e = launch_bilateral((const double *)d_rgba, width, height, window, sigma_d, sigma_r, (double *)d_out, ctx->stream);
	if (ctx->fail_hip(e, "k_bilateral")) return RT_ERR_HIP;
	return RT_OK;
} RT_CATCH(ctx)

int rt_filter_bilateral(rt_context *ctx, const double *rgba, int width, int height, int window, double sigma_d, double sigma_r, double *out)
try {
	if (!ctx || !rgba || !out) return set_err(ctx, RT_ERR_INVALID, "rt_filter_bilateral: NULL argument");
	if (width <= 0 || height <= 0) return set_err(ctx, RT_ERR_INVALID, "rt_filter_bilateral: needs width, height > 0");
	if (!use_device(ctx)) return RT_ERR_HIP;
	const size_t bytes = (size_t)width * height * 4 * sizeof(double);
	void *d_in = nullptr, *d_out = nullptr;
	int rc = RT_OK;
	if (ctx->fail_hip(hipMalloc(&d_in, bytes), "hipMalloc") || ctx->fail_hip(hipMalloc(&d_out, bytes), "hipMalloc") ||
	    ctx->fail_hip(hipMemcpy(d_in, rgba, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"))
		rc = RT_ERR_HIP;
	if (rc == RT_OK) rc = rt_filter_bilateral_device(ctx, d_in, width, height, window, sigma_d, sigma_r, d_out);
	if (rc == RT_OK && (ctx->fail_hip(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize") ||
	                    ctx->fail_hip(hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H")))
		rc = RT_ERR_HIP;
	if (d_in) (void)hipFree(d_in);
	if (d_out) (void)hipFree(d_out);
	return rc;
} RT_CATCH(ctx)

void *rt_stream(rt_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---- groups: one context per GPU (or several on one), samples of a pass split between them
} // extern "C"

struct rt_group {
	std::vector<rt_context *> members;
	// on the first member's device: the summed film, and one slice per member on another device for its film in transit
	float *d_sum = nullptr, *d_stage = nullptr;
	size_t sum_elems = 0, stage_slices = 0;
	std::vector<hipEvent_t> arrived; // per member: its film has reached its slice (created on first use, on the member's device)
	double last_sum_ms = 0.0;        // how long the last gather and sum took, on the first member's stream
	std::string error;
};

namespace {
int caught_group(rt_group *g, const char *what) noexcept
{
	try {
		if (g) g->error = std::string("host side: ") + what;
		else g_create_error = std::string("host side: ") + what;
	} catch (...) {
	}
	return RT_ERR_HOST;
}

// runs f(member index) for every member on a thread of its own (a member's calls block on its GPU) and returns the first failure
template <class F>
int for_each_member(rt_group *g, F f)
{
	const size_t n = g->members.size();
	std::vector<int> rc(n, RT_OK);
	if (n == 1) rc[0] = f(0);
	else {
		// one host thread per member; members whose thread cannot be started are served by the calling thread, after the others
		// have been joined (a vector of joinable threads must never be destroyed: that ends the process)
		std::vector<std::thread> th;
		th.reserve(n);
		size_t started = 0;
		try {
			for (; started < n; started++) th.emplace_back([&rc, &f, i = started] { rc[i] = f(i); });
		} catch (const std::system_error &) {
		}
		for (auto &t : th) t.join();
		for (size_t i = started; i < n; i++) rc[i] = f(i);
	}
	for (size_t i = 0; i < n; i++)
		if (rc[i] != RT_OK) {
			g->error = "member " + std::to_string(i) + " (device " + std::to_string(g->members[i]->device) + "): " + rt_last_error(g->members[i]);
			return rc[i];
		}
	return RT_OK;
}

// the i-th of n blocks of `count` samples starting at `first` (blocks differ by at most one sample and tile the range)
void member_samples(int i, int n, int first, int count, int &m_first, int &m_count)
{
	const int base = count / n, rem = count % n;
	m_first = first + i * base + std::min(i, rem);
	m_count = base + (i < rem ? 1 : 0);
}

int group_fail(rt_group *g, int code, const std::string &msg)
{
	g->error = msg;
	return code;
}

// Sums the member films into g->d_sum on the first member's device; every member must have finished (rt_render_sync).
// Every member on another device pushes its film over its own xGMI link into a slice of its own on the first member's device, on
// its own stream, all at the same time (the links are point to point: seven copies in turn would use one link at a time); the
// first member's stream waits for the arrivals and one kernel adds all films up.  Members on the first member's device are read
// where they lie.
int group_sum_films(rt_group *g)
{
	rt_context *root = g->members[0];
	if (!root->render.begun) return group_fail(g, RT_ERR_INVALID, "rt_group: nothing rendered");
	const size_t elems = (size_t)root->render.nx * root->render.ny * RT_NBINS, bytes = elems * sizeof(float);
	size_t remote = 0;
	for (rt_context *m : g->members) remote += m->device != root->device ? 1 : 0;
	hipError_t e = hipSetDevice(root->device);
	if (e == hipSuccess && (g->sum_elems != elems || g->stage_slices < remote)) {
		if (g->d_sum) (void)hipFree(g->d_sum);
		if (g->d_stage) (void)hipFree(g->d_stage);
		g->d_sum = g->d_stage = nullptr;
		g->sum_elems = g->stage_slices = 0;
		e = hipMalloc((void **)&g->d_sum, bytes);
		if (e == hipSuccess && remote) e = hipMalloc((void **)&g->d_stage, bytes * remote);
		if (e == hipSuccess) { g->sum_elems = elems; g->stage_slices = remote; }
	}
	if (g->arrived.size() < g->members.size()) g->arrived.resize(g->members.size(), nullptr);
	hipEvent_t t0 = nullptr, t1 = nullptr;
	if (e == hipSuccess) e = hipEventCreate(&t0);
	if (e == hipSuccess) e = hipEventCreate(&t1);
	if (e == hipSuccess) e = hipEventRecord(t0, root->stream);
	const float *src[kMaxFilmSources];
	size_t n_src = 0, slice = 0;
	for (size_t i = 0; i < g->members.size() && e == hipSuccess; i++) {
		rt_context *m = g->members[i];
		if (m->device == root->device) { src[n_src++] = m->render.d_film; continue; }
		float *dst = g->d_stage + slice++ * elems;
		e = hipSetDevice(m->device); // the copy is the sender's: it goes on the member's stream, behind the member's rendering
		if (e == hipSuccess && !g->arrived[i]) e = hipEventCreateWithFlags(&g->arrived[i], hipEventDisableTiming);
		if (e == hipSuccess) e = hipMemcpyPeerAsync(dst, root->device, m->render.d_film, m->device, bytes, m->stream);
		if (e == hipSuccess) e = hipEventRecord(g->arrived[i], m->stream);
		if (e == hipSuccess) e = hipSetDevice(root->device);
		if (e == hipSuccess) e = hipStreamWaitEvent(root->stream, g->arrived[i], 0);
		src[n_src++] = dst;
	}
	(void)hipSetDevice(root->device);
	if (e == hipSuccess) e = render_sum_films(g->d_sum, src, n_src, elems, root->stream);
	if (e == hipSuccess) e = hipEventRecord(t1, root->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(root->stream);
	float ms = 0.0f;
	if (e == hipSuccess && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) g->last_sum_ms = ms;
	if (t0) (void)hipEventDestroy(t0);
	if (t1) (void)hipEventDestroy(t1);
	if (e != hipSuccess) {
		for (rt_context *m : g->members) { (void)hipSetDevice(m->device); (void)hipStreamSynchronize(m->stream); } // nothing may still be in flight into the slices
		(void)hipSetDevice(root->device);
		return group_fail(g, RT_ERR_HIP, std::string("rt_group: summing the member films: ") + hipGetErrorString(e));
	}
	return RT_OK;
}
} // namespace

extern "C" {

int rt_group_create(const int *devices, int n, rt_group **out)
try {
	if (!out) return set_err(nullptr, RT_ERR_INVALID, "rt_group_create: out is NULL");
	*out = nullptr;
	if (!devices || n <= 0 || n > 64) return set_err(nullptr, RT_ERR_INVALID, "rt_group_create: needs 1..64 device ordinals");
	rt_group *g = new (std::nothrow) rt_group();
	if (!g) return set_err(nullptr, RT_ERR_INVALID, "rt_group_create: out of host memory");
	for (int i = 0; i < n; i++) {
		rt_context *c = nullptr;
		int rc = rt_create(devices[i], &c);
		if (rc != RT_OK) { // rt_last_error(NULL) holds the text
			rt_group_destroy(g);
			return rc;
		}
		g->members.push_back(c);
	}
	*out = g;
	return RT_OK;
} RT_CATCH_CREATE

void rt_group_destroy(rt_group *g)
{
	if (!g) return;
	if (!g->members.empty()) {
		(void)hipSetDevice(g->members[0]->device);
		if (g->d_sum) (void)hipFree(g->d_sum);
		if (g->d_stage) (void)hipFree(g->d_stage);
	}
	for (size_t i = 0; i < g->arrived.size(); i++)
		if (g->arrived[i]) { (void)hipSetDevice(g->members[i]->device); (void)hipEventDestroy(g->arrived[i]); }
	for (rt_context *c : g->members) rt_destroy(c);
	delete g;
}

const char *rt_group_last_error(const rt_group *g) { return g ? g->error.c_str() : g_create_error.c_str(); }
int rt_group_size(const rt_group *g) { return g ? (int)g->members.size() : 0; }
rt_context *rt_group_context(rt_group *g, int i) { return (g && i >= 0 && (size_t)i < g->members.size()) ? g->members[i] : nullptr; }

int rt_group_upload_mesh(rt_group *g, const rt_mesh *mesh)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_upload_mesh(g->members[i], mesh); });
} RT_CATCH_GROUP(g)
int rt_group_upload_materials(rt_group *g, const rt_material *per_object, uint32_t n_objects)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_upload_materials(g->members[i], per_object, n_objects); });
} RT_CATCH_GROUP(g)
int rt_group_upload_material_graphs(rt_group *g, const rt_material_graphs *graphs)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_upload_material_graphs(g->members[i], graphs); });
} RT_CATCH_GROUP(g)
int rt_group_set_environment(rt_group *g, const uint8_t *texels, int width, int height, int bpp, double brightness)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_set_environment(g->members[i], texels, width, height, bpp, brightness); });
} RT_CATCH_GROUP(g)
int rt_group_set_camera(rt_group *g, const rt_camera *cam)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_set_camera(g->members[i], cam); });
} RT_CATCH_GROUP(g)
int rt_group_set_triangle_test(rt_group *g, int mode)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_set_triangle_test(g->members[i], mode); });
} RT_CATCH_GROUP(g)
int rt_group_clear(rt_group *g)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_clear(g->members[i]); });
} RT_CATCH_GROUP(g)
int rt_group_render_begin(rt_group *g, int nx, int ny, int ns, int spectral_samples, int algorithm, uint64_t seed)
try {
	if (!g) return RT_ERR_INVALID;
	// every member gets the whole render's ns (the film weight is 1 / ns per sample) and the same seed: streams are keyed by (pixel, sample)
	return for_each_member(g, [&](size_t i) { return rt_render_begin(g->members[i], nx, ny, ns, spectral_samples, algorithm, seed, nullptr); });
} RT_CATCH_GROUP(g)
int rt_group_render_pass(rt_group *g, int first_sample, int n_samples)
try {
	if (!g) return RT_ERR_INVALID;
	if (first_sample < 0 || n_samples < 0) return group_fail(g, RT_ERR_INVALID, "rt_group_render_pass: negative sample range");
	const int n = (int)g->members.size();
	return for_each_member(g, [&](size_t i) {
		int f, c;
		member_samples((int)i, n, first_sample, n_samples, f, c);
		return rt_render_pass(g->members[i], f, c);
	});
} RT_CATCH_GROUP(g)
int rt_group_render_sync(rt_group *g)
try {
	if (!g) return RT_ERR_INVALID;
	return for_each_member(g, [&](size_t i) { return rt_render_sync(g->members[i]); });
} RT_CATCH_GROUP(g)
int rt_group_download_film(rt_group *g, double *spectrum_img)
try {
	if (!g || !spectrum_img) return RT_ERR_INVALID;
	if (int rc = rt_group_render_sync(g)) return rc;
	if (int rc = group_sum_films(g)) return rc;
	rt_context *root = g->members[0];
	std::string err;
	if (render_download_film(root->render, spectrum_img, root->stream, err, g->d_sum) != hipSuccess) return group_fail(g, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH_GROUP(g)
int rt_group_download_preview(rt_group *g, int samples_done, double *preview_img)
try {
	if (!g || !preview_img) return RT_ERR_INVALID;
	if (int rc = rt_group_render_sync(g)) return rc;
	if (int rc = group_sum_films(g)) return rc;
	rt_context *root = g->members[0];
	std::string err;
	if (render_download_preview(root->render, samples_done, preview_img, root->stream, err, g->d_sum) != hipSuccess) return group_fail(g, RT_ERR_HIP, err);
	return RT_OK;
} RT_CATCH_GROUP(g)

double rt_group_last_sum_ms(const rt_group *g) { return g ? g->last_sum_ms : 0.0; }

// The traversal benchmark on a group: member i answers the i-th of n contiguous ranges of the rays on its own replica of the scene
// (host buffers, like rt_trace_closest); no member needs anything from another.
int rt_group_trace_closest(rt_group *g, const rt_ray *rays, uint64_t n, rt_hit *hits, double *t64)
try {
	if (!g || (n && (!rays || !hits))) return RT_ERR_INVALID;
	const uint64_t parts = g->members.size();
	return for_each_member(g, [&](size_t i) {
		const uint64_t lo = n * i / parts, hi = n * (i + 1) / parts;
		if (hi == lo) return (int)RT_OK;
		return rt_trace_closest(g->members[i], rays + lo, hi - lo, hits + lo, t64 ? t64 + lo : nullptr);
	});
} RT_CATCH_GROUP(g)

int rt_profile_start(rt_context *ctx)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (!use_device(ctx)) return RT_ERR_HIP;
	ctx->profiler.clear();
	ctx->profiler.enabled = true;
	return RT_OK;
} RT_CATCH(ctx)

int rt_profile_stop(rt_context *ctx, rt_profile *out)
try {
	if (!ctx || !out) return RT_ERR_INVALID;
	if (!use_device(ctx)) return RT_ERR_HIP;
	if (int rc = finish_pending_paths(ctx)) return rc;
	ctx->profiler.enabled = false;
	ctx->profiler.collect(*out);
	return RT_OK;
} RT_CATCH(ctx)

int rt_render_capture_rays(rt_context *ctx, rt_ray *buf, uint64_t capacity, uint64_t *n_captured)
try {
	if (!ctx) return RT_ERR_INVALID;
	if (ctx->render.begun && use_device(ctx)) { if (int rc = finish_pending_paths(ctx)) return rc; }
	ctx->render.capture_buf = buf;
	ctx->render.capture_capacity = buf ? capacity : 0;
	ctx->render.capture_count = 0;
	ctx->render.capture_count_out = buf ? n_captured : nullptr;
	if (n_captured) *n_captured = 0;
	return RT_OK;
} RT_CATCH(ctx)

} // extern "C"
