// Wavefront integrator state and entry points (implemented in render.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "device_scene.h"
#include "host_scene.h"
#include "mi355rt.h"
#include "staged_copy.h"

namespace mi355rt {

// HIP-event timing of individual kernel launches on the stream they run on.
struct Profiler {
	enum Kind { CLOSEST = 0, ANY, GENERATE, SHADE, CONNECT, N_KINDS };
	struct Rec { int kind; uint64_t items; hipEvent_t start, stop; };
	bool enabled = false;
	std::vector<Rec> recs;
	std::vector<hipEvent_t> pool; // events free for reuse
	hipEvent_t get()
	{
		if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
		hipEvent_t e = nullptr;
		(void)hipEventCreate(&e);
		return e;
	}
	// call begin() right before a launch and end() right after it
	void begin(int kind, uint64_t items, hipStream_t s)
	{
		if (!enabled) return;
		Rec r{kind, items, get(), get()};
		(void)hipEventRecord(r.start, s);
		recs.push_back(r);
	}
	void end(hipStream_t s)
	{
		if (!enabled) return;
		(void)hipEventRecord(recs.back().stop, s);
	}
	// a launch whose item count lives on the device is recorded with a bound and corrected once the count is read back
	long last() const { return enabled ? (long)recs.size() - 1 : -1; }
	void set_items(long idx, uint64_t items) { if (idx >= 0 && (size_t)idx < recs.size()) recs[idx].items = items; }
	void clear()
	{
		for (auto &r : recs) { pool.push_back(r.start); pool.push_back(r.stop); }
		recs.clear();
	}
	void collect(rt_profile &out)
	{
		rt_kernel_time *k[N_KINDS] = {&out.closest, &out.any, &out.generate, &out.shade, &out.connect};
		for (int i = 0; i < N_KINDS; i++) *k[i] = rt_kernel_time{0.0, 0, 0};
		for (auto &r : recs) {
			(void)hipEventSynchronize(r.stop);
			float ms = 0.0f;
			(void)hipEventElapsedTime(&ms, r.start, r.stop);
			k[r.kind]->ms += ms;
			k[r.kind]->launches++;
			k[r.kind]->items += r.items;
		}
		clear();
	}
	void destroy()
	{
		clear();
		for (auto e : pool) (void)hipEventDestroy(e);
		pool.clear();
	}
};

struct RenderState {
	// film
	int nx = 0, ny = 0, ns = 0, spectral_samples = 0, algorithm = 0;
	uint64_t seed = 0;
	float *d_film = nullptr;
	bool film_owned = false;
	bool begun = false;
	// counters since rt_render_begin
	uint64_t closest_rays = 0, any_rays = 0, camera_samples = 0;
	Profiler *profiler = nullptr;   // owned by the context
	// diagnostic ray capture
	rt_ray *capture_buf = nullptr;
	uint64_t capture_capacity = 0, capture_count = 0;
	uint64_t *capture_count_out = nullptr;
	// opaque device allocations managed by render.hip
	void *impl = nullptr;
};

// upload of shading data: per-triangle float attributes, materials, lights, camera/band tables
// tri_leaf: fan triangle -> leaf-order index; d_tri_face / d_face_object: the geometry's device tables
hipError_t render_upload_scene(RenderState &rs, const HostScene &host, const std::vector<uint32_t> &tri_leaf,
                               const uint32_t *d_tri_face, const uint32_t *d_face_object, PinnedPieces &pinned, hipStream_t stream, std::string &err);
// per-hit material programs for slot a of some objects (validated by the caller); nullptr or no programs removes them
hipError_t render_upload_material_graphs(RenderState &rs, const rt_material_graphs *g, hipStream_t stream, std::string &err);
// environment texture of the naive integrator (renderer.cpp:175-193); texels == nullptr removes it
hipError_t render_set_environment(RenderState &rs, const uint8_t *texels, int width, int height, int bpp, double brightness, hipStream_t stream, std::string &err);
hipError_t render_upload_materials(RenderState &rs, const HostScene &host, const std::vector<rt_material> &mats,
                                   const std::vector<uint32_t> &light_faces, hipStream_t stream, std::string &err);
hipError_t render_begin(RenderState &rs, const rt_camera &cam, int nx, int ny, int ns, int spectral_samples, int algorithm,
                        uint64_t seed, void *d_film, hipStream_t stream, std::string &err);
hipError_t render_pass(RenderState &rs, const DeviceGeom &geom, int first_sample, int n_samples, hipStream_t stream, std::string &err);
// runs the paths that render_pass left for its next batch to completion (a no-op when there are none)
hipError_t render_flush(RenderState &rs, const DeviceGeom &geom, hipStream_t stream, std::string &err);
// film: another film of the same size on the same device to read instead of the context's own (the sum over a group of contexts)
hipError_t render_download_film(RenderState &rs, double *out, hipStream_t stream, std::string &err, const float *film = nullptr);
hipError_t render_download_preview(RenderState &rs, int samples_done, double *out, hipStream_t stream, std::string &err, const float *film = nullptr);
// dst += src over n film elements (both on the current device)
hipError_t render_accumulate_film(float *dst, const float *src, size_t n, hipStream_t stream);
// dst = src[0] + ... + src[n_src - 1] over n film elements in one pass (a group's member films; at most kMaxFilmSources)
constexpr int kMaxFilmSources = 64;
hipError_t render_sum_films(float *dst, const float *const *src, size_t n_src, size_t n, hipStream_t stream);
void render_free(RenderState &rs);

} // namespace mi355rt
