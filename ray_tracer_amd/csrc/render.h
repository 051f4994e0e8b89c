// Wavefront integrator state and entry points (implemented in render.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "device_scene.h"
#include "host_scene.h"
#include "mi355rt.h"

namespace mi355rt {

struct RenderState {
	// film
	int nx = 0, ny = 0, ns = 0, spectral_samples = 0, algorithm = 0;
	uint64_t seed = 0;
	float *d_film = nullptr;
	bool film_owned = false;
	bool begun = false;
	// counters since rt_render_begin
	uint64_t closest_rays = 0, any_rays = 0, camera_samples = 0;
	// opaque device allocations managed by render.hip
	void *impl = nullptr;
};

// upload of shading data: per-triangle float attributes, materials, lights, camera/band tables
// tri_leaf: fan triangle -> leaf-order index; d_tri_face / d_face_object: the geometry's device tables
hipError_t render_upload_scene(RenderState &rs, const HostScene &host, const std::vector<uint32_t> &tri_leaf,
                               const uint32_t *d_tri_face, const uint32_t *d_face_object, hipStream_t stream, std::string &err);
hipError_t render_upload_materials(RenderState &rs, const HostScene &host, const std::vector<rt_material> &mats,
                                   const std::vector<uint32_t> &light_faces, hipStream_t stream, std::string &err);
hipError_t render_begin(RenderState &rs, const rt_camera &cam, int nx, int ny, int ns, int spectral_samples, int algorithm,
                        uint64_t seed, void *d_film, hipStream_t stream, std::string &err);
hipError_t render_pass(RenderState &rs, const DeviceGeom &geom, int first_sample, int n_samples, hipStream_t stream, std::string &err);
hipError_t render_download_film(RenderState &rs, double *out, hipStream_t stream, std::string &err);
hipError_t render_download_preview(RenderState &rs, int samples_done, double *out, hipStream_t stream, std::string &err);
void render_free(RenderState &rs);

} // namespace mi355rt
