#pragma once
#include <hip/hip_runtime.h>
#include "device_scene.h"
#include "mi355rt.h"

namespace mi355rt {
// deepest tree the traversal kernels walk (entries of the per-lane stack when it lives in scratch); builders refuse deeper trees
constexpr uint32_t kTraceMaxDepth = 96;
// closest hit over n device-resident rays; d_t64 may be null.  When d_n is not null the kernel reads the ray count
// from it (a count produced by an earlier kernel on the same stream) and n is only an upper bound that sizes the grid.
hipError_t launch_trace_closest(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, rt_hit *d_hits, double *d_t64, hipStream_t stream,
                                const uint32_t *d_n = nullptr);
// any hit: d_out[i] = 1 when some triangle is hit inside [tmin, tmax]
hipError_t launch_trace_any(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, uint8_t *d_out, hipStream_t stream,
                            const uint32_t *d_n = nullptr);

// Staging memory that exists without any allocation: a file-scope device array of the traversal code object, one copy per GPU,
// sized for kStaticRays rays with their hits and fp64 distances.  static_stage() returns the address of the calling device's copy
// (looked up once per device, with that device current) and must be called with `device` current.
constexpr uint64_t kStaticRays = 1ull << 20;
struct StaticStage { rt_ray *rays; rt_hit *hits; double *t64; };
hipError_t static_stage(int device, StaticStage &out);
} // namespace mi355rt
