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
// closest hit over two batches of rays in one launch, each with its own results and its own count (an upper bound on the host, the
// exact count in device memory when d_n_* is not null): one launch has one tail where two have two
hipError_t launch_trace_closest_two(const DeviceGeom &g, const rt_ray *d_rays_a, uint64_t n_a, rt_hit *d_hits_a, const uint32_t *d_n_a,
                                    const rt_ray *d_rays_b, uint64_t n_b, rt_hit *d_hits_b, const uint32_t *d_n_b, hipStream_t stream);
// any hit: d_out[i] = 1 when some triangle is hit inside [tmin, tmax]
hipError_t launch_trace_any(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, uint8_t *d_out, hipStream_t stream,
                            const uint32_t *d_n = nullptr);

// Work counters of the traversal launches.  The waves of a launch are persistent; which 64-ray chunks a wave traces is not fixed at
// launch (fixed shares leave the chip half empty while the slowest waves finish theirs: rays differ in cost by an order of magnitude)
// but taken chunk by chunk from counters in device memory: kWorkGroups counters per XCD, each shared by the workgroups of one group,
// so that no counter sees more than one atomic per few hundred nanoseconds.  Every launch zeroes and uses one of kWorkSets sets;
// launches of one context on different streams may overlap, so a set is reused only behind the launch that used it last (an event,
// waited for on the device).  One per context, allocated by init() with the context's device current; no call after that allocates.
#ifndef TRACE_WORK_GROUPS
#define TRACE_WORK_GROUPS 16
#endif
constexpr int kWorkSets = 32, kWorkGroups = TRACE_WORK_GROUPS, kWorkStride = 16; // counters are kWorkStride uint32 (64 bytes) apart
struct TraceWork {
	uint32_t *d_counters = nullptr; // kWorkSets x 8 XCDs x kWorkGroups counters
	hipEvent_t done[kWorkSets] = {};
	bool used[kWorkSets] = {};
	int next = 0;
	hipError_t init();
	void destroy();
	static constexpr size_t set_words() { return (size_t)8 * kWorkGroups * kWorkStride; }
};

// Staging memory that exists without any allocation: a file-scope device array of the traversal code object, one copy per GPU,
// sized for kStaticRays rays with their hits and fp64 distances.  static_stage() returns the address of the calling device's copy
// (looked up once per device, with that device current) and must be called with `device` current.
constexpr uint64_t kStaticRays = 1ull << 20;
struct StaticStage { rt_ray *rays; rt_hit *hits; double *t64; };
hipError_t static_stage(int device, StaticStage &out);
} // namespace mi355rt
