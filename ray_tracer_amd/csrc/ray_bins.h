// Ordering of a large batch of unordered rays by the cell of their origin, for scenes far larger than the caches.
// The closest-hit kernel gives every XCD (each with an L2 of its own) one contiguous eighth of a launch's rays: when the rays of an
// eighth start in the same part of the scene, that L2 holds the part of the tree they walk.
// 64 cells in Morton order are enough (measured on a 10M-triangle soup with 16 M rays: 64 cells +13 %, 4096 cells +12 %, 2^30 cells
// +11 %; a 1M-triangle soup gains nothing), and few cells keep the runs a workgroup writes into each cell long.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "mi355rt.h"

namespace mi355rt {

constexpr int kRayBins = 64;

// counters: 2 * kRayBins words of device memory.  sorted[perm[i]] = rays[i]; the order inside a cell is not defined
// n_dev: when not null the number of rays is min(n, *n_dev), read on the device (a queue whose length the host knows only a bound of)
hipError_t bin_rays(const float4 *root_node, const rt_ray *rays, uint32_t n, rt_ray *sorted, uint32_t *perm, uint32_t *counters, hipStream_t stream, const uint32_t *n_dev = nullptr);
// hits[i] = sorted_hits[perm[i]] (and the same for t64 when both pointers are given)
hipError_t unbin_results(const uint32_t *perm, uint32_t n, const rt_hit *sorted_hits, const double *sorted_t64, rt_hit *hits, double *t64, hipStream_t stream, const uint32_t *n_dev = nullptr);
// out[i] = sorted_out[perm[i]]: the any-hit answers, one byte per ray
hipError_t unbin_bytes(const uint32_t *perm, uint32_t n, const uint8_t *sorted_out, uint8_t *out, hipStream_t stream);

} // namespace mi355rt
