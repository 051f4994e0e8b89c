// BVH construction on the device (see device_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace mi355rt {

// everything the traversal kernel needs, in HBM, owned by the caller after a successful build (hipFree)
struct DeviceBvh {
	void *nodes = nullptr;        // WideNode[n_nodes]
	uint32_t *leaf_tri = nullptr; // leaf order -> fan triangle
	uint32_t *tri_leaf = nullptr; // fan triangle -> leaf order
	void *tri_a = nullptr;        // leaf-ordered records: v0, face normal (6 doubles)
	void *tri_b = nullptr;        // leaf-ordered records: v1, v2 (6 doubles)
	uint32_t n_nodes = 0;
	uint32_t depth = 0;           // nodes on the longest root-to-leaf path
	double seconds = 0.0;         // wall time of the build, device work included
};

// d_tri_a_fan / d_tri_b_fan: the same 6-double records in fan-triangle order, already on the device
hipError_t build_bvh_device(const double *d_tri_a_fan, const double *d_tri_b_fan, uint32_t n, hipStream_t stream, DeviceBvh &out, std::string &err);

} // namespace mi355rt
