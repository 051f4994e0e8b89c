// BVH construction on the device (see device_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace mi355rt {

// everything the traversal kernel needs, in HBM, owned by the caller after a successful build (hipFree)
struct DeviceBvh {
	void *nodes = nullptr;        // WideNode[n_nodes]
	uint32_t *leaf_tri = nullptr; // leaf order -> fan triangle, n_refs entries (a triangle whose reference was cut appears once per piece)
	uint32_t *tri_leaf = nullptr; // fan triangle -> leaf order
	void *tri = nullptr;          // leaf-ordered triangle records (device_scene.h), n_refs of them
	uint32_t n_refs = 0;          // leaves of the tree: the triangles plus the additional references of those that were cut
	uint32_t tri_stride = 8;      // double2 per record
	uint32_t n_nodes = 0;
	uint32_t depth = 0;           // nodes on the longest root-to-leaf path
	double seconds = 0.0;         // wall time of the build, device work included
};

// double2 per triangle record: a line of its own (8) once the records outgrow the caches (4 MB of L2 per XCD), packed (7) below that
inline uint32_t triangle_record_stride(uint64_t n_tris) { return n_tris * 128 > (4u << 20) ? 8u : 7u; }

// d_tri_a_fan / d_tri_b_fan: 6-double records (v0 + face normal, v1 + v2) in fan-triangle order, already on the device;
// d_tri_face / d_face_object: fan triangle -> face, face -> obj object (for the records' last quarter);
// cut: references of long triangles cut into pieces on the host (presplit.h), or null
struct Presplit;
hipError_t build_bvh_device(const double *d_tri_a_fan, const double *d_tri_b_fan, const uint32_t *d_tri_face, const uint32_t *d_face_object,
                            uint32_t n, const Presplit *cut, hipStream_t stream, DeviceBvh &out, std::string &err);

} // namespace mi355rt
