// rt_context: owns one device's copy of the scene, the stream, and the render state.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "bvh_builder.h"
#include "device_scene.h"
#include "host_scene.h"
#include "mi355rt.h"
#include "render.h"
#include "staged_copy.h"
#include "trace.h"

struct rt_context {
	int device = 0;
	hipStream_t stream = nullptr;
	// rt_trace_closest on a large batch: the rays of one part go in on copy_stream and the results of another come back on down_stream
	// (a helper thread, through pinned pieces of its own) while the kernel works on a third (all created on first use)
	hipStream_t copy_stream = nullptr, down_stream = nullptr;
	mi355rt::PinnedPieces pinned_down;
	mutable std::string error;

	mi355rt::HostScene host;
	mi355rt::Bvh2 bvh;     // binary tree, host only: introspection and work counting
	mi355rt::BvhWide wide; // the tree the device walks
	bool has_mesh = false, has_materials = false, has_camera = false;
	int builder = RT_BUILDER_AUTO;
	double cut_budget = 0.25;                          // rt_set_reference_cutting
	uint64_t cut_triangles = 0, extra_references = 0;  // what the last upload cut (presplit.h)
	int triangle_test = RT_TRIANGLE_REFERENCE;
	bool built_on_device = false;
	uint64_t device_nodes = 0;     // nodes of the tree in HBM

	// geometry in HBM
	void *d_nodes = nullptr, *d_tri = nullptr, *d_leaf_tri = nullptr, *d_tri_face = nullptr,
	     *d_face_object = nullptr;
	mi355rt::DeviceGeom geom{};
	mi355rt::TraceWork trace_work; // work counters of the traversal launches (allocated by rt_create)
	uint64_t device_bytes = 0;

	// staging for the host-buffer trace entry points (grown on demand)
	void *d_stage_rays = nullptr, *d_stage_out = nullptr, *d_stage_t64 = nullptr;
	uint64_t stage_capacity = 0;
	mi355rt::PinnedPieces pinned; // the scene's large tables are uploaded through these
	// a large batch of unordered rays in a scene far larger than the caches is traced in the order of the cells its rays start in
	// (ray_bins.h): the ordered copy, the permutation, the results in that order, the cell counters (grown on demand)
	void *d_bin_rays = nullptr, *d_bin_perm = nullptr, *d_bin_hits = nullptr, *d_bin_t64 = nullptr, *d_bin_counters = nullptr;
	uint64_t bin_capacity = 0;
	// the buffers above are shared by every batch of the context: a batch enqueued on one stream must not reorder its rays into
	// them while an earlier batch on another stream still reads its results out of them.  Recorded behind the last kernel that
	// reads them; the next batch's stream waits for it (no host synchronisation: a wait on an event of the same stream is free).
	hipEvent_t bin_done = nullptr;
	bool bin_in_use = false;

	std::vector<rt_material> materials;
	std::vector<uint32_t> light_faces;
	rt_camera camera{};

	mi355rt::RenderState render;
	mi355rt::Profiler profiler;

	bool fail_hip(hipError_t e, const char *what);
	void free_geometry();
};
