#include "renderer.h"
#include <algorithm>
#include <cstdio>
#include <iostream>
#include <random>

namespace mi355host {

Renderer::Renderer(void) // renderer.cpp:12-16
{
	std::random_device rd;
	seed = ((unsigned long long)rd() << 32) | rd();
	UseDevices({0});
}

Renderer::~Renderer() { rt_group_destroy(gpu); }

bool Renderer::UseDevices(const std::vector<int> &devices)
{
	rt_group_destroy(gpu);
	gpu = nullptr;
	last_error.clear();
	if (rt_group_create(devices.data(), (int)devices.size(), &gpu) != RT_OK) {
		last_error = rt_group_last_error(nullptr);
		std::fprintf(stderr, "%s\n", last_error.c_str());
		return false;
	}
	return true;
}

void Renderer::Load(const char *objfilename) // renderer.cpp:18-22 (the parser's bool is ignored there too)
{
	obj_loader.Clear();
	obj_loader.Load(objfilename);
	tables = MeshTables();
	for (auto &p : obj_loader.v) { tables.v.push_back(p[0]); tables.v.push_back(p[1]); tables.v.push_back(p[2]); }
	for (auto &p : obj_loader.vt) { tables.vt.push_back(p[0]); tables.vt.push_back(p[1]); tables.vt.push_back(p[2]); }
	for (auto &p : obj_loader.vn) { tables.vn.push_back(p[0]); tables.vn.push_back(p[1]); tables.vn.push_back(p[2]); }
	for (auto *o : obj_loader.objects) {
		for (auto &face : o->faces) {
			for (auto &c : face) { tables.corners.push_back(c[0]); tables.corners.push_back(c[1]); tables.corners.push_back(c[2]); }
			tables.face_begin.push_back(tables.corners.size() / 3);
		}
		tables.object_begin.push_back(tables.face_begin.size() - 1);
		tables.object_names.push_back(o->name);
	}
	Upload();
}

bool Renderer::LoadPly(const char *plyfilename, const char *object_name, PlyNormals normals)
{
	if (!ply_loader.Load(plyfilename)) {
		last_error = ply_loader.error;
		std::fprintf(stderr, "%s\n", last_error.c_str());
		return false;
	}
	size_t dropped = AppendPly(ply_loader, object_name, normals, tables);
	if (dropped) std::fprintf(stderr, "%zu faces of %s dropped (fewer than 3 corners or index out of range)\n", dropped, plyfilename);
	return Upload();
}

bool Renderer::Upload(void)
{
	rt_mesh m{tables.v.data(), tables.v.size() / 3, tables.vt.data(), tables.vt.size() / 3, tables.vn.data(), tables.vn.size() / 3,
	          tables.corners.data(), tables.face_begin.data(), tables.face_begin.size() - 1, tables.object_begin.data(),
	          (uint32_t)(tables.object_begin.size() - 1)};
	if (!gpu || rt_group_upload_mesh(gpu, &m) != RT_OK) {
		last_error = rt_group_last_error(gpu);
		std::fprintf(stderr, "%s\n", last_error.c_str());
		return false;
	}
	return true;
}

bool Renderer::LoadMaterials(const std::map<std::string, std::shared_ptr<NodeMaterial>> &materials) // renderer.cpp:24-44
{
	std::vector<rt_material> recs;
	MaterialGraphTables graphs;
	for (size_t i = 0; i < tables.object_names.size(); i++) { // obj objects in file order (as the reference), then PLY objects
		auto itr = materials.find(tables.object_names[i]);
		if (itr == materials.end()) {
			std::fprintf(stderr, "Material is unset for %s\n", tables.object_names[i].c_str());
			return false;
		}
		recs.push_back(itr->second->Record());
		itr->second->EmitGraph((uint32_t)i, graphs); // per-hit program when the albedo / color slot reads the texture coordinate
	}
	if (!gpu) return false;
	int rc = rt_group_upload_materials(gpu, recs.data(), (uint32_t)recs.size());
	if (rc == RT_OK && !graphs.programs.empty()) {
		rt_material_graphs view = graphs.View();
		rc = rt_group_upload_material_graphs(gpu, &view);
	}
	if (rc != RT_OK) last_error = rt_group_last_error(gpu);
	return rc == RT_OK; // RT_ERR_NO_LIGHTS: `light_objects.size() == 0` in the reference
}

bool Renderer::SetEnvironment(const uint8_t *texels, int width, int height, int bpp, double brightness)
{
	if (!gpu || rt_group_set_environment(gpu, texels, width, height, bpp, brightness) != RT_OK) {
		last_error = gpu ? rt_group_last_error(gpu) : "no device";
		return false;
	}
	return true;
}

void Renderer::Clear(void) // renderer.cpp:47-51
{
	obj_loader.Clear();
	ply_loader.Clear();
	tables = MeshTables();
	img_updated = false;
}

void Renderer::RenderImage(int nx, int ny, int ns, int spectral_samples, bool, bool print_progress) // renderer.cpp:54-160
{
	spectrum_img.assign((size_t)nx * ny, Spectrum());
	preview_img.assign((size_t)nx * ny, dvec3());
	rendering_runnnig = true;
	rt_camera c = cam.Pod();
	if (!gpu || rt_group_set_camera(gpu, &c) != RT_OK ||
	    rt_group_render_begin(gpu, nx, ny, ns, spectral_samples, (int)algorithm_type, seed) != RT_OK) {
		last_error = gpu ? rt_group_last_error(gpu) : "no device";
		std::fprintf(stderr, "%s\n", last_error.c_str());
		rendering_runnnig = false;
		return;
	}
	std::vector<double> prev((size_t)nx * ny * 3);
	int s = 0;
	while (s < ns) {
		if (stop_rendering) break; // once per pass, renderer.cpp:70
		// samples_per_pass = 0: the library's choice.  Unwatched, a pass is one full device batch (2^28 paths where the GPU's memory allows, the
		// library halves it where not: tens of milliseconds,
		// so stop_rendering is still looked at many times a second).  With a preview or progress output wanted the reference serves
		// them once per sample (renderer.cpp:69-151); here a watched render reports at least a hundred times.
		const long long fill = std::max(1LL, (1LL << 28) / std::max(1LL, (long long)nx * ny));
		const long long want = samples_per_pass > 0 ? samples_per_pass
		                       : (preview_img_flag || print_progress) ? std::max(1LL, std::min<long long>(fill, ns / 100)) : fill;
		int b = (int)std::min<long long>(want, ns - s);
		if (rt_group_render_pass(gpu, s, b) != RT_OK) { // no sync here: the pass's last paths ride along with the next pass; the downloads finish them
			last_error = rt_group_last_error(gpu);
			std::fprintf(stderr, "%s\n", last_error.c_str());
			break;
		}
		s += b;
		if (preview_img_flag || s == ns) { // renderer.cpp:124-143
			rt_group_download_preview(gpu, s, prev.data());
			for (size_t p = 0; p < (size_t)nx * ny; p++) preview_img[p] = dvec3(prev[3 * p], prev[3 * p + 1], prev[3 * p + 2]);
			img_updated = true;
		}
		progress = static_cast<float>(s) / ns;
		if (print_progress) std::cout << progress * 100.0 << "%" << std::endl;
	}
	std::vector<double> film((size_t)nx * ny * RT_NBINS);
	rt_group_download_film(gpu, film.data());
	for (size_t p = 0; p < (size_t)nx * ny; p++)
		for (int k = 0; k < RT_NBINS; k++) spectrum_img[p].data[k] = film[p * RT_NBINS + k];
	img_updated = true;
	if (stop_rendering) stop_rendering = false;
	rendering_runnnig = false;
	if (print_progress) std::cout << "Completed" << std::endl;
}

} // namespace mi355host
