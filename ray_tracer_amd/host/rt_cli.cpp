// Headless driver for the host-side Renderer: what the reference's `--cli` mode does (main.cpp:54-99) plus the
// camera set-up the GUI performs (scene.cpp:866-867), with the linear film dumped for tests.
//   rt_cli --obj scene.obj --materials materials.json --objmatmap map.json [--width W --height H --sample S]
//          [--algorithm naive|nee|mis|bdpt|misbdpt] [--seed N] [--spectral 30] [--film out.f64] [--ppm out.ppm] [--records out.f64] [--graphs out.bin]
//          [--ply mesh.ply --ply-object NAME --ply-normals auto|file|flat|smooth]   one more object from a PLY file
//          [--devices 0,1,2,3]     render on several GPUs (ordinals may repeat), samples of each pass split between them
//   rt_cli --ply mesh.ply --dump-ply out.bin       host only: the parsed tables (u64 nv, nf, ni, has_normals; f64 [nv][6];
//                                                  u64 [nf+1]; i32 [ni]) for tests
//   rt_cli --obj scene.obj --dump-obj out.txt      host only: what the OBJ parser made of the file, as text ("ok 0|1", then mtl / v /
//                                                  vt / vn lines with hexadecimal floats, then per object "o [name] [material] s" and
//                                                  its faces "f v/vt/vn ..." 0-based, -1 absent) for tests
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include "renderer.h"

using namespace mi355host;

int main(int argc, char **argv)
{
	std::string objp, matp, mapp, film_out, ppm_out, rec_out, graphs_out, alg = "mis", plyp, ply_object = "ply", ply_normals = "auto", ply_dump, obj_dump, devices;
	int w = 64, h = 64, ns = 4, spectral = 30;
	unsigned long long seed = 1;
	for (int i = 1; i + 1 < argc; i += 2) {
		std::string k = argv[i], v = argv[i + 1];
		if (k == "--obj") objp = v; else if (k == "--materials") matp = v; else if (k == "--objmatmap") mapp = v;
		else if (k == "--width") w = std::atoi(v.c_str()); else if (k == "--height") h = std::atoi(v.c_str());
		else if (k == "--sample") ns = std::atoi(v.c_str()); else if (k == "--algorithm") alg = v;
		else if (k == "--seed") seed = std::strtoull(v.c_str(), nullptr, 10); else if (k == "--spectral") spectral = std::atoi(v.c_str());
		else if (k == "--film") film_out = v; else if (k == "--ppm") ppm_out = v; else if (k == "--records") rec_out = v; else if (k == "--graphs") graphs_out = v;
		else if (k == "--ply") plyp = v; else if (k == "--ply-object") ply_object = v; else if (k == "--ply-normals") ply_normals = v;
		else if (k == "--dump-ply") ply_dump = v; else if (k == "--dump-obj") obj_dump = v; else if (k == "--devices") devices = v;
		else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
	}
	if (!ply_dump.empty()) {
		ply p;
		if (!p.Load(plyp.c_str())) { std::fprintf(stderr, "%s\n", p.error.c_str()); return 7; }
		std::ofstream f(ply_dump, std::ios::binary);
		uint64_t head[4] = {p.n_vertices(), p.n_faces(), p.face_index.size(), p.has_normals() ? 1u : 0u};
		f.write((const char *)head, sizeof(head));
		f.write((const char *)p.vertex_data.data(), sizeof(double) * p.vertex_data.size());
		f.write((const char *)p.face_begin.data(), sizeof(uint64_t) * p.face_begin.size());
		f.write((const char *)p.face_index.data(), sizeof(int32_t) * p.face_index.size());
		for (auto &e : p.elements) std::printf("element %s %zu %zu\n", e.name.c_str(), e.num_of_elements, e.num_of_property);
		return 0;
	}
	if (!obj_dump.empty()) {
		obj o;
		const bool ok = o.Load(objp.c_str());
		std::FILE *f = std::fopen(obj_dump.c_str(), "w");
		if (!f) { std::fprintf(stderr, "cannot write %s\n", obj_dump.c_str()); return 7; }
		std::fprintf(f, "ok %d\n", ok ? 1 : 0);
		for (auto &m : o.mtl_file) std::fprintf(f, "mtl [%s]\n", m.c_str());
		for (auto &p : o.v) std::fprintf(f, "v %a %a %a\n", p.e[0], p.e[1], p.e[2]);
		for (auto &p : o.vt) std::fprintf(f, "vt %a %a %a\n", p.e[0], p.e[1], p.e[2]);
		for (auto &p : o.vn) std::fprintf(f, "vn %a %a %a\n", p.e[0], p.e[1], p.e[2]);
		for (auto *ob : o.objects) {
			std::fprintf(f, "o [%s] [%s] %d\n", ob->name.c_str(), ob->material_name.c_str(), ob->s ? 1 : 0);
			for (auto &face : ob->faces) {
				std::fprintf(f, "f");
				for (auto &c : face) std::fprintf(f, " %lld/%lld/%lld", (long long)c[0], (long long)c[1], (long long)c[2]);
				std::fprintf(f, "\n");
			}
		}
		std::fclose(f);
		return 0;
	}
	if (objp.empty() || matp.empty() || mapp.empty()) { std::fprintf(stderr, "need --obj --materials --objmatmap\n"); return 2; }
	try {
		auto named = LoadMaterialFile(matp);
		auto omm = LoadObjMatMap(mapp);
		std::map<std::string, std::shared_ptr<NodeMaterial>> by_object; // Scene::obj_materials, scene.cpp:302-315
		for (auto &kv : omm) {
			auto it = named.find(kv.second);
			if (it != named.end()) by_object[kv.first] = it->second;
		}
		if (!rec_out.empty()) { // host logic only (no device needed): folded records in obj object order
			obj parsed;
			parsed.Load(objp.c_str());
			std::ofstream f(rec_out, std::ios::binary);
			for (auto *o : parsed.objects) {
				rt_material m = by_object.at(o->name)->Record();
				double head[2] = {(double)m.type, (double)m.light_flag};
				f.write((const char *)head, sizeof(head));
				f.write((const char *)m.a, sizeof(m.a));
				f.write((const char *)m.b, sizeof(m.b));
				f.write((const char *)&m.roughness, sizeof(m.roughness));
			}
		}
		if (!graphs_out.empty()) { // host logic only: the per-hit programs, as u64 counts (programs, code pairs, spectra, vectors, scalars),
			// then u32 [programs][3], u32 [pairs][2], f64 spectra, vectors, scalars
			obj parsed;
			parsed.Load(objp.c_str());
			MaterialGraphTables t;
			uint32_t k = 0;
			for (auto *o : parsed.objects) by_object.at(o->name)->EmitGraph(k++, t);
			std::ofstream f(graphs_out, std::ios::binary);
			uint64_t head[5] = {t.programs.size(), t.code.size() / 2, t.spectra.size() / 30, t.vectors.size() / 3, t.scalars.size()};
			f.write((const char *)head, sizeof(head));
			f.write((const char *)t.programs.data(), sizeof(rt_graph_program) * t.programs.size());
			f.write((const char *)t.code.data(), sizeof(uint32_t) * t.code.size());
			f.write((const char *)t.spectra.data(), sizeof(double) * t.spectra.size());
			f.write((const char *)t.vectors.data(), sizeof(double) * t.vectors.size());
			f.write((const char *)t.scalars.data(), sizeof(double) * t.scalars.size());
		}
		Renderer r;
		if (!r.last_error.empty()) return 3;
		if (!devices.empty()) {
			std::vector<int> list;
			for (size_t at = 0; at < devices.size();) {
				size_t comma = devices.find(',', at);
				if (comma == std::string::npos) comma = devices.size();
				list.push_back(std::atoi(devices.substr(at, comma - at).c_str()));
				at = comma + 1;
			}
			if (!r.UseDevices(list)) return 3;
		}
		r.seed = seed;
		r.Load(objp.c_str());
		if (!plyp.empty()) {
			PlyNormals mode = ply_normals == "file" ? PlyNormals::File : ply_normals == "flat" ? PlyNormals::Flat
			                : ply_normals == "smooth" ? PlyNormals::Smooth : PlyNormals::Auto;
			if (!r.LoadPly(plyp.c_str(), ply_object.c_str(), mode)) return 7;
		}
		if (!r.LoadMaterials(by_object)) { std::fprintf(stderr, "LoadMaterials failed: %s\n", r.last_error.c_str()); return 4; }
		r.cam.film_height = 0.024;
		r.cam.film_width = 0.024 * w / h; // scene.cpp:668
		r.cam.SetLens(60.0, 0.4);
		r.cam.set_Camera(dvec3(0, 0, 3), dvec3(0, 0, 2), dvec3(0, 1, 0));
		r.algorithm_type = alg == "naive" ? Naive : alg == "nee" ? NEE : alg == "bdpt" ? NaiveBDPTType : alg == "misbdpt" ? MISBDPTType : MIS;
		r.preview_img_flag = false;
		r.samples_per_pass = ns;
		r.RenderImage(w, h, ns, spectral, true, false);
		if (!r.last_error.empty()) return 5;
		if (!film_out.empty()) {
			std::ofstream f(film_out, std::ios::binary);
			for (auto &s : r.spectrum_img) f.write((const char *)s.data.data(), sizeof(double) * RT_NBINS);
		}
		if (!ppm_out.empty()) { // main.cpp:84-98 / scene.cpp:835-851: 8-bit from the gamma-encoded preview
			std::ofstream f(ppm_out);
			f << "P3\n" << w << " " << h << "\n255\n";
			for (int j = h - 1; j >= 0; j--)
				for (int i = 0; i < w; i++) {
					const dvec3 &c = r.preview_img[(size_t)i * h + j];
					f << (int)(255.99 * c[0]) << " " << (int)(255.99 * c[1]) << " " << (int)(255.99 * c[2]) << "\n";
				}
		}
	} catch (const std::exception &e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 6;
	}
	return 0;
}
