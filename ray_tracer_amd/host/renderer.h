// Host-side drop-in for the reference's `Renderer` (renderer.h:21-50): same public members, same error
// behaviour, hot path on the MI355X through the C ABI of mi355rt.h.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "camera.h"
#include "materialnode.h"
#include "mi355rt.h"
#include "obj.h"
#include "ply.h"

namespace mi355host {

enum RenderingAlgorithm { Naive, NEE, MIS, NaiveBDPTType, MISBDPTType }; // renderer.h:12-18

struct Spectrum { // spectrum.h:11-24, the part callers of Renderer touch
	std::vector<double> data = std::vector<double>(RT_NBINS, 0.0);
};

class Renderer {
public:
	Renderer(void);
	~Renderer();
	// addition: render on several GPUs of the node (ordinals may repeat).  The reference's one parallel construct is the OpenMP loop
	// over image columns (renderer.cpp:74-77); here every listed device renders its share of each pass's samples and the films are
	// summed when they are read.  Call before Load; the default is device 0 alone.  False (and last_error) when a device is unusable.
	bool UseDevices(const std::vector<int> &devices);
	void Load(const char *objfilename);
	// addition (the reference parses PLY files but never renders them, object.cpp:153-233): the file becomes one more
	// object called `object_name`, appended to what Load / earlier LoadPly calls brought in; false on a parse error
	bool LoadPly(const char *plyfilename, const char *object_name, PlyNormals normals = PlyNormals::Auto);
	bool LoadMaterials(const std::map<std::string, std::shared_ptr<NodeMaterial>> &materials);
	void Clear(void);
	// the reference keeps these in globals (env_mapping_texture, env_mapping_width, env_mapping_height, env_mapping_bpp,
	// env_brightness: scene.h:20-22) that NaivePathTracing reads (renderer.cpp:175-193); texels == nullptr: no environment
	bool SetEnvironment(const uint8_t *texels, int width, int height, int bpp, double brightness);
	void RenderImage(int nx, int ny, int ns, int spectral_samples, bool enable_openmp, bool print_progress = false);
	enum RenderingAlgorithm algorithm_type = NaiveBDPTType; // renderer.h:33
	std::vector<Spectrum> spectrum_img;
	std::vector<dvec3> preview_img;
	obj obj_loader;
	ply ply_loader; // the last PLY file read by LoadPly
	LensCamera cam;
	bool preview_img_flag = true;
	bool img_updated = false;
	bool rendering_runnnig = false;
	bool stop_rendering = false;
	float progress = 0.0;
	// additions: the reference seeds its generator from random_device; a fixed seed makes runs repeatable
	unsigned long long seed;
	// samples per pixel handed to the device at a time (the reference renders and previews one at a time, renderer.cpp:70-143).  0: as
	// many as fill one batch of the device's queues (2^27 paths: 64 at 1920x1080) -- a pass is a batch, and a small batch pays the
	// launches of its late, nearly empty bounces for few samples; stop_rendering, the preview and the progress are served per pass
	int samples_per_pass = 0;
	std::string last_error;

private:
	bool Upload(void);
	MeshTables tables;              // what the device holds: obj objects first, then one object per LoadPly call
	rt_group *gpu = nullptr;        // one context per device in use
};

} // namespace mi355host
