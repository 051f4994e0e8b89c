// NodeMaterial (materialnode.cpp:1611-1638): node list + connections read from the materials JSON, folded to
// the one BSDF record per object that the device evaluates (rt_material).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "json.h"
#include "mi355rt.h"

namespace mi355host {

struct UnsupportedMaterialGraph : std::runtime_error {
	using std::runtime_error::runtime_error;
};

// tables of rt_upload_material_graphs, filled by NodeMaterial::EmitGraph for every object whose albedo / color slot
// reads the hit's texture coordinate
struct MaterialGraphTables {
	std::vector<rt_graph_program> programs;
	std::vector<uint32_t> code; // (op, arg) pairs
	std::vector<double> spectra, vectors, scalars;
	std::vector<std::vector<uint8_t>> texels; // one buffer per texture, with its descriptor below
	std::vector<rt_texture> textures;
	std::vector<std::string> texture_paths;
	rt_material_graphs View(void); // pointers into this object: keep it alive during the upload
};

class NodeMaterial {
public:
	explicit NodeMaterial(const Json &j);
	std::string name;
	bool light_flag = false;
	// throws UnsupportedMaterialGraph for nodes that are not evaluated on the device (nodes the device has no evaluator for).
	// When the albedo / color slot depends on the texture coordinate, `a` is zero and EmitGraph supplies the program.
	rt_material Record() const;
	// appends this material's per-hit program for `object` to the tables; false when its slot is a constant.
	// ImageTexture files are read as binary PPM (the reference decodes with stb_image); a missing file leaves zeros
	// as the reference's null texture does (materialnode.cpp:1003-1004)
	bool EmitGraph(uint32_t object, MaterialGraphTables &out) const;

private:
	struct Conn { int input_node, output_node; std::string input_slot, output_slot; };
	std::vector<Json> nodes;
	std::vector<std::vector<Conn>> conns; // per node, both directions, file order (materialnode.cpp:1635-1636)
	int Parent(int node, const char *slot) const; // MaterialNode::GetInputParentNode, materialnode.cpp:66-74
	std::vector<double> ComputeSpectrum(int node, int depth) const;
	std::vector<double> ComputeVec3(int node, int depth) const;
	std::vector<double> SlotSpectrum(int node, const char *slot, const char *own_key) const;
	struct Expr;
	std::shared_ptr<Expr> VecExpr(int node, int depth) const;
	std::shared_ptr<Expr> ScalarExpr(int node, int depth) const;
	std::shared_ptr<Expr> SpecExpr(int node, int depth) const;
	std::shared_ptr<Expr> SlotExpr(void) const; // the expression of the BSDF node's albedo / color slot, null for other nodes
};

// Scene::LoadMaterial (scene.cpp:249-265) and Scene::LoadObjMatMap (scene.cpp:282-300)
std::map<std::string, std::shared_ptr<NodeMaterial>> LoadMaterialFile(const std::string &path);
std::map<std::string, std::string> LoadObjMatMap(const std::string &path);

} // namespace mi355host
