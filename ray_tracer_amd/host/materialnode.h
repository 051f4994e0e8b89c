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

class NodeMaterial {
public:
	explicit NodeMaterial(const Json &j);
	std::string name;
	bool light_flag = false;
	// throws UnsupportedMaterialGraph for graphs that need the hit's texture coordinate (or GGXReflection)
	rt_material Record() const;

private:
	struct Conn { int input_node, output_node; std::string input_slot, output_slot; };
	std::vector<Json> nodes;
	std::vector<std::vector<Conn>> conns; // per node, both directions, file order (materialnode.cpp:1635-1636)
	int Parent(int node, const char *slot) const; // MaterialNode::GetInputParentNode, materialnode.cpp:66-74
	std::vector<double> ComputeSpectrum(int node, int depth) const;
	std::vector<double> ComputeVec3(int node, int depth) const;
	std::vector<double> SlotSpectrum(int node, const char *slot, const char *own_key) const;
};

// Scene::LoadMaterial (scene.cpp:249-265) and Scene::LoadObjMatMap (scene.cpp:282-300)
std::map<std::string, std::shared_ptr<NodeMaterial>> LoadMaterialFile(const std::string &path);
std::map<std::string, std::string> LoadObjMatMap(const std::string &path);

} // namespace mi355host
