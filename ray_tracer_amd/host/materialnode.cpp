#include "materialnode.h"
#include <cstring>

namespace mi355host {

namespace {
constexpr int N = RT_NBINS;
constexpr int Output_i = 1; // nodes 0/1 are always UV/Output (materialnode.h:449-450)

std::vector<double> own_array(const Json &node, const char *key)
{
	std::vector<double> out(N, 0.0);
	if (node.contains(key)) {
		const Json &a = node.at(key);
		for (size_t i = 0; i < a.size() && i < (size_t)N; i++) out[i] = a[i].number(0.0);
	}
	return out;
}
} // namespace

NodeMaterial::NodeMaterial(const Json &j)
{
	name = j.contains("name") ? j.at("name").string_or("untitled material") : "untitled material";
	light_flag = j.contains("light_flag") ? j.at("light_flag").bool_or(false) : false;
	if (j.contains("nodes")) nodes = j.at("nodes").arr;
	conns.resize(nodes.size());
	if (j.contains("connections")) {
		for (const Json &c : j.at("connections").arr) {
			Conn k{(int)c.at("input_node").number(-1), (int)c.at("output_node").number(-1), c.at("input_slot").string_or(""),
			       c.at("output_slot").string_or("")};
			if (k.output_node < 0 || k.input_node < 0 || k.output_node >= (int)nodes.size() || k.input_node >= (int)nodes.size())
				throw std::runtime_error("material " + name + ": connection to a node that does not exist");
			conns[k.output_node].push_back(k);
			conns[k.input_node].push_back(k);
		}
	}
}

int NodeMaterial::Parent(int node, const char *slot) const
{
	const size_t n = std::strlen(slot);
	for (const Conn &c : conns[node])
		if (std::strncmp(c.input_slot.c_str(), slot, n) == 0) return c.output_node;
	return -1;
}

std::vector<double> NodeMaterial::ComputeVec3(int node, int depth) const
{
	if (depth > 64) throw UnsupportedMaterialGraph("cyclic material graph in " + name);
	const std::string t = nodes[node].at("type").string_or("");
	if (t == "RGBColor") { // materialnode.cpp:931-936
		const Json &c = nodes[node].at("col");
		return {c[0].number(0), c[1].number(0), c[2].number(0)};
	}
	if (t == "UV" || t == "ImageTexture" || t == "Checkerboard" || t == "RodriguesRotation" || t == "CombineVec3Component" ||
	    t == "Addition" || t == "Multiplication")
		throw UnsupportedMaterialGraph("node " + t + " of material " + name + " needs per-hit evaluation");
	return {0.0, 0.0, 0.0};
}

std::vector<double> NodeMaterial::ComputeSpectrum(int node, int depth) const
{
	if (depth > 64) throw UnsupportedMaterialGraph("cyclic material graph in " + name);
	const std::string t = nodes[node].at("type").string_or("");
	std::vector<double> data(N, 0.0); // a default-constructed Spectrum is all zero (spectrum.cpp:107-109)
	if (t == "Spectrum") return own_array(nodes[node], "data"); // :878-881
	if (t == "RGBtoSpectrum") { // :949-957 with spectrum.cpp:219-232
		int p = Parent(node, "RGB");
		if (p < 0) return data;
		std::vector<double> rgb = ComputeVec3(p, depth + 1);
		for (int i = 0; i < N; i++) data[i] = i < N / 3 ? rgb[2] : (i < 2 * N / 3 ? rgb[1] : rgb[0]);
		return data;
	}
	if (t == "Addition") { // :1083-1109
		int p = Parent(node, "In0");
		if (p < 0) return data;
		data = ComputeSpectrum(p, depth + 1);
		p = Parent(node, "In1");
		if (p < 0) return data;
		std::vector<double> b = ComputeSpectrum(p, depth + 1);
		for (int i = 0; i < N; i++) data[i] = data[i] + b[i];
		return data;
	}
	if (t == "Multiplication") { // :1179-1203
		int p0 = Parent(node, "In0"), p1 = Parent(node, "In1");
		if (p0 < 0) return data;
		std::vector<double> a = ComputeSpectrum(p0, depth + 1);
		if (p1 < 0) return data;
		std::vector<double> b = ComputeSpectrum(p1, depth + 1);
		for (int i = 0; i < N; i++) data[i] = a[i] * b[i];
		return data;
	}
	if (t == "ScalarMultiplication") { // :1145-1166: looks up a slot called "scale", which no node has
		int p = Parent(node, "scale");
		if (p < 0) return data;
		std::vector<double> b = ComputeSpectrum(p, depth + 1);
		for (int i = 0; i < N; i++) data[i] = data[i] * b[i];
		return data;
	}
	return data; // RandomSampling writes nothing (:1251-1264); the other nodes have no Spectrum Compute (:78-80)
}

std::vector<double> NodeMaterial::SlotSpectrum(int node, const char *slot, const char *own_key) const
{
	int p = Parent(node, slot);
	if (p < 0) return own_array(nodes[node], own_key);
	return ComputeSpectrum(p, 0);
}

rt_material NodeMaterial::Record() const
{
	rt_material r;
	std::memset(&r, 0, sizeof(r));
	r.type = RT_MAT_NONE;
	r.light_flag = light_flag ? 1 : 0;
	if ((int)nodes.size() <= Output_i) return r;
	int b = Parent(Output_i, "BSDF"); // materialnode.cpp:1760
	if (b < 0) return r;
	const std::string t = nodes[b].at("type").string_or("");
	std::vector<double> a(N, 0.0), bb(N, 0.0);
	if (t == "Lambertian") { r.type = RT_MAT_LAMBERTIAN; a = SlotSpectrum(b, "albedo", "albedo"); }
	else if (t == "DiffuseLight") { r.type = RT_MAT_DIFFUSE_LIGHT; a = SlotSpectrum(b, "color", "color"); }
	else if (t == "Dielectric") { // "surface_color" never matches the slot "surface color" (:300,359)
		r.type = RT_MAT_DIELECTRIC;
		a = own_array(nodes[b], "n");
		bb = SlotSpectrum(b, "surface_color", "surface_color");
	} else if (t == "Conductor") { r.type = RT_MAT_CONDUCTOR; a = own_array(nodes[b], "n"); bb = own_array(nodes[b], "k"); }
	else if (t == "ColoredMetal") { r.type = RT_MAT_COLORED_METAL; a = SlotSpectrum(b, "albedo", "albedo"); }
	else if (t == "GGXReflection")
		throw UnsupportedMaterialGraph("GGXReflection reads an uninitialised alpha and returns no value from BSDF()/PDF() in the reference");
	else throw UnsupportedMaterialGraph("node type " + t + " wired to the BSDF output of " + name);
	for (int i = 0; i < N; i++) { r.a[i] = a[i]; r.b[i] = bb[i]; }
	return r;
}

std::map<std::string, std::shared_ptr<NodeMaterial>> LoadMaterialFile(const std::string &path)
{
	std::map<std::string, std::shared_ptr<NodeMaterial>> out;
	Json j = Json::parse_file(path);
	for (const Json &m : j.arr) {
		auto nm = std::make_shared<NodeMaterial>(m);
		out[nm->name] = nm;
	}
	return out;
}

std::map<std::string, std::string> LoadObjMatMap(const std::string &path)
{
	std::map<std::string, std::string> out;
	Json j = Json::parse_file(path);
	for (auto &kv : j.obj) out[kv.first] = kv.second.string_or("");
	return out;
}

} // namespace mi355host
