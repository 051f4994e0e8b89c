#include "obj.h"
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace mi355host {

namespace {
bool blank(const std::string &s) { return std::all_of(s.begin(), s.end(), [](unsigned char c) { return std::isspace(c); }); }
}

bool obj::Load(const char *filename)
{
	std::ifstream file(filename);
	if (!file) return false;
	std::string line;
	objobject *cur = nullptr;
	while (std::getline(file, line)) {
		if (blank(line) || line.find("#") == 0) continue;
		std::istringstream iss(line);
		std::string s;
		iss >> s;
		if (cur == nullptr) { // obj.cpp:13-30
			if (line.find("mtllib") == 0) { iss >> s; mtl_file.push_back(s); continue; }
			if (line.find("o") != 0) return false;
		}
		if (s == "o" || cur == nullptr) { // obj.cpp:53-62
			cur = new objobject();
			objects.push_back(cur);
			iss >> s;
			cur->name = s;
		} else if (s == "v" || s == "vn") { // obj.cpp:77-85, 94-102
			dvec3 p;
			std::string a;
			for (int k = 0; k < 3; k++) { iss >> a; p.e[k] = std::atof(a.c_str()); }
			(s == "v" ? v : vn).push_back(p);
		} else if (s == "vt") { // obj.cpp:86-93
			dvec3 p;
			std::string a;
			iss >> a; p.e[0] = std::atof(a.c_str());
			iss >> a; p.e[1] = std::atof(a.c_str());
			p.e[2] = 0;
			vt.push_back(p);
		} else if (s == "f") { // obj.cpp:103-128 (a trailing blank makes the loop re-read the last token)
			std::vector<std::array<int64_t, 3>> face;
			while (!iss.eof()) {
				iss >> s;
				std::istringstream vs(s);
				std::string part;
				std::array<int64_t, 3> info = {-1, -1, -1};
				for (int k = 0; k < 3; k++) {
					part.clear();
					std::getline(vs, part, '/');
					if (!blank(part)) info[k] = std::atoi(part.c_str()) - 1;
				}
				face.push_back(info);
			}
			cur->faces.push_back(face);
		} else if (s == "usemtl") {
			iss >> s;
			cur->material_name = s;
		} else if (s == "s") {
			iss >> s;
			cur->s = !(s == "off");
		} else {
			return false; // obj.cpp:139-142
		}
	}
	return true;
}

void obj::Clear(void)
{
	mtl_file.clear();
	for (auto *o : objects) delete o;
	objects.clear();
	v.clear();
	vt.clear();
	vn.clear();
}

} // namespace mi355host
