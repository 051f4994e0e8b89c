// The reference's `obj` container (obj.h:10-36) and its line parser (obj.cpp:6-152), host side, without Boost:
// an absent index is -1 instead of boost::none.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace mi355host {

struct dvec3 {
	double e[3] = {0, 0, 0};
	dvec3() {}
	dvec3(double a, double b, double c) { e[0] = a; e[1] = b; e[2] = c; }
	double operator[](int i) const { return e[i]; }
	double &operator[](int i) { return e[i]; }
};

struct objobject {
	std::vector<std::vector<std::array<int64_t, 3>>> faces; // per face, per corner: v, vt, vn (0-based, -1 absent)
	std::string name, material_name;
	bool s = false;
};

class obj {
public:
	bool Load(const char *filename); // false on the first line the reference reports as "Error"
	void Clear(void);
	std::vector<std::string> mtl_file;
	std::vector<objobject *> objects;
	std::vector<dvec3> v, vt, vn;
	~obj() { Clear(); }
};

} // namespace mi355host
