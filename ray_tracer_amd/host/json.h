// Minimal JSON reader for the reference's project / materials / object-material-map files
// (the reference vendors nlohmann's json.hpp for this; only reading is needed on this path).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mi355host {

class Json {
public:
	enum Kind { Null, Bool, Number, String, Array, Object };
	Kind kind = Null;
	bool b = false;
	double num = 0.0;
	std::string str;
	std::vector<Json> arr;
	std::vector<std::pair<std::string, Json>> obj; // file order preserved

	static Json parse(const std::string &text);
	static Json parse_file(const std::string &path);

	bool contains(const std::string &key) const;
	const Json &at(const std::string &key) const; // throws std::runtime_error when absent
	const Json &operator[](size_t i) const { return arr.at(i); }
	size_t size() const { return kind == Array ? arr.size() : obj.size(); }
	double number(double fallback) const { return kind == Number ? num : fallback; }
	std::string string_or(const std::string &fallback) const { return kind == String ? str : fallback; }
	bool bool_or(bool fallback) const { return kind == Bool ? b : fallback; }
};

} // namespace mi355host
