#include "json.h"
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace mi355host {
namespace {
struct Parser {
	const std::string &s;
	size_t i = 0;
	explicit Parser(const std::string &t) : s(t) {}
	[[noreturn]] void fail(const char *what) { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(i)); }
	void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
	Json value()
	{
		ws();
		if (i >= s.size()) fail("unexpected end");
		char c = s[i];
		Json j;
		if (c == '{') {
			j.kind = Json::Object;
			i++;
			ws();
			if (i < s.size() && s[i] == '}') { i++; return j; }
			for (;;) {
				ws();
				if (i >= s.size() || s[i] != '"') fail("expected key");
				std::string k = string();
				ws();
				if (i >= s.size() || s[i] != ':') fail("expected ':'");
				i++;
				j.obj.emplace_back(k, value());
				ws();
				if (i < s.size() && s[i] == ',') { i++; continue; }
				if (i < s.size() && s[i] == '}') { i++; return j; }
				fail("expected ',' or '}'");
			}
		}
		if (c == '[') {
			j.kind = Json::Array;
			i++;
			ws();
			if (i < s.size() && s[i] == ']') { i++; return j; }
			for (;;) {
				j.arr.push_back(value());
				ws();
				if (i < s.size() && s[i] == ',') { i++; continue; }
				if (i < s.size() && s[i] == ']') { i++; return j; }
				fail("expected ',' or ']'");
			}
		}
		if (c == '"') { j.kind = Json::String; j.str = string(); return j; }
		if (s.compare(i, 4, "true") == 0) { i += 4; j.kind = Json::Bool; j.b = true; return j; }
		if (s.compare(i, 5, "false") == 0) { i += 5; j.kind = Json::Bool; j.b = false; return j; }
		if (s.compare(i, 4, "null") == 0) { i += 4; return j; }
		char *end = nullptr;
		double v = std::strtod(s.c_str() + i, &end);
		if (end == s.c_str() + i) fail("unexpected character");
		i = (size_t)(end - s.c_str());
		j.kind = Json::Number;
		j.num = v;
		return j;
	}
	std::string string()
	{
		std::string out;
		i++; // opening quote
		while (i < s.size() && s[i] != '"') {
			char c = s[i++];
			if (c == '\\' && i < s.size()) {
				char e = s[i++];
				switch (e) {
				case 'n': out += '\n'; break;
				case 't': out += '\t'; break;
				case 'r': out += '\r'; break;
				case 'b': out += '\b'; break;
				case 'f': out += '\f'; break;
				case 'u': { // basic multilingual plane only, encoded as UTF-8
					unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
					i += 4;
					if (cp < 0x80) out += (char)cp;
					else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
					else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
					break;
				}
				default: out += e;
				}
			} else out += c;
		}
		if (i >= s.size()) fail("unterminated string");
		i++;
		return out;
	}
};
} // namespace

Json Json::parse(const std::string &text)
{
	Parser p(text);
	Json j = p.value();
	p.ws();
	if (p.i != text.size()) p.fail("trailing characters");
	return j;
}

Json Json::parse_file(const std::string &path)
{
	std::ifstream f(path);
	if (!f) throw std::runtime_error("json: cannot open " + path);
	std::stringstream ss;
	ss << f.rdbuf();
	return parse(ss.str());
}

bool Json::contains(const std::string &key) const
{
	for (auto &kv : obj) if (kv.first == key) return true;
	return false;
}

const Json &Json::at(const std::string &key) const
{
	for (auto &kv : obj) if (kv.first == key) return kv.second;
	throw std::runtime_error("json: missing key " + key);
}

} // namespace mi355host
