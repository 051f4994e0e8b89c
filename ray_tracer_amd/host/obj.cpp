// The reference's OBJ line parser (obj.cpp:6-152) restated over the file's bytes: same lines accepted and refused, same values, same
// quirks, without a stream object per line and per face corner (875 k lines: 0.97 s with streams, 0.1 s here).
//
// What the reference's stream code does, and this must do too:
//   * a line that is blank or starts with '#' is skipped; before the first object only "mtllib" lines are taken, any other line
//     must start with the letter 'o' (obj.cpp:13-30) and opens an object whatever its first word is
//   * words are separated by white space; a word that is missing leaves the string it would have been read into unchanged:
//     "v 1 2" gives (1, 2, 2), "usemtl" alone names the material "usemtl", "s" alone switches smoothing on
//   * numbers are atof / atoi of the word: the longest numeric prefix, 0 when there is none
//   * a face line is read `while (!eof)`: when white space follows the last corner, the last word is taken once more (obj.cpp:105-106)
//   * a corner is split at '/', an empty part is an absent index
//   * any other first word ends the load with false (obj.cpp:139-142)
#include "obj.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mi355host {

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

struct Word {
	const char *b = nullptr, *e = nullptr;
	bool ok() const { return b != nullptr; }
	size_t size() const { return (size_t)(e - b); }
	bool is(const char *s) const { return ok() && size() == std::strlen(s) && std::memcmp(b, s, size()) == 0; }
	std::string str() const { return std::string(b, e); }
};

// next white-space separated word of [p, end); a Word that is not ok() when there is none
inline Word next_word(const char *&p, const char *end)
{
	while (p < end && is_space(*p)) p++;
	Word w;
	if (p == end) return w;
	w.b = p;
	while (p < end && !is_space(*p)) p++;
	w.e = p;
	return w;
}

const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// atof of a word.  Plain decimals with at most 15 significant digits and a power of ten within 10^22 are exact in double
// arithmetic (one correctly rounded multiplication or division of two exact values), which is what strtod returns for them;
// everything else (long mantissas, large exponents, inf, nan, hex, junk) goes to strtod itself.
double word_to_double(const Word &w)
{
	if (!w.ok()) return 0.0;
	const char *p = w.b;
	bool neg = false;
	if (p < w.e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
	uint64_t mant = 0;
	int digits = 0, exp10 = 0;
	bool any = false;
	while (p < w.e && *p >= '0' && *p <= '9') {
		if (mant != 0 || *p != '0') { mant = mant * 10 + (uint64_t)(*p - '0'); digits++; }
		any = true;
		p++;
		if (digits > 15) goto slow;
	}
	if (p < w.e && *p == '.') {
		p++;
		while (p < w.e && *p >= '0' && *p <= '9') {
			if (mant != 0 || *p != '0') { mant = mant * 10 + (uint64_t)(*p - '0'); digits++; }
			exp10--;
			any = true;
			p++;
			if (digits > 15) goto slow;
		}
	}
	if (!any) goto slow;
	if (p < w.e && (*p == 'e' || *p == 'E')) {
		const char *q = p + 1;
		bool eneg = false;
		if (q < w.e && (*q == '-' || *q == '+')) { eneg = *q == '-'; q++; }
		if (q < w.e && *q >= '0' && *q <= '9') {
			int ex = 0;
			while (q < w.e && *q >= '0' && *q <= '9') { ex = ex * 10 + (*q - '0'); q++; if (ex > 400) goto slow; }
			exp10 += eneg ? -ex : ex;
			p = q;
		}
	}
	if (p != w.e) goto slow; // something follows the number: let strtod decide what the prefix is
	if (exp10 < -22 || exp10 > 22) goto slow;
	{
		double v = (double)mant;
		v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
		return neg ? -v : v;
	}
slow:
	return std::strtod(w.str().c_str(), nullptr);
}

// atoi of [b, e): optional sign, then digits; 0 when there are none
int64_t part_to_int(const char *b, const char *e)
{
	while (b < e && is_space(*b)) b++;
	bool neg = false;
	if (b < e && (*b == '-' || *b == '+')) { neg = *b == '-'; b++; }
	// glibc's atoi is (int)strtol: the long saturates, the conversion to int keeps the low half
	const uint64_t limit = neg ? (uint64_t)1 << 63 : ((uint64_t)1 << 63) - 1;
	uint64_t v = 0;
	while (b < e && *b >= '0' && *b <= '9') {
		const uint64_t d = (uint64_t)(*b - '0');
		v = v > (limit - d) / 10 ? limit : v * 10 + d;
		b++;
	}
	const int64_t as_long = neg ? (int64_t)(0 - v) : (int64_t)v;
	return (int64_t)(int32_t)(uint32_t)(uint64_t)as_long;
}

std::array<int64_t, 3> corner(const Word &w)
{
	std::array<int64_t, 3> info = {-1, -1, -1};
	const char *p = w.b;
	for (int k = 0; k < 3 && p <= w.e; k++) {
		const char *q = p;
		while (q < w.e && *q != '/') q++;
		bool blank = true;
		for (const char *c = p; c < q; c++) blank = blank && is_space(*c);
		if (!blank) info[k] = (int64_t)(int32_t)((uint32_t)part_to_int(p, q) - 1u); // `atoi(...) - 1` is int arithmetic in the reference
		if (q == w.e) break; // the stream is at its end: further parts stay absent
		p = q + 1;
	}
	return info;
}

} // namespace

bool obj::Load(const char *filename)
{
	std::FILE *f = std::fopen(filename, "rb");
	if (!f) return false;
	std::string data;
	if (std::fseek(f, 0, SEEK_END) == 0) { // sized in one go when the file can tell its length
		const long size = std::ftell(f);
		if (size > 0) data.reserve((size_t)size);
		std::rewind(f);
	}
	{
		char buf[1 << 16];
		size_t got;
		while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, got);
	}
	std::fclose(f);
	const char *p = data.data(), *const end = p + data.size();
	objobject *cur = nullptr;
	while (p < end) {
		const char *le = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
		if (!le) le = end;
		const char *line = p;
		p = le < end ? le + 1 : end;
		const char *c = line;
		const Word first = next_word(c, le);
		if (!first.ok() || *line == '#') continue; // blank, or a comment
		if (cur == nullptr) { // obj.cpp:13-30
			if ((size_t)(le - line) >= 6 && std::memcmp(line, "mtllib", 6) == 0) {
				const Word w = next_word(c, le);
				mtl_file.push_back(w.ok() ? w.str() : first.str());
				continue;
			}
			if (*line != 'o') return false;
		}
		if (first.is("o") || cur == nullptr) { // obj.cpp:53-62
			cur = new objobject();
			objects.push_back(cur);
			const Word w = next_word(c, le);
			cur->name = w.ok() ? w.str() : first.str();
		} else if (first.is("v") || first.is("vn")) { // obj.cpp:77-85, 94-102
			dvec3 q;
			Word a; // a missing number repeats the one before it; none at all is atof("") = 0
			for (int k = 0; k < 3; k++) {
				const Word w = next_word(c, le);
				if (w.ok()) a = w;
				q.e[k] = word_to_double(a);
			}
			(first.is("v") ? v : vn).push_back(q);
		} else if (first.is("vt")) { // obj.cpp:86-93
			dvec3 q;
			Word a;
			for (int k = 0; k < 2; k++) {
				const Word w = next_word(c, le);
				if (w.ok()) a = w;
				q.e[k] = word_to_double(a);
			}
			q.e[2] = 0;
			vt.push_back(q);
		} else if (first.is("f")) { // obj.cpp:103-128
			cur->faces.emplace_back();
			std::vector<std::array<int64_t, 3>> &face = cur->faces.back();
			face.reserve(4);
			Word last = first;
			// the stream's eof flag is up once a word has run into the end of the line; while it is down the loop reads on, and a read
			// that finds only white space leaves the previous word in place
			bool at_eof = first.e == le;
			while (!at_eof) {
				const Word w = next_word(c, le);
				if (w.ok()) { last = w; at_eof = w.e == le; }
				else at_eof = true;
				face.push_back(corner(last));
			}
		} else if (first.is("usemtl")) {
			const Word w = next_word(c, le);
			cur->material_name = w.ok() ? w.str() : first.str();
		} else if (first.is("s")) {
			const Word w = next_word(c, le);
			cur->s = !(w.ok() && w.is("off"));
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
