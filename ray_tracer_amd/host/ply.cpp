#include "ply.h"
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mi355host {

namespace {

// `iss >> s` splits at the characters std::isspace accepts in the C locale
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

struct Line {
	const char *b, *e;
};

// The std::ifstream calls of ply.cpp (getline, tellg, seekg) with libstdc++'s state rules, which decide what the
// reference reads from truncated files: an input function on a stream that is not good() sets failbit and does
// nothing else, so a getline that fails this way leaves the caller's line as it was, while one that finds the stream
// good but at its end erases the line.
struct Cursor {
	const char *begin, *p, *end;
	bool eof = false, fail = false;
	bool getline(Line &l)
	{
		if (eof || fail) { fail = true; return false; }
		if (p >= end) { eof = fail = true; l.b = l.e = end; return false; }
		const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
		l.b = p;
		l.e = nl ? nl : end;
		p = nl ? nl + 1 : end;
		if (!nl) eof = true;
		return true;
	}
	const char *tellg(void)
	{
		if (eof || fail) { fail = true; return nullptr; }
		return p;
	}
	void seekg(const char *pos)
	{
		eof = false;
		if (!fail) p = pos;
	}
};

// next whitespace-delimited token of the line; false when there is none (the caller then keeps its previous token,
// which is what a failed `iss >> s` leaves behind)
inline bool next_token(const char *&q, const char *e, const char *&tb, const char *&te)
{
	while (q < e && is_space(*q)) q++;
	if (q >= e) return false;
	tb = q;
	while (q < e && !is_space(*q)) q++;
	te = q;
	return true;
}

// atof on a token: the longest valid prefix, 0 when there is none.  from_chars is correctly rounded like strtod and
// several times faster; anything it does not take in full (leading '+', hex floats, "infinity", ...) goes to strtod
double token_atof(const char *b, const char *e)
{
	if (b == e) return 0.0;
	double v;
	auto r = std::from_chars(b, e, v);
	if (r.ec == std::errc() && r.ptr == e) return v;
	std::string s(b, e);
	return std::atof(s.c_str());
}

// atoi on a token: optional sign, then digits; stops at the first other character
int token_atoi(const char *b, const char *e)
{
	if (b == e) return 0;
	bool neg = false;
	if (*b == '+' || *b == '-') { neg = *b == '-'; b++; }
	long long v = 0;
	while (b < e && *b >= '0' && *b <= '9') { v = v * 10 + (*b - '0'); if (v > 0x7fffffffLL + 1) break; b++; }
	if (neg) v = -v;
	if (v > 0x7fffffffLL) v = 0x7fffffffLL;
	if (v < -0x80000000LL) v = -0x80000000LL;
	return (int)v;
}

inline bool starts_with(const Line &l, const char *w)
{
	size_t n = std::strlen(w);
	return (size_t)(l.e - l.b) >= n && std::memcmp(l.b, w, n) == 0;
}

struct Property {
	bool list = false;
	int size = 0, count_size = 0; // bytes of the scalar (or of the list items) and of the list count
	char kind = 'i', count_kind = 'u'; // 'f' float, 'i' signed, 'u' unsigned
};

bool scalar_type(const std::string &t, int &size, char &kind)
{
	if (t == "char" || t == "int8") { size = 1; kind = 'i'; }
	else if (t == "uchar" || t == "uint8") { size = 1; kind = 'u'; }
	else if (t == "short" || t == "int16") { size = 2; kind = 'i'; }
	else if (t == "ushort" || t == "uint16") { size = 2; kind = 'u'; }
	else if (t == "int" || t == "int32") { size = 4; kind = 'i'; }
	else if (t == "uint" || t == "uint32") { size = 4; kind = 'u'; }
	else if (t == "float" || t == "float32") { size = 4; kind = 'f'; }
	else if (t == "double" || t == "float64") { size = 8; kind = 'f'; }
	else return false;
	return true;
}

double read_scalar(const char *p, int size, char kind)
{
	switch (kind) {
	case 'f':
		if (size == 4) { float v; std::memcpy(&v, p, 4); return v; }
		{ double v; std::memcpy(&v, p, 8); return v; }
	case 'i':
		if (size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; }
		if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
		{ int32_t v; std::memcpy(&v, p, 4); return v; }
	default:
		if (size == 1) { uint8_t v; std::memcpy(&v, p, 1); return v; }
		if (size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
		{ uint32_t v; std::memcpy(&v, p, 4); return v; }
	}
}

} // namespace

void ply::Clear(void)
{
	elements.clear();
	vertex_data.clear();
	face_index.clear();
	face_begin.assign(1, 0);
	binary = false;
	error.clear();
}

bool ply::Load(const char *filename)
{
	Clear();
	int fd = ::open(filename, O_RDONLY);
	if (fd < 0) { error = std::string("cannot open ") + filename; return false; }
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size == 0) { ::close(fd); error = std::string("empty file ") + filename; return false; }
	const size_t size = (size_t)st.st_size;
	void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
	::close(fd);
	if (map == MAP_FAILED) { error = std::string("cannot map ") + filename; return false; }
	Cursor c{static_cast<const char *>(map), static_cast<const char *>(map), static_cast<const char *>(map) + size};
	bool ok = true;

	// ---- header (ply.cpp:20-35 with LoadElement, ply.cpp:121-158): every line that starts with "element" opens an
	// element (second token = name, third = count through atoi) whose properties are the lines that follow with
	// first token "property"; a line that starts with "end_header" ends the header
	std::vector<std::vector<Property>> props;
	bool little = false, big = false;
	Line l{c.begin, c.begin};
	const char *oldpos = c.tellg();
	while (c.getline(l)) {
		if (starts_with(l, "format")) {
			std::string s(l.b, l.e);
			little = s.find("binary_little_endian") != std::string::npos;
			big = s.find("binary_big_endian") != std::string::npos;
		}
		if (starts_with(l, "element")) { // LoadElement
			c.seekg(oldpos);
			const char *old2 = c.tellg();
			Line el_line{c.begin, c.begin};
			c.getline(el_line);
			element el;
			const char *q = el_line.b, *tb = el_line.b, *te = el_line.b;
			next_token(q, el_line.e, tb, te);                 // "element"
			next_token(q, el_line.e, tb, te);                 // name (a missing token repeats the previous one)
			el.name.assign(tb, te);
			next_token(q, el_line.e, tb, te);
			el.num_of_elements = (size_t)token_atoi(tb, te);
			std::vector<Property> pl;
			old2 = c.tellg();
			while (c.getline(el_line)) {
				const char *q2 = el_line.b, *b2 = el_line.b, *e2 = el_line.b;
				if (!next_token(q2, el_line.e, b2, e2) || std::string(b2, e2) != "property") break;
				el.num_of_property++;
				Property pr;
				std::string t1, t2, t3;
				if (next_token(q2, el_line.e, b2, e2)) t1.assign(b2, e2);
				if (t1 == "list") {
					pr.list = true;
					if (next_token(q2, el_line.e, b2, e2)) t2.assign(b2, e2);
					if (next_token(q2, el_line.e, b2, e2)) t3.assign(b2, e2);
					if (!scalar_type(t2, pr.count_size, pr.count_kind) || !scalar_type(t3, pr.size, pr.kind)) pr.size = 0;
				} else if (!scalar_type(t1, pr.size, pr.kind)) pr.size = 0;
				pl.push_back(pr);
				old2 = c.tellg();
			}
			c.seekg(old2); // back to the line that ended the property list, so that the header loop sees it
			elements.push_back(el);
			props.push_back(pl);
		}
		if (starts_with(l, "end_header")) break;
		oldpos = c.tellg();
	}
	if (elements.size() < 2) {
		error = "ply header declares fewer than two elements (the reference reads elements[0] as vertices and elements[1] as faces)";
		ok = false;
	}
	binary = little || big;
	if (ok && big) { error = "binary_big_endian ply files are not supported"; ok = false; }

	if (ok && !binary) {
		// ---- vertices (ply.cpp:38-59): one line each, six atof tokens; a missing token repeats the previous one
		const size_t nv = elements[0].num_of_elements;
		vertex_data.resize(nv * 6);
		for (size_t i = 0; i < nv; i++) {
			c.getline(l); // a failed getline leaves (or erases) the line as the stream rules above say, and the loop goes on
			const char *q = l.b, *tb = l.b, *te = l.b;
			for (int j = 0; j < 6; j++) {
				next_token(q, l.e, tb, te);
				vertex_data[6 * i + j] = token_atof(tb, te);
			}
		}
		// ---- faces (ply.cpp:63-79): count, then that many atoi tokens
		const size_t nf = elements[1].num_of_elements;
		face_begin.reserve(nf + 1);
		face_index.reserve(nf * 3);
		for (size_t i = 0; i < nf; i++) {
			c.getline(l);
			const char *q = l.b, *tb = l.b, *te = l.b;
			next_token(q, l.e, tb, te);
			const int cnt = token_atoi(tb, te);
			if (cnt < 0) { error = "negative corner count in face " + std::to_string(i); ok = false; break; } // (size_t)atoi: the reference would try to allocate 2^64 entries
			for (int j = 0; j < cnt; j++) {
				next_token(q, l.e, tb, te);
				face_index.push_back(token_atoi(tb, te));
			}
			face_begin.push_back(face_index.size());
		}
	} else if (ok) {
		// ---- binary little-endian (an addition): first six scalar properties of element 0, first list of element 1
		for (size_t k = 0; k < 2 && ok; k++)
			for (auto &pr : props[k])
				if (pr.size == 0) { error = "unknown property type in ply header"; ok = false; }
		const char *p = c.p;
		const size_t nv = ok ? elements[0].num_of_elements : 0;
		vertex_data.assign(nv * 6, 0.0);
		for (size_t i = 0; i < nv && ok; i++) {
			int col = 0;
			for (auto &pr : props[0]) {
				if (pr.list) {
					if (p + pr.count_size > c.end) { ok = false; break; }
					size_t n = (size_t)read_scalar(p, pr.count_size, pr.count_kind);
					p += pr.count_size + n * pr.size;
				} else {
					if (p + pr.size > c.end) { ok = false; break; }
					if (col < 6) vertex_data[6 * i + col] = read_scalar(p, pr.size, pr.kind);
					col++;
					p += pr.size;
				}
			}
			if (p > c.end) ok = false;
		}
		const size_t nf = ok ? elements[1].num_of_elements : 0;
		face_begin.reserve(nf + 1);
		for (size_t i = 0; i < nf && ok; i++) {
			bool first_list = true;
			for (auto &pr : props[1]) {
				if (pr.list) {
					if (p + pr.count_size > c.end) { ok = false; break; }
					size_t n = (size_t)read_scalar(p, pr.count_size, pr.count_kind);
					p += pr.count_size;
					if (p + n * pr.size > c.end) { ok = false; break; }
					if (first_list)
						for (size_t j = 0; j < n; j++) face_index.push_back((int32_t)read_scalar(p + j * pr.size, pr.size, pr.kind));
					first_list = false;
					p += n * pr.size;
				} else {
					p += pr.size;
				}
			}
			if (p > c.end) ok = false;
			face_begin.push_back(face_index.size());
		}
		if (!ok && error.empty()) error = "binary ply file ends before its declared elements do";
	}
	munmap(map, size);
	if (!ok) { vertex_data.clear(); face_index.clear(); face_begin.assign(1, 0); }
	return ok;
}

size_t AppendPly(const ply &p, const std::string &object_name, PlyNormals mode, MeshTables &out)
{
	if (mode == PlyNormals::Auto) mode = p.has_normals() ? PlyNormals::File : PlyNormals::Smooth;
	const size_t nv = p.n_vertices(), nf = p.n_faces();
	const int64_t v0 = (int64_t)(out.v.size() / 3), n0 = (int64_t)(out.vn.size() / 3);
	out.v.reserve(out.v.size() + 3 * nv);
	for (size_t i = 0; i < nv; i++)
		for (int k = 0; k < 3; k++) out.v.push_back(p.vertex(i)[k]);
	auto face_normal = [&](size_t f, double n[3]) { // (v1 - v0) x (v2 - v1), object.cpp:117 before normalisation
		const uint64_t b = p.face_begin[f];
		const double *a = p.vertex((size_t)p.face_index[b]), *bb = p.vertex((size_t)p.face_index[b + 1]), *cc = p.vertex((size_t)p.face_index[b + 2]);
		const double e0[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]}, e1[3] = {cc[0] - bb[0], cc[1] - bb[1], cc[2] - bb[2]};
		n[0] = e0[1] * e1[2] - e0[2] * e1[1];
		n[1] = -(e0[0] * e1[2] - e0[2] * e1[0]);
		n[2] = e0[0] * e1[1] - e0[1] * e1[0];
	};
	auto valid = [&](size_t f) {
		const uint64_t b = p.face_begin[f], e = p.face_begin[f + 1];
		if (e - b < 3) return false;
		for (uint64_t k = b; k < e; k++)
			if (p.face_index[k] < 0 || (size_t)p.face_index[k] >= nv) return false;
		return true;
	};
	std::vector<double> smooth;
	if (mode == PlyNormals::File) {
		for (size_t i = 0; i < nv; i++)
			for (int k = 0; k < 3; k++) out.vn.push_back(p.normal(i)[k]);
	} else if (mode == PlyNormals::Smooth) {
		smooth.assign(3 * nv, 0.0);
		for (size_t f = 0; f < nf; f++) {
			if (!valid(f)) continue;
			double n[3];
			face_normal(f, n); // length = twice the area of the first fan triangle: an area weight
			for (uint64_t k = p.face_begin[f]; k < p.face_begin[f + 1]; k++)
				for (int a = 0; a < 3; a++) smooth[3 * (size_t)p.face_index[k] + a] += n[a];
		}
		for (size_t i = 0; i < nv; i++) {
			double *n = &smooth[3 * i];
			if (n[0] == 0.0 && n[1] == 0.0 && n[2] == 0.0) n[2] = 1.0; // vertex used by no face (or only by degenerate ones)
			for (int k = 0; k < 3; k++) out.vn.push_back(n[k]);
		}
	}
	size_t dropped = 0;
	for (size_t f = 0; f < nf; f++) {
		if (!valid(f)) { dropped++; continue; }
		int64_t flat_index = -1;
		if (mode == PlyNormals::Flat) {
			double n[3];
			face_normal(f, n);
			flat_index = (int64_t)(out.vn.size() / 3);
			for (int k = 0; k < 3; k++) out.vn.push_back(n[k]);
		}
		for (uint64_t k = p.face_begin[f]; k < p.face_begin[f + 1]; k++) {
			const int64_t vi = p.face_index[k];
			out.corners.push_back(v0 + vi);
			out.corners.push_back(-1);
			out.corners.push_back(mode == PlyNormals::Flat ? flat_index : n0 + vi);
		}
		out.face_begin.push_back(out.corners.size() / 3);
	}
	out.object_begin.push_back(out.face_begin.size() - 1);
	out.object_names.push_back(object_name);
	return dropped;
}

} // namespace mi355host
