// TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own PLY reader (ply.cpp, compiled unmodified
// from the reference checkout) so that tests can pin the product's PLY parser against it.
#include "ply.h"
#include <cstdint>
#include <cstring>

extern "C" {

void *ref_ply_load(const char *filename)
{
	ply *p = new ply();
	p->Load(filename); // ply.cpp:13-81
	return p;
}

void ref_ply_counts(void *h, uint64_t *n_vertices, uint64_t *n_faces, uint64_t *n_indices, uint64_t *n_elements)
{
	ply *p = static_cast<ply *>(h);
	uint64_t n = 0;
	for (auto &f : p->faces) n += f.size();
	*n_vertices = p->vertices.size();
	*n_faces = p->faces.size();
	*n_indices = n;
	*n_elements = p->elements.size();
}

// vertices: [n][6] doubles (position, normal); face_begin: [n_faces + 1]; face_index: [n_indices]
void ref_ply_tables(void *h, double *vertices, uint64_t *face_begin, int32_t *face_index)
{
	ply *p = static_cast<ply *>(h);
	for (size_t i = 0; i < p->vertices.size(); i++)
		for (int k = 0; k < 6; k++) vertices[6 * i + k] = p->vertices[i][k / 3].e[k % 3];
	uint64_t n = 0;
	for (size_t i = 0; i < p->faces.size(); i++) {
		face_begin[i] = n;
		for (int idx : p->faces[i]) face_index[n++] = idx;
	}
	face_begin[p->faces.size()] = n;
}

// element k of the header: name (copied, at most cap-1 characters), declared count, number of property lines
void ref_ply_element(void *h, uint64_t k, char *name, uint64_t cap, uint64_t *count, uint64_t *properties)
{
	ply *p = static_cast<ply *>(h);
	std::strncpy(name, p->elements[k]->name.c_str(), cap - 1);
	name[cap - 1] = 0;
	*count = p->elements[k]->num_of_elements;
	*properties = p->elements[k]->num_of_property;
}

void ref_ply_free(void *h) { delete static_cast<ply *>(h); }
}
