/*
 * ref_harness_bvh.cpp -- drives the reference's own BVH code (bvh.cpp: SAHBVHNode construction,
 * BVHNode::Hit; hittablelist.cpp: HittableList::Hit), compiled unmodified, over leaf objects whose
 * Hit/BoundingBox forward to the oracle's polygon restatement.  What this pins: the oracle's
 * restated tree construction and closest-hit selection (both children, same interval, right child
 * on ties) against the compiled reference.  What it does not pin: the polygon test inside the
 * leaves, which is the oracle's on both sides.
 *
 * TEST INFRASTRUCTURE ONLY.  Built with -DMATERIALNODE_H so that hittable.h does not pull in the
 * material-graph header (which needs Boost.Preprocessor and ImNodes, absent in the build container).
 * The two out-of-line Hittable members live in the reference's object.cpp, which cannot be compiled
 * here (it includes <boost/optional.hpp> through obj.h); the definitions below only complete the link
 * and are never called.
 */
#include <memory>
#include <vector>
#include <limits>
#include "bvh.h"
#include "hittablelist.h"
#include "rt_oracle.h"

class Pdf { public: virtual ~Pdf() {} };
std::unique_ptr<Pdf> Hittable::GeneratePdfObject(const dvec3 &) { return std::unique_ptr<Pdf>(); }
void Hittable::SetMaterial(NodeMaterial *mat) { mat_ptr = mat; }

namespace {
struct Leaf : public Hittable {
	const orc_scene *scene;
	uint64_t poly;
	int64_t last_tri;
	bool Hit(const ray &r, double t_min, double t_max, HitRecord &rec) const override
	{
		double o[3] = {r.A[0], r.A[1], r.A[2]}, d[3] = {r.B[0], r.B[1], r.B[2]}, t, p[3], n[3];
		int64_t tri;
		if (!orc_polygon_hit(scene, poly, o, d, t_min, t_max, &t, p, n, &tri))
			return false;
		rec.t = t;
		rec.p = dvec3(p[0], p[1], p[2]);
		rec.normal = dvec3(n[0], n[1], n[2]);
		rec.hit_object = this;
		rec.hit_object_id = object_id;
		return true;
	}
	bool BoundingBox(AABB &box) const override
	{
		double a[3], b[3];
		orc_scene_polygon_bbox(scene, poly, a, b);
		box = AABB(dvec3(a[0], a[1], a[2]), dvec3(b[0], b[1], b[2]));
		return true;
	}
};

struct World {
	std::vector<std::shared_ptr<Hittable>> models; /* per-object SAHBVHNode, as ObjModel::models */
	std::shared_ptr<SAHBVHNode> bvh;
	HittableList list;
};

void walk(const Hittable *h, uint64_t &nodes, uint32_t depth, uint32_t &max_depth)
{
	const BVHNode *n = dynamic_cast<const BVHNode *>(h);
	if (!n) return;
	nodes++;
	if (depth > max_depth) max_depth = depth;
	walk(n->left.get(), nodes, depth + 1, max_depth);
	if (n->right.get() != n->left.get()) walk(n->right.get(), nodes, depth + 1, max_depth);
}
}

extern "C" {

/* mirrors ObjModel::ObjModel's tree construction (one SAH tree per object, one over the objects) */
void *ref_world_create(const orc_scene *scene, const uint64_t *object_begin, uint32_t n_objects)
{
	World *w = new World();
	w->models.resize(n_objects);
	for (uint32_t i = 0; i < n_objects; i++) {
		std::vector<std::shared_ptr<Hittable>> model;
		for (uint64_t f = object_begin[i]; f < object_begin[i + 1]; f++) {
			std::shared_ptr<Leaf> l = std::make_shared<Leaf>();
			l->scene = scene;
			l->poly = f;
			l->object_id = i;
			l->mat_ptr = nullptr;
			model.push_back(l);
			w->list.list.push_back(l);
		}
		w->models[i] = std::make_shared<SAHBVHNode>(model);
	}
	auto v(w->models);
	w->bvh = std::make_shared<SAHBVHNode>(v);
	return w;
}
void ref_world_free(void *p) { delete static_cast<World *>(p); }
void ref_world_stats(void *p, uint64_t *nodes, uint32_t *depth)
{
	World *w = static_cast<World *>(p);
	*nodes = 0;
	*depth = 0;
	walk(w->bvh.get(), *nodes, 1, *depth);
}
/* mode 0: BVHNode::Hit through the tree; mode 1: HittableList::Hit. rays: 8 floats each. */
uint64_t ref_world_trace(void *p, const float *rays, uint64_t n, int mode, int32_t *out_poly, double *out_t)
{
	World *w = static_cast<World *>(p);
	uint64_t hits = 0;
	for (uint64_t i = 0; i < n; i++) {
		const float *q = rays + 8 * i;
		ray r(dvec3(q[0], q[1], q[2]), dvec3(q[4], q[5], q[6]));
		double t_max = std::isinf(q[7]) ? std::numeric_limits<double>::max() : (double)q[7];
		HitRecord rec;
		bool h = mode == 0 ? w->bvh->Hit(r, q[3], t_max, rec) : w->list.Hit(r, q[3], t_max, rec);
		out_poly[i] = h ? (int32_t) static_cast<const Leaf *>(rec.hit_object)->poly : -1;
		out_t[i] = h ? rec.t : std::numeric_limits<double>::infinity();
		hits += h;
	}
	return hits;
}
int ref_world_occluded(void *p, const float *q)
{
	World *w = static_cast<World *>(p);
	ray r(dvec3(q[0], q[1], q[2]), dvec3(q[4], q[5], q[6]));
	return w->bvh->Occluded(r, q[3], q[7]) ? 1 : 0;
}
}
