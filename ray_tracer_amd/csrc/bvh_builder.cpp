#include "bvh_builder.h"
#include "accept_region.h"
#include "presplit.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <thread>

namespace mi355rt {
namespace {

struct Box {
	float lo[3], hi[3];
	void reset()
	{
		for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
	}
	void grow(const Box &b)
	{
		for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
	}
	void grow_point(const float *p)
	{
		for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
	}
	float half_area() const
	{
		float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return dx * dy + dy * dz + dz * dx;
	}
};

// largest float <= x / smallest float >= x: the device box test stays conservative
inline float round_down(double x)
{
	float f = (float)x;
	if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
	return f;
}
inline float round_up(double x)
{
	float f = (float)x;
	if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
	return f;
}

struct TempNode {
	Box box;
	int32_t left = -1, right = -1; // temp node indices; leaf when left < 0
	uint32_t first = 0, count = 0; // range in idx (a subtree owns a contiguous range: the builder partitions in place)
};

struct Builder {
	const BvhBuildOptions &opt;
	std::vector<Box> prim_box;
	std::vector<float> centroid; // 3 per prim
	std::vector<uint32_t> prim_tri; // prim (reference) -> fan triangle: the identity, followed by the additional references of triangles that were cut
	std::vector<uint32_t> idx;
	std::vector<TempNode> temp;
	std::atomic<uint32_t> next_node{0};
	std::atomic<int> live_threads{1};
	int max_threads = 1;

	explicit Builder(const BvhBuildOptions &o) : opt(o) {}

	uint32_t alloc_node() { return next_node.fetch_add(1); }

	void make_leaf(uint32_t me, uint32_t begin, uint32_t end)
	{
		temp[me].left = temp[me].right = -1;
		temp[me].first = begin;
		temp[me].count = end - begin;
	}

	void build(uint32_t me, uint32_t begin, uint32_t end, int depth)
	{
		const uint32_t n = end - begin;
		Box box, cbox;
		box.reset();
		cbox.reset();
		for (uint32_t i = begin; i < end; i++) {
			box.grow(prim_box[idx[i]]);
			cbox.grow_point(&centroid[3 * (size_t)idx[i]]);
		}
		temp[me].box = box;
		temp[me].first = begin;
		temp[me].count = n;
		if (n == 1) { make_leaf(me, begin, end); return; }

		uint32_t mid = 0;
		bool split_found = false;
		if (depth < opt.sah_depth_limit) {
			const int B = opt.bins;
			float best_cost = std::numeric_limits<float>::infinity();
			int best_axis = -1, best_bin = -1;
			std::vector<Box> bin_box(B), right_box(B);
			std::vector<uint32_t> bin_cnt(B);
			for (int axis = 0; axis < 3; axis++) {
				float c0 = cbox.lo[axis], c1 = cbox.hi[axis];
				if (!(c1 > c0)) continue;
				float scale = (float)B / (c1 - c0);
				for (int b = 0; b < B; b++) { bin_box[b].reset(); bin_cnt[b] = 0; }
				for (uint32_t i = begin; i < end; i++) {
					uint32_t p = idx[i];
					int b = (int)((centroid[3 * (size_t)p + axis] - c0) * scale);
					b = b < 0 ? 0 : (b >= B ? B - 1 : b);
					bin_box[b].grow(prim_box[p]);
					bin_cnt[b]++;
				}
				Box acc;
				acc.reset();
				for (int b = B - 1; b > 0; b--) { acc.grow(bin_box[b]); right_box[b] = acc; }
				acc.reset();
				uint32_t nl = 0;
				for (int b = 0; b < B - 1; b++) {
					acc.grow(bin_box[b]);
					nl += bin_cnt[b];
					uint32_t nr = n - nl;
					if (nl == 0 || nr == 0) continue;
					float cost = acc.half_area() * (float)nl + right_box[b + 1].half_area() * (float)nr;
					if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
				}
			}
			if (best_axis >= 0) {
				float parent_area = box.half_area();
				float split_cost = opt.cost_node + opt.cost_tri * best_cost / (parent_area > 0 ? parent_area : 1e-30f);
				float leaf_cost = opt.cost_tri * (float)n;
				if ((int)n <= opt.max_leaf && leaf_cost <= split_cost) { make_leaf(me, begin, end); return; }
				float c0 = cbox.lo[best_axis], scale = (float)opt.bins / (cbox.hi[best_axis] - c0);
				auto it = std::partition(idx.begin() + begin, idx.begin() + end, [&](uint32_t p) {
					int b = (int)((centroid[3 * (size_t)p + best_axis] - c0) * scale);
					b = b < 0 ? 0 : (b >= opt.bins ? opt.bins - 1 : b);
					return b <= best_bin;
				});
				mid = (uint32_t)(it - idx.begin());
				split_found = mid > begin && mid < end;
			}
		}
		if (!split_found) {
			if ((int)n <= opt.max_leaf) { make_leaf(me, begin, end); return; }
			// all centroids coincide, or the SAH depth budget is spent: balanced split on the widest axis
			int axis = 0;
			float ext = -1.0f;
			for (int a = 0; a < 3; a++) if (cbox.hi[a] - cbox.lo[a] > ext) { ext = cbox.hi[a] - cbox.lo[a]; axis = a; }
			mid = begin + n / 2;
			std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end, [&](uint32_t a, uint32_t b) {
				float ca = centroid[3 * (size_t)a + axis], cb = centroid[3 * (size_t)b + axis];
				return ca < cb || (ca == cb && a < b);
			});
		}
		uint32_t l = alloc_node(), r = alloc_node();
		temp[me].left = (int32_t)l;
		temp[me].right = (int32_t)r;
		bool spawn = n > 65536 && live_threads.load() < max_threads;
		if (spawn) {
			live_threads.fetch_add(1);
			auto fut = std::async(std::launch::async, [this, l, begin, mid, depth] { build(l, begin, mid, depth + 1); });
			build(r, mid, end, depth + 1);
			fut.get();
			live_threads.fetch_sub(1);
		} else {
			build(l, begin, mid, depth + 1);
			build(r, mid, end, depth + 1);
		}
	}
};

struct Flattener {
	const Builder &b;
	Bvh2 &out;
	// returns the link for temp node t; interior nodes are emitted in preorder
	int32_t emit(int32_t t, uint32_t depth)
	{
		const TempNode &tn = b.temp[t];
		if (tn.left < 0) {
			uint32_t first = (uint32_t)out.tri_index.size();
			for (uint32_t i = 0; i < tn.count; i++) out.tri_index.push_back(b.prim_tri[b.idx[tn.first + i]]);
			out.n_leaves++;
			return make_leaf_link(first, tn.count);
		}
		if (depth + 1 > out.depth) out.depth = depth + 1;
		uint32_t me = (uint32_t)out.nodes.size();
		out.nodes.emplace_back();
		const Box &b0 = b.temp[tn.left].box, &b1 = b.temp[tn.right].box;
		int32_t c0 = emit(tn.left, depth + 1);
		int32_t c1 = emit(tn.right, depth + 1);
		rt_bvh2_node &n = out.nodes[me];
		for (int a = 0; a < 3; a++) { n.lo0[a] = b0.lo[a]; n.hi0[a] = b0.hi[a]; n.lo1[a] = b1.lo[a]; n.hi1[a] = b1.hi[a]; }
		n.child0 = c0;
		n.child1 = c1;
		n.pad[0] = n.pad[1] = 0;
		return (int32_t)me;
	}
};

// Collapse to 8-wide nodes and quantise.
// Which binary-tree nodes become the children of which 8-wide node.  The greedy rule (open the child with the largest area until
// there are eight) leaves the nodes at the bottom of the tree half empty: a subtree of nine leaves becomes a full node and a node of
// two, and on a scene of 490 k triangles the average node held 4.2 children.  The cut is instead chosen to minimise the surface-area
// cost of the wide tree (Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017,
// section 4): cost[n][i] is the cheapest way to represent subtree n by at most i roots, each a leaf of at most kWideLeafMax
// triangles or an 8-wide node,
//   cost[n][1] = min(area(n) * tris(n) * c_tri  [tris(n) <= kWideLeafMax],  area(n) * c_node + min_k cost[l][k] + cost[r][8 - k])
//   cost[n][i] = min(cost[n][i - 1], min_k cost[l][k] + cost[r][i - k])
// evaluated children first (a child's index is larger than its parent's), and dec[] records which option won.
constexpr uint32_t kWideLeafMax = 4; // 8 leaf children x 4 triangles: what a node's 5-bit offsets address

struct WideCut {
	std::vector<float> cost;  // [7 * node + i - 1]
	std::vector<uint8_t> dec; // i = 1: 0 leaf, k > 0: 8-wide node whose left subtree gives k children; i > 1: 0 as for i - 1, k: left k roots
	bool enabled = false;

	void solve(const Builder &b, uint32_t n_nodes, float c_node, float c_tri)
	{
		enabled = true;
		cost.assign(7 * (size_t)n_nodes, 0.0f);
		dec.assign(7 * (size_t)n_nodes, 0);
		const float inf = std::numeric_limits<float>::infinity();
		for (uint32_t t = n_nodes; t-- > 0;) {
			const TempNode &tn = b.temp[t];
			float *c = &cost[7 * (size_t)t];
			uint8_t *d = &dec[7 * (size_t)t];
			const float area = tn.box.half_area();
			const float as_leaf = tn.count <= kWideLeafMax || tn.left < 0 ? area * (float)tn.count * c_tri : inf;
			if (tn.left < 0) {
				for (int i = 0; i < 7; i++) { c[i] = as_leaf; d[i] = 0; }
				continue;
			}
			const float *cl = &cost[7 * (size_t)tn.left], *cr = &cost[7 * (size_t)tn.right];
			float best = inf;
			uint8_t bk = 1;
			for (int k = 1; k <= 7; k++) {
				const float v = cl[k - 1] + cr[8 - k - 1];
				if (v < best) { best = v; bk = (uint8_t)k; }
			}
			const float as_node = area * c_node + best;
			if (as_leaf <= as_node) { c[0] = as_leaf; d[0] = 0; }
			else { c[0] = as_node; d[0] = bk; }
			for (int i = 2; i <= 7; i++) {
				float v = c[i - 2];
				uint8_t dk = 0;
				for (int k = 1; k < i; k++) {
					const float w = cl[k - 1] + cr[i - k - 1];
					if (w < v) { v = w; dk = (uint8_t)k; }
				}
				c[i - 1] = v;
				d[i - 1] = dk;
			}
		}
	}
	bool is_leaf(const Builder &b, int32_t t) const { return enabled ? dec[7 * (size_t)t] == 0 : b.temp[t].left < 0; }
	// the roots subtree t is cut into when it may use up to i of them
	void roots(const Builder &b, int32_t t, int i, int32_t *kids, int &nk) const
	{
		while (i > 1 && dec[7 * (size_t)t + i - 1] == 0) i--;
		if (i == 1) { kids[nk++] = t; return; }
		const int k = dec[7 * (size_t)t + i - 1];
		roots(b, b.temp[t].left, k, kids, nk);
		roots(b, b.temp[t].right, i - k, kids, nk);
	}
};

struct WideFlattener {
	const Builder &b;
	BvhWide &out;
	const WideCut &cut;

	static int floor_log2_cells(double extent) // smallest e with 2^e * 255 >= extent
	{
		if (!(extent > 0.0)) return -126;
		int e = (int)std::ceil(std::log2(extent / 255.0));
		while (std::ldexp(1.0, e) * 255.0 < extent) e++;
		if (e < -126) e = -126;
		if (e > 127) e = 127;
		return e;
	}

	void emit(int32_t t, uint32_t me, uint32_t depth)
	{
		if (depth > out.depth) out.depth = depth;
		// ---- pick up to 8 children: open the interior child with the largest area until full
		int32_t kids[8];
		int nk = 0;
		if (cut.is_leaf(b, t)) kids[nk++] = t; // the whole tree is one leaf
		else if (cut.enabled) {
			const int k = cut.dec[7 * (size_t)t];
			cut.roots(b, b.temp[t].left, k, kids, nk);
			cut.roots(b, b.temp[t].right, 8 - k, kids, nk);
		} else { kids[nk++] = b.temp[t].left; kids[nk++] = b.temp[t].right; }
		while (!cut.enabled && nk < 8) {
			int best = -1;
			float best_area = -1.0f;
			for (int i = 0; i < nk; i++)
				if (b.temp[kids[i]].left >= 0 && b.temp[kids[i]].box.half_area() > best_area) { best_area = b.temp[kids[i]].box.half_area(); best = i; }
			if (best < 0) break;
			const int32_t open = kids[best];
			kids[best] = b.temp[open].left;
			kids[nk++] = b.temp[open].right;
		}
		// ---- node box and cell sizes
		Box box;
		box.reset();
		for (int i = 0; i < nk; i++) box.grow(b.temp[kids[i]].box);
		WideNode nd;
		std::memset(&nd, 0, sizeof(nd));
		double scale[3];
		for (int a = 0; a < 3; a++) {
			nd.org[a] = box.lo[a];
			int e = floor_log2_cells((double)box.hi[a] - (double)box.lo[a]);
			nd.e[a] = (uint8_t)(e + 127);
			scale[a] = std::ldexp(1.0, e);
		}
		// ---- slots: the octant of the child's centre relative to the node's centre, nearest free slot otherwise
		int slot_of[8], child_in_slot[8];
		for (int sl = 0; sl < 8; sl++) child_in_slot[sl] = -1;
		int order[8];
		float dist[8];
		for (int i = 0; i < nk; i++) {
			order[i] = i;
			float dsum = 0.0f;
			for (int a = 0; a < 3; a++) {
				float c = 0.5f * (b.temp[kids[i]].box.lo[a] + b.temp[kids[i]].box.hi[a]), m = 0.5f * (box.lo[a] + box.hi[a]);
				dsum += std::fabs(c - m);
			}
			dist[i] = dsum;
		}
		std::sort(order, order + nk, [&](int x, int y) { return dist[x] > dist[y] || (dist[x] == dist[y] && x < y); });
		if (b.opt.slot_assignment >= 1) {
			// the assignment of children to slots that maximises the sum over the children of (centre - node centre) . (the slot's
			// octant direction), as Ylitie, Karras and Laine choose it (HPG 2017, section 4.2), found exactly: best[mask] = the best
			// total for the first popcount(mask) children placed in the slots of mask
			float score[8][8];
			for (int i = 0; i < nk; i++)
				for (int sl = 0; sl < 8; sl++) {
					float t = 0.0f;
					for (int a = 0; a < 3; a++) {
						const float c = 0.5f * (b.temp[kids[i]].box.lo[a] + b.temp[kids[i]].box.hi[a]), m = 0.5f * (box.lo[a] + box.hi[a]);
						const float w = b.opt.slot_assignment == 2 ? 1.0f / std::max(box.hi[a] - box.lo[a], 1e-30f) : 1.0f;
						t += (((sl >> a) & 1) ? c - m : m - c) * w;
					}
					score[i][sl] = t;
				}
			float best[256];
			uint8_t from[256];
			best[0] = 0.0f;
			for (int mask = 1; mask < 256; mask++) {
				const int i = __builtin_popcount((unsigned)mask) - 1; // the child this step places
				best[mask] = -std::numeric_limits<float>::infinity();
				from[mask] = 0;
				if (i >= nk) continue;
				for (int sl = 0; sl < 8; sl++)
					if ((mask >> sl) & 1) {
						const float v = best[mask ^ (1 << sl)] + score[i][sl];
						if (v > best[mask]) { best[mask] = v; from[mask] = (uint8_t)sl; }
					}
			}
			int top = -1;
			for (int mask = 0; mask < 256; mask++)
				if (__builtin_popcount((unsigned)mask) == nk && (top < 0 || best[mask] > best[top])) top = mask;
			for (int mask = top, i = nk - 1; i >= 0; i--) {
				const int sl = from[mask];
				slot_of[i] = sl;
				child_in_slot[sl] = i;
				mask ^= 1 << sl;
			}
		} else
		for (int oi = 0; oi < nk; oi++) {
			const int i = order[oi];
			int want = 0;
			for (int a = 0; a < 3; a++) {
				float c = 0.5f * (b.temp[kids[i]].box.lo[a] + b.temp[kids[i]].box.hi[a]), m = 0.5f * (box.lo[a] + box.hi[a]);
				if (c > m) want |= 1 << a;
			}
			int best_slot = -1, best_d = 99;
			for (int sl = 0; sl < 8; sl++) {
				if (child_in_slot[sl] >= 0) continue;
				int d = __builtin_popcount((unsigned)(sl ^ want));
				if (d < best_d) { best_d = d; best_slot = sl; }
			}
			slot_of[i] = best_slot;
			child_in_slot[best_slot] = i;
		}
		// ---- quantise, collect interior children and leaf triangles in slot order
		for (int a = 0; a < 3; a++)
			for (int sl = 0; sl < 8; sl++) { nd.qlo[a][sl] = 255; nd.qhi[a][sl] = 0; }
		int n_interior = 0;
		nd.tri_base = (uint32_t)out.tri_index.size();
		for (int sl = 0; sl < 8; sl++) {
			const int i = child_in_slot[sl];
			if (i < 0) continue;
			const TempNode &c = b.temp[kids[i]];
			for (int a = 0; a < 3; a++) {
				double lo = std::floor(((double)c.box.lo[a] - (double)nd.org[a]) / scale[a]);
				double hi = std::ceil(((double)c.box.hi[a] - (double)nd.org[a]) / scale[a]);
				lo = std::max(0.0, std::min(255.0, lo));
				hi = std::max(0.0, std::min(255.0, hi));
				// the cells must cover the child's box (they do by construction; the node box covers it)
				while ((double)nd.org[a] + lo * scale[a] > (double)c.box.lo[a] && lo > 0.0) lo -= 1.0;
				while ((double)nd.org[a] + hi * scale[a] < (double)c.box.hi[a] && hi < 255.0) hi += 1.0;
				nd.qlo[a][sl] = (uint8_t)lo;
				nd.qhi[a][sl] = (uint8_t)hi;
			}
			if (!cut.is_leaf(b, kids[i])) {
				nd.imask |= (uint8_t)(1u << sl);
				n_interior++;
			} else {
				const uint32_t off = (uint32_t)out.tri_index.size() - nd.tri_base;
				nd.meta[sl] = (uint8_t)((c.count << 5) | off);
				for (uint32_t k = 0; k < c.count; k++) out.tri_index.push_back(b.prim_tri[b.idx[c.first + k]]);
			}
		}
		nd.child_base = (uint32_t)out.nodes.size();
		out.nodes.resize(out.nodes.size() + n_interior);
		out.nodes[me] = nd;
		uint32_t rel = 0;
		for (int sl = 0; sl < 8; sl++) {
			const int i = child_in_slot[sl];
			if (i < 0 || cut.is_leaf(b, kids[i])) continue;
			emit(kids[i], nd.child_base + rel, depth + 1);
			rel++;
		}
		(void)slot_of;
	}
};

} // namespace

void build_bvh(const HostScene &scene, const BvhBuildOptions &opt_in, const Presplit *cut, Bvh2 &out, BvhWide &wide)
{
	auto t_start = std::chrono::steady_clock::now();
	BvhBuildOptions opt = opt_in;
	opt.max_leaf = std::max(1, std::min(opt.max_leaf, 4)); // a wide node addresses at most 32 triangles: 8 leaves x 4
	opt.bins = std::max(4, std::min(opt.bins, 256));
	out = Bvh2();
	wide = BvhWide();
	const uint64_t n = scene.n_tris;
	const float inf = std::numeric_limits<float>::infinity();
	auto empty_child = [&](rt_bvh2_node &nd, int which) {
		float *lo = which ? nd.lo1 : nd.lo0, *hi = which ? nd.hi1 : nd.hi0;
		for (int a = 0; a < 3; a++) { lo[a] = inf; hi[a] = -inf; }
	};
	if (n == 0) {
		rt_bvh2_node root;
		std::memset(&root, 0, sizeof(root));
		empty_child(root, 0);
		empty_child(root, 1);
		root.child0 = root.child1 = make_leaf_link(0, 1);
		out.nodes.push_back(root);
		out.depth = 1;
		WideNode w;
		std::memset(&w, 0, sizeof(w));
		for (int a = 0; a < 3; a++) { w.e[a] = 1; for (int sl = 0; sl < 8; sl++) { w.qlo[a][sl] = 255; w.qhi[a][sl] = 0; } }
		wide.nodes.push_back(w);
		wide.depth = 1;
		return;
	}
	Builder b(opt);
	const uint64_t m = n + (cut ? cut->extra.size() : 0); // references
	b.prim_box.resize(m);
	b.centroid.resize(3 * m);
	b.prim_tri.resize(m);
	b.idx.resize(m);
	for (uint64_t t = 0; t < n; t++) {
		const double *v0, *v1, *v2;
		scene.tri_vertices(t, v0, v1, v2);
		Box bx;
		// the box of everything the reference's test of this fan triangle can accept: the vertices' box for a planar polygon,
		// more for a polygon that is not (accept_region.h)
		double rlo[3], rhi[3];
		accept_region_box(v0, v1, v2, &scene.face_normal[3 * (size_t)scene.tri_face[t]], rlo, rhi);
		for (int a = 0; a < 3; a++) {
			double mn = std::min(v0[a], std::min(v1[a], v2[a])), mx = std::max(v0[a], std::max(v1[a], v2[a]));
			bx.lo[a] = round_down(rlo[a]);
			bx.hi[a] = round_up(rhi[a]);
			b.centroid[3 * t + a] = (float)(0.5 * (mn + mx));
		}
		b.prim_box[t] = bx;
		b.prim_tri[t] = (uint32_t)t;
		b.idx[t] = (uint32_t)t;
	}
	if (cut) {
		// a triangle that was cut keeps its reference with the box of its first piece; the other pieces are references of their own
		auto place = [&](uint64_t prim, const SplitRef &r) {
			for (int a = 0; a < 3; a++) {
				b.prim_box[prim].lo[a] = r.lo[a];
				b.prim_box[prim].hi[a] = r.hi[a];
				b.centroid[3 * prim + a] = 0.5f * (r.lo[a] + r.hi[a]);
			}
			b.prim_tri[prim] = r.tri;
			b.idx[prim] = (uint32_t)prim;
		};
		for (const SplitRef &r : cut->first) place(r.tri, r);
		for (size_t k = 0; k < cut->extra.size(); k++) place(n + k, cut->extra[k]);
	}
	b.temp.resize(2 * m);
	int hw = opt.threads > 0 ? opt.threads : (int)std::thread::hardware_concurrency();
	b.max_threads = std::max(1, std::min(hw, 32));
	uint32_t root = b.alloc_node();
	b.build(root, 0, (uint32_t)m, 0);

	out.nodes.reserve(m);
	out.tri_index.reserve(m);
	Flattener fl{b, out};
	if (b.temp[root].left < 0) {
		// the whole scene fits one leaf: a root whose second child is an empty box
		rt_bvh2_node nd;
		std::memset(&nd, 0, sizeof(nd));
		const Box &bx = b.temp[root].box;
		for (int a = 0; a < 3; a++) { nd.lo0[a] = bx.lo[a]; nd.hi0[a] = bx.hi[a]; }
		empty_child(nd, 1);
		out.nodes.push_back(nd);
		int32_t leaf = fl.emit((int32_t)root, 0);
		out.nodes[0].child0 = leaf;
		out.nodes[0].child1 = leaf;
		out.depth = 1;
	} else {
		fl.emit((int32_t)root, 0);
	}
	wide.nodes.reserve(m / 2 + 16);
	wide.tri_index.reserve(m);
	wide.nodes.resize(1);
	WideCut wide_cut;
	if (opt.optimal_cut) wide_cut.solve(b, b.next_node.load(), opt.wide_cost_node, opt.wide_cost_tri);
	WideFlattener wf{b, wide, wide_cut};
	wf.emit((int32_t)root, 0, 1);
	out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
}

} // namespace mi355rt
