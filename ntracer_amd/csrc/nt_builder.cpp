// k-d tree construction on the host: what build_kdtree / build_composite_scene do in
// the reference (src/tracer.hpp:1965-2455, a multi-threaded C++ SAH builder), built a different way.
//
// The reference decides which side of a split a simplex belongs to with a set of projection tests
// (aabb::intersects, tracer.hpp:1465-1512).  Here every (simplex, cell) pair is CLIPPED exactly: the n-1-simplex
// is a convex polytope in its own affine space (barycentric coordinates); cutting it with the 2n half-spaces of
// the cell keeps a vertex list in which every vertex carries the set of constraints that are tight at it, so that
// the edges crossed by the next cut are the vertex pairs sharing n-2 tight constraints.  The bounding box of what
// is left is the primitive's true extent inside the cell ("perfect splits"): it feeds the SAH sweep, and it IS
// the membership test -- a primitive goes left iff its part inside the cell reaches below the split.  Long thin
// simplices that cross a cell diagonally (the cones of a star polytope's cells) therefore cost what they really
// cover, not their bounding box.  Solids are placed by bounding box.
//
// Conservative by construction: clipping is done in double with a tolerance that only ever keeps more; if a
// clipped polytope grows past NT_CLIP_MAX_VERTS vertices (high dimensions) the item falls back to its bounding
// box cut to the cell.  Nearest-hit results do not depend on the tree, so none of this is part of the parity
// contract; the kernels' early exit needs only that no primitive is missing from a cell it enters.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/ntracer_hip.h"

namespace {

constexpr int NT_CLIP_MAX_VERTS = 4096;
constexpr int KD_MAX_DIM = 64;
constexpr double TRAVERSAL_COST = 1.0;
constexpr double INTERSECTION_COST = 1.0;

struct ClipVert {
    std::vector<double> x;      // position, n coordinates
    uint64_t tight[3];          // constraints tight at this vertex: bits 0..n-1 = barycentric facets, then 2 per axis
};

inline int popcount3(const uint64_t *a, const uint64_t *b) {
    return __builtin_popcountll(a[0] & b[0]) + __builtin_popcountll(a[1] & b[1]) + __builtin_popcountll(a[2] & b[2]);
}
inline void setbit(uint64_t *m, int b) { m[b >> 6] |= 1ull << (b & 63); }

struct Builder {
    int n;
    int n_items;
    const float *item_lo, *item_hi;
    const int32_t *simplex_first;
    const float *simplex_verts;
    int max_depth, split_threshold;
    double scale;

    std::vector<int32_t> node_axis, node_left, node_right, leaf_items;
    std::vector<float> node_split;

    // exact extent of simplex s inside [lo,hi]; false when nothing (of positive measure along some axis) is left
    bool clip_simplex(int s, const double *lo, const double *hi, double *cmin, double *cmax, bool &overflow) const {
        const double eps = 1e-10 * scale;
        std::vector<ClipVert> cur(n), next;
        const float *v = simplex_verts + (size_t)s * n * n;
        for (int i = 0; i < n; ++i) {
            cur[i].x.assign(v + (size_t)i * n, v + (size_t)(i + 1) * n);
            cur[i].tight[0] = cur[i].tight[1] = cur[i].tight[2] = 0;
            for (int j = 0; j < n; ++j)
                if (j != i) setbit(cur[i].tight, j);          // lambda_j = 0 at vertex i
        }
        const int need = n - 2;                               // an edge of an (n-1)-polytope: n-2 shared tight constraints
        std::vector<double> dist;
        for (int a = 0; a < n && !cur.empty(); ++a) {
            for (int side = 0; side < 2; ++side) {
                const double bound = side == 0 ? lo[a] : hi[a];
                const double sgn = side == 0 ? 1.0 : -1.0;    // keep sgn*(x_a - bound) >= 0
                const int cbit = n + 2 * a + side;
                dist.resize(cur.size());
                bool any_out = false;
                for (size_t i = 0; i < cur.size(); ++i) {
                    dist[i] = sgn * (cur[i].x[a] - bound);
                    any_out = any_out || dist[i] < -eps;
                }
                if (!any_out) continue;
                next.clear();
                for (size_t i = 0; i < cur.size(); ++i)
                    if (dist[i] >= -eps) {
                        next.push_back(cur[i]);
                        if (std::fabs(dist[i]) <= eps) setbit(next.back().tight, cbit);
                    }
                for (size_t i = 0; i < cur.size(); ++i) {
                    if (dist[i] >= -eps) continue;
                    for (size_t j = 0; j < cur.size(); ++j) {
                        if (dist[j] <= eps) continue;         // strictly inside partners only
                        if (popcount3(cur[i].tight, cur[j].tight) < need) continue;
                        const double t = dist[j] / (dist[j] - dist[i]);
                        ClipVert nv;
                        nv.x.resize(n);
                        for (int k = 0; k < n; ++k) nv.x[k] = cur[j].x[k] + t * (cur[i].x[k] - cur[j].x[k]);
                        nv.x[a] = bound;
                        for (int k = 0; k < 3; ++k) nv.tight[k] = cur[i].tight[k] & cur[j].tight[k];
                        setbit(nv.tight, cbit);
                        next.push_back(std::move(nv));
                        if ((int)next.size() > NT_CLIP_MAX_VERTS) { overflow = true; return true; }
                    }
                }
                cur.swap(next);
                if (cur.empty()) break;
            }
        }
        if (cur.empty()) return false;
        for (int k = 0; k < n; ++k) { cmin[k] = std::numeric_limits<double>::infinity(); cmax[k] = -cmin[k]; }
        for (const ClipVert &c : cur)
            for (int k = 0; k < n; ++k) { cmin[k] = std::min(cmin[k], c.x[k]); cmax[k] = std::max(cmax[k], c.x[k]); }
        return true;
    }

    // extent of item `it` inside the cell; false: not in the cell
    bool item_bounds(int it, const double *lo, const double *hi, double *cmin, double *cmax) const {
        const float *blo = item_lo + (size_t)it * n, *bhi = item_hi + (size_t)it * n;
        for (int k = 0; k < n; ++k) {
            cmin[k] = std::max<double>(blo[k], lo[k]);
            cmax[k] = std::min<double>(bhi[k], hi[k]);
            if (cmin[k] > cmax[k]) return false;
        }
        const int s0 = simplex_first[it], s1 = simplex_first[it + 1];
        if (s0 == s1) return true;                            // solids: bounding box
        double amin[KD_MAX_DIM], amax[KD_MAX_DIM], smin[KD_MAX_DIM], smax[KD_MAX_DIM];
        bool any = false;
        for (int s = s0; s < s1; ++s) {
            bool overflow = false;
            if (!clip_simplex(s, lo, hi, smin, smax, overflow)) continue;
            if (overflow) return true;                        // keep the box estimate (conservative)
            for (int k = 0; k < n; ++k) {
                amin[k] = any ? std::min(amin[k], smin[k]) : smin[k];
                amax[k] = any ? std::max(amax[k], smax[k]) : smax[k];
            }
            any = true;
        }
        if (!any) return false;
        for (int k = 0; k < n; ++k) { cmin[k] = std::max(cmin[k], amin[k]); cmax[k] = std::min(cmax[k], amax[k]); }
        return true;
    }

    static double area(int n, const double *lo, const double *hi) {
        // surface measure of an n-box up to a constant: sum over axes of the product of the other extents
        double tot = 0.0;
        for (int a = 0; a < n; ++a) {
            double p = 1.0;
            for (int k = 0; k < n; ++k)
                if (k != a) p *= std::max(hi[k] - lo[k], 0.0);
            tot += p;
        }
        return tot;
    }

    int make_leaf(const std::vector<int> &items) {
        const int idx = (int)node_axis.size();
        node_axis.push_back(-1);
        node_split.push_back(0.0f);
        node_left.push_back((int32_t)leaf_items.size());
        node_right.push_back((int32_t)items.size());
        leaf_items.insert(leaf_items.end(), items.begin(), items.end());
        return idx;
    }

    // refs: items in this cell with their extents inside it (cmin/cmax: [ref][n])
    int build(std::vector<int> &refs, std::vector<double> &cmin, std::vector<double> &cmax, const double *lo, const double *hi, int depth) {
        const int m = (int)refs.size();
        if (m <= split_threshold || depth >= max_depth) return make_leaf(refs);
        const double base = area(n, lo, hi);
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1;
        double best_pos = 0.0;
        int best_nl = 0, best_nr = 0;
        if (base > 0.0) {
            std::vector<std::pair<double, int>> ev;            // (position, kind): 0 = end, 1 = flat, 2 = start
            for (int axis = 0; axis < n; ++axis) {
                if (!(hi[axis] > lo[axis])) continue;
                ev.clear();
                for (int r = 0; r < m; ++r) {
                    const double s = cmin[(size_t)r * n + axis], e = cmax[(size_t)r * n + axis];
                    if (s == e) ev.emplace_back(s, 1);
                    else { ev.emplace_back(s, 2); ev.emplace_back(e, 0); }
                }
                std::sort(ev.begin(), ev.end());
                // box area is linear in the extent along `axis`: A0 + A1*x
                double a0 = 1.0, a1 = 0.0;
                for (int k = 0; k < n; ++k)
                    if (k != axis) a0 *= std::max(hi[k] - lo[k], 0.0);
                for (int j = 0; j < n; ++j) {
                    if (j == axis) continue;
                    double p = 1.0;
                    for (int k = 0; k < n; ++k)
                        if (k != axis && k != j) p *= std::max(hi[k] - lo[k], 0.0);
                    a1 += p;
                }
                int nl = 0, nr = m;
                size_t i = 0;
                while (i < ev.size()) {
                    const double pos = ev[i].first;
                    int ends = 0, flats = 0, starts = 0;
                    while (i < ev.size() && ev[i].first == pos) {
                        if (ev[i].second == 0) ++ends;
                        else if (ev[i].second == 1) ++flats;
                        else ++starts;
                        ++i;
                    }
                    nr -= ends + flats;
                    const double ext = hi[axis] - lo[axis];
                    if (pos > lo[axis] + 1e-7 * scale && pos < hi[axis] - 1e-7 * scale) {
                        // flat primitives lying in the plane go to both sides (the walk picks a side by the ray)
                        const int cl = nl + flats, cr = nr + flats;
                        double cost = TRAVERSAL_COST + INTERSECTION_COST * ((a0 + a1 * (pos - lo[axis])) * cl + (a0 + a1 * (hi[axis] - pos)) * cr) / base;
                        // cutting off empty space is worth more -- if it is a real slab, not the rounding gap
                        // between the fp32 split of the parent and the primitives' extents
                        const bool empty_cut = cl == 0 || cr == 0;
                        const double emptied = cl == 0 ? pos - lo[axis] : hi[axis] - pos;
                        if (empty_cut && emptied < 0.01 * ext) cost = std::numeric_limits<double>::infinity();
                        else if (empty_cut) cost *= 0.8;
                        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_pos = pos; best_nl = cl; best_nr = cr; }
                    }
                    nl += starts + flats;
                }
            }
        }
        if (best_axis < 0 || best_cost >= INTERSECTION_COST * m || (best_nl == m && best_nr == m)) return make_leaf(refs);
        const float split = (float)best_pos;
        const double sp = (double)split;                       // the kernels compare against the fp32 value
        const double eps = 1e-10 * scale;
        std::vector<double> llo(lo, lo + n), lhi(hi, hi + n), rlo(lo, lo + n), rhi(hi, hi + n);
        lhi[best_axis] = sp;
        rlo[best_axis] = sp;
        std::vector<int> lrefs, rrefs;
        std::vector<double> lmin, lmax, rmin, rmax;
        double bmin[KD_MAX_DIM], bmax[KD_MAX_DIM];
        for (int r = 0; r < m; ++r) {
            const double s = cmin[(size_t)r * n + best_axis], e = cmax[(size_t)r * n + best_axis];
            const bool flat_in_plane = s == e && std::fabs(s - sp) <= eps;
            if (s < sp - eps || flat_in_plane) {
                if (e <= sp + eps && !flat_in_plane) {          // entirely left: extents unchanged
                    lrefs.push_back(refs[r]);
                    lmin.insert(lmin.end(), &cmin[(size_t)r * n], &cmin[(size_t)r * n] + n);
                    lmax.insert(lmax.end(), &cmax[(size_t)r * n], &cmax[(size_t)r * n] + n);
                } else if (item_bounds(refs[r], llo.data(), lhi.data(), bmin, bmax)) {
                    lrefs.push_back(refs[r]);
                    lmin.insert(lmin.end(), bmin, bmin + n);
                    lmax.insert(lmax.end(), bmax, bmax + n);
                }
            }
            if (e > sp + eps || flat_in_plane) {
                if (s >= sp - eps && !flat_in_plane) {
                    rrefs.push_back(refs[r]);
                    rmin.insert(rmin.end(), &cmin[(size_t)r * n], &cmin[(size_t)r * n] + n);
                    rmax.insert(rmax.end(), &cmax[(size_t)r * n], &cmax[(size_t)r * n] + n);
                } else if (item_bounds(refs[r], rlo.data(), rhi.data(), bmin, bmax)) {
                    rrefs.push_back(refs[r]);
                    rmin.insert(rmin.end(), bmin, bmin + n);
                    rmax.insert(rmax.end(), bmax, bmax + n);
                }
            }
        }
        if ((int)lrefs.size() == m && (int)rrefs.size() == m) return make_leaf(refs);
        if (lrefs.empty() && rrefs.empty()) return make_leaf(refs);
        // free this level's arrays before descending
        std::vector<int>().swap(refs);
        std::vector<double>().swap(cmin);
        std::vector<double>().swap(cmax);
        const int idx = (int)node_axis.size();
        node_axis.push_back(best_axis);
        node_split.push_back(split);
        node_left.push_back(-1);
        node_right.push_back(-1);
        const int l = lrefs.empty() ? -1 : build(lrefs, lmin, lmax, llo.data(), lhi.data(), depth + 1);
        const int r = rrefs.empty() ? -1 : build(rrefs, rmin, rmax, rlo.data(), rhi.data(), depth + 1);
        node_left[idx] = l;
        node_right[idx] = r;
        return idx;
    }
};

template <typename T>
T *dup(const std::vector<T> &v) {
    T *p = (T *)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

extern "C" {

int nt_kdtree_build(int dimension, int n_items, const float *item_lo, const float *item_hi, const int32_t *simplex_first,
                    const float *simplex_verts, int max_depth, int split_threshold, nt_kdtree *out) {
    if (!out) return NT_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (dimension < 1 || dimension > KD_MAX_DIM || n_items < 1 || !item_lo || !item_hi || !simplex_first) return NT_E_INVALID;
    if (simplex_first[n_items] > 0 && !simplex_verts) return NT_E_INVALID;
    if (dimension + 2 * dimension > 192) return NT_E_INVALID;
    Builder b;
    b.n = dimension;
    b.n_items = n_items;
    b.item_lo = item_lo;
    b.item_hi = item_hi;
    b.simplex_first = simplex_first;
    b.simplex_verts = simplex_verts;
    b.max_depth = max_depth > 0 ? max_depth : 25;
    b.split_threshold = split_threshold > 0 ? split_threshold : 2;
    const int n = dimension;
    std::vector<double> lo(n, std::numeric_limits<double>::infinity()), hi(n, -std::numeric_limits<double>::infinity());
    double scale = 1.0;
    for (int i = 0; i < n_items; ++i)
        for (int k = 0; k < n; ++k) {
            const double a = item_lo[(size_t)i * n + k], c = item_hi[(size_t)i * n + k];
            if (!(a <= c)) return NT_E_INVALID;
            lo[k] = std::min(lo[k], a);
            hi[k] = std::max(hi[k], c);
            scale = std::max(scale, std::max(std::fabs(a), std::fabs(c)));
        }
    b.scale = scale;
    try {
        std::vector<int> refs;
        std::vector<double> cmin, cmax;
        double bmin[KD_MAX_DIM], bmax[KD_MAX_DIM];
        for (int i = 0; i < n_items; ++i) {
            if (!b.item_bounds(i, lo.data(), hi.data(), bmin, bmax)) {
                // degenerate input (e.g. a zero-volume simplex): keep it by its box
                for (int k = 0; k < n; ++k) { bmin[k] = item_lo[(size_t)i * n + k]; bmax[k] = item_hi[(size_t)i * n + k]; }
            }
            refs.push_back(i);
            cmin.insert(cmin.end(), bmin, bmin + n);
            cmax.insert(cmax.end(), bmax, bmax + n);
        }
        out->root = b.build(refs, cmin, cmax, lo.data(), hi.data(), 0);
    } catch (const std::bad_alloc &) {
        return NT_E_NOMEM;
    }
    out->n_nodes = (int32_t)b.node_axis.size();
    out->n_leaf_items = (int32_t)b.leaf_items.size();
    out->node_axis = dup(b.node_axis);
    out->node_split = dup(b.node_split);
    out->node_left = dup(b.node_left);
    out->node_right = dup(b.node_right);
    out->leaf_items = dup(b.leaf_items);
    out->aabb = (float *)std::malloc(sizeof(float) * 2 * n);
    if (!out->node_axis || !out->node_split || !out->node_left || !out->node_right || !out->leaf_items || !out->aabb) {
        nt_kdtree_free(out);
        return NT_E_NOMEM;
    }
    for (int k = 0; k < n; ++k) { out->aabb[k] = (float)lo[k]; out->aabb[n + k] = (float)hi[k]; }
    return NT_OK;
}

void nt_kdtree_free(nt_kdtree *t) {
    if (!t) return;
    std::free(t->node_axis);
    std::free(t->node_split);
    std::free(t->node_left);
    std::free(t->node_right);
    std::free(t->leaf_items);
    std::free(t->aabb);
    std::memset(t, 0, sizeof(*t));
}

}  // extern "C"
