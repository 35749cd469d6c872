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
#include <atomic>
#include <system_error>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/ntracer_hip.h"

namespace {

constexpr int NT_CLIP_MAX_VERTS = 4096;
constexpr int KD_MAX_DIM = 64;
constexpr int KD_CLIP_DIM = 16;        // exact clipping up to this dimension; beyond it items are placed by bounding box

inline int popcount3(const uint64_t *a, const uint64_t *b) {
    return __builtin_popcountll(a[0] & b[0]) + __builtin_popcountll(a[1] & b[1]) + __builtin_popcountll(a[2] & b[2]);
}
inline void setbit(uint64_t *m, int b) { m[b >> 6] |= 1ull << (b & 63); }

struct Out {                    // a (sub)tree in the flat layout
    std::vector<int32_t> node_axis, node_left, node_right, leaf_items;
    std::vector<float> node_split;
    // append `src` (a complete subtree); returns the new index of its node `root`
    int append(const Out &src, int root) {
        const int nb = (int)node_axis.size(), ib = (int)leaf_items.size();
        for (size_t k = 0; k < src.node_axis.size(); ++k) {
            const bool leaf = src.node_axis[k] < 0;
            node_axis.push_back(src.node_axis[k]);
            node_split.push_back(src.node_split[k]);
            node_left.push_back(leaf ? src.node_left[k] + ib : (src.node_left[k] >= 0 ? src.node_left[k] + nb : -1));
            node_right.push_back(leaf ? src.node_right[k] : (src.node_right[k] >= 0 ? src.node_right[k] + nb : -1));
        }
        leaf_items.insert(leaf_items.end(), src.leaf_items.begin(), src.leaf_items.end());
        return root + nb;
    }
};

// The references of one cell: item ids, each item's extent inside the cell, and -- for simplices -- what is left
// of each simplex inside the cell, as a vertex list (positions + the constraints tight at each vertex).  A child
// cell is the parent cut by ONE more plane, so the polytopes are cut incrementally on the way down.
struct Refs {
    int n = 0;
    std::vector<int> item;
    std::vector<double> cmin, cmax;          // [ref][n]
    std::vector<unsigned char> boxed;        // 1: placed by bounding box (solid, clip overflow, n > KD_CLIP_DIM)
    std::vector<int> slot_first;             // [ref+1]: first polytope slot of the ref (one slot per simplex)
    std::vector<int> vert_first;             // [slot+1]: first vertex of the polytope (empty polytope: no vertices)
    std::vector<double> vx;                  // [vertex][n]
    std::vector<uint64_t> vt;                // [vertex][3]

    int size() const { return (int)item.size(); }
    void begin(int dim) { n = dim; slot_first.assign(1, 0); vert_first.assign(1, 0); }
    void release() { *this = Refs(); }
    // append all of `o` (same n)
    void append(const Refs &o) {
        const int sb = vert_first.empty() ? 0 : (int)vert_first.size() - 1, vb = (int)(vx.size() / std::max(n, 1));
        item.insert(item.end(), o.item.begin(), o.item.end());
        cmin.insert(cmin.end(), o.cmin.begin(), o.cmin.end());
        cmax.insert(cmax.end(), o.cmax.begin(), o.cmax.end());
        boxed.insert(boxed.end(), o.boxed.begin(), o.boxed.end());
        for (size_t k = 1; k < o.slot_first.size(); ++k) slot_first.push_back(o.slot_first[k] + sb);
        for (size_t k = 1; k < o.vert_first.size(); ++k) vert_first.push_back(o.vert_first[k] + vb);
        vx.insert(vx.end(), o.vx.begin(), o.vx.end());
        vt.insert(vt.end(), o.vt.begin(), o.vt.end());
    }
};

struct Builder {
    int n;
    int n_items;
    const float *item_lo, *item_hi;
    const int32_t *simplex_first;
    const float *simplex_verts;
    int max_depth, split_threshold;
    double TRAVERSAL_COST, INTERSECTION_COST;
    double scale;
    int threads = 1;

    Out top;
    std::atomic<int> spare_threads{0};

    // Cut the polytope (nv vertices at x/t) with the half-space sgn*(x_a - bound) >= 0, append what is left to `dst`
    // as a new slot.  Returns false if the result outgrew NT_CLIP_MAX_VERTS (nothing appended).
    bool cut(const double *x, const uint64_t *t, int nv, int a, double sgn, double bound, int cbit, Refs &dst) const {
        return cut_general(x, t, nv, a, sgn, bound, cbit, n - 2, dst);   // an edge of an (n-1)-polytope: n-2 shared tight constraints
    }

    bool cut_general(const double *x, const uint64_t *t, int nv, int a, double sgn, double bound, int cbit, int need, Refs &dst) const {
        const double eps = 1e-10 * scale;
        static thread_local std::vector<double> dist;
        dist.resize(nv);
        bool any_out = false;
        for (int i = 0; i < nv; ++i) {
            dist[i] = sgn * (x[(size_t)i * n + a] - bound);
            any_out = any_out || dist[i] < -eps;
        }
        const size_t v0 = dst.vx.size() / n;
        for (int i = 0; i < nv; ++i)
            if (dist[i] >= -eps) {
                dst.vx.insert(dst.vx.end(), x + (size_t)i * n, x + (size_t)(i + 1) * n);
                dst.vt.insert(dst.vt.end(), t + (size_t)i * 3, t + (size_t)(i + 1) * 3);
                if (any_out && std::fabs(dist[i]) <= eps) setbit(&dst.vt[dst.vt.size() - 3], cbit);
            }
        if (any_out) {
            for (int i = 0; i < nv; ++i) {
                if (dist[i] >= -eps) continue;
                for (int j = 0; j < nv; ++j) {
                    if (dist[j] <= eps) continue;             // strictly inside partners only
                    if (popcount3(t + (size_t)i * 3, t + (size_t)j * 3) < need) continue;
                    const double w = dist[j] / (dist[j] - dist[i]);
                    const size_t at = dst.vx.size();
                    dst.vx.resize(at + n);
                    for (int k = 0; k < n; ++k) dst.vx[at + k] = x[(size_t)j * n + k] + w * (x[(size_t)i * n + k] - x[(size_t)j * n + k]);
                    dst.vx[at + a] = bound;
                    const size_t tt = dst.vt.size();
                    dst.vt.resize(tt + 3);
                    for (int k = 0; k < 3; ++k) dst.vt[tt + k] = t[(size_t)i * 3 + k] & t[(size_t)j * 3 + k];
                    setbit(&dst.vt[tt], cbit);
                    if ((int)(dst.vx.size() / n - v0) > NT_CLIP_MAX_VERTS) {
                        dst.vx.resize(v0 * n);
                        dst.vt.resize(v0 * 3);
                        return false;
                    }
                }
            }
        }
        dst.vert_first.push_back((int)(dst.vx.size() / n));
        return true;
    }

    // reference r of `src` restricted to the half-space; appended to `dst` if anything is left
    void restrict_ref(const Refs &src, int r, int a, int side, double bound, const double *clo, const double *chi, Refs &dst) const {
        const int it = src.item[r];
        const double sgn = side == 0 ? 1.0 : -1.0;
        double bmin[KD_MAX_DIM], bmax[KD_MAX_DIM];
        bool boxed = src.boxed[r] != 0;
        const size_t vmark = dst.vx.size(), tmark = dst.vt.size(), smark = dst.vert_first.size();
        bool any = false;
        if (!boxed) {
            for (int sl = src.slot_first[r]; sl < src.slot_first[r + 1]; ++sl) {
                const int f = src.vert_first[sl], nv = src.vert_first[sl + 1] - f;
                const size_t before = dst.vx.size() / n;
                if (nv == 0) { dst.vert_first.push_back((int)before); continue; }
                if (!cut(&src.vx[(size_t)f * n], &src.vt[(size_t)f * 3], nv, a, sgn, bound, n + 2 * a + side, dst)) { boxed = true; break; }
                for (size_t v = before; v < dst.vx.size() / n; ++v)
                    for (int k = 0; k < n; ++k) {
                        const double c = dst.vx[v * n + k];
                        bmin[k] = any ? std::min(bmin[k], c) : c;
                        bmax[k] = any ? std::max(bmax[k], c) : c;
                        if (k == n - 1) any = true;
                    }
            }
        }
        if (boxed) {
            // bounding box of the item cut to the child cell (conservative)
            dst.vx.resize(vmark);
            dst.vt.resize(tmark);
            dst.vert_first.resize(smark);
            for (int k = 0; k < n; ++k) {
                bmin[k] = std::max<double>(item_lo[(size_t)it * n + k], clo[k]);
                bmax[k] = std::min<double>(item_hi[(size_t)it * n + k], chi[k]);
                if (bmin[k] > bmax[k]) return;
            }
            for (int sl = src.slot_first[r]; sl < src.slot_first[r + 1]; ++sl) dst.vert_first.push_back((int)(dst.vx.size() / n));
        } else if (!any) {
            dst.vx.resize(vmark);
            dst.vt.resize(tmark);
            dst.vert_first.resize(smark);
            return;
        }
        dst.item.push_back(it);
        dst.boxed.push_back(boxed ? 1 : 0);
        dst.cmin.insert(dst.cmin.end(), bmin, bmin + n);
        dst.cmax.insert(dst.cmax.end(), bmax, bmax + n);
        dst.slot_first.push_back((int)dst.vert_first.size() - 1);
    }

    // reference r of `src` unchanged
    void copy_ref(const Refs &src, int r, Refs &dst) const {
        dst.item.push_back(src.item[r]);
        dst.boxed.push_back(src.boxed[r]);
        dst.cmin.insert(dst.cmin.end(), &src.cmin[(size_t)r * n], &src.cmin[(size_t)r * n] + n);
        dst.cmax.insert(dst.cmax.end(), &src.cmax[(size_t)r * n], &src.cmax[(size_t)r * n] + n);
        const int vb = (int)(dst.vx.size() / n);
        for (int sl = src.slot_first[r]; sl < src.slot_first[r + 1]; ++sl) {
            const int f = src.vert_first[sl], e = src.vert_first[sl + 1];
            dst.vx.insert(dst.vx.end(), &src.vx[(size_t)f * n], &src.vx[(size_t)f * n] + (size_t)(e - f) * n);
            dst.vt.insert(dst.vt.end(), &src.vt[(size_t)f * 3], &src.vt[(size_t)f * 3] + (size_t)(e - f) * 3);
            dst.vert_first.push_back((int)(dst.vx.size() / n));
        }
        (void)vb;
        dst.slot_first.push_back((int)dst.vert_first.size() - 1);
    }

    void root_refs(Refs &R) const {
        R.begin(n);
        for (int it = 0; it < n_items; ++it) {
            const int s0 = simplex_first[it], s1 = simplex_first[it + 1];
            const bool boxed = s0 == s1 || n > KD_CLIP_DIM;
            R.item.push_back(it);
            R.boxed.push_back(boxed ? 1 : 0);
            for (int k = 0; k < n; ++k) { R.cmin.push_back(item_lo[(size_t)it * n + k]); R.cmax.push_back(item_hi[(size_t)it * n + k]); }
            for (int s = s0; s < s1; ++s) {
                if (!boxed) {
                    const float *v = simplex_verts + (size_t)s * n * n;
                    for (int i = 0; i < n; ++i) {
                        for (int k = 0; k < n; ++k) R.vx.push_back(v[(size_t)i * n + k]);
                        uint64_t m[3] = {0, 0, 0};
                        for (int j = 0; j < n; ++j)
                            if (j != i) setbit(m, j);          // lambda_j = 0 at vertex i
                        R.vt.insert(R.vt.end(), m, m + 3);
                    }
                }
                R.vert_first.push_back((int)(R.vx.size() / n));
            }
            R.slot_first.push_back((int)R.vert_first.size() - 1);
        }
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

    static int make_leaf(Out &o, const std::vector<int> &items) {
        const int idx = (int)o.node_axis.size();
        o.node_axis.push_back(-1);
        o.node_split.push_back(0.0f);
        o.node_left.push_back((int32_t)o.leaf_items.size());
        o.node_right.push_back((int32_t)items.size());
        o.leaf_items.insert(o.leaf_items.end(), items.begin(), items.end());
        return idx;
    }

    int build(Out &o, Refs &R, const double *lo, const double *hi, int depth) {
        const int m = R.size();
        if (m <= split_threshold || depth >= max_depth) return make_leaf(o, R.item);
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
                    const double s = R.cmin[(size_t)r * n + axis], e = R.cmax[(size_t)r * n + axis];
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
                const double ext = hi[axis] - lo[axis];
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
        if (best_axis < 0 || best_cost >= INTERSECTION_COST * m || (best_nl == m && best_nr == m)) return make_leaf(o, R.item);
        const float split = (float)best_pos;
        const double sp = (double)split;                       // the kernels compare against the fp32 value
        const double eps = 1e-10 * scale;
        std::vector<double> llo(lo, lo + n), lhi(hi, hi + n), rlo(lo, lo + n), rhi(hi, hi + n);
        lhi[best_axis] = sp;
        rlo[best_axis] = sp;
        // classify every reference; straddlers are cut by the split plane (big nodes: in parallel, slice by slice)
        auto classify = [&](int r0, int r1, Refs &L, Refs &Rr) {
            L.begin(n);
            Rr.begin(n);
            for (int r = r0; r < r1; ++r) {
                const double s = R.cmin[(size_t)r * n + best_axis], e = R.cmax[(size_t)r * n + best_axis];
                const bool flat_in_plane = s == e && std::fabs(s - sp) <= eps;
                if (s < sp - eps || flat_in_plane) {
                    if (e <= sp + eps) copy_ref(R, r, L);                 // entirely left (or flat in the plane)
                    else restrict_ref(R, r, best_axis, 1, sp, llo.data(), lhi.data(), L);
                }
                if (e > sp + eps || flat_in_plane) {
                    if (s >= sp - eps) copy_ref(R, r, Rr);
                    else restrict_ref(R, r, best_axis, 0, sp, rlo.data(), rhi.data(), Rr);
                }
            }
        };
        Refs L, Rr;
        int slices = 1;
        if (m >= 4096) {
            int avail = spare_threads.load();
            const int want = std::min(avail, m / 1024);
            while (want > 0 && avail >= want && !spare_threads.compare_exchange_weak(avail, avail - want)) {}
            if (want > 0 && avail >= want) slices = want + 1;
        }
        if (slices > 1) {
            std::vector<Refs> ls(slices), rs(slices);
            std::vector<std::thread> pool;
            std::atomic<bool> failed{false};
            auto guarded = [&](int t) {
                try {
                    classify((int)((long long)m * t / slices), (int)((long long)m * (t + 1) / slices), ls[t], rs[t]);
                } catch (...) {
                    failed = true;
                }
            };
            try {
                for (int t = 1; t < slices; ++t) pool.emplace_back(guarded, t);
            } catch (...) {
                failed = true;                                  // could not start every thread
            }
            const int started = (int)pool.size() + 1;
            guarded(0);
            for (auto &th : pool) th.join();
            spare_threads.fetch_add(slices - 1);
            if (failed || started != slices) throw std::bad_alloc();
            L.begin(n);
            Rr.begin(n);
            for (int t = 0; t < slices; ++t) { L.append(ls[t]); ls[t].release(); Rr.append(rs[t]); rs[t].release(); }
        } else {
            classify(0, m, L, Rr);
        }
        if (L.size() == m && Rr.size() == m) return make_leaf(o, R.item);
        if (L.size() == 0 && Rr.size() == 0) return make_leaf(o, R.item);
        R.release();                                            // free this level's arrays before descending
        const int idx = (int)o.node_axis.size();
        o.node_axis.push_back(best_axis);
        o.node_split.push_back(split);
        o.node_left.push_back(-1);
        o.node_right.push_back(-1);
        int l = -1, r = -1;
        // big subtrees are built side by side: the left one on a new thread into its own arrays, merged afterwards
        bool forked = false;
        if (std::min(L.size(), Rr.size()) >= 512) {
            int avail = spare_threads.load();
            while (avail > 0 && !spare_threads.compare_exchange_weak(avail, avail - 1)) {}
            forked = avail > 0;
        }
        if (forked) {
            Out lo_out;
            int lroot = -1;
            bool failed = false;
            std::thread th([&] {
                try { lroot = build(lo_out, L, llo.data(), lhi.data(), depth + 1); } catch (...) { failed = true; }
            });
            try {
                r = build(o, Rr, rlo.data(), rhi.data(), depth + 1);
            } catch (...) {
                th.join();
                spare_threads.fetch_add(1);
                throw;
            }
            th.join();
            spare_threads.fetch_add(1);
            if (failed) throw std::bad_alloc();
            l = o.append(lo_out, lroot);
        } else {
            l = L.size() ? build(o, L, llo.data(), lhi.data(), depth + 1) : -1;
            r = Rr.size() ? build(o, Rr, rlo.data(), rhi.data(), depth + 1) : -1;
        }
        o.node_left[idx] = l;
        o.node_right[idx] = r;
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
                    const float *simplex_verts, const nt_kdtree_params *params, nt_kdtree *out) {
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
    b.max_depth = params && params->max_depth > 0 ? params->max_depth : 25;
    b.split_threshold = params && params->split_threshold > 0 ? params->split_threshold : 2;
    b.TRAVERSAL_COST = params && params->traversal_cost > 0.0f ? params->traversal_cost : 1.0;
    b.INTERSECTION_COST = params && params->intersection_cost > 0.0f ? params->intersection_cost : 1.0;
    if (b.max_depth > 64) b.max_depth = 64;
    b.threads = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (const char *e = getenv("NTRACER_BUILD_THREADS")) b.threads = std::max(1, atoi(e));
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
        Refs R;
        b.root_refs(R);
        b.spare_threads = b.threads - 1;
        out->root = b.build(b.top, R, lo.data(), hi.data(), 0);
    } catch (const std::bad_alloc &) {
        return NT_E_NOMEM;
    } catch (const std::system_error &) {
        return NT_E_NOMEM;          // could not start a thread
    }
    out->n_nodes = (int32_t)b.top.node_axis.size();
    out->n_leaf_items = (int32_t)b.top.leaf_items.size();
    out->node_axis = dup(b.top.node_axis);
    out->node_split = dup(b.top.node_split);
    out->node_left = dup(b.top.node_left);
    out->node_right = dup(b.top.node_right);
    out->leaf_items = dup(b.top.leaf_items);
    out->aabb = (float *)std::malloc(sizeof(float) * 2 * n);
    if (!out->node_axis || !out->node_split || !out->node_left || !out->node_right || !out->leaf_items || !out->aabb) {
        nt_kdtree_free(out);
        return NT_E_NOMEM;
    }
    for (int k = 0; k < n; ++k) { out->aabb[k] = (float)lo[k]; out->aabb[n + k] = (float)hi[k]; }
    return NT_OK;
}

int nt_polytope_clip_box(int dimension, int n_verts, const float *verts, const uint64_t *tight, int shared_for_edge, int first_free_bit,
                         const float *lo, const float *hi, float *out_lo, float *out_hi) {
    if (dimension < 1 || dimension > KD_CLIP_DIM || n_verts < 1 || !verts || !tight || !lo || !hi) return NT_E_INVALID;
    if (first_free_bit < 0 || first_free_bit + 2 * dimension > 192 || shared_for_edge < 0) return NT_E_INVALID;
    const int n = dimension;
    Builder b;
    b.n = n;
    double scale = 1.0;
    for (int i = 0; i < n_verts * n; ++i) scale = std::max(scale, (double)std::fabs(verts[i]));
    for (int k = 0; k < n; ++k) scale = std::max(scale, (double)std::max(std::fabs(lo[k]), std::fabs(hi[k])));
    b.scale = scale;
    try {
        Refs cur, next;
        cur.begin(n);
        for (int i = 0; i < n_verts; ++i) {
            for (int k = 0; k < n; ++k) cur.vx.push_back(verts[(size_t)i * n + k]);
            cur.vt.insert(cur.vt.end(), tight + (size_t)i * 3, tight + (size_t)i * 3 + 3);
        }
        cur.vert_first.push_back(n_verts);
        int nv = n_verts;
        for (int a = 0; a < n && nv > 0; ++a)
            for (int side = 0; side < 2 && nv > 0; ++side) {
                next.begin(n);
                next.vx.clear();
                next.vt.clear();
                if (!b.cut_general(cur.vx.data(), cur.vt.data(), nv, a, side == 0 ? 1.0 : -1.0, side == 0 ? lo[a] : hi[a],
                                   first_free_bit + 2 * a + side, shared_for_edge, next))
                    return NT_E_NOMEM;                                      // outgrew the vertex budget
                nv = (int)(next.vx.size() / n);
                cur.vx.swap(next.vx);
                cur.vt.swap(next.vt);
            }
        if (nv > 0 && out_lo && out_hi)
            for (int k = 0; k < n; ++k) {
                double mn = cur.vx[k], mx = cur.vx[k];
                for (int v = 1; v < nv; ++v) { mn = std::min(mn, cur.vx[(size_t)v * n + k]); mx = std::max(mx, cur.vx[(size_t)v * n + k]); }
                out_lo[k] = (float)mn;
                out_hi[k] = (float)mx;
            }
        return nv;
    } catch (const std::bad_alloc &) {
        return NT_E_NOMEM;
    }
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
