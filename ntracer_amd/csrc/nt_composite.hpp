// nt_composite.hpp -- CompositeScene kernels for compile-time N: composite_scene::calculate_color
// (src/tracer.hpp:1885-1890) = aabb_distance -> k-d closest-hit walk -> leaf / simplex / solid tests -> base_color,
// with process_pixel fused into the epilogue.  Instantiated per N by nt_inst_composite.hip; the run-time-n kernel
// (nt_var.hip) shares the traversal helpers.
#pragma once
#include "nt_pixel.hpp"

namespace {

// CompositeScene, compile-time N
// --------------------------------------------------------------------------------------
struct Hit {
    float dist;
    int item;    // (index<<2)|kind, -1: none
    int lane;    // simplex inside a batch, -1 otherwise
};

struct Stats {
    unsigned int rays, shadow_rays, branches, leaves, simplex_tests, solid_tests, hits, aabb_enter;
};

template <int N>
__device__ __forceinline__ float dotN(const float (&a)[N], const float (&b)[N]) {
    float s = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < N; ++k) s = s + a[k] * b[k];
    return s;
}

template <int N>
__device__ __forceinline__ float dotP(const float *__restrict__ a, const float (&b)[N]) {
    float s = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < N; ++k) s = s + a[k] * b[k];
    return s;
}

// Load one simplex record (d, face_normal[N], p1[N], edge_normal[N-1][N]) with 16-byte loads.
template <int N>
struct SimplexRec {
    static constexpr int LEN = N * N + N + 1;
    static constexpr int LEN4 = (LEN + 3) / 4;
    float v[LEN4 * 4];
    __device__ __forceinline__ void load(const float *__restrict__ p) {
        const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
        for (int k = 0; k < LEN4; ++k) {
            const float4 t = q[k];
            v[4 * k] = t.x;
            v[4 * k + 1] = t.y;
            v[4 * k + 2] = t.z;
            v[4 * k + 3] = t.w;
        }
    }
    __device__ __forceinline__ float d() const { return v[0]; }
    __device__ __forceinline__ float nrm(int k) const { return v[1 + k]; }
    __device__ __forceinline__ float p1(int k) const { return v[1 + N + k]; }
    __device__ __forceinline__ float edge(int i, int k) const { return v[1 + 2 * N + i * N + k]; }
};

// triangle_batch::intersects, one SIMD lane (tracer.hpp:561-581): returns t, or 0 when masked out
template <int N>
__device__ __forceinline__ float simplex_batch_form(const SimplexRec<N> &s, const float (&o)[N], const float (&d)[N]) {
    float denom = s.nrm(0) * d[0];
#pragma unroll
    for (int k = 1; k < N; ++k) denom = denom + s.nrm(k) * d[k];
    float no = s.nrm(0) * o[0];
#pragma unroll
    for (int k = 1; k < N; ++k) no = no + s.nrm(k) * o[k];
    const float t = -(no + s.d()) / denom;
    bool ok = denom != 0.0f && t >= 0.0f;
    float pside[N];
#pragma unroll
    for (int k = 0; k < N; ++k) pside[k] = s.p1(k) - (o[k] + t * d[k]);
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        float area = s.edge(i, 0) * pside[0];
#pragma unroll
        for (int k = 1; k < N; ++k) area = area + s.edge(i, k) * pside[k];
        ok = ok && area >= -NT_FUZZ;
        tot += area;
    }
    ok = ok && tot <= (1.0f + NT_FUZZ);
    return ok ? t : 0.0f;
}

// The same predicate in two stages, for wave-uniform records: stage 1 is the plane part (denom, t); the
// point-in-simplex part only runs if some lane can still accept this simplex (`want`: the caller's remaining
// conditions, e.g. t < current cutoff).  A lane that fails stage 1 or `want` is rejected either way, so the
// result is the one simplex_batch_form gives.
template <int N, typename Want>
__device__ __forceinline__ float simplex_batch_form_staged(const SimplexRec<N> &s, const float (&o)[N], const float (&d)[N], Want want) {
    float denom = s.nrm(0) * d[0];
#pragma unroll
    for (int k = 1; k < N; ++k) denom = denom + s.nrm(k) * d[k];
    float no = s.nrm(0) * o[0];
#pragma unroll
    for (int k = 1; k < N; ++k) no = no + s.nrm(k) * o[k];
    const float t = -(no + s.d()) / denom;
    bool ok = denom != 0.0f && t >= 0.0f;
    if (__builtin_amdgcn_ballot_w64(ok && want(t)) == 0ull) return 0.0f;
    float pside[N];
#pragma unroll
    for (int k = 0; k < N; ++k) pside[k] = s.p1(k) - (o[k] + t * d[k]);
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        float area = s.edge(i, 0) * pside[0];
#pragma unroll
        for (int k = 1; k < N; ++k) area = area + s.edge(i, k) * pside[k];
        ok = ok && area >= -NT_FUZZ;
        tot += area;
    }
    ok = ok && tot <= (1.0f + NT_FUZZ);
    return ok ? t : 0.0f;
}

// triangle::intersects (tracer.hpp:411-440): scalar form with the early rejects
template <int N>
__device__ __forceinline__ float simplex_scalar_form(const SimplexRec<N> &s, const float (&o)[N], const float (&d)[N], float cutoff) {
    float denom = s.nrm(0) * d[0];
#pragma unroll
    for (int k = 1; k < N; ++k) denom = denom + s.nrm(k) * d[k];
    if (denom == 0.0f) return 0.0f;
    float no = s.nrm(0) * o[0];
#pragma unroll
    for (int k = 1; k < N; ++k) no = no + s.nrm(k) * o[k];
    const float t = -(no + s.d()) / denom;
    if (t <= 0.0f || t >= cutoff) return 0.0f;
    float pside[N];
#pragma unroll
    for (int k = 0; k < N; ++k) pside[k] = s.p1(k) - (o[k] + t * d[k]);
    float tot = 0.0f;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        float area = s.edge(i, 0) * pside[0];
#pragma unroll
        for (int k = 1; k < N; ++k) area = area + s.edge(i, k) * pside[k];
        ok = ok && !(area < -NT_FUZZ || area > (1.0f + NT_FUZZ));
        tot += area;
    }
    return (ok && tot <= (1.0f + NT_FUZZ)) ? t : 0.0f;
}

// hypercube_intersects for a solid's local ray (tracer.hpp:126-152); outputs the local normal ray
template <int N>
__device__ __forceinline__ float cube_local(const float (&o)[N], const float (&d)[N], float cutoff, float (&no)[N], float (&nd)[N]) {
    bool done = false;
    float result = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float di = d[i];
        const float s = di < 0.0f ? 1.0f : -1.0f;
        const float dist = (s - o[i]) / di;
        bool ok = !done && di != 0.0f && dist > 0.0f;
        float p[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            p[j] = d[j] * dist + o[j];
            if (j != i) ok = ok && !(fabsf(p[j]) > (1.0f + NT_FUZZ));
        }
        if (ok) {
            done = true;
            if (!(dist >= cutoff)) {
                result = dist;
#pragma unroll
                for (int j = 0; j < N; ++j) { no[j] = j == i ? s : p[j]; nd[j] = j == i ? s : 0.0f; }
            }
        }
    }
    return result;
}

// hypersphere_intersects (tracer.hpp:154-173)
template <int N>
__device__ __forceinline__ float sphere_local(const float (&o)[N], const float (&d)[N], float cutoff, float (&no)[N], float (&nd)[N]) {
    const float a = dotN<N>(d, d);
    const float b = 2.0f * dotN<N>(d, o);
    const float c = dotN<N>(o, o) - 1.0f;
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return 0.0f;
    const float dist = (-b - sqrtf(disc)) / (2.0f * a);
    if (dist <= 0.0f || dist >= cutoff) return 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) { no[j] = o[j] + d[j] * dist; nd[j] = no[j]; }
    return dist;
}

// solid::intersects (tracer.hpp:251-276).  When `want_normal`, the world-space normal ray is produced.
template <int N>
__device__ __noinline__ float solid_intersects(const NtCompositeDev &sc, int idx, const float (&o)[N], const float (&d)[N], float cutoff,
                                               bool want_normal, float (&no)[N], float (&nd)[N]) {
    const float *orient = sc.solid_recs + (size_t)idx * (2 * N * N + N);
    const float *inv = orient + N * N;
    const float *pos = inv + N * N;
    float lo[N], ld[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        lo[i] = dotP<N>(inv + i * N, o) - pos[i];
        ld[i] = dotP<N>(inv + i * N, d);
    }
    float ln_o[N], ln_d[N];
    float dist;
    if (sc.solid_types[idx] == 1) dist = cube_local<N>(lo, ld, cutoff, ln_o, ln_d);
    else dist = sphere_local<N>(lo, ld, cutoff, ln_o, ln_d);
    if (dist == 0.0f) return 0.0f;
    if (want_normal) {
        float tmp[N];
#pragma unroll
        for (int i = 0; i < N; ++i) tmp[i] = ln_o[i] + pos[i];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            no[i] = dotP<N>(orient + i * N, tmp);
            nd[i] = dotP<N>(orient + i * N, ln_d);
        }
    }
    return dist;
}

// Per-wave LDS scratch, all lane-major so that an access is conflict free whatever level / slot each lane is at:
//   stack[level][64]  4-byte entries: the index of a branch whose far side is still pending
//                     (t and the far child are recomputed from the node record on pop -- same inputs, same bits)
//   ray[N][64]        (origin[axis], 1/direction[axis]) pairs for the axis-indexed branch step
//   mbox[NT_MBOX][64] direct-mapped mailbox of recently tested leaf items (the reference's `checked`
//                     list, tracer.hpp:782,832: a primitive spanning several leaves is tested once)
#define NT_MBOX 16
struct WaveLds {
    int *stack;        // stack[level*64 + lane]
    float2 *ray;       // ray[axis*64 + lane]
    int *mbox;         // mbox[slot*64 + lane]
};

__device__ __forceinline__ size_t wave_lds_bytes(int stack_depth, int n) {
    return (size_t)64 * ((size_t)stack_depth * 4 + (size_t)n * 8 + (size_t)NT_MBOX * 4);
}

__device__ __forceinline__ WaveLds wave_lds(char *base, int wave, int stack_depth, int n) {
    char *p = base + (size_t)wave * wave_lds_bytes(stack_depth, n);
    WaveLds w;
    w.ray = reinterpret_cast<float2 *>(p);                                     // 8-byte aligned first
    w.stack = reinterpret_cast<int *>(p + (size_t)64 * n * 8);
    w.mbox = w.stack + (size_t)64 * stack_depth;
    return w;
}

__device__ __forceinline__ void mbox_reset(const WaveLds &w, int lane) {
#pragma unroll
    for (int k = 0; k < NT_MBOX; ++k) w.mbox[k * 64 + lane] = -1;
}

// true when `item` was already tested for this ray (then re-testing cannot change the hit: the cutoff only
// shrinks); otherwise records it.  Evictions only cause harmless re-tests.
__device__ __forceinline__ bool mbox_seen(const WaveLds &w, int lane, int item) {
    const int slot = ((item >> 2) & (NT_MBOX - 1)) * 64 + lane;
    const bool seen = w.mbox[slot] == item;
    w.mbox[slot] = item;
    return seen;
}

template <int N>
__device__ __forceinline__ void setup_ray_table(const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // invdir = 1/direction (tracer.hpp:1174); a NaN marks direction == 0 exactly so the
        // `if(target.direction[axis])` test (tracer.hpp:1191) needs no third table column
        const float inv = d[k] != 0.0f ? 1.0f / d[k] : __int_as_float(0x7fc00000);
        w.ray[k * 64 + lane] = make_float2(o[k], inv);
    }
}

// One leaf (kd_leaf<Store,true>::intersects, tracer.hpp:977-1086) for all-opaque scenes: every
// hit tightens the cutoff, so the two-loop structure collapses to "keep the nearest, first wins".
// SCALP: leaves may hold unbatched triangles and solids (NtCompositeDev::has_scalar_prims).  Scenes made of batches alone --
// every polytope of the reference's scripts -- take the instantiations without: no call to solid_intersects in the leaf
// loops, and with it a good part of the shading kernel's registers (214 VGPRs with, two waves a SIMD).
template <int N, bool FEAT, bool STATS, bool SCALP = FEAT>
__device__ __forceinline__ bool leaf_closest(const NtCompositeDev &sc, const WaveLds &w, int lane, int start, int count,
                                             const float (&o)[N], const float (&d)[N], int skip_item, int skip_lane, Hit &hit, Stats &st) {
    bool improved = false;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        const int kind = item & 3;
        const int idx = item >> 2;
        if (mbox_seen(w, lane, item)) continue;
        if (kind == 0) {
            const int sl = item == skip_item ? skip_lane : -1;
            float min_t = hit.dist;
            int r = -1;
            const float *base = sc.batch_recs + (size_t)idx * NT_DEV_BATCH * sc.rec_stride;
#pragma unroll
            for (int l = 0; l < NT_DEV_BATCH; ++l) {
                SimplexRec<N> s;
                s.load(base + (size_t)l * sc.rec_stride);
                const float t = simplex_batch_form<N>(s, o, d);
                if (l != sl && t != 0.0f && t < min_t) { min_t = t; r = l; }
            }
            if (STATS) st.simplex_tests += NT_DEV_BATCH;
            if (r >= 0) { hit.dist = min_t; hit.item = item; hit.lane = r; improved = true; }
        } else if (SCALP && item != skip_item) {
            float t;
            if (kind == 1) {
                SimplexRec<N> s;
                s.load(sc.tri_recs + (size_t)idx * sc.rec_stride);
                t = simplex_scalar_form<N>(s, o, d, hit.dist);
                if (STATS) st.simplex_tests += 1;
            } else {
                float no[N], nd[N];
                t = solid_intersects<N>(sc, idx, o, d, hit.dist, false, no, nd);
                if (STATS) st.solid_tests += 1;
            }
            if (t != 0.0f) { hit.dist = t; hit.item = item; hit.lane = -1; improved = true; }
        }
    }
    return improved;
}

// kd_node_intersection::operator() (tracer.hpp:1179-1243) with the recursion turned into an explicit
// stack of continuations.  Entry (far, t) stands for "after the near subtree of this branch returns":
//   - the near call's return value `hit` is "o_hit improved since the push", tracked with one integer
//     (`dirty`: number of bottom stack entries that have seen an improvement);
//   - `(hit && o_hit.dist <= t) || !n_far` => the frame returns (pop again), otherwise continue into far
//     with t_near = t; far == -1 encodes the `!n_far` case so that t_far is always the top entry's t.
// split distance of branch `nd` for the ray in the table: (split - origin[axis]) * invdir[axis] (tracer.hpp:1197)
// Not in the reference: its closest-hit walk (tracer.hpp:1179-1243) only stops at a branch when the NEAR subtree
// itself reported the hit (`hit && o_hit.dist <= t`, :1213); a ray whose hit was found in a leaf that ends before
// the hit point keeps descending into every later cell up to the far end of the scene (on the 120-cell: up to
// 2 900 of the 3 600 batches per ray).  A cell whose interval starts beyond the current hit cannot hold a
// closer one as long as every primitive is listed in each cell it overlaps -- the invariant the reference's own
// early exit relies on -- so such subtrees are dropped; the margin keeps cells that start within rounding
// distance of the hit (primitives embedded in a split plane, :1217-1222).  Off with nt_render_opts.strict_reference.
__device__ __forceinline__ bool nt_beyond_hit(float hit_dist, float t_near) {
    return hit_dist < t_near - 1e-4f * (1.0f + fabsf(t_near));
}

__device__ __forceinline__ float branch_t(const WaveLds &w, int lane, const NtNode &nd, bool &gt) {
    const float2 oi = w.ray[nd.axis * 64 + lane];
    gt = oi.x > nd.split;
    return (nd.split - oi.x) * oi.y;
}

template <int N, bool FEAT, bool STATS, bool SCALP = FEAT>
__device__ __noinline__ bool trace_closest(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                              float t_near, float t_far_root, int skip_item, int skip_lane, Hit &hit, Stats &st) {
    hit.dist = FLT_MAX;
    hit.item = -1;
    hit.lane = -1;
    mbox_reset(w, lane);
    int node = sc.root;
    int sp = 0;
    int dirty = 0;
    float t_far = t_far_root;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            if (sc.prune && nt_beyond_hit(hit.dist, t_near)) { node = -1; break; }
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (STATS) st.leaves += 1;
                if (leaf_closest<N, FEAT, STATS, SCALP>(sc, w, lane, nd.left, nd.right, o, d, skip_item, skip_lane, hit, st)) dirty = sp;
                node = -1;
                break;
            }
            if (STATS) st.branches += 1;
            const float2 oi = w.ray[nd.axis * 64 + lane];
            const float oa = oi.x, inv = oi.y;
            if (inv == inv) {                       // direction[axis] != 0
                if (oa == nd.split) { node = inv > 0.0f ? nd.right : nd.left; continue; }
                const float t = (nd.split - oa) * inv;
                const bool gt = oa > nd.split;
                const int n_near = gt ? nd.right : nd.left;
                const int n_far = gt ? nd.left : nd.right;
                if (t < 0.0f || t > t_far) { node = n_near; continue; }
                if (t < t_near) { node = n_far; continue; }
                if (n_near >= 0) {
                    if (sp < max_sp) {              // always true: depth <= tree depth (host-checked)
                        w.stack[sp * 64 + lane] = node;
                        ++sp;
                    }
                    t_far = t;
                    node = n_near;
                    continue;
                }
                node = n_far;
                t_near = t;
                continue;
            }
            node = oa >= nd.split ? nd.right : nd.left;
        }
        // the current frame has returned: resume the innermost continuation
        bool resumed = false;
        while (sp > 0) {
            --sp;
            const NtNode nd = sc.nodes[w.stack[sp * 64 + lane]];
            bool gt;
            const float t = branch_t(w, lane, nd, gt);
            const int far = gt ? nd.left : nd.right;
            const bool near_hit = sp < dirty;
            if (dirty > sp) dirty = sp;
            if ((near_hit && hit.dist <= t) || far < 0) continue;     // frame returns `hit`
            node = far;
            t_near = t;
            t_far = t_far_root;
            if (sp > 0) {
                const NtNode up = sc.nodes[w.stack[(sp - 1) * 64 + lane]];
                bool g2;
                t_far = branch_t(w, lane, up, g2);
            }
            resumed = true;
            break;
        }
        if (!resumed) break;
    }
    return hit.item >= 0;
}

// kd_leaf::occludes (tracer.hpp:1088-1124), all-opaque scenes
template <int N, bool STATS, bool SCALP = true>
__device__ __forceinline__ bool leaf_occludes(const NtCompositeDev &sc, int start, int count, const float (&o)[N], const float (&d)[N],
                                              float ldistance, int skip_item, int skip_lane, Stats &st) {
    // (the id of item i + 1 is fetched before item i's records: one memory round trip less on the chain id -> records -> test)
    int next_item = count > 0 ? sc.items[start] : 0;
    for (int i = 0; i < count; ++i) {
        const int item = next_item;
        if (i + 1 < count) next_item = sc.items[start + i + 1];
        const int kind = item & 3;
        const int idx = item >> 2;
        if (kind == 0) {
            const int sl = item == skip_item ? skip_lane : -1;
            const float *base = sc.batch_recs + (size_t)idx * NT_DEV_BATCH * sc.rec_stride;
            bool any = false;
#pragma unroll
            for (int l = 0; l < NT_DEV_BATCH; ++l) {
                SimplexRec<N> s;
                s.load(base + (size_t)l * sc.rec_stride);
                const float t = simplex_batch_form<N>(s, o, d);
                any = any || (l != sl && t != 0.0f && t < ldistance);
            }
            if (STATS) st.simplex_tests += NT_DEV_BATCH;
            if (any) return true;
        } else if (SCALP && item != skip_item) {
            float t;
            if (kind == 1) {
                SimplexRec<N> s;
                s.load(sc.tri_recs + (size_t)idx * sc.rec_stride);
                t = simplex_scalar_form<N>(s, o, d, ldistance);
                if (STATS) st.simplex_tests += 1;
            } else {
                float no[N], nd[N];
                t = solid_intersects<N>(sc, idx, o, d, ldistance, false, no, nd);
                if (STATS) st.solid_tests += 1;
            }
            if (t != 0.0f) return true;
        }
    }
    return false;
}

// _occludes (tracer.hpp:1258-1307), including `if(t < ldistance) return false;` at :1298 -- the far
// child is skipped whenever the split lies nearer than the light (reference quirk, reproduced).
// (experiment switch: -DNT_OCCL_ATTR=__forceinline__ puts the shadow walk into the shading loop instead of behind a call)
#ifndef NT_OCCL_ATTR
#define NT_OCCL_ATTR __noinline__
#endif
// (the ray travels BY VALUE: up to four dimensions that is registers -- the calling convention passes an aggregate of at most 16
// dwords directly -- where two references to the caller's arrays were two round trips through scratch memory per component)
template <int N> struct RayArg { float o[N], d[N]; };
template <int N, bool STATS, bool SCALP = true>
__device__ NT_OCCL_ATTR bool trace_occluded(const NtCompositeDev &sc, const WaveLds &w, int lane, const RayArg<N> ray,
                                            float ldistance, int skip_item, int skip_lane, Stats &st) {
    const float (&o)[N] = ray.o;
    const float (&d)[N] = ray.d;
    setup_ray_table<N>(w, lane, o, d);
    int node = sc.root;
    int sp = 0;
    float t_near = 0.0f;
    float t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    if (STATS) st.shadow_rays += 1;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (STATS) st.leaves += 1;
                if (leaf_occludes<N, STATS, SCALP>(sc, nd.left, nd.right, o, d, ldistance, skip_item, skip_lane, st)) return true;
                node = -1;
                break;
            }
            if (STATS) st.branches += 1;
            const float2 oi = w.ray[nd.axis * 64 + lane];
            const float oa = oi.x, inv = oi.y;
            if (inv == inv) {
                if (oa == nd.split) { node = inv > 0.0f ? nd.right : nd.left; continue; }
                const float t = (nd.split - oa) * inv;
                const bool gt = oa > nd.split;
                const int n_near = gt ? nd.right : nd.left;
                const int n_far = gt ? nd.left : nd.right;
                if (t < 0.0f || t > t_far) { node = n_near; continue; }
                if (t < t_near) { node = n_far; continue; }
                if (n_near >= 0) {
                    if (sp < max_sp) {
                        w.stack[sp * 64 + lane] = node;
                        ++sp;
                    }
                    t_far = t;
                    node = n_near;
                    continue;
                }
                if (t < ldistance) { node = -1; break; }     // :1298 with n_near == nullptr
                t_near = t;
                node = n_far;
                continue;
            }
            node = oa >= nd.split ? nd.right : nd.left;
        }
        bool resumed = false;
        while (sp > 0) {
            --sp;
            const NtNode nd = sc.nodes[w.stack[sp * 64 + lane]];
            bool gt;
            const float t = branch_t(w, lane, nd, gt);
            const int far = gt ? nd.left : nd.right;
            if (t < ldistance || far < 0) continue;             // frame returns false
            node = far;
            t_near = t;
            t_far = FLT_MAX;
            if (sp > 0) {
                const NtNode up = sc.nodes[w.stack[(sp - 1) * 64 + lane]];
                bool g2;
                t_far = branch_t(w, lane, up, g2);
            }
            resumed = true;
            break;
        }
        if (!resumed) return false;
    }
}

// composite_scene::aabb_distance (tracer.hpp:1892-1918)
template <int N>
__device__ __forceinline__ float aabb_distance(const NtCompositeDev &sc, const float (&o)[N], const float (&d)[N]) {
    float bs[N], be[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { bs[k] = sc.aabb[k]; be[k] = sc.aabb[N + k]; }
    bool done = false;
    float result = -1.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float di = d[i];
        const float face = di > 0.0f ? bs[i] : be[i];
        float dist = (face - o[i]) / di;
        const bool neg = dist < 0.0f;
        if (neg) dist = 0.0f;
        bool ok = !done && di != 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float p = d[j] * dist + o[j];
            const bool outside = p >= be[j] || p <= bs[j];
            if (j != i) ok = ok && !outside;
            else ok = ok && !(neg && outside);      // skip = -1 when dist was clamped: axis i is tested too
        }
        if (ok) { done = true; result = dist; }
    }
    return result;
}

struct Color3 { float r, g, b; };
__device__ __forceinline__ Color3 c3(float r, float g, float b) { Color3 c; c.r = r; c.g = g; c.b = b; return c; }
__device__ __forceinline__ Color3 c3p(const float *p) { return c3(p[0], p[1], p[2]); }
__device__ __forceinline__ Color3 cadd(Color3 a, Color3 b) { return c3(a.r + b.r, a.g + b.g, a.b + b.b); }
__device__ __forceinline__ Color3 cmul(Color3 a, Color3 b) { return c3(a.r * b.r, a.g * b.g, a.b * b.b); }
__device__ __forceinline__ Color3 cscale(Color3 a, float s) { return c3(a.r * s, a.g * s, a.b * s); }

__device__ __forceinline__ const float *material_of(const NtCompositeDev &sc, int item, int lane) {
    const int kind = item & 3, idx = item >> 2;
    int m;
    if (kind == 0) m = sc.batch_mats[idx * NT_DEV_BATCH + lane];
    else if (kind == 1) m = sc.tri_mats[idx];
    else m = sc.solid_mats[idx];
    return sc.materials + 10 * m;
}

// normal ray of the recorded hit (what the reference stored in o_hit.normal)
template <int N, bool FEAT>          // (FEAT here: hits may be solids)
__device__ __forceinline__ void hit_normal(const NtCompositeDev &sc, const Hit &hit, const float (&o)[N], const float (&d)[N],
                                           float (&no)[N], float (&nd)[N]) {
    const int kind = hit.item & 3, idx = hit.item >> 2;
    if (!FEAT || kind != 2) {
        const float *rec = kind == 0 ? sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + hit.lane) * sc.rec_stride
                                     : sc.tri_recs + (size_t)idx * sc.rec_stride;
        float fn[N];
#pragma unroll
        for (int k = 0; k < N; ++k) fn[k] = rec[1 + k];
        float denom = fn[0] * d[0];
#pragma unroll
        for (int k = 1; k < N; ++k) denom = denom + fn[k] * d[k];
        const float len = sqrtf(dotN<N>(fn, fn));
#pragma unroll
        for (int k = 0; k < N; ++k) {
            no[k] = o[k] + hit.dist * d[k];
            const float u = fn[k] / len;
            nd[k] = denom > 0.0f ? -u : u;
        }
    } else {
        solid_intersects<N>(sc, idx, o, d, FLT_MAX, true, no, nd);
    }
}

// append_specular (tracer.hpp:1701-1707)
template <int N>
__device__ __forceinline__ void append_specular(Color3 &c, float &a, const float *m, Color3 light_c, const float (&target)[N],
                                                const float (&normal)[N], const float (&light_dir)[N]) {
    float tmp[N];
#pragma unroll
    for (int k = 0; k < N; ++k) tmp[k] = light_dir[k] - target[k];
    const float len = sqrtf(dotN<N>(tmp, tmp));
#pragma unroll
    for (int k = 0; k < N; ++k) tmp[k] = tmp[k] / len;
    const float base = powf(dotN<N>(normal, tmp), m[9]) * m[8];
    c = cadd(c, cscale(cscale(cmul(c3p(m + 3), light_c), base), (1.0f - a)));
    a += base * (1.0f - a);
    c = cscale(c, a);
}

struct Level {      // one frame of the base_color/ray_color recursion that is waiting on its reflection
    Color3 spec, r0, c;
    float spec_a, refl;
};

// ray_color's miss branch (tracer.hpp:1866-1867)
// v[idx] for a wave-uniform idx without indexing registers: a chain of v_cndmask on uniform masks, spelled in asm
// because LLVM folds the equivalent C select chain back into an indexed array, which lands in scratch memory
// (a vector-memory round trip per use, and scratch-using waves are admitted at half the occupancy).
template <int N>
__device__ __forceinline__ float pick_uniform(const float (&v)[N], int idx) {
    const int u = __builtin_amdgcn_readfirstlane(idx);
    float r = v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) {
        const unsigned long long is_k = __builtin_amdgcn_ballot_w64(u == k);
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(r), "v"(v[k]), "s"(is_k));
    }
    return r;
}

// v[idx] for a per-lane idx, same idea (the masks come from v_cmp through ballot)
template <int N>
__device__ __forceinline__ float pick_lane(const float (&v)[N], int idx) {
    float r = v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) {
        const unsigned long long is_k = __builtin_amdgcn_ballot_w64(idx == k);
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(r), "v"(v[k]), "s"(is_k));
    }
    return r;
}

template <int N>
__device__ __forceinline__ Color3 background_color(const NtCompositeDev &sc, const float (&d)[N]) {
    const float iv = pick_uniform<N>(d, sc.bg_axis);          // target.direction[bg_gradient_axis]
    return iv >= 0.0f ? cadd(cscale(c3p(sc.bg1), iv), cscale(c3p(sc.bg2), 1.0f - iv))
                      : cadd(cscale(c3p(sc.bg3), -iv), cscale(c3p(sc.bg2), 1.0f + iv));
}

// base_color (tracer.hpp:1768-1854) for the scripted configuration: batches only, no lights, no reflective
// material -- the camera light and its specular term.  Same operations, same order as composite_color.
template <int N>
__device__ __forceinline__ Color3 surface_color_lean(const NtCompositeDev &sc, const Hit &hit, const float (&o)[N], const float (&d)[N]) {
    float no[N], nd[N];
    hit_normal<N, false>(sc, hit, o, d, no, nd);
    const float *m = material_of(sc, hit.item, hit.lane);
    Color3 light = c3(0.0f, 0.0f, 0.0f), specular = c3(0.0f, 0.0f, 0.0f);
    float spec_a = 0.0f;
    const float sine = -dotN<N>(d, nd);
    if (sc.camera_light && sine > 0.0f) {
        light = cadd(light, c3(sine, sine, sine));
        if (m[8] != 0.0f) {
            const float base = powf(sine, m[9]) * m[8];
            specular = cadd(specular, cscale(cscale(c3p(m + 3), base), (1.0f - spec_a)));
            spec_a += base * (1.0f - spec_a);
            specular = cscale(specular, spec_a);
        }
    }
    const Color3 r0 = cadd(c3p(sc.ambient), cmul(c3p(m), light));
    return cadd(specular, cscale(r0, 1.0f - spec_a));
}

// `primary`: when not null, the closest hit of the depth-0 ray has already been found (by the packet walk) and
// is taken from there instead of being traced here.
template <int N, bool FEAT, bool STATS, bool SCALP = FEAT>
__device__ __forceinline__ Color3 composite_color(const NtCompositeDev &sc, const WaveLds &w, int lane, float (&o)[N], float (&d)[N], Stats &st,
                                                  const Hit *primary = nullptr) {
    Level levels[FEAT ? NT_DEV_MAX_REFLECT : 1];
    Color3 deep_a = c3(0.0f, 0.0f, 0.0f), deep_b = c3(1.0f, 1.0f, 1.0f);      // levels beyond the stack: colour = deep_a + deep_b * (next)
    int depth = 0;
    int skip_item = -1, skip_lane = -1;
    Color3 result;
    for (;;) {
        // ---- ray_color (tracer.hpp:1856-1883) ----
        if (STATS) st.rays += 1;
        Hit hit;
        hit.item = -1;
        hit.lane = -1;
        hit.dist = FLT_MAX;
        bool found = false;
        if (primary && depth == 0) {
            hit = *primary;
            found = hit.item >= 0;
        } else {
            const float dist = aabb_distance<N>(sc, o, d);
            if (dist >= 0.0f) {
                if (STATS && depth == 0) st.aabb_enter += 1;
                setup_ray_table<N>(w, lane, o, d);
                found = trace_closest<N, FEAT, STATS, SCALP>(sc, w, lane, o, d, dist, FLT_MAX, skip_item, skip_lane, hit, st);
            }
        }
        if (!found) {
            result = background_color<N>(sc, d);
            break;
        }
        if (STATS && depth == 0) st.hits += 1;

        // ---- base_color (tracer.hpp:1768-1854) ----
        float no[N], nd[N];
        hit_normal<N, SCALP>(sc, hit, o, d, no, nd);
        const float *m = material_of(sc, hit.item, hit.lane);
        Color3 light = c3(0.0f, 0.0f, 0.0f), specular = c3(0.0f, 0.0f, 0.0f);
        float spec_a = 0.0f;

        if (FEAT) {
            for (int li = 0; li < sc.n_point_lights; ++li) {
                const float *pos = sc.pl_pos + (size_t)li * N;
                const Color3 plc = c3p(sc.pl_color + 3 * li);
                float lv[N];
#pragma unroll
                for (int k = 0; k < N; ++k) lv[k] = no[k] - pos[k];
                const float ldist = sqrtf(dotN<N>(lv, lv));
#pragma unroll
                for (int k = 0; k < N; ++k) lv[k] = lv[k] / ldist;
                const float sine = dotN<N>(nd, lv);
                if (sine > 0.0f) {
                    const float strength = nt_falloff(ldist, N - 1);
                    if (sc.shadows) {
                        if (fmaxf(plc.r, fmaxf(plc.g, plc.b)) * strength * sine > NT_LIGHT_THRESHOLD) {
                            RayArg<N> sray;
#pragma unroll
                            for (int k = 0; k < N; ++k) { sray.o[k] = no[k]; sray.d[k] = lv[k]; }
                            if (!trace_occluded<N, STATS, SCALP>(sc, w, lane, sray, ldist, hit.item, hit.lane, st)) {
                                const Color3 filtered = cscale(plc, strength);
                                light = cadd(light, cscale(filtered, sine));
                                if (m[8] != 0.0f) append_specular<N>(specular, spec_a, m, filtered, d, nd, lv);
                            }
                        }
                    } else {
                        light = cadd(light, cscale(cscale(plc, strength), sine));
                    }
                }
            }
            for (int li = 0; li < sc.n_global_lights; ++li) {
                const float *gd = sc.gl_dir + (size_t)li * N;
                const Color3 glc = c3p(sc.gl_color + 3 * li);
                float gdir[N], neg[N];
#pragma unroll
                for (int k = 0; k < N; ++k) { gdir[k] = gd[k]; neg[k] = -gd[k]; }
                const float sine = -dotN<N>(nd, gdir);
                if (sine > 0.0f) {
                    if (sc.shadows) {
                        RayArg<N> sray;
#pragma unroll
                        for (int k = 0; k < N; ++k) { sray.o[k] = no[k]; sray.d[k] = neg[k]; }
                        if (!trace_occluded<N, STATS, SCALP>(sc, w, lane, sray, FLT_MAX, hit.item, hit.lane, st)) {
                            light = cadd(light, cscale(glc, sine));
                            if (m[8] != 0.0f) append_specular<N>(specular, spec_a, m, glc, d, nd, neg);
                        }
                    } else {
                        light = cadd(light, cscale(glc, sine));
                    }
                }
            }
        }

        const float sine = -dotN<N>(d, nd);
        if (sc.camera_light && sine > 0.0f) {
            light = cadd(light, c3(sine, sine, sine));
            if (m[8] != 0.0f) {
                const float base = powf(sine, m[9]) * m[8];
                specular = cadd(specular, cscale(cscale(c3p(m + 3), base), (1.0f - spec_a)));
                spec_a += base * (1.0f - spec_a);
                specular = cscale(specular, spec_a);
            }
        }
        const Color3 r0 = cadd(c3p(sc.ambient), cmul(c3p(m), light));

        if (FEAT && m[7] != 0.0f && depth < sc.max_reflect_depth) {
            if (depth < NT_DEV_MAX_REFLECT) {
                Level &L = levels[depth];
                L.spec = specular;
                L.spec_a = spec_a;
                L.r0 = r0;
                L.c = c3p(m);
                L.refl = m[7];
            } else {
                // deeper than the level stack: this level's colour is alpha + beta * (colour of the reflected ray); fold it
                // into the running pair for everything below the stack (same terms as the unwinding loop, associated
                // differently: the contribution is damped by sixteen reflectivities by now)
                const float k1 = 1.0f - spec_a;
                const Color3 alpha = cadd(specular, cscale(cscale(r0, 1.0f - m[7]), k1));
                const Color3 beta = cscale(cscale(c3p(m), m[7]), k1);
                deep_a = cadd(deep_a, cmul(deep_b, alpha));
                deep_b = cmul(deep_b, beta);
            }
            const float f = -2.0f * sine;
#pragma unroll
            for (int k = 0; k < N; ++k) { d[k] = d[k] - nd[k] * f; o[k] = no[k]; }
            skip_item = hit.item;
            skip_lane = hit.lane;
            ++depth;
            continue;
        }
        result = cadd(specular, cscale(r0, 1.0f - spec_a));
        break;
    }
    if (FEAT) {
        if (depth > NT_DEV_MAX_REFLECT) {
            result = cadd(deep_a, cmul(deep_b, result));
            depth = NT_DEV_MAX_REFLECT;
        }
        while (depth > 0) {
            --depth;
            const Level &L = levels[depth];
            const Color3 r = cadd(cscale(cmul(L.c, result), L.refl), cscale(L.r0, 1.0f - L.refl));
            result = cadd(L.spec, cscale(r, 1.0f - L.spec_a));
        }
    }
    return result;
}

// --------------------------------------------------------------------------------------
// Transparent materials (opacity < 1): the general ray_color / base_color recursion
// (tracer.hpp:1768-1883) with transparent-hit lists, run as an explicit frame stack.
// Slow path: lists and frames live in per-lane scratch memory.  Normals follow the "clean"
// semantics documented in DESIGN.md (a hit's normal is that of the primitive that was hit).
// --------------------------------------------------------------------------------------
#define NT_TH_MAX 24        // transparent hits kept per ray (the reference misbehaves beyond 10)
#define NT_TFRAMES 6        // ray_color frames: max_reflect_depth <= 5 when transparency is present
#define NT_STK_MARK 0x80000000u

struct THit {
    float dist;
    int item;
    int lane;
};

struct TList {
    THit e[NT_TH_MAX];
    int n;
};

__device__ __forceinline__ void tl_add(TList &l, float dist, int item, int lane) {
    if (l.n < NT_TH_MAX) {
        l.e[l.n].dist = dist;
        l.e[l.n].item = item;
        l.e[l.n].lane = lane;
        ++l.n;
    }
}

// trim_intersections (tracer.hpp:784-789) with quick_list::remove_at's swap-with-last (:723-728)
__device__ __forceinline__ void tl_trim(TList &l, float dist, int from) {
    while (from < l.n) {
        if (l.e[from].dist >= dist) {
            --l.n;
            if (from != l.n) l.e[from] = l.e[l.n];
        } else {
            ++from;
        }
    }
}

// sort_and_unique (tracer.hpp:714-721): ascending dist, adjacent equal targets collapsed
__device__ __forceinline__ void tl_sort_unique(TList &l) {
    for (int i = 1; i < l.n; ++i) {
        const THit x = l.e[i];
        int j = i - 1;
        while (j >= 0 && l.e[j].dist > x.dist) { l.e[j + 1] = l.e[j]; --j; }
        l.e[j + 1] = x;
    }
    int w = 0;
    for (int i = 0; i < l.n; ++i) {
        if (w == 0 || !(l.e[w - 1].item == l.e[i].item && l.e[w - 1].lane == l.e[i].lane)) {
            if (w != i) l.e[w] = l.e[i];
            ++w;
        }
    }
    l.n = w;
}

// one primitive test for the transparent path: returns t (0 = miss) and the batch lane
template <int N>
__device__ __forceinline__ float test_item(const NtCompositeDev &sc, int item, const float (&o)[N], const float (&d)[N], float cutoff,
                                           int skip_item, int skip_lane, int &lane_out) {
    const int kind = item & 3, idx = item >> 2;
    lane_out = -1;
    if (kind == 0) {
        const int sl = item == skip_item ? skip_lane : -1;
        float min_t = cutoff;
        int r = -1;
        const float *base = sc.batch_recs + (size_t)idx * NT_DEV_BATCH * sc.rec_stride;
#pragma unroll
        for (int l = 0; l < NT_DEV_BATCH; ++l) {
            SimplexRec<N> s;
            s.load(base + (size_t)l * sc.rec_stride);
            const float t = simplex_batch_form<N>(s, o, d);
            if (l != sl && t != 0.0f && t < min_t) { min_t = t; r = l; }
        }
        lane_out = r;
        return r >= 0 ? min_t : 0.0f;
    }
    if (kind == 1) {
        SimplexRec<N> s;
        s.load(sc.tri_recs + (size_t)idx * sc.rec_stride);
        return simplex_scalar_form<N>(s, o, d, cutoff);
    }
    float no[N], nd[N];
    return solid_intersects<N>(sc, idx, o, d, cutoff, false, no, nd);
}

// --------------------------------------------------------------------------------------
// Reference-faithful normals.  The reference's tests take the normal ray to fill in as an in/out argument, and the
// first loop of kd_leaf::intersects passes o_hit.normal itself (tracer.hpp:1001,1020).  What a test writes there and
// when, restated exactly (the oracle does the same, oracle/ntracer_oracle.c):
//   * triangle / triangle_batch: nothing unless the test succeeds (tracer.hpp:431-437, 583-596);
//   * hypersphere: nothing unless it succeeds (:154-173);
//   * hypercube: normal.origin[i] = +-1 for every axis it tries and normal.origin[j] = p_j for the coordinates it checks
//     before the first one that fails -- in the solid's LOCAL coordinates, success or not (:126-152); direction and the
//     transformation back to world space only on success (:262-275).
// So a later test that finds a transparent surface, or a later Solid cube that is missed, leaves its marks on the normal
// ray of an opaque hit found earlier -- and base_color shades that hit with them.
// --------------------------------------------------------------------------------------

// hypercube_intersects (tracer.hpp:126-152) with its writes to `no` / `nd` in the reference's order
template <int N>
__device__ __forceinline__ float cube_local_marks(const float (&o)[N], const float (&d)[N], float cutoff, float (&no)[N], float (&nd)[N]) {
    bool alive = true;          // the function has not returned yet
    float result = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float di = d[i];
        const bool tried = alive && di != 0.0f;
        const float s = di < 0.0f ? 1.0f : -1.0f;
        if (tried) no[i] = s;
        const float dist = (s - o[i]) / di;
        bool going = tried && dist > 0.0f;        // inside the j loop, no miss so far
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j != i) {
                const float p = d[j] * dist + o[j];
                if (going) no[j] = p;
                going = going && !(fabsf(p) > (1.0f + NT_FUZZ));
            }
        }
        if (going) {
            alive = false;
            if (!(dist >= cutoff)) {
                result = dist;
#pragma unroll
                for (int j = 0; j < N; ++j) nd[j] = j == i ? s : 0.0f;        // (no[i] == s)
            }
        }
    }
    return result;
}

// solid::intersects (tracer.hpp:251-276) writing through to the caller's normal ray as the reference does
template <int N>
__device__ __noinline__ float solid_intersects_marks(const NtCompositeDev &sc, int idx, const float (&o)[N], const float (&d)[N], float cutoff,
                                                     float (&no)[N], float (&nd)[N]) {
    const float *orient = sc.solid_recs + (size_t)idx * (2 * N * N + N);
    const float *inv = orient + N * N;
    const float *pos = inv + N * N;
    float lo[N], ld[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        lo[i] = dotP<N>(inv + i * N, o) - pos[i];
        ld[i] = dotP<N>(inv + i * N, d);
    }
    float dist;
    if (sc.solid_types[idx] == 1) {
        dist = cube_local_marks<N>(lo, ld, cutoff, no, nd);
    } else {
        float so[N], sd[N];
        dist = sphere_local<N>(lo, ld, cutoff, so, sd);
        if (dist != 0.0f) {
#pragma unroll
            for (int i = 0; i < N; ++i) { no[i] = so[i]; nd[i] = sd[i]; }
        }
    }
    if (dist == 0.0f) return 0.0f;
    float tmp[N], ndl[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { tmp[i] = no[i] + pos[i]; ndl[i] = nd[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        no[i] = dotP<N>(orient + i * N, tmp);
        nd[i] = dotP<N>(orient + i * N, ndl);
    }
    return dist;
}

// test_item that fills in (no, nd) the way the reference's tests fill in their `normal` argument
template <int N>
__device__ __forceinline__ float test_item_marks(const NtCompositeDev &sc, int item, const float (&o)[N], const float (&d)[N], float cutoff,
                                                 int skip_item, int skip_lane, int &lane_out, float (&no)[N], float (&nd)[N]) {
    if ((item & 3) == 2) {
        lane_out = -1;
        return solid_intersects_marks<N>(sc, item >> 2, o, d, cutoff, no, nd);
    }
    const float t = test_item<N>(sc, item, o, d, cutoff, skip_item, skip_lane, lane_out);
    if (t != 0.0f) {
        Hit h;
        h.dist = t;
        h.item = item;
        h.lane = lane_out;
        hit_normal<N, false>(sc, h, o, d, no, nd);           // (tracer.hpp:431-437, 583-596)
    }
    return t;
}

// The reference's `checked` list (tracer.hpp:782,832,1166) kept exactly: one bit per primitive for this lane.
struct Checked {
    uint32_t *bits;          // this lane's column: word w at bits[w * stride]
    long long stride;
    int words, n_batches, n_triangles;
};
__device__ __forceinline__ void checked_reset(const Checked &c) {
    for (int w = 0; w < c.words; ++w) c.bits[(long long)w * c.stride] = 0u;
}
__device__ __forceinline__ bool checked_seen(const Checked &c, int item) {
    const int kind = item & 3, idx = item >> 2;
    const int id = kind == 0 ? idx : (kind == 1 ? c.n_batches + idx : c.n_batches + c.n_triangles + idx);
    uint32_t *p = c.bits + (long long)(id >> 5) * c.stride;
    const uint32_t v = *p, m = 1u << (id & 31);
    *p = v | m;
    return (v & m) != 0u;
}

// kd_leaf<Store,true>::intersects (tracer.hpp:977-1086) with transparent hits: first loop until an opaque
// hit, then "is there anything closer?", then trim_intersections with the LAST test's dist (:1084).
// ALIAS: the hit's normal ray (hn_o, hn_d) is carried along and written to exactly as the reference writes to
// o_hit.normal; `ck` is the exact `checked` list.  Otherwise the normal is worked out from the hit afterwards.
template <int N, bool ALIAS>
__device__ __noinline__ bool leaf_closest_t(const NtCompositeDev &sc, const WaveLds &w, int lane, int start, int count, const float (&o)[N],
                                            const float (&d)[N], int skip_item, int skip_lane, Hit &hit, TList &th, const Checked &ck,
                                            float (&hn_o)[N], float (&hn_d)[N]) {
    const int h_start = th.n;
    bool found = false;
    float dist_last = 0.0f;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        if ((item & 3) != 0 && item == skip_item) continue;
        // the reference's exact `checked` list in both modes: a repeated test is not harmless here -- the trim below uses the
        // LAST test's distance (tracer.hpp:1084), so a test the reference skips changes which transparent hits survive
        if (checked_seen(ck, item)) continue;
        int l;
        float t;
        if (ALIAS) {
            if (!found) {
                // first loop: the test writes to o_hit.normal itself (tracer.hpp:1001,1020)
                t = test_item_marks<N>(sc, item, o, d, hit.dist, skip_item, skip_lane, l, hn_o, hn_d);
            } else {
                // "is there anything closer?": a separate new_normal, copied on an opaque hit (:1037-1082)
                float nn_o[N], nn_d[N];
                t = test_item_marks<N>(sc, item, o, d, hit.dist, skip_item, skip_lane, l, nn_o, nn_d);
                if (t != 0.0f && material_of(sc, item, l)[6] >= 1.0f) {
#pragma unroll
                    for (int k = 0; k < N; ++k) { hn_o[k] = nn_o[k]; hn_d[k] = nn_d[k]; }
                }
            }
        } else {
            t = test_item<N>(sc, item, o, d, hit.dist, skip_item, skip_lane, l);
        }
        dist_last = t;
        if (t != 0.0f) {
            if (material_of(sc, item, l)[6] >= 1.0f) {
                hit.dist = t;
                hit.item = item;
                hit.lane = l;
                if (!found) {           // `goto hit`: the item is re-tested against its own distance and misses
                    found = true;
                    dist_last = 0.0f;
                }
            } else {
                tl_add(th, t, item, l);
            }
        }
    }
    if (found) tl_trim(th, dist_last, h_start);
    return found;
}

// kd_node_intersection::operator() with the transparent-list trims (tracer.hpp:1211-1231).  Stack entries:
// branch index | list size at the push << 24; NT_STK_MARK marks "far side of this branch is being walked
// after a near hit: trim the list when it returns with a closer hit".
template <int N, bool ALIAS>
__device__ __noinline__ bool trace_closest_t(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                             float t_near, int skip_item, int skip_lane, Hit &hit, TList &th, const Checked &ck,
                                             float (&hn_o)[N], float (&hn_d)[N]) {
    hit.dist = FLT_MAX;
    hit.item = -1;
    hit.lane = -1;
    th.n = 0;
    checked_reset(ck);
    int node = sc.root;
    int sp = 0;
    int dirty = 0;
    float t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (leaf_closest_t<N, ALIAS>(sc, w, lane, nd.left, nd.right, o, d, skip_item, skip_lane, hit, th, ck, hn_o, hn_d)) dirty = sp;
                node = -1;
                break;
            }
            const float2 oi = w.ray[nd.axis * 64 + lane];
            const float oa = oi.x, inv = oi.y;
            if (inv == inv) {
                if (oa == nd.split) { node = inv > 0.0f ? nd.right : nd.left; continue; }
                const float t = (nd.split - oa) * inv;
                const bool gt = oa > nd.split;
                const int n_near = gt ? nd.right : nd.left;
                const int n_far = gt ? nd.left : nd.right;
                if (t < 0.0f || t > t_far) { node = n_near; continue; }
                if (t < t_near) { node = n_far; continue; }
                if (n_near >= 0) {
                    if (sp < max_sp) {
                        w.stack[sp * 64 + lane] = (int)((unsigned)node | ((unsigned)th.n << 24));
                        ++sp;
                    }
                    t_far = t;
                    node = n_near;
                    continue;
                }
                node = n_far;
                t_near = t;
                continue;
            }
            node = oa >= nd.split ? nd.right : nd.left;
        }
        bool resumed = false;
        while (sp > 0) {
            --sp;
            const unsigned e = (unsigned)w.stack[sp * 64 + lane];
            const bool improved = sp < dirty;
            if (dirty > sp) dirty = sp;
            const int h_start = (int)((e >> 24) & 0x7fu);
            if (e & NT_STK_MARK) {
                // the far call of tracer.hpp:1225 has returned
                if (improved) tl_trim(th, hit.dist, h_start);
                continue;                                           // return true
            }
            const NtNode nd = sc.nodes[e & 0xffffffu];
            bool gt;
            const float t = branch_t(w, lane, nd, gt);
            const int far = gt ? nd.left : nd.right;
            if ((improved && hit.dist <= t) || far < 0) continue;   // return hit
            if (improved) {                                         // near hit beyond the split: walk far, then trim
                w.stack[sp * 64 + lane] = (int)(e | NT_STK_MARK);
                ++sp;
            }
            node = far;
            t_near = t;
            // t_far of this frame: the split of the nearest pending (non-marker) branch below
            t_far = FLT_MAX;
            for (int k = (improved ? sp - 2 : sp - 1); k >= 0; --k) {
                const unsigned ek = (unsigned)w.stack[k * 64 + lane];
                if (!(ek & NT_STK_MARK)) {
                    const NtNode up = sc.nodes[ek & 0xffffffu];
                    bool g2;
                    t_far = branch_t(w, lane, up, g2);
                    break;
                }
            }
            resumed = true;
            break;
        }
        if (!resumed) break;
    }
    return hit.item >= 0;
}

// _occludes + kd_leaf::occludes with transparent hits collected (tracer.hpp:1088-1124, 1258-1307)
template <int N>
__device__ __noinline__ bool trace_occluded_t(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                              float ldistance, int skip_item, int skip_lane, TList &sh) {
    setup_ray_table<N>(w, lane, o, d);
    sh.n = 0;
    int node = sc.root;
    int sp = 0;
    float t_near = 0.0f;
    float t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                for (int i = 0; i < nd.right; ++i) {
                    const int item = sc.items[nd.left + i];
                    if ((item & 3) != 0 && item == skip_item) continue;
                    int l;
                    const float t = test_item<N>(sc, item, o, d, ldistance, skip_item, skip_lane, l);
                    if (t != 0.0f) {
                        if (material_of(sc, item, l)[6] >= 1.0f) return true;
                        tl_add(sh, t, item, l);
                    }
                }
                node = -1;
                break;
            }
            const float2 oi = w.ray[nd.axis * 64 + lane];
            const float oa = oi.x, inv = oi.y;
            if (inv == inv) {
                if (oa == nd.split) { node = inv > 0.0f ? nd.right : nd.left; continue; }
                const float t = (nd.split - oa) * inv;
                const bool gt = oa > nd.split;
                const int n_near = gt ? nd.right : nd.left;
                const int n_far = gt ? nd.left : nd.right;
                if (t < 0.0f || t > t_far) { node = n_near; continue; }
                if (t < t_near) { node = n_far; continue; }
                if (n_near >= 0) {
                    if (sp < max_sp) { w.stack[sp * 64 + lane] = node; ++sp; }
                    t_far = t;
                    node = n_near;
                    continue;
                }
                if (t < ldistance) { node = -1; break; }
                t_near = t;
                node = n_far;
                continue;
            }
            node = oa >= nd.split ? nd.right : nd.left;
        }
        bool resumed = false;
        while (sp > 0) {
            --sp;
            const NtNode nd = sc.nodes[w.stack[sp * 64 + lane]];
            bool gt;
            const float t = branch_t(w, lane, nd, gt);
            const int far = gt ? nd.left : nd.right;
            if (t < ldistance || far < 0) continue;
            node = far;
            t_near = t;
            t_far = FLT_MAX;
            if (sp > 0) {
                const NtNode up = sc.nodes[w.stack[(sp - 1) * 64 + lane]];
                bool g2;
                t_far = branch_t(w, lane, up, g2);
            }
            resumed = true;
            break;
        }
        if (!resumed) return false;
    }
}

// light_reaches (tracer.hpp:1750-1766)
template <int N>
__device__ __forceinline__ bool light_reaches_t(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                                float ldistance, int skip_item, int skip_lane, Color3 &filtered) {
    TList sh;
    if (trace_occluded_t<N>(sc, w, lane, o, d, ldistance, skip_item, skip_lane, sh)) return false;
    if (sh.n) {
        tl_sort_unique(sh);
        for (int i = sh.n - 1; i >= 0; --i) filtered = cscale(filtered, 1.0f - material_of(sc, sh.e[i].item, sh.e[i].lane)[6]);
    }
    return true;
}

template <int N>
struct TFrame {
    float o[N], d[N];
    int depth, skip_item, skip_lane;
    int nsurf, j;                 // surfaces: [0] = opaque hit (item < 0: background), then transparent hits far -> near
    THit surf[NT_TH_MAX + 1];
    Color3 r;                     // colour composited so far
    Color3 spec, r0, c;           // base_color of surface j, waiting for its reflection
    float spec_a, refl;
    float hn_o[N], hn_d[N];       // ALIAS: o_hit.normal as the walk left it (the normal ray surface 0 is shaded with)
};

template <int N, bool ALIAS>
__device__ __noinline__ Color3 composite_color_t(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&org)[N], const float (&dir)[N],
                                                 const Checked &ck) {
    TFrame<N> frames[NT_TFRAMES];
    int fp = 0;
    {
        TFrame<N> &F = frames[0];
#pragma unroll
        for (int k = 0; k < N; ++k) { F.o[k] = org[k]; F.d[k] = dir[k]; }
        F.depth = 0;
        F.skip_item = -1;
        F.skip_lane = -1;
    }
    int state = 0;          // 0: trace the frame's ray, 1: shade its next surface, 2: a reflection has returned
    Color3 result = c3(0.0f, 0.0f, 0.0f);
    for (;;) {
        TFrame<N> &F = frames[fp];
        if (state == 0) {
            // ---- ray_color: intersect (tracer.hpp:1861-1868)
            float o[N], d[N];
#pragma unroll
            for (int k = 0; k < N; ++k) { o[k] = F.o[k]; d[k] = F.d[k]; }
            TList th;
            th.n = 0;
            Hit hit;
            hit.item = -1; hit.lane = -1; hit.dist = FLT_MAX;
            float hn_o[N], hn_d[N];
#pragma unroll
            for (int k = 0; k < N; ++k) { hn_o[k] = 0.0f; hn_d[k] = 0.0f; }       // ray_intersection starts out zeroed (oracle: memset)
            const float dist = aabb_distance<N>(sc, o, d);
            if (dist >= 0.0f) {
                setup_ray_table<N>(w, lane, o, d);
                trace_closest_t<N, ALIAS>(sc, w, lane, o, d, dist, F.skip_item, F.skip_lane, hit, th, ck, hn_o, hn_d);
            }
            if (ALIAS) {
#pragma unroll
                for (int k = 0; k < N; ++k) { F.hn_o[k] = hn_o[k]; F.hn_d[k] = hn_d[k]; }
            }
            tl_sort_unique(th);
            F.surf[0].dist = hit.dist;
            F.surf[0].item = hit.item;
            F.surf[0].lane = hit.lane;
            for (int i = 0; i < th.n; ++i) F.surf[1 + i] = th.e[th.n - 1 - i];      // farthest first (:1874)
            F.nsurf = 1 + th.n;
            F.j = 0;
            F.r = c3(0.0f, 0.0f, 0.0f);
            state = 1;
            continue;
        }
        float o[N], d[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { o[k] = F.o[k]; d[k] = F.d[k]; }
        Color3 col;
        float opacity = 1.0f;
        if (state == 1) {
            if (F.j == F.nsurf) {
                // ---- the frame is complete: hand its colour to the waiting base_color, or finish
                result = F.r;
                if (fp == 0) break;
                --fp;
                state = 2;
                continue;
            }
            const THit sf = F.surf[F.j];
            if (sf.item < 0) {            // miss: background (only surface 0 can be this)
                F.r = background_color<N>(sc, d);
                ++F.j;
                continue;
            }
            // ---- base_color (tracer.hpp:1768-1854)
            Hit hit;
            hit.dist = sf.dist; hit.item = sf.item; hit.lane = sf.lane;
            float no[N], nd[N];
            if (ALIAS && F.j == 0) {
                // the opaque hit is shaded with o_hit.normal as the walk left it (tracer.hpp:1864)
#pragma unroll
                for (int k = 0; k < N; ++k) { no[k] = F.hn_o[k]; nd[k] = F.hn_d[k]; }
            } else {
                hit_normal<N, true>(sc, hit, o, d, no, nd);      // transparent hits carry a copy made when they were found
            }
            const float *m = material_of(sc, hit.item, hit.lane);
            opacity = m[6];
            Color3 light = c3(0.0f, 0.0f, 0.0f), specular = c3(0.0f, 0.0f, 0.0f);
            float spec_a = 0.0f;
            for (int li = 0; li < sc.n_point_lights; ++li) {
                const float *pos = sc.pl_pos + (size_t)li * N;
                const Color3 plc = c3p(sc.pl_color + 3 * li);
                float lv[N];
#pragma unroll
                for (int k = 0; k < N; ++k) lv[k] = no[k] - pos[k];
                const float ldist = sqrtf(dotN<N>(lv, lv));
#pragma unroll
                for (int k = 0; k < N; ++k) lv[k] = lv[k] / ldist;
                const float sine = dotN<N>(nd, lv);
                if (sine > 0.0f) {
                    const float strength = nt_falloff(ldist, N - 1);
                    if (sc.shadows) {
                        if (fmaxf(plc.r, fmaxf(plc.g, plc.b)) * strength * sine > NT_LIGHT_THRESHOLD) {
                            Color3 filtered = plc;
                            if (light_reaches_t<N>(sc, w, lane, no, lv, ldist, hit.item, hit.lane, filtered)) {
                                filtered = cscale(filtered, strength);
                                light = cadd(light, cscale(filtered, sine));
                                if (m[8] != 0.0f) append_specular<N>(specular, spec_a, m, filtered, d, nd, lv);
                            }
                        }
                    } else {
                        light = cadd(light, cscale(cscale(plc, strength), sine));
                    }
                }
            }
            for (int li = 0; li < sc.n_global_lights; ++li) {
                const float *gd = sc.gl_dir + (size_t)li * N;
                const Color3 glc = c3p(sc.gl_color + 3 * li);
                float gdir[N], neg[N];
#pragma unroll
                for (int k = 0; k < N; ++k) { gdir[k] = gd[k]; neg[k] = -gd[k]; }
                const float sine = -dotN<N>(nd, gdir);
                if (sine > 0.0f) {
                    if (sc.shadows) {
                        Color3 filtered = glc;
                        if (light_reaches_t<N>(sc, w, lane, no, neg, FLT_MAX, hit.item, hit.lane, filtered)) {
                            light = cadd(light, cscale(filtered, sine));
                            if (m[8] != 0.0f) append_specular<N>(specular, spec_a, m, filtered, d, nd, neg);
                        }
                    } else {
                        light = cadd(light, cscale(glc, sine));
                    }
                }
            }
            const float sine = -dotN<N>(d, nd);
            if (sc.camera_light && sine > 0.0f) {
                light = cadd(light, c3(sine, sine, sine));
                if (m[8] != 0.0f) {
                    const float base = powf(sine, m[9]) * m[8];
                    specular = cadd(specular, cscale(cscale(c3p(m + 3), base), (1.0f - spec_a)));
                    spec_a += base * (1.0f - spec_a);
                    specular = cscale(specular, spec_a);
                }
            }
            const Color3 r0 = cadd(c3p(sc.ambient), cmul(c3p(m), light));
            if (m[7] != 0.0f && F.depth < sc.max_reflect_depth && fp + 1 < NT_TFRAMES) {
                F.spec = specular;
                F.spec_a = spec_a;
                F.r0 = r0;
                F.c = c3p(m);
                F.refl = m[7];
                TFrame<N> &G = frames[fp + 1];
                const float f = -2.0f * sine;
#pragma unroll
                for (int k = 0; k < N; ++k) { G.d[k] = d[k] - nd[k] * f; G.o[k] = no[k]; }
                G.depth = F.depth + 1;
                G.skip_item = hit.item;
                G.skip_lane = hit.lane;
                ++fp;
                state = 0;
                continue;
            }
            col = cadd(specular, cscale(r0, 1.0f - spec_a));
        } else {
            // ---- state 2: the reflection of surface j returned `result` (tracer.hpp:1842-1853)
            const Color3 r = cadd(cscale(cmul(F.c, result), F.refl), cscale(F.r0, 1.0f - F.refl));
            col = cadd(F.spec, cscale(r, 1.0f - F.spec_a));
            opacity = material_of(sc, F.surf[F.j].item, F.surf[F.j].lane)[6];
            state = 1;
        }
        // ---- ray_color: the opaque hit is the base, transparent hits are blended over it (:1864, :1878)
        if (F.j == 0) F.r = col;
        else F.r = cadd(cscale(col, opacity), cscale(F.r, 1.0f - opacity));
        ++F.j;
    }
    return result;
}

// The blocks stride over the tiles (tiles_x * tiles_y * frames of them): every resident lane owns a column of the `checked`
// bitmap, and that scratch is sized by the grid, not by the image.  ALIAS: the reference's o_hit.normal handling (see above);
// otherwise a hit keeps the normal of what was hit.
template <int N, bool ALIAS>
__global__ __launch_bounds__(256) void composite_kernel_t(NtCameraFixed cam_in, NtCompositeDev sc, NtTarget tg, int tiles_x, int tiles_y, int frames) {
    extern __shared__ float2 lds_raw[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const WaveLds w = wave_lds(reinterpret_cast<char *>(lds_raw), wv, sc.stack_depth, N);
    Checked ck;
    ck.bits = sc.checked + ((long long)blockIdx.x * 256 + tid);
    ck.stride = sc.checked_lanes;
    ck.words = sc.checked_words;
    ck.n_batches = sc.n_batches;
    ck.n_triangles = sc.n_triangles;
    int px, py;
    if (tg.colors_out) { px = 0; py = 0; }
    else { px = (wv & 1) * 8 + (lane & 7); py = (wv >> 1) * 8 + (lane >> 3); }
    const long long total = (long long)tiles_x * tiles_y * frames;
    for (long long tile = (long long)blockIdx.x; tile < total; tile += gridDim.x) {
        if (nt_aborted(tg)) return;                       // (the four waves of a block are independent: no barrier below)
        const int bz = (int)(tile / ((long long)tiles_x * tiles_y));
        const int rem = (int)(tile - (long long)bz * tiles_x * tiles_y);
        const int by = rem / tiles_x;
        const int bx = rem - by * tiles_x;
        const PixelRef pr = locate_pixel_at<16, 16>(tg, bx, by, bz, px, py, tid);
        if (pr.valid) {
            NtCameraFixed cam = cam_in;
            float org[N], right[N], up[N], fwd[N], dir[N];
            if (cam.buf) cam.buf += (size_t)bz * 4 * N;            // (load_camera adds blockIdx.z, which is 0 here)
            load_camera<N>(cam, org, right, up, fwd);
            primary_dir<N>(tg, right, up, fwd, pr.x, pr.y, dir);
            const Color3 c = composite_color_t<N, ALIAS>(sc, w, lane, org, dir, ck);
            emit_pixel(tg, pr, c.r, c.g, c.b);
        }
    }
}

__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// One wave renders an 8x8 pixel tile (coherent rays -> shared k-d path and broadcast record loads);
// a 256-thread block covers 16x16 pixels.
// (experiment switch: -DNT_SHADE_OCC=__attribute__((amdgpu_waves_per_eu(3,3))) holds the kernel to a register budget)
#ifndef NT_SHADE_OCC
#define NT_SHADE_OCC
#endif
template <int N, bool FEAT, bool STATS, bool SCALP = FEAT>
__global__ __launch_bounds__(256) NT_SHADE_OCC void composite_kernel(NtCameraFixed cam, NtCompositeDev sc, NtTarget tg) {
    extern __shared__ float2 lds_raw[];
    if (nt_aborted(tg)) return;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const WaveLds w = wave_lds(reinterpret_cast<char *>(lds_raw), wv, sc.stack_depth, N);

    int px, py;
    if (tg.colors_out) { px = 0; py = 0; }
    else { px = (wv & 1) * 8 + (lane & 7); py = (wv >> 1) * 8 + (lane >> 3); }
    const PixelRef pr = locate_pixel<16, 16>(tg, px, py, tid);

    Stats st = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pr.valid) {
        float org[N], right[N], up[N], fwd[N], dir[N];
        load_camera<N>(cam, org, right, up, fwd);
        primary_dir<N>(tg, right, up, fwd, pr.x, pr.y, dir);
        Color3 c;
        bool emit = true;
        if (FEAT && tg.hits) {
            // second pass of a lit scene: the primary hit was found by the packet kernel (which has already
            // written the pixels of the rays that hit nothing)
            const float4 h = reinterpret_cast<const float4 *>(tg.hits)[pr.hit_index];
            Hit hit;
            hit.dist = h.x;
            hit.item = __float_as_int(h.y);
            hit.lane = __float_as_int(h.z);
            emit = hit.item >= 0;
            if (emit) c = composite_color<N, FEAT, STATS, SCALP>(sc, w, lane, org, dir, st, &hit);
        } else {
            c = composite_color<N, FEAT, STATS, SCALP>(sc, w, lane, org, dir, st);
        }
        if (emit) emit_pixel(tg, pr, c.r, c.g, c.b);
    }
    if (STATS && sc.stats) {
        unsigned int v[8] = {st.rays, st.shadow_rays, st.branches, st.leaves, st.simplex_tests, st.solid_tests, st.hits, st.aabb_enter};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned int s = wave_sum(v[k]);
            if (lane == 0 && s) atomicAdd(sc.stats + k, (unsigned long long)s);
        }
    }
}

// --------------------------------------------------------------------------------------
// Persistent variant of the lean composite kernel (batches only, camera light only: the scripted
// configurations).  The k-d walk is the same as trace_closest, but every lane is a little state machine
// and the wave alternates between
//   phase A (cheap, divergent)   each lane advances -- branch steps, leaf entry, mailbox, pops -- until it
//                                holds ONE pending batch to test or its ray is finished;
//   phase B (expensive, converged) all lanes with a pending batch run the 4-simplex test together.
// Lanes whose ray is finished shade + write their pixel and take the next pixel from a global counter
// (one atomicAdd per wave: __ballot/__popcll rank), so a wave is not held hostage by its slowest ray:
// leaf sizes range from 1 to ~1800 batches in the 120-cell and the per-tile variant runs at ~25 % lane use.
// Pixels are numbered tile-major (8x8) so that refilled rays stay coherent.
struct PersistArgs {
    const float *cams;              // [frame][4][N]
    unsigned long long *counter;    // zeroed by the host before the launch
    long long total;                // frames * tiles_per_frame * 64
    int tiles_x, tiles_per_frame;
};

template <int N>
__global__ __launch_bounds__(256) void composite_persistent(NtCompositeDev sc, NtTarget tg, PersistArgs pa) {
    extern __shared__ float2 lds_raw[];
    if (nt_aborted(tg)) return;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const WaveLds w = wave_lds(reinterpret_cast<char *>(lds_raw), wv, sc.stack_depth, N);
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int max_sp = sc.stack_depth;

    // per-lane ray state
    bool running = false;          // a ray is in flight in this lane
    bool finished = false;         // ... and its traversal is complete (needs shading)
    bool exhausted = false;        // the pixel pool is empty
    long long out_off = 0;
    float o[N], d[N];
    Hit hit;
    hit.dist = FLT_MAX; hit.item = -1; hit.lane = -1;
    int node = -1, sp = 0, dirty = 0;
    float t_near = 0.0f, t_far = FLT_MAX;
    int leaf_pos = 0, leaf_end = 0;      // remaining items of the current leaf: [leaf_pos, leaf_end)
    bool in_leaf = false, improved = false;
#pragma unroll
    for (int k = 0; k < N; ++k) { o[k] = 0.0f; d[k] = 0.0f; }

    for (;;) {
        // ---------------- retire finished rays, refill idle lanes ----------------
        const unsigned long long run_mask = __builtin_amdgcn_ballot_w64(running && !finished);
        const int n_busy = (int)__popcll(run_mask);
        if (n_busy <= 48) {              // refill when a quarter of the wave is idle (or nothing is running)
            if (running && finished) {
                const Color3 c = hit.item >= 0 ? surface_color_lean<N>(sc, hit, o, d) : background_color<N>(sc, d);
                uint8_t *p = tg.dest + out_off;
                // same epilogue as emit_pixel (image mode)
                if (tg.pack_mode == NT_PACK_WORD32) {
                    const uint32_t wd = pack_word32(c.r, c.g, c.b, tg);
                    if (tg.bpp == 4 && tg.aligned4) *reinterpret_cast<uint32_t *>(p) = tg.reversed ? wd : bswap32(wd);
                    else store_pixel(p, tg, (uint64_t)wd << 32, 0);
                } else if (tg.pack_mode == NT_PACK_WORD64) {
                    store_pixel(p, tg, pack_word64(c.r, c.g, c.b, tg), 0);
                } else {
                    uint64_t hi, lo;
                    pack_pixel(c.r, c.g, c.b, tg, hi, lo);
                    store_pixel(p, tg, hi, lo);
                }
                running = false;
                finished = false;
            }
            // fetch until every idle lane has a ray that needs traversal, or the pool is empty
            for (;;) {
                const bool want = !running && !exhausted;
                const unsigned long long want_mask = __builtin_amdgcn_ballot_w64(want);
                if (want_mask == 0ull) break;
                const int need = (int)__popcll(want_mask);
                const int src = (int)__builtin_ctzll(want_mask);
                unsigned long long base = 0;
                if (lane == src) base = atomicAdd(pa.counter, (unsigned long long)need);
                const unsigned int blo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(base & 0xffffffffull), src);
                const unsigned int bhi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(base >> 32), src);
                base = ((unsigned long long)bhi << 32) | blo;
                if (want) {
                    const long long id = (long long)base + (long long)__popcll(want_mask & lane_lt);
                    if (id >= pa.total) {
                        exhausted = true;
                    } else {
                        const int frame = (int)(id / ((long long)pa.tiles_per_frame * 64));
                        const int rem = (int)(id - (long long)frame * pa.tiles_per_frame * 64);
                        const int tile = rem >> 6, within = rem & 63;
                        const int ty = tile / pa.tiles_x, tx = tile - ty * pa.tiles_x;
                        const int x = tx * 8 + (within & 7);
                        const int row = ty * 8 + (within >> 3);
                        bool valid = x < tg.width && row < tg.row_count;
                        int y = tg.row_begin + row;
                        const int orow = y;
                        if (valid && tg.band_world > 1) {
                            const int band = orow / tg.band_rows;
                            y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
                        }
                        valid = valid && y < tg.height;
                        if (valid) {
                            out_off = (long long)frame * tg.frame_stride + (long long)(tg.compact ? orow : y) * tg.pitch + (long long)x * tg.bpp;
                            const float *c = pa.cams + (size_t)frame * 4 * N;
                            float right[N], up[N], fwd[N];
#pragma unroll
                            for (int k = 0; k < N; ++k) { o[k] = c[k]; right[k] = c[N + k]; up[k] = c[2 * N + k]; fwd[k] = c[3 * N + k]; }
                            primary_dir<N>(tg, right, up, fwd, x, y, d);
                            running = true;
                            finished = true;            // until the AABB test says otherwise
                            hit.dist = FLT_MAX; hit.item = -1; hit.lane = -1;
                            const float dist = aabb_distance<N>(sc, o, d);
                            if (dist >= 0.0f) {
                                setup_ray_table<N>(w, lane, o, d);
                                mbox_reset(w, lane);
                                node = sc.root; sp = 0; dirty = 0;
                                t_near = dist; t_far = FLT_MAX;
                                in_leaf = false; improved = false;
                                finished = false;
                            }
                        }
                    }
                }
                // rays that missed the scene box are shaded right away so their lanes can refill in this loop
                if (running && finished) {
                    const Color3 c = background_color<N>(sc, d);
                    uint8_t *p = tg.dest + out_off;
                    if (tg.pack_mode == NT_PACK_WORD32) {
                        const uint32_t wd = pack_word32(c.r, c.g, c.b, tg);
                        if (tg.bpp == 4 && tg.aligned4) *reinterpret_cast<uint32_t *>(p) = tg.reversed ? wd : bswap32(wd);
                        else store_pixel(p, tg, (uint64_t)wd << 32, 0);
                    } else if (tg.pack_mode == NT_PACK_WORD64) {
                        store_pixel(p, tg, pack_word64(c.r, c.g, c.b, tg), 0);
                    } else {
                        uint64_t hi, lo;
                        pack_pixel(c.r, c.g, c.b, tg, hi, lo);
                        store_pixel(p, tg, hi, lo);
                    }
                    running = false;
                    finished = false;
                }
            }
            if (__builtin_amdgcn_ballot_w64(running) == 0ull) break;     // pool empty and nothing in flight
        }

        // ---------------- phase A: advance to the next pending batch ----------------
        int pending = -1;
        while (running && !finished && pending < 0) {
            if (in_leaf) {
                if (leaf_pos < leaf_end) {
                    const int item = sc.items[leaf_pos];
                    ++leaf_pos;
                    if (!mbox_seen(w, lane, item)) pending = item;
                    continue;
                }
                in_leaf = false;
                if (improved) dirty = sp;
                node = -1;
            }
            if (node >= 0) {
                if (sc.prune && nt_beyond_hit(hit.dist, t_near)) { node = -1; continue; }
                const NtNode nd = sc.nodes[node];
                if (nd.axis < 0) {
                    in_leaf = true;
                    improved = false;
                    leaf_pos = nd.left;
                    leaf_end = nd.left + nd.right;
                    continue;
                }
                const float2 oi = w.ray[nd.axis * 64 + lane];
                const float oa = oi.x, inv = oi.y;
                if (inv == inv) {
                    if (oa == nd.split) { node = inv > 0.0f ? nd.right : nd.left; continue; }
                    const float t = (nd.split - oa) * inv;
                    const bool gt = oa > nd.split;
                    const int n_near = gt ? nd.right : nd.left;
                    const int n_far = gt ? nd.left : nd.right;
                    if (t < 0.0f || t > t_far) { node = n_near; continue; }
                    if (t < t_near) { node = n_far; continue; }
                    if (n_near >= 0) {
                        if (sp < max_sp) { w.stack[sp * 64 + lane] = node; ++sp; }
                        t_far = t;
                        node = n_near;
                        continue;
                    }
                    node = n_far;
                    t_near = t;
                    continue;
                }
                node = oa >= nd.split ? nd.right : nd.left;
                continue;
            }
            // frame returned: pop
            if (sp == 0) { finished = true; break; }
            --sp;
            const NtNode nd = sc.nodes[w.stack[sp * 64 + lane]];
            bool gt;
            const float t = branch_t(w, lane, nd, gt);
            const int far = gt ? nd.left : nd.right;
            const bool near_hit = sp < dirty;
            if (dirty > sp) dirty = sp;
            if ((near_hit && hit.dist <= t) || far < 0) continue;
            node = far;
            t_near = t;
            t_far = FLT_MAX;
            if (sp > 0) {
                const NtNode up = sc.nodes[w.stack[(sp - 1) * 64 + lane]];
                bool g2;
                t_far = branch_t(w, lane, up, g2);
            }
        }

        // ---------------- phase B: the batch test, lanes converged ----------------
        if (pending >= 0) {
            const int idx = pending >> 2;
            float min_t = hit.dist;
            int r = -1;
            const float *base = sc.batch_recs + (size_t)idx * NT_DEV_BATCH * sc.rec_stride;
#pragma unroll
            for (int l = 0; l < NT_DEV_BATCH; ++l) {
                SimplexRec<N> sr;
                sr.load(base + (size_t)l * sc.rec_stride);
                const float t = simplex_batch_form<N>(sr, o, d);
                if (t != 0.0f && t < min_t) { min_t = t; r = l; }
            }
            if (r >= 0) { hit.dist = min_t; hit.item = pending; hit.lane = r; improved = true; }
        }
    }
}

// --------------------------------------------------------------------------------------
// Packet variant of the lean composite kernel: one wave walks the k-d tree ONCE for its 8x8 tile of primary
// rays.  Primary rays share the camera origin, so which child of a branch is "near" (tracer.hpp:1199-1200) is
// the same for all 64 lanes; what differs per lane is only whether it enters near, far or both, and its
// [t_near,t_far].  Control flow (node ids, leaf items) is therefore wave-uniform: node and simplex records
// are fetched with scalar loads into SGPRs (no per-lane gathers through the vector memory pipe, which bound
// the per-lane kernels), and every lane that the reference would take through a leaf tests the leaf's
// batches in the reference's order.  Per lane the visited leaves, the cutoffs and hence the hit are those
// of the per-lane walk (trace_closest); the frame stack holds, per level, the wave-uniform far node and lane
// mask plus each lane's (t_split, t_far) pair.
struct PacketArgs {
    const float *cams;        // [frame][4][N]
    const int *order;         // nullptr, or a permutation of the quads (2x2 tiles): expensive (central) rows first
    int tiles_x, tiles_y;
    int quads_x;
    int quads;                // quads per frame
    int nframes;
    int lds_per_wave;         // bytes
    int frame_major;          // work items numbered frame-major instead of quad-rank-major
    float4 *hits_out;         // not null: write the primary hits ([frame][row][x]) instead of shading
    const float *numer;       // not null: [frame][batch][4] plane numerators -(N.o + d) (packet_numerators)
    int n_batches;
};

// The per-lane part of the frame stack is ONE register: bit k of `bothbits` says that the lane entered both
// sides of the branch pushed at level k.  The split distances the reference keeps in its recursion frames
// (t for the far call, t_far to restore) are recomputed from the level's (split, axis) when a frame is
// resumed -- same operands and operations, so the same floats -- which keeps the kernel under 64 VGPRs (a
// per-level float array cost 28 and held the kernel at 4 waves per SIMD).  LDS holds the mailbox and the
// uniform per-level record (far node, far-lane mask, split, axis).
// FEAT = true: the packet walk finds the primary hits (batches only), then every lane shades its hit with the
// general base_color -- lights, shadow rays (per-lane _occludes walks), reflections (per-lane closest-hit walks);
// those secondary walks need the per-lane LDS stack + ray table, placed after the packet's own LDS.
// Plane numerators for the packet kernel: t = -(N.o + d) / (N.dir) (tracer.hpp:560-563) and primary rays share o,
// so the numerator is computed once per (frame, simplex) here -- with exactly the operations the kernel would
// use per lane -- and fetched with the batch's other scalars: a quarter of stage 1's VALU work.
template <int N>
__global__ __launch_bounds__(256) void packet_numerators(NtCompositeDev sc, const float *cams, float *out) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;            // simplex index (batch*4 + lane)
    const long long total = (long long)sc.n_batches * NT_DEV_BATCH;
    if (k >= total) return;
    const float *rec = sc.batch_recs + (size_t)k * sc.rec_stride;
    const float *o = cams + (size_t)blockIdx.y * 4 * N;
    float no = rec[1] * o[0];
#pragma unroll
    for (int j = 1; j < N; ++j) no = no + rec[1 + j] * o[j];
    out[(size_t)blockIdx.y * total + k] = -(no + rec[0]);
}

// Wave-level mailbox of the packet kernel: the walk is wave-uniform, so "which lanes have already tested batch X"
// is one 64-bit mask per batch.  NT_WM direct-mapped entries (tag, mask) replace the per-lane 16-slot mailbox: the
// lookup is one LDS read of a uniform address, and 256 entries remember a ray's whole path through all but the
// largest leaves (the reference keeps every tested primitive in `checked`, tracer.hpp:1166; forgetting one only
// costs a repeated test, which cannot change the hit).  wm_claim returns whether this lane still has to test
// `item` and records that it will.
#define NT_WM 256
__device__ __forceinline__ void wm_reset(int *wm, int lane) {
#pragma unroll
    for (int k = 0; k < NT_WM / 64; ++k) wm[(k * 64 + lane) * 4] = -1;
}
__device__ __forceinline__ bool wm_claim(int *wm, int lane, int item, bool active) {
    int *e = wm + ((item >> 2) & (NT_WM - 1)) * 4;
    const int4 v = *reinterpret_cast<const int4 *>(e);
    const int tag = __builtin_amdgcn_readfirstlane(v.x);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane(v.y);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane(v.z);
    const unsigned long long seen = tag == item ? (((unsigned long long)hi << 32) | lo) : 0ull;
    const bool doit = active && ((seen >> lane) & 1ull) == 0ull;
    const unsigned long long add = __builtin_amdgcn_ballot_w64(doit);
    if (add != 0ull && lane == 0) {
        const unsigned long long now = seen | add;
        *reinterpret_cast<int4 *>(e) = make_int4(item, (int)(unsigned int)(now & 0xffffffffull), (int)(unsigned int)(now >> 32), 0);
    }
    return doit;
}

// A 256-thread block is four independent waves (no barrier) rendering a 2x2 quad of 8x8 tiles of one frame:
// neighbouring rays walk the same leaves, so the four waves share what their scalar loads bring into the CU's
// scalar cache (blocks of unrelated tiles ran ~12 % slower).  Quads are dispatched through a host table: quad
// rows nearest the image centre first (they hold the long walks), row-major within a row.
// (experiment switches: -DNT_PACKET_WAVES4=7 -DNT_PACKET_ATTR=__attribute__((amdgpu_num_sgpr(96))): waves per SIMD asked of the lean
// kernel up to four dimensions, and a scalar-register budget that admits them -- 106 SGPRs admit six 256-thread blocks a CU)
// the packet walk looks at the abort word on every stack pop whose node id has these bits clear: 0 = every pop.  (Measured,
// tools/abort_probe.py, the 120-cell walked strictly at 4096 x 4096, a 10.3 ms frame: with mask 63 a raised flag took 2.4-5.9 ms
// to bring nt_render back -- the long walks through the middle of the scene pop few such nodes -- with 0 it takes 0.4-1.6 ms,
// and the frame, aborted or not, takes the same time: the look is a load nothing else waits for but the branch on it.)
#ifndef NT_ABORT_POLL_MASK
#define NT_ABORT_POLL_MASK 0
#endif
#ifndef NT_PACKET_WAVES4
#define NT_PACKET_WAVES4 6
#endif
#ifndef NT_PACKET_ATTR
#define NT_PACKET_ATTR
#endif
template <int N, int DEPTH, bool FEAT, bool SCAL>
__global__ __launch_bounds__(256, FEAT ? 1 : ((N <= 4 && !SCAL) ? NT_PACKET_WAVES4 : (N <= 7 ? 5 : 4))) NT_PACKET_ATTR void composite_packet(NtCompositeDev sc, NtTarget tg, PacketArgs pa) {
    extern __shared__ float2 lds_raw[];
    if (nt_aborted(tg)) return;                           // (four independent waves: no barrier in this kernel)
    const int lane = (int)threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifdef NT_EXP_TRACE
    const unsigned long long trace_t0 = wall_clock64();
#endif
    const long long slot = (long long)blockIdx.x;
    int rank, frame;
    if (pa.frame_major) { frame = (int)(slot / pa.quads); rank = (int)(slot - (long long)frame * pa.quads); }
    else { rank = (int)(slot / pa.nframes); frame = (int)(slot - (long long)rank * pa.nframes); }
    float2 *lds_wave = reinterpret_cast<float2 *>(reinterpret_cast<char *>(lds_raw) + (size_t)wv * pa.lds_per_wave);
    WaveLds w;              // view used by the shared helpers (mailbox; FEAT: also ray table and per-lane stack)
    w.ray = lds_wave;                                                      // [N][64] float2 (FEAT only)
    w.stack = reinterpret_cast<int *>(lds_wave + (FEAT ? 64 * N : 0));    // [stack_depth][64] (FEAT only)
    w.mbox = w.stack + (FEAT ? 64 * sc.stack_depth : 0);
    int *wm = w.mbox + (FEAT ? 64 * NT_MBOX : 0); // [NT_WM][4] wave mailbox (w.mbox itself: FEAT's per-lane walks)
    int *ustack = wm + NT_WM * 4;                 // [DEPTH][8]: far node, far-lane mask lo, hi, split, axis

    // ---- this wave's tile
    const float *numer = pa.numer ? pa.numer + (size_t)frame * pa.n_batches * NT_DEV_BATCH : nullptr;     // uniform
    const int quad = pa.order ? pa.order[rank] : rank;
    const int qy = quad / pa.quads_x, qx = quad - qy * pa.quads_x;
    const int tx = qx * 2 + (wv & 1), ty = qy * 2 + (wv >> 1);
    if (tx >= pa.tiles_x || ty >= pa.tiles_y) return;
    const int x = tx * 8 + (lane & 7);
    const int row = ty * 8 + (lane >> 3);
    bool valid = x < tg.width && row < tg.row_count;
    const int orow = tg.row_begin + row;
    int y = orow;
    if (tg.band_world > 1) {
        const int band = orow / tg.band_rows;
        y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
    }
    valid = valid && y < tg.height;
    const long long out_off = (long long)frame * tg.frame_stride + (long long)(tg.compact ? orow : y) * tg.pitch + (long long)x * tg.bpp;

    float o[N], d[N], invd[N];
    {
        const float *c = pa.cams + (size_t)frame * 4 * N;
        float right[N], up[N], fwd[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { o[k] = c[k]; right[k] = c[N + k]; up[k] = c[2 * N + k]; fwd[k] = c[3 * N + k]; }
        primary_dir<N>(tg, right, up, fwd, x, y, d);
        // invdir = 1/direction (tracer.hpp:1174); NaN marks direction == 0 (see setup_ray_table)
#pragma unroll
        for (int k = 0; k < N; ++k) invd[k] = d[k] != 0.0f ? 1.0f / d[k] : __int_as_float(0x7fc00000);
    }
    Hit hit;
    hit.dist = FLT_MAX; hit.item = -1; hit.lane = -1;
    const float dist0 = aabb_distance<N>(sc, o, d);
    bool active = valid && dist0 >= 0.0f;
    float t_near = dist0, t_far = FLT_MAX;
    int dirty = 0;
    unsigned int bothbits = 0u;   // bit k: this lane entered BOTH sides of the branch pushed at stack level k
    wm_reset(wm, lane);

    int node = sc.root;      // wave-uniform
    int sp = 0;              // wave-uniform
    for (;;) {
        while (node >= 0) {
            if (sc.prune) active = active && !nt_beyond_hit(hit.dist, t_near);
            if (__builtin_amdgcn_ballot_w64(active) == 0ull) { node = -1; break; }
            const NtNode nd = sc.nodes[node];                // uniform address -> scalar load
            if (nd.axis < 0) {
                // ---- leaf: kd_leaf<Store,true>::intersects (tracer.hpp:977-1086), batches only
                bool improved = false;
                // software pipeline over the leaf's items: the id of item i+1 is fetched with item i's plane
                // records, and its mailbox lookup (LDS) is issued before item i's edge tests, so neither
                // round trip sits on the critical path
                int item = __builtin_amdgcn_readfirstlane(sc.items[nd.left]);
                bool doit = wm_claim(wm, lane, item, active);
                for (int i = 0; i < nd.right; ++i) {
                    const int cur = item;
                    const bool cur_doit = doit;
                    const bool more = i + 1 < nd.right;
                    if (more) item = __builtin_amdgcn_readfirstlane(sc.items[nd.left + i + 1]);
                    if (__builtin_amdgcn_ballot_w64(cur_doit) == 0ull) {
                        doit = false;
                        if (more) doit = wm_claim(wm, lane, item, active);
                        continue;
                    }
                    if (SCAL && (cur & 3) != 0) {
                        // an unbatched triangle or a solid (they follow the batches in every leaf, tracer.hpp:994,1149):
                        // the per-lane tests of leaf_closest, on a record every lane reads from the same address
                        doit = false;
                        if (more) doit = wm_claim(wm, lane, item, active);
                        if (cur_doit) {
                            float t;
                            if ((cur & 3) == 1) {
                                SimplexRec<N> sr;
                                sr.load(sc.tri_recs + (size_t)(cur >> 2) * sc.rec_stride);
                                t = simplex_scalar_form<N>(sr, o, d, hit.dist);
                            } else {
                                float no_[N], nd_[N];
                                t = solid_intersects<N>(sc, cur >> 2, o, d, hit.dist, false, no_, nd_);
                            }
                            if (t != 0.0f) { hit.dist = t; hit.item = cur; hit.lane = -1; improved = true; }
                        }
                        continue;
                    }
                    const float *base = sc.batch_recs + (size_t)(cur >> 2) * NT_DEV_BATCH * sc.rec_stride;
                    // stage 1 for the 4 simplices at once: only d, face_normal, p1 (9 + N-4.. floats) are fetched,
                    // so all plane tests share one scalar-memory round trip
                    float tl[NT_DEV_BATCH];
                    bool ok1[NT_DEV_BATCH];
#pragma unroll
                    for (int l = 0; l < NT_DEV_BATCH; ++l) {
                        const float *rec = base + (size_t)l * sc.rec_stride;
                        float denom = rec[1] * d[0];
#pragma unroll
                        for (int k = 1; k < N; ++k) denom = denom + rec[1 + k] * d[k];
                        float num;
                        if (numer) {
                            num = numer[(size_t)(cur >> 2) * NT_DEV_BATCH + l];
                        } else {
                            float no = rec[1] * o[0];
#pragma unroll
                            for (int k = 1; k < N; ++k) no = no + rec[1 + k] * o[k];
                            num = -(no + rec[0]);
                        }
                        tl[l] = num / denom;
                        ok1[l] = denom != 0.0f && tl[l] >= 0.0f;
                    }
                    doit = false;
                    if (more) doit = wm_claim(wm, lane, item, active);
                    float min_t = hit.dist;
                    int r = -1;
#pragma unroll
                    for (int l = 0; l < NT_DEV_BATCH; ++l) {
                        // stage 2 only if some lane can still accept this simplex (same accept rule as below)
                        const float t = tl[l];
                        if (__builtin_amdgcn_ballot_w64(cur_doit && ok1[l] && t != 0.0f && t < min_t) == 0ull) continue;
                        const float *rec = base + (size_t)l * sc.rec_stride;
                        float pside[N];
#pragma unroll
                        for (int k = 0; k < N; ++k) pside[k] = rec[1 + N + k] - (o[k] + t * d[k]);
                        bool ok = ok1[l];
                        float tot = 0.0f;
#pragma unroll
                        for (int e = 0; e < N - 1; ++e) {
                            const float *en = rec + 1 + 2 * N + e * N;
                            float area = en[0] * pside[0];
#pragma unroll
                            for (int k = 1; k < N; ++k) area = area + en[k] * pside[k];
                            ok = ok && area >= -NT_FUZZ;
                            tot += area;
                        }
                        ok = ok && tot <= (1.0f + NT_FUZZ);
                        if (ok && t != 0.0f && t < min_t) { min_t = t; r = l; }
                    }
                    if (cur_doit && r >= 0) { hit.dist = min_t; hit.item = cur; hit.lane = r; improved = true; }
                }
                if (improved) dirty = sp;
                node = -1;
                break;
            }
            // ---- branch: kd_node_intersection::operator() (tracer.hpp:1189-1240)
            const int axis = __builtin_amdgcn_readfirstlane(nd.axis);      // uniform
            float oa = o[0];                                   // the same in every lane (shared origin)
#pragma unroll
            for (int k = 1; k < N; ++k) oa = axis == k ? o[k] : oa;
            const float inv = pick_uniform<N>(invd, axis);
            const bool gt = __builtin_amdgcn_readfirstlane((int)(oa > nd.split)) != 0;
            const int n_near = gt ? nd.right : nd.left;
            const int n_far = gt ? nd.left : nd.right;
            bool go_near = false, go_far = false, both = false;
            float t = 0.0f;
            if (active) {
                if (inv == inv) {
                    if (oa == nd.split) {
                        // node = direction > 0 ? right : left; with oa == split: near = left, far = right
                        if (inv > 0.0f) go_far = true; else go_near = true;
                    } else {
                        t = (nd.split - oa) * inv;
                        if (t < 0.0f || t > t_far) go_near = true;
                        else if (t < t_near) go_far = true;
                        else both = true;
                    }
                } else {
                    // direction[axis] == 0: node = origin >= split ? right : left
                    const bool to_right = oa >= nd.split;
                    if (to_right == gt) go_near = true; else go_far = true;       // near == right iff gt
                }
            }
            // a `both` lane with no near child continues in far with t_near = t (tracer.hpp:1234-1237);
            // with no far child it returns after near (:1214)
            const bool near_lane = (go_near || both) && n_near >= 0;
            const bool far_after = (go_far || both) && n_far >= 0 && n_near >= 0;      // far AFTER a near subtree
            const unsigned long long m_near = __builtin_amdgcn_ballot_w64(near_lane);
            const unsigned long long m_far = __builtin_amdgcn_ballot_w64(far_after);
            if (m_near != 0ull) {
                if (m_far != 0ull && sp < DEPTH) {
                    if (lane == 0) {
                        ustack[sp * 8 + 0] = n_far;
                        ustack[sp * 8 + 1] = (int)(unsigned int)(m_far & 0xffffffffull);
                        ustack[sp * 8 + 2] = (int)(unsigned int)(m_far >> 32);
                        ustack[sp * 8 + 3] = __float_as_int(nd.split);
                        ustack[sp * 8 + 4] = axis;
                    }
                    bothbits = both ? (bothbits | (1u << sp)) : (bothbits & ~(1u << sp));
                    ++sp;
                }
                if (both && near_lane) t_far = t;
                active = near_lane;
                node = n_near;
            } else {
                // no lane enters near: lanes bound for far go there now
                const bool goes = (go_far || both) && n_far >= 0;
                if (both && goes) t_near = t;
                active = goes;
                node = __builtin_amdgcn_ballot_w64(goes) != 0ull ? n_far : -1;
            }
        }
        // ---- the frame returned: resume the innermost pending far side
        if (sp == 0) break;
        --sp;
        const int far = __builtin_amdgcn_readfirstlane(ustack[sp * 8 + 0]);
        // (abort: a look at the word on every pop -- NT_ABORT_POLL_MASK -- so that a wave on one of the long walks through the
        // middle of a scene does not hold an aborted frame for milliseconds)
        if (tg.abort_word != nullptr && (far & NT_ABORT_POLL_MASK) == 0 && nt_aborted(tg)) return;
        const unsigned long long m = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane(ustack[sp * 8 + 2]) << 32) |
                                     (unsigned int)__builtin_amdgcn_readfirstlane(ustack[sp * 8 + 1]);
        const bool was_both = ((bothbits >> sp) & 1u) != 0u;
        const bool near_hit = sp < dirty;
        if (dirty > sp) dirty = sp;
        bool join = ((m >> lane) & 1ull) != 0ull;
        if (__builtin_amdgcn_ballot_w64(join && was_both) != 0ull) {
            // the split distance of the branch being resumed, recomputed from its (split, axis): same operands,
            // same operations as at the push, hence the same float
            const float psplit = __int_as_float(__builtin_amdgcn_readfirstlane(ustack[sp * 8 + 3]));
            const int paxis = __builtin_amdgcn_readfirstlane(ustack[sp * 8 + 4]);
            float poa = o[0];
#pragma unroll
            for (int k = 1; k < N; ++k) poa = paxis == k ? o[k] : poa;
            const float et = (psplit - poa) * pick_uniform<N>(invd, paxis);
            // t_far of the frame being resumed = the split distance of the innermost pending branch below that this
            // lane entered on both sides (its near subtree is where we are); none: the root's t_far
            const unsigned int below = bothbits & ((1u << sp) - 1u);
            float ef = FLT_MAX;
            if (below != 0u) {
                const int ks = 31 - __clz((int)below);
                const float s2 = __int_as_float(ustack[ks * 8 + 3]);
                const int a2 = ustack[ks * 8 + 4];
                ef = (s2 - pick_lane<N>(o, a2)) * pick_lane<N>(invd, a2);
            }
            if (join && was_both) {                        // a `both` lane: (hit && o_hit.dist <= t) -> return
                if (near_hit && hit.dist <= et) join = false;
                else { t_near = et; t_far = ef; }
            }
        }
        active = join;
        node = far;
    }

#ifdef NT_EXP_TRACE
    if (lane == 0) {       // ablation builds only: per-wave residency record behind the last frame
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(tg.dest + (long long)pa.nframes * tg.frame_stride) +
                                 ((size_t)frame * pa.tiles_x * pa.tiles_y + (size_t)ty * pa.tiles_x + tx) * 4;
        tr[0] = trace_t0;
        tr[1] = wall_clock64();
        tr[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
        tr[3] = __builtin_amdgcn_ballot_w64(hit.item >= 0);
    }
#endif
    bool shade_here = valid;
    if (!FEAT && pa.hits_out) {
        // first pass of a lit scene: the shading pass (composite_kernel<N,true,false>) picks the hits up; rays
        // that hit nothing get their background colour here and are skipped there
        if (valid) pa.hits_out[((long long)frame * tg.row_count + row) * tg.width + x] =
            make_float4(hit.dist, __int_as_float(hit.item), __int_as_float(hit.lane), 0.0f);
        shade_here = valid && hit.item < 0;
    }
    if (shade_here) {
        Color3 c;
        if (FEAT) {
            Stats st = {0, 0, 0, 0, 0, 0, 0, 0};
            c = composite_color<N, true, false>(sc, w, lane, o, d, st, &hit);
        } else {
            c = hit.item >= 0 ? surface_color_lean<N>(sc, hit, o, d) : background_color<N>(sc, d);
        }
        uint8_t *p = tg.dest + out_off;
        if (tg.pack_mode == NT_PACK_WORD32) {
            const uint32_t wd = pack_word32(c.r, c.g, c.b, tg);
            if (tg.bpp == 4 && tg.aligned4) *reinterpret_cast<uint32_t *>(p) = tg.reversed ? wd : bswap32(wd);
            else store_word32_narrow(p, tg, wd, x);
        } else if (tg.pack_mode == NT_PACK_WORD64) {
            store_word64_narrow(p, tg, pack_word64(c.r, c.g, c.b, tg), x);
        } else {
            uint64_t hi, lo;
            pack_pixel(c.r, c.g, c.b, tg, hi, lo);
            store_pixel(p, tg, hi, lo);
        }
    }
}

template <int N>
int launch_composite_fixed(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) {
    NtCameraFixed cf;
    cf.buf = cam.buf;
    cf.dots = cam.dots;
    for (int k = 0; k < 4; ++k) cf.odots[k] = cam.odots[k];
    cf.n = N;
    for (int k = 0; k < 4 * N; ++k) cf.inl[k] = cam.inl[k];
    dim3 grid;
    grid_for(tg, 16, 16, li.nframes, grid);
    const size_t lds = (size_t)4 * 64 * ((size_t)sc.stack_depth * 4 + (size_t)N * 8 + (size_t)NT_MBOX * 4);
    if (lds > 160 * 1024) {
        snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "k-d tree too deep for the LDS traversal stack (depth %d)", sc.stack_depth);
        return -1;
    }
    const bool feat = sc.n_point_lights || sc.n_global_lights || sc.any_reflective || sc.has_scalar_prims;
    hipStream_t s = (hipStream_t)li.stream;
    if (sc.checked) {
        // transparent materials, or the reference's o_hit.normal handling for scenes with Solids: the kernel with the exact
        // `checked` list -- as many blocks as the scratch has lane columns for, striding over the tiles
        const long long tiles = (long long)grid.x * grid.y * grid.z;
        long long blocks = sc.checked_lanes / 256;
        if (blocks > tiles) blocks = tiles;
        if (sc.alias_normals)
            hipLaunchKernelGGL((composite_kernel_t<N, true>), dim3((unsigned)blocks), dim3(256), lds, s, cf, sc, tg, (int)grid.x, (int)grid.y, (int)grid.z);
        else
            hipLaunchKernelGGL((composite_kernel_t<N, false>), dim3((unsigned)blocks), dim3(256), lds, s, cf, sc, tg, (int)grid.x, (int)grid.y, (int)grid.z);
        return 0;
    }
    if (!sc.all_opaque) {
        snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "internal: transparent scene without the checked-list scratch");
        return -1;
    }
    // scenes with unbatched triangles or solids take the packet walk only as the first of two passes (their hits
    // are shaded by the general kernel)
    const bool two_pass_ok = feat && li.hit_buf && li.hit_frames > 0;
    if ((!sc.has_scalar_prims || two_pass_ok) && !sc.stats && !tg.colors_out && li.persist_cams && li.kernel_choice == 0 && sc.stack_depth <= 32) {
        // packet kernel: one wave per 8x8 tile, wave-uniform tree walk for the primary rays
        PacketArgs pk;
        pk.cams = li.persist_cams;
        pk.tiles_x = (tg.width + 7) / 8;
        pk.tiles_y = (tg.row_count + 7) / 8;
        pk.quads_x = (pk.tiles_x + 1) / 2;
        pk.quads = pk.quads_x * ((pk.tiles_y + 1) / 2);
        pk.nframes = li.nframes;
        pk.order = li.tile_order;
        pk.frame_major = getenv("NTRACER_FRAME_MAJOR") ? atoi(getenv("NTRACER_FRAME_MAJOR")) : 1;
        pk.hits_out = nullptr;
        pk.numer = nullptr;
        pk.n_batches = sc.n_batches;
        const bool two_pass = feat && li.hit_buf && li.hit_frames > 0;
        const bool single_feat = feat && !two_pass;
        // frames per launch: what the scratch buffers (primary hits, plane numerators) hold
        int chunk = li.nframes;
        if (two_pass && li.hit_frames < chunk) chunk = li.hit_frames;
        if (li.numer_buf && li.numer_frames > 0 && li.numer_frames < chunk) chunk = li.numer_frames;
        const size_t lds_lean = (size_t)NT_WM * 16 + (size_t)32 * 32;
        const size_t lds_feat = (size_t)64 * ((size_t)N * 8 + (size_t)sc.stack_depth * 4 + NT_MBOX * 4) + NT_WM * 16 + (size_t)32 * 32;
        for (int f0 = 0; f0 < li.nframes; f0 += chunk) {
            const int cnt = li.nframes - f0 < chunk ? li.nframes - f0 : chunk;
            NtTarget t2 = tg;
            t2.dest = tg.dest + (long long)f0 * tg.frame_stride;
            pk.cams = li.persist_cams + (size_t)f0 * 4 * N;
            pk.nframes = cnt;
            if (li.numer_buf && li.numer_frames > 0 && sc.n_batches > 0) {
                // -(N.o + d) of every simplex for every camera of the chunk: the same for all rays of a frame
                const long long total = (long long)sc.n_batches * NT_DEV_BATCH;
                hipLaunchKernelGGL((packet_numerators<N>), dim3((unsigned)((total + 255) / 256), (unsigned)cnt), dim3(256), 0, s,
                                   sc, pk.cams, li.numer_buf);
                pk.numer = li.numer_buf;
            }
            const dim3 pgrid((unsigned)((long long)pk.quads * cnt));
            if (single_feat) {
                pk.lds_per_wave = (int)lds_feat;
                hipLaunchKernelGGL((composite_packet<N, 32, true, false>), pgrid, dim3(256), (size_t)4 * pk.lds_per_wave, s, sc, t2, pk);
                continue;
            }
            // Lit scenes in two passes: the lean packet kernel (47 VGPRs, 6 waves/SIMD) finds the primary hits, then
            // the per-lane shading kernel (250 VGPRs: lights, shadow and reflection rays) starts from them.  One
            // kernel doing both ran its primary walk at the shading code's occupancy (1 wave/SIMD).
            pk.lds_per_wave = (int)lds_lean;
            pk.hits_out = two_pass ? (float4 *)li.hit_buf : nullptr;
            if (sc.has_scalar_prims)
                hipLaunchKernelGGL((composite_packet<N, 32, false, true>), pgrid, dim3(256), (size_t)4 * pk.lds_per_wave, s, sc, t2, pk);
            else
                hipLaunchKernelGGL((composite_packet<N, 32, false, false>), pgrid, dim3(256), (size_t)4 * pk.lds_per_wave, s, sc, t2, pk);
            if (two_pass) {
                t2.hits = li.hit_buf;
                NtCameraFixed c2 = cf;
                c2.buf = li.persist_cams + (size_t)f0 * 4 * N;
                dim3 g2;
                grid_for(t2, 16, 16, cnt, g2);
                // (scenes made of batches alone: the instantiation without unbatched triangles and solids in its leaf loops)
                if (sc.has_scalar_prims) hipLaunchKernelGGL((composite_kernel<N, true, false>), g2, dim3(256), lds, s, c2, sc, t2);
                else hipLaunchKernelGGL((composite_kernel<N, true, false, false>), g2, dim3(256), lds, s, c2, sc, t2);
            }
        }
        return 0;
    }
    if (!feat && !sc.stats && !tg.colors_out && li.persist_counter) {
        // persistent waves + ray refill
        PersistArgs pa;
        pa.cams = li.persist_cams;
        pa.counter = (unsigned long long *)li.persist_counter;
        pa.tiles_x = (tg.width + 7) / 8;
        pa.tiles_per_frame = pa.tiles_x * ((tg.row_count + 7) / 8);
        pa.total = (long long)li.nframes * pa.tiles_per_frame * 64;
        const size_t lds_block = lds;
        int blocks_per_cu = (int)(160 * 1024 / lds_block);
        if (blocks_per_cu > 8) blocks_per_cu = 8;
        if (blocks_per_cu < 1) blocks_per_cu = 1;
        long long want_blocks = (pa.total + 255) / 256;
        int nblocks = li.cu_count * blocks_per_cu;
        if ((long long)nblocks > want_blocks) nblocks = (int)want_blocks;
        if (nblocks < 1) nblocks = 1;
        hipLaunchKernelGGL((composite_persistent<N>), dim3((unsigned)nblocks), dim3(256), lds, s, sc, tg, pa);
        return 0;
    }
    if (sc.stats) hipLaunchKernelGGL((composite_kernel<N, true, true>), grid, dim3(256), lds, s, cf, sc, tg);
    else if (feat && !sc.has_scalar_prims) hipLaunchKernelGGL((composite_kernel<N, true, false, false>), grid, dim3(256), lds, s, cf, sc, tg);
    else if (feat) hipLaunchKernelGGL((composite_kernel<N, true, false>), grid, dim3(256), lds, s, cf, sc, tg);
    else hipLaunchKernelGGL((composite_kernel<N, false, false>), grid, dim3(256), lds, s, cf, sc, tg);
    return 0;
}

}  // namespace
