// nt_box.hpp -- BoxScene kernels for compile-time N (fixed_geometry.hpp -> registers):
//   box_cull_kernel<N> -> box_kernel<N,PLAIN,ROWS> -> box_redo_kernel<N>      box_scene::calculate_color (src/tracer.hpp:101-152)
// fused with process_pixel's conversion and packing (nt_pixel.hpp), so the only HBM traffic of a frame is the packed
// framebuffer.  Instantiated per N by nt_inst_box.hip.
#pragma once
#include <type_traits>
#include "nt_pixel.hpp"

namespace {

// box_scene::calculate_color + hypercube_intersects (tracer.hpp:101-152).
//
// The reference tries the entry face of every axis i in ascending order; face i is the hit when
// dist_i = (s_i - o_i)/d_i > 0 and |o_j + d_j*dist_i| <= 1+FUZZ for all j != i.  That predicate is evaluated
// here bit for bit, but only for the faces that can satisfy it:
//   1. a ray whose distance from the centre exceeds the cube's circumradius (0.1 % margin) satisfies it for
//      no face: such waves skip everything;
//   2. let K be the candidate face reached LAST (largest dist; found by cross-multiplication, no division).
//      A candidate i reached earlier than K by more than a sliver is still outside slab K at t = dist_i:
//      |o_K + d_K*dist_i| = 1 + |d_K|*(dist_K - dist_i) > 1 + FUZZ, so it fails the reference's own j = K check.
//      With a = |s - o|, b = |d| (dist = a/b), "more than a sliver" is  a_i*b_K < b_i*(a_K - mu),
//      mu = 1e-4*(1+|o_K|) -- ~100x the rounding error of the quantities compared and of the reference's
//      check.  Only the remaining near-ties (normally just K) get the division and the N-1 checks, in
//      ascending order, exactly as the reference computes them.
// Step 1 of the pruning above, on the UNNORMALISED direction v (|v|^2 = sq): the ray passes within the
// circumradius unless |o|^2 - (o.v)^2/|v|^2 > rad2.  Conservative (0.1 % on the radius, 1e-4 on the product),
// not bit-exact -- it only decides whether the exact predicate is evaluated at all.  Waves in which no lane
// may hit never normalise more than dir[0], the one component the background colour needs: that saves N-1 of
// the N IEEE divisions for ~85 % of the rays of the 6-D benchmark frames.
// dots: |o|^2, o.right, o.up, o.forward (host); v = forward + right*sx - up*sy, so o.v follows from three of them
__device__ __forceinline__ bool box_may_hit(int n, const float *dots, float sx, float sy, float sq) {
    const float osq = dots[0];
    const float ov = fmaf(-dots[2], sy, fmaf(dots[1], sx, dots[3]));
    const float rad2 = (float)n * (1.0f + NT_FUZZ) * (1.0f + NT_FUZZ) * 1.001f;
#ifdef NT_EXP_SKIP_SLABS
    return false;
#else
    return !((osq - rad2) * sq > ov * ov * 1.0001f);         // NaN -> maybe
#endif
}

__device__ __forceinline__ void box_background(float in, float &r, float &g, float &b) {
    // miss: i = dir[0]; i > 0 ? (i,i,i) : (0,-i,-i)   (tracer.hpp:109-113)
    if (in > 0.0f) { r = in; g = in; b = in; }
    else { r = 0.0f; g = -in; b = -in; }
}

template <int N>
__device__ __forceinline__ void box_color(const float (&o)[N], const float (&dir)[N], bool maybe, float &r, float &g, float &b) {
    bool done = false;     // a face passed the slab test (hit, or dist >= cutoff)
    float shade = 0.0f;

    {
        float num[N];
        bool pre[N];
        // candidates: dist > 0 needs a non-zero numerator with the sign of d_i; track the last-reached one
        float aK = 0.0f, bK = 1.0f, oK = 0.0f;
        bool any = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float di = dir[i];
            const float s = di < 0.0f ? 1.0f : -1.0f;
            num[i] = s - o[i];
            pre[i] = maybe && ((num[i] > 0.0f && di > 0.0f) || (num[i] < 0.0f && di < 0.0f));
            const float a = fabsf(num[i]), bb = fabsf(di);
            // a/bb > aK/bK  <=>  a*bK > aK*bb   (all positive)
            if (pre[i] && (!any || a * bK > aK * bb)) { aK = a; bK = bb; oK = o[i]; any = true; }
        }
        const float mu = 1e-4f * (1.0f + fabsf(oK));
        const float aKm = (aK - mu) * (1.0f - 1e-6f);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            // near-tie with the last-reached face (always true for that face itself); NaN-safe: !(x < y)
            const bool tie = pre[i] && !done && !(fabsf(num[i]) * bK < fabsf(dir[i]) * aKm);
            if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
                const float di = dir[i];
                const float dist = num[i] / di;
                bool ok = tie && dist > 0.0f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (j != i) {
                        const float p = dir[j] * dist + o[j];
                        ok = ok && !(fabsf(p) > (1.0f + NT_FUZZ));
                    }
                }
                if (ok) {
                    done = true;
                    // `if(dist >= cutoff) return 0` with cutoff = FLT_MAX (tracer.hpp:142): a miss
                    if (dist >= FLT_MAX) shade = -1.0f;
                    else {
                        const float sine = di * (di < 0.0f ? 1.0f : -1.0f);   // dot(dir, s*e_i)
                        shade = sine <= 0.0f ? -sine : 0.0f;
                    }
                }
            }
        }
    }
    if (done && shade >= 0.0f) {
        r = shade * 1.0f;
        g = shade * 0.5f;
        b = shade * 0.5f;
    } else {
        box_background(dir[0], r, g, b);
    }
}

// m of box_classify / box_resolve / box_cull_kernel, per unit of 1 + max|o_j|.  What it has to dominate:
// ROUNDING_FUZZ (1.2e-6), the reference's own rounding in p_j (<= (5|o_j| + 6)*2^-24) and the v_rcp_f32 arithmetic
// here (~2^-22*(1 + |o_j|)): under 2e-6*(1 + max|o_j|) together, 5x below this.  (Rounds 1 and most of 2 ran with 3e-5; the
// narrower band sends a third fewer stretches to box_redo_kernel.  Builds with -DNT_BOX_MARGIN=2e-6f, the bound itself, and
// 5e-6f render the 440 full frames of tools/box_soak.py byte for byte like the oracle; at 5e-7f frames start to differ.)
// box_tile_kernel renders everything itself -- no box_redo_kernel after it -- up to this dimension (packed RGB; fp32x3: always)
#ifndef NT_BOX_INLINE_MAX_N
#define NT_BOX_INLINE_MAX_N 8
#endif
// ... which is as far as the codes wave works out tie sets (bits 0..9 and 10..19 of a dword: at most 10)
#ifndef NT_BOX_SETS_MAX_N
#define NT_BOX_SETS_MAX_N NT_BOX_INLINE_MAX_N
#endif
#ifndef NT_BOX_MARGIN
#define NT_BOX_MARGIN 1e-5f
#endif
// Which rays need the exact evaluation at all?  Everything below works on the UNNORMALISED direction v
// (p_j(tau) = o_j + v_j*tau; the reference's dist is tau*|v|), with reciprocals from v_rcp_f32, and sorts a lane
// that may hit into one of three classes.  m = NT_BOX_MARGIN*(1 + max|o_j|) dominates the sum of ROUNDING_FUZZ and
// every rounding error involved (see NT_BOX_MARGIN).
//   miss      the ray (tau > 0) stays outside the cube grown to 1+m.  Every point the reference accepts has
//             |p_i| = 1, |p_j| <= 1+FUZZ at a dist > 0, so the reference finds no face either.
//   hit at K  K = the entry face reached last, at tau_K; tau_K is clearly positive; at tau_K every other
//             coordinate is inside 1-m/2 (the reference's test for K passes); and every other entry plane is
//             crossed while p_K is still outside 1+m ((tau_K - tau_i)*|v_K| > m), so no face before K in the
//             reference's ascending order can pass its j = K check.  The reference returns face K: shade |d_K|.
//   unclear   anything else (edges, grazing rays, origins on or inside the cube, NaN): the wave takes the exact,
//             reference-ordered evaluation in box_color.
// x = the component the colour is made of: v_K for a hit, v_0 for the background.
template <int N>
__device__ __forceinline__ void box_classify(const float (&o)[N], const float (&v)[N], float m, bool maybe, bool &hit, bool &unclear,
                                             float &x, float (&near)[N], float &tn, float &vK) {
    float tn2 = -INFINITY;                                 // tn, tn2: last and second-to-last entry, unit cube
    tn = -INFINITY;
    vK = 0.0f;
    float tnp = -INFINITY, tfp = INFINITY;                 // last entry / first exit, cube grown by m
    const float m1p = 1.0f + m;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // slab j: (+-1 - o_j)/v_j = c -+ |1/v_j| with c = -o_j/v_j; grown by m: c -+ (1 + m)|1/v_j|.  (v_j = 0: inf - inf
        // is a NaN, which drops out of fmaxf / fminf -- right for an origin inside the slab; one outside it goes
        // undetected here and ends up unclear, but box_stretch_code culls such stretches before anyone looks at a ray)
        const float inv = __builtin_amdgcn_rcpf(v[j]);
        const float c0 = -o[j] * inv;
        const float nr = c0 - fabsf(inv);
        near[j] = nr;
        tnp = fmaxf(tnp, fmaf(-fabsf(inv), m1p, c0));
        tfp = fminf(tfp, fmaf(fabsf(inv), m1p, c0));
        const bool later = nr > tn;
        tn2 = __builtin_amdgcn_fmed3f(tn, tn2, nr);
        vK = later ? v[j] : vK;
        tn = fmaxf(tn, nr);
    }
    const bool miss = tnp > tfp || tfp < 0.0f;
    const float c = 1.0f - m;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) sum = sum + fmaxf(fabsf(fmaf(v[j], tn, o[j])), c);
    // face K itself contributes 1 - c = m; the others nothing unless they are within m of their planes
    const bool inside = sum - (float)N * c <= 1.5f * m;
    const bool sole = (tn - tn2) * fabsf(vK) > m;
    const bool front = tn > 1e-3f && tn < 1e30f;
    hit = maybe && !miss && inside && sole && front;
    unclear = maybe && !miss && !hit;
#ifdef NT_EXP_NOUNCLEAR
    unclear = false;
#endif
    x = hit ? vK : v[0];
}

// The part of box_classify that box_resolve needs: entry times into the unit cube, the last of them, its axis.
template <int N>
__device__ __forceinline__ void box_entries(const float (&o)[N], const float (&v)[N], float (&near)[N], float &tn, float &vK) {
    tn = -INFINITY;
    vK = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float inv = __builtin_amdgcn_rcpf(v[j]);
        const float nr = fmaf(-o[j], inv, -fabsf(inv));          // (+-1 - o_j)/v_j, the earlier of the two
        near[j] = nr;
        vK = nr > tn ? v[j] : vK;
        tn = fmaxf(tn, nr);
    }
}

// The unclear lanes of a wave, resolved with the reference's own arithmetic -- but only the part of it that can
// matter.  With near[], tn, vK from box_classify (approximate; the margins absorb that):
//   T = { i : (tn - near_i)*|v_K| <= m }   the faces that can still be the reference's answer: any other face is
//       entered while p_K is outside 1+m and fails its j = K check (box_classify);
//   C = { j : |p_j(tn)| + |v_j|*m/|v_K| > 1 - m/2 }   the coordinates whose test some face of T could fail: every
//       face of T is entered within m/|v_K| of tn, so a coordinate outside C is inside 1-m/2 at all of them.
//       T is a subset of C (p_i(tn) is within |v_i|*m/|v_K| of +-1 for i in T).
// The faces of T are tried in ascending order exactly as hypercube_intersects does (tracer.hpp:126-152): dist from the
// IEEE quotient, then d_j*dist + o_j against 1+FUZZ for the j in C; d_j = v_j/len is computed for the axes of C only.
// A lane whose tn is not a usable number gets T = C = everything, i.e. the reference's full loop.
template <int N>
__device__ __forceinline__ void box_resolve(const float (&o)[N], const float (&v)[N], float len, float m, const float (&near)[N], float tn,
                                            float vK, bool unclear, bool &hit, float &x) {
    const float aK = fabsf(vK);
    const float slack = m * __builtin_amdgcn_rcpf(aK), lim = 1.0f - 0.5f * m;
    bool inT[N], inC[N];
    float d[N];
    uint32_t anyC = 0u;                       // coordinates in some lane's C (uniform): the others are not looked at below
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const bool early = (tn - near[j]) * aK > m;                                     // false for a NaN
        const bool robust = fmaf(fabsf(v[j]), slack, fabsf(fmaf(v[j], tn, o[j]))) <= lim;  // false for a NaN
        inT[j] = unclear && !early;
        inC[j] = unclear && !(robust && early);
        d[j] = 0.0f;
        if (__builtin_amdgcn_ballot_w64(inC[j]) != 0ull) {
            d[j] = v[j] / len;
            anyC |= 1u << j;
        }
    }
    bool done = false, found = false;
    float xs = v[0];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool cand = inT[i] && !done;
        if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
            const float di = d[i];
            const float dist = ((di < 0.0f ? 1.0f : -1.0f) - o[i]) / di;
            bool ok = cand && di != 0.0f && dist > 0.0f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j != i && ((anyC >> j) & 1u) != 0u) {
                    const float p = d[j] * dist + o[j];
                    ok = ok && !(inC[j] && fabsf(p) > (1.0f + NT_FUZZ));
                }
            }
            if (ok) {
                done = true;
                // `if(dist >= cutoff) return 0` with cutoff = FLT_MAX (tracer.hpp:142): a miss
                if (!(dist >= FLT_MAX)) { found = true; xs = v[i]; }
            }
        }
    }
    if (unclear) {
        hit = found;
        x = xs;
    }
}

// a pointer into global memory every lane agrees on, pinned to scalar registers (and so kept apart from the per-lane
// offset added to it): base of a global_store with an SGPR address
typedef __attribute__((address_space(1))) uint8_t *nt_gptr;
__device__ __forceinline__ nt_gptr uniform_ptr(uint8_t *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (nt_gptr)(((unsigned long long)hi << 32) | lo);
}
#define NT_G32(p) (*(__attribute__((address_space(1))) uint32_t *)(p))

// Three fp32 channels that are plain components (tg.plain_f32, 12-byte pixels on 4-byte-aligned rows), BoxScene colours
// (g == b): the bytes emit_pixel writes for this layout -- clamp, big-endian floats or the reversed pixel -- as one
// 12-byte store.
template <typename P>
__device__ __forceinline__ void emit_f32x3_at(const NtTarget &tg, P p, float r, float gb) {
    r = r > 0.0f ? r : 0.0f;          // simd::clamp, as in channel_value
    r = r < 1.0f ? r : 1.0f;
    gb = gb > 0.0f ? gb : 0.0f;
    gb = gb < 1.0f ? gb : 1.0f;
    uint32_t vr = __float_as_uint(r), vgb = __float_as_uint(gb);
    uint3 w;
    if (tg.plain_f32[0] == 0 && tg.plain_f32[1] == 1 && tg.plain_f32[2] == 2 && !tg.reversed) {       // (uniform) R, G, B in order
        w.x = bswap32(vr);
        w.y = bswap32(vgb);
        w.z = w.y;
    } else {
        if (!tg.reversed) { vr = bswap32(vr); vgb = bswap32(vgb); }
        w.x = tg.plain_f32[tg.reversed ? 2 : 0] == 0 ? vr : vgb;
        w.y = tg.plain_f32[1] == 0 ? vr : vgb;
        w.z = tg.plain_f32[tg.reversed ? 0 : 2] == 0 ? vr : vgb;
    }
    if constexpr (std::is_same<P, nt_gptr>::value) {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        u32x3 v;
        v.x = w.x; v.y = w.y; v.z = w.z;
        *(__attribute__((address_space(1))) u32x3 *)(p) = v;
    } else {
        *reinterpret_cast<uint3 *>(p) = w;
    }
}
__device__ __forceinline__ void emit_f32x3(const NtTarget &tg, long long offset, float r, float gb) { emit_f32x3_at(tg, tg.dest + offset, r, gb); }

// sqrtf(x) for x in [2^-96, 2^96): hipcc's correctly rounded square root is v_sqrt_f32 followed by a choice among the
// result and its two neighbours (two fma residuals), wrapped in a rescaling for x < 2^-96 and a pass-through for 0 and
// infinity; inside that range the wrapping does nothing, and this is the rest -- the same operations, hence the same float.
__device__ __forceinline__ float sqrt_in_range(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = fmaf(-s_dn, s, x), vs = fmaf(-s_up, s, x);
    float r = vp <= 0.0f ? s_dn : s;
    r = vs > 0.0f ? s_up : r;
    return r;
}
// ... for a whole wave: the plain sqrtf unless every lane is inside the range (NaN included in "outside")
__device__ __forceinline__ float sqrt_wave(float x) {
    if (__builtin_amdgcn_ballot_w64(!(x >= 0x1p-96f && x < 0x1p96f)) == 0ull) return sqrt_in_range(x);
    return sqrtf(x);
}

// One pixel of BoxScene from the unnormalised direction `dir` (|dir|^2 = sq), sx / sy as in the ray source.
// PLAIN: the format is known to be plain_rgb with at most 10 bits per channel (the launcher checks).
// DEFER: a wave with an unclear lane writes nothing and returns false (the caller hands the stretch to box_redo_kernel),
// which keeps the resolving code -- and its registers -- out of the kernel every other wave runs.
// REDO (box_redo_kernel): no sorting into clear and unclear -- box_resolve is complete by itself (a clear hit is T = C =
// {K}; a clear miss fails at a coordinate of C), and in a stretch that is here because of its unclear lanes the
// sorting of the others saves nothing.
template <int N, bool PLAIN, bool DEFER = false, bool REDO = false, bool F32 = false>
__device__ __forceinline__ bool box_pixel(const NtTarget &tg, const PixelRef &pr, const float (&org)[N], float (&dir)[N], float sq,
                                          const float (&dots)[4], float sx, float sy, float margin, bool rowhit = true, int face = -1,
                                          uint32_t sets = 0u, bool noclass = false) {
    // noclass (wave-uniform; box_tile_kernel where it has no second kernel: F32, or N <= 8): a near-tie stretch (code 14) --
    // as in box_redo_kernel, no sorting into clear and unclear first
    constexpr bool INL = (F32 || PLAIN) && !REDO && !DEFER;
    const bool redo = REDO || (INL && noclass);
    // rowhit (wave-uniform): the culling bit of this 64-pixel stretch of the row, see box_cull_kernel
    // face (wave-uniform) >= 0: every ray of the stretch is known to hit that face (a one-face row of box_tile_kernel whose
    // cheap quantisation came too close to a rounding boundary): nothing to sort, only the exact colour is wanted
    // (DEFER: the callers pass the code of the stretch as rowhit -- a stretch that survived the codes wave is next to the
    // cube, where the circumsphere test rarely spares a wave the classification and costs eight instructions every time)
    const bool maybe = redo || (rowhit && (DEFER || INL || box_may_hit(N, dots, sx, sy, sq)));
    float r, g, b;
    bool hit = false, unclear = false;
    float x = dir[0];
#ifndef NT_EXP_NOCLASSIFY
    float near[N], tn = 0.0f, vK = 0.0f;
    if (face >= 0) {
        hit = true;
#pragma unroll
        for (int j = 1; j < N; ++j) x = face == j ? dir[j] : x;
    } else if ((REDO || INL) && redo && (sets & 0x80000000u) != 0u) {
        // sets (wave-uniform, box_redo_kernel): T and C of the whole stretch from the codes wave (box_stretch_code) -- the
        // reference's arithmetic on them, as in box_resolve, without the entry times
        const float len = sqrt_wave(sq);
        float d[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            d[j] = 0.0f;
            if ((sets >> (10 + j)) & 1u) d[j] = dir[j] / len;
        }
        bool done = false;
        x = dir[0];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (((sets >> i) & 1u) != 0u && __builtin_amdgcn_ballot_w64(!done) != 0ull) {
                const float di = d[i];
                const float dist = ((di < 0.0f ? 1.0f : -1.0f) - org[i]) / di;
                bool ok = !done && di != 0.0f && dist > 0.0f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (j != i && ((sets >> (10 + j)) & 1u) != 0u) {
                        const float p = d[j] * dist + org[j];
                        ok = ok && !(fabsf(p) > (1.0f + NT_FUZZ));
                    }
                }
                if (ok) {
                    done = true;
                    // `if(dist >= cutoff) return 0` with cutoff = FLT_MAX (tracer.hpp:142): a miss
                    if (!(dist >= FLT_MAX)) { hit = true; x = dir[i]; }
                }
            }
        }
    } else if (redo) {
        box_entries<N>(org, dir, near, tn, vK);
        unclear = true;
    } else if (__builtin_amdgcn_ballot_w64(maybe) != 0ull) {
        box_classify<N>(org, dir, margin, maybe, hit, unclear, x, near, tn, vK);
    }
    // unclear lanes: the reference's arithmetic on the faces and coordinates still in question; a lane without a
    // usable entry time (origin on or inside the cube, NaN) keeps the whole wave on box_color's full evaluation
    if (DEFER) {
        if (__builtin_amdgcn_ballot_w64(unclear) != 0ull) return false;
    } else {
        const bool hard = unclear && !(tn > 1e-3f && tn < 1e30f);
        if (__builtin_amdgcn_ballot_w64(unclear) != 0ull && __builtin_amdgcn_ballot_w64(hard) == 0ull) {
            box_resolve<N>(org, dir, sqrt_wave(sq), margin, near, tn, vK, unclear, hit, x);
            unclear = false;
        }
    }
#else
    unclear = maybe;
#endif
    if (__builtin_amdgcn_ballot_w64(unclear) == 0ull) {
        // every lane's colour is |x|/len times (1,.5,.5) (hit) or (1,1,1) / (0,1,1) (background, by the sign of x)
        if (PLAIN || (plain_rgb(tg) && tg.plain_bits <= 10u)) {
            // Only round(value * maxval) is stored.  |x| * rsq(sq) is within 3*2^-23 of the reference's twice-rounded
            // |x/len| (v_rsq_f32: 1 ulp; two multiplications here; sqrt and division there), so whenever
            // t = that * maxval keeps 2^-20*(1+t) clear of every k + 1/2 the two round to the same integer: no
            // sqrt, no division.  (A hit also stores round(t/2).)  A wave with a lane inside a guard band (a few
            // per cent of them at 8 bits), or with a NaN / infinity, takes the exact division below.
            const float maxv = (float)tg.plain_maxval;
            const float t = (fabsf(x) * __builtin_amdgcn_rsqf(sq)) * maxv;
            const float tgb = hit ? t * 0.5f : t;
            const bool clear_gb = fabsf(__builtin_amdgcn_fractf(tgb) - 0.5f) > fmaf(tgb, 0x1p-20f, 0x1p-20f);
            const bool clear_r = fabsf(__builtin_amdgcn_fractf(t) - 0.5f) > fmaf(t, 0x1p-20f, 0x1p-20f);
            if (__builtin_amdgcn_ballot_w64(!(clear_gb && (clear_r || !hit))) == 0ull) {
                if (tg.plain_sel != 0u) {
                    // 8-bit fields: t + 2^23 holds round(t) in its low mantissa byte (see box_tile_kernel), the byte v_perm_b32 picks
                    const uint32_t q8gb = __float_as_uint(tgb + 8388608.0f), q8r = __float_as_uint(t + 8388608.0f);
                    *reinterpret_cast<uint32_t *>(tg.dest + pr.offset) = __builtin_amdgcn_perm((hit || x > 0.0f) ? q8r : 0u, q8gb, tg.plain_sel);
                    return true;
                }
                uint32_t qgb = (uint32_t)(tgb + 0.5f), qr = (uint32_t)(t + 0.5f);
                qgb = qgb < tg.plain_maxval ? qgb : tg.plain_maxval;    // the value may round to just above 1: clamped
                qr = qr < tg.plain_maxval ? qr : tg.plain_maxval;
                emit_plain(tg, pr, (hit || x > 0.0f) ? qr : 0u, qgb);
                return true;
            }
        }
        const float in = x / sqrtf(sq);
        if (hit) {
            // sine = d_K * (-sign d_K) <= 0, shade = -sine (tracer.hpp:105-107)
            const float shade = fabsf(in);
            r = shade * 1.0f;
            g = shade * 0.5f;
            b = shade * 0.5f;
        } else {
            box_background(in, r, g, b);
        }
    } else if (!DEFER) {
        const float len = sqrtf(sq);
#pragma unroll
        for (int j = 0; j < N; ++j) dir[j] = dir[j] / len;
        box_color<N>(org, dir, maybe, r, g, b);
    } else {
        return false;
    }
    if (F32) {
        emit_f32x3(tg, pr.offset, r, g);                                        // g == b
        return true;
    }
    if (PLAIN || plain_rgb(tg)) {
        emit_plain(tg, pr, plain_quantize(tg, r), plain_quantize(tg, g));       // g == b
        return true;
    }
    emit_pixel(tg, pr, r, g, b);
    return true;
}

// A lane renders ROWS pixels of one image column (a block: 64 columns x 4*ROWS rows; a wave still writes 64 consecutive
// pixels of a row at a time): forward + right*sx and the wave's set-up are shared by all of them.  ROWS = BoxRows<N>
// (8), or 16 for the packed-RGB kernel in large launches; probe mode (listed pixels) is one pixel per lane.
#ifndef NT_BOXROWS
#define NT_BOXROWS 8
#endif
template <int N> struct BoxRows { static constexpr int value = N <= NT_DEV_MAX_FIXED_BOX ? NT_BOXROWS : 1; };
template <int N, bool PLAIN, int ROWS = BoxRows<N>::value>
__global__ __launch_bounds__(256) void box_kernel(NtCameraFixed cam, NtTarget tg) {
    const int tid = (int)threadIdx.x;
    float org[N], right[N], up[N], fwd[N], dir[N];
    load_camera<N>(cam, org, right, up, fwd);
    float margin = fabsf(org[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) margin = fmaxf(margin, fabsf(org[j]));
    margin = NT_BOX_MARGIN * (1.0f + margin);
    float dots[4];
    if (cam.buf) {
        const float *dp = cam.dots + (size_t)blockIdx.z * 4;
        dots[0] = dp[0]; dots[1] = dp[1]; dots[2] = dp[2]; dots[3] = dp[3];
    } else {
        dots[0] = cam.odots[0]; dots[1] = cam.odots[1]; dots[2] = cam.odots[2]; dots[3] = cam.odots[3];
    }
    if ((!PLAIN && tg.colors_out) || ROWS == 1) {
        // one pixel per lane: probe mode (listed pixels)
        const PixelRef pr = locate_pixel<64, 4>(tg, tid & 63, tid >> 6, tid);
        if (!pr.valid) return;
        // flat_origin_ray_source::operator() (tracer.hpp:71-75), as primary_dir, with the normalisation split off
        const float sx = tg.fovI * ((float)pr.x - tg.half_w);
        const float sy = tg.fovI * ((float)pr.y - tg.half_h);
#pragma unroll
        for (int j = 0; j < N; ++j) dir[j] = (fwd[j] + right[j] * sx) - up[j] * sy;
        float sq = dir[0] * dir[0];
#pragma unroll
        for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
        box_pixel<N, PLAIN>(tg, pr, org, dir, sq, dots, sx, sy, margin);
        return;
    }
    // the wave's number as a scalar: everything that depends on the row alone stays on the scalar unit
    constexpr int R = ROWS;
    const int row0 = ((int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(tid >> 6)) * R;
    if (PLAIN) {
        // ---- packed RGB, at most 10 bits a channel: the lean loop ----
        if (row0 >= tg.row_count) return;             // (also keeps the row-code reads below inside the table's padding)
        // Row bookkeeping is done once, one row per lane (lane l <-> row row0 + l), and read back with v_readlane:
        // sy, up[0]*sy, the row's byte offset, whether the row exists.  Every lane stays active for that -- lanes past
        // the right edge redo the last pixel (the same dword, the same value) instead of leaving.
        const int lane = tid & 63;
        const int lorow = tg.row_begin + row0 + lane;
        int ly = lorow;
        if (tg.band_world > 1) {
            const int band = lorow / tg.band_rows;
            ly = (band * tg.band_world + tg.band_rank) * tg.band_rows + (lorow - band * tg.band_rows);
        }
        const uint32_t valid = (uint32_t)__builtin_amdgcn_ballot_w64(lane < R && row0 + lane < tg.row_count && ly < tg.height);
        const float v_sy = tg.fovI * ((float)ly - tg.half_h);
        const float v_us0 = up[0] * v_sy;
        const long long v_off = (long long)blockIdx.z * tg.frame_stride + (long long)(tg.compact ? lorow : ly) * tg.pitch;
        const int v_off_lo = (int)v_off, v_off_hi = (int)(v_off >> 32);
        int x = (int)blockIdx.x * 64 + lane;
        x = x < tg.width ? x : tg.width - 1;
        const long long xoff = (long long)x * tg.bpp;
        const float sx = tg.fovI * ((float)x - tg.half_w);
        float base[N];
#pragma unroll
        for (int j = 0; j < N; ++j) base[j] = fwd[j] + right[j] * sx;
        // what box_cull_kernel found out about the wave's rows, four bits a row (row rr in bits 4rr..4rr+3)
        static_assert(R <= 16, "sixteen row codes to a qword");
        unsigned long long rowcodes = 0ull;
        // (rows past the last one read on into the table's padding: `valid` masks them out)
        const uint32_t *cp = tg.cull + ((size_t)blockIdx.z * tg.row_count + row0) * tg.cull_words + (blockIdx.x >> 3);
        const int nibble = 4 * (blockIdx.x & 7);
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
            rowcodes |= (unsigned long long)((cp[rr * tg.cull_words] >> nibble) & 15u) << (4 * rr);
        // Background rows need |dir|^2 only to ~2^-19 (see the guard below): as a quadratic in sy,
        //   |base - up*sy|^2 = base.base - 2*sy*(base.up) + sy^2*(up.up),
        // it costs two fma per row instead of the N-1 other components and their squares.  Its absolute error is
        // ~2.7n*2^-24*(base.base + sy^2 up.up), and that is relative to the result as long as the cross term cannot
        // cancel the squares: lanes check (base.up)^2 <= base.base*up.up/16 (any sane camera: up is orthogonal to
        // forward and right), and a wave with a lane that fails it never takes the shortcut.
        float bb = 0.0f, bu = 0.0f, uu = 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            bb = fmaf(base[j], base[j], bb);
            bu = fmaf(base[j], up[j], bu);
            uu = fmaf(up[j], up[j], uu);
        }
        const float m2bu = -2.0f * bu;
        const bool fastsq = __builtin_amdgcn_ballot_w64(!(bu * bu <= bb * uu * 0.0625f)) == 0ull;
        const float maxv = (float)tg.plain_maxval;
        // rows painted as background outright / rows that are one face throughout / rows for the full treatment, as
        // masks with one bit per row AT THE ROW'S NIBBLE (bit 4rr): derived from the codes with a dozen scalar
        // operations on the whole qword
        const unsigned long long nib = 0x1111111111111111ull;
        unsigned long long validn = valid & 0xffffu;                       // bit rr -> bit 4rr
        validn = (validn | (validn << 24)) & 0x000000ff000000ffull;
        validn = (validn | (validn << 12)) & 0x000f000f000f000full;
        validn = (validn | (validn << 6)) & 0x0303030303030303ull;
        validn = (validn | (validn << 3)) & nib;
        unsigned long long quick = 0ull, inner = 0ull, todo = validn;
        // up[K]*sy of the lane's row, K = its face if it is of the second kind
        float v_usK = 0.0f;
        if (fastsq) {
            const unsigned long long n = rowcodes;
            const unsigned long long nz = (n | (n >> 1) | (n >> 2) | (n >> 3)) & nib;            // code != 0
            const unsigned long long hi3 = ((n >> 1) & (n >> 2) & (n >> 3)) & nib;               // code is 14 or 15
            const unsigned long long full = hi3 & n, skip = hi3 & ~n;                            // 15 / 14 (box_redo_kernel's from the start)
            quick = validn & ~nz;
            todo = validn & full;
            inner = validn & nz & ~hi3;
            (void)skip;
            const uint32_t lk = ((uint32_t)(rowcodes >> (4 * (lane & 15))) & 15u) - 1u;
            float upK = up[0];
#pragma unroll
            for (int j = 1; j < N; ++j) upK = lk == (uint32_t)j ? up[j] : upK;
            v_usK = upK * v_sy;
        }
        while (quick != 0ull) {
            const int rr = __builtin_ctzll(quick) >> 2;
            quick &= quick - 1ull;
            const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_sy), rr));
            const float us0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_us0), rr));
            const float d0 = base[0] - us0;                           // dir[0], bit for bit
            const float sqa = fmaf(sy, fmaf(sy, uu, m2bu), bb);
            // round(|dir[0]|/len * maxval), as in box_pixel, with the guard widened for sqa: sqa is within
            // (3.7n+4)*2^-24 of the reference's sum, so t is within ~22*2^-24 < 2^-19.4 of its value (n <= 8); guard 2^-18
            const float t = (fabsf(d0) * __builtin_amdgcn_rsqf(sqa)) * maxv;
            const bool clear = fabsf(__builtin_amdgcn_fractf(t) - 0.5f) > fmaf(t, 0x1p-18f, 0x1p-18f);
            if (__builtin_amdgcn_ballot_w64(!clear) != 0ull) {
                todo |= 1ull << (4 * rr);                                     // a lane too close to a rounding boundary
                continue;
            }
            uint32_t q = (uint32_t)(t + 0.5f);
            q = q < tg.plain_maxval ? q : tg.plain_maxval;
            PixelRef pr;
            pr.offset = (((long long)__builtin_amdgcn_readlane(v_off_hi, rr) << 32) | (unsigned)__builtin_amdgcn_readlane(v_off_lo, rr)) + xoff;
            emit_plain(tg, pr, d0 > 0.0f ? q : 0u, q);
        }
        while (inner != 0ull) {
            // every ray of the row's stretch hits face K (box_cull_kernel): the colour is |dir[K]|/len * (1, .5, .5)
            const int rr = __builtin_ctzll(inner) >> 2;
            inner &= inner - 1ull;
            const uint32_t K = ((uint32_t)(rowcodes >> (4 * rr)) & 15u) - 1u;
            const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_sy), rr));
            const float usK = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_usK), rr));
            float bK = base[0];
#pragma unroll
            for (int j = 1; j < N; ++j) bK = K == (uint32_t)j ? base[j] : bK;
            const float dK = bK - usK;                                // dir[K], bit for bit
            const float sqa = fmaf(sy, fmaf(sy, uu, m2bu), bb);
            const float t = (fabsf(dK) * __builtin_amdgcn_rsqf(sqa)) * maxv, th = t * 0.5f;
            const bool clear = fabsf(__builtin_amdgcn_fractf(t) - 0.5f) > fmaf(t, 0x1p-18f, 0x1p-18f) &&
                               fabsf(__builtin_amdgcn_fractf(th) - 0.5f) > fmaf(th, 0x1p-18f, 0x1p-18f);
            if (__builtin_amdgcn_ballot_w64(!clear) != 0ull) {
                todo |= 1ull << (4 * rr);
                continue;
            }
            uint32_t qr = (uint32_t)(t + 0.5f), qgb = (uint32_t)(th + 0.5f);
            qr = qr < tg.plain_maxval ? qr : tg.plain_maxval;
            qgb = qgb < tg.plain_maxval ? qgb : tg.plain_maxval;
            PixelRef pr;
            pr.offset = (((long long)__builtin_amdgcn_readlane(v_off_hi, rr) << 32) | (unsigned)__builtin_amdgcn_readlane(v_off_lo, rr)) + xoff;
            emit_plain(tg, pr, qr, qgb);
        }
        while (todo != 0ull) {
            const int rr = __builtin_ctzll(todo) >> 2;
            todo &= todo - 1ull;
            const bool rowhit = ((uint32_t)(rowcodes >> (4 * rr)) & 15u) != 0u;
            const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_sy), rr));
            PixelRef pr;
            pr.x = x;
            pr.y = 0;
            pr.offset = (((long long)__builtin_amdgcn_readlane(v_off_hi, rr) << 32) | (unsigned)__builtin_amdgcn_readlane(v_off_lo, rr)) + xoff;
            pr.hit_index = 0;
            pr.valid = true;
#pragma unroll
            for (int j = 0; j < N; ++j) dir[j] = base[j] - up[j] * sy;
            float sq = dir[0] * dir[0];
#pragma unroll
            for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
            if (!box_pixel<N, true, true>(tg, pr, org, dir, sq, dots, sx, sy, margin, rowhit)) {
                // a lane needs the reference's face-by-face arithmetic: leave the stretch to box_redo_kernel
                if (lane == 0)
                    atomicOr(tg.redo + ((size_t)blockIdx.z * tg.row_count + row0 + rr) * tg.redo_words + (blockIdx.x >> 5),
                             1u << (blockIdx.x & 31));
            }
        }
        return;
    }
    // ---- any other format ----
    const int x = (int)blockIdx.x * 64 + (tid & 63);
    if (x >= tg.width) return;
    const float sx = tg.fovI * ((float)x - tg.half_w);
    float base[N];
#pragma unroll
    for (int j = 0; j < N; ++j) base[j] = fwd[j] + right[j] * sx;
    // the stretch codes of the wave's rows, fetched together ahead of the loop (only "culled or not" is used here)
    uint32_t live = ~0u;
    if (tg.cull) {
        if (row0 >= tg.row_count) return;               // (keeps the reads inside the table's padding)
        const uint32_t *cp = tg.cull + ((size_t)blockIdx.z * tg.row_count + row0) * tg.cull_words + (blockIdx.x >> 3);
        live = 0u;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) live |= (((cp[rr * tg.cull_words] >> (4 * (blockIdx.x & 7))) & 15u) != 0u ? 1u : 0u) << rr;
    }
    for (int rr = 0; rr < R; ++rr) {
        const int row = row0 + rr;                      // relative to row_begin; the same for the whole wave
        if (row >= tg.row_count) return;
        const bool rowhit = (live >> rr) & 1u;
        const int orow = tg.row_begin + row;
        int y = orow;
        if (tg.band_world > 1) {
            const int band = orow / tg.band_rows;
            y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
        }
        if (y >= tg.height) continue;
        PixelRef pr;
        pr.x = x;
        pr.y = y;
        pr.offset = (long long)blockIdx.z * tg.frame_stride + (long long)(tg.compact ? orow : y) * tg.pitch + (long long)x * tg.bpp;
        pr.hit_index = 0;
        pr.valid = true;
        const float sy = tg.fovI * ((float)y - tg.half_h);
#pragma unroll
        for (int j = 0; j < N; ++j) dir[j] = base[j] - up[j] * sy;
        float sq = dir[0] * dir[0];
#pragma unroll
        for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
        box_pixel<N, false>(tg, pr, org, dir, sq, dots, sx, sy, margin, rowhit);
    }
}

// The stretches box_kernel<N, true> left behind (tg.redo): one wave per (frame, row, word of 32 stretches), every set
// bit rendered with the complete box_pixel -- classification, box_resolve, box_color.
// F32: three plain fp32 channels instead of packed RGB.  ZERO: hand the word back zeroed (the fused path's bitmap is
// marked with atomic ORs by box_tile_kernel and must be clean when the next launch starts).
// SPLIT (small launches, ZERO only): 2 or 4 waves per word, wave k for the stretches k, k + SPLIT, ... -- the marked stretches
// of a row come in runs, and with few rows in flight a run of ten is a long tail for one wave.
template <int N, bool F32 = false, bool ZERO = false, int SPLIT = 1>
__global__ __launch_bounds__(256) void box_redo_kernel(NtCameraFixed cam, NtTarget tg) {
    const int tid = (int)threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "waves per redo word");
    const int row = (int)blockIdx.y * (4 / SPLIT) + wv / SPLIT;
    if (row >= tg.row_count) return;
    uint32_t *word = tg.redo + ((size_t)blockIdx.z * tg.row_count + row) * tg.redo_words + blockIdx.x;
    const uint32_t mine = (SPLIT == 1 ? 0xffffffffu : SPLIT == 2 ? 0x55555555u : 0x11111111u) << (wv % SPLIT);
    uint32_t todo = *word & mine;
    if (todo == 0u) return;
    if (ZERO && (tid & 63) == 0) {
        if (SPLIT > 1) atomicAnd(word, ~mine);  // (the other waves may not have read the word yet: each needs its own bits only)
        else *word = 0u;
    }
    float org[N], right[N], up[N], fwd[N], dir[N];
    load_camera<N>(cam, org, right, up, fwd);
    float margin = fabsf(org[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) margin = fmaxf(margin, fabsf(org[j]));
    margin = NT_BOX_MARGIN * (1.0f + margin);
    float dots[4];
    if (cam.buf) {
        const float *dp = cam.dots + (size_t)blockIdx.z * 4;
        dots[0] = dp[0]; dots[1] = dp[1]; dots[2] = dp[2]; dots[3] = dp[3];
    } else {
        dots[0] = cam.odots[0]; dots[1] = cam.odots[1]; dots[2] = cam.odots[2]; dots[3] = cam.odots[3];
    }
    const int orow = tg.row_begin + row;
    int y = orow;
    if (tg.band_world > 1) {
        const int band = orow / tg.band_rows;
        y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
    }
    if (y >= tg.height) return;
    const float sy = tg.fovI * ((float)y - tg.half_h);
    while (todo != 0u) {
        const int bit = __builtin_ctz(todo);
        todo &= todo - 1u;
        int x = ((int)blockIdx.x * 32 + bit) * 64 + (tid & 63);
        x = x < tg.width ? x : tg.width - 1;            // as in box_kernel<N, true>
        PixelRef pr;
        pr.x = x;
        pr.y = y;
        pr.offset = (long long)blockIdx.z * tg.frame_stride + (long long)(tg.compact ? orow : y) * tg.pitch + (long long)x * tg.bpp;
        pr.hit_index = 0;
        pr.valid = true;
        const float sx = tg.fovI * ((float)x - tg.half_w);
#pragma unroll
        for (int j = 0; j < N; ++j) dir[j] = (fwd[j] + right[j] * sx) - up[j] * sy;
        float sq = dir[0] * dir[0];
#pragma unroll
        for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
        uint32_t sets = 0u;
        if (N <= 8 && tg.tie_sets) {
            const int stretch = (int)blockIdx.x * 32 + bit;
            sets = (uint32_t)__builtin_amdgcn_readfirstlane((int)tg.tie_sets[((size_t)blockIdx.z * tg.row_count + row) * ((tg.width + 63) / 64) + stretch]);
        }
        box_pixel<N, !F32, false, true, F32>(tg, pr, org, dir, sq, dots, sx, sy, margin, true, -1, sets);
    }
}

// What can be said about a whole 64-pixel stretch of a row?  One thread per stretch.  Its rays are v = vc + right*e
// with vc the direction through its middle and |e| <= 32*fovI: v_j lies in [vc_j - g_j, vc_j + g_j],
// g_j = 32*fovI*|right_j| (+1e-6 for the rounding of v itself).
//  * code 0 -- no ray can reach the cube.  A ray that comes within h = 1 + 2m + 1e-3 of the cube in every coordinate
//    at some tau > 0 (every ray the reference could call a hit does, see box_classify) satisfies
//        (vc_j + g_j)*tau >= -h - o_j     and     (vc_j - g_j)*tau <= h - o_j         for every j:
//    2n half-lines in tau; an empty intersection clears the stretch, and box_kernel paints background there without
//    looking further.  Convexity makes this sharp: what is left is within half a stretch of the cube's silhouette.
//  * code K+1 -- every ray clearly hits face K, K = the face the middle ray enters last.  With v_K of one sign over
//    the stretch, tau_K = (s_K - o_K)/v_K ranges over [tlo, thi]; if for every other j the extremes of
//    o_j + v_j*tau over that box stay inside 1 - m*(1 + |v_j|max/|v_K|min) (less 1e-4 for the arithmetic here), then for
//    each ray the reference's test of face K passes with room to spare, and every other slab was entered at least
//    m/|v_K| earlier, i.e. while p_K was outside 1+m, so no face before K can pass its j = K check (the argument of
//    box_classify).  box_kernel shades such rows from v_K alone.
//  * code 15 -- anything else: box_kernel classifies the rays one by one;  code 14 -- box_kernel skips the stretch and
//    box_redo_kernel renders it (its redo bit is set here).
// Reciprocals are approximate (v_rcp_f32); the slacks above are ~1000x their error.
// The code of one 64-pixel stretch (see the comment above): row y of the image, stretch `col` of the row.
// `sets` (box_tile_kernel): for a stretch that may end up with box_redo_kernel (codes 14 and 15), what box_resolve works out ray by ray from entry times -- the faces T that
// can still be the reference's answer and the coordinates C one of them could fail at -- as supersets valid for EVERY ray
// of the stretch (bits 0..9: T, bits 10..19: C, bit 31: valid), so that box_redo_kernel goes straight to the reference's
// arithmetic on them.  With [A_j, B_j] the range of slab j's entry time over the stretch's directions: every ray's last
// entry is at or after TN = max_j A_j; M = m / (the smallest |v_j| of an axis that can be last, B_j >= TN) is at least the
// ray's m/|v_K|; so a face within m/|v_K| of a ray's last entry has B_j >= TN - M: that is T.  The entries of T's faces,
// for any ray, lie in [min_{T} A_j, max_j B_j]; a coordinate that stays inside 1 - m over that span of tau, for all the
// stretch's directions, passes every test made there: the others, and T itself, are C.  (A face of T that a given ray
// enters long before its last entry fails at that ray's last axis, which is in T, hence in C.)  No valid sets when a candidate's v_j
// changes sign in the stretch or TN - M is not clearly positive (rays starting on or in the cube: box_color's business).
// v_max_f32 / v_min_f32 as they are: fmaxf / fminf on a value that reaches them from another basic block come with a
// canonicalising `v_max_f32 x, x, x` per operand (the compiler cannot see that the value is the result of arithmetic), and
// these instructions run at half the rate of a multiply: an eighth of box_stretch_code's instructions were such no-ops.
// Quiet NaNs drop out of the hardware's max / min as they do out of fmaxf / fminf; every operand here is the result of
// arithmetic or an infinity.
__device__ __forceinline__ float nt_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nt_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nt_vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float nt_vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float nt_vmed3(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int N, bool SETS = false>
__device__ __forceinline__ unsigned long long box_stretch_code2(const float (&org)[N], const float (&right)[N], const float (&up)[N],
                                                                const float (&fwd)[N], const NtTarget &tg, int y, int col) {
    // returns the code in the low dword and, SETS, the sets in the high one
    uint32_t sets_value = 0u;
    uint32_t *const sets = SETS ? &sets_value : nullptr;
    uint32_t code = 0u;
    float omax = fabsf(org[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) omax = fmaxf(omax, fabsf(org[j]));
    const float m = NT_BOX_MARGIN * (1.0f + omax);
    const float h = 1.0f + 2.0f * m + 1e-3f;
    const float sxc = tg.fovI * (((float)(col * 64) + 31.5f) - tg.half_w);
    const float sy = tg.fovI * ((float)y - tg.half_h);
    const float spread = 32.0f * tg.fovI;
    float tlo = 0.0f, thi = INFINITY;
    float vc[N], g[N];
    float tn = -INFINITY, tn2 = -INFINITY, vK = 0.0f, gK = 0.0f, oK = 0.0f;
    int K = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        vc[j] = (fwd[j] + right[j] * sxc) - up[j] * sy;
        g[j] = fmaf(spread, fabsf(right[j]), 1e-6f);
        const float pa = vc[j] + g[j], qa = -h - org[j];
        const float pb = vc[j] - g[j], qb = h - org[j];
        const float ra = qa * __builtin_amdgcn_rcpf(pa), rb = qb * __builtin_amdgcn_rcpf(pb);
        // pa*tau >= qa bounds tau from below when pa > 0, from above when pa < 0; pb*tau <= qb the other way round.  A zero
        // (always +0: g > 0, and x + y is -0 only for two negative zeros) goes with the positive side: ra is then
        // -inf (no bound), +inf (qa > 0: no tau at all -- tlo = inf) or a NaN, 0*inf, which drops out of max / min
        const bool ap = pa >= 0.0f, bp = pb >= 0.0f;
        const float lo_a = ap ? ra : -INFINITY, hi_a = ap ? INFINITY : ra;
        const float hi_b = bp ? rb : INFINITY, lo_b = bp ? -INFINITY : rb;
        tlo = nt_vmax3(tlo, lo_a, lo_b);
        thi = nt_vmin3(thi, hi_a, hi_b);
    }
    if (!(tlo > thi)) {                              // a NaN keeps the stretch
        code = 15u;
        // the middle ray's entries into the slabs: the last one, K, and the one before it (any K is verified below, so
        // accuracy only matters for the yield).  Only for stretches that survive: most tiles have none.
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float nr = ((vc[j] < 0.0f ? 1.0f : -1.0f) - org[j]) * __builtin_amdgcn_rcpf(vc[j]);
            tn2 = nt_vmed3(tn, tn2, nr);                        // second-to-last entry
            const bool later = nr > tn;
            vK = later ? vc[j] : vK;
            gK = later ? g[j] : gK;
            oK = later ? org[j] : oK;
            K = later ? j : K;
            tn = nt_vmax(tn, nr);
        }
        const float vKa = vK - gK, vKb = vK + gK;
        if (K < 13 && vKa * vKb > 0.0f) {            // (the code of a one-face stretch is K + 1 <= 13: 14 and 15 are taken; faces 13..15 of N > 13 go without)
            const float num = (vK < 0.0f ? 1.0f : -1.0f) - oK;
            const float t1 = num * __builtin_amdgcn_rcpf(vKa), t2 = num * __builtin_amdgcn_rcpf(vKb);
            const float t_lo = nt_vmin(t1, t2) * (1.0f - 1e-6f), t_hi = nt_vmax(t1, t2) * (1.0f + 1e-6f);
            // (the smaller of |vKa|, |vKb| with both of one sign: |vK| - gK, rounded the same way)
            const float rK = m * __builtin_amdgcn_rcpf(fabsf(vK) - gK) * (1.0f + 1e-6f);
            bool ok = t_lo > 1e-3f && t_hi < 1e30f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float va = vc[j] - g[j], vb = vc[j] + g[j];
                const float pmax = org[j] + nt_vmax(vb * t_lo, vb * t_hi);
                const float pmin = org[j] + nt_vmin(va * t_lo, va * t_hi);
                // (the larger of |va|, |vb| is |vc| + g, rounded the same way)
                const float lim = (1.0f - m - 1e-4f) - (fabsf(vc[j]) + g[j]) * rK;
                ok = ok && (j == K || (pmax <= lim && pmin >= -lim));
            }
            if (ok) code = (uint32_t)K + 1u;
        }
        // The middle ray enters two slabs within m/|v_K| of each other: box_classify would call the rays around it
        // unclear and box_kernel would hand the stretch to box_redo_kernel after classifying all of it -- send it
        // there directly (code 14; only a prediction: box_redo_kernel is right for any stretch).
        if (code == 15u && !((tn - tn2) * fabsf(vK) > m)) code = 14u;
        if (code >= 14u) {
            if (sets != nullptr && N <= 10) {
                auto range = [&](int j, float &A, float &B, float &vabs) {
                    const float va = vc[j] - g[j], vb = vc[j] + g[j];
                    const float num = (vc[j] < 0.0f ? 1.0f : -1.0f) - org[j];
                    const float e1 = num * __builtin_amdgcn_rcpf(va), e2 = num * __builtin_amdgcn_rcpf(vb);
                    const float lo = nt_vmin(e1, e2), hi = nt_vmax(e1, e2);
                    const bool same = va * vb > 0.0f;
                    A = same ? lo - fabsf(lo) * 1e-6f : -INFINITY;             // (v_rcp_f32: 1 ulp)
                    B = same ? hi + fabsf(hi) * 1e-6f : INFINITY;
                    vabs = same ? fabsf(vc[j]) - g[j] : 0.0f;                  // (the smaller of |va|, |vb|)
                };
                // (the ranges are kept for the two passes that follow: with interleaved rows nearly every wave of the middle strips
                // has a stretch that needs its sets, so this is no longer a rare path -- and working each range out three times, two
                // v_rcp_f32 each, was a twentieth of the kernel's vector instructions; 3N registers that are dead again before
                // the row loops begin)
                float rA[N], rB[N], rV[N];
                float TN = -INFINITY, TH = -INFINITY;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    range(j, rA[j], rB[j], rV[j]);
                    TN = nt_vmax(TN, rA[j]);
                    TH = nt_vmax(TH, rB[j]);
                }
                float vmin = INFINITY;                  // the smallest |v_j| an axis that can be last has anywhere in the stretch
#pragma unroll
                for (int j = 0; j < N; ++j) vmin = rB[j] >= TN ? nt_vmin(vmin, rV[j]) : vmin;
                const float M = m * __builtin_amdgcn_rcpf(vmin) * (1.0f + 1e-5f);
                // T, and the earliest entry of any of its faces for any ray: every test the redo kernel makes is made at a
                // tau in [t_lo, t_hi]
                uint32_t T = 0u, C = 0u;
                float t_lo = INFINITY;
                const float t_hi = TH;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const bool inT = rB[j] >= TN - M;
                    T |= inT ? 1u << j : 0u;
                    t_lo = inT ? nt_vmin(t_lo, rA[j]) : t_lo;
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float va = vc[j] - g[j], vb = vc[j] + g[j];
                    const float pmax = org[j] + nt_vmax(vb * t_lo, vb * t_hi);
                    const float pmin = org[j] + nt_vmin(va * t_lo, va * t_hi);
                    const float lim = 1.0f - m - 1e-4f;
                    const bool inC = ((T >> j) & 1u) != 0u || !(pmax <= lim && pmin >= -lim);
                    C |= inC ? 1u << j : 0u;
                }
                const bool valid = vmin > 0.0f && t_lo > 1e-3f && t_hi < 1e30f;         // (a NaN fails)
                *sets = valid ? (0x80000000u | (C << 10) | T) : 0u;
            }
        }
    }
    return (unsigned long long)code | ((unsigned long long)sets_value << 32);
}
template <int N>
__device__ __forceinline__ uint32_t box_stretch_code(const float (&org)[N], const float (&right)[N], const float (&up)[N], const float (&fwd)[N],
                                                     const NtTarget &tg, int y, int col) {
    return (uint32_t)box_stretch_code2<N, false>(org, right, up, fwd, tg, y, col);
}

template <int N>
__global__ __launch_bounds__(256) void box_cull_kernel(NtCameraFixed cam, NtTarget tg, uint32_t *out, int ncols) {
    float org[N], right[N], up[N], fwd[N];
    load_camera<N>(cam, org, right, up, fwd);
    // a half-wave = 32 stretches of one row: blockIdx.x = which 32, blockIdx.y = which 8 rows
    const int word = (int)blockIdx.x, row = (int)blockIdx.y * 8 + (int)(threadIdx.x >> 5);
    const int col = word * 32 + (int)(threadIdx.x & 31);
    uint32_t code = 0u;
    if (row < tg.row_count && col < ncols) {
        const int orow = tg.row_begin + row;
        int y = orow;
        if (tg.band_world > 1) {
            const int band = orow / tg.band_rows;
            y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
        }
        code = box_stretch_code<N>(org, right, up, fwd, tg, y, col);
    }
    const unsigned long long direct = __builtin_amdgcn_ballot_w64(code == 14u);      // redo bits set here
    // eight stretches to a dword
    uint32_t packed = code << (4 * (threadIdx.x & 7));
    packed |= (uint32_t)__shfl_xor((int)packed, 1, 64);
    packed |= (uint32_t)__shfl_xor((int)packed, 2, 64);
    packed |= (uint32_t)__shfl_xor((int)packed, 4, 64);
    if (row < tg.row_count) {
        if ((threadIdx.x & 7) == 0) out[((size_t)blockIdx.z * tg.row_count + row) * tg.cull_words + (col >> 3)] = packed;
        if ((threadIdx.x & 31) == 0)
            tg.redo[((size_t)blockIdx.z * tg.row_count + row) * tg.redo_words + word] = (threadIdx.x & 32) ? (uint32_t)(direct >> 32) : (uint32_t)direct;
    }
}

// --------------------------------------------------------------------------------------
// The fused path for the two formats the reference's scripts render into (packed plain RGB of <= 10 bits a channel in
// one aligned dword -- RGBX8 & co.: F32 = false; three plain fp32 channels, 12-byte pixels: F32 = true):
//
//   box_tile_kernel<N, F32, ROWS, WAVES>   a block = 64 columns x WAVES*ROWS rows (at most 64: one row per lane of the wave
//                                   that works out the stretch codes, box_stretch_code, and leaves them in LDS); after the
//                                   barrier every wave renders its ROWS rows from them with the lean loops, sixteen rows (a
//                                   qword of codes) at a time.  Per-row parameters come from the host's row table through
//                                   scalar loads (NtTarget::rowtab).  A row it cannot settle (code 14, or a lane that needs
//                                   the reference's face-by-face arithmetic) gets its bit set in the redo bitmap,
//                                   [frame][row][word of 32 stretches], with an atomic OR.
//   box_redo_kernel<N, F32>         one wave per (frame, row, word): the marked stretches with box_pixel<REDO>; it hands the
//                                   word back zeroed, so the bitmap is clean for the next launch (the host zeroes it once).
// No pre-kernel; the only scratch is the bitmap (one bit per 64 pixels).
// --------------------------------------------------------------------------------------
// ROWS x WAVES: 64 x 1 for launches of 512 rows or more with waves to spare (what depends on the column alone is set up once
// for 64 rows); otherwise 16 (8 in small launches) x 4, or x 3 when that leaves fewer idle waves below the last row of the
// launch (a rank's 136 rows of a 1080-row frame: three tiles of 48 rows instead of three of 64).
// (experiment switch: -DNT_EXP_NOSTORE times the tile kernel's lean loops without their stores)
#ifdef NT_EXP_NOSTORE
#define NT_EXP_STORE_IF if (tg.width < 0)
#else
#define NT_EXP_STORE_IF
#endif
// Occupancy (-DNT_TILE_OCC=..., set per dimension in nt_inst_box.hip).  The waves a SIMD are decided by BOTH register files:
// 512 / VGPRs and floor(800 / (ceil(SGPRs / 16) * 16 + 16)) -- with 106 SGPRs that is six, whatever `amdgpu_waves_per_eu` says
// about the vector registers (round 2 "held the kernel to 64 VGPRs for eight waves" and saw 1 % for six spilled dwords a lane:
// it was still running six).  Up to n = 6 the kernels are given an SGPR budget of 96, which admits seven.
#ifndef NT_TILE_OCC
#define NT_TILE_OCC
#endif
#ifndef NT_BOX_PIN_UP
#define NT_BOX_PIN_UP 1
#endif
template <int N, bool F32, int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) NT_TILE_OCC void box_tile_kernel(NtCameraFixed cam, NtTarget tg) {
    static_assert(ROWS == 8 || ROWS == 16 || ROWS == 32 || ROWS == 64, "sixteen row codes to a qword, one to four qwords a wave");
    static_assert(WAVES >= 1 && WAVES <= 4 && WAVES * ROWS <= 64, "the codes of a tile are the work of one wave, a row per lane");
    constexpr int R = ROWS;
    __shared__ uint32_t s_code[64];
    // ... and the same as one bit per tile row and class (culled / one face / ray by ray / near-tie): the row loops are driven by
    // 32-bit masks of row bits -- two scalar instructions to step, one to index the row table (masks of nibble positions in a
    // qword cost five and two; the scalar unit is as busy as the vector units in these loops)
    __shared__ unsigned long long s_rows[4];
    // F32: this kernel is bound by its stores (12 bytes a pixel), not by vector instructions, and renders the near-tie and
    // unclear stretches itself -- no second kernel; the tie sets of its rows stay in LDS
    // ... and so does the packed format up to eight dimensions, where the codes wave's tie sets (LDS) give the near-tie
    // stretches their short cut: the second kernel's work costs this kernel a tenth of its time and saves a third of it
    // (BoxScene(6): 423 -> 410 us a call; 75 VGPRs instead of 70).  Beyond eight -- no tie sets -- it loses: BoxScene(10) 4096^2
    // 6-12 % slower without box_redo_kernel, BoxScene(16) 15 %.
    constexpr bool ALLIN = F32 || N <= NT_BOX_INLINE_MAX_N;
    constexpr bool SETS_LDS = ALLIN && N <= NT_BOX_SETS_MAX_N;
    __shared__ uint32_t s_sets[SETS_LDS ? 64 : 1];
    __shared__ int s_abort;                  // the codes wave's reading of NtTarget::abort_word: one answer for the whole block
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef NT_EXP_TRACE
    const unsigned long long trace_t0 = wall_clock64();          // ablation builds: per-wave residency record (tools/box_wave_trace.py)
    unsigned trace_rows = 0u;                                    // rows by class: culled | one face << 8 | ray by ray << 16 | near-tie << 24
    unsigned long long trace_t1 = 0ull;                          // end of the codes phase
#endif
    // Work order.  Blocks start in grid order -- column, tile row, then z -- and a wave with sixty-four rows to sort ray by ray
    // lives ten times as long as one with none: with z = frame the long waves of the last frames were still running, almost
    // alone, 60 us after the grid had been handed out (tools/box_wave_trace.py).  The long waves are in the middle columns of
    // the image (the scripts' cameras look at the cube), so those columns run tg.lead_frames frames ahead of the outer ones: z
    // counts lead_frames + nframes slots, slot z holds the middle columns of frame z and the outer columns of frame
    // z - lead_frames; the grid ends on short waves only.  (A block whose frame does not exist leaves at once.)
    const unsigned nframes = gridDim.z - (unsigned)tg.lead_frames;
    unsigned frame = blockIdx.z;
    if (tg.lead_frames > 0) {
        const unsigned cols = gridDim.x, q = cols / 4u;
        const bool middle = blockIdx.x >= q && blockIdx.x < cols - q;
        if (!middle) frame -= (unsigned)tg.lead_frames;          // (wraps for the first slots: >= nframes)
        if (frame >= nframes) return;
    }
    float org[N], right[N], up[N], fwd[N], dir[N];
    load_camera<N>(cam, frame, org, right, up, fwd);
    float margin = fabsf(org[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) margin = fmaxf(margin, fabsf(org[j]));
    margin = NT_BOX_MARGIN * (1.0f + margin);
    float dots[4];
    if (cam.buf) {
        const float *dp = cam.dots + (size_t)frame * 4;
        dots[0] = dp[0]; dots[1] = dp[1]; dots[2] = dp[2]; dots[3] = dp[3];
    } else {
        dots[0] = cam.odots[0]; dots[1] = cam.odots[1]; dots[2] = cam.odots[2]; dots[3] = cam.odots[3];
    }
    const int tile_row0 = (int)blockIdx.y * WAVES * R;
    // ---- phase 1: the tile's stretch codes, one row per lane of one wave -- a different one from block to block, so that
    // the extra work does not always land on the same SIMD of a CU
    if (wv == (int)((blockIdx.x + blockIdx.y + frame) % (unsigned)WAVES)) {
        if (tg.abort_word != nullptr) {
            const bool ab = nt_aborted(tg);
            if (WAVES == 1) { if (ab) return; }
            else if (lane == 0) s_abort = ab ? 1 : 0;
        }
        uint32_t code = 0u;
        uint32_t row_sets = 0u;
        // lane <-> slot tile_row0 + lane = row rr of wave w; interleaved: that wave's rr-th row is w + W * rr (NtTarget::row_il)
        const int trow = tg.row_il > 0 ? ((int)blockIdx.y * WAVES + lane / R) + tg.row_il * (lane % R) : tile_row0 + lane;
        if (lane < WAVES * R && trow < tg.row_count) {
            const int orow = tg.row_begin + trow;
            int y = orow;
            if (tg.band_world > 1) {
                const int band = orow / tg.band_rows;
                y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
            }
            if (y < tg.height) {
                // (N > 8: the sets' arithmetic would raise the kernel's register allocation -- 124 VGPRs and spills at N = 10 --
                // for every row; those dimensions go without)
                const unsigned long long cs = box_stretch_code2<N, (N <= NT_BOX_SETS_MAX_N)>(org, right, up, fwd, tg, y, (int)blockIdx.x);
                code = (uint32_t)cs;
                const uint32_t sets = (uint32_t)(cs >> 32);
                // (every marked stretch gets a fresh entry: the sets here, 0 from the wave that marks a row it looked at)
                row_sets = sets;
                if (!ALLIN && N <= 8 && code >= 14u && tg.tie_sets) tg.tie_sets[((size_t)frame * tg.row_count + trow) * gridDim.x + blockIdx.x] = sets;
            }
        }
        if (SETS_LDS) s_sets[lane] = row_sets;
        {
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(code == 0u), m1 = __builtin_amdgcn_ballot_w64(code >= 1u && code <= 13u),
                                     m2 = __builtin_amdgcn_ballot_w64(code == 15u), m3 = __builtin_amdgcn_ballot_w64(code == 14u);
            if (lane == 0) {
                s_rows[0] = m0;
                s_rows[1] = m1;
                s_rows[2] = m2;
                s_rows[3] = m3;
            }
        }
        // rows of wave w in nibbles of s_code[2w] (rows 0..7) and s_code[2w + 1] (rows 8..15); R == 32: s_code[4w .. 4w + 3]
        uint32_t packed = code << (4 * (lane & 7));
        packed |= (uint32_t)__shfl_xor((int)packed, 1, 64);
        packed |= (uint32_t)__shfl_xor((int)packed, 2, 64);
        packed |= (uint32_t)__shfl_xor((int)packed, 4, 64);
        if ((lane & 7) == 0) {
            const int grp = lane >> 3;                                   // eight rows each
            const int slot = R == 8 ? 2 * grp : grp;                     // R == 8: wave w's rows are group w
            s_code[slot] = packed;
            if (R == 8) s_code[slot + 1] = 0u;
        }
    }
    __syncthreads();
    if (WAVES > 1 && tg.abort_word != nullptr && s_abort != 0) return;
#ifdef NT_EXP_TRACE
    trace_t1 = wall_clock64();
#endif
    // Sixteen rows at a time (their codes fill a qword), once or -- R == 32 -- twice per wave: what depends on the column
    // alone (forward + right*sx, the quadratic for |dir|^2) is set up once for all the wave's rows.
    constexpr int HALVES = R >= 32 ? R / 16 : 1, RH = R / HALVES;
    const int wrow0 = tile_row0 + wv * R;                     // the wave's first slot (its first row when rows are not interleaved)
    const int il = tg.row_il;                                 // (scalar) 0, or the stride between a wave's rows
    const int wfirst = il > 0 ? (int)blockIdx.y * WAVES + wv : wrow0;
    if (wfirst < tg.row_count) {
        typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(4))) const nt_u32x4 *nt_rowtab;
        uint8_t *const frame_base = tg.dest + (long long)frame * tg.frame_stride;
        // What a row loop needs to know about its row -- sy of the ray source and the row's byte offset in a frame -- comes
        // from a table the host wrote (NtTarget::rowtab, 16 bytes per owned row: sy, -, offset), read through the scalar
        // data cache (constant address space: s_load_dwordx4): no vector instruction, nothing per lane
#define NT_ROW_LOAD(rr)                     \
        const nt_u32x4 row_e = tab[(rr)];   \
        const float sy = __uint_as_float(row_e.x)
#define NT_ROW_OFF() ((long long)(((unsigned long long)row_e.w << 32) | row_e.z) + (long long)frame * tg.frame_stride)
        // a store goes to (row pointer: scalar registers) + (the lane's byte offset in the row: 32 bits) -- the addressing
        // mode of global_store with an SGPR base, no vector arithmetic on addresses
#define NT_ROW_PTR() uniform_ptr(frame_base + (long long)(((unsigned long long)row_e.w << 32) | row_e.z))
        // (instruction selection works block by block and only recognises base + zero-extended 32-bit offset when it sees the
        // extension: the empty asm keeps it from being hoisted out of the row loops)
#define NT_LANE_OFF() ({ asm volatile("" : "+v"(xoff)); xoff; })
        // mask &= ~(1 << bit) as ONE scalar instruction (the compiler writes mask & (mask - 1) as an add and an and): the lean loops
        // run about as many scalar instructions a row as vector ones
#define NT_CLEAR_BIT(mask, bit) asm("s_bitset0_b32 %0, %1" : "+s"(mask) : "s"(bit))
        int x = (int)blockIdx.x * 64 + lane;
        x = x < tg.width ? x : tg.width - 1;
        uint32_t xoff = (uint32_t)x * (uint32_t)tg.bpp;
        const float sx = tg.fovI * ((float)x - tg.half_w);
        float base[N];
#pragma unroll
        for (int j = 0; j < N; ++j) base[j] = fwd[j] + right[j] * sx;
        // `up` in vector registers: up[j] * sy has the scalar sy as its one scalar operand
        float upv[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            upv[j] = up[j];
            // (NT_BOX_PIN_UP = 0: only up[0], the one the culled rows' loop multiplies -- the others stay in scalar registers and
            // are moved over where a row needs them, which is what lets the n = 6 kernel fit seven waves' registers unspilled)
            // (the fp32 kernel, bound by its stores, keeps them pinned: unpinned it executes 8 % more instructions for nothing)
            if (NT_BOX_PIN_UP || F32 || j == 0) asm volatile("" : "+v"(upv[j]));
        }
        // packed RGB: the quadratic |dir|^2 = bb - 2 bu sy + uu sy^2 of the guarded rsq quantisation (see box_kernel<N, true>)
        float bb = 0.0f, bu = 0.0f, uu = 0.0f;
        bool fastsq = true;
        if (!F32) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                bb = fmaf(base[j], base[j], bb);
                bu = fmaf(base[j], up[j], bu);
                uu = fmaf(up[j], up[j], uu);
            }
            fastsq = __builtin_amdgcn_ballot_w64(!(bu * bu <= bb * uu * 0.0625f)) == 0ull;
        }
        const float maxv = (float)tg.plain_maxval;
        // ... scaled by 1/maxval^2, so that rsq of it is maxval/|dir|: the multiplication by maxval costs three instructions a
        // wave instead of one a row (three more roundings of 2^-24 each in a budget that had 2.6x room, see the guard)
        const float inv_maxv2 = 1.0f / (maxv * maxv);
        const float m2bu = (-2.0f * bu) * inv_maxv2;
        bb = bb * inv_maxv2;
        uu = uu * inv_maxv2;
        // the classes of the wave's rows, a bit per row
        auto rows_of = [&](int k) {
            const unsigned long long m = s_rows[k];
            return (((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m)) >> (wv * R);
        };
        const unsigned long long rows_culled = rows_of(0), rows_face = rows_of(1), rows_rays = rows_of(2), rows_tie = rows_of(3);
#pragma unroll 1
        for (int half = 0; half < HALVES; ++half) {
        const int row0 = wrow0 + 16 * half;                   // slot of the half's first row
        const int hfirst = il > 0 ? wfirst + il * 16 * half : row0;        // ... and that row
        if (hfirst >= tg.row_count) break;
        uint32_t redo_bits = 0u;                              // rows (bit rr) left to box_redo_kernel
        const int cw = (R >= 32 ? (R / 8) * wv + 2 * half : 2 * wv);
        unsigned long long rowcodes = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_code[cw + 1]) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)s_code[cw]);
        // Which of these rows exist: one row per lane (lane l <-> row row0 + l).  Every lane stays active in the row
        // loops -- lanes past the right edge redo the last pixel (same bytes, same value) instead of leaving
        const int lrow = il > 0 ? hfirst + il * lane : row0 + lane;
        const int lorow = tg.row_begin + lrow;
        int ly = lorow;
        if (tg.band_world > 1) {
            const int band = lorow / tg.band_rows;
            ly = (band * tg.band_world + tg.band_rank) * tg.band_rows + (lorow - band * tg.band_rows);
        }
        const uint32_t valid = (uint32_t)__builtin_amdgcn_ballot_w64(lane < RH && lrow < tg.row_count && ly < tg.height);
        // (interleaved rows: the table is in slot order and belongs to this launch's row range)
        const nt_rowtab tab = (nt_rowtab)tg.rowtab + (il > 0 ? row0 : tg.row_begin + row0);
        // (bit rr <-> row row0 + rr; code 14 -- a near-tie stretch -- is not looked at here unless this kernel is all there is)
        uint32_t quick = (uint32_t)(rows_culled >> (16 * half)) & valid, inner = (uint32_t)(rows_face >> (16 * half)) & valid;
        uint32_t todo = (uint32_t)((ALLIN ? rows_rays | rows_tie : rows_rays) >> (16 * half)) & valid;
        if (!ALLIN) redo_bits = (uint32_t)(rows_tie >> (16 * half)) & valid;
#ifdef NT_EXP_TRACE
        trace_rows += (unsigned)__builtin_popcount(quick) | ((unsigned)__builtin_popcount(inner) << 8) |
                      ((unsigned)__builtin_popcount((uint32_t)(rows_rays >> (16 * half)) & valid) << 16) |
                      ((unsigned)__builtin_popcount((uint32_t)(rows_tie >> (16 * half)) & valid) << 24);
#endif
        if (!F32) {
            // ---- packed RGB: guarded rsq quantisation (see box_kernel<N, true>)
            if (!fastsq) {
                todo |= quick | inner;
                quick = 0u;
                inner = 0u;
            }
            // (the two loops in two copies: 8-bit fields on byte boundaries -- v_perm_b32 packing, tg.plain_sel -- and the rest;
            // the test is made once here, not once a row)
            auto lean_rows = [&](auto sel8) {
            constexpr bool SEL8 = decltype(sel8)::value;
            while (quick != 0u) {
                const int rr = __builtin_ctz(quick);
                NT_CLEAR_BIT(quick, rr);
                NT_ROW_LOAD(rr);
                const float d0 = base[0] - upv[0] * sy;                   // dir[0], bit for bit
                const float sqa = fmaf(sy, fmaf(sy, uu, m2bu), bb);
                const float t = fabsf(d0) * __builtin_amdgcn_rsqf(sqa);          // (sqa is |dir|^2 / maxval^2)
                const bool clear = fabsf(__builtin_amdgcn_fractf(t) - 0.5f) > fmaf(t, 0x1p-18f, 0x1p-18f);
                if (__builtin_amdgcn_ballot_w64(!clear) != 0ull) {
                    todo |= 1u << rr;                                             // a lane too close to a rounding boundary
                    continue;
                }
                const nt_gptr out = NT_ROW_PTR() + NT_LANE_OFF();
                if (SEL8) {
                    // 8-bit fields: t + 2^23 has round(t) in its low mantissa byte (t < 255.5; the guard keeps t off the
                    // half-way points, so nearest-even is the reference's rounding), which is the byte v_perm_b32 picks
                    const uint32_t q = __float_as_uint(t + 8388608.0f);
                    // (the red byte is zero where dir[0] is not positive: a select on every row.  Until the end of round 3 the
                    // select sat behind a wave-uniform branch -- its sign rarely changes within a stretch -- but the branch and
                    // the jump around it are scalar instructions, and the scalar unit is the busier one in this loop: 1.4 % of
                    // the headline call)
                    const uint32_t w = __builtin_amdgcn_perm(d0 > 0.0f ? q : 0u, q, tg.plain_sel);
                    NT_EXP_STORE_IF NT_G32(out) = w;
                    continue;
                }
                uint32_t q = (uint32_t)(t + 0.5f);
                q = q < tg.plain_maxval ? q : tg.plain_maxval;
                const uint32_t w = (d0 > 0.0f ? q : 0u) * tg.plain_mul[0] + q * (tg.plain_mul[1] + tg.plain_mul[2]);      // (emit_plain)
                NT_EXP_STORE_IF NT_G32(out) = tg.reversed ? w : bswap32(w);
            }
            // (the one-face rows of a wave mostly share their face: its component of `base` is picked once)
            uint32_t K0 = 0u;
            float bK0 = base[0], uK0 = upv[0];
            if (inner != 0u) {
                K0 = ((uint32_t)(rowcodes >> (4 * __builtin_ctz(inner))) & 15u) - 1u;
#pragma unroll
                for (int j = 1; j < N; ++j) {
                    bK0 = K0 == (uint32_t)j ? base[j] : bK0;
                    uK0 = K0 == (uint32_t)j ? upv[j] : uK0;
                }
            }
            while (inner != 0u) {
                const int rr = __builtin_ctz(inner);
                NT_CLEAR_BIT(inner, rr);
                const uint32_t K = ((uint32_t)(rowcodes >> (4 * rr)) & 15u) - 1u;
                NT_ROW_LOAD(rr);
                float bK = bK0, uK = uK0;
                if (K != K0) {
                    bK = base[0];
                    uK = upv[0];
#pragma unroll
                    for (int j = 1; j < N; ++j) {
                        bK = K == (uint32_t)j ? base[j] : bK;
                        uK = K == (uint32_t)j ? upv[j] : uK;
                    }
                }
                const float dK = bK - uK * sy;                            // dir[K], bit for bit
                const float sqa = fmaf(sy, fmaf(sy, uu, m2bu), bb);
                const float t = fabsf(dK) * __builtin_amdgcn_rsqf(sqa), th = t * 0.5f;
                const bool clear = fabsf(__builtin_amdgcn_fractf(t) - 0.5f) > fmaf(t, 0x1p-18f, 0x1p-18f) &&
                                   fabsf(__builtin_amdgcn_fractf(th) - 0.5f) > fmaf(th, 0x1p-18f, 0x1p-18f);
                if (__builtin_amdgcn_ballot_w64(!clear) != 0ull) {
                    todo |= 1u << rr;
                    continue;
                }
                const nt_gptr out = NT_ROW_PTR() + NT_LANE_OFF();
                if (SEL8) {
                    NT_EXP_STORE_IF NT_G32(out) = __builtin_amdgcn_perm(__float_as_uint(t + 8388608.0f), __float_as_uint(th + 8388608.0f), tg.plain_sel);
                    continue;
                }
                uint32_t qr = (uint32_t)(t + 0.5f), qgb = (uint32_t)(th + 0.5f);
                qr = qr < tg.plain_maxval ? qr : tg.plain_maxval;
                qgb = qgb < tg.plain_maxval ? qgb : tg.plain_maxval;
                const uint32_t w = qr * tg.plain_mul[0] + qgb * (tg.plain_mul[1] + tg.plain_mul[2]);
                NT_EXP_STORE_IF NT_G32(out) = tg.reversed ? w : bswap32(w);
            }
            };
            if (tg.plain_sel != 0u) lean_rows(std::true_type{});
            else lean_rows(std::false_type{});
        } else {
            // ---- fp32 channels: the stored value IS x / sqrtf(sq), so the reference's sum, square root and division are
            // done as they stand -- but on rows whose code says which x it is, nothing else is
            while (quick != 0u) {
                // background rows: i = dir[0]; i > 0 ? (i,i,i) : (0,-i,-i) (tracer.hpp:109-113), clamped as channel_value does
                const int rr = __builtin_ctz(quick);
                quick &= quick - 1u;
                NT_ROW_LOAD(rr);
#pragma unroll
                for (int j = 0; j < N; ++j) dir[j] = base[j] - upv[j] * sy;
                float sq = dir[0] * dir[0];
#pragma unroll
                for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
                const float in = dir[0] / sqrt_wave(sq);
                float r, gb, b_;
                box_background(in, r, gb, b_);
                emit_f32x3_at(tg, NT_ROW_PTR() + NT_LANE_OFF(), r, gb);
            }
            uint32_t K0 = 0u;
            float bK0 = base[0], uK0 = upv[0];
            if (inner != 0u) {
                K0 = ((uint32_t)(rowcodes >> (4 * __builtin_ctz(inner))) & 15u) - 1u;
#pragma unroll
                for (int j = 1; j < N; ++j) {
                    bK0 = K0 == (uint32_t)j ? base[j] : bK0;
                    uK0 = K0 == (uint32_t)j ? upv[j] : uK0;
                }
            }
            while (inner != 0u) {
                // one face K throughout: sine = d_K * (-sign d_K) <= 0, shade = -sine (tracer.hpp:105-107)
                const int rr = __builtin_ctz(inner);
                inner &= inner - 1u;
                const uint32_t K = ((uint32_t)(rowcodes >> (4 * rr)) & 15u) - 1u;
                NT_ROW_LOAD(rr);
#pragma unroll
                for (int j = 0; j < N; ++j) dir[j] = base[j] - upv[j] * sy;
                float sq = dir[0] * dir[0];
#pragma unroll
                for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
                float bK = bK0, uK = uK0;
                if (K != K0) {
                    bK = base[0];
                    uK = upv[0];
#pragma unroll
                    for (int j = 1; j < N; ++j) {
                        bK = K == (uint32_t)j ? base[j] : bK;
                        uK = K == (uint32_t)j ? upv[j] : uK;
                    }
                }
                const float xk = bK - uK * sy;                            // dir[K], bit for bit (the same two operations)
                const float shade = fabsf(xk / sqrt_wave(sq));
                emit_f32x3_at(tg, NT_ROW_PTR() + NT_LANE_OFF(), shade * 1.0f, shade * 0.5f);
            }
        }
        while (todo != 0u) {
            const int rr = __builtin_ctz(todo);
            todo &= todo - 1u;
            const bool rowhit = ((uint32_t)(rowcodes >> (4 * rr)) & 15u) != 0u;
            NT_ROW_LOAD(rr);
            PixelRef pr;
            pr.x = x;
            pr.y = 0;
            pr.offset = NT_ROW_OFF() + (long long)xoff;
            pr.hit_index = 0;
            pr.valid = true;
#pragma unroll
            for (int j = 0; j < N; ++j) dir[j] = base[j] - upv[j] * sy;
            float sq = dir[0] * dir[0];
#pragma unroll
            for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
            // (a row with a face code is here because its cheap quantisation failed: the face is known)
            const int rcode = (int)((uint32_t)(rowcodes >> (4 * rr)) & 15u);
            if (ALLIN) {
                // everything here: classification, the reference's arithmetic on the faces in question (on the stretch's tie sets
                // for a near-tie stretch), box_color for rays that start on or in the cube
                uint32_t sets = 0u;
                if (SETS_LDS) sets = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_sets[wv * R + 16 * half + rr]);
                box_pixel<N, !F32, false, false, F32>(tg, pr, org, dir, sq, dots, sx, sy, margin, rowhit, rcode >= 1 && rcode <= 13 ? rcode - 1 : -1, sets,
                                                       rcode == 14);
            } else if (!box_pixel<N, true, true, false, false>(tg, pr, org, dir, sq, dots, sx, sy, margin, rowhit, rcode >= 1 && rcode <= 13 ? rcode - 1 : -1)) {
                redo_bits |= 1u << rr;
            }
        }
        // mark the rows left over in the redo bitmap (clean on entry: box_redo_kernel zeroes what it has read)
        if (!ALLIN && lane == 0) {
            while (redo_bits != 0u) {
                const int rr = __builtin_ctz(redo_bits);
                redo_bits &= redo_bits - 1u;
                const int mrow = il > 0 ? hfirst + il * rr : row0 + rr;
                if (N <= 8 && tg.tie_sets && ((uint32_t)(rowcodes >> (4 * rr)) & 15u) < 14u)
                    tg.tie_sets[((size_t)frame * tg.row_count + mrow) * gridDim.x + blockIdx.x] = 0u;
                atomicOr(tg.redo + ((size_t)frame * tg.row_count + mrow) * tg.redo_words + (blockIdx.x >> 5), 1u << (blockIdx.x & 31));
            }
        }
        }           // (sixteen rows)
    }
#ifdef NT_EXP_TRACE
    if (lane == 0) {       // the records lie behind the last frame: [frame][tile row][column][wave] x 4 qwords
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(tg.dest + (long long)nframes * tg.frame_stride) +
                                 ((((size_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WAVES + wv) * 4;
        tr[0] = trace_t0;
        tr[1] = wall_clock64();
        tr[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
        tr[3] = (unsigned long long)trace_rows | ((trace_t1 - trace_t0) << 32);
    }
#endif
#undef NT_ROW_LOAD
#undef NT_ROW_OFF
#undef NT_ROW_PTR
#undef NT_LANE_OFF
#undef NT_CLEAR_BIT
}

template <int N>
int launch_box_fixed(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg_in) {
    NtCameraFixed cf;
    cf.buf = cam.buf;
    cf.dots = cam.dots;
    for (int k = 0; k < 4; ++k) cf.odots[k] = cam.odots[k];
    cf.n = N;
    for (int k = 0; k < 4 * N; ++k) cf.inl[k] = cam.inl[k];
    NtTarget tg = tg_in;
    dim3 grid;
    grid_for(tg, 64, tg.colors_out ? 4 : 4 * BoxRows<N>::value, li.nframes, grid);
    tg.cull = nullptr;
    tg.tie_sets = nullptr;
    tg.redo = nullptr;
    tg.cull_words = 0;
    tg.redo_words = 0;
    // the two formats of the reference's scripts take the fused kernels (no pre-kernel): packed plain RGB of at most 10 bits
    // a channel in one aligned dword, and three plain fp32 channels
    const bool fmt_rgb = tg.plain_bits != 0u && tg.plain_bits <= 10u && tg.bpp == 4 && tg.aligned4;
    const bool fmt_f32 = tg.plain_f32[0] >= 0 && tg.bpp == 12 && tg.aligned4;
    if (li.cull_buf && tg.rowtab && !tg.colors_out && BoxRows<N>::value > 1 && (fmt_rgb || fmt_f32) && li.box_path != 0 && li.cull_clean) {
        hipStream_t st = (hipStream_t)li.stream;
        // rows a wave x waves a block (nt_box_tile_geom, nt_device.hpp -- the host's row table follows the same decision):
        // sixteen rows a lane once there are waves to spare (the per-wave set-up is a fifth of the work at eight), three waves a
        // block instead of four when that leaves fewer idle waves below the last row; sixty-four rows a lane, one wave a block
        // (the wave works out its own codes, and the set-up per column is shared by four times the rows) once the launch is tall
        // enough for such tiles to fit it well and has waves enough even so (four 4096 x 4096 frames are 16 384 such waves: 7 %
        // slower than with 16 rows a wave; sixteen frames: 9 % faster)
        NtBoxTileGeom geom;
        geom.rows = li.tile_rows;
        geom.waves = li.tile_waves;
        const bool r64 = geom.rows == 64, r16 = geom.rows == 16;
        const int wpb = geom.waves;
        const int tile_rows = geom.rows * geom.waves;
        dim3 tgrid((unsigned)((tg.width + 63) / 64), (unsigned)((tg.row_count + tile_rows - 1) / tile_rows), (unsigned)li.nframes);
        tg.redo_words = ((tg.width + 63) / 64 + 31) / 32;
        tg.redo = li.cull_buf;                        // [frame][row][redo_words], all zero between launches
        // the tie sets of the marked stretches, [frame][row][stretch] dwords (written with the mark)
        // (a buffer of their own: the bitmap's buffer must hold nothing but the bitmap, which has to be all zero whatever the
        // next launch's geometry is)
        tg.tie_sets = nullptr;               // (the tie sets stay in LDS: only kernels that need no second kernel have them)
        // the middle columns 96 frames ahead of the outer ones (32 .. 96 measure alike on the 160-frame call: -4 %, and on
        // BoxScene(3): -3 %; a rank's eighth of the call: -3 % with 48, -6 % with 96; 16: half of it), in launches of 8 frames
        // or more (eight 4096 x 4096 frames of BoxScene(10): 621 -> 693 Grays/s; four: no difference)
        tg.lead_frames = li.nframes >= 8 ? (li.nframes < 96 ? li.nframes : 96) : 0;
        // ... with interleaved rows (round 3) a strip's waves are all alike and little is left for the lead to do: 160-frame call
        // 358 us without, 347 with 32, 354 with 96; 32- / 64- / 320-frame calls 1-2 % better with 8..16 than without and than
        // with more; sixteen 4096 x 4096 frames of BoxScene(10) 3 % WORSE with 16 than without (tools/il_ab.py --var NTRACER_BOX_LEAD)
        if (tg.row_il > 0) tg.lead_frames = li.nframes >= 32 ? 16 : 0;
        if (const char *e = getenv("NTRACER_BOX_LEAD")) tg.lead_frames = atoi(e) > 0 && li.nframes > 1 ? atoi(e) : 0;        // (A/B)
        if ((long long)li.nframes + tg.lead_frames > 65535) tg.lead_frames = 0;        // (grid z)
        tgrid.z += (unsigned)tg.lead_frames;
        // few rows in flight: two waves per redo word
        const long long rwords = (long long)tg.row_count * li.nframes * tg.redo_words;
        int split = rwords < 48 * 1024 ? 2 : 1;                 // (87k words: one wave 3 % faster; 44k: even; 22k: two waves 2 % faster;
                                                                //  four waves a word measured slower than two)
        if (const char *e = getenv("NTRACER_BOX_SPLIT")) split = atoi(e) == 2 ? 2 : 1;        // (A/B)
        const int rpb = 4 / split;              // rows per block
        const dim3 rgrid((unsigned)tg.redo_words, (unsigned)((tg.row_count + rpb - 1) / rpb), (unsigned)li.nframes);
        if (fmt_rgb) {
            if (r64) hipLaunchKernelGGL((box_tile_kernel<N, false, 64, 1>), tgrid, dim3(64), 0, st, cf, tg);
            else if (r16 && wpb == 3) hipLaunchKernelGGL((box_tile_kernel<N, false, 16, 3>), tgrid, dim3(192), 0, st, cf, tg);
            else if (r16) hipLaunchKernelGGL((box_tile_kernel<N, false, 16, 4>), tgrid, dim3(256), 0, st, cf, tg);
            else hipLaunchKernelGGL((box_tile_kernel<N, false, 8, 4>), tgrid, dim3(256), 0, st, cf, tg);
            if (N > NT_BOX_INLINE_MAX_N) {          // (up to there the tile kernel leaves nothing behind)
                if (split == 2) hipLaunchKernelGGL((box_redo_kernel<N, false, true, 2>), rgrid, dim3(256), 0, st, cf, tg);
                else hipLaunchKernelGGL((box_redo_kernel<N, false, true, 1>), rgrid, dim3(256), 0, st, cf, tg);
            }
        } else {
            if (r64) hipLaunchKernelGGL((box_tile_kernel<N, true, 64, 1>), tgrid, dim3(64), 0, st, cf, tg);
            else if (r16 && wpb == 3) hipLaunchKernelGGL((box_tile_kernel<N, true, 16, 3>), tgrid, dim3(192), 0, st, cf, tg);
            else if (r16) hipLaunchKernelGGL((box_tile_kernel<N, true, 16, 4>), tgrid, dim3(256), 0, st, cf, tg);
            else hipLaunchKernelGGL((box_tile_kernel<N, true, 8, 4>), tgrid, dim3(256), 0, st, cf, tg);
            // (no second kernel: bound by its stores, the tile kernel has the vector instructions to spare)
        }
        return 0;
    }
    if (li.cull_buf && !tg.colors_out && BoxRows<N>::value > 1) {
        const int ncols = (tg.width + 63) / 64;
        tg.redo_words = (ncols + 31) / 32;
        tg.cull_words = 4 * tg.redo_words;
        tg.redo = li.cull_buf + ((size_t)li.nframes * tg.row_count + 16) * tg.cull_words;        // 16 rows of padding after the codes
        hipLaunchKernelGGL(box_cull_kernel<N>, dim3((unsigned)tg.redo_words, (unsigned)((tg.row_count + 7) / 8), (unsigned)li.nframes), dim3(256), 0,
                           (hipStream_t)li.stream, cf, tg, li.cull_buf, ncols);
        tg.cull = li.cull_buf;
    }
    // the common packed-RGB formats get the kernel with the format tests compiled out; it leaves the stretches that
    // need the reference's face-by-face arithmetic to a second, small launch (it needs the bitmaps for that)
    if ((tg.redo || BoxRows<N>::value == 1) && tg.plain_bits != 0u && tg.plain_bits <= 10u && tg.bpp == 4 && tg.aligned4 &&
        !tg.colors_out) {
        // sixteen rows a lane once there are waves to spare (the per-wave set-up is a fifth of the work at eight)
        const long long waves8 = (long long)grid.x * grid.y * grid.z * 4;
        if (BoxRows<N>::value > 1 && BoxRows<N>::value < 16 && waves8 >= 64 * 1024) {
            grid_for(tg, 64, 4 * 16, li.nframes, grid);
            hipLaunchKernelGGL((box_kernel<N, true, (BoxRows<N>::value > 1 ? 16 : 1)>), grid, dim3(256), 0, (hipStream_t)li.stream, cf, tg);
        } else {
            hipLaunchKernelGGL((box_kernel<N, true>), grid, dim3(256), 0, (hipStream_t)li.stream, cf, tg);
        }
        if (BoxRows<N>::value > 1)
            hipLaunchKernelGGL((box_redo_kernel<N, false, false, 1>), dim3((unsigned)tg.redo_words, (unsigned)((tg.row_count + 3) / 4), (unsigned)li.nframes),
                               dim3(256), 0, (hipStream_t)li.stream, cf, tg);
    } else {
        hipLaunchKernelGGL((box_kernel<N, false>), grid, dim3(256), 0, (hipStream_t)li.stream, cf, tg);
    }
    return 0;
}

}  // namespace
