// nt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// One ray per lane.  Two scene kernels, matching the reference's two `scene`
// implementations (src/render.hpp:8-26):
//
//   box_kernel<N> / box_kernel_var     box_scene::calculate_color        (src/tracer.hpp:101-152)
//   composite_kernel<N>                composite_scene::calculate_color  (src/tracer.hpp:1885-1890)
//
// Both fuse process_pixel's channel conversion and bit packing
// (src/render.cpp:419-462) into the epilogue, so the only HBM traffic of a frame
// is the packed framebuffer (plus, for composite scenes, the k-d nodes and
// simplex records).
//
// Arithmetic contract: identical operation order to oracle/ntracer_oracle.c,
// compiled with -ffp-contract=off, IEEE division/sqrt (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), so colours are bit-identical to the
// oracle except through powf/pow.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "nt_device.hpp"

namespace {

thread_local char g_launch_error[256] = "";

// --------------------------------------------------------------------------------------
// pixel packing: render.cpp:419-462
// --------------------------------------------------------------------------------------

// lround(v * double(maxval)) (render.cpp:439) for v in [0,1], exactly, without f64 when the double
// product is exact (bits <= 29): v = m * 2^-s, so the answer is round-half-up(m * maxval / 2^s).
__device__ __forceinline__ uint32_t quantize(float v, uint32_t maxval, uint32_t bits) {
    const uint32_t u = __float_as_uint(v);
    const uint32_t e = u >> 23;                                  // v >= 0: no sign bit
    const uint32_t m = (u & 0x7fffffu) | (e ? 0x800000u : 0u);
    const uint32_t s = (e ? 150u : 149u) - e;                    // >= 23 because v <= 1
    if (bits <= 8) {
        const uint32_t p = m * maxval;                           // < 2^32
        return s > 32u ? 0u : (((p >> (s - 1u)) + 1u) >> 1);
    }
    if (bits <= 29) {
        const uint64_t p = (uint64_t)m * maxval;                 // < 2^53
        return s > 56u ? 0u : (uint32_t)(((p >> (s - 1u)) + 1u) >> 1);
    }
    return (uint32_t)llround((double)v * (double)maxval);        // 30/31-bit channels: the f64 product rounds
}

__device__ __forceinline__ uint32_t channel_value(const NtChanDev &c, float r, float g, float b) {
    // association order of the reference build, pinned by tests/golden/packing_box3.npz (see oracle)
    float v = (c.f_g * g + c.f_b * b) + (c.f_r * r + c.f_c);
    v = v > 0.0f ? v : 0.0f;     // simd::clamp = min(max(v,0),1), SSE NaN rule
    v = v < 1.0f ? v : 1.0f;
    return c.tfloat ? __float_as_uint(v) : quantize(v, c.maxval, c.bits);
}

// generic: up to 128 bits, any channel count
__device__ __forceinline__ void pack_pixel(float r, float g, float b, const NtTarget &tg, uint64_t &hi, uint64_t &lo) {
    hi = 0;
    lo = 0;
    for (int k = 0; k < tg.nchannels; ++k) {
        const NtChanDev c = tg.chans[k];
        const uint64_t ival = channel_value(c, r, g, b);
        const int bits = (int)c.bits;
        const int off = (int)c.offset;
        const int rm = off & 63;
        const int sh = 64 - rm - bits;
        if (off < 64) {
            hi |= sh >= 0 ? ival << sh : ival >> -sh;
            if (rm + bits > 64) lo |= ival << (128 - rm - bits);
        } else {
            lo |= ival << sh;    // total <= 128 bits, so sh >= 0 here
        }
    }
}

// <= 4 live channels in one 32-bit container (RGBX8, RGB565, RGB888, ...): fully unrolled, the
// channel constants stay in SGPRs
__device__ __forceinline__ uint32_t pack_word32(float r, float g, float b, const NtTarget &tg) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < tg.nchannels) {
            const NtChanDev c = tg.chans[k];
            w |= channel_value(c, r, g, b) << (32u - c.offset - c.bits);
        }
    }
    return w;
}

__device__ __forceinline__ uint64_t pack_word64(float r, float g, float b, const NtTarget &tg) {
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < tg.nchannels) {
            const NtChanDev c = tg.chans[k];
            w |= (uint64_t)channel_value(c, r, g, b) << (64u - c.offset - c.bits);
        }
    }
    return w;
}

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

__device__ __forceinline__ uint32_t msb_byte(uint64_t hi, uint64_t lo, int j) {
    const uint64_t w = j < 8 ? hi : lo;
    return (uint32_t)(w >> ((7 - (j & 7)) * 8)) & 0xffu;
}

__device__ __forceinline__ void store_pixel(uint8_t *p, const NtTarget &tg, uint64_t hi, uint64_t lo) {
    const int bpp = tg.bpp;
    if (tg.aligned4 && (bpp & 3) == 0) {
        const uint32_t w0 = (uint32_t)(hi >> 32), w1 = (uint32_t)hi, w2 = (uint32_t)(lo >> 32), w3 = (uint32_t)lo;
        uint32_t *q = reinterpret_cast<uint32_t *>(p);
        if (!tg.reversed) {
            q[0] = bswap32(w0);
            if (bpp > 4) q[1] = bswap32(w1);
            if (bpp > 8) q[2] = bswap32(w2);
            if (bpp > 12) q[3] = bswap32(w3);
        } else {
            if (bpp == 4) { q[0] = w0; }
            else if (bpp == 8) { q[0] = w1; q[1] = w0; }
            else if (bpp == 12) { q[0] = w2; q[1] = w1; q[2] = w0; }
            else { q[0] = w3; q[1] = w2; q[2] = w1; q[3] = w0; }
        }
        return;
    }
    if (!tg.reversed) {
        for (int j = 0; j < bpp; ++j) p[j] = (uint8_t)msb_byte(hi, lo, j);
    } else {
        for (int j = 0; j < bpp; ++j) p[j] = (uint8_t)msb_byte(hi, lo, bpp - 1 - j);
    }
}

// A pixel that fits one 32-bit word but is not 4 bytes wide.  3-byte pixels (RGB24) of four neighbouring lanes --
// every image kernel puts x = ...+lane within aligned groups of 8 or 64 -- are 12 contiguous bytes: three of the
// four lanes assemble one dword each from their own and their right neighbour's pixel and store it; anything else
// (other widths, unaligned rows, a group cut by the image edge) goes out byte by byte.
__device__ __forceinline__ void store_word32_narrow(uint8_t *p, const NtTarget &tg, uint32_t w, int x) {
    if (tg.bpp == 3 && tg.aligned4 && !tg.colors_out) {
        const int lane = (int)(threadIdx.x & 63);
        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);          // lanes executing this store
        const bool whole = ((act >> (lane & ~3)) & 0xfull) == 0xfull && (x & 3) == (lane & 3);
        // the pixel's three bytes in memory order, lowest first
        const uint32_t m = tg.reversed ? (w >> 8) : (bswap32(w) & 0xffffffu);
        const uint32_t right = (uint32_t)__shfl_down((int)m, 1, 64);
        if (whole) {
            const int j = lane & 3;
            if (j < 3) {
                const uint32_t dw = (m >> (8 * j)) | (right << (24 - 8 * j));
                *reinterpret_cast<uint32_t *>(p + j) = dw;          // p = row + 3x; the group's dword j sits at row + 3*(x-j) + 4j = p + j
            }
            return;
        }
    }
    store_pixel(p, tg, (uint64_t)w << 32, 0);
}

// The same for 6-byte pixels (three 16-bit channels, the format of the reference's video export,
// scripts/polytope.py:594-599): two neighbouring lanes own 12 contiguous bytes = three dwords.
__device__ __forceinline__ void store_word64_narrow(uint8_t *p, const NtTarget &tg, uint64_t w, int x) {
    if (tg.bpp == 6 && tg.aligned4 && !tg.colors_out) {
        const int lane = (int)(threadIdx.x & 63);
        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
        const bool whole = ((act >> (lane & ~1)) & 0x3ull) == 0x3ull && (x & 1) == (lane & 1);
        // the pixel's six bytes in memory order, lowest first (w holds them MSB-first in its top 48 bits)
        const uint64_t m = tg.reversed ? (w >> 16) : (__builtin_bswap64(w) & 0xffffffffffffull);
        const uint32_t right_lo = (uint32_t)__shfl_down((int)(uint32_t)m, 1, 64);
        if (whole) {
            if ((lane & 1) == 0) {                          // p = pair base: bytes 0..7
                uint32_t *q = reinterpret_cast<uint32_t *>(p);
                q[0] = (uint32_t)m;
                q[1] = (uint32_t)(m >> 32) | (right_lo << 16);
            } else {                                        // p = pair base + 6: its bytes 2..5 are the pair's last dword
                *reinterpret_cast<uint32_t *>(p + 2) = (uint32_t)(m >> 16);
            }
            return;
        }
    }
    store_pixel(p, tg, w, 0);
}

// --------------------------------------------------------------------------------------
// pixel <-> thread mapping (worker_draw's chunking, render.cpp:468-493, becomes the grid)
// --------------------------------------------------------------------------------------
struct PixelRef {
    int x, y;
    long long offset;   // byte offset into dest, or probe index in probe mode
    long long hit_index; // record index into NtTarget::hits (image mode)
    bool valid;
};

// px,py: position inside the block's BW x BH tile
template <int BW, int BH>
__device__ __forceinline__ PixelRef locate_pixel(const NtTarget &tg, int px, int py, int tid) {
    PixelRef r;
    r.valid = false;
    r.x = 0;
    r.y = 0;
    r.offset = 0;
    r.hit_index = 0;
    if (tg.colors_out) {
        const int idx = (int)blockIdx.x * (BW * BH) + tid;
        if (idx < tg.probe_count) {
            r.x = tg.probe_xs[idx];
            r.y = tg.probe_ys[idx];
            r.offset = idx;
            r.valid = true;
        }
        return r;
    }
    const int x = (int)blockIdx.x * BW + px;
    const int row = (int)blockIdx.y * BH + py;      // relative to row_begin
    if (x >= tg.width || row >= tg.row_count) return r;
    const int orow = tg.row_begin + row;             // owned-row index
    int y = orow;
    if (tg.band_world > 1) {
        const int band = orow / tg.band_rows;
        y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
    }
    if (y >= tg.height) return r;
    r.x = x;
    r.y = y;
    r.offset = (long long)blockIdx.z * tg.frame_stride + (long long)(tg.compact ? orow : y) * tg.pitch + (long long)x * tg.bpp;
    r.hit_index = ((long long)blockIdx.z * tg.row_count + row) * tg.width + x;
    r.valid = true;
    return r;
}

__device__ __forceinline__ void emit_pixel(const NtTarget &tg, const PixelRef &pr, float r, float g, float b) {
    if (tg.colors_out) {
        float *o = tg.colors_out + 3 * pr.offset;
        o[0] = r;
        o[1] = g;
        o[2] = b;
        return;
    }
    uint8_t *p = tg.dest + pr.offset;
#ifdef NT_EXP_SKIP_PACK
    *reinterpret_cast<uint32_t *>(p) = __float_as_uint(r + g + b);
    return;
#endif
    if (tg.pack_mode == NT_PACK_WORD32) {
        const uint32_t w = pack_word32(r, g, b, tg);
        if (tg.bpp == 4 && tg.aligned4) {
            *reinterpret_cast<uint32_t *>(p) = tg.reversed ? w : bswap32(w);     // one coalesced dword per lane
            return;
        }
        store_word32_narrow(p, tg, w, pr.x);
        return;
    }
    if (tg.pack_mode == NT_PACK_WORD64) {
        store_word64_narrow(p, tg, pack_word64(r, g, b, tg), pr.x);
        return;
    }
    if (tg.plain_f32[0] >= 0 && tg.aligned4) {
        // three fp32 channels that are plain components: clamp, big-endian floats (or the reversed pixel)
        const float c[3] = {r, g, b};
        uint32_t v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float x = tg.plain_f32[k] == 0 ? c[0] : (tg.plain_f32[k] == 1 ? c[1] : c[2]);
            x = x > 0.0f ? x : 0.0f;     // simd::clamp, as in channel_value
            x = x < 1.0f ? x : 1.0f;
            v[k] = __float_as_uint(x);
        }
        uint32_t *q = reinterpret_cast<uint32_t *>(p);
        if (!tg.reversed) { q[0] = bswap32(v[0]); q[1] = bswap32(v[1]); q[2] = bswap32(v[2]); }
        else { q[0] = v[2]; q[1] = v[1]; q[2] = v[0]; }
        return;
    }
    uint64_t hi, lo;
    pack_pixel(r, g, b, tg, hi, lo);
    store_pixel(p, tg, hi, lo);
}

// BoxScene colours always have g == b (tracer.hpp:107-113: shade*(1,.5,.5), (i,i,i) or (0,-i,-i)).  For plain RGB
// layouts in one aligned dword the channel value is the component itself -- (0*g + 0*b) + (1*r + 0) == r -- so
// the G and B fields share one quantisation; same bits as emit_pixel, fewer instructions.
__device__ __forceinline__ bool plain_rgb(const NtTarget &tg) {
    return tg.plain_bits != 0u && tg.bpp == 4 && tg.aligned4 && !tg.colors_out;
}
__device__ __forceinline__ uint32_t plain_quantize(const NtTarget &tg, float v) {
    v = v > 0.0f ? v : 0.0f;     // simd::clamp, as in channel_value
    v = v < 1.0f ? v : 1.0f;
    return quantize(v, tg.plain_maxval, tg.plain_bits);
}
__device__ __forceinline__ void emit_plain(const NtTarget &tg, const PixelRef &pr, uint32_t qr, uint32_t qgb) {
    if (tg.plain_sel != 0u) {
        // 8-bit fields on byte boundaries: one v_perm_b32 puts the two values where they go, in memory order
        *reinterpret_cast<uint32_t *>(tg.dest + pr.offset) = __builtin_amdgcn_perm(qr, qgb, tg.plain_sel);
        return;
    }
    const uint32_t w = qr * tg.plain_mul[0] + qgb * (tg.plain_mul[1] + tg.plain_mul[2]);
    *reinterpret_cast<uint32_t *>(tg.dest + pr.offset) = tg.reversed ? w : bswap32(w);
}

// --------------------------------------------------------------------------------------
// BoxScene, compile-time N (fixed_geometry.hpp -> registers)
// --------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void load_camera(const NtCameraFixed &cam, float (&org)[N], float (&right)[N], float (&up)[N], float (&fwd)[N]) {
    if (cam.buf) {
        const float *c = cam.buf + (size_t)blockIdx.z * 4 * N;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            org[j] = c[j];
            right[j] = c[N + j];
            up[j] = c[2 * N + j];
            fwd[j] = c[3 * N + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            org[j] = cam.inl[j];
            right[j] = cam.inl[N + j];
            up[j] = cam.inl[2 * N + j];
            fwd[j] = cam.inl[3 * N + j];
        }
    }
}

// flat_origin_ray_source::operator() (tracer.hpp:71-75)
template <int N>
__device__ __forceinline__ void primary_dir(const NtTarget &tg, const float (&right)[N], const float (&up)[N], const float (&fwd)[N],
                                            int x, int y, float (&dir)[N]) {
    const float sx = tg.fovI * ((float)x - tg.half_w);
    const float sy = tg.fovI * ((float)y - tg.half_h);
#pragma unroll
    for (int j = 0; j < N; ++j) dir[j] = (fwd[j] + right[j] * sx) - up[j] * sy;
    float sq = dir[0] * dir[0];
#pragma unroll
    for (int j = 1; j < N; ++j) sq = sq + dir[j] * dir[j];
    const float len = sqrtf(sq);
#pragma unroll
    for (int j = 0; j < N; ++j) dir[j] = dir[j] / len;
}

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
// here (~2^-22*(1 + |o_j|)): under 2e-6*(1 + max|o_j|) together, 15x below this.
#ifndef NT_BOX_MARGIN
#define NT_BOX_MARGIN 3e-5f
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
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float inv = __builtin_amdgcn_rcpf(v[j]);
        const float a = (-1.0f - o[j]) * inv, b = (1.0f - o[j]) * inv;
        const float nr = fminf(a, b), fr = fmaxf(a, b);     // a NaN (v_j = 0 and o_j = -+1, or 0*inf) drops out
        near[j] = nr;
        const float w = m * fabsf(inv);
        tnp = fmaxf(tnp, nr - w);
        tfp = fminf(tfp, fr + w);
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
        const float nr = fminf((-1.0f - o[j]) * inv, (1.0f - o[j]) * inv);
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
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const bool early = (tn - near[j]) * aK > m;                                     // false for a NaN
        const bool robust = fmaf(fabsf(v[j]), slack, fabsf(fmaf(v[j], tn, o[j]))) <= lim;  // false for a NaN
        inT[j] = unclear && !early;
        inC[j] = unclear && !(robust && early);
        d[j] = 0.0f;
        if (__builtin_amdgcn_ballot_w64(inC[j]) != 0ull) d[j] = v[j] / len;
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
                if (j != i) {
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

// One pixel of BoxScene from the unnormalised direction `dir` (|dir|^2 = sq), sx / sy as in the ray source.
// PLAIN: the format is known to be plain_rgb with at most 10 bits per channel (the launcher checks).
// DEFER: a wave with an unclear lane writes nothing and returns false (the caller hands the stretch to box_redo_kernel),
// which keeps the resolving code -- and its registers -- out of the kernel every other wave runs.
// REDO (box_redo_kernel): no sorting into clear and unclear -- box_resolve is complete by itself (a clear hit is T = C =
// {K}; a clear miss fails at a coordinate of C), and in a stretch that is here because of its unclear lanes the
// sorting of the others saves nothing.
template <int N, bool PLAIN, bool DEFER = false, bool REDO = false>
__device__ __forceinline__ bool box_pixel(const NtTarget &tg, const PixelRef &pr, const float (&org)[N], float (&dir)[N], float sq,
                                          const float (&dots)[4], float sx, float sy, float margin, bool rowhit = true) {
    // rowhit (wave-uniform): the culling bit of this 64-pixel stretch of the row, see box_cull_kernel
    const bool maybe = REDO || (rowhit && box_may_hit(N, dots, sx, sy, sq));
    float r, g, b;
    bool hit = false, unclear = false;
    float x = dir[0];
#ifndef NT_EXP_NOCLASSIFY
    float near[N], tn = 0.0f, vK = 0.0f;
    if (REDO) {
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
            box_resolve<N>(org, dir, sqrtf(sq), margin, near, tn, vK, unclear, hit, x);
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
template <int N> struct BoxRows { static constexpr int value = N <= 10 ? NT_BOXROWS : 1; };
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
        const float *dp = cam.buf + (size_t)gridDim.z * 4 * N + (size_t)blockIdx.z * 4;
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
template <int N>
__global__ __launch_bounds__(256) void box_redo_kernel(NtCameraFixed cam, NtTarget tg) {
    const int tid = (int)threadIdx.x;
    const int row = (int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(tid >> 6);
    if (row >= tg.row_count) return;
    uint32_t todo = tg.redo[((size_t)blockIdx.z * tg.row_count + row) * tg.redo_words + blockIdx.x];
    if (todo == 0u) return;
    float org[N], right[N], up[N], fwd[N], dir[N];
    load_camera<N>(cam, org, right, up, fwd);
    float margin = fabsf(org[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) margin = fmaxf(margin, fabsf(org[j]));
    margin = NT_BOX_MARGIN * (1.0f + margin);
    float dots[4];
    if (cam.buf) {
        const float *dp = cam.buf + (size_t)gridDim.z * 4 * N + (size_t)blockIdx.z * 4;
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
        box_pixel<N, true, false, true>(tg, pr, org, dir, sq, dots, sx, sy, margin);
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
        float omax = fabsf(org[0]);
#pragma unroll
        for (int j = 1; j < N; ++j) omax = fmaxf(omax, fabsf(org[j]));
        const float m = NT_BOX_MARGIN * (1.0f + omax);
        const float h = 1.0f + 2.0f * m + 1e-3f;
        const float sxc = tg.fovI * (((float)(col * 64) + 31.5f) - tg.half_w);
        const float sy = tg.fovI * ((float)y - tg.half_h);
        const float spread = 32.0f * tg.fovI;
        float tlo = 0.0f, thi = INFINITY;
        bool dead = false;
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
            // pa*tau >= qa bounds tau from below when pa > 0, from above when pa < 0; pb*tau <= qb the other way round
            // (a NaN -- 0*inf -- drops out of fmaxf / fminf)
            const float lo_a = pa > 0.0f ? ra : -INFINITY, hi_a = pa < 0.0f ? ra : INFINITY;
            const float hi_b = pb > 0.0f ? rb : INFINITY, lo_b = pb < 0.0f ? rb : -INFINITY;
            tlo = fmaxf(tlo, fmaxf(lo_a, lo_b));
            thi = fminf(thi, fminf(hi_a, hi_b));
            dead = dead || (pa == 0.0f && qa > 0.0f) || (pb == 0.0f && qb < 0.0f);
            // the middle ray's entry into slab j (any K is verified below, so accuracy only matters for the yield)
            const float nr = ((vc[j] < 0.0f ? 1.0f : -1.0f) - org[j]) * __builtin_amdgcn_rcpf(vc[j]);
            tn2 = __builtin_amdgcn_fmed3f(tn, tn2, nr);         // second-to-last entry
            const bool later = nr > tn;
            vK = later ? vc[j] : vK;
            gK = later ? g[j] : gK;
            oK = later ? org[j] : oK;
            K = later ? j : K;
            tn = fmaxf(tn, nr);
        }
        if (!(dead || tlo > thi)) {                      // a NaN keeps the stretch
            code = 15u;
            const float vKa = vK - gK, vKb = vK + gK;
            if (N <= 14 && vKa * vKb > 0.0f) {
                const float num = (vK < 0.0f ? 1.0f : -1.0f) - oK;
                const float t1 = num * __builtin_amdgcn_rcpf(vKa), t2 = num * __builtin_amdgcn_rcpf(vKb);
                const float t_lo = fminf(t1, t2) * (1.0f - 1e-6f), t_hi = fmaxf(t1, t2) * (1.0f + 1e-6f);
                const float rK = m * __builtin_amdgcn_rcpf(fminf(fabsf(vKa), fabsf(vKb))) * (1.0f + 1e-6f);
                bool ok = t_lo > 1e-3f && t_hi < 1e30f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float va = vc[j] - g[j], vb = vc[j] + g[j];
                    const float pmax = org[j] + fmaxf(vb * t_lo, vb * t_hi);
                    const float pmin = org[j] + fminf(va * t_lo, va * t_hi);
                    const float lim = (1.0f - m - 1e-4f) - fmaxf(fabsf(va), fabsf(vb)) * rK;
                    ok = ok && (j == K || (pmax <= lim && pmin >= -lim));
                }
                if (ok) code = (uint32_t)K + 1u;
            }
            // The middle ray enters two slabs within m/|v_K| of each other: box_classify would call the rays around it
            // unclear and box_kernel would hand the stretch to box_redo_kernel after classifying all of it -- send it
            // there directly (code 14; only a prediction: box_redo_kernel is right for any stretch).
            if (code == 15u && !((tn - tn2) * fabsf(vK) > m)) code = 14u;
        }
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
// BoxScene, run-time n (var_geometry.hpp -> per-lane n-vector in LDS, [j][lane] so that a
// wave's accesses to component j hit 64 consecutive banks)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_kernel_var(NtCamera cam, NtTarget tg) {
    extern __shared__ float lds_dir[];    // [n][256] per-lane direction, then [4][n] camera rows (broadcast reads)
    const int tid = (int)threadIdx.x;
    const int n = cam.n;
    float *camrow = lds_dir + (size_t)n * 256;
    {
        // stage the camera rows once per block: run-time-indexed kernel arguments would be one scalar load each
        const float *src = cam.buf ? cam.buf + (size_t)blockIdx.z * 4 * n : nullptr;
        for (int k = tid; k < 4 * n; k += 256) camrow[k] = src ? src[k] : cam.inl[k];
    }
    __syncthreads();
    const PixelRef pr = locate_pixel<64, 4>(tg, tid & 63, tid >> 6, tid);
    if (!pr.valid) return;
    const float *c = camrow;              // origin, right, up, forward
    float *dir = lds_dir + tid;           // dir[j] at dir[j*256]
    const float sx = tg.fovI * ((float)pr.x - tg.half_w);
    const float sy = tg.fovI * ((float)pr.y - tg.half_h);
    float sq = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float v = (c[3 * n + j] + c[n + j] * sx) - c[2 * n + j] * sy;
        dir[j * 256] = v;
        sq = j == 0 ? v * v : sq + v * v;
    }
    const float len = sqrtf(sq);

    bool done = false;
    float shade = 0.0f;
    // Same exact pruning as box_color<N> (see there): circumsphere rejection per wave on the unnormalised
    // direction (box_may_hit), then only the faces in a near-tie with the last-reached candidate K get the
    // division and the n-1 checks.  Waves that cannot hit normalise dir[0] only.
    float dots[4];
    if (cam.buf) {
        const float *dp = cam.buf + (size_t)gridDim.z * 4 * n + (size_t)blockIdx.z * 4;
        dots[0] = dp[0]; dots[1] = dp[1]; dots[2] = dp[2]; dots[3] = dp[3];
    } else {
        dots[0] = cam.odots[0]; dots[1] = cam.odots[1]; dots[2] = cam.odots[2]; dots[3] = cam.odots[3];
    }
    const float osq = dots[0];
    const float ov = fmaf(-dots[2], sy, fmaf(dots[1], sx, dots[3]));
    const float rad2 = (float)n * (1.0f + NT_FUZZ) * (1.0f + NT_FUZZ) * 1.001f;
    const bool maybe = !((osq - rad2 * 1.0001f - 1e-5f * osq) * sq > ov * ov * 1.0001f);   // FMA rounding covered by the margins
    const bool wave_maybe = __builtin_amdgcn_ballot_w64(maybe) != 0ull;
    if (wave_maybe) {
        for (int j = 0; j < n; ++j) dir[j * 256] = dir[j * 256] / len;
    } else {
        dir[0] = dir[0] / len;
    }
    if (wave_maybe) {
        float aK = 0.0f, bK = 1.0f, oK = 0.0f;
        bool any = false;
        for (int i = 0; i < n; ++i) {
            const float di = dir[i * 256];
            const float oi = c[i];
            const float num = (di < 0.0f ? 1.0f : -1.0f) - oi;
            const bool pre = maybe && ((num > 0.0f && di > 0.0f) || (num < 0.0f && di < 0.0f));
            const float a = fabsf(num), bb = fabsf(di);
            if (pre && (!any || a * bK > aK * bb)) { aK = a; bK = bb; oK = oi; any = true; }
        }
        const float mu = 1e-4f * (1.0f + fabsf(oK));
        const float aKm = (aK - mu) * (1.0f - 1e-6f);
        for (int i = 0; i < n; ++i) {
            const float di = dir[i * 256];
            const float oi = c[i];
            const float s = di < 0.0f ? 1.0f : -1.0f;
            const float num = s - oi;
            const bool pre = maybe && ((num > 0.0f && di > 0.0f) || (num < 0.0f && di < 0.0f));
            const bool tie = pre && !done && !(fabsf(num) * bK < fabsf(di) * aKm);
            if (__builtin_amdgcn_ballot_w64(tie) == 0ull) continue;
            const float dist = num / di;
            bool ok = tie && dist > 0.0f;
            for (int j = 0; j < n; ++j) {
                if (j != i) {
                    const float oj = c[j];
                    const float p = dir[j * 256] * dist + oj;
                    ok = ok && !(fabsf(p) > (1.0f + NT_FUZZ));
                }
            }
            if (ok) {
                done = true;
                if (dist >= FLT_MAX) shade = -1.0f;
                else {
                    const float sine = di * s;
                    shade = sine <= 0.0f ? -sine : 0.0f;
                }
            }
        }
    }
    float r, g, b;
    if (done && shade >= 0.0f) {
        r = shade * 1.0f;
        g = shade * 0.5f;
        b = shade * 0.5f;
    } else {
        const float in = dir[0];
        if (in > 0.0f) { r = in; g = in; b = in; }
        else { r = 0.0f; g = -in; b = -in; }
    }
    if (plain_rgb(tg)) {
        emit_plain(tg, pr, plain_quantize(tg, r), plain_quantize(tg, g));            // g == b
        return;
    }
    emit_pixel(tg, pr, r, g, b);
}

// --------------------------------------------------------------------------------------
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
template <int N, bool FEAT, bool STATS>
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
        } else if (FEAT && item != skip_item) {
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

template <int N, bool FEAT, bool STATS>
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
                if (leaf_closest<N, FEAT, STATS>(sc, w, lane, nd.left, nd.right, o, d, skip_item, skip_lane, hit, st)) dirty = sp;
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
template <int N, bool STATS>
__device__ __forceinline__ bool leaf_occludes(const NtCompositeDev &sc, int start, int count, const float (&o)[N], const float (&d)[N],
                                              float ldistance, int skip_item, int skip_lane, Stats &st) {
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
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
        } else if (item != skip_item) {
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
template <int N, bool STATS>
__device__ __noinline__ bool trace_occluded(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                            float ldistance, int skip_item, int skip_lane, Stats &st) {
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
                if (leaf_occludes<N, STATS>(sc, nd.left, nd.right, o, d, ldistance, skip_item, skip_lane, st)) return true;
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
template <int N, bool FEAT>
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
template <int N, bool FEAT, bool STATS>
__device__ __forceinline__ Color3 composite_color(const NtCompositeDev &sc, const WaveLds &w, int lane, float (&o)[N], float (&d)[N], Stats &st,
                                                  const Hit *primary = nullptr) {
    Level levels[FEAT ? NT_DEV_MAX_REFLECT : 1];
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
                found = trace_closest<N, FEAT, STATS>(sc, w, lane, o, d, dist, FLT_MAX, skip_item, skip_lane, hit, st);
            }
        }
        if (!found) {
            result = background_color<N>(sc, d);
            break;
        }
        if (STATS && depth == 0) st.hits += 1;

        // ---- base_color (tracer.hpp:1768-1854) ----
        float no[N], nd[N];
        hit_normal<N, FEAT>(sc, hit, o, d, no, nd);
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
                    const float strength = (float)(1.0 / pow((double)ldist, (double)(N - 1)));
                    if (sc.shadows) {
                        if (fmaxf(plc.r, fmaxf(plc.g, plc.b)) * strength * sine > NT_LIGHT_THRESHOLD) {
                            if (!trace_occluded<N, STATS>(sc, w, lane, no, lv, ldist, hit.item, hit.lane, st)) {
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
                        if (!trace_occluded<N, STATS>(sc, w, lane, no, neg, FLT_MAX, hit.item, hit.lane, st)) {
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

        if (FEAT && m[7] != 0.0f && depth < sc.max_reflect_depth && depth < NT_DEV_MAX_REFLECT) {
            Level &L = levels[depth];
            L.spec = specular;
            L.spec_a = spec_a;
            L.r0 = r0;
            L.c = c3p(m);
            L.refl = m[7];
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

// kd_leaf<Store,true>::intersects (tracer.hpp:977-1086) with transparent hits: first loop until an opaque
// hit, then "is there anything closer?", then trim_intersections with the LAST test's dist (:1084).
template <int N>
__device__ __noinline__ bool leaf_closest_t(const NtCompositeDev &sc, const WaveLds &w, int lane, int start, int count, const float (&o)[N],
                                            const float (&d)[N], int skip_item, int skip_lane, Hit &hit, TList &th) {
    const int h_start = th.n;
    bool found = false;
    float dist_last = 0.0f;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        if ((item & 3) != 0 && item == skip_item) continue;
        if (mbox_seen(w, lane, item)) continue;
        int l;
        const float t = test_item<N>(sc, item, o, d, hit.dist, skip_item, skip_lane, l);
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
template <int N>
__device__ __noinline__ bool trace_closest_t(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                             float t_near, int skip_item, int skip_lane, Hit &hit, TList &th) {
    hit.dist = FLT_MAX;
    hit.item = -1;
    hit.lane = -1;
    th.n = 0;
    mbox_reset(w, lane);
    int node = sc.root;
    int sp = 0;
    int dirty = 0;
    float t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (leaf_closest_t<N>(sc, w, lane, nd.left, nd.right, o, d, skip_item, skip_lane, hit, th)) dirty = sp;
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
};

template <int N>
__device__ __noinline__ Color3 composite_color_t(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&org)[N], const float (&dir)[N]) {
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
            const float dist = aabb_distance<N>(sc, o, d);
            if (dist >= 0.0f) {
                setup_ray_table<N>(w, lane, o, d);
                trace_closest_t<N>(sc, w, lane, o, d, dist, F.skip_item, F.skip_lane, hit, th);
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
            hit_normal<N, true>(sc, hit, o, d, no, nd);
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
                    const float strength = (float)(1.0 / pow((double)ldist, (double)(N - 1)));
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

template <int N>
__global__ __launch_bounds__(256) void composite_kernel_t(NtCameraFixed cam, NtCompositeDev sc, NtTarget tg) {
    extern __shared__ float2 lds_raw[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const WaveLds w = wave_lds(reinterpret_cast<char *>(lds_raw), wv, sc.stack_depth, N);
    int px, py;
    if (tg.colors_out) { px = 0; py = 0; }
    else { px = (wv & 1) * 8 + (lane & 7); py = (wv >> 1) * 8 + (lane >> 3); }
    const PixelRef pr = locate_pixel<16, 16>(tg, px, py, tid);
    if (pr.valid) {
        float org[N], right[N], up[N], fwd[N], dir[N];
        load_camera<N>(cam, org, right, up, fwd);
        primary_dir<N>(tg, right, up, fwd, pr.x, pr.y, dir);
        const Color3 c = composite_color_t<N>(sc, w, lane, org, dir);
        emit_pixel(tg, pr, c.r, c.g, c.b);
    }
}

// --------------------------------------------------------------------------------------
// CompositeScene, run-time n (9..64): the var_geometry.hpp path.  Per-lane kernel; the ray's n-vectors
// (origin, direction, 1/direction, scratch) live in LDS as [k][lane], simplex records are read from global
// memory component by component.  Feature set of the scripted configurations: batches and unbatched
// triangles, opaque, camera light; anything else is refused by the host for n > 8.  Operation order is the
// oracle's, so results are identical to the fixed-N kernels' where both exist.
// --------------------------------------------------------------------------------------
struct VarLds {
    float2 *ray;     // [n][64] (origin, 1/direction)
    float *dv;       // [n][64] direction
    float *ps;       // [n][64] scratch: pside / camera rows
    int *stack;      // [depth][64]
    int *mbox;       // [NT_MBOX][64]
};

__device__ __forceinline__ size_t var_lds_bytes(int depth, int n) {
    return (size_t)64 * ((size_t)n * 16 + (size_t)depth * 4 + (size_t)NT_MBOX * 4);
}

// triangle_batch::intersects lane / triangle::intersects with run-time n (tracer.hpp:411-440, 561-581)
__device__ __forceinline__ float simplex_var(const float *__restrict__ rec, int n, const VarLds &L, int lane, bool scalar_form, float cutoff) {
    float denom = 0.0f, no = 0.0f;
    for (int k = 0; k < n; ++k) {
        const float fn = rec[1 + k];
        const float pd = fn * L.dv[k * 64 + lane];
        const float po = fn * L.ray[k * 64 + lane].x;
        denom = k == 0 ? pd : denom + pd;
        no = k == 0 ? po : no + po;
    }
    if (scalar_form && denom == 0.0f) return 0.0f;
    const float t = -(no + rec[0]) / denom;
    if (scalar_form && (t <= 0.0f || t >= cutoff)) return 0.0f;
    bool ok = scalar_form ? true : (denom != 0.0f && t >= 0.0f);
    for (int k = 0; k < n; ++k) L.ps[k * 64 + lane] = rec[1 + n + k] - (L.ray[k * 64 + lane].x + t * L.dv[k * 64 + lane]);
    float tot = 0.0f;
    for (int e = 0; e < n - 1; ++e) {
        const float *en = rec + 1 + 2 * n + (size_t)e * n;
        float area = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float p = en[k] * L.ps[k * 64 + lane];
            area = k == 0 ? p : area + p;
        }
        if (scalar_form) ok = ok && !(area < -NT_FUZZ || area > (1.0f + NT_FUZZ));
        else ok = ok && area >= -NT_FUZZ;
        tot += area;
    }
    ok = ok && tot <= (1.0f + NT_FUZZ);
    return ok ? t : 0.0f;
}

__global__ __launch_bounds__(64) void composite_kernel_var(NtCamera cam, NtCompositeDev sc, NtTarget tg, int n) {
    extern __shared__ float2 lds_raw[];
    const int lane = (int)threadIdx.x;
    const int depth = sc.stack_depth;
    VarLds L;
    {
        char *p = reinterpret_cast<char *>(lds_raw);
        L.ray = reinterpret_cast<float2 *>(p);
        L.dv = reinterpret_cast<float *>(p + (size_t)64 * n * 8);
        L.ps = L.dv + (size_t)64 * n;
        L.stack = reinterpret_cast<int *>(L.ps + (size_t)64 * n);
        L.mbox = L.stack + (size_t)64 * depth;
    }
    WaveLds w;
    w.ray = L.ray;
    w.stack = L.stack;
    w.mbox = L.mbox;
    const PixelRef pr = tg.colors_out ? locate_pixel<8, 8>(tg, 0, 0, lane) : locate_pixel<8, 8>(tg, lane & 7, lane >> 3, lane);
    if (!pr.valid) return;
    const float *c = cam.buf ? cam.buf + (size_t)blockIdx.z * 4 * n : nullptr;

    // ---- primary ray (tracer.hpp:60-76), direction into LDS
    const float sx = tg.fovI * ((float)pr.x - tg.half_w);
    const float sy = tg.fovI * ((float)pr.y - tg.half_h);
    float sq = 0.0f;
    for (int k = 0; k < n; ++k) {
        const float rk = c ? c[n + k] : cam.inl[n + k];
        const float uk = c ? c[2 * n + k] : cam.inl[2 * n + k];
        const float fk = c ? c[3 * n + k] : cam.inl[3 * n + k];
        const float v = (fk + rk * sx) - uk * sy;
        L.dv[k * 64 + lane] = v;
        sq = k == 0 ? v * v : sq + v * v;
    }
    const float len = sqrtf(sq);
    for (int k = 0; k < n; ++k) {
        const float dk = L.dv[k * 64 + lane] / len;
        L.dv[k * 64 + lane] = dk;
        const float ok_ = c ? c[k] : cam.inl[k];
        L.ray[k * 64 + lane] = make_float2(ok_, dk != 0.0f ? 1.0f / dk : __int_as_float(0x7fc00000));
    }

    // ---- aabb_distance (tracer.hpp:1892-1918)
    float dist0 = -1.0f;
    for (int i = 0; i < n && dist0 < 0.0f; ++i) {
        const float di = L.dv[i * 64 + lane];
        if (di == 0.0f) continue;
        const float oi = L.ray[i * 64 + lane].x;
        const float face = di > 0.0f ? sc.aabb[i] : sc.aabb[n + i];
        float dist = (face - oi) / di;
        int skip = i;
        if (dist < 0.0f) { dist = 0.0f; skip = -1; }
        bool ok = true;
        for (int j = 0; j < n; ++j) {
            if (j != skip) {
                const float p = L.dv[j * 64 + lane] * dist + L.ray[j * 64 + lane].x;
                if (p >= sc.aabb[n + j] || p <= sc.aabb[j]) { ok = false; break; }
            }
        }
        if (ok) dist0 = dist;
    }

    Hit hit;
    hit.dist = FLT_MAX; hit.item = -1; hit.lane = -1;
    if (dist0 >= 0.0f) {
        // ---- kd_node_intersection (same continuation stack as trace_closest)
        mbox_reset(w, lane);
        int node = sc.root, sp = 0, dirty = 0;
        float t_near = dist0, t_far = FLT_MAX;
        for (;;) {
            while (node >= 0) {
                if (sc.prune && nt_beyond_hit(hit.dist, t_near)) { node = -1; break; }
                const NtNode nd = sc.nodes[node];
                if (nd.axis < 0) {
                    bool improved = false;
                    for (int i = 0; i < nd.right; ++i) {
                        const int item = sc.items[nd.left + i];
                        if (mbox_seen(w, lane, item)) continue;
                        const int kind = item & 3, idx = item >> 2;
                        if (kind == 0) {
                            float min_t = hit.dist;
                            int r = -1;
                            for (int l = 0; l < NT_DEV_BATCH; ++l) {
                                const float t = simplex_var(sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + l) * sc.rec_stride, n, L, lane, false, 0.0f);
                                if (t != 0.0f && t < min_t) { min_t = t; r = l; }
                            }
                            if (r >= 0) { hit.dist = min_t; hit.item = item; hit.lane = r; improved = true; }
                        } else {
                            const float t = simplex_var(sc.tri_recs + (size_t)idx * sc.rec_stride, n, L, lane, true, hit.dist);
                            if (t != 0.0f) { hit.dist = t; hit.item = item; hit.lane = -1; improved = true; }
                        }
                    }
                    if (improved) dirty = sp;
                    node = -1;
                    break;
                }
                const float2 oi = L.ray[nd.axis * 64 + lane];
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
                        if (sp < depth) { L.stack[sp * 64 + lane] = node; ++sp; }
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
                const NtNode nd = sc.nodes[L.stack[sp * 64 + lane]];
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
                    const NtNode up = sc.nodes[L.stack[(sp - 1) * 64 + lane]];
                    bool g2;
                    t_far = branch_t(w, lane, up, g2);
                }
                resumed = true;
                break;
            }
            if (!resumed) break;
        }
    }

    // ---- shading: ray_color's miss branch / base_color with the camera light (tracer.hpp:1829-1853, 1866)
    Color3 col;
    if (hit.item < 0) {
        const float iv = L.dv[sc.bg_axis * 64 + lane];
        col = iv >= 0.0f ? cadd(cscale(c3p(sc.bg1), iv), cscale(c3p(sc.bg2), 1.0f - iv))
                         : cadd(cscale(c3p(sc.bg3), -iv), cscale(c3p(sc.bg2), 1.0f + iv));
    } else {
        const int kind = hit.item & 3, idx = hit.item >> 2;
        const float *rec = kind == 0 ? sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + hit.lane) * sc.rec_stride
                                     : sc.tri_recs + (size_t)idx * sc.rec_stride;
        float denom = 0.0f, fsq = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float fn = rec[1 + k];
            const float pd = fn * L.dv[k * 64 + lane];
            denom = k == 0 ? pd : denom + pd;
            fsq = k == 0 ? fn * fn : fsq + fn * fn;
        }
        const float flen = sqrtf(fsq);
        float dn = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float u = rec[1 + k] / flen;
            const float ndk = denom > 0.0f ? -u : u;
            const float p = L.dv[k * 64 + lane] * ndk;
            dn = k == 0 ? p : dn + p;
        }
        const float sine = -dn;
        const float *m = material_of(sc, hit.item, hit.lane);
        Color3 light = c3(0.0f, 0.0f, 0.0f), specular = c3(0.0f, 0.0f, 0.0f);
        float spec_a = 0.0f;
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
        col = cadd(specular, cscale(r0, 1.0f - spec_a));
    }
    emit_pixel(tg, pr, col.r, col.g, col.b);
}

__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// One wave renders an 8x8 pixel tile (coherent rays -> shared k-d path and broadcast record loads);
// a 256-thread block covers 16x16 pixels.
template <int N, bool FEAT, bool STATS>
__global__ __launch_bounds__(256) void composite_kernel(NtCameraFixed cam, NtCompositeDev sc, NtTarget tg) {
    extern __shared__ float2 lds_raw[];
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
            if (emit) c = composite_color<N, FEAT, STATS>(sc, w, lane, org, dir, st, &hit);
        } else {
            c = composite_color<N, FEAT, STATS>(sc, w, lane, org, dir, st);
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
template <int N, int DEPTH, bool FEAT, bool SCAL>
__global__ __launch_bounds__(256, FEAT ? 1 : ((N <= 4 && !SCAL) ? 6 : (N <= 7 ? 5 : 4))) void composite_packet(NtCompositeDev sc, NtTarget tg, PacketArgs pa) {
    extern __shared__ float2 lds_raw[];
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

template <typename T>
void set_error(const char *what, T err) {
    snprintf(g_launch_error, sizeof(g_launch_error), "%s: %s", what, hipGetErrorString((hipError_t)err));
}

void grid_for(const NtTarget &tg, int bw, int bh, int nframes, dim3 &grid) {
    if (tg.colors_out) {
        grid = dim3((unsigned)((tg.probe_count + bw * bh - 1) / (bw * bh)), 1, 1);
    } else {
        grid = dim3((unsigned)((tg.width + bw - 1) / bw), (unsigned)((tg.row_count + bh - 1) / bh), (unsigned)nframes);
    }
}

template <int N>
int launch_box_fixed(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg_in) {
    NtCameraFixed cf;
    cf.buf = cam.buf;
    for (int k = 0; k < 4; ++k) cf.odots[k] = cam.odots[k];
    cf.n = N;
    for (int k = 0; k < 4 * N; ++k) cf.inl[k] = cam.inl[k];
    NtTarget tg = tg_in;
    dim3 grid;
    grid_for(tg, 64, tg.colors_out ? 4 : 4 * BoxRows<N>::value, li.nframes, grid);
    tg.cull = nullptr;
    tg.redo = nullptr;
    tg.cull_words = 0;
    tg.redo_words = 0;
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
            hipLaunchKernelGGL(box_redo_kernel<N>, dim3((unsigned)tg.redo_words, (unsigned)((tg.row_count + 3) / 4), (unsigned)li.nframes),
                               dim3(256), 0, (hipStream_t)li.stream, cf, tg);
    } else {
        hipLaunchKernelGGL((box_kernel<N, false>), grid, dim3(256), 0, (hipStream_t)li.stream, cf, tg);
    }
    return 0;
}

template <int N>
int launch_composite_fixed(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) {
    NtCameraFixed cf;
    cf.buf = cam.buf;
    for (int k = 0; k < 4; ++k) cf.odots[k] = cam.odots[k];
    cf.n = N;
    for (int k = 0; k < 4 * N; ++k) cf.inl[k] = cam.inl[k];
    dim3 grid;
    grid_for(tg, 16, 16, li.nframes, grid);
    const size_t lds = (size_t)4 * 64 * ((size_t)sc.stack_depth * 4 + (size_t)N * 8 + (size_t)NT_MBOX * 4);
    if (lds > 160 * 1024) {
        snprintf(g_launch_error, sizeof(g_launch_error), "k-d tree too deep for the LDS traversal stack (depth %d)", sc.stack_depth);
        return -1;
    }
    const bool feat = sc.n_point_lights || sc.n_global_lights || sc.any_reflective || sc.has_scalar_prims;
    hipStream_t s = (hipStream_t)li.stream;
    if (!sc.all_opaque) {
        hipLaunchKernelGGL((composite_kernel_t<N>), grid, dim3(256), lds, s, cf, sc, tg);
        return 0;
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
                hipLaunchKernelGGL((composite_kernel<N, true, false>), g2, dim3(256), lds, s, c2, sc, t2);
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
    else if (feat) hipLaunchKernelGGL((composite_kernel<N, true, false>), grid, dim3(256), lds, s, cf, sc, tg);
    else hipLaunchKernelGGL((composite_kernel<N, false, false>), grid, dim3(256), lds, s, cf, sc, tg);
    return 0;
}

int finish_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what, e);
        return -1;
    }
    return 0;
}

}  // namespace

const char *nt_launch_error() { return g_launch_error; }

// NTRACER_FORCE_VAR=1: use the run-time-n kernels for every dimension (they are the only ones above
// NT_DEV_MAX_FIXED; the switch lets tests compare them with the compile-time-N kernels on the same scene)
static bool force_var() {
    const char *e = getenv("NTRACER_FORCE_VAR");
    return e && atoi(e) != 0;
}

int nt_launch_box(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg) {
    switch (force_var() ? 0 : li.n) {
        case 3: launch_box_fixed<3>(li, cam, tg); break;
        case 4: launch_box_fixed<4>(li, cam, tg); break;
        case 5: launch_box_fixed<5>(li, cam, tg); break;
        case 6: launch_box_fixed<6>(li, cam, tg); break;
        case 7: launch_box_fixed<7>(li, cam, tg); break;
        case 8: launch_box_fixed<8>(li, cam, tg); break;
        case 9: launch_box_fixed<9>(li, cam, tg); break;
        case 10: launch_box_fixed<10>(li, cam, tg); break;
        default: {
            dim3 grid;
            grid_for(tg, 64, 4, li.nframes, grid);
            const size_t lds = ((size_t)li.n * 256 + (size_t)4 * li.n) * sizeof(float);
            hipLaunchKernelGGL(box_kernel_var, grid, dim3(256), lds, (hipStream_t)li.stream, cam, tg);
        }
    }
    return finish_launch("box kernel launch");
}

int nt_launch_composite(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) {
    int r;
    switch (force_var() ? 0 : li.n) {
        case 3: r = launch_composite_fixed<3>(li, cam, sc, tg); break;
        case 4: r = launch_composite_fixed<4>(li, cam, sc, tg); break;
        case 5: r = launch_composite_fixed<5>(li, cam, sc, tg); break;
        case 6: r = launch_composite_fixed<6>(li, cam, sc, tg); break;
        case 7: r = launch_composite_fixed<7>(li, cam, sc, tg); break;
        case 8: r = launch_composite_fixed<8>(li, cam, sc, tg); break;
        case 9: r = launch_composite_fixed<9>(li, cam, sc, tg); break;
        case 10: r = launch_composite_fixed<10>(li, cam, sc, tg); break;
        default: {
            if (li.n < 3 || li.n > NT_DEV_MAX_DIM) {
                snprintf(g_launch_error, sizeof(g_launch_error), "unsupported dimension %d", li.n);
                return -2;
            }
            if (sc.n_point_lights || sc.n_global_lights || sc.any_reflective || sc.n_solids || !sc.all_opaque) {
                snprintf(g_launch_error, sizeof(g_launch_error), "the run-time-n kernel renders opaque, non-reflective simplices lit by the camera light only");
                return -2;
            }
            // run-time-n kernel: one wave per 8x8 tile (probe mode: 64 probes per block)
            dim3 grid;
            grid_for(tg, 8, 8, li.nframes, grid);
            const size_t lds = (size_t)64 * ((size_t)li.n * 16 + (size_t)sc.stack_depth * 4 + (size_t)NT_MBOX * 4);
            if (lds > 160 * 1024) {
                snprintf(g_launch_error, sizeof(g_launch_error), "scene too deep for the LDS budget (n %d, depth %d)", li.n, sc.stack_depth);
                return -1;
            }
            hipLaunchKernelGGL(composite_kernel_var, grid, dim3(64), lds, (hipStream_t)li.stream, cam, sc, tg, li.n);
            r = 0;
        }
    }
    if (r) return r;
    return finish_launch("composite kernel launch");
}
