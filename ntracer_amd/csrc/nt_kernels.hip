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

// --------------------------------------------------------------------------------------
// pixel <-> thread mapping (worker_draw's chunking, render.cpp:468-493, becomes the grid)
// --------------------------------------------------------------------------------------
struct PixelRef {
    int x, y;
    long long offset;   // byte offset into dest, or probe index in probe mode
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
    if (tg.pack_mode == NT_PACK_WORD32) {
        const uint32_t w = pack_word32(r, g, b, tg);
        if (tg.bpp == 4 && tg.aligned4) {
            *reinterpret_cast<uint32_t *>(p) = tg.reversed ? w : bswap32(w);     // one coalesced dword per lane
            return;
        }
        store_pixel(p, tg, (uint64_t)w << 32, 0);
        return;
    }
    if (tg.pack_mode == NT_PACK_WORD64) {
        store_pixel(p, tg, pack_word64(r, g, b, tg), 0);
        return;
    }
    uint64_t hi, lo;
    pack_pixel(r, g, b, tg, hi, lo);
    store_pixel(p, tg, hi, lo);
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
// dist_i = (s_i - o_i)/d_i > 0 and |o_j + d_j*dist_i| <= 1+FUZZ for all j != i.  The same predicate is
// evaluated here, bit for bit, but in an order that lets whole waves skip work:
//   1. a ray whose distance from the centre exceeds the cube's circumradius (with a 0.1 % margin, far
//      above any rounding) cannot satisfy the predicate for any face -> waves of such rays skip all faces;
//   2. the AND over j is order-independent, so each face is first checked against ONE wave-uniform axis K
//      (the axis entered last by the wave's first candidate ray); faces that fail it for every lane --
//      all but one or two in a coherent wave -- skip the remaining N-2 checks;
//   3. faces no lane can enter (dist <= 0 for the whole wave) skip their division.
template <int N>
__device__ __forceinline__ void box_color(const float (&o)[N], const float (&dir)[N], float &r, float &g, float &b) {
    bool done = false;     // a face passed the slab test (hit, or dist >= cutoff)
    float shade = 0.0f;

    float osq = o[0] * o[0], od = o[0] * dir[0];
#pragma unroll
    for (int j = 1; j < N; ++j) { osq += o[j] * o[j]; od += o[j] * dir[j]; }
    const float rad2 = (float)N * (1.0f + NT_FUZZ) * (1.0f + NT_FUZZ) * 1.001f;
    const bool maybe = !((osq - od * od) > rad2);            // NaN -> maybe

    if (__builtin_amdgcn_ballot_w64(maybe) != 0ull) {
        float dist[N];
        bool cand[N];
        float dmax = -1.0f;
        int kmax = -1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float di = dir[i];
            const float s = di < 0.0f ? 1.0f : -1.0f;
            const float num = s - o[i];
            // dist > 0 needs a non-zero numerator with the sign of di
            const bool pre = maybe && di != 0.0f && ((num > 0.0f && di > 0.0f) || (num < 0.0f && di < 0.0f));
            dist[i] = 0.0f;
            cand[i] = false;
            if (__builtin_amdgcn_ballot_w64(pre) != 0ull) {
                dist[i] = num / di;
                cand[i] = pre && dist[i] > 0.0f;
            }
            if (cand[i] && dist[i] > dmax) { dmax = dist[i]; kmax = i; }
        }
        const unsigned long long has = __builtin_amdgcn_ballot_w64(kmax >= 0);
        if (has != 0ull) {
            const int K = __builtin_amdgcn_readlane(kmax, (int)__builtin_ctzll(has));
            float dK = dir[0], oK = o[0];
#pragma unroll
            for (int k = 1; k < N; ++k) if (K == k) { dK = dir[k]; oK = o[k]; }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                bool ok = cand[i] && !done;
                if (i != K) {
                    const float p = dK * dist[i] + oK;
                    ok = ok && !(fabsf(p) > (1.0f + NT_FUZZ));
                }
                if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        if (j != i) {
                            const float p = dir[j] * dist[i] + o[j];
                            ok = ok && !(fabsf(p) > (1.0f + NT_FUZZ));
                        }
                    }
                    if (ok) {
                        done = true;
                        // `if(dist >= cutoff) return 0` with cutoff = FLT_MAX (tracer.hpp:142): a miss
                        if (dist[i] >= FLT_MAX) shade = -1.0f;
                        else {
                            const float sine = dir[i] * (dir[i] < 0.0f ? 1.0f : -1.0f);   // dot(dir, s*e_i)
                            shade = sine <= 0.0f ? -sine : 0.0f;
                        }
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
        const float in = dir[0];
        if (in > 0.0f) { r = in; g = in; b = in; }
        else { r = 0.0f; g = -in; b = -in; }
    }
}

template <int N>
__global__ __launch_bounds__(256) void box_kernel(NtCameraFixed cam, NtTarget tg) {
    const int tid = (int)threadIdx.x;
    const PixelRef pr = locate_pixel<64, 4>(tg, tid & 63, tid >> 6, tid);
    if (!pr.valid) return;
    float org[N], right[N], up[N], fwd[N], dir[N];
    load_camera<N>(cam, org, right, up, fwd);
    primary_dir<N>(tg, right, up, fwd, pr.x, pr.y, dir);
    float r, g, b;
    box_color<N>(org, dir, r, g, b);
    emit_pixel(tg, pr, r, g, b);
}

// --------------------------------------------------------------------------------------
// BoxScene, run-time n (var_geometry.hpp -> per-lane n-vector in LDS, [j][lane] so that a
// wave's accesses to component j hit 64 consecutive banks)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_kernel_var(NtCamera cam, NtTarget tg) {
    extern __shared__ float lds_dir[];    // [n][256]
    const int tid = (int)threadIdx.x;
    const int n = cam.n;
    const PixelRef pr = locate_pixel<64, 4>(tg, tid & 63, tid >> 6, tid);
    if (!pr.valid) return;
    const float *c = cam.buf ? cam.buf + (size_t)blockIdx.z * 4 * n : nullptr;
    float *dir = lds_dir + tid;           // dir[j] at dir[j*256]
    const float sx = tg.fovI * ((float)pr.x - tg.half_w);
    const float sy = tg.fovI * ((float)pr.y - tg.half_h);
    float sq = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float rj = c ? c[n + j] : cam.inl[n + j];
        const float uj = c ? c[2 * n + j] : cam.inl[2 * n + j];
        const float fj = c ? c[3 * n + j] : cam.inl[3 * n + j];
        const float v = (fj + rj * sx) - uj * sy;
        dir[j * 256] = v;
        sq = j == 0 ? v * v : sq + v * v;
    }
    const float len = sqrtf(sq);
    for (int j = 0; j < n; ++j) dir[j * 256] = dir[j * 256] / len;

    bool done = false;
    float shade = 0.0f;
    for (int i = 0; i < n && !done; ++i) {
        const float di = dir[i * 256];
        if (di == 0.0f) continue;
        const float oi = c ? c[i] : cam.inl[i];
        const float s = di < 0.0f ? 1.0f : -1.0f;
        const float dist = (s - oi) / di;
        if (!(dist > 0.0f)) continue;
        bool ok = true;
        for (int j = 0; j < n; ++j) {
            if (j != i) {
                const float oj = c ? c[j] : cam.inl[j];
                const float p = dir[j * 256] * dist + oj;
                if (fabsf(p) > (1.0f + NT_FUZZ)) { ok = false; break; }
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

// Per-wave LDS scratch: a traversal stack [depth][64] of (node, t) pairs and the ray table
// [N][64] of (origin[axis], 1/direction[axis]) pairs used by the axis-indexed branch step.
// 8-byte entries at lane stride: ds_read_b64/ds_write_b64 are conflict free whatever level
// each lane is at (bank = f(lane) only).
struct WaveLds {
    float2 *stack;     // stack[level*64 + lane]
    float2 *ray;       // ray[axis*64 + lane]
};

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
__device__ __forceinline__ bool leaf_closest(const NtCompositeDev &sc, int start, int count, const float (&o)[N], const float (&d)[N],
                                             int skip_item, int skip_lane, Hit &hit, Stats &st) {
    bool improved = false;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        const int kind = item & 3;
        const int idx = item >> 2;
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
template <int N, bool FEAT, bool STATS>
__device__ __forceinline__ bool trace_closest(const NtCompositeDev &sc, const WaveLds &w, int lane, const float (&o)[N], const float (&d)[N],
                                              float t_near, float t_far_root, int skip_item, int skip_lane, Hit &hit, Stats &st) {
    hit.dist = FLT_MAX;
    hit.item = -1;
    hit.lane = -1;
    int node = sc.root;
    int sp = 0;
    int dirty = 0;
    float t_far = t_far_root;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (STATS) st.leaves += 1;
                if (leaf_closest<N, FEAT, STATS>(sc, nd.left, nd.right, o, d, skip_item, skip_lane, hit, st)) dirty = sp;
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
                        w.stack[sp * 64 + lane] = make_float2(__int_as_float(n_far), t);
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
            const float2 e = w.stack[sp * 64 + lane];
            const int far = __float_as_int(e.x);
            const float t = e.y;
            const bool near_hit = sp < dirty;
            if (dirty > sp) dirty = sp;
            if ((near_hit && hit.dist <= t) || far < 0) continue;     // frame returns `hit`
            node = far;
            t_near = t;
            t_far = sp > 0 ? w.stack[(sp - 1) * 64 + lane].y : t_far_root;
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
                        w.stack[sp * 64 + lane] = make_float2(__int_as_float(n_far), t);
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
            const float2 e = w.stack[sp * 64 + lane];
            const int far = __float_as_int(e.x);
            const float t = e.y;
            if (t < ldistance || far < 0) continue;             // frame returns false
            node = far;
            t_near = t;
            t_far = sp > 0 ? w.stack[(sp - 1) * 64 + lane].y : FLT_MAX;
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

template <int N, bool FEAT, bool STATS>
__device__ __forceinline__ Color3 composite_color(const NtCompositeDev &sc, const WaveLds &w, int lane, float (&o)[N], float (&d)[N], Stats &st) {
    Level levels[FEAT ? NT_DEV_MAX_REFLECT : 1];
    int depth = 0;
    int skip_item = -1, skip_lane = -1;
    Color3 result;
    for (;;) {
        // ---- ray_color (tracer.hpp:1856-1883) ----
        if (STATS) st.rays += 1;
        const float dist = aabb_distance<N>(sc, o, d);
        Hit hit;
        hit.item = -1;
        hit.lane = -1;
        hit.dist = FLT_MAX;
        bool found = false;
        if (dist >= 0.0f) {
            if (STATS && depth == 0) st.aabb_enter += 1;
            setup_ray_table<N>(w, lane, o, d);
            found = trace_closest<N, FEAT, STATS>(sc, w, lane, o, d, dist, FLT_MAX, skip_item, skip_lane, hit, st);
        }
        if (!found) {
            // target.direction[bg_gradient_axis]: select chain instead of indexing registers
            float iv = d[0];
#pragma unroll
            for (int k = 1; k < N; ++k) iv = sc.bg_axis == k ? d[k] : iv;
            result = iv >= 0.0f ? cadd(cscale(c3p(sc.bg1), iv), cscale(c3p(sc.bg2), 1.0f - iv))
                                : cadd(cscale(c3p(sc.bg3), -iv), cscale(c3p(sc.bg2), 1.0f + iv));
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
    const int per_wave = (sc.stack_depth + N) * 64;
    WaveLds w;
    w.stack = lds_raw + (size_t)wv * per_wave;
    w.ray = w.stack + (size_t)sc.stack_depth * 64;

    int px, py;
    if (tg.colors_out) { px = 0; py = 0; }
    else { px = (wv & 1) * 8 + (lane & 7); py = (wv >> 1) * 8 + (lane >> 3); }
    const PixelRef pr = locate_pixel<16, 16>(tg, px, py, tid);

    Stats st = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pr.valid) {
        float org[N], right[N], up[N], fwd[N], dir[N];
        load_camera<N>(cam, org, right, up, fwd);
        primary_dir<N>(tg, right, up, fwd, pr.x, pr.y, dir);
        const Color3 c = composite_color<N, FEAT, STATS>(sc, w, lane, org, dir, st);
        emit_pixel(tg, pr, c.r, c.g, c.b);
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
int launch_box_fixed(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg) {
    NtCameraFixed cf;
    cf.buf = cam.buf;
    cf.n = N;
    for (int k = 0; k < 4 * N; ++k) cf.inl[k] = cam.inl[k];
    dim3 grid;
    grid_for(tg, 64, 4, li.nframes, grid);
    hipLaunchKernelGGL(box_kernel<N>, grid, dim3(256), 0, (hipStream_t)li.stream, cf, tg);
    return 0;
}

template <int N>
int launch_composite_fixed(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) {
    NtCameraFixed cf;
    cf.buf = cam.buf;
    cf.n = N;
    for (int k = 0; k < 4 * N; ++k) cf.inl[k] = cam.inl[k];
    dim3 grid;
    grid_for(tg, 16, 16, li.nframes, grid);
    const size_t lds = (size_t)4 * (sc.stack_depth + N) * 64 * sizeof(float2);
    if (lds > 160 * 1024) {
        snprintf(g_launch_error, sizeof(g_launch_error), "k-d tree too deep for the LDS traversal stack (depth %d)", sc.stack_depth);
        return -1;
    }
    const bool feat = sc.n_point_lights || sc.n_global_lights || sc.any_reflective || sc.has_scalar_prims;
    hipStream_t s = (hipStream_t)li.stream;
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

int nt_launch_box(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg) {
    switch (li.n) {
        case 3: launch_box_fixed<3>(li, cam, tg); break;
        case 4: launch_box_fixed<4>(li, cam, tg); break;
        case 5: launch_box_fixed<5>(li, cam, tg); break;
        case 6: launch_box_fixed<6>(li, cam, tg); break;
        case 7: launch_box_fixed<7>(li, cam, tg); break;
        case 8: launch_box_fixed<8>(li, cam, tg); break;
        default: {
            dim3 grid;
            grid_for(tg, 64, 4, li.nframes, grid);
            const size_t lds = (size_t)li.n * 256 * sizeof(float);
            hipLaunchKernelGGL(box_kernel_var, grid, dim3(256), lds, (hipStream_t)li.stream, cam, tg);
        }
    }
    return finish_launch("box kernel launch");
}

int nt_launch_composite(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) {
    int r;
    switch (li.n) {
        case 3: r = launch_composite_fixed<3>(li, cam, sc, tg); break;
        case 4: r = launch_composite_fixed<4>(li, cam, sc, tg); break;
        case 5: r = launch_composite_fixed<5>(li, cam, sc, tg); break;
        case 6: r = launch_composite_fixed<6>(li, cam, sc, tg); break;
        case 7: r = launch_composite_fixed<7>(li, cam, sc, tg); break;
        case 8: r = launch_composite_fixed<8>(li, cam, sc, tg); break;
        default:
            snprintf(g_launch_error, sizeof(g_launch_error), "composite scenes with dimension %d are not supported yet (3..8)", li.n);
            return -2;
    }
    if (r) return r;
    return finish_launch("composite kernel launch");
}
