// nt_pixel.hpp -- device code shared by every kernel of libntracer_hip.so (gfx950, wave64): process_pixel's channel
// conversion and bit packing (src/render.cpp:419-462), the pixel <-> thread mapping that replaces worker_draw's chunk
// queue (render.cpp:468-493), the camera rows and the primary ray (src/tracer.hpp:60-76); plus the launch helpers.
//
// Arithmetic contract: identical operation order to oracle/ntracer_oracle.c, compiled with -ffp-contract=off, IEEE
// division/sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), so colours are bit-identical to the oracle
// except through powf/pow.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "nt_device.hpp"

// thread-local message of the last failed launch (nt_launch.cpp)
char *nt_launch_error_buf();
#define NT_LAUNCH_ERROR_LEN 256

namespace {

// has the caller raised the abort flag?  (NtTarget::abort_word: a system-scope load, past L1 / L2 -- the word lives in host
// memory and is written by the host while the kernel runs; every lane reads the same address)
__device__ __forceinline__ bool nt_aborted(const NtTarget &tg) {
    return tg.abort_word != nullptr && __hip_atomic_load(tg.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
}

// point_light::strength (tracer.hpp:1686-1688): 1 / std::pow(distance, dimension - 1) -- float and int arguments, so the
// reference's pow is the double one.  The exponent is a small positive integer: x^k by k - 1 multiplications in double agrees
// with the library's pow to an ulp or two of a DOUBLE, i.e. gives the same float after the final rounding (all but ~1e-8 of
// the time), and needs a handful of registers where the library's f64 pow -- logarithm, exponential, special cases -- needs
// dozens of register pairs: it alone set the shading kernels' register allocation.
__device__ __forceinline__ float nt_falloff(float distance, int k) {
    const double x = (double)distance;
    double p = x;
    for (int i = 1; i < k; ++i) p *= x;
    return (float)(1.0 / p);
}

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

// px,py: position inside the BW x BH tile of block (bx, by) of frame bz
template <int BW, int BH>
__device__ __forceinline__ PixelRef locate_pixel_at(const NtTarget &tg, int bx, int by, int bz, int px, int py, int tid) {
    PixelRef r;
    r.valid = false;
    r.x = 0;
    r.y = 0;
    r.offset = 0;
    r.hit_index = 0;
    if (tg.colors_out) {
        const int idx = bx * (BW * BH) + tid;
        if (idx < tg.probe_count) {
            r.x = tg.probe_xs[idx];
            r.y = tg.probe_ys[idx];
            r.offset = idx;
            r.valid = true;
        }
        return r;
    }
    const int x = bx * BW + px;
    const int row = by * BH + py;                    // relative to row_begin
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
    r.offset = (long long)bz * tg.frame_stride + (long long)(tg.compact ? orow : y) * tg.pitch + (long long)x * tg.bpp;
    r.hit_index = ((long long)bz * tg.row_count + row) * tg.width + x;
    r.valid = true;
    return r;
}

template <int BW, int BH>
__device__ __forceinline__ PixelRef locate_pixel(const NtTarget &tg, int px, int py, int tid) {
    return locate_pixel_at<BW, BH>(tg, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, px, py, tid);
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
__device__ __forceinline__ void load_camera(const NtCameraFixed &cam, unsigned frame, float (&org)[N], float (&right)[N], float (&up)[N], float (&fwd)[N]) {
    if (cam.buf) {
        const float *c = cam.buf + (size_t)frame * 4 * N;
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
template <int N>
__device__ __forceinline__ void load_camera(const NtCameraFixed &cam, float (&org)[N], float (&right)[N], float (&up)[N], float (&fwd)[N]) {
    load_camera<N>(cam, blockIdx.z, org, right, up, fwd);          // the frame is the grid's z almost everywhere
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

// --------------------------------------------------------------------------------------
// launch helpers
// --------------------------------------------------------------------------------------
template <typename T>
void set_error(const char *what, T err) {
    snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "%s: %s", what, hipGetErrorString((hipError_t)err));
}

void grid_for(const NtTarget &tg, int bw, int bh, int nframes, dim3 &grid) {
    if (tg.colors_out) {
        grid = dim3((unsigned)((tg.probe_count + bw * bh - 1) / (bw * bh)), 1, 1);
    } else {
        grid = dim3((unsigned)((tg.width + bw - 1) / bw), (unsigned)((tg.row_count + bh - 1) / bh), (unsigned)nframes);
    }
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
