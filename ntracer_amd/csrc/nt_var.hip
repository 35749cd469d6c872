// nt_var.hip -- the run-time-n kernels (the reference's generic `tracern` / var_geometry.hpp path: n-vectors in LDS as
// [k][lane]) and the dispatch over the dimension: nt_launch_box / nt_launch_composite.
#include "nt_box.hpp"
#include "nt_composite.hpp"

// compile-time-N launchers, one translation unit per N (nt_inst_box.hip / nt_inst_composite.hip)
#define NT_DECLARE_FIXED(N)                                                                              \
    int nt_box_fixed_##N(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);               \
    int nt_composite_fixed_##N(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg);
NT_DECLARE_FIXED(3) NT_DECLARE_FIXED(4) NT_DECLARE_FIXED(5) NT_DECLARE_FIXED(6)
NT_DECLARE_FIXED(7) NT_DECLARE_FIXED(8) NT_DECLARE_FIXED(9) NT_DECLARE_FIXED(10)
// (BoxScene alone: 11..24)
int nt_box_fixed_14(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_15(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_16(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_11(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_12(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_13(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_17(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_18(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_19(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_20(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_21(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_22(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_23(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_box_fixed_24(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);

namespace {

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
        const float *dp = cam.dots + (size_t)blockIdx.z * 4;
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
// BoxScene, run-time n, packed plain RGB (RGBX8 & co.): the structure of box_tile_kernel (nt_box.hpp) with the n-vectors in
// LDS.  A block = 64 columns x 32 rows (four waves of eight rows); one wave works out the stretch codes of the tile
// (box_stretch_code_var: what box_stretch_code computes, with loops over n); then every lane works out the three dot products
// of forward + right*sx that give |dir|^2 as a quadratic in sy (the vector itself is recomputed where a row needs it), and
//   * rows the codes call background or one face throughout cost a handful of operations per pixel WHATEVER n is
//     (guarded rsq quantisation as in box_tile_kernel),
//   * the others are evaluated ray by ray like box_kernel_var does (the reference's arithmetic on the faces in a near-tie).
// --------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t box_stretch_code_var(int n, const float *camrow, const NtTarget &tg, int y, int col) {
    const float *org = camrow, *right = camrow + n, *up = camrow + 2 * n, *fwd = camrow + 3 * n;
    float omax = fabsf(org[0]);
    for (int j = 1; j < n; ++j) omax = fmaxf(omax, fabsf(org[j]));
    const float m = NT_BOX_MARGIN * (1.0f + omax);
    const float h = 1.0f + 2.0f * m + 1e-3f;
    const float sxc = tg.fovI * (((float)(col * 64) + 31.5f) - tg.half_w);
    const float sy = tg.fovI * ((float)y - tg.half_h);
    const float spread = 32.0f * tg.fovI;
    float tlo = 0.0f, thi = INFINITY;
    bool dead = false;
    float tn = -INFINITY, tn2 = -INFINITY, vK = 0.0f, gK = 0.0f, oK = 0.0f;
    int K = 0;
    for (int j = 0; j < n; ++j) {
        const float vc = (fwd[j] + right[j] * sxc) - up[j] * sy;
        const float g = fmaf(spread, fabsf(right[j]), 1e-6f);
        const float pa = vc + g, qa = -h - org[j];
        const float pb = vc - g, qb = h - org[j];
        const float ra = qa * __builtin_amdgcn_rcpf(pa), rb = qb * __builtin_amdgcn_rcpf(pb);
        const float lo_a = pa > 0.0f ? ra : -INFINITY, hi_a = pa < 0.0f ? ra : INFINITY;
        const float hi_b = pb > 0.0f ? rb : INFINITY, lo_b = pb < 0.0f ? rb : -INFINITY;
        tlo = fmaxf(tlo, fmaxf(lo_a, lo_b));
        thi = fminf(thi, fminf(hi_a, hi_b));
        dead = dead || (pa == 0.0f && qa > 0.0f) || (pb == 0.0f && qb < 0.0f);
        const float nr = ((vc < 0.0f ? 1.0f : -1.0f) - org[j]) * __builtin_amdgcn_rcpf(vc);
        tn2 = __builtin_amdgcn_fmed3f(tn, tn2, nr);
        const bool later = nr > tn;
        vK = later ? vc : vK;
        gK = later ? g : gK;
        oK = later ? org[j] : oK;
        K = later ? j : K;
        tn = fmaxf(tn, nr);
    }
    if (dead || tlo > thi) return 0u;                  // a NaN keeps the stretch
    uint32_t code = 15u;
    const float vKa = vK - gK, vKb = vK + gK;
    if (K <= 12 && vKa * vKb > 0.0f) {                 // (codes 1 .. 13 name the face)
        const float num = (vK < 0.0f ? 1.0f : -1.0f) - oK;
        const float t1 = num * __builtin_amdgcn_rcpf(vKa), t2 = num * __builtin_amdgcn_rcpf(vKb);
        const float t_lo = fminf(t1, t2) * (1.0f - 1e-6f), t_hi = fmaxf(t1, t2) * (1.0f + 1e-6f);
        const float rK = m * __builtin_amdgcn_rcpf(fminf(fabsf(vKa), fabsf(vKb))) * (1.0f + 1e-6f);
        bool ok = t_lo > 1e-3f && t_hi < 1e30f;
        for (int j = 0; j < n; ++j) {
            const float vc = (fwd[j] + right[j] * sxc) - up[j] * sy;
            const float g = fmaf(spread, fabsf(right[j]), 1e-6f);
            const float va = vc - g, vb = vc + g;
            const float pmax = org[j] + fmaxf(vb * t_lo, vb * t_hi);
            const float pmin = org[j] + fminf(va * t_lo, va * t_hi);
            const float lim = (1.0f - m - 1e-4f) - fmaxf(fabsf(va), fabsf(vb)) * rK;
            ok = ok && (j == K || (pmax <= lim && pmin >= -lim));
        }
        if (ok) code = (uint32_t)K + 1u;
    }
    return code;
}

__global__ __launch_bounds__(256) void box_rows_kernel_var(NtCamera cam, NtTarget tg) {
    constexpr int R = 8;
    // dirs [n][256], camera rows [4][n], codes [4].  (Until round 3 `forward + right*sx` sat in LDS as well, [n][256]: with it a
    // block needed 2n KB, and from n = 25 on it was LDS that set the occupancy -- two blocks a CU at n = 32, one at n = 64, where a
    // wave alone issues an instruction every 8.4 cycles.  The two operations are simply done again where the value is used.)
    extern __shared__ float lds_var[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = cam.n;
    float *dirs = lds_var + tid;          // component j at [j * 256]
    float *camrow = lds_var + (size_t)n * 256;
    uint32_t *s_code = reinterpret_cast<uint32_t *>(camrow + 4 * n);
    {
        const float *src = cam.buf ? cam.buf + (size_t)blockIdx.z * 4 * n : nullptr;
        for (int k = tid; k < 4 * n; k += 256) camrow[k] = src ? src[k] : cam.inl[k];
    }
    __syncthreads();
    const float *org = camrow, *right = camrow + n, *up = camrow + 2 * n, *fwd = camrow + 3 * n;
    const int tile_row0 = (int)blockIdx.y * 4 * R;
    if (wv == (int)((blockIdx.x + blockIdx.y + blockIdx.z) & 3u)) {
        uint32_t code = 0u;
        const int trow = tile_row0 + lane;
        if (lane < 4 * R && trow < tg.row_count) {
            const int orow = tg.row_begin + trow;
            int y = orow;
            if (tg.band_world > 1) {
                const int band = orow / tg.band_rows;
                y = (band * tg.band_world + tg.band_rank) * tg.band_rows + (orow - band * tg.band_rows);
            }
            if (y < tg.height) code = box_stretch_code_var(n, camrow, tg, y, (int)blockIdx.x);
        }
        uint32_t packed = code << (4 * (lane & 7));
        packed |= (uint32_t)__shfl_xor((int)packed, 1, 64);
        packed |= (uint32_t)__shfl_xor((int)packed, 2, 64);
        packed |= (uint32_t)__shfl_xor((int)packed, 4, 64);
        if ((lane & 7) == 0 && lane < 4 * R) s_code[lane >> 3] = packed;          // wave w's eight rows: s_code[w]
    }
    __syncthreads();
    const int row0 = tile_row0 + wv * R;
    if (row0 >= tg.row_count) return;
    const uint32_t rowcodes = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_code[wv]);
    // row bookkeeping: one row per lane, read back with v_readlane (see box_tile_kernel)
    const int lorow = tg.row_begin + row0 + lane;
    int ly = lorow;
    if (tg.band_world > 1) {
        const int band = lorow / tg.band_rows;
        ly = (band * tg.band_world + tg.band_rank) * tg.band_rows + (lorow - band * tg.band_rows);
    }
    const uint32_t valid = (uint32_t)__builtin_amdgcn_ballot_w64(lane < R && row0 + lane < tg.row_count && ly < tg.height);
    const float v_sy = tg.fovI * ((float)ly - tg.half_h);
    const long long v_off = (long long)blockIdx.z * tg.frame_stride + (long long)(tg.compact ? lorow : ly) * tg.pitch;
    const int v_off_lo = (int)v_off, v_off_hi = (int)(v_off >> 32);
    int x = (int)blockIdx.x * 64 + lane;
    x = x < tg.width ? x : tg.width - 1;                 // lanes past the right edge redo the last pixel
    const long long xoff = (long long)x * tg.bpp;
    const float sx = tg.fovI * ((float)x - tg.half_w);
    float bb = 0.0f, bu = 0.0f, uu = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float b = fwd[j] + right[j] * sx;
        bb = fmaf(b, b, bb);
        bu = fmaf(b, up[j], bu);
        uu = fmaf(up[j], up[j], uu);
    }
    const float base0 = fwd[0] + right[0] * sx, up0 = up[0];
    const float m2bu = -2.0f * bu;
    // (the quadratic's error grows with n: (3.7n + 4) * 2^-24 relative -- the guard below is sized for it)
    const bool fastsq = __builtin_amdgcn_ballot_w64(!(bu * bu <= bb * uu * 0.0625f)) == 0ull;
    const float guard = (5.55f * (float)n + 21.0f) * 0x1p-24f;         // three times the error bound (1.85n + 7) * 2^-24 of t: 2^-18 at n = 8
    const float maxv = (float)tg.plain_maxval;
    float dots[4];
    if (cam.buf) {
        const float *dp = cam.dots + (size_t)blockIdx.z * 4;
        dots[0] = dp[0]; dots[1] = dp[1]; dots[2] = dp[2]; dots[3] = dp[3];
    } else {
        dots[0] = cam.odots[0]; dots[1] = cam.odots[1]; dots[2] = cam.odots[2]; dots[3] = cam.odots[3];
    }
    for (int rr = 0; rr < R; ++rr) {
        if (!((valid >> rr) & 1u)) continue;
        const uint32_t code = (rowcodes >> (4 * rr)) & 15u;
        const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_sy), rr));
        PixelRef pr;
        pr.x = x;
        pr.y = 0;
        pr.offset = (((long long)__builtin_amdgcn_readlane(v_off_hi, rr) << 32) | (unsigned)__builtin_amdgcn_readlane(v_off_lo, rr)) + xoff;
        pr.hit_index = 0;
        pr.valid = true;
        if (fastsq && code <= 13u) {
            // background (code 0) or face K = code - 1 throughout: |x| / |dir| from x = dir[0] or dir[K] alone
            const int K = code == 0u ? 0 : (int)code - 1;
            const float xk = (code == 0u ? base0 : fwd[K] + right[K] * sx) - (code == 0u ? up0 : up[K]) * sy;       // dir[K], bit for bit
            const float sqa = fmaf(sy, fmaf(sy, uu, m2bu), bb);
            const float t = (fabsf(xk) * __builtin_amdgcn_rsqf(sqa)) * maxv, th = t * 0.5f;
            const bool clear = fabsf(__builtin_amdgcn_fractf(t) - 0.5f) > fmaf(t, guard, guard) &&
                               (code == 0u || fabsf(__builtin_amdgcn_fractf(th) - 0.5f) > fmaf(th, guard, guard));
            if (__builtin_amdgcn_ballot_w64(!clear) == 0ull) {
                uint32_t q = (uint32_t)(t + 0.5f), qh = (uint32_t)(th + 0.5f);
                q = q < tg.plain_maxval ? q : tg.plain_maxval;
                qh = qh < tg.plain_maxval ? qh : tg.plain_maxval;
                if (code == 0u) emit_plain(tg, pr, xk > 0.0f ? q : 0u, q);
                else emit_plain(tg, pr, q, qh);
                continue;
            }
        }
        // ---- ray by ray: box_kernel_var's evaluation (the reference's arithmetic on the faces in a near-tie with the
        // last-reached one), with the direction in `dirs`
        float sq = 0.0f;
        for (int j = 0; j < n; ++j) {
            const float v = (fwd[j] + right[j] * sx) - up[j] * sy;
            dirs[j * 256] = v;
            sq = j == 0 ? v * v : sq + v * v;
        }
        const float len = sqrtf(sq);
        const float osq = dots[0];
        const float ov = fmaf(-dots[2], sy, fmaf(dots[1], sx, dots[3]));
        const float rad2 = (float)n * (1.0f + NT_FUZZ) * (1.0f + NT_FUZZ) * 1.001f;
        const bool maybe = code != 0u && !((osq - rad2 * 1.0001f - 1e-5f * osq) * sq > ov * ov * 1.0001f);
        const bool wave_maybe = __builtin_amdgcn_ballot_w64(maybe) != 0ull;
        bool done = false;
        float shade = 0.0f;
        if (wave_maybe) {
            for (int j = 0; j < n; ++j) dirs[j * 256] = dirs[j * 256] / len;
            float aK = 0.0f, bK = 1.0f, oK = 0.0f;
            bool any = false;
            for (int i = 0; i < n; ++i) {
                const float di = dirs[i * 256];
                const float oi = org[i];
                const float num = (di < 0.0f ? 1.0f : -1.0f) - oi;
                const bool pre = maybe && ((num > 0.0f && di > 0.0f) || (num < 0.0f && di < 0.0f));
                const float a = fabsf(num), bq = fabsf(di);
                if (pre && (!any || a * bK > aK * bq)) { aK = a; bK = bq; oK = oi; any = true; }
            }
            const float mu = 1e-4f * (1.0f + fabsf(oK));
            const float aKm = (aK - mu) * (1.0f - 1e-6f);
            for (int i = 0; i < n; ++i) {
                const float di = dirs[i * 256];
                const float oi = org[i];
                const float s_ = di < 0.0f ? 1.0f : -1.0f;
                const float num = s_ - oi;
                const bool pre = maybe && ((num > 0.0f && di > 0.0f) || (num < 0.0f && di < 0.0f));
                const bool tie = pre && !done && !(fabsf(num) * bK < fabsf(di) * aKm);
                if (__builtin_amdgcn_ballot_w64(tie) == 0ull) continue;
                const float dist = num / di;
                bool ok = tie && dist > 0.0f;
                for (int j = 0; j < n; ++j) {
                    if (j != i) {
                        const float pj = dirs[j * 256] * dist + org[j];
                        ok = ok && !(fabsf(pj) > (1.0f + NT_FUZZ));
                    }
                }
                if (ok) {
                    done = true;
                    if (dist >= FLT_MAX) shade = -1.0f;
                    else {
                        const float sine = di * s_;
                        shade = sine <= 0.0f ? -sine : 0.0f;
                    }
                }
            }
        } else {
            dirs[0] = dirs[0] / len;
        }
        float r, g, b;
        if (done && shade >= 0.0f) {
            r = shade * 1.0f;
            g = shade * 0.5f;
            b = shade * 0.5f;
        } else {
            const float in = dirs[0];
            if (in > 0.0f) { r = in; g = in; b = in; }
            else { r = 0.0f; g = -in; b = -in; }
        }
        (void)b;
        emit_plain(tg, pr, plain_quantize(tg, r), plain_quantize(tg, g));            // g == b
    }
}

// --------------------------------------------------------------------------------------
// CompositeScene, run-time n (up to 64): the reference's generic `tracern` module (var_geometry.hpp).  Per-lane
// kernel, the whole of composite_scene::calculate_color for opaque scenes: batches, unbatched triangles, Solids,
// point / global / camera lights, shadow rays (with _occludes' far-child rule), reflection.  The current ray's n-vectors
// (origin, 1/direction, direction, a scratch vector) live in LDS as [k][lane] -- every record component is applied to all
// of them in turn -- while the vectors shading needs once per hit (normal ray, light vector, the saved view direction,
// a Solid's local ray) sit in per-lane scratch memory.  Simplex and solid records are read component by component.
// Operation order is the oracle's, so results equal the compile-time-N kernels' where both exist (NTRACER_FORCE_VAR=1).
// Transparent materials, and the reference's o_hit.normal handling for scenes with Solids: composite_kernel_var_t below.
// --------------------------------------------------------------------------------------
struct VarLds {
    float2 *ray;     // [n][64] (origin, 1/direction; NaN marks direction == 0)
    float *dv;       // [n][64] direction
    float *ps;       // [n][64] scratch: pside
    int *stack;      // [depth][64]
    int *mbox;       // [NT_MBOX][64]
};

__device__ __forceinline__ size_t var_lds_bytes(int depth, int n) {
    return (size_t)64 * ((size_t)n * 16 + (size_t)depth * 4 + (size_t)NT_MBOX * 4);
}

struct VarCtx {
    const NtCompositeDev &sc;
    VarLds L;
    WaveLds w;
    int n, lane;
};

#define VO(k) (cx.L.ray[(k) * 64 + cx.lane].x)       // origin[k] of the current ray
#define VD(k) (cx.L.dv[(k) * 64 + cx.lane])          // direction[k]

// make (o, d) the current ray: origin, direction and invdir = 1/direction (tracer.hpp:1174)
__device__ __forceinline__ void var_set_ray(const VarCtx &cx, const float *o, const float *d) {
    for (int k = 0; k < cx.n; ++k) {
        const float dk = d[k];
        cx.L.dv[k * 64 + cx.lane] = dk;
        cx.L.ray[k * 64 + cx.lane] = make_float2(o[k], dk != 0.0f ? 1.0f / dk : __int_as_float(0x7fc00000));
    }
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

// solid::intersects (tracer.hpp:251-276) on the current ray, hypercube_intersects / hypersphere_intersects (:126-173) on
// the local ray.  want_normal: (no, nd) receive the world-space normal ray.
__device__ __noinline__ float solid_var(const VarCtx &cx, int idx, float cutoff, bool want_normal, float *no, float *nd) {
    const int n = cx.n;
    const float *orient = cx.sc.solid_recs + (size_t)idx * (2 * n * n + n);
    const float *inv = orient + n * n;
    const float *pos = inv + n * n;
    float lo[NT_DEV_MAX_DIM], ld[NT_DEV_MAX_DIM], ln_o[NT_DEV_MAX_DIM], ln_d[NT_DEV_MAX_DIM];
    for (int i = 0; i < n; ++i) {
        float so = 0.0f, sd = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float m = inv[i * n + k];
            const float po = m * VO(k), pd = m * VD(k);
            so = k == 0 ? po : so + po;
            sd = k == 0 ? pd : sd + pd;
        }
        lo[i] = so - pos[i];
        ld[i] = sd;
    }
    float dist = 0.0f;
    if (cx.sc.solid_types[idx] == 1) {
        for (int i = 0; i < n; ++i) {
            const float di = ld[i];
            if (di == 0.0f) continue;
            const float s = di < 0.0f ? 1.0f : -1.0f;
            const float t = (s - lo[i]) / di;
            if (!(t > 0.0f)) continue;
            bool ok = true;
            for (int j = 0; j < n; ++j) {
                if (j != i) {
                    const float p = ld[j] * t + lo[j];
                    ln_o[j] = p;
                    if (fabsf(p) > (1.0f + NT_FUZZ)) { ok = false; break; }
                }
            }
            if (!ok) continue;
            if (t >= cutoff) return 0.0f;
            dist = t;
            ln_o[i] = s;
            for (int j = 0; j < n; ++j) ln_d[j] = j == i ? s : 0.0f;
            break;
        }
    } else {
        float a = 0.0f, b = 0.0f, c = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float pa = ld[k] * ld[k], pb = ld[k] * lo[k], pc = lo[k] * lo[k];
            a = k == 0 ? pa : a + pa;
            b = k == 0 ? pb : b + pb;
            c = k == 0 ? pc : c + pc;
        }
        b = 2.0f * b;
        c = c - 1.0f;
        const float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return 0.0f;
        const float t = (-b - sqrtf(disc)) / (2.0f * a);
        if (t <= 0.0f || t >= cutoff) return 0.0f;
        dist = t;
        for (int j = 0; j < n; ++j) { ln_o[j] = lo[j] + ld[j] * t; ln_d[j] = ln_o[j]; }
    }
    if (dist == 0.0f) return 0.0f;
    if (want_normal) {
        for (int i = 0; i < n; ++i) ln_o[i] = ln_o[i] + pos[i];
        for (int i = 0; i < n; ++i) {
            float so = 0.0f, sd = 0.0f;
            for (int k = 0; k < n; ++k) {
                const float m = orient[i * n + k];
                const float po = m * ln_o[k], pd = m * ln_d[k];
                so = k == 0 ? po : so + po;
                sd = k == 0 ? pd : sd + pd;
            }
            no[i] = so;
            nd[i] = sd;
        }
    }
    return dist;
}

// composite_scene::aabb_distance (tracer.hpp:1892-1918) for the current ray
__device__ __forceinline__ float aabb_distance_var(const VarCtx &cx) {
    const int n = cx.n;
    const float *aabb = cx.sc.aabb;
    for (int i = 0; i < n; ++i) {
        const float di = VD(i);
        if (di == 0.0f) continue;
        const float face = di > 0.0f ? aabb[i] : aabb[n + i];
        float dist = (face - VO(i)) / di;
        int skip = i;
        if (dist < 0.0f) { dist = 0.0f; skip = -1; }
        bool ok = true;
        for (int j = 0; j < n; ++j) {
            if (j != skip) {
                const float p = VD(j) * dist + VO(j);
                if (p >= aabb[n + j] || p <= aabb[j]) { ok = false; break; }
            }
        }
        if (ok) return dist;
    }
    return -1.0f;
}

// kd_leaf<Store,true>::intersects for all-opaque scenes (see leaf_closest in nt_composite.hpp)
__device__ __forceinline__ bool leaf_closest_var(const VarCtx &cx, int start, int count, int skip_item, int skip_lane, Hit &hit) {
    const NtCompositeDev &sc = cx.sc;
    bool improved = false;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        const int kind = item & 3, idx = item >> 2;
        if (mbox_seen(cx.w, cx.lane, item)) continue;
        if (kind == 0) {
            const int sl = item == skip_item ? skip_lane : -1;
            float min_t = hit.dist;
            int r = -1;
            for (int l = 0; l < NT_DEV_BATCH; ++l) {
                const float t = simplex_var(sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + l) * sc.rec_stride, cx.n, cx.L, cx.lane, false, 0.0f);
                if (l != sl && t != 0.0f && t < min_t) { min_t = t; r = l; }
            }
            if (r >= 0) { hit.dist = min_t; hit.item = item; hit.lane = r; improved = true; }
        } else if (item != skip_item) {
            float t;
            if (kind == 1) t = simplex_var(sc.tri_recs + (size_t)idx * sc.rec_stride, cx.n, cx.L, cx.lane, true, hit.dist);
            else t = solid_var(cx, idx, hit.dist, false, nullptr, nullptr);
            if (t != 0.0f) { hit.dist = t; hit.item = item; hit.lane = -1; improved = true; }
        }
    }
    return improved;
}

// kd_node_intersection::operator() (tracer.hpp:1179-1243): the continuation stack of trace_closest (nt_composite.hpp)
__device__ __noinline__ bool trace_closest_var(const VarCtx &cx, float t_near, int skip_item, int skip_lane, Hit &hit) {
    const NtCompositeDev &sc = cx.sc;
    const WaveLds &w = cx.w;
    const int lane = cx.lane;
    hit.dist = FLT_MAX;
    hit.item = -1;
    hit.lane = -1;
    mbox_reset(w, lane);
    int node = sc.root, sp = 0, dirty = 0;
    float t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            if (sc.prune && nt_beyond_hit(hit.dist, t_near)) { node = -1; break; }
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (leaf_closest_var(cx, nd.left, nd.right, skip_item, skip_lane, hit)) dirty = sp;
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
                node = n_far;
                t_near = t;
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
            resumed = true;
            break;
        }
        if (!resumed) break;
    }
    return hit.item >= 0;
}

// kd_leaf::occludes (tracer.hpp:1088-1124), all-opaque scenes
__device__ __forceinline__ bool leaf_occludes_var(const VarCtx &cx, int start, int count, float ldistance, int skip_item, int skip_lane) {
    const NtCompositeDev &sc = cx.sc;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        const int kind = item & 3, idx = item >> 2;
        if (kind == 0) {
            const int sl = item == skip_item ? skip_lane : -1;
            bool any = false;
            for (int l = 0; l < NT_DEV_BATCH; ++l) {
                const float t = simplex_var(sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + l) * sc.rec_stride, cx.n, cx.L, cx.lane, false, 0.0f);
                any = any || (l != sl && t != 0.0f && t < ldistance);
            }
            if (any) return true;
        } else if (item != skip_item) {
            float t;
            if (kind == 1) t = simplex_var(sc.tri_recs + (size_t)idx * sc.rec_stride, cx.n, cx.L, cx.lane, true, ldistance);
            else t = solid_var(cx, idx, ldistance, false, nullptr, nullptr);
            if (t != 0.0f) return true;
        }
    }
    return false;
}

// _occludes (tracer.hpp:1258-1307) for the current ray, `if(t < ldistance) return false;` (:1298) included
__device__ __noinline__ bool trace_occluded_var(const VarCtx &cx, float ldistance, int skip_item, int skip_lane) {
    const NtCompositeDev &sc = cx.sc;
    const WaveLds &w = cx.w;
    const int lane = cx.lane;
    int node = sc.root, sp = 0;
    float t_near = 0.0f, t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (leaf_occludes_var(cx, nd.left, nd.right, ldistance, skip_item, skip_lane)) return true;
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

__device__ __forceinline__ float dot_var(int n, const float *a, const float *b) {
    float s = 0.0f;
    for (int k = 0; k < n; ++k) {
        const float p = a[k] * b[k];
        s = k == 0 ? p : s + p;
    }
    return s;
}

// append_specular (tracer.hpp:1701-1707)
__device__ __forceinline__ void append_specular_var(int n, Color3 &c, float &a, const float *m, Color3 light_c, const float *target, const float *normal,
                                                    const float *light_dir) {
    float tmp[NT_DEV_MAX_DIM];
    for (int k = 0; k < n; ++k) tmp[k] = light_dir[k] - target[k];
    const float len = sqrtf(dot_var(n, tmp, tmp));
    for (int k = 0; k < n; ++k) tmp[k] = tmp[k] / len;
    const float base = powf(dot_var(n, normal, tmp), m[9]) * m[8];
    c = cadd(c, cscale(cscale(cmul(c3p(m + 3), light_c), base), (1.0f - a)));
    a += base * (1.0f - a);
    c = cscale(c, a);
}

// ray_color + base_color (tracer.hpp:1768-1883), the reflection recursion as a level stack (see composite_color).
// On entry the primary ray is the current ray.
__device__ __noinline__ Color3 composite_color_var(const VarCtx &cx) {
    const NtCompositeDev &sc = cx.sc;
    const int n = cx.n;
    Level levels[NT_DEV_MAX_REFLECT];
    Color3 deep_a = c3(0.0f, 0.0f, 0.0f), deep_b = c3(1.0f, 1.0f, 1.0f);
    int depth = 0;
    int skip_item = -1, skip_lane = -1;
    Color3 result;
    float dir[NT_DEV_MAX_DIM], no[NT_DEV_MAX_DIM], nd[NT_DEV_MAX_DIM], lv[NT_DEV_MAX_DIM];
    for (;;) {
        // ---- ray_color (tracer.hpp:1856-1883)
        Hit hit;
        hit.dist = FLT_MAX; hit.item = -1; hit.lane = -1;
        bool found = false;
        const float dist0 = aabb_distance_var(cx);
        if (dist0 >= 0.0f) found = trace_closest_var(cx, dist0, skip_item, skip_lane, hit);
        if (!found) {
            const float iv = VD(sc.bg_axis);
            result = iv >= 0.0f ? cadd(cscale(c3p(sc.bg1), iv), cscale(c3p(sc.bg2), 1.0f - iv))
                                : cadd(cscale(c3p(sc.bg3), -iv), cscale(c3p(sc.bg2), 1.0f + iv));
            break;
        }
        // ---- the hit's normal ray (what the reference's tests stored in o_hit.normal)
        const int kind = hit.item & 3, idx = hit.item >> 2;
        for (int k = 0; k < n; ++k) dir[k] = VD(k);
        if (kind != 2) {
            const float *rec = kind == 0 ? sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + hit.lane) * sc.rec_stride
                                         : sc.tri_recs + (size_t)idx * sc.rec_stride;
            float denom = 0.0f, fsq = 0.0f;
            for (int k = 0; k < n; ++k) {
                const float fn = rec[1 + k];
                const float pd = fn * dir[k];
                denom = k == 0 ? pd : denom + pd;
                fsq = k == 0 ? fn * fn : fsq + fn * fn;
            }
            const float flen = sqrtf(fsq);
            for (int k = 0; k < n; ++k) {
                no[k] = VO(k) + hit.dist * dir[k];
                const float u = rec[1 + k] / flen;
                nd[k] = denom > 0.0f ? -u : u;
            }
        } else {
            solid_var(cx, idx, FLT_MAX, true, no, nd);
        }
        // ---- base_color (tracer.hpp:1768-1854)
        const float *m = material_of(sc, hit.item, hit.lane);
        Color3 light = c3(0.0f, 0.0f, 0.0f), specular = c3(0.0f, 0.0f, 0.0f);
        float spec_a = 0.0f;
        for (int li = 0; li < sc.n_point_lights; ++li) {
            const float *pos = sc.pl_pos + (size_t)li * n;
            const Color3 plc = c3p(sc.pl_color + 3 * li);
            for (int k = 0; k < n; ++k) lv[k] = no[k] - pos[k];
            const float ldist = sqrtf(dot_var(n, lv, lv));
            for (int k = 0; k < n; ++k) lv[k] = lv[k] / ldist;
            const float sine = dot_var(n, nd, lv);
            if (sine > 0.0f) {
                const float strength = nt_falloff(ldist, n - 1);
                if (sc.shadows) {
                    if (fmaxf(plc.r, fmaxf(plc.g, plc.b)) * strength * sine > NT_LIGHT_THRESHOLD) {
                        var_set_ray(cx, no, lv);
                        if (!trace_occluded_var(cx, ldist, hit.item, hit.lane)) {
                            const Color3 filtered = cscale(plc, strength);
                            light = cadd(light, cscale(filtered, sine));
                            if (m[8] != 0.0f) append_specular_var(n, specular, spec_a, m, filtered, dir, nd, lv);
                        }
                    }
                } else {
                    light = cadd(light, cscale(cscale(plc, strength), sine));
                }
            }
        }
        for (int li = 0; li < sc.n_global_lights; ++li) {
            const float *gd = sc.gl_dir + (size_t)li * n;
            const Color3 glc = c3p(sc.gl_color + 3 * li);
            const float sine = -dot_var(n, nd, gd);
            if (sine > 0.0f) {
                if (sc.shadows) {
                    for (int k = 0; k < n; ++k) lv[k] = -gd[k];
                    var_set_ray(cx, no, lv);
                    if (!trace_occluded_var(cx, FLT_MAX, hit.item, hit.lane)) {
                        light = cadd(light, cscale(glc, sine));
                        if (m[8] != 0.0f) append_specular_var(n, specular, spec_a, m, glc, dir, nd, lv);
                    }
                } else {
                    light = cadd(light, cscale(glc, sine));
                }
            }
        }
        const float sine = -dot_var(n, dir, nd);
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
        if (m[7] != 0.0f && depth < sc.max_reflect_depth) {
            if (depth < NT_DEV_MAX_REFLECT) {
                Level &Lv = levels[depth];
                Lv.spec = specular;
                Lv.spec_a = spec_a;
                Lv.r0 = r0;
                Lv.c = c3p(m);
                Lv.refl = m[7];
            } else {                        // beyond the level stack: folded into a running affine pair (see composite_color)
                const float k1 = 1.0f - spec_a;
                const Color3 alpha = cadd(specular, cscale(cscale(r0, 1.0f - m[7]), k1));
                const Color3 beta = cscale(cscale(c3p(m), m[7]), k1);
                deep_a = cadd(deep_a, cmul(deep_b, alpha));
                deep_b = cmul(deep_b, beta);
            }
            const float f = -2.0f * sine;
            for (int k = 0; k < n; ++k) dir[k] = dir[k] - nd[k] * f;
            var_set_ray(cx, no, dir);
            skip_item = hit.item;
            skip_lane = hit.lane;
            ++depth;
            continue;
        }
        result = cadd(specular, cscale(r0, 1.0f - spec_a));
        break;
    }
    if (depth > NT_DEV_MAX_REFLECT) {
        result = cadd(deep_a, cmul(deep_b, result));
        depth = NT_DEV_MAX_REFLECT;
    }
    while (depth > 0) {
        --depth;
        const Level &Lv = levels[depth];
        const Color3 r = cadd(cscale(cmul(Lv.c, result), Lv.refl), cscale(Lv.r0, 1.0f - Lv.refl));
        result = cadd(Lv.spec, cscale(r, 1.0f - Lv.spec_a));
    }
    return result;
}

// --------------------------------------------------------------------------------------
// Transparent materials, and the reference's o_hit.normal handling, at run-time n: composite_kernel_t (nt_composite.hpp)
// restated with the current ray in LDS.  The ray_color frames -- one per reflection level, each with its ray, the walk's
// normal ray and its surface list -- live in global scratch, [frame][word][lane slot], sized by the host for
// max_reflect_depth + 1 levels: no depth limit here, which is why scenes with transparency whose depth is beyond the
// compile-time-N kernel's frame stack are sent here as well, whatever their n.
// --------------------------------------------------------------------------------------
struct VarFrames {
    float *base;             // this lane's column
    long long stride;        // lane slots
    int fw;                  // words per frame
};
// frame words: o[n] d[n] hn_o[n] hn_d[n], then the scalars below, then (NT_TH_MAX + 1) surfaces of 3 words
enum { VF_DEPTH = 0, VF_SKIP_ITEM, VF_SKIP_LANE, VF_NSURF, VF_J, VF_R, VF_SPEC = VF_R + 3, VF_R0 = VF_SPEC + 3, VF_C = VF_R0 + 3,
       VF_SPEC_A = VF_C + 3, VF_REFL, VF_SURF, VF_WORDS = VF_SURF + 3 * (NT_TH_MAX + 1) };
__host__ __device__ __forceinline__ int var_frame_words(int n) { return 4 * n + VF_WORDS; }

#define FRW(f, w) (fr.base[((long long)(f) * fr.fw + (w)) * fr.stride])
#define FRI(f, w) (reinterpret_cast<int *>(fr.base)[((long long)(f) * fr.fw + (w)) * fr.stride])

__device__ __forceinline__ Color3 fr_get3(const VarFrames &fr, int f, int w) { return c3(FRW(f, w), FRW(f, w + 1), FRW(f, w + 2)); }
__device__ __forceinline__ void fr_put3(const VarFrames &fr, int f, int w, Color3 c) { FRW(f, w) = c.r; FRW(f, w + 1) = c.g; FRW(f, w + 2) = c.b; }

// test_item (nt_composite.hpp) on the current ray
__device__ __forceinline__ float test_item_var(const VarCtx &cx, int item, float cutoff, int skip_item, int skip_lane, int &lane_out) {
    const NtCompositeDev &sc = cx.sc;
    const int kind = item & 3, idx = item >> 2;
    lane_out = -1;
    if (kind == 0) {
        const int sl = item == skip_item ? skip_lane : -1;
        float min_t = cutoff;
        int r = -1;
        for (int l = 0; l < NT_DEV_BATCH; ++l) {
            const float t = simplex_var(sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + l) * sc.rec_stride, cx.n, cx.L, cx.lane, false, 0.0f);
            if (l != sl && t != 0.0f && t < min_t) { min_t = t; r = l; }
        }
        lane_out = r;
        return r >= 0 ? min_t : 0.0f;
    }
    if (kind == 1) return simplex_var(sc.tri_recs + (size_t)idx * sc.rec_stride, cx.n, cx.L, cx.lane, true, cutoff);
    return solid_var(cx, idx, cutoff, false, nullptr, nullptr);
}

// the normal ray of a hit on the current ray (hit_normal in nt_composite.hpp)
__device__ __forceinline__ void hit_normal_var(const VarCtx &cx, const Hit &hit, float *no, float *nd) {
    const NtCompositeDev &sc = cx.sc;
    const int n = cx.n;
    const int kind = hit.item & 3, idx = hit.item >> 2;
    if (kind == 2) {
        solid_var(cx, idx, FLT_MAX, true, no, nd);
        return;
    }
    const float *rec = kind == 0 ? sc.batch_recs + ((size_t)idx * NT_DEV_BATCH + hit.lane) * sc.rec_stride
                                 : sc.tri_recs + (size_t)idx * sc.rec_stride;
    float denom = 0.0f, fsq = 0.0f;
    for (int k = 0; k < n; ++k) {
        const float fn = rec[1 + k];
        const float pd = fn * VD(k);
        denom = k == 0 ? pd : denom + pd;
        fsq = k == 0 ? fn * fn : fsq + fn * fn;
    }
    const float flen = sqrtf(fsq);
    for (int k = 0; k < n; ++k) {
        no[k] = VO(k) + hit.dist * VD(k);
        const float u = rec[1 + k] / flen;
        nd[k] = denom > 0.0f ? -u : u;
    }
}

// solid::intersects writing through to the caller's normal ray as the reference does (solid_intersects_marks /
// cube_local_marks in nt_composite.hpp; tracer.hpp:126-152, 251-276)
__device__ __noinline__ float solid_marks_var(const VarCtx &cx, int idx, float cutoff, float *no, float *nd) {
    const int n = cx.n;
    const float *orient = cx.sc.solid_recs + (size_t)idx * (2 * n * n + n);
    const float *inv = orient + n * n;
    const float *pos = inv + n * n;
    float lo[NT_DEV_MAX_DIM], ld[NT_DEV_MAX_DIM];
    for (int i = 0; i < n; ++i) {
        float so = 0.0f, sd = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float m = inv[i * n + k];
            const float po = m * VO(k), pd = m * VD(k);
            so = k == 0 ? po : so + po;
            sd = k == 0 ? pd : sd + pd;
        }
        lo[i] = so - pos[i];
        ld[i] = sd;
    }
    float dist = 0.0f;
    if (cx.sc.solid_types[idx] == 1) {
        for (int i = 0; i < n; ++i) {
            const float di = ld[i];
            if (di == 0.0f) continue;
            const float s = di < 0.0f ? 1.0f : -1.0f;
            no[i] = s;
            const float t = (s - lo[i]) / di;
            if (!(t > 0.0f)) continue;
            bool ok = true;
            for (int j = 0; j < n; ++j) {
                if (j != i) {
                    const float p = ld[j] * t + lo[j];
                    no[j] = p;
                    if (fabsf(p) > (1.0f + NT_FUZZ)) { ok = false; break; }
                }
            }
            if (!ok) continue;
            if (t >= cutoff) return 0.0f;
            dist = t;
            for (int j = 0; j < n; ++j) nd[j] = j == i ? s : 0.0f;
            break;
        }
    } else {
        float a = 0.0f, b = 0.0f, c = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float pa = ld[k] * ld[k], pb = ld[k] * lo[k], pc = lo[k] * lo[k];
            a = k == 0 ? pa : a + pa;
            b = k == 0 ? pb : b + pb;
            c = k == 0 ? pc : c + pc;
        }
        b = 2.0f * b;
        c = c - 1.0f;
        const float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return 0.0f;
        const float t = (-b - sqrtf(disc)) / (2.0f * a);
        if (t <= 0.0f || t >= cutoff) return 0.0f;
        dist = t;
        for (int j = 0; j < n; ++j) { no[j] = lo[j] + ld[j] * t; nd[j] = no[j]; }
    }
    if (dist == 0.0f) return 0.0f;
    // (lo, ld are done with: they hold the local normal ray while it is turned back to world space)
    for (int i = 0; i < n; ++i) { lo[i] = no[i] + pos[i]; ld[i] = nd[i]; }
    for (int i = 0; i < n; ++i) {
        float so = 0.0f, sd = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float m = orient[i * n + k];
            const float po = m * lo[k], pd = m * ld[k];
            so = k == 0 ? po : so + po;
            sd = k == 0 ? pd : sd + pd;
        }
        no[i] = so;
        nd[i] = sd;
    }
    return dist;
}

__device__ __forceinline__ float test_item_marks_var(const VarCtx &cx, int item, float cutoff, int skip_item, int skip_lane, int &lane_out,
                                                     float *no, float *nd) {
    if ((item & 3) == 2) {
        lane_out = -1;
        return solid_marks_var(cx, item >> 2, cutoff, no, nd);
    }
    const float t = test_item_var(cx, item, cutoff, skip_item, skip_lane, lane_out);
    if (t != 0.0f) {
        Hit h;
        h.dist = t;
        h.item = item;
        h.lane = lane_out;
        hit_normal_var(cx, h, no, nd);
    }
    return t;
}

// leaf_closest_t (nt_composite.hpp): kd_leaf<Store,true>::intersects with transparent hits (tracer.hpp:977-1086)
template <bool ALIAS>
__device__ __noinline__ bool leaf_closest_var_t(const VarCtx &cx, int start, int count, int skip_item, int skip_lane, Hit &hit, TList &th,
                                                const Checked &ck, float *hn_o, float *hn_d, float *nn_o, float *nn_d) {
    const NtCompositeDev &sc = cx.sc;
    const int h_start = th.n;
    bool found = false;
    float dist_last = 0.0f;
    for (int i = 0; i < count; ++i) {
        const int item = sc.items[start + i];
        if ((item & 3) != 0 && item == skip_item) continue;
        if (checked_seen(ck, item)) continue;
        int l;
        float t;
        if (ALIAS) {
            if (!found) {
                t = test_item_marks_var(cx, item, hit.dist, skip_item, skip_lane, l, hn_o, hn_d);
            } else {
                t = test_item_marks_var(cx, item, hit.dist, skip_item, skip_lane, l, nn_o, nn_d);
                if (t != 0.0f && material_of(sc, item, l)[6] >= 1.0f) {
                    for (int k = 0; k < cx.n; ++k) { hn_o[k] = nn_o[k]; hn_d[k] = nn_d[k]; }
                }
            }
        } else {
            t = test_item_var(cx, item, hit.dist, skip_item, skip_lane, l);
        }
        dist_last = t;
        if (t != 0.0f) {
            if (material_of(sc, item, l)[6] >= 1.0f) {
                hit.dist = t;
                hit.item = item;
                hit.lane = l;
                if (!found) {
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

// trace_closest_t (nt_composite.hpp) for the current ray
template <bool ALIAS>
__device__ __noinline__ bool trace_closest_var_t(const VarCtx &cx, float t_near, int skip_item, int skip_lane, Hit &hit, TList &th,
                                                 const Checked &ck, float *hn_o, float *hn_d, float *nn_o, float *nn_d) {
    const NtCompositeDev &sc = cx.sc;
    const WaveLds &w = cx.w;
    const int lane = cx.lane;
    hit.dist = FLT_MAX;
    hit.item = -1;
    hit.lane = -1;
    th.n = 0;
    checked_reset(ck);
    int node = sc.root, sp = 0, dirty = 0;
    float t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                if (leaf_closest_var_t<ALIAS>(cx, nd.left, nd.right, skip_item, skip_lane, hit, th, ck, hn_o, hn_d, nn_o, nn_d)) dirty = sp;
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
                if (improved) tl_trim(th, hit.dist, h_start);
                continue;
            }
            const NtNode nd = sc.nodes[e & 0xffffffu];
            bool gt;
            const float t = branch_t(w, lane, nd, gt);
            const int far = gt ? nd.left : nd.right;
            if ((improved && hit.dist <= t) || far < 0) continue;
            if (improved) {
                w.stack[sp * 64 + lane] = (int)(e | NT_STK_MARK);
                ++sp;
            }
            node = far;
            t_near = t;
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

// trace_occluded_t (nt_composite.hpp) for the current ray: transparent hits are collected, an opaque one blocks
__device__ __noinline__ bool trace_occluded_var_t(const VarCtx &cx, float ldistance, int skip_item, int skip_lane, TList &sh) {
    const NtCompositeDev &sc = cx.sc;
    const WaveLds &w = cx.w;
    const int lane = cx.lane;
    sh.n = 0;
    int node = sc.root, sp = 0;
    float t_near = 0.0f, t_far = FLT_MAX;
    const int max_sp = sc.stack_depth;
    for (;;) {
        while (node >= 0) {
            const NtNode nd = sc.nodes[node];
            if (nd.axis < 0) {
                for (int i = 0; i < nd.right; ++i) {
                    const int item = sc.items[nd.left + i];
                    if ((item & 3) != 0 && item == skip_item) continue;
                    int l;
                    const float t = test_item_var(cx, item, ldistance, skip_item, skip_lane, l);
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

// light_reaches (tracer.hpp:1750-1766); the shadow ray is the current ray
__device__ __forceinline__ bool light_reaches_var_t(const VarCtx &cx, float ldistance, int skip_item, int skip_lane, Color3 &filtered) {
    TList sh;
    if (trace_occluded_var_t(cx, ldistance, skip_item, skip_lane, sh)) return false;
    if (sh.n) {
        tl_sort_unique(sh);
        for (int i = sh.n - 1; i >= 0; --i) filtered = cscale(filtered, 1.0f - material_of(cx.sc, sh.e[i].item, sh.e[i].lane)[6]);
    }
    return true;
}

// composite_color_t (nt_composite.hpp): ray_color / base_color as a state machine over the frame stack.  On entry the
// primary ray is the current ray; `nframes` frames are there (max_reflect_depth + 1 when anything reflects).
template <bool ALIAS>
__device__ __noinline__ Color3 composite_color_var_t(const VarCtx &cx, const VarFrames &fr, const Checked &ck, int nframes) {
    const NtCompositeDev &sc = cx.sc;
    const int n = cx.n;
    const int S = 4 * n;                  // first scalar word of a frame
    float a0[NT_DEV_MAX_DIM], a1[NT_DEV_MAX_DIM], a2[NT_DEV_MAX_DIM], a3[NT_DEV_MAX_DIM];
    for (int k = 0; k < n; ++k) { FRW(0, k) = VO(k); FRW(0, n + k) = VD(k); }
    FRI(0, S + VF_DEPTH) = 0;
    FRI(0, S + VF_SKIP_ITEM) = -1;
    FRI(0, S + VF_SKIP_LANE) = -1;
    int fp = 0;
    int state = 0;          // 0: trace the frame's ray, 1: shade its next surface, 2: a reflection has returned
    Color3 result = c3(0.0f, 0.0f, 0.0f);
    for (;;) {
        if (state == 0) {
            // ---- ray_color: intersect (tracer.hpp:1861-1868); the frame's ray becomes the current ray
            float *hn_o = a0, *hn_d = a1;
            for (int k = 0; k < n; ++k) {
                const float dk = FRW(fp, n + k);
                cx.L.dv[k * 64 + cx.lane] = dk;
                cx.L.ray[k * 64 + cx.lane] = make_float2(FRW(fp, k), dk != 0.0f ? 1.0f / dk : __int_as_float(0x7fc00000));
                hn_o[k] = 0.0f;
                hn_d[k] = 0.0f;
            }
            TList th;
            th.n = 0;
            Hit hit;
            hit.item = -1; hit.lane = -1; hit.dist = FLT_MAX;
            const float dist = aabb_distance_var(cx);
            if (dist >= 0.0f)
                trace_closest_var_t<ALIAS>(cx, dist, FRI(fp, S + VF_SKIP_ITEM), FRI(fp, S + VF_SKIP_LANE), hit, th, ck, hn_o, hn_d, a2, a3);
            if (ALIAS) {
                for (int k = 0; k < n; ++k) { FRW(fp, 2 * n + k) = hn_o[k]; FRW(fp, 3 * n + k) = hn_d[k]; }
            }
            tl_sort_unique(th);
            FRW(fp, S + VF_SURF) = hit.dist;
            FRI(fp, S + VF_SURF + 1) = hit.item;
            FRI(fp, S + VF_SURF + 2) = hit.lane;
            for (int i = 0; i < th.n; ++i) {          // farthest first (:1874)
                const THit e = th.e[th.n - 1 - i];
                FRW(fp, S + VF_SURF + 3 * (1 + i)) = e.dist;
                FRI(fp, S + VF_SURF + 3 * (1 + i) + 1) = e.item;
                FRI(fp, S + VF_SURF + 3 * (1 + i) + 2) = e.lane;
            }
            FRI(fp, S + VF_NSURF) = 1 + th.n;
            FRI(fp, S + VF_J) = 0;
            fr_put3(fr, fp, S + VF_R, c3(0.0f, 0.0f, 0.0f));
            state = 1;
            continue;
        }
        const int j = FRI(fp, S + VF_J);
        Color3 col;
        float opacity = 1.0f;
        if (state == 1) {
            if (j == FRI(fp, S + VF_NSURF)) {
                // ---- the frame is complete: hand its colour to the waiting base_color, or finish
                result = fr_get3(fr, fp, S + VF_R);
                if (fp == 0) break;
                --fp;
                state = 2;
                continue;
            }
            Hit hit;
            hit.dist = FRW(fp, S + VF_SURF + 3 * j);
            hit.item = FRI(fp, S + VF_SURF + 3 * j + 1);
            hit.lane = FRI(fp, S + VF_SURF + 3 * j + 2);
            float *dir = a0, *no = a1, *nd = a2, *lv = a3;
            for (int k = 0; k < n; ++k) dir[k] = FRW(fp, n + k);
            if (hit.item < 0) {            // miss: background (only surface 0 can be this)
                const float iv = dir[sc.bg_axis];
                fr_put3(fr, fp, S + VF_R, iv >= 0.0f ? cadd(cscale(c3p(sc.bg1), iv), cscale(c3p(sc.bg2), 1.0f - iv))
                                                     : cadd(cscale(c3p(sc.bg3), -iv), cscale(c3p(sc.bg2), 1.0f + iv)));
                FRI(fp, S + VF_J) = j + 1;
                continue;
            }
            // ---- base_color (tracer.hpp:1768-1854)
            if (ALIAS && j == 0) {
                // the opaque hit is shaded with o_hit.normal as the walk left it (tracer.hpp:1864)
                for (int k = 0; k < n; ++k) { no[k] = FRW(fp, 2 * n + k); nd[k] = FRW(fp, 3 * n + k); }
            } else {
                // (shadow rays and reflections have replaced the current ray since the frame was traced)
                for (int k = 0; k < n; ++k) {
                    cx.L.dv[k * 64 + cx.lane] = dir[k];
                    cx.L.ray[k * 64 + cx.lane].x = FRW(fp, k);
                }
                hit_normal_var(cx, hit, no, nd);
            }
            const float *m = material_of(sc, hit.item, hit.lane);
            opacity = m[6];
            Color3 light = c3(0.0f, 0.0f, 0.0f), specular = c3(0.0f, 0.0f, 0.0f);
            float spec_a = 0.0f;
            for (int li = 0; li < sc.n_point_lights; ++li) {
                const float *pos = sc.pl_pos + (size_t)li * n;
                const Color3 plc = c3p(sc.pl_color + 3 * li);
                for (int k = 0; k < n; ++k) lv[k] = no[k] - pos[k];
                const float ldist = sqrtf(dot_var(n, lv, lv));
                for (int k = 0; k < n; ++k) lv[k] = lv[k] / ldist;
                const float sine = dot_var(n, nd, lv);
                if (sine > 0.0f) {
                    const float strength = nt_falloff(ldist, n - 1);
                    if (sc.shadows) {
                        if (fmaxf(plc.r, fmaxf(plc.g, plc.b)) * strength * sine > NT_LIGHT_THRESHOLD) {
                            Color3 filtered = plc;
                            var_set_ray(cx, no, lv);
                            if (light_reaches_var_t(cx, ldist, hit.item, hit.lane, filtered)) {
                                filtered = cscale(filtered, strength);
                                light = cadd(light, cscale(filtered, sine));
                                if (m[8] != 0.0f) append_specular_var(n, specular, spec_a, m, filtered, dir, nd, lv);
                            }
                        }
                    } else {
                        light = cadd(light, cscale(cscale(plc, strength), sine));
                    }
                }
            }
            for (int li = 0; li < sc.n_global_lights; ++li) {
                const float *gd = sc.gl_dir + (size_t)li * n;
                const Color3 glc = c3p(sc.gl_color + 3 * li);
                const float sine = -dot_var(n, nd, gd);
                if (sine > 0.0f) {
                    if (sc.shadows) {
                        for (int k = 0; k < n; ++k) lv[k] = -gd[k];
                        Color3 filtered = glc;
                        var_set_ray(cx, no, lv);
                        if (light_reaches_var_t(cx, FLT_MAX, hit.item, hit.lane, filtered)) {
                            light = cadd(light, cscale(filtered, sine));
                            if (m[8] != 0.0f) append_specular_var(n, specular, spec_a, m, filtered, dir, nd, lv);
                        }
                    } else {
                        light = cadd(light, cscale(glc, sine));
                    }
                }
            }
            const float sine = -dot_var(n, dir, nd);
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
            const int depth = FRI(fp, S + VF_DEPTH);
            if (m[7] != 0.0f && depth < sc.max_reflect_depth && fp + 1 < nframes) {
                fr_put3(fr, fp, S + VF_SPEC, specular);
                FRW(fp, S + VF_SPEC_A) = spec_a;
                fr_put3(fr, fp, S + VF_R0, r0);
                fr_put3(fr, fp, S + VF_C, c3p(m));
                FRW(fp, S + VF_REFL) = m[7];
                const float f = -2.0f * sine;
                for (int k = 0; k < n; ++k) { FRW(fp + 1, n + k) = dir[k] - nd[k] * f; FRW(fp + 1, k) = no[k]; }
                FRI(fp + 1, S + VF_DEPTH) = depth + 1;
                FRI(fp + 1, S + VF_SKIP_ITEM) = hit.item;
                FRI(fp + 1, S + VF_SKIP_LANE) = hit.lane;
                ++fp;
                state = 0;
                continue;
            }
            col = cadd(specular, cscale(r0, 1.0f - spec_a));
        } else {
            // ---- state 2: the reflection of surface j returned `result` (tracer.hpp:1842-1853)
            const float refl = FRW(fp, S + VF_REFL);
            const Color3 r = cadd(cscale(cmul(fr_get3(fr, fp, S + VF_C), result), refl), cscale(fr_get3(fr, fp, S + VF_R0), 1.0f - refl));
            col = cadd(fr_get3(fr, fp, S + VF_SPEC), cscale(r, 1.0f - FRW(fp, S + VF_SPEC_A)));
            opacity = material_of(sc, FRI(fp, S + VF_SURF + 3 * j + 1), FRI(fp, S + VF_SURF + 3 * j + 2))[6];
            state = 1;
        }
        // ---- ray_color: the opaque hit is the base, transparent hits are blended over it (:1864, :1878)
        if (j == 0) fr_put3(fr, fp, S + VF_R, col);
        else fr_put3(fr, fp, S + VF_R, cadd(cscale(col, opacity), cscale(fr_get3(fr, fp, S + VF_R), 1.0f - opacity)));
        FRI(fp, S + VF_J) = j + 1;
    }
    return result;
}

#undef FRW
#undef FRI

// The blocks stride over the 8x8-pixel tiles of the launch (the scratch -- `checked` columns and frame stacks -- is sized by
// the grid, not by the image), one wave per block as in composite_kernel_var.
template <bool ALIAS>
__global__ __launch_bounds__(64) void composite_kernel_var_t(NtCamera cam, NtCompositeDev sc, NtTarget tg, int n, int tiles_x, int tiles_y,
                                                             int frames) {
    extern __shared__ float2 lds_raw[];
    const int lane = (int)threadIdx.x;
    VarLds L;
    {
        char *p = reinterpret_cast<char *>(lds_raw);
        L.ray = reinterpret_cast<float2 *>(p);
        L.dv = reinterpret_cast<float *>(p + (size_t)64 * n * 8);
        L.ps = L.dv + (size_t)64 * n;
        L.stack = reinterpret_cast<int *>(L.ps + (size_t)64 * n);
        L.mbox = L.stack + (size_t)64 * sc.stack_depth;
    }
    WaveLds w;
    w.ray = L.ray;
    w.stack = L.stack;
    w.mbox = L.mbox;
    const long long slot = (long long)blockIdx.x * 64 + lane;
    Checked ck;
    ck.bits = sc.checked + slot;
    ck.stride = sc.checked_lanes;
    ck.words = sc.checked_words;
    ck.n_batches = sc.n_batches;
    ck.n_triangles = sc.n_triangles;
    VarFrames fr;
    fr.base = sc.tframes + slot;
    fr.stride = sc.checked_lanes;
    fr.fw = var_frame_words(n);
    const VarCtx cx = {sc, L, w, n, lane};
    const long long total = (long long)tiles_x * tiles_y * frames;
    for (long long tile = (long long)blockIdx.x; tile < total; tile += gridDim.x) {
        if (nt_aborted(tg)) return;                       // (one wave a block)
        const int bz = (int)(tile / ((long long)tiles_x * tiles_y));
        const int rem = (int)(tile - (long long)bz * tiles_x * tiles_y);
        const int by = rem / tiles_x;
        const int bx = rem - by * tiles_x;
        const PixelRef pr = tg.colors_out ? locate_pixel_at<8, 8>(tg, bx, by, bz, 0, 0, lane)
                                          : locate_pixel_at<8, 8>(tg, bx, by, bz, lane & 7, lane >> 3, lane);
        if (!pr.valid) continue;
        const float *c = cam.buf ? cam.buf + (size_t)bz * 4 * n : nullptr;
        // ---- primary ray (tracer.hpp:60-76) becomes the current ray
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
        const Color3 col = composite_color_var_t<ALIAS>(cx, fr, ck, sc.tframe_count);
        emit_pixel(tg, pr, col.r, col.g, col.b);
    }
}

__global__ __launch_bounds__(64) void composite_kernel_var(NtCamera cam, NtCompositeDev sc, NtTarget tg, int n) {
    extern __shared__ float2 lds_raw[];
    if (nt_aborted(tg)) return;
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

    // ---- primary ray (tracer.hpp:60-76) becomes the current ray
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
    const VarCtx cx = {sc, L, w, n, lane};
    const Color3 col = composite_color_var(cx);
    emit_pixel(tg, pr, col.r, col.g, col.b);
}

#undef VO
#undef VD

}  // namespace

// The camera table of a multi-frame launch, from pinned host memory (which the device reads in place) to device memory:
// a kernel on the launch stream instead of a copy-engine transfer the next kernel would have to wait for across queues.
namespace {
__global__ __launch_bounds__(256) void upload_kernel(const float *__restrict__ src, float *__restrict__ dst, int count) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i < count) dst[i] = src[i];
}
}  // namespace

int nt_launch_upload(void *stream, const float *src_pinned, float *dst, int count) {
    hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_pinned, dst, count);
    return finish_launch("camera upload");
}

// NTRACER_FORCE_VAR=1: use the run-time-n kernels for every dimension (they are the only ones above
// NT_DEV_MAX_FIXED; the switch lets tests compare them with the compile-time-N kernels on the same scene)
static bool force_var() {
    const char *e = getenv("NTRACER_FORCE_VAR");
    return e && atoi(e) != 0;
}

int nt_launch_box(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg) {
    switch (force_var() ? 0 : li.n) {
        case 3: nt_box_fixed_3(li, cam, tg); break;
        case 4: nt_box_fixed_4(li, cam, tg); break;
        case 5: nt_box_fixed_5(li, cam, tg); break;
        case 6: nt_box_fixed_6(li, cam, tg); break;
        case 7: nt_box_fixed_7(li, cam, tg); break;
        case 8: nt_box_fixed_8(li, cam, tg); break;
        case 9: nt_box_fixed_9(li, cam, tg); break;
        case 10: nt_box_fixed_10(li, cam, tg); break;
        case 11: nt_box_fixed_11(li, cam, tg); break;
        case 12: nt_box_fixed_12(li, cam, tg); break;
        case 13: nt_box_fixed_13(li, cam, tg); break;
        case 14: nt_box_fixed_14(li, cam, tg); break;
        case 15: nt_box_fixed_15(li, cam, tg); break;
        case 16: nt_box_fixed_16(li, cam, tg); break;
        case 17: nt_box_fixed_17(li, cam, tg); break;
        case 18: nt_box_fixed_18(li, cam, tg); break;
        case 19: nt_box_fixed_19(li, cam, tg); break;
        case 20: nt_box_fixed_20(li, cam, tg); break;
        case 21: nt_box_fixed_21(li, cam, tg); break;
        case 22: nt_box_fixed_22(li, cam, tg); break;
        case 23: nt_box_fixed_23(li, cam, tg); break;
        case 24: nt_box_fixed_24(li, cam, tg); break;
        default: {
            // packed plain RGB of <= 10 bits in one aligned dword: the rows kernel (codes + lean loops), if its n-vectors fit LDS
            const size_t lds_rows = ((size_t)li.n * 256 + (size_t)4 * li.n + 4) * sizeof(float);
            const char *er = getenv("NTRACER_BOX_VAR_ROWS");
            if (!tg.colors_out && tg.plain_bits != 0u && tg.plain_bits <= 10u && tg.bpp == 4 && tg.aligned4 && lds_rows <= 160 * 1024 &&
                !(er && atoi(er) == 0)) {
                const dim3 grid((unsigned)((tg.width + 63) / 64), (unsigned)((tg.row_count + 31) / 32), (unsigned)li.nframes);
                if (lds_rows > 64 * 1024)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(box_rows_kernel_var), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
                hipLaunchKernelGGL(box_rows_kernel_var, grid, dim3(256), lds_rows, (hipStream_t)li.stream, cam, tg);
                break;
            }
            dim3 grid;
            grid_for(tg, 64, 4, li.nframes, grid);
            const size_t lds = ((size_t)li.n * 256 + (size_t)4 * li.n) * sizeof(float);
            hipLaunchKernelGGL(box_kernel_var, grid, dim3(256), lds, (hipStream_t)li.stream, cam, tg);
        }
    }
    return finish_launch("box kernel launch");
}

// words per ray_color frame of composite_kernel_var_t (the host sizes NtCompositeDev::tframes with it)
int nt_var_frame_words(int n) { return var_frame_words(n); }

int nt_launch_composite(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg) {
    int r;
    if (sc.tframes) {
        // transparent materials / the reference's normal handling at run-time n (or beyond the fixed kernels' frame stack)
        if (li.n < 3 || li.n > NT_DEV_MAX_DIM || !sc.checked || sc.checked_lanes < 64) {
            snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "run-time-n transparency kernel: bad launch (n %d)", li.n);
            return -2;
        }
        const size_t lds = (size_t)64 * ((size_t)li.n * 16 + (size_t)sc.stack_depth * 4 + (size_t)NT_MBOX * 4);
        if (lds > 160 * 1024) {
            snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "scene too deep for the LDS budget (n %d, depth %d)", li.n, sc.stack_depth);
            return -1;
        }
        dim3 grid;
        grid_for(tg, 8, 8, li.nframes, grid);
        const unsigned blocks = (unsigned)(sc.checked_lanes / 64);
        if (sc.alias_normals) {
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(composite_kernel_var_t<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(composite_kernel_var_t<true>, dim3(blocks), dim3(64), lds, (hipStream_t)li.stream, cam, sc, tg, li.n, (int)grid.x,
                               (int)grid.y, (int)grid.z);
        } else {
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(composite_kernel_var_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(composite_kernel_var_t<false>, dim3(blocks), dim3(64), lds, (hipStream_t)li.stream, cam, sc, tg, li.n, (int)grid.x,
                               (int)grid.y, (int)grid.z);
        }
        return finish_launch("composite kernel launch");
    }
    switch (force_var() ? 0 : li.n) {
        case 3: r = nt_composite_fixed_3(li, cam, sc, tg); break;
        case 4: r = nt_composite_fixed_4(li, cam, sc, tg); break;
        case 5: r = nt_composite_fixed_5(li, cam, sc, tg); break;
        case 6: r = nt_composite_fixed_6(li, cam, sc, tg); break;
        case 7: r = nt_composite_fixed_7(li, cam, sc, tg); break;
        case 8: r = nt_composite_fixed_8(li, cam, sc, tg); break;
        case 9: r = nt_composite_fixed_9(li, cam, sc, tg); break;
        case 10: r = nt_composite_fixed_10(li, cam, sc, tg); break;
        default: {
            if (li.n < 3 || li.n > NT_DEV_MAX_DIM) {
                snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "unsupported dimension %d", li.n);
                return -2;
            }
            if (!sc.all_opaque) {
                snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "the run-time-n kernel does not render transparent materials");
                return -2;
            }
            // run-time-n kernel: one wave per 8x8 tile (probe mode: 64 probes per block)
            dim3 grid;
            grid_for(tg, 8, 8, li.nframes, grid);
            const size_t lds = (size_t)64 * ((size_t)li.n * 16 + (size_t)sc.stack_depth * 4 + (size_t)NT_MBOX * 4);
            if (lds > 160 * 1024) {
                snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "scene too deep for the LDS budget (n %d, depth %d)", li.n, sc.stack_depth);
                return -1;
            }
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(composite_kernel_var), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(composite_kernel_var, grid, dim3(64), lds, (hipStream_t)li.stream, cam, sc, tg, li.n);
            r = 0;
        }
    }
    if (r) return r;
    return finish_launch("composite kernel launch");
}