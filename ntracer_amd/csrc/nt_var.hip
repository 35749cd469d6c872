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

}  // namespace

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
            if (sc.n_point_lights || sc.n_global_lights || sc.any_reflective || sc.n_solids || !sc.all_opaque) {
                snprintf(nt_launch_error_buf(), NT_LAUNCH_ERROR_LEN, "the run-time-n kernel renders opaque, non-reflective simplices lit by the camera light only");
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
            hipLaunchKernelGGL(composite_kernel_var, grid, dim3(64), lds, (hipStream_t)li.stream, cam, sc, tg, li.n);
            r = 0;
        }
    }
    if (r) return r;
    return finish_launch("composite kernel launch");
}