// nt_api.cpp -- host side of libntracer_hip.so: the C ABI declared in include/ntracer_hip.h.
//
// Mirrors, for the ray-cast path only, what the reference does in C++ above its scenes:
//   ImageFormat/Channel validation      src/render.cpp:120-164, 187-209, 249-288
//   renderer frame loop and protocol    src/render.cpp:853-923 (busy / lock / abort)
//   box_scene / composite_scene state   src/tracer.hpp:83-123, 1710-1748
// The per-pixel work itself is in nt_box.hpp / nt_composite.hpp / nt_var.hip.  There is NO CPU fallback: without a HIP
// device every render entry point fails with NT_E_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ntracer_hip.h"
#include "nt_device.hpp"

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? NT_E_NOMEM : NT_E_DEVICE, \
                                          "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 256);
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ChanTable {
    std::vector<NtChanDev> host;
    NtChanDev *dev = nullptr;
};

// NtTarget::rowtab: what the BoxScene tile kernel needs to know about a row, per owned row of one (view, band split, pitch) --
// or, with interleaved rows (NtTarget::row_il), per slot of one launch geometry
struct RowTable {
    int height = 0, pitch = 0, rank = 0, world = 1, rows = 0, compact = 0;
    int il = 0, il_rows = 0, il_count = 0;   // interleave stride (waves per column strip), rows a wave, rows of the launch
    uint32_t half_h = 0, fovI = 0;       // float bits
    void *dev = nullptr;
};

// everything a scene keeps on one HIP device
struct DeviceState {
    int device = -1;
    hipStream_t stream = nullptr;        // used by the host-buffer entry points
    // The per-device scratch of a scene (camera table, BoxScene's redo bitmap, hit records ...) is shared by its launches, which
    // are ordered by their stream.  A call that arrives on another stream than the one before first waits (on the host) for
    // that stream to drain: see use_stream
    hipStream_t last_stream = nullptr;
    bool have_last_stream = false;
    // nt_render's abort: a dword in device memory (NtTarget::abort_word: read past the caches by every block that starts, a
    // microsecond from HBM -- from mapped host memory the same read made a 120-cell frame four times as long) that the host
    // raises, by a 4-byte copy on a stream of its own, when the caller's flag goes up; and the event the host waits on
    DevBuf abort_word;
    int *abort_one = nullptr;            // pinned source of that copy: the value 1
    hipStream_t side_stream = nullptr;
    hipEvent_t frame_done = nullptr;
    bool scene_uploaded = false;
    DevBuf nodes, items, batch_recs, batch_mats, tri_recs, tri_mats, solid_recs, solid_types, solid_mats, materials, aabb;
    DevBuf lights;                       // pl_pos | pl_color | gl_dir | gl_color
    unsigned long long lights_version = 0;
    DevBuf framebuffer, cams, probes, stats, counter;
    DevBuf hits;                         // primary-hit records between the two passes of a lit render
    DevBuf stats_frame;                  // where the statistics launch of a "faithful" scene draws (discarded)
    DevBuf numer;                        // packet kernel: -(N.o + d) per (frame, simplex)
    DevBuf cull;                         // BoxScene: row culling bits
    bool cull_clean = false;             // `cull` is all zero (what the fused BoxScene path needs and leaves behind)
    DevBuf checked;                      // reference-faithful normals: the exact `checked` bitmap, one column per resident lane
    DevBuf ties;                         // BoxScene: tie sets of the marked stretches (fused path)
    DevBuf tframes;                      // run-time-n transparency kernel: the ray_color frame stacks, one column per resident lane
    // camera tables travel through pinned host memory (a pageable source makes hipMemcpyAsync wait for the copy on the
    // host, which stalls the launch pipeline of back-to-back calls): a ring of slots, each guarded by an event
    struct Stage { void *host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool in_flight = false; };
    Stage stage[8];
    unsigned stage_next = 0;
    struct TileOrder { int tx = 0, ty = 0; DevBuf buf; };
    std::vector<std::unique_ptr<TileOrder>> tile_orders;   // packet kernel: tiles sorted centre-out, per tile grid
    int cu_count = 0;
    std::vector<std::unique_ptr<ChanTable>> chan_tables;
    std::vector<std::unique_ptr<RowTable>> row_tables;
};

struct Format {                          // validated image_format (render.cpp:167-172)
    int width = 0, height = 0, pitch = 0, bpp = 0, reversed = 0;
    int pack_mode = NT_PACK_GENERIC;
    // "plain RGB" layouts (every live channel is exactly one of r, g, b, same bit size, one 32-bit word): the three
    // multipliers place a quantised component into all the fields that carry it; 0 bits = not such a layout
    uint32_t plain_bits = 0, plain_maxval = 0, plain_mul[3] = {0, 0, 0};
    // three fp32 channels, each exactly one of r, g, b (12-byte pixels): component of float k, or -1 if not this layout
    int plain_f32[3] = {-1, -1, -1};
    std::vector<NtChanDev> chans;        // live channels only (all-zero channels contribute no bits)
};

}  // namespace

struct nt_scene {
    bool composite = false;
    int n = 0;
    std::mutex mu;
    int locked = 0;
    bool busy = false;
    float fov = 0.8f;                    // tracer.hpp:91,1731
    std::vector<float> origin, axes;     // camera<Store>: origin[n], t_orientation[n][n] (camera.hpp:7-15)

    // composite_scene (tracer.hpp:1713-1740)
    int root = -1;
    int depth = 0;
    std::vector<NtNode> nodes;
    std::vector<int32_t> items;
    int rec_len = 0, rec_stride = 0;
    std::vector<float> batch_recs, tri_recs, solid_recs, materials, aabb;
    std::vector<int32_t> batch_mats, tri_mats, solid_types, solid_mats;
    int n_batches = 0, n_triangles = 0, n_solids = 0, n_materials = 0;
    bool all_opaque = true, any_reflective = false, has_scalar = false;
    int shadows = 0, camera_light = 1, max_reflect_depth = 4, bg_axis = 1;
    float ambient[3] = {0, 0, 0}, bg1[3] = {1, 1, 1}, bg2[3] = {0, 0, 0}, bg3[3] = {0, 1, 1};
    std::vector<float> pl_pos, pl_color, gl_dir, gl_color;
    unsigned long long lights_version = 1;

    std::map<int, std::unique_ptr<DeviceState>> devs;
    nt_stats last_stats{};
    bool have_stats = false;
    int stats_device = -1;
};

namespace {

// ---------------------------------------------------------------------------------------------
// formats: Channel.__new__ (render.cpp:120-164), im_set_channels (:192-209), ImageFormat.__new__
// (:249-288), im_check_buffer_size (:187-190)
// ---------------------------------------------------------------------------------------------
int parse_format(const nt_image_format *f, Format &out) {
    if (!f) return fail(NT_E_INVALID, "format is NULL");
    if (f->nchannels < 0 || (f->nchannels > 0 && !f->channels)) return fail(NT_E_INVALID, "invalid channel list");
    long bits = 0;
    out.chans.clear();
    for (int i = 0; i < f->nchannels; ++i) {
        const nt_channel &c = f->channels[i];
        if (c.tfloat) {
            if (c.bit_size != 32) return fail(NT_E_INVALID, "if \"tfloat\" is true, \"bit_size\" can only be 32");
        } else {
            if (c.bit_size > NT_MAX_BITSIZE) return fail(NT_E_INVALID, "\"bit_size\" cannot be greater than %d (unless \"tfloat\" is true)", NT_MAX_BITSIZE);
            if (c.bit_size < 1) return fail(NT_E_INVALID, "\"bit_size\" cannot be less than 1");
        }
        NtChanDev d;
        d.f_r = c.f_r; d.f_g = c.f_g; d.f_b = c.f_b; d.f_c = c.f_c;
        d.bits = c.bit_size;
        d.tfloat = c.tfloat ? 1u : 0u;
        d.offset = (uint32_t)bits;
        d.maxval = c.tfloat ? 0u : (0xffffffffu >> (32 - c.bit_size));
        bits += c.bit_size;
        // clamp((0*g + 0*b) + (0*r + 0)) is 0 for every colour (NaN included): such a channel (the X of
        // RGBX8, padding) writes no bits, so it is dropped from the device table
        const bool dead = c.f_r == 0.0f && c.f_g == 0.0f && c.f_b == 0.0f && c.f_c == 0.0f;
        if (!dead) out.chans.push_back(d);
    }
    if (bits > NT_MAX_PIXELSIZE * 8) return fail(NT_E_INVALID, "Too many bytes per pixel. The maximum is %d.", NT_MAX_PIXELSIZE);
    out.bpp = (int)((bits + 7) / 8);
    out.pack_mode = NT_PACK_GENERIC;
    if (out.chans.size() <= 4 && bits <= 32) out.pack_mode = NT_PACK_WORD32;
    else if (out.chans.size() <= 4 && bits <= 64) out.pack_mode = NT_PACK_WORD64;
    if (out.chans.size() == 3 && bits == 96) {
        int comp[3] = {-1, -1, -1};
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            const NtChanDev &d = out.chans[k];
            const float f[3] = {d.f_r, d.f_g, d.f_b};
            int ones = 0, zeros = 0;
            for (int c = 0; c < 3; ++c) {
                if (f[c] == 1.0f) { comp[k] = c; ++ones; }
                else if (f[c] == 0.0f && !std::signbit(f[c])) ++zeros;
            }
            ok = ones == 1 && zeros == 2 && d.f_c == 0.0f && !std::signbit(d.f_c) && d.tfloat && d.offset == (uint32_t)(32 * k);
        }
        if (ok) for (int k = 0; k < 3; ++k) out.plain_f32[k] = comp[k];
    }
    if (out.pack_mode == NT_PACK_WORD32 && !out.chans.empty()) {
        bool plain = true;
        uint32_t mul[3] = {0, 0, 0};
        for (const NtChanDev &d : out.chans) {
            const float f[3] = {d.f_r, d.f_g, d.f_b};
            int comp = -1, ones = 0, zeros = 0;
            for (int k = 0; k < 3; ++k) {
                if (f[k] == 1.0f) { comp = k; ++ones; }
                else if (f[k] == 0.0f && !std::signbit(f[k])) ++zeros;
            }
            if (ones != 1 || zeros != 2 || d.f_c != 0.0f || std::signbit(d.f_c) || d.tfloat || d.bits != out.chans[0].bits) { plain = false; break; }
            mul[comp] |= 1u << (32u - d.offset - d.bits);
        }
        if (plain) {
            out.plain_bits = out.chans[0].bits;
            out.plain_maxval = out.chans[0].maxval;
            for (int k = 0; k < 3; ++k) out.plain_mul[k] = mul[k];
        }
    }
    if (f->width < 1 || f->height < 1) return fail(NT_E_INVALID, "width and height must be at least 1");
    if (f->pitch < 0) return fail(NT_E_INVALID, "pitch cannot be negative");
    out.width = f->width;
    out.height = f->height;
    out.reversed = f->reversed ? 1 : 0;
    if (f->pitch) {
        if (f->pitch < f->width * out.bpp) return fail(NT_E_INVALID, "\"pitch\" must be at least \"width\" times the size of one pixel in bytes");
        out.pitch = f->pitch;
    } else {
        out.pitch = f->width * out.bpp;
    }
    return NT_OK;
}

struct Bands {
    int rank = 0, world = 1, rows = NT_RENDER_CHUNK_SIZE, compact = 0;
    int owned_rows = 0;
};

int parse_bands(const nt_render_opts *o, int height, Bands &b) {
    if (o) {
        b.world = o->band_world > 1 ? o->band_world : 1;
        b.rank = b.world > 1 ? o->band_rank : 0;
        b.rows = o->band_rows > 0 ? o->band_rows : NT_RENDER_CHUNK_SIZE;
        b.compact = o->compact ? 1 : 0;
        if (b.rank < 0 || b.rank >= b.world) return fail(NT_E_INVALID, "band_rank %d out of range for band_world %d", o->band_rank, b.world);
    }
    if (b.world == 1) {
        b.owned_rows = height;
    } else {
        // rows of bands rank, rank+world, ... clipped to the image
        int owned = 0;
        const int nbands = (height + b.rows - 1) / b.rows;
        for (int band = b.rank; band < nbands; band += b.world) owned += std::min(b.rows, height - band * b.rows);
        b.owned_rows = owned;
    }
    return NT_OK;
}

size_t required_len(const Format &f, const Bands &b) {
    return (size_t)f.pitch * (size_t)(b.compact ? b.owned_rows : f.height);
}

// ---------------------------------------------------------------------------------------------
// devices
// ---------------------------------------------------------------------------------------------
int pick_device(const nt_render_opts *o, int explicit_dev, int &dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1)
        return fail(NT_E_DEVICE, "no HIP device available (%s): the ray-cast path has no CPU fallback", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    dev = explicit_dev;
    if (o && o->device >= 0) dev = o->device;
    if (dev < 0) {
        HIP_TRY(hipGetDevice(&dev));
    }
    if (dev >= count) return fail(NT_E_INVALID, "device %d does not exist (%d devices)", dev, count);
    HIP_TRY(hipSetDevice(dev));
    return NT_OK;
}

int device_state(nt_scene *s, int dev, DeviceState *&out) {
    auto it = s->devs.find(dev);
    if (it == s->devs.end()) {
        auto ds = std::make_unique<DeviceState>();
        ds->device = dev;
        HIP_TRY(hipDeviceGetAttribute(&ds->cu_count, hipDeviceAttributeMultiprocessorCount, dev));
        it = s->devs.emplace(dev, std::move(ds)).first;
    }
    out = it->second.get();
    return NT_OK;
}

// The stream of the entry points that take none (nt_render, nt_colors_at), made when one of them first needs it: a scene that
// is only ever drawn through the device entry points -- on the caller's streams -- creates no stream of its own.  (The device
// has a handful of hardware queues and the runtime deals its streams out over them; streams nobody uses still take their turn,
// and two of the caller's streams that end up on one queue no longer overlap: a process that had made a dozen scenes lost the
// overlap of alternating calls, 48 -> 88 us a step.)
int own_stream(DeviceState *ds) {
    if (!ds->stream) HIP_TRY(hipStreamCreateWithFlags(&ds->stream, hipStreamNonBlocking));
    return NT_OK;
}

// launches of one scene on one device are ordered by their stream; when the stream changes, the old one is drained first
int use_stream(DeviceState *ds, hipStream_t st) {
    if (ds->have_last_stream && ds->last_stream != st) HIP_TRY(hipStreamSynchronize(ds->last_stream));
    ds->last_stream = st;
    ds->have_last_stream = true;
    return NT_OK;
}

// a pinned slot of at least `bytes`, free to be overwritten (its previous copy has left the host)
int stage_slot(DeviceState *ds, size_t bytes, DeviceState::Stage *&out) {
    DeviceState::Stage &st = ds->stage[ds->stage_next++ % 8];
    if (st.in_flight) {
        HIP_TRY(hipEventSynchronize(st.done));
        st.in_flight = false;
    }
    if (st.cap < bytes) {
        if (st.host) { (void)hipHostFree(st.host); st.host = nullptr; st.cap = 0; }
        const size_t want = std::max<size_t>(bytes, 4096);
        HIP_TRY(hipHostMalloc(&st.host, want, hipHostMallocDefault));
        st.cap = want;
    }
    if (!st.done) HIP_TRY(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    out = &st;
    return NT_OK;
}

template <typename T>
int upload(DevBuf &b, const std::vector<T> &v) {
    if (int r = b.ensure(std::max<size_t>(v.size() * sizeof(T), 16))) return r;
    if (!v.empty()) HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return NT_OK;
}

int upload_scene(nt_scene *s, DeviceState *ds) {
    if (!s->composite) return NT_OK;
    if (!ds->scene_uploaded) {
        if (int r = upload(ds->nodes, s->nodes)) return r;
        if (int r = upload(ds->items, s->items)) return r;
        if (int r = upload(ds->batch_recs, s->batch_recs)) return r;
        if (int r = upload(ds->batch_mats, s->batch_mats)) return r;
        if (int r = upload(ds->tri_recs, s->tri_recs)) return r;
        if (int r = upload(ds->tri_mats, s->tri_mats)) return r;
        if (int r = upload(ds->solid_recs, s->solid_recs)) return r;
        if (int r = upload(ds->solid_types, s->solid_types)) return r;
        if (int r = upload(ds->solid_mats, s->solid_mats)) return r;
        if (int r = upload(ds->materials, s->materials)) return r;
        if (int r = upload(ds->aabb, s->aabb)) return r;
        ds->scene_uploaded = true;
    }
    if (ds->lights_version != s->lights_version) {
        std::vector<float> all;
        all.insert(all.end(), s->pl_pos.begin(), s->pl_pos.end());
        all.insert(all.end(), s->pl_color.begin(), s->pl_color.end());
        all.insert(all.end(), s->gl_dir.begin(), s->gl_dir.end());
        all.insert(all.end(), s->gl_color.begin(), s->gl_color.end());
        // a fresh allocation: launches already enqueued keep reading the old one
        DevBuf fresh;
        if (int r = upload(fresh, all)) return r;
        // the previous buffer may still be in use by enqueued work: free it only after the device drains
        if (ds->lights.p) { (void)hipDeviceSynchronize(); ds->lights.release(); }
        ds->lights = fresh;
        ds->lights_version = s->lights_version;
    }
    return NT_OK;
}

int chan_table(DeviceState *ds, const Format &f, const NtChanDev *&dev_ptr) {
    for (auto &t : ds->chan_tables) {
        if (t->host.size() == f.chans.size() &&
            (f.chans.empty() || std::memcmp(t->host.data(), f.chans.data(), f.chans.size() * sizeof(NtChanDev)) == 0)) {
            dev_ptr = t->dev;
            return NT_OK;
        }
    }
    auto t = std::make_unique<ChanTable>();
    t->host = f.chans;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(f.chans.size() * sizeof(NtChanDev), 16)));
    t->dev = (NtChanDev *)p;
    if (!f.chans.empty()) HIP_TRY(hipMemcpy(p, f.chans.data(), f.chans.size() * sizeof(NtChanDev), hipMemcpyHostToDevice));
    dev_ptr = t->dev;
    ds->chan_tables.push_back(std::move(t));
    return NT_OK;
}

// The row table of a launch geometry (cached: a render loop keeps its view, band split and pitch).  Entry i belongs to owned
// row i: sy exactly as the ray source computes it (tracer.hpp:72-74: fovI * (y - half_h), two fp32 operations), whether
// the row exists, and where it starts in a frame; 64 entries of padding, as a wave reads its sixteen rows unclamped.
// Interleaved rows (tg.row_il = W > 0, il_rows = rows a wave): entry w * il_rows + rr belongs to row w + W * rr of the launch.
int row_table(DeviceState *ds, const NtTarget &tg, int il_rows, const void *&dev_ptr) {
    uint32_t hh, fi;
    std::memcpy(&hh, &tg.half_h, 4);
    std::memcpy(&fi, &tg.fovI, 4);
    const int world = std::max(tg.band_world, 1);
    const int il = tg.row_il;
    for (auto &t : ds->row_tables) {
        if (t->height == tg.height && t->pitch == tg.pitch && t->rank == tg.band_rank && t->world == world && t->rows == tg.band_rows &&
            t->compact == tg.compact && t->half_h == hh && t->fovI == fi && t->il == il &&
            (il == 0 || (t->il_rows == il_rows && t->il_count == tg.row_count))) {
            dev_ptr = t->dev;
            return NT_OK;
        }
    }
    struct Entry { float sy; uint32_t valid; long long off; };
    static_assert(sizeof(Entry) == 16, "16-byte row entries");
    std::vector<Entry> host;
    const int rows = std::max(tg.band_rows, 1);
    auto entry_of = [&](int orow, Entry &e) {            // false: no such band
        const int band = orow / rows;
        const int y = world > 1 ? (band * world + tg.band_rank) * rows + (orow - band * rows) : orow;
        if ((world > 1 ? (band * world + tg.band_rank) * rows : orow) >= tg.height) return false;
        e.sy = tg.fovI * ((float)y - tg.half_h);
        e.valid = y < tg.height ? 1u : 0u;
        e.off = (long long)(tg.compact ? orow : y) * tg.pitch;
        return true;
    };
    if (il > 0) {
        for (int w = 0; w < il; ++w)
            for (int rr = 0; rr < il_rows; ++rr) {
                Entry e{0.0f, 0u, 0};
                const int row = w + il * rr;
                if (row >= tg.row_count || !entry_of(tg.row_begin + row, e)) e = Entry{0.0f, 0u, 0};
                host.push_back(e);
            }
    } else {
        for (int orow = 0;; ++orow) {
            Entry e;
            if (!entry_of(orow, e)) break;
            host.push_back(e);
        }
    }
    for (int k = 0; k < 64; ++k) host.push_back(Entry{0.0f, 0u, 0});
    if (ds->row_tables.size() >= 8) {                       // a handful of geometries at a time
        (void)hipDeviceSynchronize();                       // (nothing in flight may still read the one that goes)
        if (ds->row_tables.front()->dev) (void)hipFree(ds->row_tables.front()->dev);
        ds->row_tables.erase(ds->row_tables.begin());
    }
    auto t = std::make_unique<RowTable>();
    t->height = tg.height; t->pitch = tg.pitch; t->rank = tg.band_rank; t->world = world; t->rows = tg.band_rows; t->compact = tg.compact;
    t->half_h = hh; t->fovI = fi;
    t->il = il; t->il_rows = il_rows; t->il_count = tg.row_count;
    HIP_TRY(hipMalloc(&t->dev, host.size() * sizeof(Entry)));
    HIP_TRY(hipMemcpy(t->dev, host.data(), host.size() * sizeof(Entry), hipMemcpyHostToDevice));
    dev_ptr = t->dev;
    ds->row_tables.push_back(std::move(t));
    return NT_OK;
}

// |o|^2, o.right, o.up, o.forward (see NtCamera)
void camera_dots(int n, const float *origin, const float *axes, float out[4]) {
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < n; ++k) {
        q[0] += (double)origin[k] * origin[k];
        for (int r = 0; r < 3; ++r) q[1 + r] += (double)origin[k] * axes[(size_t)r * n + k];
    }
    for (int r = 0; r < 4; ++r) out[r] = (float)q[r];
}

// camera rows the ray source needs: origin, right, up, forward (camera.hpp:40-45)
void pack_camera(int n, const float *origin, const float *axes, float *out) {
    std::memcpy(out, origin, sizeof(float) * n);
    std::memcpy(out + n, axes, sizeof(float) * n);            // right  = t_orientation[0]
    std::memcpy(out + 2 * n, axes + n, sizeof(float) * n);    // up     = t_orientation[1]
    std::memcpy(out + 3 * n, axes + 2 * n, sizeof(float) * n);// forward= t_orientation[2]
}

void fill_view(NtTarget &tg, const nt_scene *s, int w, int h) {
    // flat_origin_ray_source::set_params (tracer.hpp:65-69)
    tg.width = w;
    tg.height = h;
    tg.half_w = float(w) / float(2);
    tg.half_h = float(h) / float(2);
    tg.fovI = std::tan(s->fov / 2) / tg.half_w;
}

void fill_composite(const nt_scene *s, const DeviceState *ds, NtCompositeDev &c, bool stats) {
    std::memset(&c, 0, sizeof(c));
    c.nodes = (const NtNode *)ds->nodes.p;
    c.items = (const int *)ds->items.p;
    c.batch_recs = (const float *)ds->batch_recs.p;
    c.batch_mats = (const int *)ds->batch_mats.p;
    c.tri_recs = (const float *)ds->tri_recs.p;
    c.tri_mats = (const int *)ds->tri_mats.p;
    c.solid_recs = (const float *)ds->solid_recs.p;
    c.solid_types = (const int *)ds->solid_types.p;
    c.solid_mats = (const int *)ds->solid_mats.p;
    c.materials = (const float *)ds->materials.p;
    c.aabb = (const float *)ds->aabb.p;
    c.rec_stride = s->rec_stride;
    c.root = s->root;
    c.stack_depth = std::max(s->depth + 1, 2);
    c.shadows = s->shadows;
    c.camera_light = s->camera_light;
    c.max_reflect_depth = s->max_reflect_depth;
    c.bg_axis = s->bg_axis;
    for (int k = 0; k < 3; ++k) { c.ambient[k] = s->ambient[k]; c.bg1[k] = s->bg1[k]; c.bg2[k] = s->bg2[k]; c.bg3[k] = s->bg3[k]; }
    const float *lp = (const float *)ds->lights.p;
    c.n_point_lights = (int)(s->pl_color.size() / 3);
    c.n_global_lights = (int)(s->gl_color.size() / 3);
    c.pl_pos = lp;
    c.pl_color = lp + s->pl_pos.size();
    c.gl_dir = c.pl_color + s->pl_color.size();
    c.gl_color = c.gl_dir + s->gl_dir.size();
    c.all_opaque = s->all_opaque;
    c.any_reflective = s->any_reflective;
    c.has_scalar_prims = s->has_scalar;
    c.n_batches = s->n_batches;
    c.n_solids = s->n_solids;
    c.n_triangles = s->n_triangles;
    c.stats = stats ? (unsigned long long *)ds->stats.p : nullptr;
    c.checked = nullptr;
    c.alias_normals = 0;
    c.checked_words = 0;
    c.checked_lanes = 0;
    c.tframes = nullptr;
    c.tframe_count = 0;
}

int check_renderable(const nt_scene *s) {
    if (s->composite) {
        if (s->nodes.size() >= (1u << 24)) return fail(NT_E_UNSUPPORTED, "k-d trees with 2^24 or more nodes are not supported");
    }
    return NT_OK;
}

// the renderer protocol of obj_BlockingRenderer_render (render.cpp:878-904): refuse re-entry, lock the scene
struct RenderGuard {
    nt_scene *s;
    bool held = false;
    explicit RenderGuard(nt_scene *sc) : s(sc) {}
    int acquire() {
        std::lock_guard<std::mutex> g(s->mu);
        if (s->busy) return fail(NT_E_BUSY, "the renderer is already running");
        s->busy = true;
        ++s->locked;
        held = true;
        return NT_OK;
    }
    ~RenderGuard() {
        if (held) {
            std::lock_guard<std::mutex> g(s->mu);
            s->busy = false;
            --s->locked;
        }
    }
};

struct FrameJob {
    const Format *fmt;
    Bands bands;
    void *dest_dev;
    size_t frame_stride;
    int nframes;
    const float *cam_buf;     // device [nframes][4][n] or nullptr
    const float *cam_dots = nullptr;   // with cam_buf: device [nframes][4], the cameras' dot products (camera_dots)
    hipStream_t stream;
    bool stats;
    bool strict = false;      // nt_render_opts.strict_reference
    const int *abort_word = nullptr;   // NtTarget::abort_word
    int overlapped = 0;                // nt_render_opts::overlapped
    bool counters_pass = false;        // (enqueue's own) the statistics launch of a scene whose pixels come from the faithful kernels
    int row_begin, row_count; // owned-row range
    // probe mode
    float *colors_out = nullptr;
    const int *xs = nullptr, *ys = nullptr;
    int probe_count = 0;
    int view_w = 0, view_h = 0;
};

int enqueue(nt_scene *s, DeviceState *ds, const FrameJob &job_in) {
    FrameJob job = job_in;
    if (s->composite && job.stats && !job.counters_pass && !job.colors_out) {
        // Scenes with transparent materials or Solids are drawn by the kernels that reproduce the reference's o_hit.normal
        // handling (below), which keep no counters.  Asking for statistics must not change the pixels: the frame is drawn as
        // always, and the counters come from a launch of their own -- the counting kernel (a hit keeps the normal of what
        // was hit, 16-slot mailbox) into a scratch frame.  They describe THAT traversal: the same tree and cells, a few
        // repeated tests after mailbox evictions.  Transparent materials have no counting kernel at all: refused.
        if (s->n > NT_MAX_FIXED_DIM) return fail(NT_E_UNSUPPORTED, "collect_stats is not available above %d dimensions (the run-time-n kernels keep no counters)", NT_MAX_FIXED_DIM);
        const char *ecl0 = getenv("NTRACER_CLEAN_NORMALS");
        const bool clean0 = ecl0 && atoi(ecl0) != 0;
        if (!s->all_opaque)
            return fail(NT_E_UNSUPPORTED, "collect_stats is not available for scenes with transparent materials (their kernels keep no counters)");
        if (s->n_solids > 0 && !clean0) {
            const size_t bytes = (size_t)job.fmt->pitch * (size_t)(job.bands.compact ? job.bands.owned_rows : job.fmt->height);
            if (int e = ds->stats_frame.ensure(std::max<size_t>(bytes, 16))) return e;
            FrameJob cj = job;
            cj.counters_pass = true;
            cj.dest_dev = ds->stats_frame.p;
            cj.frame_stride = 0;
            if (job.nframes > 1) return fail(NT_E_UNSUPPORTED, "collect_stats on a multi-frame launch is not available for scenes with Solids");
            if (int e = enqueue(s, ds, cj)) return e;
            job.stats = false;
        }
    }
    NtTarget tg;
    std::memset(&tg, 0, sizeof(tg));
    if (job.colors_out) {
        fill_view(tg, s, job.view_w, job.view_h);
        tg.colors_out = job.colors_out;
        tg.probe_xs = job.xs;
        tg.probe_ys = job.ys;
        tg.probe_count = job.probe_count;
        tg.band_world = 1;
        tg.band_rows = NT_RENDER_CHUNK_SIZE;
    } else {
        const Format &f = *job.fmt;
        fill_view(tg, s, f.width, f.height);
        tg.dest = (uint8_t *)job.dest_dev;
        tg.frame_stride = (long long)job.frame_stride;
        const NtChanDev *chans = nullptr;
        if (int r = chan_table(ds, f, chans)) return r;
        tg.chans = chans;
        tg.nchannels = (int)f.chans.size();
        tg.pack_mode = f.pack_mode;
        tg.plain_bits = f.plain_bits;
        tg.plain_maxval = f.plain_maxval;
        for (int k = 0; k < 3; ++k) tg.plain_mul[k] = f.plain_mul[k];
        tg.plain_sel = 0;
        if (f.plain_bits == 8 && ((f.plain_mul[0] | f.plain_mul[1] | f.plain_mul[2]) & ~0x01010101u) == 0) {
            // byte b of the MSB-first pixel word holds the component whose multiplier has bit 8b set; memory byte k is
            // byte 3-k of that word (or byte k when the pixel's bytes are reversed)
            uint32_t sel = 0;
            for (int k = 0; k < 4; ++k) {
                const int b = f.reversed ? k : 3 - k;
                uint32_t pick = 0x0c;                                            // constant 0
                if ((f.plain_mul[0] >> (8 * b)) & 1u) pick = 4;                  // byte 0 of src0: R
                else if (((f.plain_mul[1] | f.plain_mul[2]) >> (8 * b)) & 1u) pick = 0;     // byte 0 of src1: G = B
                sel |= pick << (8 * k);
            }
            tg.plain_sel = sel;
        }
        for (int k = 0; k < 3; ++k) tg.plain_f32[k] = f.plain_f32[k];
        tg.bpp = f.bpp;
        tg.reversed = f.reversed;
        tg.pitch = f.pitch;
        tg.band_rank = job.bands.rank;
        tg.band_world = job.bands.world;
        tg.band_rows = job.bands.rows;
        tg.compact = job.bands.compact;
        tg.row_begin = job.row_begin;
        tg.row_count = job.row_count;
        tg.aligned4 = ((uintptr_t)job.dest_dev % 4 == 0) && (f.pitch % 4 == 0) && (job.frame_stride % 4 == 0);
        tg.abort_word = job.abort_word;
        if (tg.row_count <= 0 || f.bpp == 0) return NT_OK;      // nothing to draw
    }
    NtCamera cam;
    cam.buf = job.cam_buf;
    cam.dots = job.cam_dots;
    cam.n = s->n;
    if (!job.cam_buf) pack_camera(s->n, s->origin.data(), s->axes.data(), cam.inl);
    camera_dots(s->n, s->origin.data(), s->axes.data(), cam.odots);
    NtLaunchInfo li;
    li.n = s->n;
    li.nframes = job.nframes;
    li.stream = job.stream;
    li.persist_counter = nullptr;
    li.persist_cams = nullptr;
    li.cu_count = ds->cu_count;
    li.kernel_choice = 0;
    li.tile_order = nullptr;
    li.hit_buf = nullptr;
    li.hit_frames = 0;
    li.numer_buf = nullptr;
    li.numer_frames = 0;
    li.cull_buf = nullptr;
    li.tie_buf = nullptr;
    li.cull_clean = 0;
    li.tile_rows = 0;
    li.tile_waves = 0;
    li.box_path = 1;
    if (const char *bp = getenv("NTRACER_BOX_PATH")) li.box_path = atoi(bp);
    if (const char *kc = getenv("NTRACER_COMPOSITE_KERNEL")) li.kernel_choice = atoi(kc);
    int r;
    if (s->composite) {
        NtCompositeDev c;
        fill_composite(s, ds, c, job.stats);
        // closest-hit walks drop subtrees beyond the current hit unless the caller (or NTRACER_STRICT_REFERENCE=1)
        // asks for the reference's exact walk; the pixels are the same (nt_beyond_hit in nt_composite.hpp)
        const char *es = getenv("NTRACER_STRICT_REFERENCE");
        const bool env_strict = es && atoi(es) != 0;
        // ... and never for scenes with Solids: trees from the reference's own builder leave solids out of some cells
        // they reach (its goldens show it), i.e. they break the invariant the shortcut relies on
        c.prune = (job.strict || env_strict || s->n_solids > 0) ? 0 : 1;
        if (c.root < 0) c.root = -1;
        // Scenes with transparent materials or Solids are rendered with the reference's own handling of o_hit.normal (its
        // first leaf loop lets every primitive test write to the current hit's normal ray, tracer.hpp:1001,1020 -- see
        // composite_kernel_t<N, true>), which needs the reference's exact `checked` list: a bitmap column per resident
        // lane.  NTRACER_CLEAN_NORMALS=1 selects the intended semantics instead (a hit keeps the normal of what was hit).
        const char *ecl = getenv("NTRACER_CLEAN_NORMALS");
        const bool clean = ecl && atoi(ecl) != 0;
        const bool faithful = !job.counters_pass && (!s->all_opaque || (s->n_solids > 0 && !clean));
        if (faithful) {
            // (transparent materials need the exact list in either mode: the reference trims its transparent hits with the
            // distance of the LAST test, so a repeated test is not harmless there)
            // The compile-time-N kernel keeps NT_TFRAMES ray_color frames in registers/scratch; above NT_MAX_FIXED_DIM, and
            // for reflection deeper than that among transparent things, the run-time-n kernel with its frames in global
            // scratch takes over (one wave per block there).
            const char *efv = getenv("NTRACER_FORCE_VAR");
            const int nframes_stack = s->any_reflective ? s->max_reflect_depth + 1 : 1;
            const bool var_t = s->n > NT_MAX_FIXED_DIM || (efv && atoi(efv) != 0) || nframes_stack > 6;
            const long long lpb = var_t ? 64 : 256;           // lanes per block
            const long long tw = var_t ? 8 : 16;              // tile edge
            const long long words = ((long long)s->n_batches + s->n_triangles + s->n_solids + 31) / 32;
            const long long fwords = var_t ? (long long)nt_var_frame_words(s->n) * nframes_stack : 0;
            long long tiles = job.colors_out ? (job.probe_count + lpb - 1) / lpb
                                             : (long long)((tg.width + tw - 1) / tw) * ((tg.row_count + tw - 1) / tw) * job.nframes;
            long long blocks = std::min<long long>(std::max<long long>(tiles, 1), var_t ? 8192 : 4096);
            while (blocks > 64 && blocks * lpb * (words + fwords) * 4 > ((long long)512 << 20)) blocks /= 2;
            if (int e = ds->checked.ensure((size_t)(blocks * lpb * words * 4))) return e;
            c.checked = (uint32_t *)ds->checked.p;
            c.checked_words = (int)words;
            c.checked_lanes = (int)(blocks * lpb);
            c.alias_normals = clean ? 0 : 1;
            if (var_t) {
                if (int e = ds->tframes.ensure((size_t)(blocks * lpb * fwords * 4))) return e;
                c.tframes = (float *)ds->tframes.p;
                c.tframe_count = nframes_stack;
            }
        }
        // image renders of opaque scenes made of batches go through the packet kernel (primary rays share the
        // camera origin): it needs the camera table in device memory (and, for the persistent variant, a counter)
        const bool packetable = c.all_opaque != 0 && !faithful;
        if (packetable && !job.stats && !job.colors_out && s->n <= NT_MAX_FIXED_DIM && li.kernel_choice != 2) {
            // persistent kernel: a zeroed work counter and the camera table in device memory (stream ordered)
            if (int e = ds->counter.ensure(8)) return e;
            HIP_TRY(hipMemsetAsync(ds->counter.p, 0, 8, job.stream));
            li.persist_counter = ds->counter.p;
            if (job.cam_buf) {
                li.persist_cams = job.cam_buf;
            } else {
                if (int e = ds->cams.ensure(sizeof(float) * 4 * s->n)) return e;
                HIP_TRY(hipMemcpyAsync(ds->cams.p, cam.inl, sizeof(float) * 4 * s->n, hipMemcpyHostToDevice, job.stream));
                li.persist_cams = (const float *)ds->cams.p;
            }
        }
        const char *enum_ = getenv("NTRACER_NUMERATORS");
        if (li.persist_cams && !tg.colors_out && s->n_batches > 0 && !(enum_ && atoi(enum_) == 0)) {
            // plane numerators of the packet kernel: as many frames as fit in 256 MB, at least one
            const size_t per_frame = (size_t)s->n_batches * NT_BATCH_SIZE * sizeof(float);
            const size_t frames = std::max<size_t>(1, std::min<size_t>((size_t)job.nframes, ((size_t)256 << 20) / per_frame));
            if (int e = ds->numer.ensure(frames * per_frame)) return e;
            li.numer_buf = (float *)ds->numer.p;
            li.numer_frames = (int)frames;
            if (const char *cf = getenv("NTRACER_CHUNK_FRAMES")) li.numer_frames = std::max(1, std::min(li.numer_frames, atoi(cf)));   // tests
        }
        const bool lit = !s->pl_color.empty() || !s->gl_color.empty() || c.any_reflective || c.has_scalar_prims;
        const char *e2p = getenv("NTRACER_TWO_PASS");
        if (li.persist_cams && !tg.colors_out && lit && !(e2p && atoi(e2p) == 0)) {
            // scratch for the primary hits of a two-pass render: as many frames as fit in 512 MB, at least one
            const size_t per_frame = (size_t)16 * tg.width * tg.row_count;
            size_t frames = std::max<size_t>(1, std::min<size_t>((size_t)job.nframes, ((size_t)512 << 20) / std::max<size_t>(per_frame, 1)));
            if (int e = ds->hits.ensure(frames * per_frame)) return e;
            li.hit_buf = ds->hits.p;
            li.hit_frames = (int)frames;
            if (const char *cf = getenv("NTRACER_CHUNK_FRAMES")) li.hit_frames = std::max(1, std::min(li.hit_frames, atoi(cf)));
        }
        const char *eto = getenv("NTRACER_TILE_ORDER");
        if (li.persist_cams && !tg.colors_out && !(eto && atoi(eto) == 0)) {
            // dispatch order of the packet kernel's quads (2x2 tiles of 8x8 pixels): the waves that walk the middle
            // of the scene run longest, so the quad rows nearest the centre go first; row-major within a row keeps
            // neighbouring blocks on neighbouring rays
            const int tx = ((tg.width + 7) / 8 + 1) / 2, ty = ((tg.row_count + 7) / 8 + 1) / 2;
            DeviceState::TileOrder *to = nullptr;
            for (auto &e : ds->tile_orders)
                if (e->tx == tx && e->ty == ty) to = e.get();
            if (!to) {
                std::vector<int> order((size_t)tx * ty);
                for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
                auto key = [&](int t) {
                    const long long dy = 2 * (t / tx) - (ty - 1);
                    return dy < 0 ? -dy : dy;
                };
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
                if (ds->tile_orders.size() >= 16) {
                    // tables may still be read by launches queued on other streams
                    HIP_TRY(hipDeviceSynchronize());
                    for (auto &e : ds->tile_orders) e->buf.release();
                    ds->tile_orders.clear();
                }
                std::unique_ptr<DeviceState::TileOrder> e(new DeviceState::TileOrder);
                if (int err = e->buf.ensure(order.size() * sizeof(int))) return err;
                HIP_TRY(hipMemcpy(e->buf.p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
                e->tx = tx;
                e->ty = ty;
                to = e.get();
                ds->tile_orders.push_back(std::move(e));
            }
            li.tile_order = (const int *)to->buf.p;
        }
        r = nt_launch_composite(li, cam, c, tg);
    } else {
        const char *ec = getenv("NTRACER_BOX_CULL");
        if (!tg.colors_out && !(ec && atoi(ec) == 0)) {
            // one bit per 64-pixel stretch of a row: can any of its rays reach the cube? (box_cull_kernel)
            const size_t words = (size_t)(((tg.width + 63) / 64 + 31) / 32);
            // stretch codes (4 words per redo word), 16 rows of padding (box_kernel reads a wave's rows without
            // clamping), redo bits
            const size_t need = ((size_t)5 * job.nframes * tg.row_count + 64) * words * sizeof(uint32_t);
            if (need > ds->cull.cap || !ds->cull.p) ds->cull_clean = false;
            if (int e = ds->cull.ensure(need)) return e;
            // The fused kernels keep their redo bitmap at the start of this buffer and leave it zeroed; after anything else
            // has written there (a fresh allocation, the cull / box / redo kernels) it is zeroed here, in stream order.
            // (the formats launch_box_fixed sends there: plain RGB of <= 10 bits in one aligned dword, or three plain fp32 channels)
            const bool fused = li.box_path != 0 && s->n <= NT_MAX_FIXED_BOX_DIM && tg.aligned4 &&
                               ((tg.plain_bits != 0u && tg.plain_bits <= 10u && tg.bpp == 4) || (tg.plain_f32[0] >= 0 && tg.bpp == 12));
            if (fused && !ds->cull_clean) {
                HIP_TRY(hipMemsetAsync(ds->cull.p, 0, ds->cull.cap, job.stream));
                ds->cull_clean = true;
            } else if (!fused) {
                ds->cull_clean = false;
            }
            li.cull_clean = fused ? 1 : 0;
            li.cull_buf = (uint32_t *)ds->cull.p;
            if (fused) {
                // interleaved rows (the default for launches that start at their first owned row; NTRACER_BOX_INTERLEAVE=0: A/B):
                // the waves of a column strip deal the rows out among themselves, so that the rows that need ray-by-ray work --
                // which come in runs of dozens -- are spread over all of them instead of making a few waves ten times as long as
                // the rest
                const NtBoxTileGeom geom = nt_box_tile_geom(tg.width, tg.row_count, job.nframes, job.overlapped);
                li.tile_rows = geom.rows;
                li.tile_waves = geom.waves;
                const char *eil = getenv("NTRACER_BOX_INTERLEAVE");
                const int tile_rows = geom.rows * geom.waves;
                tg.row_il = (tg.row_begin == 0 && !(eil && atoi(eil) == 0)) ? (tg.row_count + tile_rows - 1) / tile_rows * geom.waves : 0;
                if (int e = row_table(ds, tg, geom.rows, tg.rowtab)) return e;
                li.tie_buf = nullptr;               // (round 2's end: the tie sets never leave the tile kernel)
            }
        }
        r = nt_launch_box(li, cam, tg);
    }
    if (r) return fail(r == -2 ? NT_E_UNSUPPORTED : NT_E_DEVICE, "%s", nt_launch_error());
    return NT_OK;
}

int prepare_stats(DeviceState *ds, hipStream_t st, bool on) {
    if (!on) return NT_OK;
    if (int r = ds->stats.ensure(8 * sizeof(unsigned long long))) return r;
    HIP_TRY(hipMemsetAsync(ds->stats.p, 0, 8 * sizeof(unsigned long long), st));
    return NT_OK;
}

int validate_desc(const nt_scene_desc *d) {
    if (!d) return fail(NT_E_INVALID, "scene description is NULL");
    const int n = d->dimension;
    if (n < 3 || n > NT_MAX_DIM) return fail(NT_E_INVALID, "dimension must be between 3 and %d", NT_MAX_DIM);
    if (d->n_nodes < 0 || d->n_items < 0 || d->n_batches < 0 || d->n_triangles < 0 || d->n_solids < 0 || d->n_materials < 0)
        return fail(NT_E_INVALID, "negative count in scene description");
    if (d->root < -1 || d->root >= d->n_nodes) return fail(NT_E_INVALID, "root index out of range");
    if (!d->aabb_start || !d->aabb_end) return fail(NT_E_INVALID, "scene boundary is required");
    if (d->n_nodes && (!d->node_axis || !d->node_split || !d->node_left || !d->node_right)) return fail(NT_E_INVALID, "node arrays are NULL");
    if (d->n_items && !d->items) return fail(NT_E_INVALID, "items array is NULL");
    if (d->n_batches && (!d->batch_recs || !d->batch_mats)) return fail(NT_E_INVALID, "batch arrays are NULL");
    if (d->n_triangles && (!d->tri_recs || !d->tri_mats)) return fail(NT_E_INVALID, "triangle arrays are NULL");
    if (d->n_solids && (!d->solid_recs || !d->solid_types || !d->solid_mats)) return fail(NT_E_INVALID, "solid arrays are NULL");
    if ((d->n_batches || d->n_triangles || d->n_solids) && (!d->materials || d->n_materials < 1)) return fail(NT_E_INVALID, "materials are required");
    for (int i = 0; i < d->n_nodes; ++i) {
        const int ax = d->node_axis[i];
        if (ax >= n || ax < -1) return fail(NT_E_INVALID, "node %d: axis %d out of range", i, ax);
        if (ax < 0) {
            const long st = d->node_left[i], cnt = d->node_right[i];
            if (st < 0 || cnt < 1 || st + cnt > d->n_items) return fail(NT_E_INVALID, "leaf %d: item range out of bounds", i);
        } else {
            const int l = d->node_left[i], r = d->node_right[i];
            if (l < -1 || l >= d->n_nodes || r < -1 || r >= d->n_nodes) return fail(NT_E_INVALID, "branch %d: child index out of range", i);
            if (l < 0 && r < 0) return fail(NT_E_INVALID, "branch %d: both children are empty", i);
        }
    }
    for (int i = 0; i < d->n_items; ++i) {
        const int it = d->items[i];
        const int kind = it & 3, idx = it >> 2;
        const int lim = kind == NT_KIND_BATCH ? d->n_batches : kind == NT_KIND_TRIANGLE ? d->n_triangles : kind == NT_KIND_SOLID ? d->n_solids : -1;
        if (it < 0 || idx >= lim) return fail(NT_E_INVALID, "item %d: primitive reference out of range", i);
    }
    auto check_mats = [&](const int32_t *m, long cnt) {
        for (long i = 0; i < cnt; ++i) if (m[i] < 0 || m[i] >= d->n_materials) return false;
        return true;
    };
    if (!check_mats(d->batch_mats, (long)d->n_batches * NT_BATCH_SIZE) || !check_mats(d->tri_mats, d->n_triangles) || !check_mats(d->solid_mats, d->n_solids))
        return fail(NT_E_INVALID, "material index out of range");
    for (int i = 0; i < d->n_solids; ++i)
        if (d->solid_types[i] != NT_SOLID_CUBE && d->solid_types[i] != NT_SOLID_SPHERE) return fail(NT_E_INVALID, "solid %d: unknown type", i);
    return NT_OK;
}

// depth of the tree + cycle check (every node reachable at most once)
int tree_depth(const nt_scene_desc *d, int &depth_out) {
    depth_out = 0;
    if (d->root < 0) return NT_OK;
    std::vector<char> seen((size_t)d->n_nodes, 0);
    std::vector<std::pair<int, int>> stack;
    stack.emplace_back(d->root, 1);
    while (!stack.empty()) {
        auto [node, dep] = stack.back();
        stack.pop_back();
        if (seen[node]) return fail(NT_E_INVALID, "node %d is referenced more than once (the k-d tree must be a tree)", node);
        seen[node] = 1;
        depth_out = std::max(depth_out, dep);
        if (d->node_axis[node] >= 0) {
            if (d->node_left[node] >= 0) stack.emplace_back(d->node_left[node], dep + 1);
            if (d->node_right[node] >= 0) stack.emplace_back(d->node_right[node], dep + 1);
        }
    }
    return NT_OK;
}

void pad_records(const float *src, long count, int rec_len, int stride, std::vector<float> &out) {
    out.assign((size_t)count * stride, 0.0f);
    for (long i = 0; i < count; ++i) std::memcpy(out.data() + (size_t)i * stride, src + (size_t)i * rec_len, sizeof(float) * rec_len);
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *nt_version(void) { return "ntracer_hip 0.1 (gfx950)"; }

const char *nt_last_error(void) { return g_error.c_str(); }

int nt_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

nt_scene_t *nt_box_scene_create(int dimension) {
    if (dimension < 3 || dimension > NT_MAX_DIM) {
        fail(NT_E_INVALID, "dimension must be between 3 and %d", NT_MAX_DIM);
        return nullptr;
    }
    nt_scene *s = new (std::nothrow) nt_scene();
    if (!s) { fail(NT_E_NOMEM, "out of memory"); return nullptr; }
    s->composite = false;
    s->n = dimension;
    s->origin.assign(dimension, 0.0f);                 // camera(d): origin 0, identity axes (camera.hpp:11)
    s->axes.assign((size_t)dimension * dimension, 0.0f);
    for (int i = 0; i < dimension; ++i) s->axes[(size_t)i * dimension + i] = 1.0f;
    return s;
}

nt_scene_t *nt_composite_scene_create(const nt_scene_desc *d) {
    if (validate_desc(d)) return nullptr;
    int depth = 0;
    if (tree_depth(d, depth)) return nullptr;
    nt_scene *s = new (std::nothrow) nt_scene();
    if (!s) { fail(NT_E_NOMEM, "out of memory"); return nullptr; }
    const int n = d->dimension;
    s->composite = true;
    s->n = n;
    s->origin.assign(n, 0.0f);
    s->axes.assign((size_t)n * n, 0.0f);
    for (int i = 0; i < n; ++i) s->axes[(size_t)i * n + i] = 1.0f;
    s->root = d->root;
    s->depth = depth;
    s->nodes.resize((size_t)d->n_nodes);
    for (int i = 0; i < d->n_nodes; ++i) {
        s->nodes[i].split = d->node_split[i];
        s->nodes[i].axis = d->node_axis[i];
        s->nodes[i].left = d->node_left[i];
        s->nodes[i].right = d->node_right[i];
    }
    s->items.assign(d->items, d->items + d->n_items);
    s->rec_len = n * n + n + 1;
    s->rec_stride = (s->rec_len + 3) / 4 * 4;
    s->n_batches = d->n_batches;
    s->n_triangles = d->n_triangles;
    s->n_solids = d->n_solids;
    s->n_materials = d->n_materials;
    pad_records(d->batch_recs, (long)d->n_batches * NT_BATCH_SIZE, s->rec_len, s->rec_stride, s->batch_recs);
    pad_records(d->tri_recs, d->n_triangles, s->rec_len, s->rec_stride, s->tri_recs);
    s->batch_mats.assign(d->batch_mats, d->batch_mats + (size_t)d->n_batches * NT_BATCH_SIZE);
    s->tri_mats.assign(d->tri_mats, d->tri_mats + d->n_triangles);
    s->solid_recs.assign(d->solid_recs, d->solid_recs + (size_t)d->n_solids * (2 * n * n + n));
    s->solid_types.assign(d->solid_types, d->solid_types + d->n_solids);
    s->solid_mats.assign(d->solid_mats, d->solid_mats + d->n_solids);
    s->materials.resize((size_t)d->n_materials * 10);
    for (int i = 0; i < d->n_materials; ++i) {
        const nt_material &m = d->materials[i];
        float *o = s->materials.data() + (size_t)i * 10;
        o[0] = m.color[0]; o[1] = m.color[1]; o[2] = m.color[2];
        o[3] = m.specular[0]; o[4] = m.specular[1]; o[5] = m.specular[2];
        o[6] = m.opacity; o[7] = m.reflectivity; o[8] = m.specular_intensity; o[9] = m.specular_exp;
        if (!(m.opacity >= 1.0f)) s->all_opaque = false;       // primitive::opaque (tracer.hpp:187)
        if (m.reflectivity != 0.0f) s->any_reflective = true;
    }
    s->has_scalar = false;
    for (int i = 0; i < d->n_items; ++i) if ((d->items[i] & 3) != NT_KIND_BATCH) s->has_scalar = true;
    s->aabb.assign(d->aabb_start, d->aabb_start + n);
    s->aabb.insert(s->aabb.end(), d->aabb_end, d->aabb_end + n);
    return s;
}

void nt_scene_destroy(nt_scene_t *s) {
    if (!s) return;
    for (auto &kv : s->devs) {
        DeviceState *ds = kv.second.get();
        if (hipSetDevice(ds->device) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        for (DevBuf *b : {&ds->nodes, &ds->items, &ds->batch_recs, &ds->batch_mats, &ds->tri_recs, &ds->tri_mats, &ds->solid_recs,
                          &ds->solid_types, &ds->solid_mats, &ds->materials, &ds->aabb, &ds->lights, &ds->framebuffer, &ds->cams, &ds->counter,
                          &ds->probes, &ds->stats, &ds->hits, &ds->stats_frame, &ds->numer, &ds->cull, &ds->checked, &ds->tframes, &ds->ties})
            b->release();
        for (auto &t : ds->chan_tables) if (t->dev) (void)hipFree(t->dev);
        for (auto &t : ds->row_tables) if (t->dev) (void)hipFree(t->dev);
        for (auto &t : ds->tile_orders) t->buf.release();
        for (auto &st : ds->stage) {
            if (st.host) (void)hipHostFree(st.host);
            if (st.done) (void)hipEventDestroy(st.done);
        }
        if (ds->stream) (void)hipStreamDestroy(ds->stream);
        ds->abort_word.release();
        if (ds->abort_one) (void)hipHostFree(ds->abort_one);
        if (ds->side_stream) (void)hipStreamDestroy(ds->side_stream);
        if (ds->frame_done) (void)hipEventDestroy(ds->frame_done);
    }
    delete s;
}

int nt_scene_dimension(const nt_scene_t *s) { return s ? s->n : fail(NT_E_INVALID, "scene is NULL"); }
int nt_scene_is_composite(const nt_scene_t *s) { return s ? (s->composite ? 1 : 0) : fail(NT_E_INVALID, "scene is NULL"); }

int nt_scene_set_camera(nt_scene_t *s, const float *origin, const float *axes) {
    if (!s || !origin || !axes) return fail(NT_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> g(s->mu);
    if (s->locked) return fail(NT_E_LOCKED, "the scene is locked for reading");
    s->origin.assign(origin, origin + s->n);
    s->axes.assign(axes, axes + (size_t)s->n * s->n);
    return NT_OK;
}

int nt_scene_get_camera(const nt_scene_t *s, float *origin, float *axes) {
    if (!s || !origin || !axes) return fail(NT_E_INVALID, "NULL argument");
    std::memcpy(origin, s->origin.data(), sizeof(float) * s->n);
    std::memcpy(axes, s->axes.data(), sizeof(float) * s->n * s->n);
    return NT_OK;
}

int nt_scene_set_fov(nt_scene_t *s, float fov) {
    if (!s) return fail(NT_E_INVALID, "scene is NULL");
    std::lock_guard<std::mutex> g(s->mu);
    if (s->locked) return fail(NT_E_LOCKED, "the scene is locked for reading");
    s->fov = fov;
    return NT_OK;
}

float nt_scene_get_fov(const nt_scene_t *s) { return s ? s->fov : 0.0f; }

int nt_scene_set_params(nt_scene_t *s, const nt_scene_params *p) {
    if (!s || !p) return fail(NT_E_INVALID, "NULL argument");
    if (!s->composite) return fail(NT_E_INVALID, "BoxScene has no lighting parameters");
    if (p->bg_gradient_axis < 0 || p->bg_gradient_axis >= s->n) return fail(NT_E_INVALID, "bg_gradient_axis out of range");
    if (p->n_point_lights < 0 || p->n_global_lights < 0) return fail(NT_E_INVALID, "negative light count");
    if ((p->n_point_lights && (!p->point_light_pos || !p->point_light_color)) || (p->n_global_lights && (!p->global_light_dir || !p->global_light_color)))
        return fail(NT_E_INVALID, "light arrays are NULL");
    std::lock_guard<std::mutex> g(s->mu);
    if (s->locked) return fail(NT_E_LOCKED, "the scene is locked for reading");
    s->shadows = p->shadows ? 1 : 0;
    s->camera_light = p->camera_light ? 1 : 0;
    s->max_reflect_depth = p->max_reflect_depth;
    s->bg_axis = p->bg_gradient_axis;
    for (int k = 0; k < 3; ++k) { s->ambient[k] = p->ambient[k]; s->bg1[k] = p->bg1[k]; s->bg2[k] = p->bg2[k]; s->bg3[k] = p->bg3[k]; }
    const int n = s->n;
    s->pl_pos.assign(p->point_light_pos, p->point_light_pos + (size_t)p->n_point_lights * n);
    s->pl_color.assign(p->point_light_color, p->point_light_color + (size_t)p->n_point_lights * 3);
    s->gl_dir.assign(p->global_light_dir, p->global_light_dir + (size_t)p->n_global_lights * n);
    s->gl_color.assign(p->global_light_color, p->global_light_color + (size_t)p->n_global_lights * 3);
    ++s->lights_version;
    return NT_OK;
}

int nt_scene_lock(nt_scene_t *s) {
    if (!s) return fail(NT_E_INVALID, "scene is NULL");
    std::lock_guard<std::mutex> g(s->mu);
    ++s->locked;
    return NT_OK;
}

int nt_scene_unlock(nt_scene_t *s) {
    if (!s) return fail(NT_E_INVALID, "scene is NULL");
    std::lock_guard<std::mutex> g(s->mu);
    if (s->locked <= 0) return fail(NT_E_INVALID, "the scene is not locked");
    --s->locked;
    return NT_OK;
}

int nt_scene_locked(const nt_scene_t *s) { return s ? (s->locked > 0 ? 1 : 0) : fail(NT_E_INVALID, "scene is NULL"); }

int nt_format_bytes_per_pixel(const nt_image_format *fmt) {
    Format f;
    if (int r = parse_format(fmt, f)) return r;
    return f.bpp;
}

int nt_render(nt_scene_t *s, void *dest, size_t dest_len, const nt_image_format *fmt, const nt_render_opts *opts, volatile int *abort_flag) {
    if (!s || !dest) return fail(NT_E_INVALID, "NULL argument");
    Format f;
    if (int r = parse_format(fmt, f)) return r;
    Bands b;
    if (int r = parse_bands(opts, f.height, b)) return r;
    const size_t need = required_len(f, b);
    if (dest_len < need) return fail(NT_E_INVALID, "the buffer is too small for an image with the given dimensions");
    if (int r = check_renderable(s)) return r;
    RenderGuard guard(s);
    if (int r = guard.acquire()) return r;
    if (abort_flag && *abort_flag) return NT_ABORTED;           // (before anything touches `dest` or the device)
    int dev;
    if (int r = pick_device(opts, -1, dev)) return r;
    DeviceState *ds;
    if (int r = device_state(s, dev, ds)) return r;
    if (int r = upload_scene(s, ds)) return r;
    if (int r = own_stream(ds)) return r;
    if (int r = use_stream(ds, ds->stream)) return r;
    if (int r = ds->framebuffer.ensure(std::max<size_t>(need, 16))) return r;
    const bool stats = opts && opts->collect_stats;
    if (int r = prepare_stats(ds, ds->stream, stats)) return r;
    // pitch padding bytes are not written by the kernels: carry the caller's bytes through
    if (f.pitch != f.width * f.bpp || (b.world > 1 && !b.compact)) HIP_TRY(hipMemcpyAsync(ds->framebuffer.p, dest, need, hipMemcpyHostToDevice, ds->stream));

    FrameJob job{};
    job.fmt = &f;
    job.bands = b;
    job.dest_dev = ds->framebuffer.p;
    job.frame_stride = 0;
    job.nframes = 1;
    job.cam_buf = nullptr;
    job.stream = ds->stream;
    job.stats = stats;
    job.strict = opts && opts->strict_reference;
    // Abort (the reference's workers poll renderer::CANCEL per pixel, render.cpp:412): ONE launch for the frame -- cutting it
    // into slabs cost a 120-cell frame a kernel tail per slab (9.3 ms instead of 1.2) -- whose blocks read a dword in device
    // memory when they start, and the packet kernel's waves every few dozen nodes (NtTarget::abort_word).  The host waits for the
    // frame with an eye on the caller's flag and raises that word when the flag goes up: what has not started leaves at once,
    // so an abort costs what the waves in flight need to reach their next look at the word.  An aborted frame is incomplete
    // in no particular order; nothing of it is copied back -- the caller's buffer stays as it was.
    if (abort_flag) {
        if (!ds->abort_one) {
            if (int r = ds->abort_word.ensure(64)) return r;
            void *p = nullptr;
            HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocDefault));
            ds->abort_one = (int *)p;
            *ds->abort_one = 1;
            HIP_TRY(hipStreamCreateWithFlags(&ds->side_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ds->frame_done, hipEventDisableTiming));
        }
        HIP_TRY(hipMemsetAsync(ds->abort_word.p, 0, 4, ds->stream));
        job.abort_word = (const int *)ds->abort_word.p;
    }
    job.row_begin = 0;
    job.row_count = b.owned_rows;
    if (int r = enqueue(s, ds, job)) { (void)hipStreamSynchronize(ds->stream); return r; }
    bool aborted = false;
    if (abort_flag) {
        HIP_TRY(hipEventRecord(ds->frame_done, ds->stream));
        while (hipEventQuery(ds->frame_done) == hipErrorNotReady) {
            if (!aborted && *abort_flag) {
                (void)hipMemcpyAsync(ds->abort_word.p, ds->abort_one, 4, hipMemcpyHostToDevice, ds->side_stream);
                aborted = true;
            }
            std::this_thread::yield();
        }
    }
    if (!aborted && need) HIP_TRY(hipMemcpyAsync(dest, ds->framebuffer.p, need, hipMemcpyDeviceToHost, ds->stream));
    HIP_TRY(hipStreamSynchronize(ds->stream));
    if (aborted) HIP_TRY(hipStreamSynchronize(ds->side_stream));
    if (stats) {
        unsigned long long v[8];
        HIP_TRY(hipMemcpy(v, ds->stats.p, sizeof(v), hipMemcpyDeviceToHost));
        s->last_stats.rays = v[0]; s->last_stats.shadow_rays = v[1]; s->last_stats.branches = v[2]; s->last_stats.leaves = v[3];
        s->last_stats.simplex_tests = v[4]; s->last_stats.solid_tests = v[5]; s->last_stats.hits = v[6]; s->last_stats.aabb_enter = v[7];
        s->have_stats = true;
    }
    return aborted ? NT_ABORTED : NT_OK;
}

int nt_render_device(nt_scene_t *s, void *dest_dev, size_t dest_len, const nt_image_format *fmt, const nt_render_opts *opts, void *hip_stream) {
    if (!s || !dest_dev) return fail(NT_E_INVALID, "NULL argument");
    Format f;
    if (int r = parse_format(fmt, f)) return r;
    Bands b;
    if (int r = parse_bands(opts, f.height, b)) return r;
    if (dest_len < required_len(f, b)) return fail(NT_E_INVALID, "the buffer is too small for an image with the given dimensions");
    if (int r = check_renderable(s)) return r;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->busy) return fail(NT_E_BUSY, "the renderer is already running");
    int dev;
    if (int r = pick_device(opts, -1, dev)) return r;
    DeviceState *ds;
    if (int r = device_state(s, dev, ds)) return r;
    if (int r = upload_scene(s, ds)) return r;
    if (int r = use_stream(ds, (hipStream_t)hip_stream)) return r;
    const bool stats = opts && opts->collect_stats;
    if (int r = prepare_stats(ds, (hipStream_t)hip_stream, stats)) return r;
    if (stats) { s->have_stats = false; s->stats_device = dev; }
    FrameJob job{};
    job.fmt = &f;
    job.bands = b;
    job.dest_dev = dest_dev;
    job.nframes = 1;
    job.stream = (hipStream_t)hip_stream;
    job.stats = stats;
    job.strict = opts && opts->strict_reference;
    job.abort_word = opts ? (const int *)opts->abort_device : nullptr;
    job.overlapped = opts ? opts->overlapped : 0;
    job.row_begin = 0;
    job.row_count = b.owned_rows;
    return enqueue(s, ds, job);
}

int nt_render_frames_device(nt_scene_t *s, void *dest_dev, size_t frame_stride, int nframes, const float *origins, const float *axes,
                            const nt_image_format *fmt, const nt_render_opts *opts, void *hip_stream) {
    if (!s || !dest_dev || !origins || !axes) return fail(NT_E_INVALID, "NULL argument");
    if (nframes < 1 || nframes > 65535) return fail(NT_E_INVALID, "nframes must be between 1 and 65535");
    Format f;
    if (int r = parse_format(fmt, f)) return r;
    Bands b;
    if (int r = parse_bands(opts, f.height, b)) return r;
    if (frame_stride < required_len(f, b)) return fail(NT_E_INVALID, "frame_stride is smaller than one frame");
    if (int r = check_renderable(s)) return r;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->busy) return fail(NT_E_BUSY, "the renderer is already running");
    int dev;
    if (int r = pick_device(opts, -1, dev)) return r;
    DeviceState *ds;
    if (int r = device_state(s, dev, ds)) return r;
    if (int r = upload_scene(s, ds)) return r;
    {
        // This entry point stages the caller's host arrays in a pinned slot that later calls reuse: a graph would replay the
        // launch, not the staging.  Refused while `hip_stream` is being captured -- a camera table (nt_render_table_device), whose
        // cameras live in device memory, is what a graph wants (tests: test_render_calls_captured_in_a_hip_graph).
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)hip_stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(NT_E_UNSUPPORTED, "nt_render_frames_device stages host cameras per call and cannot be captured into a graph: use a camera table");
    }
    if (int r = use_stream(ds, (hipStream_t)hip_stream)) return r;
    const int n = s->n;
    const size_t cam_floats = (size_t)nframes * 4 * n + (size_t)nframes * 4;
    DeviceState::Stage *st = nullptr;
    if (int r = stage_slot(ds, cam_floats * sizeof(float), st)) return r;
    float *packed = (float *)st->host;
    for (int fidx = 0; fidx < nframes; ++fidx) {
        pack_camera(n, origins + (size_t)fidx * n, axes + (size_t)fidx * n * n, packed + (size_t)fidx * 4 * n);
        camera_dots(n, origins + (size_t)fidx * n, axes + (size_t)fidx * n * n, packed + (size_t)nframes * 4 * n + (size_t)fidx * 4);
    }
    // a camera table that earlier launches may still read must not be overwritten: grow-only buffer,
    // refilled only after the stream that used it has drained (same-stream ordering)
    if (int r = ds->cams.ensure(cam_floats * sizeof(float))) return r;
    {
        // NTRACER_CAM_UPLOAD: "kernel" (default) a copy kernel on the launch stream reading the pinned slot in place;
        // "memcpy" hipMemcpyAsync (copy engine)
        const char *eu = getenv("NTRACER_CAM_UPLOAD");
        if (eu && eu[0] == 'm') {
            HIP_TRY(hipMemcpyAsync(ds->cams.p, packed, cam_floats * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)hip_stream));
        } else if (nt_launch_upload(hip_stream, packed, (float *)ds->cams.p, (int)cam_floats)) {
            return fail(NT_E_DEVICE, "%s", nt_launch_error());
        }
    }
    {
        const char *ese = getenv("NTRACER_STAGE_EVENT");     // (experiment: 0 = no event behind the upload; unsafe beyond 8 calls in flight)
        if (!(ese && atoi(ese) == 0)) {
            HIP_TRY(hipEventRecord(st->done, (hipStream_t)hip_stream));
            st->in_flight = true;
        }
    }
    const bool stats = opts && opts->collect_stats;
    if (int r = prepare_stats(ds, (hipStream_t)hip_stream, stats)) return r;
    if (stats) { s->have_stats = false; s->stats_device = dev; }
    FrameJob job{};
    job.fmt = &f;
    job.bands = b;
    job.dest_dev = dest_dev;
    job.frame_stride = frame_stride;
    job.nframes = nframes;
    job.cam_buf = (const float *)ds->cams.p;
    job.cam_dots = job.cam_buf + (size_t)nframes * 4 * s->n;
    job.stream = (hipStream_t)hip_stream;
    job.stats = stats;
    job.strict = opts && opts->strict_reference;
    job.abort_word = opts ? (const int *)opts->abort_device : nullptr;
    job.overlapped = opts ? opts->overlapped : 0;
    job.row_begin = 0;
    job.row_count = b.owned_rows;
    return enqueue(s, ds, job);
}

struct nt_camera_table {
    int n = 0, nframes = 0, device = -1;
    float *dev = nullptr;                // [nframes][4][n] camera rows, then [nframes][4] dot products (NtCamera::buf)
};

nt_camera_table_t *nt_camera_table_create(int dimension, int nframes, const float *origins, const float *axes, int device) {
    if (dimension < 3 || dimension > NT_MAX_DIM || nframes < 1 || nframes > 65535 || !origins || !axes) {
        fail(NT_E_INVALID, "invalid camera table arguments");
        return nullptr;
    }
    int dev;
    if (pick_device(nullptr, device, dev)) return nullptr;
    const int n = dimension;
    const size_t cam_floats = (size_t)nframes * 4 * n + (size_t)nframes * 4;
    std::vector<float> packed(cam_floats);
    for (int f = 0; f < nframes; ++f) {
        pack_camera(n, origins + (size_t)f * n, axes + (size_t)f * n * n, packed.data() + (size_t)f * 4 * n);
        camera_dots(n, origins + (size_t)f * n, axes + (size_t)f * n * n, packed.data() + (size_t)nframes * 4 * n + (size_t)f * 4);
    }
    void *p = nullptr;
    if (hipMalloc(&p, cam_floats * sizeof(float)) != hipSuccess) { fail(NT_E_NOMEM, "hipMalloc failed for the camera table"); return nullptr; }
    if (hipMemcpy(p, packed.data(), cam_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p);
        fail(NT_E_DEVICE, "camera table upload failed");
        return nullptr;
    }
    nt_camera_table *t = new (std::nothrow) nt_camera_table();
    if (!t) { (void)hipFree(p); fail(NT_E_NOMEM, "out of memory"); return nullptr; }
    t->n = n; t->nframes = nframes; t->device = dev; t->dev = (float *)p;
    return t;
}

void nt_camera_table_destroy(nt_camera_table_t *t) {
    if (!t) return;
    if (t->dev && hipSetDevice(t->device) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(t->dev); }
    delete t;
}

int nt_camera_table_frames(const nt_camera_table_t *t) { return t ? t->nframes : fail(NT_E_INVALID, "table is NULL"); }

int nt_render_table_device(nt_scene_t *s, void *dest_dev, size_t frame_stride, const nt_camera_table_t *table, int first, int count,
                           const nt_image_format *fmt, const nt_render_opts *opts, void *hip_stream) {
    if (!s || !dest_dev || !table) return fail(NT_E_INVALID, "NULL argument");
    if (table->n != s->n) return fail(NT_E_INVALID, "the camera table is for %d dimensions, the scene has %d", table->n, s->n);
    if (first < 0 || count < 1 || first > table->nframes - count) return fail(NT_E_INVALID, "frames %d..%d are not in a table of %d", first, first + count - 1, table->nframes);
    Format f;
    if (int r = parse_format(fmt, f)) return r;
    Bands b;
    if (int r = parse_bands(opts, f.height, b)) return r;
    if (frame_stride < required_len(f, b)) return fail(NT_E_INVALID, "frame_stride is smaller than one frame");
    if (int r = check_renderable(s)) return r;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->busy) return fail(NT_E_BUSY, "the renderer is already running");
    int dev;
    if (int r = pick_device(opts, -1, dev)) return r;
    if (dev != table->device) return fail(NT_E_INVALID, "the camera table lives on device %d, the render is for device %d", table->device, dev);
    DeviceState *ds;
    if (int r = device_state(s, dev, ds)) return r;
    if (int r = upload_scene(s, ds)) return r;
    if (int r = use_stream(ds, (hipStream_t)hip_stream)) return r;
    const bool stats = opts && opts->collect_stats;
    if (int r = prepare_stats(ds, (hipStream_t)hip_stream, stats)) return r;
    if (stats) { s->have_stats = false; s->stats_device = dev; }
    FrameJob job{};
    job.fmt = &f;
    job.bands = b;
    job.dest_dev = dest_dev;
    job.frame_stride = frame_stride;
    job.nframes = count;
    job.cam_buf = table->dev + (size_t)first * 4 * table->n;                                  // (the table: all cameras, then all dot products)
    job.cam_dots = table->dev + (size_t)table->nframes * 4 * table->n + (size_t)first * 4;
    job.stream = (hipStream_t)hip_stream;
    job.stats = stats;
    job.strict = opts && opts->strict_reference;
    job.abort_word = opts ? (const int *)opts->abort_device : nullptr;
    job.overlapped = opts ? opts->overlapped : 0;
    job.row_begin = 0;
    job.row_count = b.owned_rows;
    return enqueue(s, ds, job);
}

int nt_colors_at(nt_scene_t *s, int width, int height, int count, const int32_t *xs, const int32_t *ys, float *rgb, int device) {
    if (!s || (count > 0 && (!xs || !ys || !rgb))) return fail(NT_E_INVALID, "NULL argument");
    if (width < 1 || height < 1 || count < 0) return fail(NT_E_INVALID, "invalid view size or count");
    if (count == 0) return NT_OK;
    if (int r = check_renderable(s)) return r;
    RenderGuard guard(s);   // Scene.calculate_color locks the scene for the call (render.cpp:599-603)
    if (int r = guard.acquire()) return r;
    int dev;
    if (int r = pick_device(nullptr, device, dev)) return r;
    DeviceState *ds;
    if (int r = device_state(s, dev, ds)) return r;
    if (int r = upload_scene(s, ds)) return r;
    if (int r = own_stream(ds)) return r;
    if (int r = use_stream(ds, ds->stream)) return r;
    const size_t ibytes = (size_t)count * sizeof(int32_t);
    const size_t cbytes = (size_t)count * 3 * sizeof(float);
    if (int r = ds->probes.ensure(2 * ibytes + cbytes)) return r;
    char *base = (char *)ds->probes.p;
    HIP_TRY(hipMemcpyAsync(base, xs, ibytes, hipMemcpyHostToDevice, ds->stream));
    HIP_TRY(hipMemcpyAsync(base + ibytes, ys, ibytes, hipMemcpyHostToDevice, ds->stream));
    FrameJob job{};
    job.nframes = 1;
    job.stream = ds->stream;
    job.colors_out = (float *)(base + 2 * ibytes);
    job.xs = (const int *)base;
    job.ys = (const int *)(base + ibytes);
    job.probe_count = count;
    job.view_w = width;
    job.view_h = height;
    if (int r = enqueue(s, ds, job)) { (void)hipStreamSynchronize(ds->stream); return r; }
    HIP_TRY(hipMemcpyAsync(rgb, base + 2 * ibytes, cbytes, hipMemcpyDeviceToHost, ds->stream));
    HIP_TRY(hipStreamSynchronize(ds->stream));
    return NT_OK;
}

int nt_calculate_color(nt_scene_t *s, int x, int y, int width, int height, float rgb[3]) {
    const int32_t xs = x, ys = y;
    return nt_colors_at(s, width, height, 1, &xs, &ys, rgb, -1);
}

#ifdef NT_DEBUG_SCRATCH
// Diagnostic builds only (tools/box_census.py; not part of include/ntracer_hip.h): the BoxScene scratch of the last launch
// on `device` (stretch codes / redo words of the cull / box / redo path), after the device has drained.
long long nt_debug_box_scratch(nt_scene_t *s, int device, void *out, size_t bytes) {
    if (!s || !out) return fail(NT_E_INVALID, "NULL argument");
    auto it = s->devs.find(device);
    if (it == s->devs.end() || !it->second->cull.p) return fail(NT_E_INVALID, "no BoxScene launch on this device yet");
    if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(NT_E_DEVICE, "device synchronisation failed");
    const size_t n = std::min(bytes, it->second->cull.cap);
    if (hipMemcpy(out, it->second->cull.p, n, hipMemcpyDeviceToHost) != hipSuccess) return fail(NT_E_DEVICE, "copy failed");
    return (long long)n;
}
#endif

int nt_scene_last_stats(const nt_scene_t *cs, nt_stats *out) {
    nt_scene *s = const_cast<nt_scene *>(cs);
    if (!s || !out) return fail(NT_E_INVALID, "NULL argument");
    if (!s->have_stats) {
        if (s->stats_device < 0) return fail(NT_E_INVALID, "no render with collect_stats has run on this scene");
        auto it = s->devs.find(s->stats_device);
        if (it == s->devs.end() || !it->second->stats.p) return fail(NT_E_INVALID, "no statistics available");
        HIP_TRY(hipSetDevice(s->stats_device));
        HIP_TRY(hipDeviceSynchronize());
        unsigned long long v[8];
        HIP_TRY(hipMemcpy(v, it->second->stats.p, sizeof(v), hipMemcpyDeviceToHost));
        s->last_stats.rays = v[0]; s->last_stats.shadow_rays = v[1]; s->last_stats.branches = v[2]; s->last_stats.leaves = v[3];
        s->last_stats.simplex_tests = v[4]; s->last_stats.solid_tests = v[5]; s->last_stats.hits = v[6]; s->last_stats.aabb_enter = v[7];
        s->have_stats = true;
    }
    *out = s->last_stats;
    return NT_OK;
}

}  // extern "C"
