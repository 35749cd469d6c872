// nt_device.hpp -- structures shared by the host API (nt_api.cpp) and the HIP kernels
// (nt_box.hpp, nt_composite.hpp, nt_var.hip).  gfx950 only.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#define NT_DEV_MAX_DIM 64
#define NT_DEV_MAX_FIXED 10
#define NT_DEV_MAX_FIXED_BOX 24        // BoxScene kernels are also compiled for N = 11..24
#define NT_DEV_BATCH 4
#define NT_DEV_MAX_REFLECT 16

// reference: ROUNDING_FUZZ = numeric_limits<float>::epsilon()*10 (src/tracer.hpp:25)
#define NT_FUZZ (1.1920928955078125e-07f * 10.0f)
// reference: LIGHT_THRESHOLD (src/tracer.hpp:31)
#define NT_LIGHT_THRESHOLD (1.0f / 512.0f)

struct NtChanDev {          // render.cpp:95-99 channel, plus host-precomputed packing constants
    float f_r, f_g, f_b, f_c;
    uint32_t bits;
    uint32_t tfloat;
    uint32_t offset;        // first bit of the channel, counted from the pixel's most significant bit
    uint32_t maxval;        // 0xffffffff >> (32 - bits): the integer scale of render.cpp:439
};

// how a pixel is assembled (chosen on the host from the format)
#define NT_PACK_GENERIC 0   // up to 128 bits, any number of channels
#define NT_PACK_WORD32 1    // <= 32 bits and <= 4 live channels: one 32-bit container, unrolled
#define NT_PACK_WORD64 2    // <= 64 bits and <= 4 live channels: one 64-bit container, unrolled

// Where the pixels of one launch go.  Rows are dealt to ranks in bands of `band_rows`
// (RENDER_CHUNK_SIZE, render.cpp:43): owned row r -> image row
//   y = ((r / band_rows) * band_world + band_rank) * band_rows + r % band_rows.
struct NtTarget {
    uint8_t *dest;
    long long frame_stride;   // bytes between frames (blockIdx.z)
    const NtChanDev *chans;
    int nchannels, bpp, reversed, pitch;
    int pack_mode;            // NT_PACK_*; channels whose value is identically 0 are dropped from `chans`
    // plain RGB layouts in one dword (RGBX8, BGRA8, ...): plain_bits != 0, and component k goes to the fields
    // plain_mul[k] marks (a quantised component times plain_mul[k] is that component shifted into all of them)
    uint32_t plain_bits, plain_maxval, plain_mul[3];
    // ... and for 8-bit fields on byte boundaries (RGBX8, BGRA8, ...): the v_perm_b32 selector that builds the dword as it
    // lies in memory from (src0 = quantised R, src1 = quantised G = B); 0 when the layout is not of that kind
    uint32_t plain_sel;
    int plain_f32[3];         // 12-byte pixels of three fp32 channels that are plain components: component of float k; else -1
    int width, height;        // view size: set_view_size(w,h) (tracer.hpp:65-69)
    float half_w, half_h, fovI;
    int band_rank, band_world, band_rows, compact;
    int row_begin, row_count; // owned-row range rendered by this launch
    int aligned4;             // dest, pitch and frame_stride are 4-byte aligned
    // probe mode (nt_colors_at / nt_calculate_color): fp32 colours of listed pixels
    float *colors_out;
    const int *probe_xs, *probe_ys;
    int probe_count;
    // two-pass renders of lit scenes: primary hits found by the packet kernel, [frame][row][x] records of 16 bytes
    // (dist, item, lane, -); nullptr otherwise
    const void *hits;
    // BoxScene: four bits per (frame, owned row, 64-pixel stretch), eight stretches to a dword, from box_cull_kernel:
    //   0: no ray of the stretch can reach the cube; 1..8: every ray of it clearly hits face K = code - 1;
    //   14: left to box_redo_kernel (redo bit already set); 15: look at each ray.  [frame][row][cull_words] dwords, or nullptr
    const uint32_t *cull;
    int cull_words;
    // one bit per stretch, [frame][row][redo_words] dwords: box_kernel sets the bit of a stretch it leaves to
    // box_redo_kernel (a lane needed the reference's own face-by-face arithmetic); cleared by box_cull_kernel
    uint32_t *redo;
    int redo_words;
    // box_tile_kernel -> box_redo_kernel: for a marked stretch, the faces and coordinates the reference's arithmetic is
    // needed on, for all its rays (box_stretch_code; 0: work them out ray by ray); [frame][row][stretch] dwords, or nullptr
    uint32_t *tie_sets;
    // BoxScene tile kernel: per owned row (index = owned-row number; 64 entries of padding) 16 bytes {float sy = fovI*(y -
    // half_h); uint32 y < height; int64 byte offset of the row within a frame}, read with scalar loads; or nullptr
    const void *rowtab;
    // box_tile_kernel, interleaved rows: 0 = a wave renders ROWS consecutive rows (table entry = owned-row number); W > 0 = the
    // W waves of a column strip (W = gridDim.y * WAVES) deal the rows out among themselves, wave w renders rows w, w + W,
    // w + 2W, ... -- every wave of a strip then holds the same share of the rows that need ray-by-ray work -- and the row
    // table is in SLOT order: entry w * ROWS + rr belongs to row w + W * rr (valid = 0 past the last row).
    int row_il;
    // Abort word (renderer::CANCEL, polled per pixel by the reference: render.cpp:412): nullptr, or a device-visible dword --
    // pinned host memory mapped into the device's address space -- that the kernels read past the caches when a block (or a
    // tile of a striding block) starts; non-zero: the block leaves without drawing
    const int *abort_word;
    // box_tile_kernel: the middle columns of the image are started `lead_frames` frames ahead of the outer ones (see the kernel); 0: off
    int lead_frames;
};

// Camera rows used by the ray source (camera.hpp:40-45): origin, right, up, forward.
// Either inline in the kernel arguments (single frame) or from a device buffer
// [frame][4][n] (multi-frame launches).
// Four ray-independent dot products travel with the camera for BoxScene's circumsphere rejection (conservative,
// not part of the exact predicate): |origin|^2, origin.right, origin.up, origin.forward -- `odots` for the inline
// camera, four floats per frame after the last camera of a table.
struct NtCamera {
    const float *buf;         // nullptr => use `inl`; else [nframes][4][n] cameras of the launch's frames
    const float *dots;        // with buf: [nframes][4] dot products of the same frames (|o|^2, o.right, o.up, o.forward)
    int n;
    float odots[4];
    float inl[4 * NT_DEV_MAX_DIM];
};
struct NtCameraFixed {        // N <= 8: 4*8 floats inline
    const float *buf;
    const float *dots;
    int n;
    float odots[4];
    float inl[4 * NT_DEV_MAX_FIXED_BOX];
};

struct NtNode {               // 16-byte k-d node record
    float split;
    int axis;                 // -1: leaf
    int left;                 // branch: child or -1; leaf: first item
    int right;                // branch: child or -1; leaf: item count
};

struct NtCompositeDev {
    const NtNode *nodes;
    const int *items;
    const float *batch_recs;  // [n_batches*4][rec_stride]
    const int *batch_mats;
    const float *tri_recs;    // [n_triangles][rec_stride]
    const int *tri_mats;
    const float *solid_recs;  // [n_solids][2*n*n+n]
    const int *solid_types;
    const int *solid_mats;
    const float *materials;   // [n_materials][10]
    const float *aabb;        // start[n], end[n]
    int rec_stride;           // floats per simplex record, multiple of 4
    int root;
    int stack_depth;          // LDS stack entries per lane
    int shadows, camera_light, max_reflect_depth, bg_axis;
    float ambient[3], bg1[3], bg2[3], bg3[3];
    int n_point_lights;
    const float *pl_pos;
    const float *pl_color;
    int n_global_lights;
    const float *gl_dir;
    const float *gl_color;
    int all_opaque;           // every material has opacity >= 1
    int any_reflective;
    int has_scalar_prims;     // leaves hold unbatched triangles or solids
    int n_batches, n_solids;
    int prune;                // 1: closest-hit walks drop subtrees that start clearly beyond the current hit (nt_beyond_hit)
    unsigned long long *stats;  // nullptr or 8 counters (nt_stats order)
    // Reference-faithful normals (composite_kernel_t<N, true>): the reference's first leaf loop hands o_hit.normal itself to
    // the primitive tests (tracer.hpp:1001,1020), so which tests run -- its exact `checked` list (:782,:832) -- matters.
    // One bit per (resident lane, primitive): checked[word * checked_lanes + lane slot]; nullptr selects the "clean"
    // semantics with the 16-slot mailbox.
    uint32_t *checked;
    int alias_normals;        // 1: the reference's o_hit.normal handling; 0: a hit keeps the normal of what was hit (NTRACER_CLEAN_NORMALS)
    int checked_words;        // ceil((n_batches + n_triangles + n_solids) / 32)
    int checked_lanes;        // lane slots = blocks of the launch * 256
    int n_triangles;
    // run-time-n transparency kernel (composite_kernel_var_t): the ray_color frame stacks in global scratch,
    // tframes[(frame * words + word) * checked_lanes + lane slot], words = var_frame_words(n); nullptr = not that kernel
    float *tframes;
    int tframe_count;         // frames per lane slot: max_reflect_depth + 1 if anything reflects, else 1
};

// ---- launchers implemented in nt_var.hip (dispatch) over nt_inst_box.hip / nt_inst_composite.hip ----
struct NtLaunchInfo {
    int n;                    // dimension
    int nframes;
    void *stream;             // hipStream_t
    // persistent composite kernel (optional): a zeroed 8-byte work counter, the camera table
    // [nframes][4][n] in device memory, and the CU count of the device
    void *persist_counter;
    const float *persist_cams;
    int cu_count;
    int kernel_choice;        // 0: default (packet kernel for lean scenes), 1: persistent per-lane kernel, 2: tile kernel
    const int *tile_order;    // packet kernel: device permutation of the 16x16-pixel quads of the launch (or nullptr)
    void *hit_buf;            // scratch for two-pass renders: hit_frames * width * rows * 16 bytes (or nullptr)
    int hit_frames;
    float *numer_buf;         // scratch for the packet kernel's plane numerators: numer_frames * n_batches * 4 floats
    int numer_frames;
    int tile_rows, tile_waves; // BoxScene, fused path: box_tile_kernel's block shape (nt_box_tile_geom; the row table follows it)
    int box_path;             // BoxScene: 1 = fused tile kernel for the scripted formats (default), 0 = cull / box / redo kernels
    int cull_clean;           // cull_buf is all zero (the fused path's redo bitmap lives at its start)
    uint32_t *tie_buf;        // BoxScene: scratch for the tie sets of the fused path, nframes * row_count * ceil(width/64) dwords (or nullptr)
    uint32_t *cull_buf;       // BoxScene: scratch for the row culling bits, 5 * nframes * row_count * ceil(ceil(width/64)/32) dwords: stretch codes, then redo bits (or nullptr)
};

// box_tile_kernel's block shape for a launch, decided in one place because the host's row table (nt_api.cpp) follows it:
// 64 rows a wave and one wave a block for tall launches with waves to spare; otherwise 16 rows a wave (8 in small launches),
// four waves a block, or three when that leaves fewer idle waves below the last row (see launch_box_fixed).
// `overlapped` (nt_render_opts::overlapped): the caller keeps two or more streams busy with calls like this one, so the
// ramp and the tail of a call are filled by its neighbours.  Long waves -- whose tail is what makes them lose on a short
// launch that runs alone -- are then the better shape from 64 rows up: a rank's 136 rows of the 160 headline frames take
// 66 us alone with 16 x 3 and 86 with 64 x 1, 54 and 48 when consecutive calls alternate between two streams.
struct NtBoxTileGeom { int rows, waves; };
static inline NtBoxTileGeom nt_box_tile_geom(int width, int row_count, int nframes, int overlapped) {
    const long long cols = (width + 63) / 64;
    const long long waves8 = cols * ((row_count + 31) / 32) * nframes * 4;
    const bool r16 = waves8 >= 64 * 1024;
    int wpb = 4;
    if (r16) {
        const int groups = (row_count + 15) / 16;                   // waves with rows, per column
        if ((groups + 2) / 3 * 3 < (groups + 3) / 4 * 4) wpb = 3;
    }
    const long long waves64 = cols * ((row_count + 63) / 64) * nframes;
    bool r64 = r16 && row_count >= 512 && waves64 >= 32 * 1024;
    if (overlapped && r16 && row_count >= 64 && waves64 >= 8 * 1024) r64 = true;
    if (const char *e = getenv("NTRACER_BOX_R64")) r64 = r16 && atoi(e) != 0;        // (A/B)
    NtBoxTileGeom g;
    g.rows = r64 ? 64 : (r16 ? 16 : 8);
    g.waves = r64 ? 1 : wpb;
    return g;
}

int nt_launch_box(const NtLaunchInfo &li, const NtCamera &cam, const NtTarget &tg);
int nt_launch_composite(const NtLaunchInfo &li, const NtCamera &cam, const NtCompositeDev &sc, const NtTarget &tg);
int nt_launch_upload(void *stream, const float *src_pinned, float *dst, int count);
int nt_var_frame_words(int n);   // floats per ray_color frame of composite_kernel_var_t
const char *nt_launch_error();
