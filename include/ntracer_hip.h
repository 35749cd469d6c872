/*
 * ntracer_hip.h -- C ABI of libntracer_hip.so, the MI355X-native replacement for
 * NTracer's per-pixel ray-cast path.
 *
 * The reference has no C ABI: the path sits behind the in-process C++ plugin
 * interface `class scene` (reference src/render.hpp:8-26) and is driven by
 * `BlockingRenderer.render` / `CallbackRenderer.begin_render` /
 * `Scene.calculate_color` (src/render.cpp:853-909, :651-700, :586-614).
 * Each entry point below names the reference interface it replaces; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns NT_OK (0)
 * or a negative nt_status (never throws across the ABI); nt_last_error() gives
 * a thread-local message for the last failure on the calling thread.  Inputs
 * are copied -- the caller keeps ownership.  `dest` buffers are borrowed for the
 * duration of the call.  One render at a time per scene handle (NT_E_BUSY
 * otherwise, the reference's already_running_error, render.cpp:87-92); different
 * handles may be used concurrently from different threads.
 */
#ifndef NTRACER_HIP_H
#define NTRACER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NT_MAX_DIM 64            /* run-time-n kernels: n-vectors live in LDS */
#define NT_MAX_FIXED_DIM 10      /* compile-time-N kernels: 3..10 (reference: setup.py --optimize-dimensions, default 3..8) */
#define NT_MAX_FIXED_BOX_DIM 24  /* ... BoxScene's also for 11..24 */
#define NT_BATCH_SIZE 4          /* tracern.BATCH_SIZE of the SSE reference build (tracer.hpp:34-38) */
#define NT_MAX_PIXELSIZE 16      /* bytes per pixel, render.cpp:50 */
#define NT_MAX_BITSIZE 31        /* integer channel bits, render.cpp:48 */
#define NT_RENDER_CHUNK_SIZE 32  /* render.cpp:43 -- also the multi-GPU band height */

typedef enum {
    NT_OK = 0,
    NT_ABORTED = 1,              /* render stopped by the abort flag (BlockingRenderer.render -> False) */
    NT_E_INVALID = -1,           /* ValueError / TypeError in the reference */
    NT_E_BUSY = -2,              /* already_running_error (render.cpp:87-92) */
    NT_E_LOCKED = -3,            /* render.LockedError (ntracer_body.hpp:235-240) */
    NT_E_DEVICE = -4,            /* HIP runtime failure / no GPU: the product path never falls back to CPU */
    NT_E_NOMEM = -5,             /* MemoryError */
    NT_E_UNSUPPORTED = -6
} nt_status;

typedef struct nt_scene nt_scene_t;

/* render.Channel (render.cpp:95-99,120-164) */
typedef struct {
    float f_r, f_g, f_b, f_c;
    uint8_t bit_size;            /* 1..31, or 32 when tfloat */
    uint8_t tfloat;              /* raw IEEE-754 bits of the clamped value */
    uint8_t _pad[2];
} nt_channel;

/* render.ImageFormat (render.cpp:167-172,249-288) */
typedef struct {
    int32_t width, height;
    int32_t pitch;               /* bytes per row; 0 => width * bytes_per_pixel */
    int32_t nchannels;
    const nt_channel *channels;
    int32_t reversed;            /* emit each pixel's bytes in reverse order */
} nt_image_format;

/* render.Material (render.hpp:56-73; defaults render.cpp:1249-1252,1269) */
typedef struct {
    float color[3];
    float specular[3];
    float opacity, reflectivity, specular_intensity, specular_exp;
} nt_material;

/* leaf item encoding: (index << 2) | kind */
#define NT_KIND_BATCH 0          /* tracern.TriangleBatch: index = batch number, 4 simplices */
#define NT_KIND_TRIANGLE 1       /* tracern.Triangle (unbatched leftover) */
#define NT_KIND_SOLID 2          /* tracern.Solid */
#define NT_SOLID_CUBE 1          /* wrapper.CUBE */
#define NT_SOLID_SPHERE 2        /* wrapper.SPHERE */

/*
 * Flat description of a composite_scene (tracer.hpp:1710-1740): the k-d tree of
 * KDBranch/KDLeaf objects, its primitives and materials.
 * Simplex record (n*n + n + 1 floats): d, face_normal[n], p1[n], edge_normal[n-1][n]
 * (tracer.hpp:399-401,539-541).  Solid record (2*n*n + n floats): orientation[n][n],
 * inv_orientation[n][n], position[n] (tracer.hpp:237-239).
 */
typedef struct {
    int32_t dimension;
    int32_t root;                /* node index, or -1 for an empty scene */
    int32_t n_nodes;
    const int32_t *node_axis;    /* branch: split axis; leaf: -1 */
    const float *node_split;
    const int32_t *node_left;    /* branch: child (< split) or -1; leaf: first item */
    const int32_t *node_right;   /* branch: child (>= split) or -1; leaf: item count */
    int32_t n_items;
    const int32_t *items;        /* per leaf: batches first (tracer.hpp:1149) */
    int32_t n_batches;
    const float *batch_recs;     /* [n_batches][NT_BATCH_SIZE][record] */
    const int32_t *batch_mats;   /* [n_batches][NT_BATCH_SIZE] material index */
    int32_t n_triangles;
    const float *tri_recs;
    const int32_t *tri_mats;
    int32_t n_solids;
    const float *solid_recs;
    const int32_t *solid_types;
    const int32_t *solid_mats;
    int32_t n_materials;
    const nt_material *materials;
    const float *aabb_start;     /* scene boundary (tracern.AABB) */
    const float *aabb_end;
} nt_scene_desc;

/* composite_scene attributes (tracer.hpp:1713-1725; setters ntracer_body.hpp:833-933) */
typedef struct {
    int32_t shadows;             /* default 0 */
    int32_t camera_light;        /* default 1 */
    int32_t max_reflect_depth;   /* default 4 */
    int32_t bg_gradient_axis;    /* default 1 */
    float ambient[3];            /* default 0,0,0 */
    float bg1[3], bg2[3], bg3[3];/* default (1,1,1) (0,0,0) (0,1,1) */
    int32_t n_point_lights;
    const float *point_light_pos;    /* [n][dimension] */
    const float *point_light_color;  /* [n][3] */
    int32_t n_global_lights;
    const float *global_light_dir;   /* [n][dimension] */
    const float *global_light_color; /* [n][3] */
} nt_scene_params;

typedef struct {
    int32_t device;              /* HIP device ordinal; -1 => current device */
    int32_t band_rank;           /* this caller renders bands b with b % band_world == band_rank */
    int32_t band_world;          /* 0 or 1 => whole image */
    int32_t band_rows;           /* 0 => NT_RENDER_CHUNK_SIZE */
    int32_t compact;             /* 1: dest holds only the owned rows, packed in band order */
    int32_t strict_reference;    /* 0: closest-hit walks skip k-d cells that start beyond the current hit (same pixels, far fewer
                                    tests).  1: walk exactly the cells the reference walks
                                    (src/tracer.hpp:1179-1243).  NTRACER_STRICT_REFERENCE=1 in the environment forces 1,
                                    also for nt_colors_at / nt_calculate_color, which take no options. */
    int32_t collect_stats;       /* 1: count rays/nodes/tests with device atomics (slower).  Never changes the pixels: scenes whose
                                    frames come from kernels without counters (Solids with the reference's normal handling) are
                                    drawn as always and counted by a launch of their own (the counters then describe the
                                    clean-normal traversal); scenes with transparent materials, and n > 10: NT_E_UNSUPPORTED */
    int32_t overlapped;          /* device entry points: 1 = the caller keeps two or more streams busy with calls like this one
                                    (consecutive calls overlap on the device: the ramp and the tail of a call are filled by its
                                    neighbours), so the library shapes its launches for throughput rather than for the time of a
                                    call that runs alone (BoxScene: longer waves).  Never changes the pixels.  0: default */
    /* Abort for the device entry points (nt_render_device / nt_render_frames_device), which only enqueue: NULL, or a dword
       the DEVICE can read while the kernels run -- best in device memory, raised by a 4-byte copy on another stream
       (pinned host memory works too, but every block's look at it is then a PCIe round trip) -- that the caller sets
       non-zero to cancel; blocks that have not started then leave without drawing (the reference polls its CANCEL state
       per pixel, src/render.cpp:412).  nt_render keeps such a word itself and relays the caller's `abort_flag` to it;
       this field is ignored there. */
    const volatile int32_t *abort_device;
} nt_render_opts;

/* counters gathered when collect_stats is set (SURVEY section 8d byte model) */
typedef struct {
    uint64_t rays;               /* primary + reflection rays */
    uint64_t shadow_rays;
    uint64_t branches;
    uint64_t leaves;
    uint64_t simplex_tests;
    uint64_t solid_tests;
    uint64_t hits;
    uint64_t aabb_enter;
} nt_stats;

/* ---- library ---- */
const char *nt_version(void);
const char *nt_last_error(void);             /* thread-local; "" when none */
int nt_device_count(void);                   /* number of HIP devices, 0 when none */

/* ---- scenes ---- */
/* tracern.BoxScene(dimension): box_scene (tracer.hpp:83-123; ntracer_body.hpp:676-715) */
nt_scene_t *nt_box_scene_create(int dimension);
/* tracern.CompositeScene(boundary,data): composite_scene (tracer.hpp:1710-1740; ntracer_body.hpp:720-933) */
nt_scene_t *nt_composite_scene_create(const nt_scene_desc *desc);
void nt_scene_destroy(nt_scene_t *s);
int nt_scene_dimension(const nt_scene_t *s);
int nt_scene_is_composite(const nt_scene_t *s);

/* Scene.set_camera / get_camera (ntracer_body.hpp:676-700): origin[n], axes row-major [n][n]
   (rows: right, up, forward, ...; camera.hpp:40-45).  NT_E_LOCKED while a render holds the scene. */
int nt_scene_set_camera(nt_scene_t *s, const float *origin, const float *axes);
int nt_scene_get_camera(const nt_scene_t *s, float *origin, float *axes);
/* Scene.set_fov / .fov (radians; default 0.8, tracer.hpp:91,1731) */
int nt_scene_set_fov(nt_scene_t *s, float fov);
float nt_scene_get_fov(const nt_scene_t *s);
/* CompositeScene.set_shadows/set_camera_light/set_max_reflect_depth/set_ambient_color/
   set_background/add_light rolled into one call */
int nt_scene_set_params(nt_scene_t *s, const nt_scene_params *p);
/* class scene::lock()/unlock() (render.hpp:18-22) and the Python `locked` attribute */
int nt_scene_lock(nt_scene_t *s);
int nt_scene_unlock(nt_scene_t *s);
int nt_scene_locked(const nt_scene_t *s);

/* ---- rendering ---- */
/* ImageFormat.bytes_per_pixel (render.cpp:192-209); negative status on an invalid format */
int nt_format_bytes_per_pixel(const nt_image_format *fmt);

/* BlockingRenderer.render(dest, format, scene) (render.cpp:853-909): dest is HOST memory of at
   least pitch*height bytes (or the compact size).  abort_flag (may be NULL) is polled on the host
   while the frame's one launch runs and relayed to a dword the kernels read when a block starts: blocks that have not
   started leave without drawing.  Returns NT_ABORTED if the flag became non-zero before the frame was finished
   (signal_abort, render.cpp:911-923); an aborted frame is not copied back -- `dest` stays as the caller had it. */
int nt_render(nt_scene_t *s, void *dest, size_t dest_len, const nt_image_format *fmt,
              const nt_render_opts *opts, volatile int *abort_flag);

/* (Streams: the launches of one scene on one device share that scene's scratch buffers and are ordered by the stream they
   are enqueued on.  A call that names another stream than the scene's previous call on that device first waits, on the
   host, for the previous stream to drain -- alternate streams per scene handle, not within one.) */
/* Same frame loop, but dest is DEVICE memory on opts->device and the launch is only enqueued on
   `hip_stream` (a hipStream_t; NULL = the legacy default stream); no host synchronisation.  Used
   by the bench (framebuffer resident in HBM) and by the multi-GPU gather.  The scene must stay
   alive and unmodified until the stream has drained. */
int nt_render_device(nt_scene_t *s, void *dest_dev, size_t dest_len, const nt_image_format *fmt,
                     const nt_render_opts *opts, void *hip_stream);

/* Render `nframes` frames with per-frame cameras in ONE launch (the RotatingCamera loop of the
   reference's scripts/polytope.py:522-556 without a launch per frame).  origins [nframes][n],
   axes [nframes][n][n]; frame f goes to dest_dev + f*frame_stride. */
int nt_render_frames_device(nt_scene_t *s, void *dest_dev, size_t frame_stride, int nframes,
                            const float *origins, const float *axes, const nt_image_format *fmt,
                            const nt_render_opts *opts, void *hip_stream);

/* A camera path resident in device memory: the cameras of a sequence (the RotatingCamera loop of scripts/polytope.py:522-556
   has 160) packed and uploaded ONCE; nt_render_table_device then renders frames [first, first + count) of it exactly like
   nt_render_frames_device, but with nothing to pack or upload per call -- one kernel launch instead of two, which is what a
   render loop over a fixed path, and a rank's small share of a tiled frame, spend a tenth of their time on.
   The table belongs to the scene's dimension and to one device; it may be used by any scene of that dimension.
   After a warm-up call such a call only launches kernels and may be captured into a HIP graph; nt_render_frames_device, which
   stages host memory per call, returns NT_E_UNSUPPORTED on a capturing stream. */
typedef struct nt_camera_table nt_camera_table_t;
nt_camera_table_t *nt_camera_table_create(int dimension, int nframes, const float *origins, const float *axes, int device);
void nt_camera_table_destroy(nt_camera_table_t *t);
int nt_camera_table_frames(const nt_camera_table_t *t);
int nt_render_table_device(nt_scene_t *s, void *dest_dev, size_t frame_stride, const nt_camera_table_t *table, int first, int count,
                           const nt_image_format *fmt, const nt_render_opts *opts, void *hip_stream);

/* Scene.calculate_color(x,y,width,height) (render.cpp:586-614): unpacked fp32 colour of one pixel,
   computed by the same device code as nt_render. */
int nt_calculate_color(nt_scene_t *s, int x, int y, int width, int height, float rgb[3]);
/* batched form: `count` pixels (xs, ys) -> rgb[count][3] */
int nt_colors_at(nt_scene_t *s, int width, int height, int count, const int32_t *xs, const int32_t *ys,
                 float *rgb, int device);

/* statistics of the last render on this scene that had collect_stats set */
int nt_scene_last_stats(const nt_scene_t *s, nt_stats *out);

/* ---- scene construction (host only, no device needed) --------------------------------------------------
   build_kdtree / build_composite_scene of the reference (src/tracer.hpp:1965-2455; Python entry points
   src/ntracer_body.hpp:3250-3357).  Items are primitives or batches: a bounding box each, plus -- for
   simplices -- their vertices, simplex_first[i] .. simplex_first[i+1] indexing simplex_verts [count][n][n]
   (an item without simplices, e.g. a Solid, is placed by its box).  The tree comes back in the flat layout of
   nt_scene_desc (leaf: axis -1, left = first entry of leaf_items, right = count); leaf_items holds ITEM
   indices.  Arrays are malloc'ed; release with nt_kdtree_free. */
typedef struct {
    int32_t root;
    int32_t n_nodes;
    int32_t n_leaf_items;
    int32_t *node_axis;
    float *node_split;
    int32_t *node_left;
    int32_t *node_right;
    int32_t *leaf_items;
    float *aabb;                 /* start[n], end[n] */
} nt_kdtree;

/* kd_tree_params (tracer.hpp:1948-1961).  Zero / negative fields select the defaults: depth 25, threshold 2,
   costs 1 : 1 (the reference's per-dimension constants, :1933-1946, were tuned for its CPU walk; on the GPU a
   branch step is cheap next to a 4-simplex batch test). */
typedef struct {
    int32_t max_depth;           /* <= 64 */
    int32_t split_threshold;
    float traversal_cost;
    float intersection_cost;
} nt_kdtree_params;

int nt_kdtree_build(int dimension, int n_items, const float *item_lo, const float *item_hi, const int32_t *simplex_first,
                    const float *simplex_verts, const nt_kdtree_params *params, nt_kdtree *out);
void nt_kdtree_free(nt_kdtree *t);

/* aabb::intersects(prototype) of the reference (src/tracer.hpp:1465-1700), done by exact clipping: a convex polytope
   -- `n_verts` vertices [n_verts][dimension], each with the set of its facets' ids as a 192-bit mask `tight`
   [n_verts][3]; two vertices span an edge when they share `shared_for_edge` facets -- is cut by the 2*dimension
   half-spaces of the box [lo, hi] (which get the facet ids first_free_bit ...).  Returns the number of vertices of
   what is left (0: disjoint; out_lo / out_hi then untouched, else its bounding box), or a negative status.
   dimension <= 16.  A simplex: vertex i is on every facet but i, shared_for_edge = dimension-2; a parallelotope
   (solid cube): vertex on one facet of each of the dimension pairs, shared_for_edge = dimension-1. */
int nt_polytope_clip_box(int dimension, int n_verts, const float *verts, const uint64_t *tight, int shared_for_edge, int first_free_bit,
                         const float *lo, const float *hi, float *out_lo, float *out_hi);

#ifdef __cplusplus
}
#endif
#endif /* NTRACER_HIP_H */
