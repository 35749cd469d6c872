/*
 * ntracer_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ntracer_oracle.h).
 *
 * Plain-C restatement of the reference's ray-cast path.  Every function cites
 * the reference lines (relative to /root/reference) it follows.  The structure
 * deliberately mirrors the reference's recursion and its in/out "normal"
 * buffers, including the places where a failed or transparent test scribbles on
 * the current opaque hit's normal (tracer.hpp:1001,1020 pass o_hit.normal), so
 * that golden pixels captured from the compiled reference are reproduced.
 *
 * Known, intentional deviations (hard part 3):
 *   - quick_list::check_capacity's byte-count memcpy (tracer.hpp:670-679) is NOT
 *     reproduced: lists grow correctly here.
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off, no -ffast-math)
 */
#define _POSIX_C_SOURCE 200809L      /* clock_gettime, sysconf under -std=c99 */
#include "ntracer_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define MAXD NTO_MAX_DIM
#define ROUNDING_FUZZ (FLT_EPSILON * 10.0f)       /* tracer.hpp:25 */
#define LIGHT_THRESHOLD (1.0f / 512.0f)           /* tracer.hpp:31 */
#define RENDER_CHUNK_SIZE 32                      /* render.cpp:43 */

typedef struct { float origin[MAXD]; float direction[MAXD]; } ray_t;
typedef struct { int item; int lane; } target_t;                 /* intersection_target<Store,true>: tracer.hpp:744-763 */
typedef struct { float dist; target_t target; ray_t normal; } isect_t;   /* ray_intersection: tracer.hpp:765-779 */
typedef struct { isect_t *data; size_t size, cap; } isect_list;  /* quick_list<ray_intersection> */
typedef struct { int *data; size_t size, cap; } prim_list;       /* quick_list<void*> mailbox */
typedef struct { float r, g, b; } color_t;

typedef struct {
    const nto_scene *s;
    nto_counters *c;
} ctx_t;

/* ---------------- small helpers (geometry.hpp:131-283, light.hpp) ---------------- */

static float dotn(int n, const float *a, const float *b) {          /* geometry.hpp:279-282 */
    float s = a[0] * b[0];
    for (int i = 1; i < n; ++i) s = s + a[i] * b[i];
    return s;
}

static color_t col(float r, float g, float b) { color_t c = {r, g, b}; return c; }
static color_t cadd(color_t a, color_t b) { return col(a.r + b.r, a.g + b.g, a.b + b.b); }
static color_t cmul(color_t a, color_t b) { return col(a.r * b.r, a.g * b.g, a.b * b.b); }
static color_t cscale(color_t a, float s) { return col(a.r * s, a.g * s, a.b * s); }
static color_t col3(const float *p) { return col(p[0], p[1], p[2]); }

static void il_add(isect_list *l, const isect_t *it) {
    if (l->size >= l->cap) {
        l->cap = l->cap ? l->cap * 2 : 10;
        l->data = (isect_t *)realloc(l->data, l->cap * sizeof(isect_t));
    }
    l->data[l->size++] = *it;
}
static void il_remove_at(isect_list *l, size_t i) {                 /* tracer.hpp:723-728 */
    --l->size;
    if (i != l->size) l->data[i] = l->data[l->size];
}
static void pl_add(prim_list *l, int item) {
    if (l->size >= l->cap) {
        l->cap = l->cap ? l->cap * 2 : 20;
        l->data = (int *)realloc(l->data, l->cap * sizeof(int));
    }
    l->data[l->size++] = item;
}
static int has(const prim_list *l, int item) {                      /* tracer.hpp:832-834 */
    for (size_t i = 0; i < l->size; ++i) if (l->data[i] == item) return 1;
    return 0;
}
static void trim_intersections(isect_list *hits, float dist, size_t from) {   /* tracer.hpp:784-789 */
    while (from < hits->size) {
        if (hits->data[from].dist >= dist) il_remove_at(hits, from);
        else ++from;
    }
}

static int rec_len(int n) { return n * n + n + 1; }

static const float *material_of(const nto_scene *s, target_t t) {   /* intersection_target::mat: tracer.hpp:752-762 */
    int kind = t.item & 3, idx = t.item >> 2, m;
    if (kind == NTO_KIND_BATCH) m = s->batch_mats[idx * s->batch_size + t.lane];
    else if (kind == NTO_KIND_TRIANGLE) m = s->tri_mats[idx];
    else m = s->solid_mats[idx];
    return s->materials + 10 * m;
}
static int opaque(const nto_scene *s, target_t t) { return material_of(s, t)[6] >= 1.0f; }   /* tracer.hpp:187,211 */

/* ---------------- ray generation: tracer.hpp:60-76 ---------------- */

static void primary_dir(const nto_scene *s, int w, int h, float x, float y, float *dir) {
    int n = s->n;
    float half_w = (float)w / 2.0f;
    float half_h = (float)h / 2.0f;
    float fovI = tanf(s->fov / 2.0f) / half_w;
    const float *right = s->axes, *up = s->axes + n, *fwd = s->axes + 2 * n;
    float sx = fovI * (x - half_w);
    float sy = fovI * (y - half_h);
    float tmp[MAXD];
    for (int j = 0; j < n; ++j) tmp[j] = (fwd[j] + right[j] * sx) - up[j] * sy;
    float len = sqrtf(dotn(n, tmp, tmp));
    for (int j = 0; j < n; ++j) dir[j] = tmp[j] / len;
}

void nto_primary_dir(const nto_scene *s, int x, int y, int w, int h, float *dir_out) {
    primary_dir(s, w, h, (float)x, (float)y, dir_out);
}

/* ---------------- hypercube / hypersphere: tracer.hpp:126-173 ---------------- */

static float hypercube_intersects(int n, const ray_t *target, ray_t *normal, float cutoff) {
    for (int i = 0; i < n; ++i) {
        if (target->direction[i] != 0.0f) {
            normal->origin[i] = target->direction[i] < 0 ? 1.0f : -1.0f;
            float dist = (normal->origin[i] - target->origin[i]) / target->direction[i];
            if (dist > 0) {
                int miss = 0;
                for (int j = 0; j < n; ++j) {
                    if (i != j) {
                        normal->origin[j] = target->direction[j] * dist + target->origin[j];
                        if (fabsf(normal->origin[j]) > (1 + ROUNDING_FUZZ)) { miss = 1; break; }
                    }
                }
                if (!miss) {
                    if (dist >= cutoff) return 0;
                    for (int j = 0; j < n; ++j) normal->direction[j] = 0.0f;
                    normal->direction[i] = normal->origin[i];
                    return dist;
                }
            }
        }
    }
    return 0;
}

static float hypersphere_intersects(int n, const ray_t *target, ray_t *normal, float cutoff) {
    float a = dotn(n, target->direction, target->direction);
    float b = 2 * dotn(n, target->direction, target->origin);
    float c = dotn(n, target->origin, target->origin) - 1;
    float discriminant = b * b - 4 * a * c;
    if (discriminant < 0) return 0;
    float dist = (-b - sqrtf(discriminant)) / (2 * a);
    if (dist <= 0 || dist >= cutoff) return 0;
    for (int j = 0; j < n; ++j) normal->direction[j] = normal->origin[j] = target->origin[j] + target->direction[j] * dist;
    return dist;
}

/* ---------------- box_scene::calculate_color: tracer.hpp:101-114 ---------------- */

static color_t box_color(const nto_scene *s, int w, int h, int x, int y, nto_counters *c) {
    int n = s->n;
    ray_t view, normal;
    memcpy(view.origin, s->origin, sizeof(float) * n);
    primary_dir(s, w, h, (float)x, (float)y, view.direction);
    if (c) c->rays++;
    if (hypercube_intersects(n, &view, &normal, FLT_MAX) != 0.0f) {
        float sine = dotn(n, view.direction, normal.direction);
        if (c) c->hits++;
        return cscale(col(1.0f, 0.5f, 0.5f), sine <= 0 ? -sine : 0.0f);
    }
    float intensity = view.direction[0];
    return intensity > 0 ? col(intensity, intensity, intensity) : col(0, -intensity, -intensity);
}

/* ---------------- primitives ---------------- */

/* triangle::intersects: tracer.hpp:411-440 */
static float triangle_intersects(const nto_scene *s, const float *rec, const ray_t *target, ray_t *normal, float cutoff) {
    int n = s->n;
    float d = rec[0];
    const float *face_normal = rec + 1, *p1 = rec + 1 + n, *edges = rec + 1 + 2 * n;
    float denom = dotn(n, face_normal, target->direction);
    if (denom == 0.0f) return 0;
    float t = -(dotn(n, face_normal, target->origin) + d) / denom;
    if (t <= 0 || t >= cutoff) return 0;
    float P[MAXD], pside[MAXD];
    for (int k = 0; k < n; ++k) P[k] = target->origin[k] + t * target->direction[k];
    for (int k = 0; k < n; ++k) pside[k] = p1[k] - P[k];
    float tot_area = 0;
    for (int i = 0; i < n - 1; ++i) {
        float area = dotn(n, edges + i * n, pside);
        if (area < -ROUNDING_FUZZ || area > (1 + ROUNDING_FUZZ)) return 0;
        tot_area += area;
    }
    if (tot_area <= (1 + ROUNDING_FUZZ)) {
        memcpy(normal->origin, P, sizeof(float) * n);
        float len = sqrtf(dotn(n, face_normal, face_normal));
        for (int k = 0; k < n; ++k) normal->direction[k] = face_normal[k] / len;
        if (denom > 0) for (int k = 0; k < n; ++k) normal->direction[k] = -normal->direction[k];
        return t;
    }
    return 0;
}

/* triangle_batch::intersects: tracer.hpp:551-599 (one SIMD lane per simplex) */
static float batch_intersects(ctx_t *cx, int batch, const ray_t *target, ray_t *normal, int *index, float cutoff) {
    const nto_scene *s = cx->s;
    int n = s->n, B = s->batch_size, rl = rec_len(n);
    float tl[16], denoml[16];
    if (cx->c) { cx->c->batch_tests++; cx->c->simplex_tests += (uint64_t)B; }
    for (int l = 0; l < B; ++l) {
        const float *rec = s->batch_recs + ((size_t)batch * B + l) * rl;
        float d = rec[0];
        const float *face_normal = rec + 1, *p1 = rec + 1 + n, *edges = rec + 1 + 2 * n;
        float denom = dotn(n, face_normal, target->direction);
        int mask = denom != 0.0f;
        float t = -(dotn(n, face_normal, target->origin) + d) / denom;
        mask = mask && t >= 0.0f;
        float pside[MAXD];
        for (int k = 0; k < n; ++k) pside[k] = p1[k] - (target->origin[k] + t * target->direction[k]);
        float tot_area = 0;
        for (int i = 0; i < n - 1; ++i) {
            float area = dotn(n, edges + i * n, pside);
            mask = mask && area >= -ROUNDING_FUZZ;
            tot_area += area;
        }
        mask = mask && tot_area <= (1 + ROUNDING_FUZZ);
        tl[l] = mask ? t : 0.0f;
        denoml[l] = denom;
    }
    float min_t = cutoff;
    int r_index = -1;
    for (int i = 0; i < B; ++i) {
        if (i != *index && tl[i] != 0.0f && tl[i] < min_t) { min_t = tl[i]; r_index = i; }
    }
    if (r_index == -1) return 0;
    *index = r_index;
    const float *rec = s->batch_recs + ((size_t)batch * B + r_index) * rl;
    const float *face_normal = rec + 1;
    for (int k = 0; k < n; ++k) normal->origin[k] = target->origin[k] + min_t * target->direction[k];
    float len = sqrtf(dotn(n, face_normal, face_normal));
    for (int k = 0; k < n; ++k) normal->direction[k] = face_normal[k] / len;
    if (denoml[r_index] > 0) for (int k = 0; k < n; ++k) normal->direction[k] = -normal->direction[k];
    return min_t;
}

/* solid::intersects: tracer.hpp:251-276 */
static float solid_intersects(const nto_scene *s, int idx, const ray_t *target, ray_t *normal, float cutoff) {
    int n = s->n;
    const float *orientation = s->solid_recs + (size_t)idx * (2 * n * n + n);
    const float *inv_orientation = orientation + n * n;
    const float *position = inv_orientation + n * n;
    ray_t transformed;
    for (int i = 0; i < n; ++i) {
        transformed.origin[i] = dotn(n, inv_orientation + i * n, target->origin) - position[i];
        transformed.direction[i] = dotn(n, inv_orientation + i * n, target->direction);
    }
    float dist;
    if (s->solid_types[idx] == NTO_SOLID_CUBE) dist = hypercube_intersects(n, &transformed, normal, cutoff);
    else dist = hypersphere_intersects(n, &transformed, normal, cutoff);
    if (dist == 0.0f) return 0;
    float tmp[MAXD], o2[MAXD], d2[MAXD];
    for (int i = 0; i < n; ++i) tmp[i] = normal->origin[i] + position[i];
    for (int i = 0; i < n; ++i) { o2[i] = dotn(n, orientation + i * n, tmp); d2[i] = dotn(n, orientation + i * n, normal->direction); }
    memcpy(normal->origin, o2, sizeof(float) * n);
    memcpy(normal->direction, d2, sizeof(float) * n);
    return dist;
}

/* primitive::intersects dispatch: tracer.hpp:508-516 */
static float primitive_intersects(ctx_t *cx, int item, const ray_t *target, ray_t *normal, float cutoff) {
    const nto_scene *s = cx->s;
    int kind = item & 3, idx = item >> 2;
    if (kind == NTO_KIND_TRIANGLE) {
        if (cx->c) cx->c->simplex_tests++;
        return triangle_intersects(s, s->tri_recs + (size_t)idx * rec_len(s->n), target, normal, cutoff);
    }
    if (cx->c) cx->c->solid_tests++;
    return solid_intersects(s, idx, target, normal, cutoff);
}

/* ---------------- k-d leaf: tracer.hpp:977-1124 ---------------- */

static int leaf_intersects(ctx_t *cx, int node, const ray_t *target, target_t skip, isect_t *o_hit,
                           isect_list *t_hits, prim_list *checked) {
    const nto_scene *s = cx->s;
    int n = s->n;
    const int *items = s->items + s->node_left[node];
    int size = s->node_right[node];
    size_t h_start = t_hits->size;
    float dist = 0;
    int i = 0;
    int hit = 0;
    ray_t clean_tmp;
    /* the reference hands o_hit.normal itself to the tests of the first loop (tracer.hpp:1001,1020) */
    ray_t *scratch = s->clean_normals ? &clean_tmp : &o_hit->normal;
    if (cx->c) cx->c->leaves++;

    for (; i < size; ++i) {
        int item = items[i];
        if ((item & 3) == NTO_KIND_BATCH) {
            if (!has(checked, item)) {
                int index = skip.item == item ? skip.lane : -1;
                dist = batch_intersects(cx, item >> 2, target, scratch, &index, o_hit->dist);
                if (dist != 0.0f) {
                    target_t tg = {item, index};
                    if (opaque(s, tg)) {
                        o_hit->dist = dist;
                        o_hit->target = tg;
                        if (scratch != &o_hit->normal) o_hit->normal = *scratch;
                        hit = 1;
                        break;   /* goto hit: `i` is NOT advanced (see below) */
                    }
                    isect_t th; th.dist = dist; th.target = tg; th.normal = *scratch;
                    il_add(t_hits, &th);
                }
                pl_add(checked, item);
            }
        } else if (item != skip.item && !has(checked, item)) {
            dist = primitive_intersects(cx, item, target, scratch, o_hit->dist);
            if (dist != 0.0f) {
                target_t tg = {item, -1};
                if (opaque(s, tg)) {
                    o_hit->dist = dist;
                    o_hit->target = tg;
                    if (scratch != &o_hit->normal) o_hit->normal = *scratch;
                    hit = 1;
                    break;
                }
                isect_t th; th.dist = dist; th.target = tg; th.normal = *scratch;
                il_add(t_hits, &th);
            }
            pl_add(checked, item);
        }
    }
    if (!hit) return 0;

    /* "is there anything closer?" -- tracer.hpp:1037-1085.  NOTE the reference jumps here
       with `i` still pointing at the item that hit and without adding it to `checked`, so
       that item is tested once more against the tightened cutoff (and fails: t < t is false),
       then joins `checked`. */
    ray_t new_normal;
    for (; i < size; ++i) {
        int item = items[i];
        if ((item & 3) == NTO_KIND_BATCH) {
            if (!has(checked, item)) {
                int index = skip.item == item ? skip.lane : -1;
                dist = batch_intersects(cx, item >> 2, target, &new_normal, &index, o_hit->dist);
                if (dist != 0.0f) {
                    target_t tg = {item, index};
                    if (opaque(s, tg)) {
                        o_hit->dist = dist;
                        memcpy(o_hit->normal.origin, new_normal.origin, sizeof(float) * n);
                        memcpy(o_hit->normal.direction, new_normal.direction, sizeof(float) * n);
                        o_hit->target = tg;
                    } else {
                        isect_t th; th.dist = dist; th.target = tg; th.normal = new_normal;
                        il_add(t_hits, &th);
                    }
                }
                pl_add(checked, item);
            }
        } else if (item != skip.item && !has(checked, item)) {
            dist = primitive_intersects(cx, item, target, &new_normal, o_hit->dist);
            if (dist != 0.0f) {
                target_t tg = {item, -1};
                if (opaque(s, tg)) {
                    o_hit->dist = dist;
                    memcpy(o_hit->normal.origin, new_normal.origin, sizeof(float) * n);
                    memcpy(o_hit->normal.direction, new_normal.direction, sizeof(float) * n);
                    o_hit->target = tg;
                } else {
                    isect_t th; th.dist = dist; th.target = tg; th.normal = new_normal;
                    il_add(t_hits, &th);
                }
            }
            pl_add(checked, item);
        }
    }
    trim_intersections(t_hits, dist, h_start);   /* sic: the LAST test's dist (tracer.hpp:1084) */
    return 1;
}

static int leaf_occludes(ctx_t *cx, int node, const ray_t *target, float ldistance, target_t skip, isect_list *hits) {
    const nto_scene *s = cx->s;
    const int *items = s->items + s->node_left[node];
    int size = s->node_right[node];
    ray_t normal;
    if (cx->c) cx->c->leaves++;
    for (int i = 0; i < size; ++i) {
        int item = items[i];
        if ((item & 3) == NTO_KIND_BATCH) {
            int index = skip.item == item ? skip.lane : -1;
            float dist = batch_intersects(cx, item >> 2, target, &normal, &index, ldistance);
            if (dist != 0.0f) {
                target_t tg = {item, index};
                if (opaque(s, tg)) return 1;
                isect_t th; th.dist = dist; th.target = tg; th.normal = normal;
                il_add(hits, &th);
            }
        } else if (item != skip.item) {
            float dist = primitive_intersects(cx, item, target, &normal, ldistance);
            if (dist != 0.0f) {
                target_t tg = {item, -1};
                if (opaque(s, tg)) return 1;
                isect_t th; th.dist = dist; th.target = tg; th.normal = normal;
                il_add(hits, &th);
            }
        }
    }
    return 0;
}

/* ---------------- k-d traversal: tracer.hpp:1159-1256 ---------------- */

typedef struct {
    ctx_t *cx;
    const ray_t *target;
    float invdir[MAXD];
    target_t skip;
    isect_t *o_hit;
    isect_list *t_hits;
    prim_list checked;
} kd_isect_t;

static int kd_visit(kd_isect_t *k, int node, float t_near, float t_far) {
    const nto_scene *s = k->cx->s;
    while (node >= 0) {
        /* optional (NOT in the reference; see nto_scene.prune_beyond_hit): a cell that begins clearly beyond
           the nearest hit so far cannot hold a closer one */
        if (s->prune_beyond_hit && k->o_hit->dist < t_near - 1e-4f * (1.0f + fabsf(t_near))) return 0;
        int axis = s->node_axis[node];
        if (axis < 0) return leaf_intersects(k->cx, node, k->target, k->skip, k->o_hit, k->t_hits, &k->checked);
        if (k->cx->c) k->cx->c->branches++;
        float split = s->node_split[node];
        int left = s->node_left[node], right = s->node_right[node];
        float o = k->target->origin[axis], d = k->target->direction[axis];
        if (d != 0.0f) {
            if (o == split) { node = d > 0 ? right : left; continue; }
            float t = (split - o) * k->invdir[axis];
            int n_near = o > split ? right : left;
            int n_far = o > split ? left : right;
            if (t < 0 || t > t_far) { node = n_near; continue; }
            if (t < t_near) { node = n_far; continue; }
            if (n_near >= 0) {
                size_t h_start = k->t_hits->size;
                int hit = kd_visit(k, n_near, t_near, t);
                if ((hit && k->o_hit->dist <= t) || n_far < 0) return hit;
                if (hit) {
                    if (kd_visit(k, n_far, t, t_far)) trim_intersections(k->t_hits, k->o_hit->dist, h_start);
                    return 1;
                }
            }
            node = n_far;
            t_near = t;
            continue;
        }
        node = o >= split ? right : left;
    }
    return 0;
}

static int kd_intersects(ctx_t *cx, int root, const ray_t *target, target_t skip, isect_t *o_hit, isect_list *t_hits,
                         float t_near, float t_far) {
    kd_isect_t k;
    k.cx = cx; k.target = target; k.skip = skip; k.o_hit = o_hit; k.t_hits = t_hits;
    k.checked.data = NULL; k.checked.size = k.checked.cap = 0;
    for (int i = 0; i < cx->s->n; ++i) k.invdir[i] = 1.0f / target->direction[i];   /* tracer.hpp:1174 */
    int r = kd_visit(&k, root, t_near, t_far);
    free(k.checked.data);
    return r;
}

/* _occludes: tracer.hpp:1258-1311 (including the far-child quirk at :1298) */
static int kd_occludes_rec(ctx_t *cx, int node, const ray_t *target, const float *invdir, float ldistance, target_t skip,
                           isect_list *hits, float t_near, float t_far) {
    const nto_scene *s = cx->s;
    while (node >= 0) {
        int axis = s->node_axis[node];
        if (axis < 0) return leaf_occludes(cx, node, target, ldistance, skip, hits);
        if (cx->c) cx->c->branches++;
        float split = s->node_split[node];
        int left = s->node_left[node], right = s->node_right[node];
        float o = target->origin[axis], d = target->direction[axis];
        if (d != 0.0f) {
            if (o == split) { node = d > 0 ? right : left; continue; }
            float t = (split - o) * invdir[axis];
            int n_near = left, n_far = right;
            if (o > split) { n_near = right; n_far = left; }
            if (t < 0 || t > t_far) { node = n_near; continue; }
            if (t < t_near) { node = n_far; continue; }
            if (n_near >= 0) {
                if (n_far < 0) { t_far = t; node = n_near; continue; }
                if (kd_occludes_rec(cx, n_near, target, invdir, ldistance, skip, hits, t_near, t)) return 1;
            }
            if (t < ldistance) return 0;
            t_near = t;
            node = n_far;
            continue;
        }
        node = o >= split ? right : left;
    }
    return 0;
}

static int kd_occludes(ctx_t *cx, int root, const ray_t *target, float ldistance, target_t skip, isect_list *hits,
                       float t_near, float t_far) {
    float invdir[MAXD];
    for (int i = 0; i < cx->s->n; ++i) invdir[i] = 1.0f / target->direction[i];
    return kd_occludes_rec(cx, root, target, invdir, ldistance, skip, hits, t_near, t_far);
}

/* ---------------- composite_scene shading: tracer.hpp:1678-1918 ---------------- */

static int cmp_isect(const void *a, const void *b) {
    float da = ((const isect_t *)a)->dist, db = ((const isect_t *)b)->dist;
    return da < db ? -1 : (da > db ? 1 : 0);
}
static void sort_and_unique(isect_list *l) {                       /* tracer.hpp:714-721 */
    /* std::sort is not stable; ties in dist between different targets are a
       measure-zero event in the fixtures */
    qsort(l->data, l->size, sizeof(isect_t), cmp_isect);
    size_t w = 0;
    for (size_t i = 0; i < l->size; ++i) {
        if (w == 0 || !(l->data[w - 1].target.item == l->data[i].target.item && l->data[w - 1].target.lane == l->data[i].target.lane)) {
            if (w != i) l->data[w] = l->data[i];
            ++w;
        }
    }
    l->size = w;
}

static color_t ray_color(ctx_t *cx, const ray_t *target, int depth, target_t source);

static int light_reaches(ctx_t *cx, const ray_t *target, float ldistance, target_t skip, color_t *filtered) {   /* :1750-1766 */
    isect_list th = {NULL, 0, 0};
    if (cx->c) cx->c->shadow_rays++;
    if (kd_occludes(cx, cx->s->root, target, ldistance, skip, &th, 0.0f, FLT_MAX)) { free(th.data); return 0; }
    if (th.size) {
        sort_and_unique(&th);
        for (size_t i = th.size; i-- > 0;) {
            float op = material_of(cx->s, th.data[i].target)[6];
            *filtered = cscale(*filtered, 1 - op);
        }
    }
    free(th.data);
    return 1;
}

static void append_specular(int n, color_t *c, float *a, const float *m, color_t light_c, const float *target,
                            const float *normal, const float *light_dir) {           /* :1701-1707 */
    float tmp[MAXD];
    for (int k = 0; k < n; ++k) tmp[k] = light_dir[k] - target[k];
    float len = sqrtf(dotn(n, tmp, tmp));
    for (int k = 0; k < n; ++k) tmp[k] = tmp[k] / len;
    float base = powf(dotn(n, normal, tmp), m[9]) * m[8];
    *c = cadd(*c, cscale(cscale(cmul(col3(m + 3), light_c), base), (1 - *a)));
    *a += base * (1 - *a);
    *c = cscale(*c, *a);
}

static color_t base_color(ctx_t *cx, const ray_t *target, const ray_t *normal, target_t source, int depth) {   /* :1768-1854 */
    const nto_scene *s = cx->s;
    int n = s->n;
    const float *m = material_of(s, source);
    color_t light = col(0, 0, 0), specular = col(0, 0, 0);
    float spec_a = 0;

    for (int li = 0; li < s->n_point_lights; ++li) {
        const float *pos = s->pl_pos + (size_t)li * n;
        color_t plc = col3(s->pl_color + 3 * li);
        float lv[MAXD];
        for (int k = 0; k < n; ++k) lv[k] = normal->origin[k] - pos[k];
        float dist = sqrtf(dotn(n, lv, lv));
        for (int k = 0; k < n; ++k) lv[k] = lv[k] / dist;
        float sine = dotn(n, normal->direction, lv);
        if (sine > 0) {
            float strength = (float)(1 / pow((double)dist, (double)(n - 1)));        /* point_light::strength :1686-1688 */
            if (s->shadows) {
                if (fmaxf(plc.r, fmaxf(plc.g, plc.b)) * strength * sine > LIGHT_THRESHOLD) {
                    color_t filtered = plc;
                    ray_t sr;
                    memcpy(sr.origin, normal->origin, sizeof(float) * n);
                    memcpy(sr.direction, lv, sizeof(float) * n);
                    if (light_reaches(cx, &sr, dist, source, &filtered)) {
                        filtered = cscale(filtered, strength);
                        light = cadd(light, cscale(filtered, sine));
                        if (m[8] != 0.0f) append_specular(n, &specular, &spec_a, m, filtered, target->direction, normal->direction, lv);
                    }
                }
            } else {
                light = cadd(light, cscale(cscale(plc, strength), sine));
            }
        }
    }
    for (int li = 0; li < s->n_global_lights; ++li) {
        const float *gd = s->gl_dir + (size_t)li * n;
        color_t glc = col3(s->gl_color + 3 * li);
        float sine = -dotn(n, normal->direction, gd);
        if (sine > 0) {
            if (s->shadows) {
                color_t filtered = glc;
                ray_t sr;
                float neg[MAXD];
                for (int k = 0; k < n; ++k) neg[k] = -gd[k];
                memcpy(sr.origin, normal->origin, sizeof(float) * n);
                memcpy(sr.direction, neg, sizeof(float) * n);
                if (light_reaches(cx, &sr, FLT_MAX, source, &filtered)) {
                    light = cadd(light, cscale(filtered, sine));
                    if (m[8] != 0.0f) append_specular(n, &specular, &spec_a, m, filtered, target->direction, normal->direction, neg);
                }
            } else {
                light = cadd(light, cscale(glc, sine));
            }
        }
    }

    float sine = -dotn(n, target->direction, normal->direction);
    if (s->camera_light && sine > 0) {
        light = cadd(light, col(sine, sine, sine));
        if (m[8] != 0.0f) {
            float base = powf(sine, m[9]) * m[8];
            specular = cadd(specular, cscale(cscale(col3(m + 3), base), (1 - spec_a)));
            spec_a += base * (1 - spec_a);
            specular = cscale(specular, spec_a);
        }
    }

    color_t r = cadd(col3(s->ambient), cmul(col3(m), light));

    if (m[7] != 0.0f && depth < s->max_reflect_depth) {
        ray_t refl;
        memcpy(refl.origin, normal->origin, sizeof(float) * n);
        float f = -2 * sine;
        for (int k = 0; k < n; ++k) refl.direction[k] = target->direction[k] - normal->direction[k] * f;
        color_t rc = ray_color(cx, &refl, depth + 1, source);
        r = cadd(cscale(cmul(col3(m), rc), m[7]), cscale(r, 1 - m[7]));
    }
    return cadd(specular, cscale(r, 1 - spec_a));
}

/* composite_scene::aabb_distance: tracer.hpp:1892-1918 */
static float aabb_distance(const nto_scene *s, const ray_t *target) {
    int n = s->n;
    for (int i = 0; i < n; ++i) {
        if (target->direction[i] != 0.0f) {
            float o = target->direction[i] > 0 ? s->aabb_start[i] : s->aabb_end[i];
            float dist = (o - target->origin[i]) / target->direction[i];
            int skip = i;
            if (dist < 0) { dist = 0; skip = -1; }
            int miss = 0;
            for (int j = 0; j < n; ++j) {
                if (j != skip) {
                    o = target->direction[j] * dist + target->origin[j];
                    if (o >= s->aabb_end[j] || o <= s->aabb_start[j]) { miss = 1; break; }
                }
            }
            if (!miss) return dist;
        }
    }
    return -1;
}

static color_t ray_color(ctx_t *cx, const ray_t *target, int depth, target_t source) {      /* :1856-1883 */
    const nto_scene *s = cx->s;
    isect_t hit;
    isect_list th = {NULL, 0, 0};
    color_t r;
    memset(&hit.normal, 0, sizeof(hit.normal));
    hit.target.item = -1; hit.target.lane = -1;
    if (cx->c) cx->c->rays++;

    float dist = aabb_distance(s, target);
    hit.dist = FLT_MAX;
    if (cx->c && depth == 0 && dist >= 0) cx->c->aabb_enter++;
    if (dist >= 0 && kd_intersects(cx, s->root, target, source, &hit, &th, dist, FLT_MAX)) {
        if (cx->c && depth == 0) cx->c->hits++;
        r = base_color(cx, target, &hit.normal, hit.target, depth);
    } else {
        float intensity = target->direction[s->bg_gradient_axis];
        r = intensity >= 0 ? cadd(cscale(col3(s->bg1), intensity), cscale(col3(s->bg2), 1 - intensity))
                           : cadd(cscale(col3(s->bg3), -intensity), cscale(col3(s->bg2), 1 + intensity));
    }
    if (th.size) {
        sort_and_unique(&th);
        for (size_t i = th.size; i-- > 0;) {
            float op = material_of(s, th.data[i].target)[6];
            color_t base = base_color(cx, target, &th.data[i].normal, th.data[i].target, depth);
            r = cadd(cscale(base, op), cscale(r, 1 - op));
        }
    }
    free(th.data);
    return r;
}

static color_t composite_color(ctx_t *cx, int w, int h, int x, int y) {                      /* :1885-1890 */
    const nto_scene *s = cx->s;
    ray_t view;
    memcpy(view.origin, s->origin, sizeof(float) * s->n);
    primary_dir(s, w, h, (float)x, (float)y, view.direction);
    target_t none = {-1, -1};
    return ray_color(cx, &view, 0, none);
}

static color_t pixel_color(const nto_scene *s, int w, int h, int x, int y, nto_counters *c) {
    if (!s->is_composite) return box_color(s, w, h, x, y, c);
    ctx_t cx = {s, c};
    return composite_color(&cx, w, h, x, y);
}

void nto_calculate_color(const nto_scene *s, int x, int y, int w, int h, float rgb[3]) {
    color_t c = pixel_color(s, w, h, x, y, NULL);
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
}

void nto_colors_at(const nto_scene *s, int w, int h, int count, const int32_t *xs, const int32_t *ys, float *rgb_out,
                   nto_counters *counters) {
    if (counters) memset(counters, 0, sizeof(*counters));
    for (int i = 0; i < count; ++i) {
        color_t c = pixel_color(s, w, h, xs[i], ys[i], counters);
        rgb_out[3 * i] = c.r; rgb_out[3 * i + 1] = c.g; rgb_out[3 * i + 2] = c.b;
    }
}

/* ---------------- pixel packing: render.cpp:419-462 ---------------- */

void nto_pack_pixel(const float rgb[3], int nchannels, const nto_channel *ch, int reversed, int bytes_per_pixel, uint8_t *out) {
    uint64_t temp[2] = {0, 0};
    int b_offset = 0;
    for (int k = 0; k < nchannels; ++k) {
        /* `f_r*r + f_g*g + f_b*b + f_c` (render.cpp:427) is compiled -ffast-math in the reference; the
           association its build uses, pinned bit-exactly by tests/golden/packing_box3.npz (31-bit and
           30-bit channels), is (f_g*g + f_b*b) + (f_r*r + f_c). */
        float v = (ch[k].f_g * rgb[1] + ch[k].f_b * rgb[2]) + (ch[k].f_r * rgb[0] + ch[k].f_c);
        /* simd::clamp(x,0,1) = min(max(x,0),1) with SSE semantics: NaN -> second operand */
        v = v > 0.0f ? v : 0.0f;
        v = v < 1.0f ? v : 1.0f;
        uint64_t ival;
        if (ch[k].tfloat) {
            uint32_t bits;
            memcpy(&bits, &v, 4);
            ival = bits;
        } else {
            ival = (uint64_t)lround((double)v * (double)(0xffffffffu >> (32 - ch[k].bit_size)));
        }
        int o = b_offset / 64;
        int rm = b_offset % 64;
        int sh = 64 - rm - ch[k].bit_size;
        temp[o] |= sh >= 0 ? ival << sh : ival >> -sh;
        if (rm + ch[k].bit_size > 64) temp[o + 1] = ival << (128 - rm - ch[k].bit_size);
        b_offset += ch[k].bit_size;
    }
    if (reversed) {
        for (int j = bytes_per_pixel - 1; j >= 0; --j) *out++ = (uint8_t)(temp[j / 8] >> ((7 - (j % 8)) * 8));
    } else {
        for (int j = 0; j < bytes_per_pixel; ++j) *out++ = (uint8_t)(temp[j / 8] >> ((7 - (j % 8)) * 8));
    }
}

/* ---------------- frame loop: render.cpp:468-493, 801-909 ---------------- */

typedef struct {
    const nto_scene *s;
    uint8_t *dest;
    int w, h, pitch, nchannels, reversed, bpp;
    const nto_channel *ch;
    unsigned int chunk;          /* atomic */
    int want_counters;
} job_t;

/* blocking_renderer (render.cpp:769-838): `threads` workers that live as long as the renderer and sleep on
   start_cond between frames; the caller draws too and then waits on finish_cond. */
struct nto_renderer {
    pthread_mutex_t mut;
    pthread_cond_t start_cond, finish_cond;
    int nworkers;
    pthread_t *workers;
    nto_counters *counters;      /* [nworkers + 1], the caller's last */
    unsigned int job;            /* frame number, bumped under `mut` (render.cpp:896) */
    unsigned int busy_threads;
    int quit;
    job_t cur;
};

typedef struct { nto_renderer *r; int index; } worker_arg_t;

static void worker_draw(job_t *r, nto_counters *c) {
    int chunks_x = (r->w + RENDER_CHUNK_SIZE - 1) / RENDER_CHUNK_SIZE;
    int chunks_y = (r->h + RENDER_CHUNK_SIZE - 1) / RENDER_CHUNK_SIZE;
    for (;;) {
        int chunk = (int)__atomic_fetch_add(&r->chunk, 1u, __ATOMIC_RELAXED);
        int start_y = chunk / chunks_x;
        int start_x = chunk % chunks_x;
        if (start_y >= chunks_y) break;     /* the reference tests `>` (off by one, harmless): render.cpp:478 */
        start_x *= RENDER_CHUNK_SIZE;
        start_y *= RENDER_CHUNK_SIZE;
        int y1 = start_y + RENDER_CHUNK_SIZE < r->h ? start_y + RENDER_CHUNK_SIZE : r->h;
        int x1 = start_x + RENDER_CHUNK_SIZE < r->w ? start_x + RENDER_CHUNK_SIZE : r->w;
        for (int y = start_y; y < y1; ++y) {
            uint8_t *px = r->dest + (size_t)y * r->pitch + (size_t)start_x * r->bpp;
            for (int x = start_x; x < x1; ++x) {
                color_t cc = pixel_color(r->s, r->w, r->h, x, y, c);
                float rgb[3] = {cc.r, cc.g, cc.b};
                nto_pack_pixel(rgb, r->nchannels, r->ch, r->reversed, r->bpp, px);
                px += r->bpp;
            }
        }
    }
}

/* wait_for_job (render.cpp:790-801); `mut` held */
static int wait_for_job(nto_renderer *r) {
    unsigned int finished;
    do {
        if (r->quit) return 0;
        finished = r->job;
        pthread_cond_wait(&r->start_cond, &r->mut);
    } while (finished == r->job);
    return !r->quit;
}

/* blocking_worker (render.cpp:803-827) */
static void *blocking_worker(void *arg) {
    worker_arg_t *wa = (worker_arg_t *)arg;
    nto_renderer *r = wa->r;
    nto_counters *c = &r->counters[wa->index];
    free(wa);
    pthread_mutex_lock(&r->mut);
    if (!r->busy_threads && !wait_for_job(r)) { pthread_mutex_unlock(&r->mut); return NULL; }
    pthread_mutex_unlock(&r->mut);
    for (;;) {
        worker_draw(&r->cur, r->cur.want_counters ? c : NULL);
        pthread_mutex_lock(&r->mut);
        if (--r->busy_threads == 0) pthread_cond_signal(&r->finish_cond);
        if (!wait_for_job(r)) { pthread_mutex_unlock(&r->mut); return NULL; }
        pthread_mutex_unlock(&r->mut);
    }
}

/* blocking_renderer::blocking_renderer (render.cpp:829-838): threads < 0 => hardware_concurrency() - 1 */
nto_renderer *nto_renderer_create(int threads) {
    if (threads < 0) {
        long hc = sysconf(_SC_NPROCESSORS_ONLN);
        threads = hc > 1 ? (int)hc - 1 : 0;
    }
    nto_renderer *r = (nto_renderer *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    pthread_mutex_init(&r->mut, NULL);
    pthread_cond_init(&r->start_cond, NULL);
    pthread_cond_init(&r->finish_cond, NULL);
    r->counters = (nto_counters *)calloc((size_t)threads + 1, sizeof(nto_counters));
    r->workers = (pthread_t *)calloc((size_t)threads + 1, sizeof(pthread_t));
    for (int i = 0; i < threads; ++i) {
        worker_arg_t *wa = (worker_arg_t *)malloc(sizeof(*wa));
        wa->r = r;
        wa->index = i;
        if (pthread_create(&r->workers[i], NULL, blocking_worker, wa) != 0) { free(wa); break; }
        r->nworkers = i + 1;
    }
    return r;
}

int nto_renderer_threads(const nto_renderer *r) { return r ? r->nworkers + 1 : 0; }

/* blocking_renderer::~blocking_renderer (render.cpp:840-851) */
void nto_renderer_destroy(nto_renderer *r) {
    if (!r) return;
    pthread_mutex_lock(&r->mut);
    r->quit = 1;
    pthread_cond_broadcast(&r->start_cond);
    pthread_mutex_unlock(&r->mut);
    for (int i = 0; i < r->nworkers; ++i) pthread_join(r->workers[i], NULL);
    pthread_mutex_destroy(&r->mut);
    pthread_cond_destroy(&r->start_cond);
    pthread_cond_destroy(&r->finish_cond);
    free(r->counters);
    free(r->workers);
    free(r);
}

/* obj_BlockingRenderer_render (render.cpp:853-909) */
int nto_renderer_render(nto_renderer *r, const nto_scene *s, uint8_t *dest, int w, int h, int pitch, int nchannels,
                        const nto_channel *ch, int reversed, nto_counters *counters) {
    long bits = 0;
    for (int k = 0; k < nchannels; ++k) bits += ch[k].bit_size;
    if (!r || bits > 128 || w < 1 || h < 1) return -1;
    job_t job;
    job.s = s; job.dest = dest; job.w = w; job.h = h; job.nchannels = nchannels; job.reversed = reversed;
    job.ch = ch; job.bpp = (int)((bits + 7) / 8); job.chunk = 0; job.want_counters = counters != NULL;
    job.pitch = pitch ? pitch : w * job.bpp;
    if (job.pitch < w * job.bpp) return -1;
    pthread_mutex_lock(&r->mut);
    if (r->busy_threads) { pthread_mutex_unlock(&r->mut); return -2; }      /* already_running_error */
    if (counters) memset(r->counters, 0, ((size_t)r->nworkers + 1) * sizeof(nto_counters));
    r->cur = job;
    r->busy_threads = (unsigned int)r->nworkers;
    pthread_cond_broadcast(&r->start_cond);
    ++r->job;
    pthread_mutex_unlock(&r->mut);

    worker_draw(&r->cur, counters ? &r->counters[r->nworkers] : NULL);

    pthread_mutex_lock(&r->mut);
    while (r->busy_threads) pthread_cond_wait(&r->finish_cond, &r->mut);
    pthread_mutex_unlock(&r->mut);
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        for (int i = 0; i <= r->nworkers; ++i) {
            const uint64_t *src = (const uint64_t *)&r->counters[i];
            uint64_t *dst = (uint64_t *)counters;
            for (size_t k = 0; k < sizeof(nto_counters) / sizeof(uint64_t); ++k) dst[k] += src[k];
        }
    }
    return 0;
}

/* one renderer for the call: BlockingRenderer(threads).render(dest, format, scene) */
int nto_render(const nto_scene *s, uint8_t *dest, int w, int h, int pitch, int nchannels, const nto_channel *ch,
               int reversed, int threads, nto_counters *counters) {
    if (threads < 0) threads = 0;
    nto_renderer *r = nto_renderer_create(threads);
    if (!r) return -1;
    int rc = nto_renderer_render(r, s, dest, w, h, pitch, nchannels, ch, reversed, counters);
    nto_renderer_destroy(r);
    return rc;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* The RotatingCamera loop of scripts/polytope.py:522-556 around one renderer: frame f is drawn with camera
   (origins[f % ncams], axes[f % ncams]) into the same `dest`; the workers sleep on start_cond between frames.
   Stops after `nframes` frames or once `max_seconds` (> 0) have passed; seconds_out[f] = wall time of frame f.
   Returns the number of frames drawn, or a negative error. */
int nto_renderer_render_frames(nto_renderer *r, const nto_scene *s, uint8_t *dest, int w, int h, int pitch, int nchannels,
                               const nto_channel *ch, int reversed, int nframes, int ncams, const float *origins,
                               const float *axes, double max_seconds, double *seconds_out) {
    if (!r || nframes < 0 || ncams < 1) return -1;
    nto_scene sc = *s;
    const int n = s->n;
    const double t_begin = now_s();
    int f = 0;
    for (; f < nframes; ++f) {
        sc.origin = origins + (size_t)(f % ncams) * n;
        sc.axes = axes + (size_t)(f % ncams) * n * n;
        const double t0 = now_s();
        int rc = nto_renderer_render(r, &sc, dest, w, h, pitch, nchannels, ch, reversed, NULL);
        if (rc) return rc;
        const double t1 = now_s();
        if (seconds_out) seconds_out[f] = t1 - t0;
        if (max_seconds > 0 && t1 - t_begin > max_seconds) { ++f; break; }
    }
    return f;
}

/* ---------------- per-stage entry points (ntracer_body.hpp:1412-1496) ---------------- */

int nto_kd_intersects(const nto_scene *s, const float *origin, const float *direction, float t_near, float t_far,
                      int skip_item, int skip_lane, float *out_dist, int *out_kind, int *out_index, int *out_lane,
                      float *out_normal_origin, float *out_normal_dir, int *n_transparent) {
    ctx_t cx = {s, NULL};
    ray_t target;
    isect_t hit;
    isect_list th = {NULL, 0, 0};
    memcpy(target.origin, origin, sizeof(float) * s->n);
    memcpy(target.direction, direction, sizeof(float) * s->n);
    memset(&hit.normal, 0, sizeof(hit.normal));
    hit.dist = FLT_MAX;
    hit.target.item = -1; hit.target.lane = -1;
    target_t skip = {skip_item, skip_lane};
    int r = kd_intersects(&cx, s->root, &target, skip, &hit, &th, t_near, t_far);
    if (n_transparent) *n_transparent = (int)th.size;
    free(th.data);
    if (r) {
        *out_dist = hit.dist;
        *out_kind = hit.target.item & 3;
        *out_index = hit.target.item >> 2;
        *out_lane = hit.target.lane;
        if (out_normal_origin) memcpy(out_normal_origin, hit.normal.origin, sizeof(float) * s->n);
        if (out_normal_dir) memcpy(out_normal_dir, hit.normal.direction, sizeof(float) * s->n);
    }
    return r;
}

int nto_kd_occludes(const nto_scene *s, const float *origin, const float *direction, float distance, float t_near,
                    float t_far, int skip_item, int skip_lane, int *n_transparent) {
    ctx_t cx = {s, NULL};
    ray_t target;
    isect_list th = {NULL, 0, 0};
    memcpy(target.origin, origin, sizeof(float) * s->n);
    memcpy(target.direction, direction, sizeof(float) * s->n);
    target_t skip = {skip_item, skip_lane};
    int r = kd_occludes(&cx, s->root, &target, distance, skip, &th, t_near, t_far);
    if (n_transparent) *n_transparent = (int)th.size;
    free(th.data);
    return r;
}
