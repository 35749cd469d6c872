/* A plain C99 client of include/ntracer_hip.h: compiled and run by tests/test_abi.py (no GPU needed). */
#include <stdio.h>
#include "ntracer_hip.h"
int main(void) {
    printf("%s devices=%d\n", nt_version(), nt_device_count());
    nt_scene_t *s = nt_box_scene_create(6);
    if (!s) { printf("create failed: %s\n", nt_last_error()); return 1; }
    printf("dimension %d composite %d fov %.2f\n", nt_scene_dimension(s), nt_scene_is_composite(s), nt_scene_get_fov(s));
    float lo[6] = {0,0,0, 5,5,5}, hi[6] = {1,1,0, 6,6,5};
    float tris[2][3][3] = {{{0,0,0},{1,0,0},{0,1,0}}, {{5,5,5},{6,5,5},{5,6,5}}};
    int32_t first[3] = {0, 1, 2};
    nt_kdtree t;
    int r = nt_kdtree_build(3, 2, lo, hi, first, &tris[0][0][0], NULL, &t);
    printf("kdtree: status %d nodes %d items %d\n", r, t.n_nodes, t.n_leaf_items);
    nt_kdtree_free(&t);
    nt_scene_destroy(s);
    return 0;
}
