/* oracle_pool_driver.c -- drives the oracle's renderer pool (nto_renderer: the reference's blocking_renderer restated,
   oracle/ntracer_oracle.c) for the ThreadSanitizer run of tools/sanitize.sh: workers that persist between frames, the chunk
   counter, the condition-variable hand-offs; BoxScene(6) frames of a turning camera, twice over with two pools. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/ntracer_oracle.h"

int main(void) {
    enum { N = 6, W = 192, H = 120 };
    float origin[N] = {0}, axes[N * N] = {0};
    for (int i = 0; i < N; ++i) axes[i * N + i] = 1.0f;
    origin[2] = -9.0f;
    nto_scene sc;
    memset(&sc, 0, sizeof(sc));
    sc.n = N; sc.origin = origin; sc.axes = axes; sc.fov = 0.8f; sc.batch_size = 4; sc.camera_light = 1;
    nto_channel ch[4] = {{1, 0, 0, 0, 8, 0}, {0, 1, 0, 0, 8, 0}, {0, 0, 1, 0, 8, 0}, {0, 0, 0, 0, 8, 0}};
    unsigned char *buf = malloc((size_t)W * H * 4);
    unsigned long sum = 0;
    for (int pool = 0; pool < 2; ++pool) {
        nto_renderer *r = nto_renderer_create(3 + pool);
        if (!r) return 2;
        for (int f = 0; f < 6; ++f) {
            const float a = 0.3f * (float)f;
            axes[2 * N + 2] = cosf(a); axes[2 * N + 0] = sinf(a); axes[0 * N + 0] = cosf(a); axes[0 * N + 2] = -sinf(a);
            for (int k = 0; k < N; ++k) origin[k] = -9.0f * axes[2 * N + k];
            if (nto_renderer_render(r, &sc, buf, W, H, W * 4, 4, ch, 0, NULL) != 0) return 3;
            for (int i = 0; i < W * H * 4; i += 97) sum += buf[i];
        }
        nto_renderer_destroy(r);
    }
    printf("oracle pool: 12 frames, checksum %lu\n", sum);
    free(buf);
    return 0;
}
