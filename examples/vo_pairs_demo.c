/*
 * vo_pairs_demo.c -- a plain C consumer of libvo_mi355.so (include/vo_mi355.h): the reference's per-frame loop
 * (/root/reference/vo.py:63-80: get_matches -> get_pose -> chain the pose) for a file of raw 8-bit gray frames, through
 * the batched sequence entry point.  No Python, no torch, no HIP headers: the host side only needs the C ABI.
 *
 *   vo_pairs_demo FRAMES.raw WIDTH HEIGHT NFRAMES FX FY CX CY [chunk_pairs]
 *
 * FRAMES.raw = NFRAMES frames of WIDTH*HEIGHT bytes, consecutive.  Prints one line per frame pair
 * (status, keypoints, matches, inliers, cheirality count, the 12 numbers of [R|t]) and the chained camera position.
 * Build: make -C examples   (gcc, links libvo_mi355.so by path with an rpath).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vo_mi355.h"

/* cur = cur * inv(T) for a rigid T = [R t; 0 1] (vo.py:80: cur_pose = cur_pose @ inv(transf)) */
static void chain_pose(double cur[16], const double T[16]) {
    double inv[16] = {0}, out[16];
    int i, j, k;
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++) inv[4 * i + j] = T[4 * j + i];
    for (i = 0; i < 3; i++) inv[4 * i + 3] = -(inv[4 * i] * T[3] + inv[4 * i + 1] * T[7] + inv[4 * i + 2] * T[11]);
    inv[15] = 1.0;
    for (i = 0; i < 4; i++)
        for (j = 0; j < 4; j++) {
            double s = 0;
            for (k = 0; k < 4; k++) s += cur[4 * i + k] * inv[4 * k + j];
            out[4 * i + j] = s;
        }
    memcpy(cur, out, sizeof(out));
}

int main(int argc, char** argv) {
    if (argc < 9) {
        fprintf(stderr, "usage: %s FRAMES.raw WIDTH HEIGHT NFRAMES FX FY CX CY [chunk_pairs]\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]);
    const double K[9] = {atof(argv[5]), 0, atof(argv[7]), 0, atof(argv[6]), atof(argv[8]), 0, 0, 1};
    const int chunk = argc > 9 ? atoi(argv[9]) : 8;
    if (w <= 0 || h <= 0 || nf < 2) { fprintf(stderr, "bad size / frame count\n"); return 2; }
    const size_t bytes = (size_t)w * h * nf;
    uint8_t* frames = (uint8_t*)malloc(bytes);
    FILE* fp = fopen(argv[1], "rb");
    if (!frames || !fp || fread(frames, 1, bytes, fp) != bytes) { fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, argv[1]); return 2; }
    fclose(fp);

    vo_orb_params prm;
    vo_default_orb_params(&prm);                     /* cv2.ORB_create(3000) */
    prm.nfeatures = 3000;
    vo_ctx* ctx = NULL;
    int rc = vo_ctx_create(0, w, h, chunk, nf - 1 > chunk ? nf - 1 : chunk, &prm, &ctx);
    if (rc != VO_OK) { fprintf(stderr, "vo_ctx_create failed (%d): %s\n", rc, ctx ? vo_last_error(ctx) : "no context"); return 1; }

    vo_pair_result* res = (vo_pair_result*)calloc((size_t)nf - 1, sizeof(vo_pair_result));
    rc = vo_sequence_run(ctx, frames, nf, w, h, w, K, 0.45, res);     /* ratio 0.45: libs/vo_utils.py:83 */
    if (rc != VO_OK) { fprintf(stderr, "vo_sequence_run failed (%d): %s\n", rc, vo_last_error(ctx)); return 1; }

    double cur[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int i, j;
    for (i = 0; i < nf - 1; i++) {
        const vo_pair_result* r = &res[i];
        printf("pair %d status %d kp %d %d matches %d inliers %d good %d T", i, r->status, r->n_kp1, r->n_kp2, r->n_match, r->n_inlier, r->n_good);
        for (j = 0; j < 12; j++) printf(" %.17g", r->T[j]);
        printf("\n");
        if (r->status == VO_OK) chain_pose(cur, r->T);
    }
    printf("position %.17g %.17g %.17g\n", cur[3], cur[7], cur[11]);
    vo_ctx_destroy(ctx);
    free(res);
    free(frames);
    return 0;
}
