/* The hot path driven from plain C through include/yolo355.h alone (host pointers: y355_forward_host), no Python and no HIP
 * header in the client.  tests/test_c_header.py writes the model blob, runs this program on the GPU box and compares what it
 * wrote with the oracle.  Blob (little-endian): int32 H, W, classes, anchors, batch; float conf, nms; float anchors[2 A];
 * int32 sa[11]; 10 x { int32 cout, cin, e_w, e_b; int8 q_w[cout cin 9]; int32 q_b[cout] }; float x[batch 3 H W].
 * Output: int32 max_det; int32 count[batch]; float boxes[batch max_det 4]; float scores[batch max_det]; int32 cls[batch max_det];
 * int64 saturated, guard. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "yolo355.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, y355_last_error()); return 10; } } while (0)

static int rd(FILE *f, void *dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : 1; }

int main(int argc, char **argv) {
    FILE *f, *o;
    int32_t hdr[5], sa[11], lh[4];
    float thr[2];
    y355_config cfg;
    y355_engine *h = NULL;
    int i, md, B;
    size_t nx;
    float *x, *boxes, *scores;
    int32_t *cls, *count;
    int64_t ctr[2];
    if (argc != 3 || !(f = fopen(argv[1], "rb"))) return 1;
    if (rd(f, hdr, sizeof hdr) || rd(f, thr, sizeof thr)) return 2;
    memset(&cfg, 0, sizeof cfg);
    cfg.height = hdr[0]; cfg.width = hdr[1]; cfg.num_classes = hdr[2]; cfg.num_anchors = hdr[3]; cfg.max_batch = B = hdr[4];
    cfg.conf_thresh = thr[0]; cfg.nms_thresh = thr[1]; cfg.own_stream = 1;
    if (cfg.num_anchors > Y355_MAX_ANCHORS || rd(f, cfg.anchors, sizeof(float) * 2 * (size_t)cfg.num_anchors) || rd(f, sa, sizeof sa)) return 3;
    CHECK(y355_create(&cfg, &h));
    for (i = 0; i < 10; ++i) {
        int8_t *qw;
        int32_t *qb;
        size_t nw;
        if (rd(f, lh, sizeof lh)) return 4;
        nw = (size_t)lh[0] * (size_t)lh[1] * 9;
        qw = (int8_t *)malloc(nw);
        qb = (int32_t *)malloc(sizeof(int32_t) * (size_t)lh[0]);
        if (!qw || !qb || rd(f, qw, nw) || rd(f, qb, sizeof(int32_t) * (size_t)lh[0])) return 5;
        CHECK(y355_load_layer(h, i, qw, qb, lh[0], lh[1], lh[2], lh[3]));
        free(qw);
        free(qb);
    }
    CHECK(y355_set_act_exponents(h, sa));
    nx = (size_t)B * 3 * (size_t)cfg.height * (size_t)cfg.width;
    x = (float *)malloc(sizeof(float) * nx);
    if (!x || rd(f, x, sizeof(float) * nx)) return 6;
    fclose(f);
    md = y355_max_det(h);
    boxes = (float *)malloc(sizeof(float) * 4 * (size_t)B * (size_t)md);
    scores = (float *)malloc(sizeof(float) * (size_t)B * (size_t)md);
    cls = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * (size_t)md);
    count = (int32_t *)malloc(sizeof(int32_t) * (size_t)B);
    if (!boxes || !scores || !cls || !count) return 7;
    CHECK(y355_forward_host(h, x, B, 0, boxes, scores, cls, count));
    CHECK(y355_forward_counters(h, &ctr[0], &ctr[1]));
    if (!(o = fopen(argv[2], "wb"))) return 8;
    fwrite(&md, sizeof md, 1, o);
    fwrite(count, sizeof(int32_t), (size_t)B, o);
    fwrite(boxes, sizeof(float), 4 * (size_t)B * (size_t)md, o);
    fwrite(scores, sizeof(float), (size_t)B * (size_t)md, o);
    fwrite(cls, sizeof(int32_t), (size_t)B * (size_t)md, o);
    fwrite(ctr, sizeof(int64_t), 2, o);
    fclose(o);
    y355_destroy(h);
    printf("ok %d images, max_det %d, first count %d\n", B, md, (int)count[0]);
    return 0;
}
