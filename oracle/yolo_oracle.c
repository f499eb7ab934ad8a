/*
 * CPU oracle, plain C (OpenMP): the "c_embedding-style" restatement of the quantized
 * slim-YOLOv2 path.  TEST INFRASTRUCTURE ONLY -- built by oracle/Makefile into
 * oracle/_build/libyolo_oracle.so and used by tests/ and by bench.py's cpu_baseline leg as the
 * reported CPU baseline ("port"); the product (yolo355) never links or calls it.
 *
 * Parity status: PINNED through oracle/yolo_oracle.py (itself pinned to outputs of the
 * reference, tests/golden/): tests/test_oracle_c.py requires bit-equal int8 feature maps and
 * tolerance-equal detections.  The reference's own C (c_embedding/yolo_forward.c) cannot be
 * compiled here (missing weight.h, hbird SDK, FPGA intrinsics, a redeclaration at :1055/:1058;
 * SURVEY.md 8c), so there is no oracle/_ref.
 *
 * What it follows:
 *   layer schedule + fused activ/pool flags ... c_embedding/yolo_forward.c:1202-1262
 *                                               (= models/slim_yolo_v2.py:212-328)
 *   shift composition ......................... c_embedding/yolo_forward.c:233-257
 *   requantisation = round-half-even, no clamp  models/slim_yolo_v2.py:33-38 (clamp optional)
 *   head channel layout ....................... c_embedding/yolo_forward.c:1271-1273,
 *                                               models/slim_yolo_v2.py:330-341
 *   decode / score / threshold / greedy NMS ... models/slim_yolo_v2.py:111-210
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t cout, cin, e_w, e_b;
    const int8_t *q_w;   /* [cout][cin][3][3] */
    const int32_t *q_b;  /* [cout] */
} yo_layer;

static const int kPool[10] = {1, 1, 0, 1, 0, 1, 0, 0, 0, 0};
static const int kLeaky[10] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 0};

static int64_t rne_shift(int64_t t, int sh) {
    if (sh <= 0) return t * ((int64_t)1 << (-sh));
    return (t + (((int64_t)1 << (sh - 1)) - 1) + ((t >> sh) & 1)) >> sh;
}

/* one output channel `co` of one fused layer on padded int8 planes: in [cin][H+2][W+2] -> plane co of out [cout][Ho+2][Wo+2];
   returns the number of values outside +-127 */
static int64_t conv_plane(const int8_t *in, int cin, int H, int W, const yo_layer *L, int sa_in, int sa_out,
                          int leaky, int pool, int saturate, int8_t *out, int Ho, int Wo, int co) {
    const int F = (sa_in + L->e_w > L->e_b) ? sa_in + L->e_w : L->e_b;
    const int shl = F - sa_in - L->e_w, bshl = F - L->e_b;
    const int sh = F + (leaky ? 3 : 0) - sa_out;
    const int Wp = W + 2;
    int64_t sat_total = 0;
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * W);
    int16_t *q16 = (int16_t *)malloc(sizeof(int16_t) * (size_t)H * W);
    memset(acc, 0, sizeof(int32_t) * (size_t)H * W);
    for (int ci = 0; ci < cin; ++ci) {
        const int8_t *pl = in + (size_t)ci * (H + 2) * Wp;
        const int8_t *wk = L->q_w + ((size_t)co * cin + ci) * 9;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int w = wk[ky * 3 + kx];
                if (!w) continue;
                for (int y = 0; y < H; ++y) {
                    const int8_t *src = pl + (size_t)(y + ky) * Wp + kx;
                    int32_t *a = acc + (size_t)y * W;
                    for (int x = 0; x < W; ++x) a[x] += w * src[x];
                }
            }
    }
    const int64_t bias = (int64_t)L->q_b[co] * ((int64_t)1 << bshl);
    for (int i = 0; i < H * W; ++i) {
        int64_t t = (int64_t)acc[i] * ((int64_t)1 << shl) + bias;
        if (leaky && t >= 0) t *= 8;
        int64_t v = rne_shift(t, sh);
        if (v > 127 || v < -127) {
            ++sat_total;
            if (saturate) v = v > 127 ? 127 : -127;
        }
        q16[i] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    }
    int8_t *op = out + (size_t)co * (Ho + 2) * (Wo + 2);
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
            int v;
            if (pool) {
                const int16_t *p0 = q16 + (size_t)(2 * y) * W + 2 * x, *p1 = p0 + W;
                int a = p0[0] > p0[1] ? p0[0] : p0[1], b = p1[0] > p1[1] ? p1[0] : p1[1];
                v = a > b ? a : b;
            } else {
                v = q16[(size_t)y * W + x];
            }
            op[(size_t)(y + 1) * (Wo + 2) + x + 1] = (int8_t)v;   /* callers use saturate=1 for int8 */
        }
    free(acc);
    free(q16);
    return sat_total;
}

/* x fp32 [B][3][H][W] -> pred int8 [B][PC][H/16][W/16]; sa[11]; nsat[11] accumulates.
   Two levels of OpenMP teams: images across an outer team of min(B, T) threads, each image's output channels across an
   inner team of T / min(B, T) threads (T = omp_get_max_threads()), so that a batch of 64 uses all 256 hardware threads of the
   GPU box's host while an image's planes stay with one team (round 4; one flat loop over image x channel was measured
   slower there, 13.8 against 35.9 images/s: every thread streaming a different image's planes thrashes the caches). */
int yo_backbone(const float *x, int B, int H, int W, const yo_layer *layers, const int32_t *sa, int saturate,
                int8_t *pred, int64_t *nsat) {
    if (H % 16 || W % 16) return -1;
    for (int i = 0; i < 11; ++i) nsat[i] = 0;
    const float s0 = ldexpf(1.0f, sa[0]);
    size_t maxel = 0;
    {
        int h = H, w = W;
        size_t e = (size_t)3 * (h + 2) * (w + 2);
        maxel = e;
        for (int k = 0; k < 10; ++k) {
            const int ho = kPool[k] ? h / 2 : h, wo = kPool[k] ? w / 2 : w;
            e = (size_t)layers[k].cout * (ho + 2) * (wo + 2);
            if (e > maxel) maxel = e;
            h = ho;
            w = wo;
        }
    }
    int64_t ns_all[11] = {0};
    int fail = 0;
    const int T = omp_get_max_threads();
    const int outer = B < T ? B : T, inner = T / outer > 1 ? T / outer : 1;
    omp_set_max_active_levels(2);
#pragma omp parallel for schedule(dynamic, 1) num_threads(outer)
    for (int b = 0; b < B; ++b) {
        int8_t *bufA = (int8_t *)calloc(maxel, 1), *bufB = (int8_t *)calloc(maxel, 1);
        if (!bufA || !bufB) { fail = 1; free(bufA); free(bufB); continue; }
        int64_t ns_img[11] = {0};
        int h = H, w = W;
        for (int c = 0; c < 3; ++c)
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx) {
                    float r = rintf(x[(((size_t)b * 3 + c) * H + y) * W + xx] * s0);   /* RNE (:35) */
                    if (r > 127.f || r < -127.f) {
                        ++ns_img[0];
                        if (saturate) r = r > 0 ? 127.f : -127.f;
                    }
                    bufA[(size_t)c * (H + 2) * (W + 2) + (size_t)(y + 1) * (W + 2) + xx + 1] = (int8_t)r;
                }
        int8_t *in = bufA, *out = bufB;
        int cin = 3;
        for (int k = 0; k < 10; ++k) {
            const int ho = kPool[k] ? h / 2 : h, wo = kPool[k] ? w / 2 : w, cout = layers[k].cout;
            memset(out, 0, (size_t)cout * (ho + 2) * (wo + 2));
            int64_t ns = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : ns) num_threads(inner)
            for (int co = 0; co < cout; ++co)
                ns += conv_plane(in, cin, h, w, &layers[k], sa[k], sa[k + 1], kLeaky[k], kPool[k], saturate, out, ho, wo, co);
            ns_img[k + 1] += ns;
            int8_t *t = in;
            in = out;
            out = t;
            cin = cout;
            h = ho;
            w = wo;
        }
        const int PC = layers[9].cout;
        for (int c = 0; c < PC; ++c)
            for (int y = 0; y < h; ++y)
                for (int xx = 0; xx < w; ++xx)
                    pred[(((size_t)b * PC + c) * h + y) * w + xx] = in[(size_t)c * (h + 2) * (w + 2) + (size_t)(y + 1) * (w + 2) + xx + 1];
        free(bufA);
        free(bufB);
#pragma omp critical
        for (int i = 0; i < 11; ++i) ns_all[i] += ns_img[i];
    }
    for (int i = 0; i < 11; ++i) nsat[i] = ns_all[i];
    return fail ? -2 : 0;
}

typedef struct { float score; int idx; } yo_key;
static int key_cmp(const void *a, const void *b) {
    const yo_key *x = (const yo_key *)a, *y = (const yo_key *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;   /* score desc */
    return x->idx - y->idx;                                          /* anchor index asc */
}

/* pred int8 [B][A*(5+C)][Hs][Ws] -> padded detections (anchor-index order) */
int yo_head_nms(const int8_t *pred, int B, int Hs, int Ws, int A, int C, const float *anchors, int sa_pred,
                float conf_thresh, float nms_thresh, int in_h, int in_w, int max_det, float *boxes, float *scores,
                int32_t *cls, int32_t *count) {
    const int HW = Hs * Ws, N = HW * A, PC = A * (5 + C);
    const float dq = ldexpf(1.0f, -sa_pred);
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float *bx = (float *)malloc(sizeof(float) * 4 * N), *sc = (float *)malloc(sizeof(float) * N);
        int *cl = (int *)malloc(sizeof(int) * N);
        char *keep = (char *)calloc(N, 1);
        yo_key *keys = (yo_key *)malloc(sizeof(yo_key) * N);
        int nk = 0;
        const int8_t *pb = pred + (size_t)b * PC * HW;
        for (int cell = 0; cell < HW; ++cell)
            for (int a = 0; a < A; ++a) {
                const int n = cell * A + a, gy = cell / Ws, gx = cell % Ws;
#define P(ch) ((float)pb[(size_t)(ch) * HW + cell] * dq)
                const float obj = 1.0f / (1.0f + expf(-P(a)));
                float m = -3.0e38f, sum = 0.f, best = -1.f;
                int bc = 0;
                for (int c = 0; c < C; ++c) { const float v = P(A + a * C + c); if (v > m) m = v; }
                for (int c = 0; c < C; ++c) sum += expf(P(A + a * C + c) - m);
                for (int c = 0; c < C; ++c) {
                    const float s = (expf(P(A + a * C + c) - m) / sum) * obj;
                    if (s > best) { best = s; bc = c; }
                }
                const int t0 = A * (1 + C) + a * 4;
                const float cx = (1.0f / (1.0f + expf(-P(t0))) + (float)gx) * 16.0f;
                const float cy = (1.0f / (1.0f + expf(-P(t0 + 1))) + (float)gy) * 16.0f;
                const float bw = (expf(P(t0 + 2)) * anchors[2 * a]) * 16.0f;
                const float bh = (expf(P(t0 + 3)) * anchors[2 * a + 1]) * 16.0f;
#undef P
                float v[4] = {(cx - bw / 2) / (float)in_w, (cy - bh / 2) / (float)in_h,
                              (cx + bw / 2) / (float)in_w, (cy + bh / 2) / (float)in_h};
                for (int k = 0; k < 4; ++k) bx[4 * n + k] = v[k] < 0.f ? 0.f : (v[k] > 1.f ? 1.f : v[k]);
                sc[n] = best;
                cl[n] = bc;
                if (best >= conf_thresh) { keys[nk].score = best; keys[nk].idx = n; ++nk; }
            }
        qsort(keys, nk, sizeof(yo_key), key_cmp);
        char *dead = (char *)calloc(nk, 1);
        for (int i = 0; i < nk; ++i) {
            if (dead[i]) continue;
            const int ni = keys[i].idx;
            keep[ni] = 1;
            const float *bi = bx + 4 * ni;
            const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
            for (int j = i + 1; j < nk; ++j) {
                const int nj = keys[j].idx;
                if (dead[j] || cl[nj] != cl[ni]) continue;
                const float *bj = bx + 4 * nj;
                const float xx1 = bi[0] > bj[0] ? bi[0] : bj[0], yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
                const float xx2 = bi[2] < bj[2] ? bi[2] : bj[2], yy2 = bi[3] < bj[3] ? bi[3] : bj[3];
                float w = xx2 - xx1, h = yy2 - yy1;
                if (!(w > 1e-28f)) w = 1e-28f;
                if (!(h > 1e-28f)) h = 1e-28f;
                const float inter = w * h, aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
                const float ovr = inter / (ai + aj - inter);
                if (!(ovr <= nms_thresh)) dead[j] = 1;
            }
        }
        int out = 0;
        for (int n = 0; n < N; ++n)
            if (keep[n] && out < max_det) {
                memcpy(boxes + ((size_t)b * max_det + out) * 4, bx + 4 * n, sizeof(float) * 4);
                scores[(size_t)b * max_det + out] = sc[n];
                cls[(size_t)b * max_det + out] = cl[n];
                ++out;
            }
        count[b] = out;
        free(bx); free(sc); free(cl); free(keep); free(keys); free(dead);
    }
    return 0;
}
