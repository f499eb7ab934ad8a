// yolo355 -- detection head on the GPU: decode + score + threshold + per-class greedy NMS,
// batched (the reference post-processes image 0 only, on the CPU, in Python).
//
// Replaces models/slim_yolo_v2.py:330-358:
//   head split :330-341   (channel layout [A obj | A*C cls | A*4 txtytwth], anchor-major)
//   decode     :111-143   cx=(sig(tx)+gx)*16, w=exp(tw)*aw*16, x1y1x2y2, /[w,h,w,h], clamp
//   score      :348-350   sigmoid(obj) * softmax(cls)
//   postprocess:176-210   argmax class, score >= conf_thresh, per-class NMS, anchor order
//   nms        :145-174   greedy by descending score; suppressed unless iou <= nms_thresh
// and the C head of c_embedding/yolo_forward.c:965-1147 (get_boxes / conf_sort / NMS).
//
// Tie order (undefined in the reference: unstable argsort): (score desc, anchor index asc).
//
// Greedy NMS is a walk over the "i suppresses j" relation in score order.  The relation is
// sparse (a box only interacts with boxes of similar size whose centre is close), so:
//   head_kernel  one workgroup per image: decode, threshold, and counting-sort the candidates
//                into (anchor, centre-bin) order -- bins are the Hs x Ws grid over the CLAMPED
//                box centres -- plus per-anchor extents (max w/h, min/max area);
//   pairs_kernel one thread per candidate walks only the bins a suppressor can sit in
//                (|dcx| < (1-thr)(wi+wj)/2, same in y, area ratio > thr), each unordered pair
//                once, and appends the suppressing pairs to the image's edge list;
//   resolve_kernel one workgroup per (image, class): candidates without conflicts are kept
//                outright; the others are sorted by (score desc, index asc), get CSR lists of
//                their EARLIER conflicting ranks from the edge list, and are settled 64 ranks
//                per step;
//   emit_kernel  survivors in anchor-index order into the padded outputs.
// The pruning is exact: a pair is skipped only when real IoU < 0.999*thr and the union is not
// degenerate, where the fp32 formula of the reference cannot exceed thr.
#include "y355_common.h"
#include <cstdlib>

#define NMS_CAP Y355_NMS_CAP   // max anchors per image handled by this head (416x416: 3380)
#define NBLK (NMS_CAP / 64)
#define MAXA Y355_HEAD_MAXA
#define EDGE_CAP (NMS_CAP * 64)   // edges per image the global list holds (the old bit-matrix footprint / 2)
#define WG_EDGE_CAP 8192         // edges one pairs workgroup buffers in LDS

struct HeadWork {
    float *cbox;          // [B][CAP][4]  compacted candidates, (anchor, bin) order
    float *cscore;        // [B][CAP]
    int *ccls;            // [B][CAP]
    int *corig;           // [B][CAP]     anchor index n = cell*A + a of compact position p
    int *count;           // [B]          candidates per image
    unsigned int *edges;  // [B][EDGE_CAP] suppressing pairs (p << 12) | q with p < q (compact positions)
    int *nedges;          // [B][2]        number of edges; overflow flag (a list did not fit)
    unsigned long long *confl;    // [B][64]       bit per compact position: has a nonzero row
    int *binstart;        // [B][CAP+8]   first compact position of bin (a*HW + by*Ws + bx)
    float *astat;         // [B][MAXA][4] per anchor: wmax, hmax, amin, amax (clamped boxes)
    int *tiny;            // [B][CAP]     positions of candidates with area < AREA_MIN
    int *ntiny;           // [B]
    unsigned long long *keepw;    // [B][64]       resolved survivors among the conflicted
    unsigned int *clsflag;        // [B][8]        bit per class: has at least one suppressing pair
};

#define PRUNE_MARGIN 1.001f
#define PRUNE_EPS 1e-6f
#define AREA_MIN 1e-10f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(1024) void head_kernel(const HeadParams p, const HeadWork wk) {
    __shared__ int hist[NMS_CAP];          // bin counts -> bin starts
    __shared__ int wsum[16];
    __shared__ unsigned int sstat[MAXA][4];
    __shared__ int ntiny_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = p.A, C = p.C;
    const int HW0 = p.lev[0].Hs * p.lev[0].Ws, N0 = HW0 * A;
    const int HW1 = p.nlev > 1 ? p.lev[1].Hs * p.lev[1].Ws : 0;
    const int N = N0 + HW1 * A;
    const int HWb = p.Hb * p.Wb;
    for (int i = tid; i < NMS_CAP; i += 1024) hist[i] = 0;
    if (tid < MAXA) {
        sstat[tid][0] = 0u;                 // wmax
        sstat[tid][1] = 0u;                 // hmax
        sstat[tid][2] = 0x7f7fffffu;        // amin
        sstat[tid][3] = 0u;                 // amax
    }
    if (tid == 0) ntiny_s = 0;
    if (tid < 64) {
        wk.confl[(size_t)b * 64 + tid] = 0ull;
        wk.keepw[(size_t)b * 64 + tid] = 0ull;
    }
    if (tid < 2) wk.nedges[b * 2 + tid] = 0;
    if (tid < 8) wk.clsflag[b * 8 + tid] = 0u;
    __syncthreads();

    float box[4][4], score[4];
    int cls[4], orig[4], key[4], rk[4];
    bool valid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int np = tid * 4 + u;          // (level, anchor, cell) enumeration
        valid[u] = false;
        score[u] = 0.f;
        cls[u] = 0;
        orig[u] = 0;
        key[u] = 0;
        rk[u] = 0;
        if (np < N) {
            const int lv = np >= N0 ? 1 : 0;
            const HeadLevel &L = p.lev[lv];
            const int npl = np - lv * N0, HWl = lv ? HW1 : HW0;
            const int a = npl / HWl, cell = npl % HWl;
            const int n = lv * N0 + cell * A + a;      // the reference's anchor index (:337-341)
            const int kt = lv * A + a;                 // anchor type: own bins and extents
            const int gy = cell / L.Ws, gx = cell % L.Ws;
            const size_t po = ((size_t)(b * L.Hs + gy) * L.Ws + gx) * L.cstride;
            const int8_t *pq = L.pred ? L.pred + po : nullptr;
            const float *pf = L.pred_f + po;
            const float dq = L.dq;
            auto ld = [&](int c) -> float { return pq ? (float)pq[c] * dq : pf[c]; };
            const float conf = ld(a);
            const float obj = sigmoidf_(conf);
            const int c0 = A + a * C;
            float m = -3.0e38f;
            for (int c = 0; c < C; ++c) m = fmaxf(m, ld(c0 + c));
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += expf(ld(c0 + c) - m);
            float best = -1.f;
            int bc = 0;
            for (int c = 0; c < C; ++c) {
                const float s = (expf(ld(c0 + c) - m) / sum) * obj;
                if (s > best) { best = s; bc = c; }
            }
            const int t0 = A * (1 + C) + a * 4;
            const float tx = ld(t0), tyy = ld(t0 + 1), tw = ld(t0 + 2), th = ld(t0 + 3);
            const float cx = (sigmoidf_(tx) + (float)gx) * L.stride;
            const float cy = (sigmoidf_(tyy) + (float)gy) * L.stride;
            const float bw = (expf(tw) * L.anchors[2 * a]) * p.wh_mul;
            const float bh = (expf(th) * L.anchors[2 * a + 1]) * p.wh_mul;
            box[u][0] = fminf(fmaxf((cx - bw / 2) / p.in_w, 0.f), 1.f);
            box[u][1] = fminf(fmaxf((cy - bh / 2) / p.in_h, 0.f), 1.f);
            box[u][2] = fminf(fmaxf((cx + bw / 2) / p.in_w, 0.f), 1.f);
            box[u][3] = fminf(fmaxf((cy + bh / 2) / p.in_h, 0.f), 1.f);
            score[u] = best;
            cls[u] = bc;
            orig[u] = n;
            valid[u] = best >= p.conf_thresh;
            if (p.cand_score) {      // full per-anchor tap (parity tests)
                p.cand_score[(size_t)b * N + n] = best;
                p.cand_cls[(size_t)b * N + n] = bc;
#pragma unroll
                for (int k = 0; k < 4; ++k) p.cand_box[((size_t)b * N + n) * 4 + k] = box[u][k];
            }
            if (valid[u]) {
                // bin of the clamped centre; per-anchor extents of the clamped boxes
                const float w = box[u][2] - box[u][0], h = box[u][3] - box[u][1], ar = w * h;
                const float ccx = 0.5f * (box[u][0] + box[u][2]), ccy = 0.5f * (box[u][1] + box[u][3]);
                const int bx = min(p.Wb - 1, max(0, (int)(ccx * (float)p.Wb)));
                const int by = min(p.Hb - 1, max(0, (int)(ccy * (float)p.Hb)));
                key[u] = kt * HWb + by * p.Wb + bx;
                rk[u] = atomicAdd(&hist[key[u]], 1);
                atomicMax(&sstat[kt][0], __float_as_uint(w));
                atomicMax(&sstat[kt][1], __float_as_uint(h));
                atomicMin(&sstat[kt][2], __float_as_uint(ar));
                atomicMax(&sstat[kt][3], __float_as_uint(ar));
            }
        }
    }
    __syncthreads();
    // ---- exclusive scan of the bin counts: 4 consecutive bins per thread
    const int lane = tid & 63, wave = tid >> 6;
    int c4[4], mine = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c4[u] = hist[tid * 4 + u];
        mine += c4[u];
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int w = 0; w < 16; ++w) { const int t = wsum[w]; wsum[w] = s; s += t; }
        wk.count[b] = s;
    }
    __syncthreads();
    int run = wsum[wave] + incl - mine;
    int *bs = wk.binstart + (size_t)b * (NMS_CAP + 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        hist[tid * 4 + u] = run;
        bs[tid * 4 + u] = run;
        run += c4[u];
    }
    if (tid == 1023) bs[NMS_CAP] = run;
    __syncthreads();
    // ---- scatter into (anchor, bin) order
    float *cb = wk.cbox + (size_t)b * NMS_CAP * 4;
    float *cs = wk.cscore + (size_t)b * NMS_CAP;
    int *cc = wk.ccls + (size_t)b * NMS_CAP;
    int *co = wk.corig + (size_t)b * NMS_CAP;
    int *tl = wk.tiny + (size_t)b * NMS_CAP;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (valid[u]) {
            const int pos = hist[key[u]] + rk[u];
            *(float4 *)(cb + (size_t)pos * 4) = make_float4(box[u][0], box[u][1], box[u][2], box[u][3]);
            cs[pos] = score[u];
            cc[pos] = cls[u];
            co[pos] = orig[u];
            if ((box[u][2] - box[u][0]) * (box[u][3] - box[u][1]) < AREA_MIN) tl[atomicAdd(&ntiny_s, 1)] = pos;
        }
    }
    if (tid < MAXA) {
        float *as = wk.astat + ((size_t)b * MAXA + tid) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) as[k] = __uint_as_float(sstat[tid][k]);
    }
    __syncthreads();
    if (tid == 0) wk.ntiny[b] = ntiny_s;
}

// ---- the reference's suppression test (slim_yolo_v2.py:159-171), same class assumed
__device__ __forceinline__ bool suppresses_exact(const float4 a, float area_a, const float4 c, float area_c, float thr) {
    const float xx1 = fmaxf(a.x, c.x), yy1 = fmaxf(a.y, c.y);
    const float xx2 = fminf(a.z, c.z), yy2 = fminf(a.w, c.w);
    const float w = fmaxf(1e-28f, xx2 - xx1), h = fmaxf(1e-28f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (area_a + area_c - inter);
    return !(ovr <= thr);
}
// same predicate; the correctly rounded division is only issued when the reciprocal estimate
// lands within 8 ulp-ish of the threshold (or the union is degenerate)
__device__ __forceinline__ bool suppresses(const float4 a, float area_a, const float4 c, float area_c, float thr,
                                           float thr_hi, float thr_lo) {
    const float xx1 = fmaxf(a.x, c.x), yy1 = fmaxf(a.y, c.y);
    const float xx2 = fminf(a.z, c.z), yy2 = fminf(a.w, c.w);
    const float w = fmaxf(1e-28f, xx2 - xx1), h = fmaxf(1e-28f, yy2 - yy1);
    const float inter = w * h;
    const float den = area_a + area_c - inter;
    const float q = inter * __builtin_amdgcn_rcpf(den);
    const bool normal = den > 1e-30f && den < 1e30f;
    const bool hi = normal && q > thr_hi, lo = normal && q < thr_lo;
    bool r = hi;
    if (__any(!hi && !lo)) {
        const bool ex = !(inter / den <= thr);
        r = (hi || lo) ? hi : ex;
    }
    return r;
}

// Exact pruning bounds.  For boxes with IoU > thr:  overlap_x > thr*max(w) hence
// |dcx| < (wi+wj)/2 - thr*max(wi,wj) <= (1-thr)*(wi+wj)/2 (same in y), and min(area)/max(area) > thr.
// With a 0.1% margin the fp32 evaluation of the reference formula cannot land above thr either,
// provided the union is not degenerate (area sum >= 1e-10) and thr >= 1e-4.
// grid (CAP/1024, batch), 1024 threads: one thread per candidate i.  Every unordered pair is
// visited once, from its lower compact position: partners q > i inside the window.  The image's
// candidates and bin starts are staged in LDS (97 KB) so the walk never waits on global memory;
// suppressing pairs are buffered in LDS and appended to the image's edge list with one atomic
// per workgroup.
__global__ __launch_bounds__(1024) void pairs_kernel(const HeadParams p, const HeadWork wk, float thr) {
    extern __shared__ __attribute__((aligned(16))) char plds[];
    float4 *sbox = (float4 *)plds;                                   // [CAP]
    int *scls = (int *)(plds + NMS_CAP * 16);                        // [CAP]
    int *sbin = (int *)(plds + NMS_CAP * 20);                        // [CAP + 8]
    unsigned int *sedge = (unsigned int *)(plds + NMS_CAP * 20 + (NMS_CAP + 8) * 4);   // [WG_EDGE_CAP]
    __shared__ unsigned long long sconf[64];
    __shared__ unsigned int sclsf[8];
    __shared__ int nedge_s, gbase_s;
    const int b = blockIdx.y;
    const int M = wk.count[b];
    if ((int)blockIdx.x * 1024 >= M) return;
    const int tid = threadIdx.x;
    const int A = p.A * p.nlev, Ws = p.Wb, Hs = p.Hb, HW = Hs * Ws;   // anchor types, bin grid
    const int N0 = p.lev[0].Hs * p.lev[0].Ws * p.A;
    {
        const float4 *cbx4 = (const float4 *)(wk.cbox + (size_t)b * NMS_CAP * 4);
        const int *ccl = wk.ccls + (size_t)b * NMS_CAP;
        const int *bs = wk.binstart + (size_t)b * (NMS_CAP + 8);
        for (int q = tid; q < M; q += 1024) { sbox[q] = cbx4[q]; scls[q] = ccl[q]; }
        for (int q = tid; q <= A * HW; q += 1024) sbin[q] = bs[q];
        if (tid < 64) sconf[tid] = 0ull;
        if (tid < 8) sclsf[tid] = 0u;
        if (tid == 0) nedge_s = 0;
    }
    __syncthreads();
    const int i = blockIdx.x * 1024 + tid;
    const bool vi = i < M;
    const bool fast = thr >= 1e-4f && thr < 1e4f;
    const float kr = (1.0f - thr) * 0.5f * PRUNE_MARGIN, thr_lo = thr * 0.999f;
    const float q_hi = thr * (1.0f + 8e-6f), q_lo = thr * (1.0f - 8e-6f);
    const float *as = wk.astat + (size_t)b * MAXA * 4;
    const float4 bi = vi ? sbox[i] : make_float4(0, 0, 0, 0);
    const int ci = vi ? scls[i] : -1;
    const float wi = bi.z - bi.x, hi = bi.w - bi.y, ai = wi * hi;
    const float cxi = 0.5f * (bi.x + bi.z), cyi = 0.5f * (bi.y + bi.w);
    bool lost = false;
    auto visit = [&](int q) {
        const float4 bj = sbox[q];
        const int cj = scls[q];
        const float wj = bj.z - bj.x, hj = bj.w - bj.y, aj = wj * hj;
        bool cand = (q > i) && (cj == ci);
        if (fast) {
            const float dx = fabsf(cxi - 0.5f * (bj.x + bj.z)), dy = fabsf(cyi - 0.5f * (bj.y + bj.w));
            const bool far = dx >= kr * (wi + wj) + PRUNE_EPS || dy >= kr * (hi + hj) + PRUNE_EPS ||
                             fminf(ai, aj) <= thr_lo * fmaxf(ai, aj);
            cand = cand && !(far && (ai + aj >= AREA_MIN));
        }
        if (cand) {
            // the reference's predicate (slim_yolo_v2.py:159-171); IEEE division only near the threshold
            const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
            const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
            const float iw = fmaxf(1e-28f, xx2 - xx1), ih = fmaxf(1e-28f, yy2 - yy1);
            const float inter = iw * ih, den = ai + aj - inter;
            const float qv = inter * __builtin_amdgcn_rcpf(den);
            bool s;
            if (fast && den > 1e-30f && den < 1e30f && (qv > q_hi || qv < q_lo)) s = qv > q_hi;
            else s = !(inter / den <= thr);
            if (s) {
                const int e = atomicAdd(&nedge_s, 1);
                if (e < WG_EDGE_CAP) sedge[e] = ((unsigned int)i << 12) | (unsigned int)q;
                else lost = true;
                atomicOr(&sconf[i >> 6], 1ull << (i & 63));
                atomicOr(&sconf[q >> 6], 1ull << (q & 63));
                atomicOr(&sclsf[(ci >> 5) & 7], 1u << (ci & 31));
            }
        }
    };
    if (vi) {
        // own anchor and bin (same formula as head_kernel)
        const int n_i = wk.corig[(size_t)b * NMS_CAP + i];
        const int a_i = n_i < N0 ? n_i % p.A : p.A + (n_i - N0) % p.A;
        const int by_i = min(Hs - 1, max(0, (int)(cyi * (float)Hs)));
        for (int a2 = a_i; a2 < A; ++a2) {              // lower anchors only hold positions < i
            const float wmax = as[a2 * 4 + 0], hmax = as[a2 * 4 + 1], amin = as[a2 * 4 + 2], amax = as[a2 * 4 + 3];
            if (amax < amin) continue;                         // no candidate of this anchor
            int bx0 = 0, bx1 = Ws - 1, by0 = 0, by1 = Hs - 1;
            if (fast && (ai + amin >= AREA_MIN)) {
                if (ai <= thr_lo * amin || amax <= thr_lo * ai) continue;    // area ratio rules the anchor out
                const float rx = kr * (wi + wmax) + PRUNE_EPS, ry = kr * (hi + hmax) + PRUNE_EPS;
                bx0 = max(0, (int)floorf((cxi - rx) * (float)Ws));
                bx1 = min(Ws - 1, (int)floorf((cxi + rx) * (float)Ws));
                by0 = max(0, (int)floorf((cyi - ry) * (float)Hs));
                by1 = min(Hs - 1, (int)floorf((cyi + ry) * (float)Hs));
            }
            if (a2 == a_i) by0 = max(by0, by_i);               // earlier bin rows of my anchor are < i
            for (int by = by0; by <= by1; ++by) {
                const int k0 = a2 * HW + by * Ws;
                const int q0 = max(sbin[k0 + bx0], i + 1), q1 = sbin[k0 + bx1 + 1];
                for (int q = q0; q < q1; ++q) visit(q);
            }
        }
        if (fast && ai < AREA_MIN) {                               // degenerate boxes see each other
            const int nt = wk.ntiny[b];
            const int *tl = wk.tiny + (size_t)b * NMS_CAP;
            for (int t = 0; t < nt; ++t) visit(tl[t]);
        }
    }
    __syncthreads();
    const int ne = min(nedge_s, WG_EDGE_CAP);
    if (tid == 0) {
        gbase_s = atomicAdd(&wk.nedges[b * 2], ne);
        if (nedge_s > WG_EDGE_CAP) wk.nedges[b * 2 + 1] = 1;
    }
    if (lost) wk.nedges[b * 2 + 1] = 1;
    __syncthreads();
    const int gb = gbase_s;
    unsigned int *ge = wk.edges + (size_t)b * EDGE_CAP;
    for (int e = tid; e < ne; e += 1024) {
        if (gb + e < EDGE_CAP) ge[gb + e] = sedge[e];
        else wk.nedges[b * 2 + 1] = 1;
    }
    if (tid < 64 && sconf[tid]) atomicOr(&wk.confl[(size_t)b * 64 + tid], sconf[tid]);
    if (tid < 8 && sclsf[tid]) atomicOr(&wk.clsflag[b * 8 + tid], sclsf[tid]);
}
#define PAIRS_LDS (NMS_CAP * 20 + (NMS_CAP + 8) * 4 + WG_EDGE_CAP * 4)

// ---- resolve_kernel: grid (classes, batch), one workgroup per (image, class).
// Conflicts only exist inside a class, so every class is an independent greedy walk.  The
// class's conflicted candidates are sorted by (score desc, anchor index asc) and every one
// gets the list of its conflicting candidates that come EARLIER in that order (u16 ranks in
// LDS).  The walk takes 64 ranks per step, one per lane: a lane is suppressed if one of its
// earlier neighbours from a previous step was kept (parallel check against the kept bit-set);
// neighbours inside the same step are settled by a scalar find-first-set loop over the few
// lanes that have any.
#define RES_EDGE_CAP 36864          // earlier-neighbour edges kept in LDS (u16 ranks, 72 KiB)

__global__ __launch_bounds__(1024) void resolve_kernel(const HeadWork wk, float thr) {
    __shared__ unsigned long long keys[NMS_CAP];      // gathered, then sorted
    __shared__ unsigned short rank_of[NMS_CAP];
    __shared__ int eoff[NMS_CAP + 1];                 // per rank: start of its earlier-neighbour list
    __shared__ int efill[NMS_CAP];                    // per rank: count, then insertion cursor
    __shared__ unsigned short edges[RES_EDGE_CAP];
    __shared__ unsigned long long keptw[64];          // kept ranks, 64 per word
    __shared__ int wsum[16];
    __shared__ int nconf_s;
    const int b = blockIdx.y, cls = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // nothing of this class conflicts (and no edge list overflowed): everything is kept by emit_kernel
    if (wk.nedges[b * 2 + 1] == 0 && ((wk.clsflag[b * 8 + ((cls >> 5) & 7)] >> (cls & 31)) & 1u) == 0u) return;
    const int M = wk.count[b];
    const float *cs = wk.cscore + (size_t)b * NMS_CAP;
    const int *cc = wk.ccls + (size_t)b * NMS_CAP;
    const int *co = wk.corig + (size_t)b * NMS_CAP;
    const unsigned long long *cf = wk.confl + (size_t)b * 64;
    const int ne_img = min(wk.nedges[b * 2], EDGE_CAP);
    const bool lossy = wk.nedges[b * 2 + 1] != 0;     // some edge list overflowed: brute-force path
    const unsigned int *ge = wk.edges + (size_t)b * EDGE_CAP;
    if (tid == 0) nconf_s = 0;
    if (tid < 64) keptw[tid] = 0ull;
    for (int i = tid; i < NMS_CAP; i += 1024) efill[i] = 0;
    __syncthreads();
    for (int pos = tid; pos < M; pos += 1024) {
        const bool conflicted = lossy || ((cf[pos >> 6] >> (pos & 63)) & 1ull);
        if (conflicted && cc[pos] == cls) {
            const int k = atomicAdd(&nconf_s, 1);
            keys[k] = ((unsigned long long)(~__float_as_uint(cs[pos])) << 32) |
                      ((unsigned long long)(unsigned int)co[pos] << 12) | (unsigned int)pos;
        }
    }
    __syncthreads();
    const int nconf = nconf_s;
    if (nconf == 0) return;
    // ---- sort by (score desc, anchor index asc)
    {
        int P2 = 1;
        while (P2 < nconf) P2 <<= 1;
        for (int i = nconf + tid; i < P2; i += 1024) keys[i] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= P2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P2; i += 1024) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long x = keys[i], y = keys[ixj];
                        const bool asc = (i & k) == 0;
                        if ((x > y) == asc) { keys[i] = y; keys[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        }
    }
    const float *cbx = wk.cbox + (size_t)b * NMS_CAP * 4;
    if (lossy) {
        // ---- fallback (an edge list overflowed): the textbook walk, one kept candidate at a time
        __shared__ unsigned char dead[NMS_CAP];
        for (int r = tid; r < nconf; r += 1024) dead[r] = 0;
        __syncthreads();
        for (int r = 0; r < nconf; ++r) {
            if (dead[r]) continue;                                  // uniform: read after the barrier below
            const int pi = (int)(keys[r] & 0xfffu);
            if (tid == 0) atomicOr(&wk.keepw[(size_t)b * 64 + (pi >> 6)], 1ull << (pi & 63));
            const float4 bi = *(const float4 *)(cbx + (size_t)pi * 4);
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
            for (int r2 = r + 1 + tid; r2 < nconf; r2 += 1024) {
                if (!dead[r2]) {
                    const int pj = (int)(keys[r2] & 0xfffu);
                    const float4 bj = *(const float4 *)(cbx + (size_t)pj * 4);
                    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                    if (suppresses_exact(bi, ai, bj, aj, thr)) dead[r2] = 1;
                }
            }
            __syncthreads();
        }
        return;
    }
    for (int r = tid; r < nconf; r += 1024) rank_of[(int)(keys[r] & 0xfffu)] = (unsigned short)r;
    __syncthreads();
    // ---- per rank: the conflicting candidates that come EARLIER in the order (CSR in LDS)
    for (int e = tid; e < ne_img; e += 1024) {
        const unsigned int pq = ge[e];
        const int pi = (int)(pq >> 12), pj = (int)(pq & 0xfffu);
        if (cc[pi] == cls) {
            const int ri = rank_of[pi], rj = rank_of[pj];
            atomicAdd(&efill[max(ri, rj)], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the counts in rank order (rank = tid + t*1024: scan per t, chained)
    constexpr int RPT = NMS_CAP / 1024;
    int base = 0;
#pragma unroll 1
    for (int t = 0; t < RPT; ++t) {
        const int r = tid + t * 1024;
        const int c = r < nconf ? efill[r] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wb = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int v = wsum[w]; if (w < wave) wb += v; tot += v; }
        if (r < nconf) { eoff[r] = base + wb + incl - c; efill[r] = base + wb + incl - c; }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) eoff[nconf] = base;
    const bool big = base > RES_EDGE_CAP;               // same value in every thread
    if (!big) {
        for (int e = tid; e < ne_img; e += 1024) {
            const unsigned int pq = ge[e];
            const int pi = (int)(pq >> 12), pj = (int)(pq & 0xfffu);
            if (cc[pi] == cls) {
                const int ri = rank_of[pi], rj = rank_of[pj];
                edges[atomicAdd(&efill[max(ri, rj)], 1)] = (unsigned short)min(ri, rj);
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- ordered walk, 64 ranks per step.  sup: an earlier neighbour of an earlier step is kept;
    //      own: earlier neighbours inside this step, settled by the scalar loop
    const int nblk = (nconf + 63) >> 6;
    for (int k = 0; k < nblk; ++k) {
        const int r = k * 64 + lane;
        const bool v = r < nconf;
        const int pos = v ? (int)(keys[r] & 0xfffu) : 0;
        bool sup = false;
        unsigned long long own = 0ull;
        if (!big) {
            const int e0 = v ? eoff[r] : 0, e1 = v ? eoff[r + 1] : 0;
            for (int e = e0; e < e1; e += 4) {          // 4 edges per trip: the LDS reads overlap
                int rq[4];
                unsigned long long kw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rq[u] = edges[min(e + u, RES_EDGE_CAP - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) kw[u] = keptw[(rq[u] >> 6) & 63];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (e + u < e1) {
                        const unsigned long long bit = 1ull << (rq[u] & 63);
                        if ((rq[u] >> 6) < k) sup = sup || (kw[u] & bit) != 0ull;
                        else own |= bit;
                    }
                }
            }
        } else {
            // more earlier-neighbour edges than LDS holds: scan the image's edge list (slow, exact)
            for (int e = 0; e < ne_img; ++e) {
                const unsigned int pq = ge[e];
                const int pi = (int)(pq >> 12), pj = (int)(pq & 0xfffu);
                if (v && (pi == pos || pj == pos)) {
                    const int rq = rank_of[pi == pos ? pj : pi];
                    if (rq < r) {
                        const unsigned long long bit = 1ull << (rq & 63);
                        if ((rq >> 6) < k) sup = sup || (keptw[rq >> 6] & bit) != 0ull;
                        else own |= bit;
                    }
                }
            }
        }
        const unsigned long long alive = __ballot(v && !sup);
        unsigned long long keptk = __ballot(v && !sup && own == 0ull);
        unsigned long long hard = alive & ~keptk;
        while (hard) {
            const int u = __ffsll((long long)hard) - 1;
            hard &= hard - 1;
            const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)own, u);
            const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(own >> 32), u);
            if (((((unsigned long long)hi << 32) | lo) & keptk) == 0ull) keptk |= 1ull << u;
        }
        if (lane == 0) keptw[k] = keptk;
        __builtin_amdgcn_wave_barrier();
        if (v && ((keptk >> lane) & 1ull)) atomicOr(&wk.keepw[(size_t)b * 64 + (pos >> 6)], 1ull << (pos & 63));
    }
}

// ---- emit_kernel: one workgroup per image: survivors = conflict-free candidates + resolved
// ones, written in anchor-index order into the padded outputs.
__global__ __launch_bounds__(1024) void emit_kernel(const HeadParams p, const HeadWork wk) {
    __shared__ unsigned long long keepw[64];        // by compact position
    __shared__ unsigned long long keepn[64];        // by anchor index
    __shared__ int wbase[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int M = wk.count[b];
    const int nw = (M + 63) >> 6;
    const float *cb = wk.cbox + (size_t)b * NMS_CAP * 4;
    const float *cs = wk.cscore + (size_t)b * NMS_CAP;
    const int *cc = wk.ccls + (size_t)b * NMS_CAP;
    const int *co = wk.corig + (size_t)b * NMS_CAP;
    if (tid < 64) {
        const unsigned long long vm = tid < nw ? ((tid == nw - 1 && (M & 63)) ? ((1ull << (M & 63)) - 1ull) : ~0ull) : 0ull;
        const unsigned long long cf = tid < nw ? wk.confl[(size_t)b * 64 + tid] : 0ull;
        keepw[tid] = (vm & ~cf) | (wk.keepw[(size_t)b * 64 + tid] & cf & vm);
        keepn[tid] = 0ull;
    }
    __syncthreads();
    for (int pos = tid; pos < M; pos += 1024) {
        if ((keepw[pos >> 6] >> (pos & 63)) & 1ull) {
            const int n = co[pos];
            atomicOr(&keepn[n >> 6], 1ull << (n & 63));
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int cnt = __popcll(keepn[tid]);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        wbase[tid] = incl - cnt;
        if (tid == 63) p.out_count[b] = incl < p.max_det ? incl : p.max_det;
    }
    __syncthreads();
    float *ob = p.out_box + (size_t)b * p.max_det * 4;
    float *os = p.out_score + (size_t)b * p.max_det;
    int *oc = p.out_cls + (size_t)b * p.max_det;
    for (int pos = tid; pos < M; pos += 1024) {
        if ((keepw[pos >> 6] >> (pos & 63)) & 1ull) {
            const int n = co[pos];
            const unsigned long long bits = keepn[n >> 6];
            const int dst = wbase[n >> 6] + __popcll(bits & ((1ull << (n & 63)) - 1ull));
            if (dst < p.max_det) {
                *(float4 *)(ob + (size_t)dst * 4) = *(const float4 *)(cb + (size_t)pos * 4);
                os[dst] = cs[pos];
                oc[dst] = cc[pos];
            }
        }
    }
}

int y355_prepare_head(void) {
    return (int)hipFuncSetAttribute((const void *)pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PAIRS_LDS);
}

void y355_launch_head_nms(const HeadParams &p, int batch, const y355_head_ws &ws, hipStream_t s, hipEvent_t mid) {
    HeadWork wk;
    wk.cbox = (float *)ws.cbox;
    wk.cscore = (float *)ws.cscore;
    wk.ccls = (int *)ws.ccls;
    wk.corig = (int *)ws.corig;
    wk.count = (int *)ws.count;
    wk.edges = (unsigned int *)ws.mask;
    wk.nedges = (int *)ws.rowvalid;
    wk.confl = (unsigned long long *)ws.confl;
    wk.binstart = (int *)ws.binstart;
    wk.astat = (float *)ws.astat;
    wk.tiny = (int *)ws.tiny;
    wk.ntiny = (int *)ws.ntiny;
    wk.keepw = (unsigned long long *)ws.keepw;
    wk.clsflag = (unsigned int *)ws.rmask;
    hipLaunchKernelGGL(head_kernel, dim3(batch), dim3(1024), 0, s, p, wk);
    if (mid) (void)hipEventRecord(mid, s);
    hipLaunchKernelGGL(pairs_kernel, dim3(NMS_CAP / 1024, batch), dim3(1024), PAIRS_LDS, s, p, wk, p.nms_thresh);
    hipLaunchKernelGGL(resolve_kernel, dim3(p.C, batch), dim3(1024), 0, s, wk, p.nms_thresh);
    hipLaunchKernelGGL(emit_kernel, dim3(batch), dim3(1024), 0, s, p, wk);
}
