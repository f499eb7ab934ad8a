// yolo355 -- detection head on the GPU: decode + score + threshold + sort + per-class greedy
// NMS, batched (the reference post-processes image 0 only, on the CPU, in Python).
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
// Three kernels per batch:
//   head_kernel  one workgroup per image: decode, compact candidates in anchor order,
//                bitonic-sort their (score, position) keys in LDS;
//   mask_kernel  all-pairs "j suppresses/is suppressed by i" bit matrix, 64x64 blocks, upper
//                triangle only (the transposed block falls out of the wave ballots);
//   scan_kernel  one wave per image walks candidates in score order holding the "removed"
//                bit-set one word per lane, rows prefetched 16 ranks ahead; then compacts the
//                survivors in anchor-index order into the padded outputs.
#include "y355_common.h"

#define NMS_CAP Y355_NMS_CAP   // max anchors per image handled by this head (416x416: 3380)

struct HeadWork {
    float *cbox;          // [B][CAP][4]  compacted candidates, anchor order
    float *cscore;        // [B][CAP]
    int *ccls;            // [B][CAP]
    int *order;           // [B][CAP]     compact position by descending (score, -pos)
    int *count;           // [B]          candidates per image
    unsigned long long *mask;  // [B][CAP][64]
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(1024) void head_kernel(const HeadParams p, const HeadWork wk) {
    __shared__ unsigned long long keys[NMS_CAP];
    __shared__ int wsum[16];
    __shared__ int total;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = p.A, C = p.C;
    const int N = p.Hs * p.Ws * A;

    float box[4][4], score[4];
    int cls[4];
    bool valid[4];
    int nvalid = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = tid * 4 + u;
        valid[u] = false;
        score[u] = 0.f;
        cls[u] = 0;
        if (n < N) {
            const int cell = n / A, a = n % A;
            const int gy = cell / p.Ws, gx = cell % p.Ws;
            const int8_t *pp = p.pred + ((size_t)(b * p.Hs + gy) * p.Ws + gx) * p.cstride;
            const float conf = (float)pp[a] * p.dq;
            const float obj = sigmoidf_(conf);
            const int8_t *pc = pp + A + a * C;
            float m = -3.0e38f;
            for (int c = 0; c < C; ++c) m = fmaxf(m, (float)pc[c] * p.dq);
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += expf((float)pc[c] * p.dq - m);
            float best = -1.f;
            int bc = 0;
            for (int c = 0; c < C; ++c) {
                const float s = (expf((float)pc[c] * p.dq - m) / sum) * obj;
                if (s > best) { best = s; bc = c; }
            }
            const int8_t *pt = pp + A * (1 + C) + a * 4;
            const float tx = (float)pt[0] * p.dq, tyy = (float)pt[1] * p.dq;
            const float tw = (float)pt[2] * p.dq, th = (float)pt[3] * p.dq;
            const float cx = (sigmoidf_(tx) + (float)gx) * 16.0f;
            const float cy = (sigmoidf_(tyy) + (float)gy) * 16.0f;
            const float bw = (expf(tw) * p.anchors[2 * a]) * 16.0f;
            const float bh = (expf(th) * p.anchors[2 * a + 1]) * 16.0f;
            box[u][0] = fminf(fmaxf((cx - bw / 2) / p.in_w, 0.f), 1.f);
            box[u][1] = fminf(fmaxf((cy - bh / 2) / p.in_h, 0.f), 1.f);
            box[u][2] = fminf(fmaxf((cx + bw / 2) / p.in_w, 0.f), 1.f);
            box[u][3] = fminf(fmaxf((cy + bh / 2) / p.in_h, 0.f), 1.f);
            score[u] = best;
            cls[u] = bc;
            valid[u] = best >= p.conf_thresh;
            nvalid += valid[u] ? 1 : 0;
            if (p.cand_score) {      // full per-anchor tap (parity tests)
                p.cand_score[(size_t)b * N + n] = best;
                p.cand_cls[(size_t)b * N + n] = bc;
#pragma unroll
                for (int k = 0; k < 4; ++k) p.cand_box[((size_t)b * N + n) * 4 + k] = box[u][k];
            }
        }
    }
    // ---- block exclusive scan of nvalid (wave scan + 16 wave totals)
    const int lane = tid & 63, wave = tid >> 6;
    int incl = nvalid;
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
        total = s;
    }
    __syncthreads();
    int pos = wsum[wave] + incl - nvalid;
    const int M = total;
    float *cb = wk.cbox + (size_t)b * NMS_CAP * 4;
    float *cs = wk.cscore + (size_t)b * NMS_CAP;
    int *cc = wk.ccls + (size_t)b * NMS_CAP;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (valid[u]) {
            *(float4 *)(cb + (size_t)pos * 4) = make_float4(box[u][0], box[u][1], box[u][2], box[u][3]);
            cs[pos] = score[u];
            cc[pos] = cls[u];
            keys[pos] = ((unsigned long long)(~__float_as_uint(score[u])) << 32) | (unsigned int)pos;
            ++pos;
        }
    }
    int P2 = 64;
    while (P2 < M) P2 <<= 1;
    __syncthreads();
    for (int i = M + tid; i < P2; i += 1024) keys[i] = ~0ull;
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
    int *ord = wk.order + (size_t)b * NMS_CAP;
    for (int i = tid; i < M; i += 1024) ord[i] = (int)(keys[i] & 0xffffffffu);
    if (tid == 0) wk.count[b] = M;
}

// suppression relation of slim_yolo_v2.py:159-171 between two boxes of the same class
__device__ __forceinline__ bool suppresses(const float4 a, float area_a, const float4 c, float area_c, float thr) {
    const float xx1 = fmaxf(a.x, c.x), yy1 = fmaxf(a.y, c.y);
    const float xx2 = fminf(a.z, c.z), yy2 = fminf(a.w, c.w);
    const float w = fmaxf(1e-28f, xx2 - xx1), h = fmaxf(1e-28f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (area_a + area_c - inter);
    return !(ovr <= thr);
}

__global__ __launch_bounds__(256) void mask_kernel(const HeadWork wk, float thr) {
    __shared__ float4 sbox[4][64];
    __shared__ int scls[4][64];
    const int b = blockIdx.y;
    const int M = wk.count[b];
    const int nblk = (M + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x / (NMS_CAP / 64 / 4);
    const int cb = (blockIdx.x % (NMS_CAP / 64 / 4)) * 4 + wave;
    if (rb >= nblk || cb >= nblk || cb < rb) return;   // whole-wave exit, no barrier below
    const float *cbx = wk.cbox + (size_t)b * NMS_CAP * 4;
    const int *ccl = wk.ccls + (size_t)b * NMS_CAP;
    const int i = rb * 64 + lane, jg = cb * 64 + lane;
    const bool vi = i < M;
    const float4 bi = vi ? *(const float4 *)(cbx + (size_t)i * 4) : make_float4(0, 0, 0, 0);
    const int ci = vi ? ccl[i] : -1;
    const float ai = (bi.z - bi.x) * (bi.w - bi.y);
    sbox[wave][lane] = (jg < M) ? *(const float4 *)(cbx + (size_t)jg * 4) : make_float4(0, 0, 0, 0);
    scls[wave][lane] = (jg < M) ? ccl[jg] : -2;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own-wave LDS writes visible to own-wave reads
    __builtin_amdgcn_wave_barrier();
    unsigned long long rowbits = 0, mycol = 0;
    for (int j = 0; j < 64; ++j) {
        const float4 bj = sbox[wave][j];
        const int cj = scls[wave][j];
        const float aj = (bj.z - bj.x) * (bj.w - bj.y);
        const bool s = (ci == cj) && (i != cb * 64 + j) && suppresses(bi, ai, bj, aj, thr);
        rowbits |= s ? (1ull << j) : 0ull;
        const unsigned long long colbits = __ballot(s);   // bits over i for column j
        if (lane == j) mycol = colbits;
    }
    unsigned long long *mk = wk.mask + (size_t)b * NMS_CAP * 64;
    if (vi) mk[(size_t)i * 64 + cb] = rowbits;
    if (cb != rb && jg < M) mk[(size_t)jg * 64 + rb] = mycol;
}

__global__ __launch_bounds__(64) void scan_kernel(const HeadParams p, const HeadWork wk) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int M = wk.count[b];
    const int nw = (M + 63) >> 6;
    const int *ord = wk.order + (size_t)b * NMS_CAP;
    const unsigned long long *mk = wk.mask + (size_t)b * NMS_CAP * 64;
    constexpr int CH = 16;
    unsigned long long removed = 0, keep = 0;
    unsigned long long rows[CH], nrows[CH];
    int idx[CH], nidx[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        idx[k] = (k < M) ? ord[k] : 0;
        rows[k] = (k < M && lane < nw) ? mk[(size_t)idx[k] * 64 + lane] : 0ull;
    }
    for (int r0 = 0; r0 < M; r0 += CH) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int r = r0 + CH + k;
            nidx[k] = (r < M) ? ord[r] : 0;
            nrows[k] = (r < M && lane < nw) ? mk[(size_t)nidx[k] * 64 + lane] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (r0 + k < M) {
                const int i = __builtin_amdgcn_readfirstlane(idx[k]);
                const int w = i >> 6, bit = i & 63;
                const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)removed, w);
                const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(removed >> 32), w);
                const unsigned long long rw = ((unsigned long long)hi << 32) | lo;
                if (!((rw >> bit) & 1ull)) {
                    if (lane == w) keep |= 1ull << bit;
                    removed |= rows[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) { idx[k] = nidx[k]; rows[k] = nrows[k]; }
    }
    // ---- survivors, in anchor-index (= compact) order
    const int cnt = __popcll(keep);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int tot = __shfl(incl, 63, 64);
    const int excl = incl - cnt;
    const float *cb = wk.cbox + (size_t)b * NMS_CAP * 4;
    const float *cs = wk.cscore + (size_t)b * NMS_CAP;
    const int *cc = wk.ccls + (size_t)b * NMS_CAP;
    float *ob = p.out_box + (size_t)b * p.max_det * 4;
    float *os = p.out_score + (size_t)b * p.max_det;
    int *oc = p.out_cls + (size_t)b * p.max_det;
    for (int w = 0; w < nw; ++w) {
        const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)keep, w);
        const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(keep >> 32), w);
        const unsigned long long bits = ((unsigned long long)hi << 32) | lo;
        const int base = __builtin_amdgcn_readlane(excl, w);
        if ((bits >> lane) & 1ull) {
            const int dst = base + __popcll(bits & ((1ull << lane) - 1ull));
            const int src = w * 64 + lane;
            if (dst < p.max_det) {
                *(float4 *)(ob + (size_t)dst * 4) = *(const float4 *)(cb + (size_t)src * 4);
                os[dst] = cs[src];
                oc[dst] = cc[src];
            }
        }
    }
    if (lane == 0) p.out_count[b] = tot < p.max_det ? tot : p.max_det;
}

void y355_launch_head_nms(const HeadParams &p, int batch, void *cbox, void *cscore, void *ccls, void *order,
                          void *count, void *mask, hipStream_t s, hipEvent_t mid) {
    HeadWork wk;
    wk.cbox = (float *)cbox;
    wk.cscore = (float *)cscore;
    wk.ccls = (int *)ccls;
    wk.order = (int *)order;
    wk.count = (int *)count;
    wk.mask = (unsigned long long *)mask;
    hipLaunchKernelGGL(head_kernel, dim3(batch), dim3(1024), 0, s, p, wk);
    if (mid) hipEventRecord(mid, s);
    hipLaunchKernelGGL(mask_kernel, dim3((NMS_CAP / 64) * (NMS_CAP / 64 / 4), batch), dim3(256), 0, s, wk, p.nms_thresh);
    hipLaunchKernelGGL(scan_kernel, dim3(batch), dim3(64), 0, s, p, wk);
}
