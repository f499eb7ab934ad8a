// yolo355 -- conv3_1 -> conv3_2 + pool3 of the q_bf path in ONE launch (models/slim_yolo_v2.py:246-267: conv3_1, a_tracker3_1,
// conv3_2, a_tracker3_2, pool3 are a straight chain with no other consumer; the FPGA's conv_normal calls 3 and 4,
// c_embedding/yolo_forward.c:1214-1225).  Round 5.
//
// convpx.hip runs the two layers as two launches: conv3_1's 64-channel map (44 MB at B = 64) goes out to HBM and comes back
// through an LDS-DMA ring, and each launch pays its own weights-into-registers prologue and drain.  Here a workgroup owns a
// BAND of pooled output rows of one image and alternates two phases per step of 192 pooling windows (3.7 pooled rows at 52 per row):
//   A  conv3_1 for the 2 x rows + 2 rows of its map the step needs (8 new rows per step; a band recomputes one row above and
//      one below itself), straight into an LDS ring of 16 map rows -- int8, requantised, exactly the bytes the unfused path
//      stores; the map never exists in HBM;
//   B  conv3_2 + 2x2 max-pool over those rows (convpx's pooled form: the 4x4 neighbourhood of a window read once and fed to
//      the four conv outputs of the window), stored to HBM.
// Every wave does both phases (no producer / consumer roles: the wave-role microbenchmark of this round says a SIMD gains
// nothing from them) and keeps BOTH layers' weights of its 32-channel block in registers: 10 + 18 fragments = 112 VGPRs.
//
// Layouts this kernel owns on both sides, so they are built for its access patterns:
//   * the map ring is PLANAR: a row is four planes of 16-byte chunks (plane c = channels 16 c .. 16 c + 15 of every pixel),
//     slot of padded column x = x ^ ((x >> 4) & 1).  Phase B's lanes (16 windows = every second pixel, one chunk each) and
//     phase A's (16 consecutive pixels) both touch 16 distinct 16-byte bank groups: no LDS bank conflicts (convpx's pooled
//     layers, whose rows arrive by LDS-DMA as whole 64-byte pixels, measure 0.50);
//   * phase A walks ROWS: a wave takes consecutive 16-pixel groups of a row, so every LDS address is (per-row VGPR) +
//     immediate -- 5 vector adds per row instead of ~70 address instructions per group in convpx's flat walk;
//   * input rows (32-byte pixels) arrive by LDS-DMA into a ring of 12 rows, one step ahead: the rows of step s + 1 are issued
//     behind the barrier that ends phase A of step s and have all of phase B to land; the step boundary waits with a counted
//     vmcnt that leaves phase B's own output stores in flight.
// Epilogues: front.hip's fp32 form on exact integers, FOLD 1 / 2 per layer; the hot passes do not clamp and
// track the LeakyReLU branches' extremes, a cold pass redoes a wave's share of the phase clamped and counts (per layer, and for
// conv3_1 only on the rows the band owns) when a value left [-127, 127].  Integer semantics bit for bit those of convpx.hip.
#include "y355_common.h"
#include <cstring>
#include <type_traits>

namespace {
constexpr float MAGIC = 12582912.0f;                 // 1.5 * 2^23
constexpr float QLO = 12582785.0f, QHI = 12583039.0f;
constexpr int RMID = 16;                             // map ring, rows (power of two: phase B's lanes mask)
constexpr int RIN = 12;                              // input ring, rows (scalar modulo)
constexpr int NGMAX = 7;                             // 16-pixel groups per map row: W <= 112
constexpr int SROWS = 4;                             // pooled rows per step of a band
typedef unsigned int v2u __attribute__((ext_vector_type(2)));     // (a uint2 store into LDS gets an s_waitcnt vmcnt(0) in front of it: the
                                                                  // compiler cannot tell it from an LDS-DMA destination; an ext_vector store does not)

__device__ __forceinline__ void qglds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
__device__ __forceinline__ float qvmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float qvmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float qvmin3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <int B>
__device__ __forceinline__ void qmax_to_byte(unsigned int &w, float a, float b) {
    if constexpr (B == 0)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 1)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 2)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned int qpack4(float a, float b, float c, float d) {
    const unsigned int ab = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x0c0c0400u);
    const unsigned int cd = __builtin_amdgcn_perm(__float_as_uint(d), __float_as_uint(c), 0x04000c0cu);
    return ab | cd;
}
__device__ __forceinline__ void qwait_vmcnt(int n) {              // s_waitcnt needs an immediate; n is wave-uniform
#define QW_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        QW_CASE(0) QW_CASE(1) QW_CASE(2) QW_CASE(3) QW_CASE(4) QW_CASE(5) QW_CASE(6) QW_CASE(7) QW_CASE(8)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef QW_CASE
}
// the fp32 epilogue's constants of one layer (VGPR operands: an SGPR source takes a vector instruction off the fast issue path)
struct Epi {
    float sp, sn, cp, cn;
};
template <int FOLD>
__device__ __forceinline__ Epi make_epi(const Requant &rq) {
    const float s_pos = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ldexpf(1.0f, rq.lk - rq.sh))));
    const float s_neg = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)rq.neg_mul * ldexpf(1.0f, -rq.sh))));
    Epi e;
    e.sp = s_pos;
    e.sn = s_neg;
    e.cp = FOLD == 2 ? MAGIC - MAGIC * s_pos : MAGIC;
    e.cn = FOLD == 2 ? MAGIC - MAGIC * s_neg : MAGIC;
    asm volatile("" : "+v"(e.sp), "+v"(e.sn), "+v"(e.cp), "+v"(e.cn));
    return e;
}
}  // namespace

#ifndef PAIR_ABL
#define PAIR_ABL 0                 // timing ablations (WRONG RESULTS): 1 phase A without MFMAs, 2 phase A without the epilogue's arithmetic,
#endif                             // 4 phase B without MFMAs, 8 phase B without the epilogue's arithmetic, 16 no phase A at all, 32 no phase B at all
#ifndef PAIR_DIAG
#define PAIR_DIAG 0                // 1: s_memrealtime stamps (100 MHz) of thread 0 at the phase boundaries (y355_debug_stamps, layer 2)
#endif

template <int F1, int F2>
__global__ __launch_bounds__(512, 2) void pxpair3_kernel(const PairParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int nstamp = 0;
    auto stamp = [&]() {
#if PAIR_DIAG
        if (p.stamps && tid == 0 && nstamp < 32) p.stamps[(size_t)blockIdx.x * 32 + nstamp++] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    (void)nstamp;
    stamp();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 1, ps = wave >> 1;            // 32-channel block of both layers; pixel / window stream 0 .. 3
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
    const int NG = (W + 15) >> 4;                        // 16-pixel groups per map row
    const int MS = W + 2, PLANE = MS * 16, MPITCH = 4 * PLANE;
    const int PPR = (W + 2 + 31) >> 5, IPITCH = PPR * 1024;
    char *const mid = smem;                              // [RMID][4 planes][MS] 16-byte chunks
    char *const inp = smem + RMID * MPITCH;              // [RIN][PPR] 1 KiB pieces of 32 pixels x 32 bytes

    // absolute padded input rows [ra, rb) of image b -> ring slots row % RIN; piece q = 32 pixels of one row, by wave q % 8
    auto dma_rows = [&](int b, int ra, int rb) {
        const int np = (rb - ra) * PPR;
        int n = 0;
        for (int q = wave; q < np; q += 8) {
            const int rr = q / PPR, pc = q - rr * PPR, row = ra + rr;
            const int px = min(pc * 32 + (lane >> 1), W + 1);
            const int8_t *src = p.in + ((size_t)(b * (H + 2) + row) * (W + 2) + px) * 32 + (lane & 1) * 16;
            qglds16(src, inp + (row % RIN) * IPITCH + pc * 1024);
            ++n;
        }
        return n;
    };

    const int G_ = gridDim.x, Rtot = p.B * Ho;
    const int rbeg = (int)((long long)Rtot * blockIdx.x / G_), rend = (int)((long long)Rtot * (blockIdx.x + 1) / G_);
    unsigned int nsat1 = 0, nsat2 = 0;
    bool w2_pending = true;
    // A band = pooled rows [j0, j1) of image b, walked in steps of SROWS = 4 pooled rows: phase A adds the 8 map rows they need
    // (whole rows, two per wave of a channel block; 10 in a band's first step), phase B takes the band's windows, flat and
    // row-major, in groups of 16 -- a multiple of four groups per step (one to four windows' worth of groups wait for the next
    // step), so that both phases are balanced over the four streams.  At most 12 map rows are live in the ring of 16, at most 12
    // input rows in the ring of 12.
    struct Band { int b, j0, j1; };
    auto band_at = [&](int r0) {
        Band q;
        q.b = r0 / Ho;
        q.j0 = r0 - q.b * Ho;
        q.j1 = min(Ho, q.j0 + (rend - r0));
        return q;
    };
    auto first_rows = [&](const Band &q) {               // input rows of the band's first step: issued one phase B ahead
        const int je = min(q.j0 + SROWS - 1, q.j1), hi = min(2 * je + 2, H + 1) + 1;
        dma_rows(q.b, max(2 * q.j0, 1) - 1, hi);
        return hi;
    };
    Band bd = band_at(rbeg);
    int in_hi = first_rows(bd);                          // input rows of the band below in_hi are in the ring or in flight
    // ---- both layers' weights of this wave's channel block: A fragments, registers for the whole launch
    v4i wf1[5][2], wf2[9][2];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int n = 0; n < 2; ++n) wf1[ks][n] = *(const v4i *)(p.w1 + ((size_t)(cb * 5 + ks) * 2 + n) * 1024 + lane * 16);
    // accumulator register r of n-tile n of lane group g = channel 32 cb + 8 g + 4 n + r (convpx.hip, NTN = 2)
    v4i cin1[2], cin2[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const v4i b1 = *(const v4i *)(p.bias1 + cb * 32 + 8 * g + 4 * n), b2 = *(const v4i *)(p.bias2 + cb * 32 + 8 * g + 4 * n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cin1[n][r] = F1 == 2 ? b1[r] + 0x4B400000 : b1[r];
            cin2[n][r] = F2 == 2 ? b2[r] + 0x4B400000 : b2[r];
        }
    }
    const Epi e1 = make_epi<F1>(p.rq1), e2 = make_epi<F2>(p.rq2);
    const float invWo = 1.0f / (float)Wo;

    // ---- lane constants of phase A.  k-step ks of conv3_1 = taps 2 ks, 2 ks + 1 (lane groups 0-1 / 2-3; tap 9 multiplies zero
    // weights and reads tap 8's bytes), 16 channels per lane: byte offset inside an input row of the lane's operand for pixel li
    // of group 0 (group k adds 512).  Only k-step 1 (taps 2 | 3) reads different rows in the two lane halves.
    int cl[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = min(2 * ks + (g >> 1), 8);
        cl[ks] = (li + tap % 3) * 32 + 16 * (g & 1);
    }
    // the map ring's slot of padded column xp = 16 k + li + 1 is xp ^ ((xp >> 4) & 1): for even / odd k
    //   li <= 14: 16 k + (li + 1) / 16 k + ((li + 1) ^ 1);   li = 15: 16 k + 17 / 16 k + 16
    // byte offset inside a map row of this lane's 8 channels (chunk 2 cb + (g >> 1), half g & 1) of group 0's pixel, k even / odd
    int wc[2];
    wc[0] = (2 * cb + (g >> 1)) * PLANE + 8 * (g & 1) + 16 * (li < 15 ? li + 1 : 17);
    wc[1] = (2 * cb + (g >> 1)) * PLANE + 8 * (g & 1) + 16 * (li < 15 ? ((li + 1) ^ 1) : 16);
    const bool lastok = 16 * (NG - 1) + li < W;          // the row's last group: lanes past the row's end store nothing

    // ---- the ring starts as zeros: the halo columns (slots of padded columns 0 and W + 1) are never written afterwards
    for (int i = tid * 16; i < RMID * MPITCH; i += 512 * 16) *(v4i *)(mid + i) = (v4i){0, 0, 0, 0};

    // conv3_2's 18 fragments are not needed before the first phase B: issued by hand BEHIND the arrival of everything the first
    // phase A needs (the compiler does not see these loads: a wait of its own for conv3_1's weights issued under them would wait
    // for them as well), waited for by hand in front of the first phase B -- the first step's phase A runs while they arrive
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first input rows, conv3_1's weights, the biases
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(wf1[ks][n]));
#pragma unroll
    for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(cin1[n]), "+v"(cin2[n]));
    {
        const int8_t *w2p = p.w2 + (size_t)cb * 9 * 2 * 1024 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 9; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(wf2[ks][n]) : "v"(w2p + (size_t)(ks >> 1) * 4096), "n"(((ks & 1) * 2 + n) * 1024) : "memory");
    }
    for (int r0 = rbeg; r0 < rend;) {
        const int b = bd.b, j0 = bd.j0, j1 = bd.j1;
        const int nsteps = (j1 - j0 + 1 + SROWS - 1) / SROWS;       // the first step takes SROWS - 1 pooled rows: with the row above them 8 map rows, two per stream
        r0 += j1 - j0;
        if (!w2_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the first time: waited for above, conv3_2's weights stay in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // the first rows (and, the first time, the zero fill) have landed
        stamp();
        int8_t *const outb = p.out + (((size_t)b * (Ho + 2) + 1) * (Wo + 2) + 1) * 64 + cb * 32;     // wave-uniform
        int PA = 2 * j0;                                 // map rows (padded) of this band below PA are in the ring
        int wdone = 0;                                   // windows of this band already pooled and stored
        for (int s = 0; s < nsteps; ++s) {
            const int je = min(j0 + SROWS * (s + 1) - 1, j1);
            // ================= phase A: map rows [PA, PB) (padded) into the ring; rows 0 and H + 1 are the map's zero halo
            const int PB = 2 * je + 2;
            if (PA == 0 && tid * 16 < MPITCH) *(v4i *)(mid + tid * 16) = (v4i){0, 0, 0, 0};
            if (PB == H + 2 && tid * 16 < MPITCH) *(v4i *)(mid + ((H + 1) & (RMID - 1)) * MPITCH + tid * 16) = (v4i){0, 0, 0, 0};
            const int pa = max(PA, 1), pb = min(PB, H + 1);
            PA = PB;
            float ymx = MAGIC, ymn = MAGIC;
            // One row = NG items of 16 pixels, software-pipelined with everything static: the operands of item k + 1 are read, and
            // item k - 1 is requantised and stored, under the MFMAs of item k; every LDS address is a per-row VGPR + an immediate.
            // (v1: read -> wait -> 10 MFMAs -> wait -> 32 VALU -> store per item; v2: the same pipeline over a flat item list
            // with run-time addresses, 60 VALU + 25 SALU per item: both ~440 cycles per item and SIMD for 160 of MFMA --
            // profiles/r05_notes.md.)
            auto row_a = [&](int P, auto coldc) {
                constexpr bool COLD = decltype(coldc)::value;
                // conv3_1's output row P - 1 reads padded input rows P - 1, P, P + 1
                const int rb0 = ((P - 1) % RIN) * IPITCH, rb1 = (P % RIN) * IPITCH, rb2 = ((P + 1) % RIN) * IPITCH;
                const int a0 = rb0 + cl[0], a1 = (g < 2 ? rb0 : rb1) + cl[1], a2 = rb1 + cl[2], a3 = rb2 + cl[3], a4 = rb2 + cl[4];
                const int wrow = (P & (RMID - 1)) * MPITCH;
                const int w0 = wrow + wc[0], w1 = wrow + wc[1];
                const bool owned = P >= 2 * j0 + 1 && P < 2 * j1 + 1;      // the rows above / below belong to the neighbouring bands
                auto rd = [&](v4i (&bq)[5], int k) {
                    bq[0] = *(const v4i *)(inp + a0 + k * 512);
                    bq[1] = *(const v4i *)(inp + a1 + k * 512);
                    bq[2] = *(const v4i *)(inp + a2 + k * 512);
                    bq[3] = *(const v4i *)(inp + a3 + k * 512);
                    bq[4] = *(const v4i *)(inp + a4 + k * 512);
                };
                // outputs (n, 2 rr) and (n, 2 rr + 1) of an item: the LeakyReLU branches, their extremes, the bytes
                auto out2 = [&](const v4i (&acc)[2], unsigned int (&word)[2], int n_, int rr, bool cok) {
                    float pos[2], neg[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int v = acc[n_][2 * rr + u];
                        const float tf = F1 == 2 ? __int_as_float(v) : (float)v;
                        pos[u] = fmaf(tf, e1.sp, e1.cp);
                        neg[u] = fmaf(tf, e1.sn, e1.cn);
                    }
                    if constexpr (!COLD) {
                        ymx = qvmax3(ymx, pos[0], pos[1]);
                        ymn = qvmin3(ymn, neg[0], neg[1]);
                        if (rr == 0) {
                            qmax_to_byte<0>(word[n_], pos[0], neg[0]);
                            qmax_to_byte<1>(word[n_], pos[1], neg[1]);
                        } else {
                            qmax_to_byte<2>(word[n_], pos[0], neg[0]);
                            qmax_to_byte<3>(word[n_], pos[1], neg[1]);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float y = qvmax(pos[u], neg[u]), yc = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                            nsat1 += (cok && y != yc) ? 1u : 0u;
                            const unsigned int by = __float_as_uint(yc) & 0xffu;
                            word[n_] = (rr == 0 && u == 0) ? by : (word[n_] | (by << (8 * (2 * rr + u))));
                        }
                    }
                };
                auto wr = [&](int k, const unsigned int (&word)[2]) {        // item k's 8 channels -> the ring
                    *(v2u *)(mid + ((k & 1) ? w1 : w0) + k * 256) = (v2u){word[0], word[1]};
                };
                // the MFMAs of item k (bq -> acc); in their shadow (k > 0) the epilogue of item k - 1 (pacc), which is never the
                // row's last item here, so all its lanes are real pixels
                auto stage = [&](int k, const v4i (&bq)[5], v4i (&acc)[2], const v4i (&pacc)[2]) {
                    acc[0] = cin1[0];
                    acc[1] = cin1[1];
                    unsigned int word[2] = {0u, 0u};
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
                        for (int n_ = 0; n_ < 2; ++n_) acc[n_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf1[ks][n_], bq[ks], acc[n_], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (k > 0 && ks < 4) {
                            out2(pacc, word, ks >> 1, ks & 1, owned);                // two MFMAs, two outputs of the item before
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (k > 0) wr(k - 1, word);
                };
                v4i bq[2][5], acc[2][2];
                if constexpr (COLD) {                                             // rare: one item at a time
#pragma unroll
                    for (int k = 0; k < NGMAX; ++k) {
                        if (k >= NG) break;
                        rd(bq[0], k);
                        stage(0, bq[0], acc[0], acc[0]);
                        unsigned int word[2] = {0u, 0u};
                        const bool cok = owned && (k + 1 < NG || lastok);
#pragma unroll
                        for (int q = 0; q < 4; ++q) out2(acc[0], word, q >> 1, q & 1, cok);
                        if (k + 1 < NG || lastok) wr(k, word);
                    }
                    return;
                }
                rd(bq[0], 0);
#pragma unroll
                for (int k = 0; k < NGMAX; ++k) {
                    if (k >= NG) break;                                            // wave-uniform
                    if (k + 1 < NGMAX && k + 1 < NG) rd(bq[(k + 1) & 1], k + 1);
                    stage(k, bq[k & 1], acc[k & 1], acc[(k + 1) & 1]);
                }
                {   // the row's last item: lanes past the row's end store nothing
                    unsigned int word[2] = {0u, 0u};
                    if ((NG - 1) & 1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) out2(acc[1], word, q >> 1, q & 1, false);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) out2(acc[0], word, q >> 1, q & 1, false);
                    }
                    if (lastok) *(v2u *)(mid + (((NG - 1) & 1) ? w1 : w0) + (NG - 1) * 256) = (v2u){word[0], word[1]};
                }
            };
            if (!(PAIR_ABL & 16))
                for (int P = pa + ps; P < pb; P += 4) row_a(P, std::false_type{});
            if (__builtin_amdgcn_ballot_w64(ymx > QHI || ymn < QLO) != 0ull)          // cold: the same rows, clamped and counted
                for (int P = pa + ps; P < pb; P += 4) row_a(P, std::true_type{});
            stamp();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stamp();
            // ---- the next step's new input rows (or the next band's first ones): in flight under phase B -- phase A is done
            // with every row but the last two, the next band's rows are another image's or further down
            if (s + 1 < nsteps) {
                const int hi = min(2 * min(j0 + SROWS * (s + 2) - 1, j1) + 2, H + 1) + 1;
                dma_rows(b, in_hi, hi);
                in_hi = max(in_hi, hi);
            } else if (r0 < rend) {
                bd = band_at(r0);
                in_hi = first_rows(bd);
            }
            // the band's windows that have all their map rows and are not done: a multiple of four groups of 16 now, the rest
            // (less than a row) with the next step; the band's last step takes what is left
            // (narrow maps take everything at once: there 63 waiting windows would be more than the two rows the ring has room for)
            const int avail = (je - j0) * Wo - wdone;
            const bool all = s + 1 == nsteps || Wo < 32;
            const int ngb = all ? (avail + 15) >> 4 : ((avail >> 4) & ~3);
            const int nwin = all ? avail : ngb * 16, wlo = wdone;
            wdone += nwin;
            // ================= phase B: conv3_2 + pool over these windows, groups of 16, stream ps takes every 4th
            float zmx = MAGIC, zmn = MAGIC;
            auto locate = [&](int grp, int &oyr, int &ox) {
                const int wi = wlo + min(grp * 16 + li, nwin - 1);     // padding lanes of the band's last group repeat its last window
                oyr = (int)(((float)wi + 0.5f) * invWo);               // wi / Wo (exact: wi < 2^16); row relative to the band
                ox = wi - oyr * Wo;
            };
            auto issue = [&](int grp, v4i (&acc)[4][2]) {
                int oyr, ox;
                locate(grp, oyr, ox);
                const int ar = 2 * (j0 + oyr), x0 = 2 * ox;            // padded map row / column of the neighbourhood's corner
                int xoff[4], roff[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int x = x0 + c;
                    xoff[c] = g * PLANE + ((x ^ ((x >> 4) & 1)) << 4);
                    roff[c] = ((ar + c) & (RMID - 1)) * MPITCH;
                }
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[v][n] = cin2[n];
                // neighbourhood row r feeds conv output (dy, dx) with filter tap (r - dy, c - dx); order pinned as in convpx.hip:
                // rows 0 and 1 are read, then row r's MFMAs run over the reads of row r + 2
                v4i bq[4][4];
                auto rd = [&](int r) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) bq[r][c] = *(const v4i *)(mid + roff[r] + xoff[c]);
                };
                auto mm = [&](int r) {
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int ty = r - dy;
                        if (ty < 0 || ty > 2) continue;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const int tx = c - dx;
                                if (tx < 0 || tx > 2) continue;
#pragma unroll
                                for (int n = 0; n < 2; ++n) {
                                    if (PAIR_ABL & 4) acc[2 * dy + dx][n] = acc[2 * dy + dx][n] + bq[r][c];
                                    else acc[2 * dy + dx][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf2[ty * 3 + tx][n], bq[r][c], acc[2 * dy + dx][n], 0, 0, 0);
                                }
                            }
                    }
                };
                rd(0);
                rd(1);
                mm(0);
                rd(2);
                mm(1);
                rd(3);
                mm(2);
                mm(3);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);
            };
            auto finish = [&](int grp, const v4i (&acc)[4][2], auto coldc) {
                constexpr bool COLD = decltype(coldc)::value;
                int oyr, ox;
                locate(grp, oyr, ox);
                unsigned int word[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (PAIR_ABL & 8) {
                        word[n] = (unsigned int)(acc[0][n][0] ^ acc[1][n][1] ^ acc[2][n][2] ^ acc[3][n][3]);
                        continue;
                    }
                    float pos[4], neg[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = max(max(acc[0][n][r], acc[1][n][r]), max(acc[2][n][r], acc[3][n][r]));
                        const float tf = F2 == 2 ? __int_as_float(m) : (float)m;
                        pos[r] = fmaf(tf, e2.sp, e2.cp);
                        neg[r] = fmaf(tf, e2.sn, e2.cn);
                    }
                    if constexpr (!COLD) {
                        zmx = qvmax3(qvmax3(zmx, pos[0], pos[1]), pos[2], pos[3]);
                        zmn = qvmin3(qvmin3(zmn, neg[0], neg[1]), neg[2], neg[3]);
                        qmax_to_byte<0>(word[n], pos[0], neg[0]);
                        qmax_to_byte<1>(word[n], pos[1], neg[1]);
                        qmax_to_byte<2>(word[n], pos[2], neg[2]);
                        qmax_to_byte<3>(word[n], pos[3], neg[3]);
                    } else {
                        float yc[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float y = qvmax(pos[r], neg[r]);
                            yc[r] = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                            nsat2 += (grp * 16 + li < nwin && y != yc[r]) ? 1u : 0u;
                        }
                        word[n] = qpack4(yc[0], yc[1], yc[2], yc[3]);
                    }
                }
                // unconditional: the padding lanes rewrite the band's last window with the same bytes, so the number of stores a wave
                // has in flight is a function of its group count (the counted wait below)
                int8_t *dst = outb + (((j0 + oyr) * (Wo + 2) + ox) * 64 + 8 * g);
                *(v2u *)dst = (v2u){word[0], word[1]};
            };
            if (w2_pending) {                                // the launch's first phase B: conv3_2's weights (and, this once, the rows just issued)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < 9; ++ks)
#pragma unroll
                    for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(wf2[ks][n]));     // their values exist from here on
                w2_pending = false;
            }
            int nstores = 0;
            {
                v4i acc[4][2];
#pragma unroll 1
                for (int grp = ps; grp < ((PAIR_ABL & 32) ? 0 : ngb); grp += 4) {
                    issue(grp, acc);
                    finish(grp, acc, std::false_type{});
                    ++nstores;
                }
            }
            if (__builtin_amdgcn_ballot_w64(zmx > QHI || zmn < QLO) != 0ull) {   // cold: the rows are still in the ring
                v4i accC[4][2];
#pragma unroll 1
                for (int grp = ps; grp < ngb; grp += 4) {
                    issue(grp, accC);
                    finish(grp, accC, std::true_type{});
                }
                nstores = -1;
            }
            // the next step's input rows were issued BEFORE this phase's stores: a counted wait leaves the stores in flight
            stamp();
            qwait_vmcnt(nstores);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stamp();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    if (nsat1) atomicAdd(&p.ctr1->sat, (unsigned long long)nsat1);
    if (nsat2) atomicAdd(&p.ctr2->sat, (unsigned long long)nsat2);
}

// ==========================================================================================
// The same pair with the two layers on DIFFERENT WAVES of every SIMD (variant 1 of y355_launch_pair3).
//
// The homogeneous kernel above gives every wave both layers' weights of a 32-channel block (112 VGPRs) and alternates the
// two phases between barriers; its stamps say: conv3_1 (a third of the MACs) is half of the time -- 10 MFMAs and 8 output bytes
// per item make it instruction-bound -- and each phase ends in a tail in which the SIMD's second wave runs alone.  Here
//   * waves 0-3 (the older wave of every SIMD) ONLY run conv3_1, one map row per wave and interval, and hold all of its 64
//     output channels (20 fragments): 20 MFMAs and 16 output bytes per item, half the items, one 16-byte LDS store per lane;
//   * waves 4-7 ONLY run conv3_2 + pool (two 32-channel blocks x two window streams, 18 fragments each), one interval behind;
//   * one barrier per interval of two pooled rows (four map rows) instead of two per step of four: while the conv3_2 wave of a
//     SIMD runs its 72-MFMA bursts the conv3_1 wave has the vector issue slots, and the matrix pipe is never idle because one
//     role is between items.
// Interval t: waves 0-3 issue the input rows of step t + 1 (LDS-DMA), compute the four map rows of step t into the ring and wait
// for their DMA; waves 4-7 pool the windows whose rows step t - 1 completed (a multiple of two groups per channel block, the
// rest waits).  Live in the ring of 16 map rows: 8 being read + 4 being written; in the input ring of 12: 6 + 4 arriving.
// A band ends with one interval in which only waves 4-7 work (the next band's first input rows arrive meanwhile).
template <int F1, int F2>
__global__ __launch_bounds__(512, 2) void pxpair3r_kernel(const PairParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int nstamp = 0;
    auto stamp = [&]() {
#if PAIR_DIAG
        if (p.stamps && (tid == 0 || tid == 256) && nstamp < 16) p.stamps[(size_t)blockIdx.x * 32 + (tid >> 8) * 16 + nstamp++] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    (void)nstamp;
    stamp();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool roleA = wave < 4;
    const int aw = wave & 3;                             // role A: row of the step; role B: (cb, st)
    const int cb = aw & 1, st = aw >> 1;
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
    const int NG = (W + 15) >> 4;
    const int MS = W + 2, PLANE = MS * 16, MPITCH = 4 * PLANE;
    const int PPR = (W + 2 + 31) >> 5, IPITCH = PPR * 1024;
    char *const mid = smem;
    char *const inp = smem + RMID * MPITCH;
    auto dma_rows = [&](int b, int ra, int rb) {          // role A only: piece q by wave q % 4
        const int np = (rb - ra) * PPR;
        for (int q = aw; q < np; q += 4) {
            const int rr = q / PPR, pc = q - rr * PPR, row = ra + rr;
            const int px = min(pc * 32 + (lane >> 1), W + 1);
            const int8_t *src = p.in + ((size_t)(b * (H + 2) + row) * (W + 2) + px) * 32 + (lane & 1) * 16;
            qglds16(src, inp + (row % RIN) * IPITCH + pc * 1024);
        }
    };
    const int G_ = gridDim.x, Rtot = p.B * Ho;
    const int rbeg = (int)((long long)Rtot * blockIdx.x / G_), rend = (int)((long long)Rtot * (blockIdx.x + 1) / G_);
    struct Band { int b, j0, j1; };
    auto band_at = [&](int r0) {
        Band q;
        q.b = r0 / Ho;
        q.j0 = r0 - q.b * Ho;
        q.j1 = min(Ho, q.j0 + (rend - r0));
        return q;
    };
    // step t of a band: pooled rows below je(t) are complete behind it; the first step takes one pooled row (four map rows with
    // the row above it), every other one two
    auto je_of = [&](const Band &q, int t) { return min(q.j0 + 1 + 2 * t, q.j1); };
    auto first_rows = [&](const Band &q) {                // input rows of steps 0 and (nothing else): map rows [2 j0, 2 je0 + 2)
        const int hi = min(2 * je_of(q, 0) + 2, H + 1) + 1;
        dma_rows(q.b, max(2 * q.j0, 1) - 1, hi);
        return hi;
    };
    Band bd = band_at(rbeg);
    int in_hi = 0;
    if (roleA) in_hi = first_rows(bd);
    unsigned int nsat = 0;

    for (int i = tid * 16; i < RMID * MPITCH; i += 512 * 16) *(v4i *)(mid + i) = (v4i){0, 0, 0, 0};
#ifdef PAIR_PRIO
    if ((PAIR_PRIO == 1) != roleA) __builtin_amdgcn_s_setprio(1);        // experiment: 1 = the conv3_2 waves, 2 = the conv3_1 waves at priority 1
#endif

    if (roleA) {
        // ---- conv3_1's weights: all 64 channels (convpx.hip's PX_C3_1 packing), 20 fragments
        v4i wfa[5][4], cina[4];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n) wfa[ks][n] = *(const v4i *)(p.w1 + ((size_t)ks * 4 + n) * 1024 + lane * 16);
#pragma unroll
        for (int n = 0; n < 4; ++n) {                     // accumulator register r of n-tile n of lane group g = channel 16 g + 4 n + r
            const v4i b1 = *(const v4i *)(p.bias1 + 16 * g + 4 * n);
#pragma unroll
            for (int r = 0; r < 4; ++r) cina[n][r] = F1 == 2 ? b1[r] + 0x4B400000 : b1[r];
        }
        const Epi e1 = make_epi<F1>(p.rq1);
        // lane constants (see the kernel above; here a lane stores its pixel's whole 16-byte chunk g)
        int cl[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = min(2 * ks + (g >> 1), 8);
            cl[ks] = (li + tap % 3) * 32 + 16 * (g & 1);
        }
        int wc[2];
        wc[0] = g * PLANE + 16 * (li < 15 ? li + 1 : 17);
        wc[1] = g * PLANE + 16 * (li < 15 ? ((li + 1) ^ 1) : 16);
        const bool lastok = 16 * (NG - 1) + li < W;
        for (int r0 = rbeg; r0 < rend;) {
            const int b = bd.b, j0 = bd.j0, j1 = bd.j1;
            const int nA = 1 + (j1 - j0 - 1 + 1) / 2;         // steps: 1 pooled row, then 2 each
            r0 += j1 - j0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // the band's first input rows (and, the first time, the zero fill) have landed
            stamp();
            for (int t = 0; t <= nA; ++t) {
                if (t < nA) {
                    // ---- the input rows of step t + 1, then the map rows [PA, PB) of step t: wave aw takes row PA + aw
                    const int PA = t == 0 ? 2 * j0 : 2 * je_of(bd, t - 1) + 2, PB = 2 * je_of(bd, t) + 2;
                    if (t + 1 < nA) {
                        const int hi = min(2 * je_of(bd, t + 1) + 2, H + 1) + 1;
                        dma_rows(b, in_hi, hi);
                        in_hi = max(in_hi, hi);
                    }
                    const int P = PA + aw;
                    if (P < PB && (P == 0 || P == H + 1)) {                        // the map's zero halo rows
                        for (int o = lane * 16; o < MPITCH; o += 1024) *(v4i *)(mid + (P & (RMID - 1)) * MPITCH + o) = (v4i){0, 0, 0, 0};
                    } else if (P < PB) {
                        float ymx = MAGIC, ymn = MAGIC;
                        auto row_a = [&](auto coldc) {
                            constexpr bool COLD = decltype(coldc)::value;
                            const int rb0 = ((P - 1) % RIN) * IPITCH, rb1 = (P % RIN) * IPITCH, rb2 = ((P + 1) % RIN) * IPITCH;
                            const int a0 = rb0 + cl[0], a1 = (g < 2 ? rb0 : rb1) + cl[1], a2 = rb1 + cl[2], a3 = rb2 + cl[3], a4 = rb2 + cl[4];
                            const int wrow = (P & (RMID - 1)) * MPITCH;
                            const int w0 = wrow + wc[0], w1 = wrow + wc[1];
                            const bool owned = P >= 2 * j0 + 1 && P < 2 * j1 + 1;
                            auto rd = [&](v4i (&bq)[5], int k) {
                                bq[0] = *(const v4i *)(inp + a0 + k * 512);
                                bq[1] = *(const v4i *)(inp + a1 + k * 512);
                                bq[2] = *(const v4i *)(inp + a2 + k * 512);
                                bq[3] = *(const v4i *)(inp + a3 + k * 512);
                                bq[4] = *(const v4i *)(inp + a4 + k * 512);
                            };
                            auto out2 = [&](const v4i (&acc)[4], v4i &word, int n_, int rr, bool cok) {
                                float pos[2], neg[2];
    #pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const int v = acc[n_][2 * rr + u];
                                    const float tf = F1 == 2 ? __int_as_float(v) : (float)v;
                                    pos[u] = fmaf(tf, e1.sp, e1.cp);
                                    neg[u] = fmaf(tf, e1.sn, e1.cn);
                                }
                                unsigned int w = (unsigned int)word[n_];
                                if constexpr (!COLD) {
                                    ymx = qvmax3(ymx, pos[0], pos[1]);
                                    ymn = qvmin3(ymn, neg[0], neg[1]);
                                    if (rr == 0) {
                                        qmax_to_byte<0>(w, pos[0], neg[0]);
                                        qmax_to_byte<1>(w, pos[1], neg[1]);
                                    } else {
                                        qmax_to_byte<2>(w, pos[0], neg[0]);
                                        qmax_to_byte<3>(w, pos[1], neg[1]);
                                    }
                                } else {
    #pragma unroll
                                    for (int u = 0; u < 2; ++u) {
                                        const float y = qvmax(pos[u], neg[u]), yc = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                                        nsat += (cok && y != yc) ? 1u : 0u;
                                        const unsigned int by = __float_as_uint(yc) & 0xffu;
                                        w = (rr == 0 && u == 0) ? by : (w | (by << (8 * (2 * rr + u))));
                                    }
                                }
                                word[n_] = (int)w;
                            };
                            auto wr = [&](int k, const v4i &word) { *(v4i *)(mid + ((k & 1) ? w1 : w0) + k * 256) = word; };
                            auto stage = [&](int k, const v4i (&bq)[5], v4i (&acc)[4], const v4i (&pacc)[4]) {
    #pragma unroll
                                for (int n_ = 0; n_ < 4; ++n_) acc[n_] = cina[n_];
                                v4i word = {0, 0, 0, 0};
    #pragma unroll
                                for (int ks = 0; ks < 5; ++ks) {
    #pragma unroll
                                    for (int h = 0; h < 2; ++h) {                  // two MFMAs (32 pipe cycles), then two outputs of the item before
                                                                                   // (8 VALU): a wave cannot issue past an MFMA that waits for the
                                                                                   // pipe, so the vector work sits BETWEEN the MFMAs
    #pragma unroll
                                        for (int n_ = 2 * h; n_ < 2 * h + 2; ++n_)
                                            acc[n_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfa[ks][n_], bq[ks], acc[n_], 0, 0, 0);
                                        __builtin_amdgcn_sched_barrier(0);
                                        if (k > 0 && ks < 4) {
                                            out2(pacc, word, ks, h, owned);
                                            __builtin_amdgcn_sched_barrier(0);
                                        }
                                    }
                                }
                                if (k > 0) wr(k - 1, word);
                            };
                            v4i bq[2][5], acc[2][4];
                            if constexpr (COLD) {
    #pragma unroll
                                for (int k = 0; k < NGMAX; ++k) {
                                    if (k >= NG) break;
                                    rd(bq[0], k);
                                    stage(0, bq[0], acc[0], acc[0]);
                                    v4i word = {0, 0, 0, 0};
                                    const bool cok = owned && (k + 1 < NG || lastok);
    #pragma unroll
                                    for (int q = 0; q < 8; ++q) out2(acc[0], word, q >> 1, q & 1, cok);
                                    if (k + 1 < NG || lastok) wr(k, word);
                                }
                                return;
                            }
                            rd(bq[0], 0);
    #pragma unroll
                            for (int k = 0; k < NGMAX; ++k) {
                                if (k >= NG) break;
                                if (k + 1 < NGMAX && k + 1 < NG) rd(bq[(k + 1) & 1], k + 1);
                                stage(k, bq[k & 1], acc[k & 1], acc[(k + 1) & 1]);
                            }
                            {
                                v4i word = {0, 0, 0, 0};
                                if ((NG - 1) & 1) {
    #pragma unroll
                                    for (int q = 0; q < 8; ++q) out2(acc[1], word, q >> 1, q & 1, false);
                                } else {
    #pragma unroll
                                    for (int q = 0; q < 8; ++q) out2(acc[0], word, q >> 1, q & 1, false);
                                }
                                if (lastok) *(v4i *)(mid + (((NG - 1) & 1) ? w1 : w0) + (NG - 1) * 256) = word;
                            }
                        };
                        row_a(std::false_type{});
                        if (__builtin_amdgcn_ballot_w64(ymx > QHI || ymn < QLO) != 0ull) row_a(std::true_type{});
                    }
                } else if (r0 < rend) {                  // the band's last interval: only the pooling waves work; the next band's first rows
                    bd = band_at(r0);
                    in_hi = first_rows(bd);
                }
                stamp();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stamp();
            }
        }

    } else {
        // ---- conv3_2's weights: 32 channels, 18 fragments
        v4i wfb[9][2], cinb[2];
#pragma unroll
        for (int ks = 0; ks < 9; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) wfb[ks][n] = *(const v4i *)(p.w2 + ((size_t)(cb * 9 + ks) * 2 + n) * 1024 + lane * 16);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const v4i b2 = *(const v4i *)(p.bias2 + cb * 32 + 8 * g + 4 * n);
#pragma unroll
            for (int r = 0; r < 4; ++r) cinb[n][r] = F2 == 2 ? b2[r] + 0x4B400000 : b2[r];
        }
        const Epi e2 = make_epi<F2>(p.rq2);
        const float invWo = 1.0f / (float)Wo;
        for (int r0 = rbeg; r0 < rend;) {
            const int b = bd.b, j0 = bd.j0, j1 = bd.j1;
            const int nA = 1 + (j1 - j0 - 1 + 1) / 2;         // steps: 1 pooled row, then 2 each
            r0 += j1 - j0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // the band's first input rows (and, the first time, the zero fill) have landed
            stamp();
            int8_t *const outb = p.out + (((size_t)b * (Ho + 2) + 1) * (Wo + 2) + 1) * 64 + cb * 32;
            int wdone = 0;
            for (int t = 0; t <= nA; ++t) {
                if (t == nA && r0 < rend) bd = band_at(r0);                        // (kept in step with the other role)
                if (t >= 1) {
                    // ---- the band's windows that step t - 1 completed and that are not done: a multiple of two groups per channel block
                    const int je = je_of(Band{b, j0, j1}, t - 1);
                    const int avail = (je - j0) * Wo - wdone;
                    const bool all = t == nA || Wo < 32;
                    const int ngb = all ? (avail + 15) >> 4 : ((avail >> 4) & ~1);
                    const int nwin = all ? avail : ngb * 16, wlo = wdone;
                    wdone += nwin;
                    float zmx = MAGIC, zmn = MAGIC;
                    auto locate = [&](int grp, int &oyr, int &ox) {
                        const int wi = wlo + min(grp * 16 + li, nwin - 1);
                        oyr = (int)(((float)wi + 0.5f) * invWo);
                        ox = wi - oyr * Wo;
                    };
                    auto issue = [&](int grp, v4i (&acc)[4][2]) {
                        int oyr, ox;
                        locate(grp, oyr, ox);
                        const int ar = 2 * (j0 + oyr), x0 = 2 * ox;
                        // (ar and x0 are even: row + 1 never wraps in the ring, columns x and x + 1 share bit 4 and their slots differ in bit 0)
                        int xoff[4], roff[4];
                        roff[0] = (ar & (RMID - 1)) * MPITCH;
                        roff[1] = roff[0] + MPITCH;
                        roff[2] = ((ar + 2) & (RMID - 1)) * MPITCH;
                        roff[3] = roff[2] + MPITCH;
                        xoff[0] = g * PLANE + ((x0 ^ ((x0 >> 4) & 1)) << 4);
                        xoff[1] = xoff[0] ^ 16;
                        xoff[2] = g * PLANE + (((x0 + 2) ^ (((x0 + 2) >> 4) & 1)) << 4);
                        xoff[3] = xoff[2] ^ 16;
    #pragma unroll
                        for (int v = 0; v < 4; ++v)
    #pragma unroll
                            for (int n = 0; n < 2; ++n) acc[v][n] = cinb[n];
                        v4i bq[4][4];
                        auto rd = [&](int r) {
    #pragma unroll
                            for (int c = 0; c < 4; ++c) bq[r][c] = *(const v4i *)(mid + roff[r] + xoff[c]);
                        };
                        auto mm = [&](int r) {
    #pragma unroll
                            for (int dy = 0; dy < 2; ++dy) {
                                const int ty = r - dy;
                                if (ty < 0 || ty > 2) continue;
    #pragma unroll
                                for (int c = 0; c < 4; ++c)
    #pragma unroll
                                    for (int dx = 0; dx < 2; ++dx) {
                                        const int tx = c - dx;
                                        if (tx < 0 || tx > 2) continue;
    #pragma unroll
                                        for (int n = 0; n < 2; ++n)
                                            acc[2 * dy + dx][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfb[ty * 3 + tx][n], bq[r][c], acc[2 * dy + dx][n], 0, 0, 0);
                                    }
                            }
                        };
                        rd(0);
                        rd(1);
                        mm(0);
                        rd(2);
                        mm(1);
                        rd(3);
                        mm(2);
                        mm(3);
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);
                    };
                    auto finish = [&](int grp, const v4i (&acc)[4][2], auto coldc) {
                        constexpr bool COLD = decltype(coldc)::value;
                        int oyr, ox;
                        locate(grp, oyr, ox);
                        unsigned int word[2];
    #pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            float pos[4], neg[4];
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int m = max(max(acc[0][n][r], acc[1][n][r]), max(acc[2][n][r], acc[3][n][r]));
                                const float tf = F2 == 2 ? __int_as_float(m) : (float)m;
                                pos[r] = fmaf(tf, e2.sp, e2.cp);
                                neg[r] = fmaf(tf, e2.sn, e2.cn);
                            }
                            if constexpr (!COLD) {
                                zmx = qvmax3(qvmax3(zmx, pos[0], pos[1]), pos[2], pos[3]);
                                zmn = qvmin3(qvmin3(zmn, neg[0], neg[1]), neg[2], neg[3]);
                                qmax_to_byte<0>(word[n], pos[0], neg[0]);
                                qmax_to_byte<1>(word[n], pos[1], neg[1]);
                                qmax_to_byte<2>(word[n], pos[2], neg[2]);
                                qmax_to_byte<3>(word[n], pos[3], neg[3]);
                            } else {
                                float yc[4];
    #pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float y = qvmax(pos[r], neg[r]);
                                    yc[r] = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                                    nsat += (grp * 16 + li < nwin && y != yc[r]) ? 1u : 0u;
                                }
                                word[n] = qpack4(yc[0], yc[1], yc[2], yc[3]);
                            }
                        }
                        int8_t *dst = outb + (((j0 + oyr) * (Wo + 2) + ox) * 64 + 8 * g);
                        *(v2u *)dst = (v2u){word[0], word[1]};
                    };
                    {
                        // (the hot pass software-pipelined INSIDE the wave -- the finish of group i - 1 and the addresses of group i + 1 in
                        // slots between the MFMA pairs of group i, two accumulator sets -- is bit-exact and SLOWER: 39.5 against 37.8 us,
                        // scratch/pxpair_r5_bpipe.hip, profiles/r05_notes.md section 7)
                        v4i acc[4][2];
    #pragma unroll 1
                        for (int grp = st; grp < ngb; grp += 2) {
                            issue(grp, acc);
                            finish(grp, acc, std::false_type{});
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(zmx > QHI || zmn < QLO) != 0ull) {
                        v4i accC[4][2];
    #pragma unroll 1
                        for (int grp = st; grp < ngb; grp += 2) {
                            issue(grp, accC);
                            finish(grp, accC, std::true_type{});
                        }
                    }
                }
                stamp();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stamp();
            }
        }

    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nsat) atomicAdd(roleA ? &p.ctr1->sat : &p.ctr2->sat, (unsigned long long)nsat);
}

// ==========================================================================================
// conv4_1 -> conv4_2 + pool4 (models/slim_yolo_v2.py:268-289; conv_normal calls 5 and 6, c_embedding/yolo_forward.c:1226-1236) in
// one launch, the same schedule: the two layers on different waves of every SIMD.  Only this schedule fits the pair: a wave
// that ran both layers would need 18 + 36 fragments of its 32-channel block = 216 VGPRs of weights.
//   * waves 0-3 run conv4_1 (64 -> 128 at 52 x 52): wave w holds output channels 32 w .. 32 w + 31 (9 taps x 2 n-tiles = 18
//     fragments) and walks ALL four map rows of the interval, row by row in groups of 16 pixels (a 52-pixel row = 3 full groups
//     and a quarter: this role has the lighter load, 18 MFMAs per item against the other's 144 per group);
//   * waves 4-7 run conv4_2 + pool (128 -> 128): wave 4 + w holds output channels 32 w .. 32 w + 31 (9 taps x 2 k-halves x 2
//     n-tiles = 36 fragments = 144 VGPRs) and takes ALL the interval's pooling windows in groups of 16.
// Map ring: 16 rows x 8 planes of 16-byte chunks (slot = x ^ ((x >> 4) & 1)); input ring: 12 rows of 64-byte pixels by LDS-DMA,
// XOR-swizzled on the source side as in convpx.hip (chunk ^ 2 ((x >> 2) & 1)).  Everything else as pxpair3r_kernel.
template <int F1, int F2>
__global__ __launch_bounds__(512, 2) void pxpair4r_kernel(const PairParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int nstamp = 0;
    auto stamp = [&]() {
#if PAIR_DIAG
        if (p.stamps && (tid == 0 || tid == 256) && nstamp < 16) p.stamps[(size_t)blockIdx.x * 32 + (tid >> 8) * 16 + nstamp++] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    (void)nstamp;
    stamp();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool roleA = wave < 4;
    const int cb = wave & 3;                              // 32-channel block of this wave's layer
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
    const int NG = (W + 15) >> 4;                         // 16-pixel groups per map row (<= 4)
    const int MS = W + 2, PLANE = MS * 16, MPITCH = 8 * PLANE;
    const int PPR = (W + 2 + 15) >> 4, IPITCH = PPR * 1024;
    char *const mid = smem;                               // [RMID][8 planes][MS] 16-byte chunks
    char *const inp = smem + RMID * MPITCH;               // [RIN][PPR] 1 KiB pieces of 16 pixels x 64 bytes
    auto dma_rows = [&](int b, int ra, int rb) {          // role A only: piece q by wave q % 4
        const int np = (rb - ra) * PPR;
        for (int q = cb; q < np; q += 4) {
            const int rr = q / PPR, pc = q - rr * PPR, row = ra + rr;
            const int col = pc * 16 + (lane >> 2);
            const int sch = (lane & 3) ^ (((col >> 2) & 1) << 1);
            const int8_t *src = p.in + ((size_t)(b * (H + 2) + row) * (W + 2) + min(col, W + 1)) * 64 + sch * 16;
            qglds16(src, inp + (row % RIN) * IPITCH + pc * 1024);
        }
    };
    const int G_ = gridDim.x, Rtot = p.B * Ho;
    const int rbeg = (int)((long long)Rtot * blockIdx.x / G_), rend = (int)((long long)Rtot * (blockIdx.x + 1) / G_);
    struct Band { int b, j0, j1; };
    auto band_at = [&](int r0) {
        Band q;
        q.b = r0 / Ho;
        q.j0 = r0 - q.b * Ho;
        q.j1 = min(Ho, q.j0 + (rend - r0));
        return q;
    };
    auto je_of = [&](const Band &q, int t) { return min(q.j0 + 1 + 2 * t, q.j1); };
    auto first_rows = [&](const Band &q) {
        const int hi = min(2 * je_of(q, 0) + 2, H + 1) + 1;
        dma_rows(q.b, max(2 * q.j0, 1) - 1, hi);
        return hi;
    };
    Band bd = band_at(rbeg);
    int in_hi = 0;
    if (roleA) in_hi = first_rows(bd);
    unsigned int nsat = 0;

    for (int i = tid * 16; i < RMID * MPITCH; i += 512 * 16) *(v4i *)(mid + i) = (v4i){0, 0, 0, 0};

    if (roleA) {
        // ---- conv4_1's weights of this wave's 32 channels: [tap][n-tile]
        v4i wfa[9][2], cina[2];
#pragma unroll
        for (int ks = 0; ks < 9; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) wfa[ks][n] = *(const v4i *)(p.w1 + ((size_t)(cb * 9 + ks) * 2 + n) * 1024 + lane * 16);
#pragma unroll
        for (int n = 0; n < 2; ++n) {                     // accumulator register r of n-tile n of lane group g = channel 32 cb + 8 g + 4 n + r
            const v4i b1 = *(const v4i *)(p.bias1 + cb * 32 + 8 * g + 4 * n);
#pragma unroll
            for (int r = 0; r < 4; ++r) cina[n][r] = F1 == 2 ? b1[r] + 0x4B400000 : b1[r];
        }
        const Epi e1 = make_epi<F1>(p.rq1);
        // byte offset inside an input row of chunk g of pixel li + tx of group 0 (group k adds 1024): the swizzle bit of a pixel is
        // bit 2 of its column, which 16 k does not touch
        int cx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) cx[tx] = (li + tx) * 64 + 16 * (g ^ ((((li + tx) >> 2) & 1) << 1));
        // byte offset inside a map row of this lane's 8 channels (chunk 2 cb + (g >> 1), half g & 1) of group 0's pixel, k even / odd
        int wc[2];
        wc[0] = (2 * cb + (g >> 1)) * PLANE + 8 * (g & 1) + 16 * (li < 15 ? li + 1 : 17);
        wc[1] = (2 * cb + (g >> 1)) * PLANE + 8 * (g & 1) + 16 * (li < 15 ? ((li + 1) ^ 1) : 16);
        const bool lastok = 16 * (NG - 1) + li < W;
        for (int r0 = rbeg; r0 < rend;) {
            const int b = bd.b, j0 = bd.j0, j1 = bd.j1;
            const int nA = 1 + (j1 - j0) / 2;
            r0 += j1 - j0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stamp();
            for (int t = 0; t <= nA; ++t) {
                if (t < nA) {
                    const int PA = t == 0 ? 2 * j0 : 2 * je_of(bd, t - 1) + 2, PB = 2 * je_of(bd, t) + 2;
                    if (t + 1 < nA) {
                        const int hi = min(2 * je_of(bd, t + 1) + 2, H + 1) + 1;
                        dma_rows(b, in_hi, hi);
                        in_hi = max(in_hi, hi);
                    }
                    float ymx = MAGIC, ymn = MAGIC;
                    // one map row: NG items of 16 pixels, pipelined as in pxpair3r_kernel (operands of item k + 1 read, item k - 1
                    // requantised and stored, under the 18 MFMAs of item k)
                    auto row_a = [&](int P, auto coldc) {
                        constexpr bool COLD = decltype(coldc)::value;
                        int a[3][3];
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty) {
                            const int rb = ((P - 1 + ty) % RIN) * IPITCH;
#pragma unroll
                            for (int tx = 0; tx < 3; ++tx) a[ty][tx] = rb + cx[tx];
                        }
                        const int wrow = (P & (RMID - 1)) * MPITCH;
                        const int w0 = wrow + wc[0], w1 = wrow + wc[1];
                        const bool owned = P >= 2 * j0 + 1 && P < 2 * j1 + 1;
                        auto rd = [&](v4i (&bq)[9], int k) {
#pragma unroll
                            for (int tap = 0; tap < 9; ++tap) bq[tap] = *(const v4i *)(inp + a[tap / 3][tap % 3] + k * 1024);
                        };
                        auto out2 = [&](const v4i (&acc)[2], unsigned int (&word)[2], int n_, int rr, bool cok) {
                            float pos[2], neg[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int v = acc[n_][2 * rr + u];
                                const float tf = F1 == 2 ? __int_as_float(v) : (float)v;
                                pos[u] = fmaf(tf, e1.sp, e1.cp);
                                neg[u] = fmaf(tf, e1.sn, e1.cn);
                            }
                            if constexpr (!COLD) {
                                ymx = qvmax3(ymx, pos[0], pos[1]);
                                ymn = qvmin3(ymn, neg[0], neg[1]);
                                if (rr == 0) {
                                    qmax_to_byte<0>(word[n_], pos[0], neg[0]);
                                    qmax_to_byte<1>(word[n_], pos[1], neg[1]);
                                } else {
                                    qmax_to_byte<2>(word[n_], pos[0], neg[0]);
                                    qmax_to_byte<3>(word[n_], pos[1], neg[1]);
                                }
                            } else {
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const float y = qvmax(pos[u], neg[u]), yc = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                                    nsat += (cok && y != yc) ? 1u : 0u;
                                    const unsigned int by = __float_as_uint(yc) & 0xffu;
                                    word[n_] = (rr == 0 && u == 0) ? by : (word[n_] | (by << (8 * (2 * rr + u))));
                                }
                            }
                        };
                        auto wr = [&](int k, const unsigned int (&word)[2]) {
                            *(v2u *)(mid + ((k & 1) ? w1 : w0) + k * 256) = (v2u){word[0], word[1]};
                        };
                        auto stage = [&](int k, const v4i (&bq)[9], v4i (&acc)[2], const v4i (&pacc)[2]) {
                            acc[0] = cina[0];
                            acc[1] = cina[1];
                            unsigned int word[2] = {0u, 0u};
#pragma unroll
                            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                                for (int n_ = 0; n_ < 2; ++n_) acc[n_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfa[tap][n_], bq[tap], acc[n_], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if (k > 0 && (tap & 1) && tap < 8) {               // behind MFMA pairs 1, 3, 5, 7: two outputs of the item before
                                    out2(pacc, word, tap >> 2, (tap >> 1) & 1, owned);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            if (k > 0) wr(k - 1, word);
                        };
                        v4i bq[2][9], acc[2][2];
                        if constexpr (COLD) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if (k >= NG) break;
                                rd(bq[0], k);
                                stage(0, bq[0], acc[0], acc[0]);
                                unsigned int word[2] = {0u, 0u};
                                const bool cok = owned && (k + 1 < NG || lastok);
#pragma unroll
                                for (int q = 0; q < 4; ++q) out2(acc[0], word, q >> 1, q & 1, cok);
                                if (k + 1 < NG || lastok) wr(k, word);
                            }
                            return;
                        }
                        rd(bq[0], 0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (k >= NG) break;
                            if (k + 1 < 4 && k + 1 < NG) rd(bq[(k + 1) & 1], k + 1);
                            stage(k, bq[k & 1], acc[k & 1], acc[(k + 1) & 1]);
                        }
                        {
                            unsigned int word[2] = {0u, 0u};
                            if ((NG - 1) & 1) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) out2(acc[1], word, q >> 1, q & 1, false);
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) out2(acc[0], word, q >> 1, q & 1, false);
                            }
                            if (lastok) *(v2u *)(mid + (((NG - 1) & 1) ? w1 : w0) + (NG - 1) * 256) = (v2u){word[0], word[1]};
                        }
                    };
                    for (int P = PA; P < PB; ++P) {
                        if (P == 0 || P == H + 1) {                                // the map's zero halo rows: each wave its own planes
                            for (int o = lane * 16; o < 2 * PLANE; o += 1024)
                                *(v4i *)(mid + (P & (RMID - 1)) * MPITCH + 2 * cb * PLANE + o) = (v4i){0, 0, 0, 0};
                        } else {
                            row_a(P, std::false_type{});
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(ymx > QHI || ymn < QLO) != 0ull)
                        for (int P = max(PA, 1); P < min(PB, H + 1); ++P) row_a(P, std::true_type{});
                } else if (r0 < rend) {
                    bd = band_at(r0);
                    in_hi = first_rows(bd);
                }
                stamp();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stamp();
            }
        }
    } else {
        // ---- conv4_2's weights of this wave's 32 channels: [tap * 2 + k-half][n-tile]
        v4i wfb[18][2], cinb[2];
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) wfb[ks][n] = *(const v4i *)(p.w2 + ((size_t)(cb * 18 + ks) * 2 + n) * 1024 + lane * 16);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const v4i b2 = *(const v4i *)(p.bias2 + cb * 32 + 8 * g + 4 * n);
#pragma unroll
            for (int r = 0; r < 4; ++r) cinb[n][r] = F2 == 2 ? b2[r] + 0x4B400000 : b2[r];
        }
        const Epi e2 = make_epi<F2>(p.rq2);
        const float invWo = 1.0f / (float)Wo;
        for (int r0 = rbeg; r0 < rend;) {
            const int b = bd.b, j0 = bd.j0, j1 = bd.j1;
            const int nA = 1 + (j1 - j0) / 2;
            r0 += j1 - j0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stamp();
            int8_t *const outb = p.out + (((size_t)b * (Ho + 2) + 1) * (Wo + 2) + 1) * 128 + cb * 32;
            int wdone = 0;
            for (int t = 0; t <= nA; ++t) {
                if (t == nA && r0 < rend) bd = band_at(r0);
                if (t >= 1) {
                    const int je = je_of(Band{b, j0, j1}, t - 1);
                    const int avail = (je - j0) * Wo - wdone;
                    const bool all = t == nA || Wo < 16;
                    const int ngb = all ? (avail + 15) >> 4 : (avail >> 4);        // whole groups; the rest (less than a row) waits
                    const int nwin = all ? avail : ngb * 16, wlo = wdone;
                    wdone += nwin;
                    float zmx = MAGIC, zmn = MAGIC;
                    auto locate = [&](int grp, int &oyr, int &ox) {
                        const int wi = wlo + min(grp * 16 + li, nwin - 1);
                        oyr = (int)(((float)wi + 0.5f) * invWo);
                        ox = wi - oyr * Wo;
                    };
                    auto issue = [&](int grp, v4i (&acc)[4][2]) {
                        int oyr, ox;
                        locate(grp, oyr, ox);
                        const int ar = 2 * (j0 + oyr), x0 = 2 * ox;
                        int xoff[4], roff[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int x = x0 + c;
                            xoff[c] = g * PLANE + ((x ^ ((x >> 4) & 1)) << 4);
                            roff[c] = ((ar + c) & (RMID - 1)) * MPITCH;
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int n = 0; n < 2; ++n) acc[v][n] = cinb[n];
                        // neighbourhood row r: 4 columns x 2 channel halves read, then its MFMAs (a row at a time: 144 VGPRs of weights
                        // leave room for one row of operands)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v4i bq[4][2];
#pragma unroll
                            for (int c = 0; c < 4; ++c)
#pragma unroll
                                for (int h = 0; h < 2; ++h) bq[c][h] = *(const v4i *)(mid + roff[r] + xoff[c] + h * 4 * PLANE);
#pragma unroll
                            for (int dy = 0; dy < 2; ++dy) {
                                const int ty = r - dy;
                                if (ty < 0 || ty > 2) continue;
#pragma unroll
                                for (int c = 0; c < 4; ++c)
#pragma unroll
                                    for (int dx = 0; dx < 2; ++dx) {
                                        const int tx = c - dx;
                                        if (tx < 0 || tx > 2) continue;
#pragma unroll
                                        for (int h = 0; h < 2; ++h)
#pragma unroll
                                            for (int n = 0; n < 2; ++n)
                                                acc[2 * dy + dx][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfb[(ty * 3 + tx) * 2 + h][n], bq[c][h], acc[2 * dy + dx][n], 0, 0, 0);
                                    }
                            }
                        }
                    };
                    auto finish = [&](int grp, const v4i (&acc)[4][2], auto coldc) {
                        constexpr bool COLD = decltype(coldc)::value;
                        int oyr, ox;
                        locate(grp, oyr, ox);
                        unsigned int word[2];
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            float pos[4], neg[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int m = max(max(acc[0][n][r], acc[1][n][r]), max(acc[2][n][r], acc[3][n][r]));
                                const float tf = F2 == 2 ? __int_as_float(m) : (float)m;
                                pos[r] = fmaf(tf, e2.sp, e2.cp);
                                neg[r] = fmaf(tf, e2.sn, e2.cn);
                            }
                            if constexpr (!COLD) {
                                zmx = qvmax3(qvmax3(zmx, pos[0], pos[1]), pos[2], pos[3]);
                                zmn = qvmin3(qvmin3(zmn, neg[0], neg[1]), neg[2], neg[3]);
                                qmax_to_byte<0>(word[n], pos[0], neg[0]);
                                qmax_to_byte<1>(word[n], pos[1], neg[1]);
                                qmax_to_byte<2>(word[n], pos[2], neg[2]);
                                qmax_to_byte<3>(word[n], pos[3], neg[3]);
                            } else {
                                float yc[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float y = qvmax(pos[r], neg[r]);
                                    yc[r] = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                                    nsat += (grp * 16 + li < nwin && y != yc[r]) ? 1u : 0u;
                                }
                                word[n] = qpack4(yc[0], yc[1], yc[2], yc[3]);
                            }
                        }
                        int8_t *dst = outb + (((j0 + oyr) * (Wo + 2) + ox) * 128 + 8 * g);
                        *(v2u *)dst = (v2u){word[0], word[1]};
                    };
                    {
                        v4i acc[4][2];
#pragma unroll 1
                        for (int grp = 0; grp < ngb; ++grp) {
                            issue(grp, acc);
                            finish(grp, acc, std::false_type{});
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(zmx > QHI || zmn < QLO) != 0ull) {
                        v4i accC[4][2];
#pragma unroll 1
                        for (int grp = 0; grp < ngb; ++grp) {
                            issue(grp, accC);
                            finish(grp, accC, std::true_type{});
                        }
                    }
                }
                stamp();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stamp();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nsat) atomicAdd(roleA ? &p.ctr1->sat : &p.ctr2->sat, (unsigned long long)nsat);
}

// ------------------------------------------------------------------------------------------
size_t y355_pair3_packed_bytes(void) { return (size_t)2 * 5 * 2 * 1024; }

// conv3_1's weights q_w [64][32][3][3] in this kernel's fragment order (convpx.hip's with 16 NTN = 32 channels per block):
// fragment ((cb * 5 + ks) * 2 + n), lane (i = l & 15, g = l >> 4), 16 bytes: row i = output channel 32 cb + 8 (i >> 2) + 4 n + (i & 3),
// k = tap 2 ks + (g >> 1) (tap 9: zeros), input channels 16 (g & 1) .. + 15
void y355_pack_pair3(const int8_t *q_w, int8_t *dst) {
    memset(dst, 0, y355_pair3_packed_bytes());
    for (int cb = 0; cb < 2; ++cb)
        for (int ks = 0; ks < 5; ++ks)
            for (int n = 0; n < 2; ++n)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4;
                    const int ch = cb * 32 + 8 * (i >> 2) + 4 * n + (i & 3);
                    const int tap = 2 * ks + (g >> 1), c0 = 16 * (g & 1);
                    if (tap > 8) continue;
                    for (int kk = 0; kk < 16; ++kk)
                        dst[(((size_t)cb * 5 + ks) * 2 + n) * 1024 + l * 16 + kk] = q_w[((size_t)ch * 32 + c0 + kk) * 9 + tap];
                }
}

namespace {
size_t pair_lds(int W) {
    const int MPITCH = 4 * (W + 2) * 16, IPITCH = ((W + 2 + 31) >> 5) * 1024;
    return (size_t)RMID * MPITCH + (size_t)RIN * IPITCH;
}
int fold_of(const Requant &rq) {
    if (rq.shl != 0) return 0;
    return (rq.tmax_log2 <= 22 && rq.sh <= 22 && rq.sh - rq.lk >= -8) ? 2 : 1;
}
template <int F1, int F2>
void launch_(const PairParams &p, int grid, size_t lds, hipStream_t s) {
    PairParams q = p;
    q.ev_start = q.ev_stop = nullptr;
    if (p.variant == 1) Y355_LAUNCH((pxpair3r_kernel<F1, F2>), dim3(grid), dim3(512), lds, s, p.ev_start, p.ev_stop, q);
    else Y355_LAUNCH((pxpair3_kernel<F1, F2>), dim3(grid), dim3(512), lds, s, p.ev_start, p.ev_stop, q);
}
}  // namespace

int y355_prepare_pair3(void) {
    int e = (int)hipFuncSetAttribute((const void *)pxpair3_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3r_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3r_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3r_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair3r_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

// what the two layers' own launches would have to be for the fused one to stand in: plain runs on the fp32-exact epilogue
// with the accumulator shift 0, a map narrow enough for the two rings
bool y355_pair3_eligible(const Requant &rq1, const Requant &rq2, int H, int W) {
    for (const Requant *rq : {&rq1, &rq2}) {
        if (rq->wide || rq->tmax_log2 > 24 || fold_of(*rq) == 0) return false;
        if (rq->neg_mul < 0 || rq->neg_mul > (1 << rq->lk)) return false;
    }
    if ((H | W) & 1 || W < 16 || W > 16 * NGMAX || H < 2) return false;
    return pair_lds(W) <= 160 * 1024;
}

// false = not available for this launch: the caller runs the two layers' own launches
bool y355_launch_pair3(const PairParams &p, hipStream_t s) {
    if (!y355_pair3_eligible(p.rq1, p.rq2, p.H, p.W)) return false;
    if ((long long)p.B * (p.H + 2) * (p.W + 2) * 32 >= (1ll << 31)) return false;      // 32-bit row arithmetic
    const int total = p.B * (p.H / 2);
    int grid = 256;                                                // one 8-wave workgroup per CU
    if (p.grid_limit > 0 && p.grid_limit < grid) grid = p.grid_limit;
    if (grid > total) grid = total;
    const size_t lds = pair_lds(p.W);
    const int f1 = fold_of(p.rq1), f2 = fold_of(p.rq2);
    if (f1 == 2 && f2 == 2) launch_<2, 2>(p, grid, lds, s);
    else if (f1 == 2) launch_<2, 1>(p, grid, lds, s);
    else if (f2 == 2) launch_<1, 2>(p, grid, lds, s);
    else launch_<1, 1>(p, grid, lds, s);
    return true;
}

// ---- conv4_1 -> conv4_2 + pool4 (pxpair4r_kernel)
size_t y355_pair4_packed_bytes(void) { return (size_t)4 * 9 * 2 * 1024; }
// conv4_1's weights q_w [128][64][3][3]: fragment ((cb * 9 + tap) * 2 + n), lane (i = l & 15, g = l >> 4), 16 bytes:
// row i = output channel 32 cb + 8 (i >> 2) + 4 n + (i & 3), input channels 16 g .. + 15 of tap
void y355_pack_pair4(const int8_t *q_w, int8_t *dst) {
    memset(dst, 0, y355_pair4_packed_bytes());
    for (int cb = 0; cb < 4; ++cb)
        for (int tap = 0; tap < 9; ++tap)
            for (int n = 0; n < 2; ++n)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4;
                    const int ch = cb * 32 + 8 * (i >> 2) + 4 * n + (i & 3);
                    for (int kk = 0; kk < 16; ++kk)
                        dst[(((size_t)cb * 9 + tap) * 2 + n) * 1024 + l * 16 + kk] = q_w[((size_t)ch * 64 + 16 * g + kk) * 9 + tap];
                }
}
namespace {
size_t pair4_lds(int W) {
    const int MPITCH = 8 * (W + 2) * 16, IPITCH = ((W + 2 + 15) >> 4) * 1024;
    return (size_t)RMID * MPITCH + (size_t)RIN * IPITCH;
}
template <int F1, int F2>
void launch4_(const PairParams &p, int grid, size_t lds, hipStream_t s) {
    PairParams q = p;
    q.ev_start = q.ev_stop = nullptr;
    Y355_LAUNCH((pxpair4r_kernel<F1, F2>), dim3(grid), dim3(512), lds, s, p.ev_start, p.ev_stop, q);
}
}  // namespace
int y355_prepare_pair4(void) {
    int e = (int)hipFuncSetAttribute((const void *)pxpair4r_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair4r_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair4r_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)pxpair4r_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}
bool y355_pair4_eligible(const Requant &rq1, const Requant &rq2, int H, int W) {
    for (const Requant *rq : {&rq1, &rq2}) {
        if (rq->wide || rq->tmax_log2 > 24 || fold_of(*rq) == 0) return false;
        if (rq->neg_mul < 0 || rq->neg_mul > (1 << rq->lk)) return false;
    }
    if ((H | W) & 1 || W < 16 || W > 64 || H < 2) return false;
    return pair4_lds(W) <= 160 * 1024;
}
bool y355_launch_pair4(const PairParams &p, hipStream_t s) {
    if (!y355_pair4_eligible(p.rq1, p.rq2, p.H, p.W)) return false;
    if ((long long)p.B * (p.H + 2) * (p.W + 2) * 64 >= (1ll << 31)) return false;
    const int total = p.B * (p.H / 2);
    int grid = 256;
    if (p.grid_limit > 0 && p.grid_limit < grid) grid = p.grid_limit;
    if (grid > total) grid = total;
    const size_t lds = pair4_lds(p.W);
    const int f1 = fold_of(p.rq1), f2 = fold_of(p.rq2);
    if (f1 == 2 && f2 == 2) launch4_<2, 2>(p, grid, lds, s);
    else if (f1 == 2) launch4_<2, 1>(p, grid, lds, s);
    else if (f2 == 2) launch4_<1, 2>(p, grid, lds, s);
    else launch4_<1, 1>(p, grid, lds, s);
    return true;
}
