// yolo355 -- conv3_1 -> conv3_2 + pool3 of the q_bf path in ONE launch (models/slim_yolo_v2.py:246-267: conv3_1, a_tracker3_1,
// conv3_2, a_tracker3_2, pool3 are a straight chain with no other consumer; the FPGA's conv_normal calls 3 and 4,
// c_embedding/yolo_forward.c:1214-1225).
//
// convpx.hip runs the two layers as two launches: conv3_1's 64-channel map (44 MB at B = 64) goes out to HBM and comes back
// through an LDS-DMA ring, and each launch pays its own weights-into-registers prologue and drain.  Here a workgroup owns a
// BAND of pooled output rows of one image:
//   conv3_1 for the map rows the next pooled rows need (a band recomputes one row above and one below itself), straight into
//      an LDS ring of 16 map rows -- int8, requantised, exactly the bytes the unfused path stores; the map never exists in HBM;
//   conv3_2 + 2x2 max-pool over those rows (convpx's pooled form: the 4x4 neighbourhood of a window read once and fed to
//      the four conv outputs of the window), stored to HBM.
// The two layers run on DIFFERENT WAVES of every SIMD (pxpair3r_kernel below).  Round 5 built three schedules; this is the one
// that pays (+4.5 % images/s against two launches).  The other two -- every wave alternating between the layers (+2.2 %), and the
// same layer-role schedule for conv4_1 -> conv4_2 + pool4 (bit-exact, NO faster than its two launches: every wave of a role reads
// every pixel, the LDS is the bound) -- were taken out of the library in round 6; their source is
// scratch/pxpair_r5_all_three_kernels.hip, their measurements profiles/r05_notes.md sections 3-5.
//
// Layouts this kernel owns on both sides, so they are built for its access patterns:
//   * the map ring is PLANAR: a row is four planes of 16-byte chunks (plane c = channels 16 c .. 16 c + 15 of every pixel),
//     slot of padded column x = x ^ ((x >> 4) & 1).  The conv3_2 lanes (16 windows = every second pixel, one chunk each) and
//     the conv3_1 lanes (16 consecutive pixels) both touch 16 distinct 16-byte bank groups: no LDS bank conflicts (convpx's
//     pooled layers, whose rows arrive by LDS-DMA as whole 64-byte pixels, measure 0.50);
//   * conv3_1 walks ROWS: a wave takes consecutive 16-pixel groups of a row, so every LDS address is (per-row VGPR) +
//     immediate -- 5 vector adds per row instead of ~70 address instructions per group in convpx's flat walk;
//   * input rows (32-byte pixels) arrive by LDS-DMA into a ring of 12 rows, one interval ahead (the conv3_1 role waits for its own DMA with
//     vmcnt(0): it issues no other vector-memory operation).
// Epilogues: front.hip's fp32 form on exact integers, FOLD 1 / 2 per layer; the hot passes do not clamp and
// track the LeakyReLU branches' extremes, a cold pass redoes a wave's share clamped and counts (per layer, and for conv3_1 only
// on the rows the band owns) when a value left [-127, 127].  Integer semantics bit for bit those of convpx.hip.
#include "y355_common.h"
#include <cstring>
#include <type_traits>

namespace {
constexpr float MAGIC = 12582912.0f;                 // 1.5 * 2^23
constexpr float QLO = 12582785.0f, QHI = 12583039.0f;
constexpr int RMID = 16;                             // map ring, rows (power of two: phase B's lanes mask)
constexpr int RIN = 12;                              // input ring, rows (scalar modulo)
constexpr int NGMAX = 7;                             // 16-pixel groups per map row: W <= 112
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

// ==========================================================================================
// The pair with the two layers on DIFFERENT WAVES of every SIMD.
//
// A homogeneous schedule (round 5's first: every wave holds both layers' weights of a 32-channel block, 112 VGPRs, and alternates
// the two phases between barriers) measured: conv3_1 (a third of the MACs) is half of the time -- 10 MFMAs and 8 output bytes
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
        // lane constants: a lane stores its pixel's whole 16-byte chunk g
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
                                // the lanes of the last group past the row's end computed on whatever the input ring holds beyond it:
                                // nothing of theirs is stored or counted, and it must not reach the clamp detection either (a
                                // spurious cold pass costs time, not correctness: ADVICE r5)
                                const float ymx_in = ymx, ymn_in = ymn;
                                v4i word = {0, 0, 0, 0};
                                if ((NG - 1) & 1) {
    #pragma unroll
                                    for (int q = 0; q < 8; ++q) out2(acc[1], word, q >> 1, q & 1, false);
                                } else {
    #pragma unroll
                                    for (int q = 0; q < 8; ++q) out2(acc[0], word, q >> 1, q & 1, false);
                                }
                                if (lastok) *(v4i *)(mid + (((NG - 1) & 1) ? w1 : w0) + (NG - 1) * 256) = word;
                                ymx = lastok ? ymx : ymx_in;
                                ymn = lastok ? ymn : ymn_in;
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


// ------------------------------------------------------------------------------------------
namespace {
size_t pair_lds(int W) {
    const int MPITCH = 4 * (W + 2) * 16, IPITCH = ((W + 2 + 31) >> 5) * 1024;
    return (size_t)RMID * MPITCH + (size_t)RIN * IPITCH + 1024;    // + one piece: the last group's lanes past the row end read (and discard) it
}
int fold_of(const Requant &rq) {
    if (rq.shl != 0) return 0;
    return (rq.tmax_log2 <= 22 && rq.sh <= 22 && rq.sh - rq.lk >= -8) ? 2 : 1;
}
template <int F1, int F2>
void launch_(const PairParams &p, int grid, size_t lds, hipStream_t s) {
    PairParams q = p;
    q.ev_start = q.ev_stop = nullptr;
    Y355_LAUNCH((pxpair3r_kernel<F1, F2>), dim3(grid), dim3(512), lds, s, p.ev_start, p.ev_stop, q);
}
}  // namespace

int y355_prepare_pair3(void) {
    int e = (int)hipFuncSetAttribute((const void *)pxpair3r_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    int grid = y355_cu_count();                                    // one 8-wave workgroup per CU
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
