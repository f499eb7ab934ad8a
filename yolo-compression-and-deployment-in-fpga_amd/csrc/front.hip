// yolo355 -- fused front end of the q_bf path: input quantise -> conv1(3->16) -> bias -> LeakyReLU(0.125) ->
// requantise -> 2x2 max-pool -> conv2(16->32) -> bias -> LeakyReLU -> requantise -> 2x2 max-pool, ONE kernel.
//
// Replaces models/slim_yolo_v2.py:218-244 (a_tracker_in, conv1, a_tracker1, pool1, conv2, a_tracker2, pool2)
// and the FPGA driver's first_conv + second_conv (c_embedding/yolo_forward.c:269-572: the same fusion
// boundary -- the camera frame goes in, the 32-channel quarter-resolution map comes out).
//
// Round 3 rewrite.  The round-2 kernel was instruction-issue-bound (profiles/r02_notes.md: 1 940 instructions per
// wave and tile, 17 VALU per conv1 MFMA); this one spends about 0.6 of that:
//   * the 2x2 pooling window is the unit of work.  A window of a 3x3 / pad-1 convolution followed by a 2x2 pool reads
//     a 4x4 input neighbourhood; the FOUR conv outputs of the window are four MFMAs over the SAME neighbourhood
//     operand with four weight fragments (the 3x3 filter placed at the four offsets inside the 4x4 neighbourhood,
//     zeros elsewhere).  conv1: 4x4 px x 4 B = exactly one 64-deep k-step; conv2: four k-steps (neighbourhood rows)
//     of 4 px x 16 ch.  One set of LDS reads feeds all four pool partners;
//   * weights are the MFMA's A operand (rows = output channels), pixels the B operand (columns = 16 windows): a lane
//     then holds FOUR CHANNELS of ONE window in an accumulator, the pool is an element-wise max over the four MFMA
//     results (no cross-lane step), and the int8 results leave as one packed ds_write_b32 / b64 per lane;
//   * the biases ride in as the MFMAs' C operand (when the layer's accumulator shift is zero: template FOLD);
//   * the requantisation runs in fp32 on exact integers: t < 2^24 (host-checked, Requant::tmax_log2), so
//         q = low byte of med3(max(fma(t, 2^(lk-sh), M), fma(t, 2^-sh, M)), M - 127, M + 127),   M = 1.5 * 2^23
//     is RNE(t' * 2^-sh) clamped, bit for bit the integer pipeline of DESIGN.md section 2: the fma rounds the exact
//     product once, to the integer grid of [2^23, 2^24), ties to even; five VALU operations instead of eight;
//   * the input quantisation uses the same fma: q = low byte of fma(x, 2^sa0, M); three values are packed with two
//     v_perm_b32; clamped inputs are detected from max |x| and handled (and counted exactly) in a cold pass.
// Integer semantics are those of conv1.hip / conv3x3.hip, bit for bit; saturation is detected with one op per output
// and counted exactly (own pixels only) in a cold second pass.
#include "y355_common.h"
#include <type_traits>
#include <cstring>
#include <cmath>
#ifndef FRONT_OCC
#define FRONT_OCC 4
#endif
#ifndef FRONT_P0
#define FRONT_P0 72              // patch pitch in pixels (dwords): 8 mod 64 puts the eight neighbourhood rows a half-wave's
#endif                           // ds_read_b64 touches (4 window rows x 2 lane groups, 8 dwords each) on disjoint banks
#ifndef FRONT_HOTCOLD
#define FRONT_HOTCOLD 1          // 1: the hot pass does not clamp (running max / min of the rounded values detect a clamp; a cold pass
#endif                           //    then rewrites the wave's outputs clamped and counts); 0: clamp + detect per output in the hot pass
#ifndef FRONT_DIAG
#define FRONT_DIAG 0             // 1: s_memtime stamps at the phase boundaries of each workgroup's first tiles (y355_debug_stamps)
#endif

namespace {
constexpr int TOY = 13, TOX = 13;                    // pooled conv2 outputs per tile
constexpr int P1H = 2 * TOY + 2, P1W = 2 * TOX + 2;  // pooled conv1 tile with its halo (windows of conv1)
constexpr int PH0 = 2 * P1H + 2;                     // input patch rows (= columns used)
constexpr int P0 = FRONT_P0;
constexpr int P1P = P1W;                             // p1 pitch in 16-byte pixels (28: rows two apart differ by 8 mod 16 slots)
constexpr int P1ROWS = P1H + 2;                      // slack rows: the clamped padding windows of C2 stay inside
// 784 conv1 windows = 49 groups of 16: 7 x 7 blocks of 4 x 4 windows
constexpr int BW1 = P1W / 4, BH1 = P1H / 4;          // blocks per row / rows of blocks of the window grid
constexpr int NW2 = TOY * TOX;                       // 169 conv2 windows = 11 groups of 16 (7 padding slots)
constexpr int NG2 = (NW2 + 15) / 16;
constexpr int QITEMS = 4;                            // input items (row, 4-pixel group) per thread: 16 wave-items of 4 rows
constexpr float MAGIC = 12582912.0f;                 // 1.5 * 2^23
constexpr float QLO = 12582785.0f, QHI = 12583039.0f;   // MAGIC -+ 127
static_assert(P1H % 4 == 0 && P1W % 4 == 0 && PH0 <= 4 * 4 * QITEMS && P0 % 4 == 0 && P0 >= PH0 + 2, "front geometry");

__device__ __forceinline__ void front_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// bare instructions: hipcc canonicalises (quiets) both operands of fmaxf / fabsf chains
__device__ __forceinline__ float vmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float vmax3abs(float a, float b, float c) {      // max(a, |b|, |c|)
    float d;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// lane i of each row of 16 lanes receives lane i + 1's value (lane 15: zero)
__device__ __forceinline__ unsigned int row_next(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
}
// bytes 0 of four registers -> one dword
__device__ __forceinline__ unsigned int pack4(float a, float b, float c, float d) {
    const unsigned int ab = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x0c0c0400u);
    const unsigned int cd = __builtin_amdgcn_perm(__float_as_uint(d), __float_as_uint(c), 0x04000c0cu);
    return ab | cd;
}
// (r, g, b) bytes 0 -> pixel word (r, g, b, 0)
__device__ __forceinline__ unsigned int pack3(float r, float g, float b) {
    const unsigned int rg = __builtin_amdgcn_perm(__float_as_uint(g), __float_as_uint(r), 0x0c0c0400u);
    return __builtin_amdgcn_perm(__float_as_uint(b), rg, 0x0c040100u);
}

// fp32 form of the epilogue of one layer (see the header): q = low byte of yc
// FOLD (accumulator shift 0, |t| < 2^22, |sh| small: y355_launch_front): the MFMAs' C operand is bias + 0x4B400000, so the
// int32 accumulator IS the bit pattern of the float M + t (no v_cvt), and M + t * s = fma(M + t, s, M * (1 - s)) exactly
// (M * (1 - s) is representable for 2^-22 <= s <= 2^8).
struct RqF {
    float s_pos, s_neg;      // 2^(lk - sh), neg_mul * 2^-sh
    float scl;               // !FOLD: 2^shl
};
template <bool FOLD>
__device__ __forceinline__ RqF make_rqf(const Requant &rq) {
    RqF r;
    // wave-uniform: the two scales wait in SGPRs between the phases that use them
    r.s_pos = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ldexpf(1.0f, rq.lk - rq.sh))));
    r.s_neg = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)rq.neg_mul * ldexpf(1.0f, -rq.sh))));
    r.scl = ldexpf(1.0f, rq.shl);
    return r;
}
// pooled accumulator -> the two branches of the LeakyReLU, each M + rne(t * scale), unclamped.  y = max(pos, neg) (RNE is
// monotone: round(max) = max(round)); with 0 <= s_neg <= s_pos (y355_front_eligible): t >= 0 -> pos >= neg >= M, t < 0 ->
// pos <= neg <= M, so  y > M + 127 <=> pos > M + 127  and  y < M - 127 <=> neg < M - 127.
// The scales and addends are VGPR operands on purpose: an SGPR source takes a vector instruction off the fast issue path
// (scratch/ubench/valu_rates.hip: v_fma_f32 3.0 cycles per SIMD with VGPR sources, 4.6 with one SGPR source).
struct RqV {
    float sp, sn, cp, cn;    // the scales; the addends: FOLD M * (1 - s), else M
};
template <bool FOLD>
__device__ __forceinline__ RqV make_rqv(const RqF &r) {
    RqV v = {r.s_pos, r.s_neg, FOLD ? MAGIC - MAGIC * r.s_pos : MAGIC, FOLD ? MAGIC - MAGIC * r.s_neg : MAGIC};
    asm volatile("" : "+v"(v.sp), "+v"(v.sn), "+v"(v.cp), "+v"(v.cn));
    return v;
}
template <bool FOLD>
__device__ __forceinline__ void rq_pair(int m, float biasf, const RqF &r, const RqV &v, float &pos, float &neg) {
    const float tf = FOLD ? __int_as_float(m) : fmaf((float)m, r.scl, biasf);   // (float)m exact: |t| < 2^24
    pos = fmaf(tf, v.sp, v.cp);
    neg = fmaf(tf, v.sn, v.cn);
}
// byte B of w = bits [7:0] of max(a, b), the other bytes kept (B = 0: zeroed): the LeakyReLU's max and the int8 pack in one
// SDWA instruction per output (the same issue cost as the plain v_max_f32)
template <int B>
__device__ __forceinline__ void max_to_byte(unsigned int &w, float a, float b) {
    if constexpr (B == 0)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 1)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 2)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
}
// The epilogue of four pooled accumulators of one lane -> packed int8 word.
//   hot (CLAMP = false): unclamped low bytes; ymx / ymn track the branches that can leave [-127, 127] (two ops per four outputs each)
//   CLAMP: clamped bytes; nbad = outputs that were clamped
//   NEGSAFE (y355_launch_front: no accumulator the weights allow can drive the negative branch below -127): ymn is not tracked
template <bool FOLD, bool CLAMP, bool NEGSAFE>
__device__ __forceinline__ unsigned int rq_word(const int (&m)[4], const float (&biasf)[4], const RqF &r, const RqV &v, float &ymx,
                                                float &ymn, unsigned int &nbad) {
    float pos[4], neg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rq_pair<FOLD>(m[q], biasf[q], r, v, pos[q], neg[q]);
    unsigned int w;
    if constexpr (!CLAMP) {
        ymx = vmax3(vmax3(ymx, pos[0], pos[1]), pos[2], pos[3]);
        if constexpr (!NEGSAFE) ymn = vmin3(vmin3(ymn, neg[0], neg[1]), neg[2], neg[3]);
        max_to_byte<0>(w, pos[0], neg[0]);
        max_to_byte<1>(w, pos[1], neg[1]);
        max_to_byte<2>(w, pos[2], neg[2]);
        max_to_byte<3>(w, pos[3], neg[3]);
    } else {
        float yc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float y = vmax(pos[q], neg[q]);
            yc[q] = __builtin_amdgcn_fmed3f(y, QLO, QHI);
            nbad += y != yc[q] ? 1u : 0u;
        }
        w = pack4(yc[0], yc[1], yc[2], yc[3]);
    }
    return w;
}
}  // namespace

template <bool U8, bool FOLD, bool NEGSAFE>
__global__ __launch_bounds__(256, FRONT_OCC) void front_kernel(const FrontParams p, const int total_tiles) {
    // separate LDS objects: the compiler then knows that the writes of one phase do not alias the reads of the same phase
    __shared__ __attribute__((aligned(16))) unsigned int patch[PH0 * P0];
    __shared__ __attribute__((aligned(16))) char p1[P1ROWS * P1P * 16];
    __shared__ __attribute__((aligned(16))) char stg[NG2 * 16 * 32];
    __shared__ __attribute__((aligned(16))) unsigned int lut[U8 ? 3 * 256 : 4];
    __shared__ __attribute__((aligned(16))) char wl[U8 ? 16 : 4096 + 64];       // fp32 input: conv1's fragments and biases

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;
    const int Hp = H >> 1, Wp = W >> 1, Ho = H >> 2, Wo = W >> 2;
    const size_t plane = (size_t)H * W;
    const float sc = p.in_scale;
    const float in_thr = p.in_thr;                     // 127.5 / sc: |x| >= thr  <=>  rne(|x| * 2^sa0) > 127 (sc is a power of two)

    // ---- nothing but a few constants stays in registers across phases: the weight fragments and biases (16 KiB + 192 B,
    // L2-resident) are re-read per tile just ahead of the phase that uses them (128 registers per lane at four workgroups per CU)
    const Requant rq1 = p.rq1, rq2 = p.rq2;
    const RqF f1 = make_rqf<FOLD>(rq1), f2 = make_rqf<FOLD>(rq2);
    if constexpr (U8) {
        // normalise + quantise is a function of the byte: per channel a 256-entry table built with the reference's
        // fp32 operations in the reference's order ((u/255 - mean)/std, data/__init__.py:43-45; round(x * 2^sa),
        // slim_yolo_v2.py:35); entry = the int8 value already in the pixel word's byte c, bit 24 = "was clamped"
        // (byte 3 of a pixel multiplies zero weights)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = (float)tid;
            t /= 255.0f;
            t -= p.nmean[c];
            t /= p.nstd[c];
            const float r = rintf(t * sc);
            const float rc = fminf(fmaxf(r, -127.f), 127.f);
            lut[c * 256 + tid] = (((unsigned int)(int)rc & 0xffu) << (8 * c)) | (rc != r ? (1u << 24) : 0u);
        }
    }

    // ---- tile-independent per-thread geometry
    // Q: wave-item q = wave + 4 k covers patch rows 4 q .. 4 q + 3; lane = 16 * (row in the item) + 4-pixel group j
    // (j = 15 is an idle slot: 15 groups = 60 pixels per row are loaded)
    int qr0 = 4 * wave + g;                            // row of item k: qr0 + 16 k
    int qj = li;
    const int G_ = gridDim.x;
    if (p.zero_next != nullptr && blockIdx.x == 0)          // the next forward's counters (CounterSets, y355_common.h): idle until this launch is done
        for (int i = tid; i < p.zero_n; i += 256) p.zero_next[i] = 0ull;
    int tile = y355_xcd_remap(blockIdx.x, G_);
    if (tile >= total_tiles) return;
    int nstamp = 0;
    auto stamp = [&]() {
#if FRONT_DIAG
        if (p.stamps && tid == 0 && nstamp < 32) p.stamps[(size_t)blockIdx.x * 32 + nstamp++] = __builtin_amdgcn_s_memtime();
#endif
    };
    (void)nstamp;
    if constexpr (!U8) {
        *(v4i *)(wl + tid * 16) = *(const v4i *)(p.wf + tid * 16);
        if (tid < 4) *(v4i *)(wl + 4096 + 16 * tid) = *(const v4i *)(p.bias1 + 4 * tid);
    }
    front_lds_barrier();                                // the table / the fragments are complete
    unsigned int nsat_in = 0, nsat1 = 0, nsat2 = 0;

    // ---- a tile's input patch: QITEMS x (3 x float4 | 12 bytes) per thread, all in flight together.  (Issuing the next
    // tile's under C2 of the current one was measured slower in every form -- all of it, half of it, three or four
    // workgroups per CU: profiles/r03_notes.md; the other workgroups of the CU cover the wait.)
    float4 vf[U8 ? 1 : QITEMS][3];
    uint3 vu[U8 ? QITEMS : 1];
    // SAFE: every row / 4-pixel group the loads touch lies inside the image (interior tiles): no clamps, no zero padding
    auto load_input = [&](int tx, int ty, int b, auto safec) {
        constexpr bool SAFE = decltype(safec)::value;
        const int y0p = 4 * TOY * ty - 3, x0p = 4 * TOX * tx - 4;
        // wave-uniform bases (SGPR pairs) + 32-bit lane offsets: the loads take the saddr form, no 64-bit vector arithmetic
        const char *bu8 = (const char *)p.x_u8 + (size_t)b * plane * 3;
        const char *bf[3] = {(const char *)(p.x + ((size_t)b * 3 + 0) * plane), (const char *)(p.x + ((size_t)b * 3 + 1) * plane),
                             (const char *)(p.x + ((size_t)b * 3 + 2) * plane)};
#pragma unroll
        for (int k = 0; k < QITEMS; ++k) {
            const int r = qr0 + 16 * k;
            int gy = y0p + r, gx = x0p + 4 * qj;
            if constexpr (!SAFE) {
                gy = min(max(gy, 0), H - 1);                  // rows / groups past the patch or the image re-read valid data
                gx = min(max(gx, 0), W - 4);
            }
            const unsigned int o = (unsigned int)(gy * W + gx);
            if constexpr (U8) {
                vu[k] = *(const uint3 *)(bu8 + o * 3u);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) vf[k][c] = *(const float4 *)(bf[c] + o * 4u);
            }
        }
    };
    using TT = std::true_type;
    using FF = std::false_type;

    // tile -> (tx, ty, b) ONCE; every further tile of this workgroup is G_ tiles on, and the host has split G_ into the same
    // mixed radix (p.step_x / step_y / step_b): three carries per tile instead of three integer divisions by run-time values
    // (round 6: the divisions and what hung off them were ~85 vector + ~175 scalar instructions per tile and wave, 13 % of the
    // kernel's issue time -- profiles/r06_notes.md)
    int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    for (int trip = 0;; tile += G_, ++trip) {
        // the per-thread bases are made opaque once per tile: otherwise every address derived from them is hoisted
        // out of the tile loop as a loop invariant and held in registers
        int li_ = li, g_ = g, lane_ = lane, tid_ = tid;
        asm volatile("" : "+v"(qr0), "+v"(qj), "+v"(li_), "+v"(g_), "+v"(lane_), "+v"(tid_));
        const int y0p = 4 * TOY * ty - 3, x0p = 4 * TOX * tx - 4;
        const bool border = ty == 0 || tx == 0 || ty == p.tiles_y - 1 || tx == p.tiles_x - 1;
        // the loads of all QITEMS items (rows y0p .. y0p + 63, columns x0p .. x0p + 63) stay inside the image
        const bool qsafe = y0p >= 0 && x0p >= 0 && y0p + 16 * QITEMS <= H && x0p + 64 <= W;
        stamp();

        if (qsafe) load_input(tx, ty, b, TT{});
        else load_input(tx, ty, b, FF{});
#if FRONT_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();
#endif

        // ---- Q: quantise into the LDS patch of 4-byte pixels (r, g, b, 0): q = clamp(rne(x * 2^sa0))  (slim_yolo_v2.py:33-35).
        // Patch column L holds global column x0p + 1 + L: the 4x4 neighbourhood of every conv1 window then starts on an
        // 8-byte boundary (one pixel to the left of the aligned 4-pixel groups the loads use: the fourth word of an
        // LDS group comes from the next lane).
        auto quantise = [&](auto clampc, auto safec) {
            constexpr bool CLAMP = decltype(clampc)::value;      // cold: clamp (and count the tile's own clamped values)
            constexpr bool SAFE = decltype(safec)::value;
            float am = 0.f;
            unsigned int sato = 0;
            float scv = sc, mg = MAGIC;                          // VGPR operands: the fast issue path of v_fma_f32
            asm volatile("" : "+v"(scv), "+v"(mg));
#pragma unroll
            for (int k = 0; k < QITEMS; ++k) {
                const int r = qr0 + 16 * k;
                unsigned int w[4];
                if constexpr (U8) {
                    const unsigned int d[3] = {vu[k].x, vu[k].y, vu[k].z};
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        // pixel px = bytes 3 px .. 3 px + 2 (B, G, R); RGB channel c = BGR byte 2 - c
                        unsigned int e = 0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int bi = 3 * px + (2 - c);
                            const unsigned int u = (d[bi >> 2] >> (8 * (bi & 3))) & 0xffu;
                            e |= lut[c * 256 + u];
                        }
                        sato |= e;
                        w[px] = e & 0x00ffffffu;
                    }
                } else {
                    if constexpr (!CLAMP) {                          // max |x| of the item's twelve values: two per instruction
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            am = vmax3abs(am, vf[k][c].x, vf[k][c].y);
                            am = vmax3abs(am, vf[k][c].z, vf[k][c].w);
                        }
                    }
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        const float xr = px == 0 ? vf[k][0].x : px == 1 ? vf[k][0].y : px == 2 ? vf[k][0].z : vf[k][0].w;
                        const float xg = px == 0 ? vf[k][1].x : px == 1 ? vf[k][1].y : px == 2 ? vf[k][1].z : vf[k][1].w;
                        const float xb = px == 0 ? vf[k][2].x : px == 1 ? vf[k][2].y : px == 2 ? vf[k][2].z : vf[k][2].w;
                        float yr = fmaf(xr, scv, mg), yg = fmaf(xg, scv, mg), yb = fmaf(xb, scv, mg);
                        if constexpr (CLAMP) {
                            yr = __builtin_amdgcn_fmed3f(yr, QLO, QHI);
                            yg = __builtin_amdgcn_fmed3f(yg, QLO, QHI);
                            yb = __builtin_amdgcn_fmed3f(yb, QLO, QHI);
                        }
                        w[px] = pack3(yr, yg, yb);
                    }
                }
                const int gy = y0p + r, gx = x0p + 4 * qj;
                const bool inside = SAFE || ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W);
                if constexpr (!SAFE) {
                    const bool zero = border && !inside;          // pixels outside the image are conv1's zero padding
#pragma unroll
                    for (int px = 0; px < 4; ++px) w[px] = zero ? 0u : w[px];
                }
                v4i wv;
                wv[0] = (int)w[1];
                wv[1] = (int)w[2];
                wv[2] = (int)w[3];
                wv[3] = (int)row_next(w[0]);
                if (r < PH0 && qj < 15) *(v4i *)(patch + r * P0 + 4 * qj) = wv;
                if constexpr (CLAMP) {
                    // exact count over the pixels this tile OWNS (rows [3, 3 + 4 TOY), load groups [1, TOX] of the
                    // patch: the tiles' exclusive input areas partition the image)
                    const bool own = inside && r >= 3 && r < 3 + 4 * TOY && qj >= 1 && qj <= TOX;
                    if constexpr (U8) {
                        const unsigned int d[3] = {vu[k].x, vu[k].y, vu[k].z};
#pragma unroll
                        for (int bi = 0; bi < 12; ++bi) {
                            const unsigned int u = (d[bi >> 2] >> (8 * (bi & 3))) & 0xffu;
                            nsat_in += (own && (lut[(2 - bi % 3) * 256 + u] >> 24)) ? 1u : 0u;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float xs[4] = {vf[k][c].x, vf[k][c].y, vf[k][c].z, vf[k][c].w};
#pragma unroll
                            for (int px = 0; px < 4; ++px) nsat_in += (own && !(fabsf(xs[px]) < in_thr)) ? 1u : 0u;
                        }
                    }
                }
            }
            return U8 ? (sato >> 24) != 0u : !(am < in_thr);          // also true for NaN / Inf inputs
        };
        const bool qsat = qsafe ? quantise(FF{}, TT{}) : quantise(FF{}, FF{});
        if (__builtin_amdgcn_ballot_w64(qsat) != 0ull) {
            load_input(tx, ty, b, FF{});                      // cold: the input registers were given up after the hot pass
            (void)quantise(TT{}, FF{});
        }
        // conv1: weight variant (dy, dx) = v >> 1, v & 1 and the biases as the MFMAs' C operand (accumulator register r of lane
        // (li, g) = channel 4 g + r).  fp32 input: from the copy in LDS (the input registers leave no room to hold them
        // through Q at 128 registers per lane); uint8 input (the table takes the LDS): from global memory, in flight across
        // the barrier
        v4i w1[4], b1v;
        if constexpr (U8) {
#pragma unroll
            for (int v = 0; v < 4; ++v) w1[v] = *(const v4i *)(p.wf + v * 1024 + lane_ * 16);
            b1v = *(const v4i *)(p.bias1 + 4 * g_);
        }
        stamp();
        front_lds_barrier();                              // B1: patch complete
        stamp();
        if constexpr (!U8) {
#pragma unroll
            for (int v = 0; v < 4; ++v) w1[v] = *(const v4i *)(wl + v * 1024 + lane_ * 16);
            b1v = *(const v4i *)(wl + 4096 + 16 * g_);
        }
        v4i cin1;
        float bf1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cin1[r] = FOLD ? b1v[r] + 0x4B400000 : 0;
            bf1[r] = (float)b1v[r];
        }

        // conv2 fragments [n-tile][filter row][dx]: issued behind C1, they land under the barrier (held through C1 they would
        // take the registers that let two blocks of C1 be in flight)
        v4i w2[2][3][2];
        // ---- C1: conv1 + pool1 -> p1.  Group = one 4 x 4 block of the 28 x 28 window grid (7 x 7 blocks).  A wave owns the block
        // rows wrot and wrot + 4 (wrot = 3: one row only) and walks them left to right: with the row folded into the lane's base
        // addresses, every block's LDS addresses are base + compile-time immediate -- no address arithmetic per block, vector
        // or scalar (scalar instructions take issue slots too: scratch/ubench/valu_issue.hip).  wrot rotates with the
        // workgroup's tile count, so over four tiles every wave (= every SIMD of the CU) computes 49 blocks.
        // Lane (li, g): window (li >> 2, li & 3) of the block, neighbourhood row g.
        const int gyp0 = 2 * TOY * ty - 1, gxp0 = 2 * TOX * tx - 1;     // pooled coordinates of window (0, 0)
        const int wrot = (wave + trip) & 3;
        const int wyl = li_ >> 2, wxl = li_ & 3;
        const int lsrc = (2 * wyl + g_) * P0 + 2 * wxl + 8 * P0 * wrot;              // dwords into `patch`
        const int ldst = ((wyl + 4 * wrot) * P1P + wxl) * 16 + 4 * g_;               // bytes into `p1`
        // Passes (COLD = false / true).  FRONT_HOTCOLD: hot = round, pack and write UNCLAMPED, tracking the running max / min of
        // the rounded values (two ops per four outputs); when they leave [-127, 127] (rare) the cold pass rewrites this
        // wave's blocks clamped and counts the tile's own clamped outputs.  Otherwise: hot = clamp + detect per output,
        // cold = count only.  Windows outside the image (conv2's zero padding; tiles on the image's edge only) are zeroed by
        // the wave that wrote them, behind its hot pass.
        auto c1 = [&](auto coldc) {
            constexpr bool COLD = decltype(coldc)::value;
            constexpr bool CLAMP = COLD || !FRONT_HOTCOLD, WRITE = !COLD || FRONT_HOTCOLD;
            unsigned int satx = 0;
            float ymx = MAGIC, ymn = MAGIC;
            const RqV v1 = make_rqv<FOLD>(f1);
            auto body = [&](int r2, int bx) {                      // hot: compile-time (unrolled); cold: scalars
                // two ds_read_b64 (2 LDS cycles each, 64 banks, 32-lane groups: conflict-free with P0 = 8 mod 64), not the
                // ds_read2_b64 the compiler merges plain loads into (8 cycles, 32 banks, 16-lane groups: 2-way conflicts on top)
                const unsigned int *src = patch + lsrc + (32 * P0 * r2 + 8 * bx);
                typedef const volatile unsigned long long __attribute__((address_space(3))) lds_cv64;
                const unsigned long long lo = *(lds_cv64 *)src, hi = *(lds_cv64 *)(src + 2);
                const v4i bq = {(int)(unsigned int)lo, (int)(unsigned int)(lo >> 32), (int)(unsigned int)hi, (int)(unsigned int)(hi >> 32)};
                v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1[0], bq, cin1, 0, 0, 0);
                v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1[1], bq, cin1, 0, 0, 0);
                v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1[2], bq, cin1, 0, 0, 0);
                v4i a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1[3], bq, cin1, 0, 0, 0);
                int m[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) m[r] = max(max(a0[r], a1[r]), max(a2[r], a3[r]));
                unsigned int nbad = 0;
                unsigned int word = rq_word<FOLD, CLAMP, NEGSAFE>(m, bf1, f1, v1, ymx, ymn, nbad);
                if constexpr (!COLD) {
                    if constexpr (!FRONT_HOTCOLD) satx += nbad;
                } else {
                    const int py = 4 * (wrot + 4 * r2) + wyl, px = 4 * bx + wxl;
                    const bool inimg = (unsigned)(gyp0 + py) < (unsigned)Hp && (unsigned)(gxp0 + px) < (unsigned)Wp;
                    const bool own = inimg && py >= 1 && py < P1H - 1 && px >= 1 && px < P1W - 1;
                    satx += own ? nbad : 0u;
                }
                if constexpr (WRITE) *(unsigned int *)(p1 + ldst + (16 * P1P * r2 + 4 * bx) * 16) = word;
            };
            if constexpr (!COLD) {
                // two blocks between scheduling fences: one block's epilogue runs under the other's MFMAs, and the live ranges
                // of the straight-line code stay inside the register budget
#pragma unroll
                for (int bx = 0; bx < BW1; ++bx) {
                    body(0, bx);
                    if (bx & 1) __builtin_amdgcn_sched_barrier(0);
                }
                if (wrot + 4 < BH1) {
#pragma unroll
                    for (int bx = 0; bx < BW1; ++bx) {
                        body(1, bx);
                        if (bx & 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll 1
                for (int r2 = 0; r2 < 2; ++r2) {
                    if (wrot + 4 * r2 >= BH1) break;
#pragma unroll 1
                    for (int bx = 0; bx < BW1; ++bx) body(r2, bx);
                }
            }
            if constexpr (!COLD && FRONT_HOTCOLD) satx = (ymx > QHI || (!NEGSAFE && ymn < QLO)) ? 1u : 0u;
            return satx;
        };
        if (__builtin_amdgcn_ballot_w64(c1(FF{}) != 0) != 0ull) nsat1 += c1(TT{});
        if (border) {
            // the wave's own windows (block rows wrot, wrot + 4: 8 x 28) that lie outside the image -> zero (same wave, same LDS
            // queue: ordered behind the writes above)
            v4i zero4 = {0, 0, 0, 0};
            asm volatile("" : "+v"(zero4));                        // materialised here, not held across the tile loop
#pragma unroll 1
            for (int idx = lane_; idx < 8 * P1W; idx += 64) {
                const int rl = (idx * 2341) >> 16;                    // idx / 28 for idx < 784
                const int px = idx - rl * P1W;
                const int py = 4 * wrot + (rl & 3) + 16 * (rl >> 2);
                const bool inimg = (unsigned)(gyp0 + py) < (unsigned)Hp && (unsigned)(gxp0 + px) < (unsigned)Wp;
                if (py < P1H && !inimg) *(v4i *)(p1 + (py * P1P + px) * 16) = zero4;
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    w2[n][ky][dx] = *(const v4i *)(p.wf + 4096 + ((3 * n + ky) * 2 + dx) * 1024 + lane_ * 16);
        const v4i b2v[2] = {*(const v4i *)(p.bias2 + 8 * g_), *(const v4i *)(p.bias2 + 8 * g_ + 4)};   // channel 8 g + 4 n + r
        stamp();
        front_lds_barrier();                              // B2: p1 complete
        stamp();

        // ---- C2: conv2 + pool2 -> staged int8 tile.  Group = 16 consecutive windows of the 13 x 13 grid (the last group's
        // padding slots repeat window 168); k-step t = neighbourhood row t, lane group g = neighbourhood column.  One set of LDS
        // reads feeds both n-tiles (16 output channels each: channel 8 g + 4 n + r in register r of lane group g), whose
        // eight int8 results leave as one ds_write_b64.
        auto c2 = [&](auto coldc) {
            constexpr bool COLD = decltype(coldc)::value;
            constexpr bool CLAMP = COLD || !FRONT_HOTCOLD, WRITE = !COLD || FRONT_HOTCOLD;
            unsigned int satx = 0;
            float ymx = MAGIC, ymn = MAGIC;
            const RqV v2 = make_rqv<FOLD>(f2);
            v4i cin2[2];
            float bf2[2][4];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cin2[n][r] = FOLD ? b2v[n][r] + 0x4B400000 : 0;
                    bf2[n][r] = (float)b2v[n][r];
                }
#pragma unroll 1
            for (int grp = wave; grp < NG2; grp += 4) {
                const int wraw = grp * 16 + li_;
                const int w = min(wraw, NW2 - 1);
                const int wy = (w * 5042) >> 16;                  // w / 13 for w < 169
                const int wx = w - wy * TOX;
                const char *src = p1 + ((2 * wy) * P1P + 2 * wx + g_) * 16;
                v4i bq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bq[t] = *(const v4i *)(src + t * P1P * 16);
                unsigned int word[2];
                unsigned int nbad = 0;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    v4i acc[2][2];                                // [dy][dx]
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) acc[dy][dx] = cin2[n];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) {
                            const int ky = t - dy;
                            if (ky < 0 || ky > 2) continue;
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                acc[dy][dx] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2[n][ky][dx], bq[t], acc[dy][dx], 0, 0, 0);
                        }
                    int m[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = max(max(acc[0][0][r], acc[0][1][r]), max(acc[1][0][r], acc[1][1][r]));
                    word[n] = rq_word<FOLD, CLAMP, NEGSAFE>(m, bf2[n], f2, v2, ymx, ymn, nbad);
                }
                if constexpr (!COLD) {
                    if constexpr (!FRONT_HOTCOLD) satx += nbad;
                } else {
                    const bool own = wraw < NW2 && TOY * ty + wy < Ho && TOX * tx + wx < Wo;
                    satx += own ? nbad : 0u;
                }
                if constexpr (WRITE) *(uint2 *)(stg + wraw * 32 + 8 * g_) = make_uint2(word[0], word[1]);
            }
            if constexpr (!COLD && FRONT_HOTCOLD) satx = (ymx > QHI || (!NEGSAFE && ymn < QLO)) ? 1u : 0u;
            return satx;
        };
        if (__builtin_amdgcn_ballot_w64(c2(FF{}) != 0) != 0ull) nsat2 += c2(TT{});
        stamp();
        front_lds_barrier();                              // B3: staged tile complete
        stamp();

        // ---- OUT: NHWC32 with halo, 16 bytes per thread and item (item = 2 * window + half)
        {
            int8_t *outb = p.out + (((size_t)b * (Ho + 2) + TOY * ty + 1) * (Wo + 2) + TOX * tx + 1) * 32;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int item = tid_ + 256 * k;
                const int wdw = item >> 1;
                const int row = (wdw * 5042) >> 16, col = wdw - row * TOX;        // wdw / 13 for wdw < 256
                if (item < NW2 * 2 && TOY * ty + row < Ho && TOX * tx + col < Wo)
                    *(v4i *)(outb + (row * (Wo + 2) + col) * 32 + (item & 1) * 16) = *(const v4i *)(stg + item * 16);
            }
        }
        if (tile + G_ >= total_tiles) break;
        tx += p.step_x;
        if (tx >= p.tiles_x) { tx -= p.tiles_x; ++ty; }
        ty += p.step_y;
        if (ty >= p.tiles_y) { ty -= p.tiles_y; ++b; }
        b += p.step_b;
        // the next tile's Q phase writes `patch` (last read before B2) and its C1 writes `p1` (last read before B3):
        // both are behind a barrier every wave has passed; `stg` is rewritten only after B2 of the next tile
    }
    if (nsat_in) atomicAdd(&p.ctr[0].in_sat, (unsigned long long)nsat_in);
    if (nsat1) atomicAdd(&p.ctr[0].sat, (unsigned long long)nsat1);
    if (nsat2) atomicAdd(&p.ctr[1].sat, (unsigned long long)nsat2);
}

void y355_front_tiles(int H, int W, int *tx, int *ty) {
    *tx = (W / 4 + TOX - 1) / TOX;
    *ty = (H / 4 + TOY - 1) / TOY;
}

// Weight fragments of the fused front end (16 KiB): MFMA A operands, lane (i = l & 15: accumulator row, g = l >> 4), 16 bytes.
//   conv1, variant v = 2 dy + dx (offset of the conv output inside the pooling window), at v * 1024:
//     row i = output channel i; g = neighbourhood row; byte 4 nc + c = w[i][c][g - dy][nc - dx] (zero outside the 3x3 filter)
//   conv2, fragment ((n * 3 + ky) * 2 + dx) at 4096 + ... * 1024:
//     row i = output channel 8 (i >> 2) + 4 n + (i & 3); g = neighbourhood column; byte ci = w[ch][ci][ky][g - dx]
void y355_pack_front(const int8_t *q_w1 /*[16][3][3][3]*/, const int8_t *q_w2 /*[32][16][3][3]*/, int8_t *dst /*16384*/) {
    memset(dst, 0, 16384);
    if (q_w1) {
        for (int v = 0; v < 4; ++v)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, g = l >> 4, ky = g - (v >> 1);
                for (int kk = 0; kk < 16; ++kk) {
                    const int kx = (kk >> 2) - (v & 1), c = kk & 3;
                    if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3 && c < 3)
                        dst[v * 1024 + l * 16 + kk] = q_w1[((i * 3 + c) * 3 + ky) * 3 + kx];
                }
            }
    }
    if (q_w2) {
        for (int n = 0; n < 2; ++n)
            for (int ky = 0; ky < 3; ++ky)
                for (int dx = 0; dx < 2; ++dx)
                    for (int l = 0; l < 64; ++l) {
                        const int i = l & 15, g = l >> 4, kx = g - dx;
                        const int ch = 8 * (i >> 2) + 4 * n + (i & 3);
                        if (kx < 0 || kx > 2) continue;
                        for (int ci = 0; ci < 16; ++ci)
                            dst[4096 + ((n * 3 + ky) * 2 + dx) * 1024 + l * 16 + ci] = q_w2[((ch * 16 + ci) * 3 + ky) * 3 + kx];
                    }
    }
}

// true when the fused launch covers these two layers: 32-bit epilogues whose t stays below 2^24 (exact in fp32)
// and a LeakyReLU slope in [0, 1] (the epilogue takes max(t * s_pos, t * s_neg) and reads a clamp off the branch that can reach it)
bool y355_front_eligible(const Requant &rq1, const Requant &rq2) {
    auto slope_ok = [](const Requant &rq) { return rq.neg_mul >= 0 && rq.neg_mul <= (1 << rq.lk); };
    return !rq1.wide && !rq2.wide && rq1.tmax_log2 <= 24 && rq2.tmax_log2 <= 24 && slope_ok(rq1) && slope_ok(rq2);
}

void y355_launch_front(const FrontParams &p, hipStream_t s) {
    const int total = p.tiles_x * p.tiles_y * p.B;
#ifndef FRONT_GRID
#define FRONT_GRID (y355_cu_count() * FRONT_OCC)
#endif
    int grid = FRONT_GRID;
    if (grid > total) grid = total;
    auto foldable = [](const Requant &rq) { return rq.shl == 0 && rq.tmax_log2 <= 22 && rq.sh <= 22 && rq.sh - rq.lk >= -8; };
    const bool fold = foldable(p.rq1) && foldable(p.rq2);
    FrontParams q = p;
    q.ev_start = q.ev_stop = nullptr;
    q.in_thr = 127.5f / p.in_scale;
    q.step_x = grid % p.tiles_x;                       // the walk's stride (one grid) in the tile index's mixed radix
    q.step_y = (grid / p.tiles_x) % p.tiles_y;
    q.step_b = grid / (p.tiles_x * p.tiles_y);
    // no accumulator these weights allow (|t| < 2^tmax_log2, Requant) takes the LeakyReLU's negative branch below -127.5: the hot
    // passes then do not track its minimum (2 of the 24 vector instructions per four outputs; the positive branch, 8 x steeper,
    // is always tracked).  True for conv1 / conv2 of the benchmark fixture and of the reference's trained exponents.
    const bool ns = p.rq1.negsafe && p.rq2.negsafe;                  // Requant::negsafe: the host's check on the exact bound
#define FRONT_GO(U8_, FOLD_, NS_) Y355_LAUNCH((front_kernel<U8_, FOLD_, NS_>), dim3(grid), dim3(256), 0, s, p.ev_start, p.ev_stop, q, total)
    if (p.x) {
        if (fold) { if (ns) FRONT_GO(false, true, true); else FRONT_GO(false, true, false); }
        else { if (ns) FRONT_GO(false, false, true); else FRONT_GO(false, false, false); }
    } else {
        if (fold) { if (ns) FRONT_GO(true, true, true); else FRONT_GO(true, true, false); }
        else { if (ns) FRONT_GO(true, false, true); else FRONT_GO(true, false, false); }
    }
#undef FRONT_GO
}
