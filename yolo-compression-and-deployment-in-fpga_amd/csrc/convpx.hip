// yolo355 -- conv3_1 (32 -> 64 channels on the quarter-resolution map, models/slim_yolo_v2.py:246-256; the FPGA's
// third conv_normal call, c_embedding/yolo_forward.c:1214-1218): production kernel, round 3.
//
// K = 9 taps x 32 channels = 288 is too shallow for the deep layers' ring kernel (its per-tile start-up and its staged
// two-pass epilogue were the kernel: 31.8 us for 5 us of MFMA work and 66 MB of compulsory traffic).
// This one turns the GEMM round, as the fused front end does (front.hip):
//   * the WEIGHTS are the MFMA's A operand (rows = output channels) and stay in registers for the whole launch
//     (5 k-steps x 4 n-tiles x 4 VGPRs = 80); the PIXELS are the B operand (columns = 16 consecutive pixels of a row),
//     read from an LDS slab with one ds_read_b128 per k-step (k-step = two taps x 32 channels; 16 pixels at a 32-byte
//     pitch are conflict-free under gfx950's 4 x 16 lane grouping);
//   * output channel 16 g + 4 n + r sits in register r of n-tile n of lane group g (weight rows are permuted so on the
//     host), so a lane ends up with 16 CONSECUTIVE channels of ONE pixel: one packed global_store_dwordx4 per lane,
//     1 KiB contiguous per wave-instruction -- no LDS staging, no second pass;
//   * the input slab of a tile (TH + 2 whole padded rows) is ONE contiguous range of the NHWC32 buffer with its zero
//     halo: a linear LDS-DMA copy (global_load_lds_dwordx4), double-buffered across the tiles a workgroup walks;
//   * the epilogue is front.hip's fp32 form (two fma + max + med3 on exact integers; FOLD: bias + 0x4B400000 as the
//     MFMAs' C operand).
// Integer semantics: DESIGN.md section 2, bit for bit those of conv3x3_v2.hip / conv3x3.hip.
#include "y355_common.h"
#include <cstring>
#include <type_traits>

#ifndef PX_ABL
#define PX_ABL 0                 // timing ablations (WRONG RESULTS): 1 no stores, 2 no epilogue arithmetic, 4 no MFMAs, 8 no LDS reads
#endif
#ifndef PX_DIAG
#define PX_DIAG 0                // 1: s_memrealtime stamps (100 MHz) per workgroup at the phase boundaries (y355_debug_stamps, layer 2)
#endif

namespace {
constexpr int CIN = 32, COUT = 64, KS = 5, NTN = 4;
constexpr int PXB = CIN;                             // bytes per input pixel
constexpr float MAGIC = 12582912.0f;                 // 1.5 * 2^23
constexpr float QLO = 12582785.0f, QHI = 12583039.0f;

__device__ __forceinline__ void pglds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
__device__ __forceinline__ float pvmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float pvmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float pvmin3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned int ppack4(float a, float b, float c, float d) {
    const unsigned int ab = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x0c0c0400u);
    const unsigned int cd = __builtin_amdgcn_perm(__float_as_uint(d), __float_as_uint(c), 0x04000c0cu);
    return ab | cd;
}
}  // namespace

// TH rows x the whole map width per tile; NW waves; groups of 16 consecutive (row-major) pixels of the tile, wave w owns
// groups w, w + NW, ...
template <int TH, int NW, bool FOLD>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void convpx32_kernel(const ConvParams p, const int total_tiles, const int slab_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // two slabs of slab_bytes (a multiple of 1 KiB)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;
    const int PW = W + 2;
    const int tiles_y = p.tiles_y;

    // ---- weights: A fragments [k-step][n-tile], registers for the whole launch
    v4i wf[KS][NTN];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int n = 0; n < NTN; ++n) wf[ks][n] = *(const v4i *)(p.w + (ks * NTN + n) * 1024 + lane * 16);
    // accumulator register r of n-tile n of lane group g = channel 16 g + 4 n + r
    v4i cin[NTN];
    float bf[NTN][4];
#pragma unroll
    for (int n = 0; n < NTN; ++n) {
        const v4i bv = *(const v4i *)(p.bias_t + 16 * g + 4 * n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cin[n][r] = FOLD ? bv[r] + 0x4B400000 : 0;
            bf[n][r] = (float)bv[r];
        }
    }
    const Requant rq = p.rq;
    const float s_pos = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ldexpf(1.0f, rq.lk - rq.sh))));
    const float s_neg = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)rq.neg_mul * ldexpf(1.0f, -rq.sh))));
    const float c_pos = FOLD ? MAGIC - MAGIC * s_pos : MAGIC, c_neg = FOLD ? MAGIC - MAGIC * s_neg : MAGIC;
    const float scl = ldexpf(1.0f, rq.shl);
    // B operand of k-step ks: tap 2 ks + (g >> 1), channel half g & 1 (tap 9 multiplies zero weights: it re-reads tap 8)
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tap = min(2 * ks + (g >> 1), 8);
        koff[ks] = ((tap / 3) * PW + tap % 3) * PXB + 16 * (g & 1);
    }
    const float invW = 1.0f / (float)W;

    auto tile_rows = [&](int t, int &b, int &y0) {
        b = t / tiles_y;
        y0 = (t - b * tiles_y) * TH;
        return min(TH, H - y0);
    };
    // slab of tile t = padded rows y0 .. y0 + rows + 1, contiguous in the input buffer; whole 1 KiB pieces, piece q by wave q % NW
    auto issue_slab = [&](int t, int buf) {
        int b, y0;
        const int rows = tile_rows(t, b, y0);
        const int8_t *src = p.in + ((size_t)b * (H + 2) + y0) * PW * PXB;
        const int npiece = ((rows + 2) * PW * PXB + 1023) >> 10;
        for (int q = wave; q < npiece; q += NW) pglds16(src + (size_t)q * 1024 + lane * 16, smem + buf * slab_bytes + q * 1024);
    };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int nstamp = 0;
    auto stamp = [&]() {
#if PX_DIAG
        if (p.stamps && tid == 0 && nstamp < 32) p.stamps[(size_t)blockIdx.x * 32 + nstamp++] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    (void)nstamp;
    stamp();
    issue_slab(tile, 0);
    unsigned int nsat = 0;
    int buf = 0;
    for (;; tile += gridDim.x, buf ^= 1) {
        // the slab of this tile has landed (and the previous tile's stores have drained); every wave is past its reads of the
        // other buffer, which the next tile's slab now overwrites
        stamp();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        const bool more = tile + (int)gridDim.x < total_tiles;
        if (more) issue_slab(tile + gridDim.x, buf ^ 1);
        int b, y0;
        const int rows = tile_rows(tile, b, y0);
        const int npix = rows * W;
        const int ngrp = (npix + 15) >> 4;
        const char *slab = smem + buf * slab_bytes;
        int8_t *outb = p.out + (((size_t)b * (H + 2) + y0 + 1) * PW + 1) * COUT;      // wave-uniform; the lane's part is a 32-bit offset

        // Hot pass: round, pack and store UNCLAMPED, tracking the running max / min of the rounded values (two ops per four
        // outputs).  When they leave [-127, 127] (rare) the cold pass recomputes this wave's groups from the slab, which is still
        // in LDS, stores them clamped and counts the clamped outputs of real pixels.
        float ymx = MAGIC, ymn = MAGIC;
        auto body = [&](int grp, auto coldc) {
            constexpr bool COLD = decltype(coldc)::value;
            const int pr = grp * 16 + li;
            const int pc = min(pr, npix - 1);                      // padding lanes of the last group repeat its last pixel
            const int py = (int)(((float)pc + 0.5f) * invW);       // pc / W (exact: pc < 2^16)
            const int px = pc - py * W;
            const char *src = slab + (py * PW + px) * PXB;
            v4i acc[NTN];
#pragma unroll
            for (int n = 0; n < NTN; ++n) acc[n] = cin[n];
            v4i bq[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (PX_ABL & 8) bq[ks] = (v4i){py, px, ks, grp};
                else bq[ks] = *(const v4i *)(src + koff[ks]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int n = 0; n < NTN; ++n) {
                    if constexpr (PX_ABL & 4) asm volatile("" : "+v"(acc[n]) : "v"(wf[ks][n]), "v"(bq[ks]));
                    else acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[ks][n], bq[ks], acc[n], 0, 0, 0);
                }
            v4i word;
#pragma unroll
            for (int n = 0; n < NTN; ++n) {
                float y[4], yc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tf = FOLD ? __int_as_float(acc[n][r]) : fmaf((float)acc[n][r], scl, bf[n][r]);
                    if constexpr (PX_ABL & 2) y[r] = tf;
                    else y[r] = pvmax(fmaf(tf, s_pos, c_pos), fmaf(tf, s_neg, c_neg));
                    yc[r] = COLD ? __builtin_amdgcn_fmed3f(y[r], QLO, QHI) : y[r];
                    if constexpr (COLD) nsat += (pr < npix && y[r] != yc[r]) ? 1u : 0u;
                }
                if constexpr (!COLD && !(PX_ABL & 2)) {
                    ymx = pvmax3(pvmax3(ymx, y[0], y[1]), y[2], y[3]);
                    ymn = pvmin3(pvmin3(ymn, y[0], y[1]), y[2], y[3]);
                }
                if constexpr (PX_ABL & 2) word[n] = (int)(__float_as_uint(yc[0]) ^ __float_as_uint(yc[1]) ^ __float_as_uint(yc[2]) ^ __float_as_uint(yc[3]));
                else word[n] = (int)ppack4(yc[0], yc[1], yc[2], yc[3]);
            }
            if constexpr (PX_ABL & 1) asm volatile("" :: "v"(word));
            else if (pr < npix) *(v4i *)(outb + ((py * PW + px) * COUT + 16 * g)) = word;
        };
        {
            int grp = wave;
#pragma unroll 1
            for (; grp + NW < ngrp; grp += 2 * NW) {               // two groups per trip: one's epilogue under the other's MFMAs
                body(grp, std::false_type{});
                body(grp + NW, std::false_type{});
            }
            if (grp < ngrp) body(grp, std::false_type{});
        }
        if (__builtin_amdgcn_ballot_w64(ymx > QHI || ymn < QLO) != 0ull) {
#pragma unroll 1
            for (int grp = wave; grp < ngrp; grp += NW) body(grp, std::true_type{});
        }
        stamp();
        if (!more) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
}

// A fragments of conv3_1 for convpx32_kernel (20 KiB): fragment (ks * 4 + n), lane (i = l & 15, g = l >> 4), 16 bytes:
// row i = output channel 16 (i >> 2) + 4 n + (i & 3); k = tap 2 ks + (g >> 1), input channels 16 (g & 1) .. + 15
void y355_pack_px32(const int8_t *q_w /*[64][32][3][3]*/, int8_t *dst /*20480*/) {
    memset(dst, 0, KS * NTN * 1024);
    for (int ks = 0; ks < KS; ++ks)
        for (int n = 0; n < NTN; ++n)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, g = l >> 4, tap = 2 * ks + (g >> 1);
                const int ch = 16 * (i >> 2) + 4 * n + (i & 3);
                if (tap > 8) continue;
                for (int kk = 0; kk < 16; ++kk) {
                    const int ci = 16 * (g & 1) + kk;
                    dst[(ks * NTN + n) * 1024 + l * 16 + kk] = q_w[((size_t)ch * CIN + ci) * 9 + tap];
                }
            }
}

namespace {
#ifndef PX_NW_
#define PX_NW_ 8
#endif
constexpr int PX_TH = 13, PX_NW = PX_NW_;
size_t px_slab_bytes(int W) { return ((size_t)(PX_TH + 2) * (W + 2) * PXB + 1023) / 1024 * 1024; }
template <bool FOLD>
void px_launch(const ConvParams &p_in, hipStream_t s) {
    ConvParams p = p_in;
    p.tiles_y = (p.H + PX_TH - 1) / PX_TH;
    p.ev_start = p.ev_stop = nullptr;
    const int total = p.tiles_y * p.B;
    const int slab = (int)px_slab_bytes(p.W);
    int grid = 256;                                         // one 8-wave workgroup per CU; at B = 64, 416 x 416: two tiles each
    if (grid > total) grid = total;
    Y355_LAUNCH((convpx32_kernel<PX_TH, PX_NW, FOLD>), dim3(grid), dim3(PX_NW * 64), 2 * (size_t)slab, s, p_in.ev_start, p_in.ev_stop, p, total, slab);
}
}  // namespace

int y355_prepare_conv_px(void) {
    int e = (int)hipFuncSetAttribute((const void *)convpx32_kernel<PX_TH, PX_NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!e) e = (int)hipFuncSetAttribute((const void *)convpx32_kernel<PX_TH, PX_NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

// conv3_1 only; false = not available for this launch (statistics mode, head-room guard, 64-bit epilogue, t beyond fp32's
// exact range, a map too wide for two slabs in LDS): the caller falls back to conv3x3_v2.hip / conv3x3.hip.
// `p.w` must be the y355_pack_px32 layout.
bool y355_launch_conv_px(int kid, const ConvParams &p, hipStream_t s) {
    if (kid != Y355_K_CONV3_1 || (p.mode & 0xff) != 0 || p.rq.wide || p.guard || p.rq.tmax_log2 > 24) return false;
    if (p.cstride != COUT || !p.out_halo || 2 * px_slab_bytes(p.W) > 160 * 1024 || p.W < 8) return false;
    const bool fold = p.rq.shl == 0 && p.rq.tmax_log2 <= 22 && p.rq.sh <= 22 && p.rq.sh - p.rq.lk >= -8;
    if (fold) px_launch<true>(p, s);
    else px_launch<false>(p, s);
    return true;
}
