// yolo355 -- conv3_1, conv3_2 (+pool), conv4_1, conv4_2 (+pool) of the q_bf path (models/slim_yolo_v2.py:246-288; the
// FPGA's conv_normal calls 3-6, c_embedding/yolo_forward.c:1214-1236): production kernels, round 3.
//
// These layers have K = 288 .. 1152: too shallow for the deep layers' ring kernel, whose per-tile start-up, per-k-step
// barrier and staged two-pass epilogue were most of their time (22-31 us per launch for 5-10 us of MFMA work).
// Here the GEMM is turned round, as in the fused front end (front.hip):
//   * the WEIGHTS are the MFMA's A operand (rows = output channels).  A wave owns a block of 16 NTN output channels and keeps
//     ALL of that block's weight fragments in registers for the whole launch (72-144 VGPRs): no weight traffic, no barrier
//     inside a chunk of work;
//   * the PIXELS are the B operand (columns = 16 consecutive pixels, or 16 consecutive 2x2 pooling windows), read from an
//     LDS slab with ds_read_b128.  Pooled layers read the 4x4 input neighbourhood of a window once and feed it to the four
//     conv outputs of the window (four accumulator sets, same B registers); the pool is an element-wise max;
//   * output channel (4 NTN) g + 4 n + r sits in register r of n-tile n of lane group g (weight rows are permuted so on the
//     host), so a lane ends up with 4 NTN CONSECUTIVE channels of ONE pixel: one packed 8 / 16-byte global store per lane,
//     contiguous across the wave -- no LDS staging, no second pass;
//   * work is dealt in groups of 16 pixels / windows over the row-major image; workgroup i owns a contiguous, equal share
//     of all groups of the batch and walks it in chunks of a few rounds of its waves;
//   * the input lives in LDS as a ROLLING RING of whole padded rows (slot = absolute padded row & (R - 1)): every row is
//     copied once per workgroup by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces that each stay inside one row: the LDS
//     pitch is padded to a piece multiple), the rows the NEXT chunk adds are in flight while this chunk computes, and the
//     chunk boundary waits with a COUNTED vmcnt that leaves the chunk's own output stores in flight.  64- and 128-byte
//     pixels are XOR-swizzled on the source side so that every ds_read_b128 is conflict-free under gfx950's 4 x 16 lane
//     grouping (chunk ^ 2 ((x >> 2) & 1) resp. chunk ^ 2 ((x >> 1) & 3));
//   * the epilogue is front.hip's fp32 form (two fma + max on exact integers).  FOLD 2 (accumulator shift 0, |t| < 2^22): bias +
//     0x4B400000 rides in as the MFMAs' C operand and the accumulator IS the float 1.5 * 2^23 + t; FOLD 1 (shift 0, |t| < 2^24):
//     the bias rides in, one v_cvt; FOLD 0: v_cvt + fma; the hot pass stores unclamped and tracks max / min, a cold pass re-does a wave's groups clamped and
//     counts when a value left [-127, 127].
// Integer semantics: DESIGN.md section 2, bit for bit those of conv3x3_ring.hip / conv3x3.hip.
#include "y355_common.h"
#include <cstring>
#include <type_traits>
#ifndef PX_R31
#define PX_R31 11                // groups per wave and chunk, conv3_1 / conv4_1: two chunks per workgroup at B = 64, 416 x 416
#define PX_R41 6
#endif
#ifndef PX_DIAG
#define PX_DIAG 0                // 1: s_memrealtime stamps (100 MHz) per workgroup at the chunk boundaries (y355_debug_stamps)
#endif

namespace {
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
// byte B of w = bits [7:0] of max(a, b), the other bytes kept (B = 0: zeroed): the LeakyReLU's max and the int8 pack in one
// SDWA instruction per output (front.hip)
template <int B>
__device__ __forceinline__ void pmax_to_byte(unsigned int &w, float a, float b) {
    if constexpr (B == 0)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 1)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 2)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned int ppack4(float a, float b, float c, float d) {
    const unsigned int ab = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x0c0c0400u);
    const unsigned int cd = __builtin_amdgcn_perm(__float_as_uint(d), __float_as_uint(c), 0x04000c0cu);
    return ab | cd;
}

// geometry shared by the kernel, the launcher and the weight packing
template <int CIN, int NTN, int NCB, bool POOL>
struct PxGeom {
    static constexpr int PXB = CIN;                              // bytes per input pixel
    static constexpr int PPP = 1024 / PXB;                       // pixels per 1 KiB DMA piece (= LDS pitch granule)
    static constexpr int KPP = CIN >= 64 ? CIN / 64 : 1;         // k-steps per tap
    static constexpr int KS = CIN >= 64 ? 9 * KPP : 5;           // CIN = 32: two taps per k-step
    static constexpr int CPB = 16 * NTN;                         // output channels per block (per wave)
    static constexpr int COUT = CPB * NCB;
    static constexpr int NFRAG = NCB * KS * NTN;
    static_assert(CIN == 32 || CIN == 64 || CIN == 128, "input channels");
    static_assert(!(POOL && CIN < 64), "pooled layers: one tap per k-step");
};

struct PxArgs {
    int total_groups;     // groups of 16 pixels / windows in the batch (ngi per image)
    int ngi;              // groups per image
    int cg;               // groups per chunk (a multiple of the pixel streams of a workgroup)
    int pwl;              // LDS row pitch in pixels (a multiple of PPP, >= W + 2)
    int logr;             // ring of 2^logr rows
    int ppg;              // DMA pieces a wave issues behind each of its groups
};
// image, groups [g0, g1) of it, absolute padded input rows [lo, hi) it reads (row = b * (H + 2) + padded row of the image)
struct PxChunk { int b, g0, g1, lo, hi; };

// s_waitcnt needs an immediate; n is wave-uniform
__device__ __forceinline__ void pwait_vmcnt(int n) {
#define PW_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        PW_CASE(0) PW_CASE(1) PW_CASE(2) PW_CASE(3) PW_CASE(4) PW_CASE(5) PW_CASE(6) PW_CASE(7) PW_CASE(8) PW_CASE(9)
        PW_CASE(10) PW_CASE(11) PW_CASE(12) PW_CASE(13) PW_CASE(14) PW_CASE(15) PW_CASE(16)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef PW_CASE
}
}  // namespace

template <int CIN, int NTN, int NCB, bool POOL, int NW, int FOLD>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void convpx_kernel(const ConvParams p, const PxArgs a) {
    using G = PxGeom<CIN, NTN, NCB, POOL>;
    constexpr int PXB = G::PXB, PPP = G::PPP, KPP = G::KPP, KS = G::KS, CPB = G::CPB, COUT = G::COUT;
    constexpr int NPS = NW / NCB;                        // pixel streams: waves that share a channel block
    constexpr int NV = POOL ? 4 : 1;                     // conv outputs per column (pooling window)
    static_assert(NW % NCB == 0, "waves per channel block");
    static_assert(NTN == 4 || NTN == 2, "16- or 8-byte stores");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // ring of 2^logr rows of PWL * PXB bytes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % NCB, ps = wave / NCB;
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;
    const int PW = W + 2, PWL = a.pwl;
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const int npw = Ho * Wo;                             // pixels / windows per image
    const int rowb = PWL * PXB;                          // bytes per slab row

    // ---- weights of this wave's channel block: A fragments [k-step][n-tile], registers for the whole launch
    v4i wf[KS][NTN];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int n = 0; n < NTN; ++n) wf[ks][n] = *(const v4i *)(p.w + ((size_t)(cb * KS + ks) * NTN + n) * 1024 + lane * 16);
    // accumulator register r of n-tile n of lane group g = channel cb * CPB + 4 NTN g + 4 n + r
    v4i cin[NTN];
    float bf[NTN][4];
#pragma unroll
    for (int n = 0; n < NTN; ++n) {
        const v4i bv = *(const v4i *)(p.bias_t + cb * CPB + 4 * NTN * g + 4 * n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cin[n][r] = FOLD == 2 ? bv[r] + 0x4B400000 : FOLD == 1 ? bv[r] : 0;
            bf[n][r] = (float)bv[r];
        }
    }
    const Requant rq = p.rq;
    const float s_pos = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ldexpf(1.0f, rq.lk - rq.sh))));
    const float s_neg = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)rq.neg_mul * ldexpf(1.0f, -rq.sh))));
    const float c_pos = FOLD == 2 ? MAGIC - MAGIC * s_pos : MAGIC, c_neg = FOLD == 2 ? MAGIC - MAGIC * s_neg : MAGIC;
    // VGPR operands for the epilogue's fma: an SGPR source takes a vector instruction off the fast issue path
    // (scratch/ubench/valu_rates.hip: 3.0 cycles per SIMD against 4.6)
    float spv = s_pos, snv = s_neg, cpv = c_pos, cnv = c_neg;
    asm volatile("" : "+v"(spv), "+v"(snv), "+v"(cpv), "+v"(cnv));
    const float scl = ldexpf(1.0f, rq.shl);
    const float invWo = 1.0f / (float)Wo;

    // ---- this workgroup's share of the batch's groups, walked in chunks of <= cg groups that stay inside one image
    const int G_ = gridDim.x;
    const int gbeg = (int)((long long)a.total_groups * blockIdx.x / G_), gend = (int)((long long)a.total_groups * (blockIdx.x + 1) / G_);
    if (gbeg >= gend) return;
    constexpr int MUL = POOL ? 2 : 1;
    const int R = 1 << a.logr, RM = R - 1;
    auto chunk_at = [&](int gg) {
        PxChunk c;
        c.b = gg / a.ngi;
        c.g0 = gg - c.b * a.ngi;
        c.g1 = min(min(c.g0 + a.cg, a.ngi), c.g0 + (gend - gg));
        const int ya = (16 * c.g0) / Wo, yb = (min(16 * c.g1, npw) - 1) / Wo;
        c.lo = c.b * (H + 2) + MUL * ya;
        c.hi = c.b * (H + 2) + MUL * yb + (POOL ? 4 : 3);
        return c;
    };
    // absolute padded rows [r0, r1) -> ring slots row & RM; piece q = 1 KiB = PPP pixels of ONE row, by wave q % NW.
    // Lane l writes LDS chunk l % CPX of pixel l / CPX of the piece and reads the source chunk the swizzle puts there.
    constexpr int CPX = PXB / 16;                        // 16-byte chunks per pixel
    const int dpx = lane / CPX, dch = lane % CPX;
    const int ppr = PWL / PPP;                           // pieces per row
    // Who sends which piece: a wave owns ONE piece column pc0 (its lanes' pixel, swizzled chunk and byte offset inside a row are
    // launch constants) and every RS-th row of it -- ppr <= NW: NW / ppr rows per round of the waves (the waves beyond RS * ppr
    // send nothing); wider rows: every row, the columns pc0, pc0 + NW, ...  A piece then costs one scalar multiply-add for the
    // row's offsets and the DMA (before: a (row, column) cursor with a wrap loop, ~17 scalar and 6 vector instructions per
    // piece, 900 - 1 600 scalar instructions per wave and launch: profiles/r04_notes.md 13).
    const int RS = ppr < NW ? NW / ppr : 1;
    const int CPW = ppr < NW ? 1 : (ppr + NW - 1) / NW;
    const int pc0 = ppr < NW ? wave % ppr : wave;
    const int rr0 = ppr < NW ? (wave / ppr < RS ? wave / ppr : (1 << 28)) : 0;
    auto lane_off = [&](int pc) {                        // byte offset of this lane's 16 bytes inside a padded input row
        const int col = pc * PPP + dpx;
        int sch = dch;
        if constexpr (CPX == 4) sch ^= ((col >> 2) & 1) << 1;
        if constexpr (CPX == 8) sch ^= ((col >> 1) & 3) << 1;
        return min(col, PW - 1) * PXB + 16 * sch;
    };
    const int goff0 = lane_off(pc0);
    // rows r0 + rr, rr = cursor, cursor + RS, ... < nrows: at most about `count` pieces; the cursor travels with the caller
    auto issue_pieces = [&](int r0, int nrows, int &rr, int count) {
        int done = 0;
        for (; rr < nrows && done < count; rr += RS) {
            const int row = r0 + rr;
            const int8_t *src = p.in + (size_t)row * (size_t)(PW * PXB);
            char *dst = smem + (row & RM) * rowb;
            pglds16(src + goff0, dst + pc0 * 1024);
            ++done;
            for (int j = 1; j < CPW; ++j) {              // rows wider than NW pieces (not the shapes of this network)
                const int pc = pc0 + j * NW;
                if (pc < ppr) {
                    pglds16(src + lane_off(pc), dst + pc * 1024);
                    ++done;
                }
            }
        }
        return done;
    };
    auto issue_rows = [&](int r0, int r1) {
        int rr = rr0;
        issue_pieces(r0, r1 - r0, rr, 1 << 30);
    };

    int nstamp = 0;
    auto stamp = [&]() {
#if PX_DIAG
#if PX_DIAG == 2      // every wave, 16 stamps each (256 workgroups x 8 waves)
        if (p.stamps && lane == 0 && nstamp < 16 && blockIdx.x < 128) p.stamps[((size_t)blockIdx.x * NW + wave) * 16 + nstamp++] = __builtin_amdgcn_s_memrealtime();
#else
        if (p.stamps && tid == 0 && nstamp < 32) p.stamps[(size_t)blockIdx.x * 32 + nstamp++] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
    };
    (void)nstamp;
    stamp();
    PxChunk ch = chunk_at(gbeg);
    issue_rows(ch.lo, ch.hi);
    int loaded = ch.hi;                                  // rows below `loaded` (and not yet overwritten) are in the ring or in flight
    unsigned int nsat = 0;
    int gg = gbeg, nstores = -1;                         // stores this wave issued in the previous chunk (-1: wait for everything)
    for (;;) {
        // Everything this chunk reads has landed: the rows were issued BEFORE the previous chunk's output stores, so a counted
        // wait leaves those stores in flight.  Behind the barrier every wave is done with the previous chunk's rows.
        stamp();
        pwait_vmcnt(nstores);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        __builtin_amdgcn_s_barrier();
        if (loaded < ch.hi) {                            // rows that could not be issued ahead (the ring was full: image boundaries)
            issue_rows(max(loaded, ch.lo), ch.hi);
            loaded = ch.hi;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        stamp();
        const int gnext = gg + (ch.g1 - ch.g0);
        const bool more = gnext < gend;
        PxChunk nx = ch;
        if (more) nx = chunk_at(gnext);
        // the next chunk's new rows, as far as they fit beside the rows this chunk still reads, go out a few pieces at a time
        // behind each group's MFMAs (the SIMD's other wave computes while this one issues); the chunk boundary may leave in
        // flight only the stores issued after the LAST piece
        int pf_r0 = 0, pf_nr = 0;
        if (more) {
            const int top = min(nx.hi, ch.lo + R);
            pf_r0 = max(loaded, nx.lo);
            if (top > pf_r0) {
                pf_nr = top - pf_r0;
                loaded = top;
            }
        }
        int pfc = rr0;
        auto prefetch = [&](int count) {
            if (issue_pieces(pf_r0, pf_nr, pfc, count) > 0) nstores = 0;
        };
        stamp();
        int8_t *outb = p.out + (((size_t)ch.b * (Ho + 2) + 1) * (Wo + 2) + 1) * COUT + cb * CPB;    // wave-uniform
        const int rbase = ch.b * (H + 2);

        float ymx = MAGIC, ymn = MAGIC;
        // ---- one group = issue (addresses, B reads, MFMAs) + finish (pool, requantise, pack, store).  (A software pipeline over
        // two accumulator sets with the finish interleaved behind the next group's MFMAs, pinned read-then-MFMA orders and a
        // raised priority for the SIMD's second wave were each measured: no change, profiles/r03_notes.md.)
        auto locate = [&](int grp, int &oy, int &ox) {
            const int pc = min(grp * 16 + li, npw - 1);            // padding lanes of an image's last group repeat its last pixel
            oy = (int)(((float)pc + 0.5f) * invWo);                // pc / Wo (exact: pc < 2^16)
            ox = pc - oy * Wo;
        };
        auto issue = [&](int grp, v4i (&acc)[NV][NTN]) {
            int oy, ox;
            locate(grp, oy, ox);
            const int ar = rbase + MUL * oy, x0 = MUL * ox;        // absolute padded row / padded column of the neighbourhood's corner
            // byte offset inside a row of neighbourhood column c: pixel x0 + c, chunk g (CIN = 32: the half is added per
            // k-step), swizzled by the pixel's x; byte offset of neighbourhood row r in the ring
            constexpr int NC = POOL ? 4 : 3;
            int xoff[NC], roff[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int x = x0 + c;
                int sch = CIN == 32 ? 0 : g;
                if constexpr (CPX == 4) sch ^= ((x >> 2) & 1) << 1;
                if constexpr (CPX == 8) sch ^= ((x >> 1) & 3) << 1;
                xoff[c] = x * PXB + 16 * sch;
                roff[c] = ((ar + c) & RM) * rowb;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int n = 0; n < NTN; ++n) acc[v][n] = cin[n];
            if constexpr (!POOL) {
                v4i bq[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if constexpr (CIN == 32) {                     // k-step = taps 2 ks, 2 ks + 1 (lane groups 0-1 / 2-3), 32 channels each
                        const int t0 = 2 * ks, t1 = 2 * ks + 1 < 9 ? 2 * ks + 1 : 8;   // tap 9 multiplies zero weights
                        const int o0 = roff[t0 / 3] + xoff[t0 % 3], o1 = roff[t1 / 3] + xoff[t1 % 3];
                        bq[ks] = *(const v4i *)(smem + (g < 2 ? o0 : o1) + 16 * (g & 1));
                    } else {
                        const int tap = ks / KPP, h = ks % KPP;
                        bq[ks] = *(const v4i *)(smem + roff[tap / 3] + (xoff[tap % 3] ^ (h << 6)));
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int n = 0; n < NTN; ++n) acc[0][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[ks][n], bq[ks], acc[0][n], 0, 0, 0);
                // pinned: every read of the group, then its MFMAs (one exposed LDS wait per group instead of three)
                __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, KS * NTN, 0);
            } else {
                // neighbourhood row r feeds conv output (dy, dx) with filter tap (r - dy, c - dx).  The order is pinned
                // (sched_group_barrier): rows 0 and 1 are read, then row r's MFMAs run over the reads of row r + 2 -- left to
                // itself the scheduler puts every read right in front of the MFMAs that use it, nine exposed LDS waits per group
                v4i bq[4][4][KPP];
                auto rd = [&](int r) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int h = 0; h < KPP; ++h) bq[r][c][h] = *(const v4i *)(smem + roff[r] + (xoff[c] ^ (h << 6)));
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
                                for (int h = 0; h < KPP; ++h)
#pragma unroll
                                    for (int n = 0; n < NTN; ++n)
                                        acc[2 * dy + dx][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[(ty * 3 + tx) * KPP + h][n], bq[r][c][h],
                                                                                                   acc[2 * dy + dx][n], 0, 0, 0);
                            }
                    }
                };
                if constexpr (KPP == 1) {
                    rd(0);
                    rd(1);
                    mm(0);
                    rd(2);
                    mm(1);
                    rd(3);
                    mm(2);
                    mm(3);
                    constexpr int RDS = 4 * KPP, M03 = 6 * KPP * NTN, M12 = 12 * KPP * NTN;     // reads per row; MFMAs of rows 0 / 3 and 1 / 2
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * RDS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, M03, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, RDS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, M12, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, RDS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, M12 + M03, 0);
                } else {                                           // two k-steps per tap: the weights leave no room for two rows in flight
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rd(r);
                        mm(r);                                     // (a row's reads pinned in front of its MFMAs: measured, no change)
                    }
                }
            }
        };
        auto finish = [&](int grp, const v4i (&acc)[NV][NTN], auto coldc) {
            constexpr bool COLD = decltype(coldc)::value;
            int oy, ox;
            locate(grp, oy, ox);
            unsigned int word[NTN];
#pragma unroll
            for (int n = 0; n < NTN; ++n) {
                // the two branches of the LeakyReLU, each M + rne(t * scale); y = max(pos, neg).  0 <= s_neg <= s_pos (launcher), so
                // y > M + 127 <=> pos > M + 127 and y < M - 127 <=> neg < M - 127: the hot pass tracks the branches and packs the
                // unclamped low bytes with one SDWA max per output
                float pos[4], neg[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = acc[0][n][r];
                    if constexpr (POOL) m = max(max(m, acc[1][n][r]), max(acc[2][n][r], acc[3][n][r]));
                    const float tf = FOLD == 2 ? __int_as_float(m) : FOLD == 1 ? (float)m : fmaf((float)m, scl, bf[n][r]);
                    pos[r] = fmaf(tf, spv, cpv);
                    neg[r] = fmaf(tf, snv, cnv);
                }
                if constexpr (!COLD) {
                    ymx = pvmax3(pvmax3(ymx, pos[0], pos[1]), pos[2], pos[3]);
                    ymn = pvmin3(pvmin3(ymn, neg[0], neg[1]), neg[2], neg[3]);
                    pmax_to_byte<0>(word[n], pos[0], neg[0]);
                    pmax_to_byte<1>(word[n], pos[1], neg[1]);
                    pmax_to_byte<2>(word[n], pos[2], neg[2]);
                    pmax_to_byte<3>(word[n], pos[3], neg[3]);
                } else {
                    float yc[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y = pvmax(pos[r], neg[r]);
                        yc[r] = __builtin_amdgcn_fmed3f(y, QLO, QHI);
                        nsat += (grp * 16 + li < npw && y != yc[r]) ? 1u : 0u;
                    }
                    word[n] = ppack4(yc[0], yc[1], yc[2], yc[3]);
                }
            }
            // unconditional: the padding lanes rewrite the image's last pixel with the same bytes, and the number of stores a
            // wave has in flight stays a function of its group count (the counted wait above)
            int8_t *dst = outb + ((oy * (Wo + 2) + ox) * COUT + 4 * NTN * g);
            if constexpr (NTN == 4) *(v4i *)dst = (v4i){(int)word[0], (int)word[1], (int)word[2], (int)word[3]};
            else *(uint2 *)dst = make_uint2(word[0], word[NTN - 1]);
        };
        nstores = 0;                                               // stores issued after the last prefetched piece
        {
            v4i acc[NV][NTN];
#pragma unroll 1
            for (int grp = ch.g0 + ps; grp < ch.g1; grp += NPS) {
                issue(grp, acc);
                prefetch(a.ppg);
                finish(grp, acc, std::false_type{});
                ++nstores;
            }
            prefetch(1 << 30);
        }
        if (__builtin_amdgcn_ballot_w64(ymx > QHI || ymn < QLO) != 0ull) {   // cold: the rows are still in the ring
            v4i accC[NV][NTN];
#pragma unroll 1
            for (int grp = ch.g0 + ps; grp < ch.g1; grp += NPS) {
                issue(grp, accC);
                finish(grp, accC, std::true_type{});
            }
            nstores = -1;
        }
        stamp();
        if (!more) break;
        gg = gnext;
        ch = nx;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
}

// ------------------------------------------------------------------------------------------
namespace {
// A fragments: fragment ((cb * KS + ks) * NTN + n), lane (i = l & 15, g = l >> 4), 16 bytes:
//   row i = output channel cb * CPB + 4 NTN (i >> 2) + 4 n + (i & 3)
//   CIN = 32:  k = tap 2 ks + (g >> 1), input channels 16 (g & 1) .. + 15
//   CIN >= 64: k = tap ks / KPP, input channels 64 (ks % KPP) + 16 g .. + 15
template <int CIN, int NTN, int NCB, bool POOL>
void px_pack(const int8_t *q_w, int cout, int8_t *dst) {
    using G = PxGeom<CIN, NTN, NCB, POOL>;
    memset(dst, 0, (size_t)G::NFRAG * 1024);
    for (int cb = 0; cb < NCB; ++cb)
        for (int ks = 0; ks < G::KS; ++ks)
            for (int n = 0; n < NTN; ++n)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4;
                    const int ch = cb * G::CPB + 4 * NTN * (i >> 2) + 4 * n + (i & 3);
                    const int tap = CIN == 32 ? 2 * ks + (g >> 1) : ks / G::KPP;
                    const int c0 = CIN == 32 ? 16 * (g & 1) : 64 * (ks % G::KPP) + 16 * g;
                    if (tap > 8 || ch >= cout) continue;
                    for (int kk = 0; kk < 16; ++kk)
                        dst[(((size_t)cb * G::KS + ks) * NTN + n) * 1024 + l * 16 + kk] = q_w[((size_t)ch * CIN + c0 + kk) * 9 + tap];
                }
}

template <int CIN, int NTN, int NCB, bool POOL, int NW>
struct PxInst {
    using G = PxGeom<CIN, NTN, NCB, POOL>;
    static constexpr int NPS = NW / NCB;
    // rounds: groups per pixel stream per chunk
    static PxArgs args(const ConvParams &p, int rounds) {
        PxArgs a;
        const int Ho = POOL ? p.H / 2 : p.H, Wo = POOL ? p.W / 2 : p.W;
        a.ngi = (Ho * Wo + 15) / 16;
        a.total_groups = a.ngi * p.B;
        a.cg = rounds * NPS;
        a.pwl = (p.W + 2 + G::PPP - 1) / G::PPP * G::PPP;
        // two consecutive chunks of an image are in the ring together: MUL * (output rows they touch) + 2 (+ 1 pooled) rows
        const int rows2 = (2 * 16 * a.cg + Wo - 1) / Wo + 1;
        const int need = (POOL ? 2 : 1) * rows2 + (POOL ? 2 : 2);
        a.logr = 2;
        while ((1 << a.logr) < need) ++a.logr;
        // a chunk adds about MUL * 16 cg / Wo rows = that many * pwl / PPP pieces, dealt over NW waves and `rounds` groups each
        const int newrows = (POOL ? 2 : 1) * ((16 * a.cg + Wo - 1) / Wo + 1);
        const int ppr = a.pwl / G::PPP, rs = ppr < NW ? NW / ppr : 1, cpw = ppr < NW ? 1 : (ppr + NW - 1) / NW;
        const int per_wave = (newrows + rs - 1) / rs * cpw;         // a wave sends one piece column of every rs-th row
        a.ppg = (per_wave + rounds - 1) / rounds;
        return a;
    }
    static size_t lds_bytes(const PxArgs &a) { return ((size_t)a.pwl * G::PXB) << a.logr; }
    template <int FOLD>
    static void launch_(const ConvParams &p_in, const PxArgs &a, hipStream_t s) {
        ConvParams p = p_in;
        p.ev_start = p.ev_stop = nullptr;
        int grid = y355_cu_count();                                           // one NW-wave workgroup per CU
        // throughput mode (Y355_OPT_RING_WORKGROUPS: several handles share the GPU): fewer workgroups, each walking a longer share of
        // the groups -- the weights-into-registers prologue (up to 288 KB per workgroup) is paid half as often and the launch leaves
        // CUs to the other handles' kernels: 64 / 96 / 128 / 160 / 256 workgroups -> 272.7 / 286.2 / 293.5 / 289.2 / 285.8 k img/s
        if (p_in.grid_limit > 0 && p_in.grid_limit < grid) grid = p_in.grid_limit;
        if (grid > a.total_groups) grid = a.total_groups;
        Y355_LAUNCH((convpx_kernel<CIN, NTN, NCB, POOL, NW, FOLD>), dim3(grid), dim3(NW * 64), lds_bytes(a), s, p_in.ev_start, p_in.ev_stop, p, a);
    }
    static bool launch(const ConvParams &p, int rounds, hipStream_t s) {
        if (p.cstride != G::COUT || !p.out_halo || p.W < 16 || (POOL && ((p.H | p.W) & 1))) return false;
        if ((long long)p.B * (p.H + 2) * (p.W + 2) * G::PXB >= (1ll << 31)) return false;     // 32-bit row arithmetic
        const PxArgs a = args(p, rounds);
        if (lds_bytes(a) > 160 * 1024) return false;
        if (p.rq.shl == 0 && p.rq.tmax_log2 <= 22 && p.rq.sh <= 22 && p.rq.sh - p.rq.lk >= -8) launch_<2>(p, a, s);
        else if (p.rq.shl == 0) launch_<1>(p, a, s);
        else launch_<0>(p, a, s);
        return true;
    }
    static int prepare() {
        int e = (int)hipFuncSetAttribute((const void *)convpx_kernel<CIN, NTN, NCB, POOL, NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (!e) e = (int)hipFuncSetAttribute((const void *)convpx_kernel<CIN, NTN, NCB, POOL, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (!e) e = (int)hipFuncSetAttribute((const void *)convpx_kernel<CIN, NTN, NCB, POOL, NW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }
};
//                   CIN NTN NCB POOL  NW
using PX_C3_1 = PxInst<32, 4, 1, false, 8>;       // 32 -> 64
using PX_C3_2 = PxInst<64, 2, 2, true, 8>;        // 64 -> 64, pooled: two 32-channel blocks x four window streams
using PX_C4_1 = PxInst<64, 4, 2, false, 8>;       // 64 -> 128
using PX_C4_2 = PxInst<128, 2, 4, true, 8>;       // 128 -> 128, pooled: four 32-channel blocks x two window streams
}  // namespace

int y355_prepare_conv_px(void) {
    int e = PX_C3_1::prepare();
    if (!e) e = PX_C3_2::prepare();
    if (!e) e = PX_C4_1::prepare();
    if (!e) e = PX_C4_2::prepare();
    return e;
}

size_t y355_px_packed_bytes(int kid) {
    switch (kid) {
    case Y355_K_CONV3_1: return (size_t)PX_C3_1::G::NFRAG * 1024;
    case Y355_K_CONV3_2: return (size_t)PX_C3_2::G::NFRAG * 1024;
    case Y355_K_CONV4_1: return (size_t)PX_C4_1::G::NFRAG * 1024;
    case Y355_K_CONV4_2: return (size_t)PX_C4_2::G::NFRAG * 1024;
    default: return 0;
    }
}

// q_w [cout][cin][3][3] of layer `kid` -> the fragment order convpx_kernel streams; false: the layer has no such kernel
bool y355_pack_px(int kid, const int8_t *q_w, int cout, int cin, int8_t *dst) {
    switch (kid) {
    case Y355_K_CONV3_1: if (cin != 32 || cout != 64) return false; px_pack<32, 4, 1, false>(q_w, cout, dst); return true;
    case Y355_K_CONV3_2: if (cin != 64 || cout != 64) return false; px_pack<64, 2, 2, true>(q_w, cout, dst); return true;
    case Y355_K_CONV4_1: if (cin != 64 || cout != 128) return false; px_pack<64, 4, 2, false>(q_w, cout, dst); return true;
    case Y355_K_CONV4_2: if (cin != 128 || cout != 128) return false; px_pack<128, 2, 4, true>(q_w, cout, dst); return true;
    default: return false;
    }
}

// false = not available for this launch (statistics mode, head-room guard, 64-bit epilogue, t beyond fp32's exact range, a
// map too wide for two slabs in LDS): the caller falls back to the ring / v2 / generic kernels.  `p.w` = y355_pack_px layout.
bool y355_launch_conv_px(int kid, const ConvParams &p, hipStream_t s) {
    if ((p.mode & 0xff) != 0 || p.rq.wide || p.guard || p.rq.tmax_log2 > 24) return false;
    if (p.rq.neg_mul < 0 || p.rq.neg_mul > (1 << p.rq.lk)) return false;      // the epilogue assumes a LeakyReLU slope in [0, 1]
    switch (kid) {
    case Y355_K_CONV3_1: return PX_C3_1::launch(p, PX_R31, s);
    case Y355_K_CONV3_2: return PX_C3_2::launch(p, 2, s);
    case Y355_K_CONV4_1: return PX_C4_1::launch(p, PX_R41, s);
    case Y355_K_CONV4_2: return PX_C4_2::launch(p, 2, s);
    default: return false;
    }
}
