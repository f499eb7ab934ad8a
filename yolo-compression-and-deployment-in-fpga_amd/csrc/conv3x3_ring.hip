// yolo355 -- fused int8 3x3 convolution, production kernel for layers with >= 64 input channels
// (conv3_2 .. conv7, pred): persistent workgroups, LDS-DMA rings with a DEEP weight prefetch.
//
// Measured on the round-1 two-slot ring kernel (s_memtime stamps, profiles/README.md): a k-step's MFMAs take ~700-900
// cycles, but every step also waited 300-2700 cycles for its LDS-DMA data -- an LDS-DMA issued behind
// other traffic lands 1-1.5 us later, and the v2 ring kept only two k-steps in flight.  This kernel
// keeps PF (5) k-steps of weights in flight:
//   * weight ring of PF + 2 slots; step s issues W(s+1+PF) into the slot read in step s-2 (LDS-DMA
//     WAR rule: restage two barriers after the last ds_read), waits (counted vmcnt) for W(s+1), and
//     reads W(s+1)'s B fragments into registers under its own MFMAs -- nothing waits for LDS behind
//     a barrier;
//   * the next 64-channel activation slab goes out one 1 KiB piece per wave per step (t = 1..PPW)
//     instead of as one blob, so no weight piece queues behind 30-57 KiB of slab in the in-order
//     vmcnt stream;
//   * the LDS that buys this comes from the epilogue: the int8 output tile is staged in the slab
//     slot that just died (in two passes where it is larger than a slot), not in its own buffer.
// Same math, tile geometry, weight packing and epilogue arithmetic as conv3x3.hip.
#include "y355_common.h"
#ifndef Y355_RING_XCD_SHARE
#define Y355_RING_XCD_SHARE 1              // 0: plain work-item order (A/B builds)
#endif
#include <type_traits>
#include <hip/hip_ext.h>
#include <cstdlib>
#ifndef Y355_DIAG
#define Y355_DIAG 0                 // 1 / 2: s_memtime stamps per workgroup / per wave, 3: six phase stamps per wave + where it ran (y355_debug_stamps, scratch/stamps_ring_pairs.py); never in the production build
#endif
#ifndef Y355_DIAG12
#define Y355_DIAG12 (Y355_DIAG == 1 || Y355_DIAG == 2)
#endif
#ifndef Y355_RING_PF
#define Y355_RING_PF 4              // k-steps of weights in flight (ring of PF + 2 = 6 slots: divides the 18 / 36 k-steps of the deep layers' tiles, which makes every slot a compile-time constant -- STATIC below); 4..7 measured equal before that (profiles/r02_notes.md)
#endif
// The timing ablations, the hand-placed (volatile asm) k-step, the row-dependent chunk swizzle, the refill-position and
// half-tile variants of round 2 live in scratch/ring_experiments/conv3x3_ring_r2_experiments.hip; none of them paid
// (profiles/r02_notes.md) and the production kernel keeps ONE body.

__device__ __forceinline__ void rglds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void rwait_vmcnt() {
    static_assert(N >= 0, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

// fp32 epilogue on exact integers (FPE): 1.5 * 2^23 and the clamp bounds around it
constexpr float RMAGIC = 12582912.0f, RQLO = 12582785.0f, RQHI = 12583039.0f;
__device__ __forceinline__ float rvmax(float a, float b) {       // v_max_f32 without the canonicalising multiply
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float rvmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float rvmin3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// byte B of w = bits [7:0] of max(a, b), the other bytes kept (B = 0: zeroed) -- front.hip
template <int B>
__device__ __forceinline__ void rmax_to_byte(unsigned int &w, float a, float b) {
    if constexpr (B == 0)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 1)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else if constexpr (B == 2)
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
    else
        asm("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned int rpack4(float a, float b, float c, float d) {   // low bytes of four floats M + q
    const unsigned int ab = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x0c0c0400u);
    const unsigned int cd = __builtin_amdgcn_perm(__float_as_uint(d), __float_as_uint(c), 0x04000c0cu);
    return ab | cd;
}

// s_waitcnt needs an immediate; callers pass values that are constants after unrolling, so the switch
// folds to one instruction
__device__ __forceinline__ void rwait_vmcnt_dyn(int n) {
#define RW_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 63 ? 63 : n)) {
        RW_CASE(0) RW_CASE(1) RW_CASE(2) RW_CASE(3) RW_CASE(4) RW_CASE(5) RW_CASE(6) RW_CASE(7) RW_CASE(8) RW_CASE(9)
        RW_CASE(10) RW_CASE(11) RW_CASE(12) RW_CASE(13) RW_CASE(14) RW_CASE(15) RW_CASE(16) RW_CASE(17) RW_CASE(18)
        RW_CASE(19) RW_CASE(20) RW_CASE(21) RW_CASE(22) RW_CASE(23) RW_CASE(24) RW_CASE(25) RW_CASE(26) RW_CASE(27)
        RW_CASE(28) RW_CASE(29) RW_CASE(30) RW_CASE(31) RW_CASE(32) RW_CASE(33) RW_CASE(34) RW_CASE(35) RW_CASE(36)
        RW_CASE(37) RW_CASE(38) RW_CASE(39) RW_CASE(40) RW_CASE(41) RW_CASE(42) RW_CASE(43) RW_CASE(44) RW_CASE(45)
        RW_CASE(46) RW_CASE(47) RW_CASE(48) RW_CASE(49) RW_CASE(50) RW_CASE(51) RW_CASE(52) RW_CASE(53) RW_CASE(54)
        RW_CASE(55) RW_CASE(56) RW_CASE(57) RW_CASE(58) RW_CASE(59) RW_CASE(60) RW_CASE(61) RW_CASE(62) RW_CASE(63)
    }
#undef RW_CASE
}

// slab pieces issued in steps lo..hi (step u issues one when 1 <= (u mod 9) <= ppw); negative steps
// are the previous tile's (none before the first tile: its slab went out whole in the prologue)
constexpr int ring_sp(int lo, int hi, int ppw, bool prev) {
    int n = 0;
    for (int u = lo; u <= hi; ++u) {
        if (u < 0 && !prev) continue;
        const int t = ((u % 9) + 9) % 9;
        if (t >= 1 && t <= ppw) ++n;
    }
    return n;
}

// FPE: the requantisation runs in fp32 on exact integers (launcher-proved: accumulator shift 0, no left requant shift, right
// shift <= 17, so every t = acc + bias that does not saturate is below 2^24 and converts exactly; a larger one converts to
// something at least as large and saturates either way).  The biases come from a 1 KiB LDS copy that the prologue's first
// LDS-DMA makes (oldest in the vmcnt stream: every later wait covers it; no global load in the epilogue).
template <int CIN, int BN, int TH, int TW, bool POOL, int WM, int WN, int PF, bool ROLL, bool DIRECT_REQ, bool FPE = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4) ? 2 : 1) void conv3x3_i8_ring_kernel(const ConvParams p, const int total_tiles) {
    constexpr int NW = WM * WN;
    constexpr int NTHR = NW * 64;
    constexpr int NCH = CIN / 64, SPC = 9, KS = NCH * SPC;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int PWL = (PW + 7) / 8 * 8;
    constexpr int NPIX = PH * PWL;
    constexpr int BM = TH * TW;
    constexpr int MT_TOT = (BM + 15) / 16;
    constexpr int MT = (MT_TOT + WM - 1) / WM;
    constexpr int NT = BN / 16 / WN;
    constexpr int SLABB = (NPIX * 64 + 1023) / 1024 * 1024;
    constexpr int NPIECE = SLABB / 1024;
    constexpr int PPW = (NPIECE + NW - 1) / NW;                    // slab pieces per wave
    constexpr int WB = (BN / 16) * 1024;
    constexpr int NFR = BN / 16;
    constexpr int WPW = (NFR + NW - 1) / NW;                       // weight pieces per wave per k-step
    constexpr int WSLOTS = PF + 2;
    constexpr int OFF_W = 2 * SLABB;
    constexpr int OFF_DUMMY = OFF_W + WSLOTS * WB;
    constexpr int OFF_BIAS = OFF_DUMMY + 1024;                     // FPE: the layer's biases (<= 256 int32)
    constexpr int SROWS = POOL ? MT * WM * 4 : MT * WM * 16;
    constexpr int SSTR = BN + 16;
    constexpr int OROWS = POOL ? BM / 4 : BM;
    // The int8 output tile is staged through the slab slot that just died when it fits (pooled layers);
    // otherwise every lane stores its own 4 adjacent channels directly (16 lanes = 64 contiguous bytes
    // of a pixel) -- measured faster than staging in two passes, which serialises the requantisation
    constexpr bool DIRECT = DIRECT_REQ && (SROWS * SSTR > SLABB);
    constexpr int NPASS = (!DIRECT && SROWS * SSTR > SLABB) ? 2 : 1;
    constexpr int RP = SROWS / NPASS;                              // staged rows per pass
    constexpr int CG = BN / 16;
    constexpr int NITP = (RP * CG + NTHR - 1) / NTHR;              // output stores per thread per pass
    constexpr int NIT = DIRECT ? (POOL ? MT : MT * 4) : NPASS * NITP;   // output stores per thread per tile (static)
    static_assert(CIN % 64 == 0 && NT == 4, "64-channel chunks, four n-tiles per wave");
    static_assert(PPW <= 8 && PF >= 2 && PF <= 8, "slab pieces go out at t = 1..PPW");
    static_assert(!(FPE && (DIRECT_REQ || ROLL)), "the fp32 epilogue is written for the staged path and the unrolled chunk loop");
    static_assert(DIRECT || (RP * SSTR <= SLABB && MT % NPASS == 0), "staging fits the dead slot");
    constexpr int UNRC = ROLL ? 1 : NCH;
    // STATIC: the weight ring and the two slab slots are back at slot 0 when a tile ends (the ring's length divides the tile's
    // k-steps, the chunk count is even), so in the unrolled loop every LDS slot is a compile-time constant: the scalar
    // bookkeeping of the ring (compare / select / add: ~12 scalar instructions per step and wave, which take issue slots beside
    // the MFMAs: scratch/ubench/valu_issue.hip) and the vector address of the B fragments disappear -- 46 instead of 62
    // instructions per wave and k-step.  The order of what is left is pinned (sched_group_barrier below): left to itself the
    // scheduler moves each fragment read in front of the MFMAs that use it and the gain is gone (profiles/r04_notes.md 11).
    constexpr bool STATIC = !ROLL && KS % WSLOTS == 0 && NCH % 2 == 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;

    // slab DMA pieces of this wave: piece q = wave + NW*j covers LDS bytes [q*1024, +1024); lane l owns
    // the 16 bytes at pixel q*16 + l/4, XOR-swizzled 16-byte group (l&3) ^ ((l>>3)&3).  Piece j is
    // NW*16/PWL patch rows below piece 0, so one (row, column) pair describes them all.
    static_assert((NW * 16) % PWL == 0 && NPIX % 16 == 0, "slab pieces step by whole patch rows");
    constexpr int PSTEP = NW * 16 / PWL;
    const int pix0 = wave * 16 + (lane >> 2);
    const int ppy0 = pix0 / PWL, ppx0 = min(pix0 % PWL, PW - 1);     // pitch padding re-reads column PW-1
    const int pwithin = ((lane & 3) ^ ((lane >> 3) & 3)) << 4;
    auto decode = [&](int tile, int &b, int &y0, int &x0, int &nb) {
        // Workgroups are dealt round-robin over the 8 XCDs (workgroup w runs on XCD w % 8), each XCD with its own L2.  The plain
        // order puts the work items that read the SAME input -- the n-blocks of a tile (the same slab), the tiles of an image
        // (each other's halo rows) -- on neighbouring XCDs: the slab is fetched into several L2s.  Within every group of
        // 8 x share consecutive items the walk is permuted so that the `share` items of one input go to workgroups w, w + 8, ..
        // -- one XCD, dispatched together: the later readers find the slab in L2 (conv6 / conv7: 32.6 -> 18.5 MB fetched per
        // launch, profiles/r06_notes.md).  share = 2^xcd_share_log2 is the launcher's (0: plain order).
        if (p.xcd_share_log2 > 0) {
            const int gs = 8 << p.xcd_share_log2;
            if (tile < (total_tiles & ~(gs - 1))) {
                const int j = tile & (gs - 1);
                tile = (tile & ~(gs - 1)) | ((j & 7) << p.xcd_share_log2) | (j >> 3);
            }
        }
        nb = tile % p.nblk;
        tile /= p.nblk;
        x0 = (tile % p.tiles_x) * TW;
        tile /= p.tiles_x;
        y0 = (tile % p.tiles_y) * TH;
        b = tile / p.tiles_y;
    };
    auto issue_slab_piece = [&](int b, int y0, int x0, int c, int slot, int j) {
        const int q = wave + NW * j;
        const int8_t *inb = p.in + (size_t)b * (H + 2) * (W + 2) * CIN + c * 64;
        const int gy = min(y0 + ppy0 + j * PSTEP, H + 1), gx = min(x0 + ppx0, W + 1);
        const int8_t *src = inb + ((size_t)gy * (W + 2) + gx) * CIN + pwithin;   // pad pieces read a valid row too
        char *dst = (q < NPIECE) ? smem + slot * SLABB + q * 1024 : smem + OFF_DUMMY;
        rglds16(src, dst);
    };
    auto issue_w = [&](int nb, int ks, int slot) {
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int f = wave + NW * j;
            const bool ok = NFR % NW == 0 || f < NFR;             // whole rounds of pieces: no dummy destination, no select
            const int8_t *src = p.w + ((size_t)(nb * KS + ks) * NFR + (ok ? f : 0)) * 1024 + lane * 16;
            char *dst = ok ? smem + OFF_W + slot * WB + f * 1024 : smem + OFF_DUMMY;
            rglds16(src, dst);
        }
    };
    auto wrap = [](int s) { return s >= WSLOTS ? s - WSLOTS : s; };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (Y355_DIAG12) {
#if Y355_DIAG == 2
            if (p.stamps && lane == 0 && blockIdx.x < 1024 / NW && nstamp < 32)
                p.stamps[(size_t)(blockIdx.x * NW + wave) * 32 + nstamp++] = __builtin_amdgcn_s_memtime();
#else
            if (p.stamps && tid == 0 && nstamp < 32) p.stamps[(size_t)blockIdx.x * 32 + nstamp++] = __builtin_amdgcn_s_memtime();
#endif
        }
    };
    stamp();
    stamp();
    // Y355_DIAG == 3: six phase stamps per wave on the 100 MHz clock (s_memrealtime): 0 entry, 1 prologue issued, 2 first data
    // landed (past the first barrier), 3 k-loop done, 4 epilogue's stores issued, 5 stores retired
    auto pstamp = [&](int i) {
#if Y355_DIAG == 3
        if (p.stamps && lane == 0 && blockIdx.x < Y355_STAMP_ROWS / NW) {
            unsigned long long *row = p.stamps + (size_t)(blockIdx.x * NW + wave) * 32;
            row[i] = __builtin_amdgcn_s_memrealtime();
            // shader-clock twins of stamps 2 / 3 (k-loop start / end): cycles of the k-loop and, with the 100 MHz stamps, the clock it ran at
            if (i == 2 || i == 3) row[10 + i] = __builtin_amdgcn_s_memtime();
            if (i == 0) {                                       // where the wave runs: HW_ID (wave, SIMD, CU, SE), XCC_ID, (workgroup, wave)
                row[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                row[10] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                row[11] = blockIdx.x * 16 + wave;
            }
        }
#endif
    };
    pstamp(0);
    int b, y0, x0, nb;
    decode(tile, b, y0, x0, nb);
    int sl = 0;                                                // slab slot of the current chunk
    int wq = 0;                                                // ring slot of W(s) at step s
    // ---- prologue: (FPE: the biases,) slab 0 whole, then W(0) .. W(PF)
    if constexpr (FPE) rglds16(p.bias_t + min(lane * 4, p.cstride - 4), wave == 0 ? smem + OFF_BIAS : smem + OFF_DUMMY);
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_slab_piece(b, y0, x0, 0, 0, j);
#pragma unroll
    for (int k = 0; k <= PF; ++k) {
        if (k < KS) issue_w(nb, k, k);
        else issue_w(nb, k - KS, k);                           // KS > PF for every layer here; keeps counts static
    }
    pstamp(1);
    const Requant rq = p.rq;
    unsigned int nsat = 0;
    bool first = true;

    for (;;) {
        int ntile = tile + gridDim.x;
        const bool more = ntile < total_tiles;
        if (!more) ntile = tile;                               // keep the operation counts static
        int b2, y2, x2, nb2;
        decode(ntile, b2, y2, x2, nb2);
        // per-lane A-fragment bases (integer divisions): computed while the prologue's (or the previous tile's prefetch) DMAs
        // are in flight, per tile from an opaque copy of the lane id so that they are not live across the epilogue
        int abase[MT][3];
        {
            int li_a = li, g_a = g;
            asm volatile("" : "+v"(li_a), "+v"(g_a));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int row = (wm * MT + m) * 16 + li_a;
                row = min(row, BM - 1);
                int oy, ox;
                if constexpr (POOL) {
                    const int w = row >> 2, r = row & 3;
                    oy = 2 * (w / (TW / 2)) + (r >> 1);
                    ox = 2 * (w % (TW / 2)) + (r & 1);
                } else {
                    oy = row / TW;
                    ox = row % TW;
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    abase[m][dx] = (oy * PWL + ox + dx) * 64 + ((g_a ^ (((ox + dx) >> 1) & 3)) << 4);
            }
        }

        v4i acc[MT][NT];
        if constexpr (!FPE) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = (v4i){0, 0, 0, 0};
        }

        // ---- pre-phase: publish W(0) (and slab 0) and read W(0)'s B fragments.
        // vmcnt is in issue order: "at most N younger operations may still fly".  Younger than W(0):
        // W(1..PF), the slab pieces issued with them, and the previous tile's NIT output stores.
        v4i bfb[2][NT];
        v4i afp[2];
        if (first) rwait_vmcnt<PF * WPW>();
        else rwait_vmcnt<PF * WPW + ring_sp(-PF, -1, PPW, true) + NIT>();
        __builtin_amdgcn_s_barrier();
        pstamp(first ? 2 : 6);
        if constexpr (FPE) {
            // the accumulators start at the bias (t = acc + bias is what the epilogue wants: one v_add per output less there);
            // the LDS copy of the biases is the prologue's oldest DMA: landed and published by the barrier above
            const v4i bv = *(const v4i *)(smem + OFF_BIAS + (nb * BN + wn * (NT * 16) + li * NT) * 4);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = (v4i){bv[t], bv[t], bv[t], bv[t]};
        }
        {
            const char *wb0 = smem + OFF_W + (STATIC ? 0 : wq) * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) bfb[0][tt] = *(const v4i *)(wb0 + tt * 1024);
        }
        // The chunk loop stays rolled (register pressure, code size): everything that depends on the chunk
        // index is wave-uniform run-time data; the counted waits are per-t constants with two variants
        // (the tile's first chunk, which also sees the previous tile's stores, and the others).
#pragma unroll UNRC
        for (int c = 0; c < NCH; ++c) {
            const int slc = STATIC ? (c & 1) : sl;              // slab slot of this chunk
            const int soff = slc * SLABB;                       // wave-uniform: one v_add per A read (STATIC: an immediate)
            const bool lastc = (c + 1 == NCH);
#pragma unroll
            for (int t = 0; t < SPC; ++t) {
                const int s_idx = c * SPC + t;
                // ---- W(s+1) has landed (own pieces) -> barrier -> everybody's has.  Younger than
                // W(s+1) (issued at step s-PF): the W pieces and slab pieces of steps s-PF+1 .. s-1,
                // and the previous tile's stores while s < PF.  At a chunk's first step the slab must
                // be complete too: its last piece went out at t = PPW of the previous chunk, followed
                // by 9-PPW steps of W pieces (and the stores, at a tile's first chunk).
                {
                    constexpr int n_slab = (9 - PPW) * WPW;
                    const bool fine = Y355_DIAG12 && first && c == (NCH > 1 ? 1 : 0);
                    if (fine) stamp();
                    if constexpr (ROLL) {
                        // chunk 0 of the first tile / of a later tile; any later chunk (s >= 9 > PF)
                        const int n0f = (PF - 1) * WPW + ring_sp(t - PF + 1, t - 1, PPW, false);
                        int n0l = (PF - 1) * WPW + ring_sp(t - PF + 1, t - 1, PPW, true) + (t < PF ? NIT : 0);
                        int nr = (PF - 1) * WPW + ring_sp(9 + t - PF + 1, 9 + t - 1, PPW, true);
                        if (t == 0) {
                            if (n_slab + NIT < n0l) n0l = n_slab + NIT;
                            if (n_slab < nr) nr = n_slab;
                        }
                        if (c == 0) {
                            if (first) rwait_vmcnt_dyn(n0f);
                            else rwait_vmcnt_dyn(n0l);
                        } else {
                            rwait_vmcnt_dyn(nr);
                        }
                    } else {
                        const int lo = s_idx - PF + 1, hi = s_idx - 1;
                        int n_first = (PF - 1) * WPW + ring_sp(lo, hi, PPW, false);
                        int n_later = (PF - 1) * WPW + ring_sp(lo, hi, PPW, true) + (s_idx < PF ? NIT : 0);
                        if (t == 0) {
                            if (c > 0 && n_slab < n_first) n_first = n_slab;
                            const int n_slab2 = n_slab + (c == 0 ? NIT : 0);
                            if (n_slab2 < n_later) n_later = n_slab2;
                        }
                        // one wait where the two agree (every step past the tile's first PF): a branch diamond
                        // here would end the scheduling region between the MFMAs and the next step's setup
                        if (n_first == n_later) rwait_vmcnt_dyn(n_later);
                        else if (first) rwait_vmcnt_dyn(n_first);
                        else rwait_vmcnt_dyn(n_later);
                    }
                    if (fine) stamp();
                }
                __builtin_amdgcn_s_barrier();
                if (Y355_DIAG12 && first && c == (NCH > 1 ? 1 : 0)) stamp();
                // ---- refill: one slab piece (t = 1..PPW) into the slot that died two barriers ago,
                // W(s+1+PF) into the ring slot read in step s-2
                const int wqs = STATIC ? s_idx % WSLOTS : wq;
                {
                    if (t >= 1 && t <= PPW) {
                        issue_slab_piece(lastc ? b2 : b, lastc ? y2 : y0, lastc ? x2 : x0, lastc ? 0 : c + 1, slc ^ 1, t - 1);
                    }
                    const int ksn = s_idx + 1 + PF;
                    const bool nxt = ksn >= KS;
                    issue_w(nxt ? nb2 : nb, nxt ? ksn - KS : ksn, wrap(wqs + PF + 1));
                }
                const int ko = (t / 3) * PWL * 64;
                const int acol = t % 3;
                const int cur = ROLL ? (t & 1) : (s_idx & 1);     // rolled: every chunk starts with its B fragments in bfb[0]
                if constexpr (!STATIC) wq = wrap(wq + 1);
                v4i af[MT];
                {
                    if (t == 0) {
                        af[0] = *(const v4i *)(smem + abase[0][acol] + soff + ko);
                        if constexpr (MT > 1) af[1] = *(const v4i *)(smem + abase[1][acol] + soff + ko);
                    } else {
                        af[0] = afp[0];
                        if constexpr (MT > 1) af[1] = afp[1];
                    }
                    if constexpr (STATIC) {
                        // pinned order: this step's remaining A fragments, the MFMAs of the two m-tiles whose fragments are
                        // already here, then the next step's B fragments and first A fragments two reads per m-tile
#pragma unroll
                        for (int m = 2; m < MT; ++m) af[m] = *(const v4i *)(smem + abase[m][acol] + soff + ko);
                        auto mf = [&](int m) {
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt)
                                acc[m][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[m], bfb[cur][tt], acc[m][tt], 0, 0, 0);
                        };
                        mf(0);
                        if constexpr (MT > 1) mf(1);
                        if (s_idx + 1 < KS) {
                            const char *wbn = smem + OFF_W + wrap(wqs + 1) * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) bfb[cur ^ 1][tt] = *(const v4i *)(wbn + tt * 1024);
                        }
                        if (t + 1 < SPC) {
                            const int ko2 = ((t + 1) / 3) * PWL * 64;
                            const int acol2 = (t + 1) % 3;
                            afp[0] = *(const v4i *)(smem + abase[0][acol2] + soff + ko2);
                            if constexpr (MT > 1) afp[1] = *(const v4i *)(smem + abase[1][acol2] + soff + ko2);
                        }
#pragma unroll
                        for (int m = 2; m < MT; ++m) mf(m);
                        if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, MT - 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
#pragma unroll
                        for (int m = 2; m < MT; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                        }
                    } else {
                    if (s_idx + 1 < KS) {                            // B fragments of step s+1, under this step's MFMAs
                        const char *wbn = smem + OFF_W + wrap(wqs + 1) * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) bfb[cur ^ 1][tt] = *(const v4i *)(wbn + tt * 1024);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (m + 2 < MT) af[m + 2] = *(const v4i *)(smem + abase[m + 2][acol] + soff + ko);
                        if (m == MT - 1 && t + 1 < SPC) {            // next step's first A fragments (same slab)
                            const int ko2 = ((t + 1) / 3) * PWL * 64;
                            const int acol2 = (t + 1) % 3;
                            afp[0] = *(const v4i *)(smem + abase[0][acol2] + soff + ko2);
                            if constexpr (MT > 1) afp[1] = *(const v4i *)(smem + abase[1][acol2] + soff + ko2);
                        }
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            acc[m][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[m], bfb[cur][tt], acc[m][tt], 0, 0, 0);
                    }
                    }
                }
            }
            // 9 steps: the last one (cur = 0) read the next chunk's first fragments into bfb[1]
            if constexpr (ROLL) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) bfb[0][tt] = bfb[1][tt];
            }
            if constexpr (!STATIC) sl ^= 1;
        }

        if (Y355_DIAG12 && first) { nstamp = 24; stamp(); }
        pstamp(first ? 3 : 7);
        // ---- epilogue: integer pipeline (32-bit path of conv3x3.hip) into registers, then through the
        // slab slot that just died (slot sl ^ 1: `sl` already points at the next tile's chunk 0)
        {
            // the epilogue's lane-dependent addresses are re-derived here from an opaque copy of the thread id: computed once
            // per launch they would be live (or spilled) across the whole k-loop
            int tid_e = threadIdx.x;
            asm volatile("" : "+v"(tid_e));
            const int tid = tid_e, lane = tid_e & 63, li = lane & 15, g = lane >> 4;
            (void)lane;
            const int ncol = wn * (NT * 16) + li * NT;
            int bias[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bias[t] = FPE ? 0 : p.bias_t[nb * BN + ncol + t];
            constexpr int RPM = POOL ? 1 : 4;                   // staged rows per m-tile and lane
            // sh_l (a left requant shift; sh_r = 0 then) folded into the accumulator shift and the bias:
            // ((t' << sh_l) + hm1 + rb) >> sh_r with t' = max(t, t << lk) equals the same form on T = t << sh_l
            const int shl2 = rq.shl + rq.sh_l;
            int bias2[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bias2[t] = bias[t] << rq.sh_l;
            auto requant = [&](int v, int t) {
                int x = (v << shl2) + bias2[t];
                x = max(x, x << rq.lk);
                const int rb = (int)__builtin_amdgcn_ubfe((unsigned int)x, (unsigned int)rq.sh_r, (unsigned int)rq.bw);
                return (x + rq.hm1 + rb) >> rq.sh_r;
            };
            // FPE: t = acc + bias in integers (exact), then fp32; the two scales are powers of two built in scalar registers
            const float s_pos = __int_as_float((127 + rq.lk - rq.sh_r) << 23), s_neg = __int_as_float((127 - rq.sh_r) << 23);
            int biasf[NT];
            if constexpr (FPE) {
#pragma unroll
                for (int t = 0; t < NT; ++t) biasf[t] = 0;           // already in the accumulators (their initial value)
            }
            // VGPR operands: an SGPR source takes the fma off the fast issue path (scratch/ubench/valu_rates.hip: 3.0 -> 4.6 cycles)
            float spv = s_pos, snv = s_neg, mgv = RMAGIC;
            if constexpr (FPE) asm volatile("" : "+v"(spv), "+v"(snv), "+v"(mgv));
            // the two branches of the LeakyReLU, each M + rne(t * scale); y = max(pos, neg).  s_neg = s_pos / 8 here, so
            // y > M + 127 <=> pos > M + 127 and y < M - 127 <=> neg < M - 127: a running max / min of the branches (two
            // instructions per four outputs) tells whether anything was clamped
            auto requantf = [&](int v, int t, float &pos, float &neg) {
                const float tf = (float)(v + biasf[t]);
                pos = fmaf(tf, spv, mgv);
                neg = fmaf(tf, snv, mgv);
                return 0.f;
            };
            float ymx = RMAGIC, ymn = RMAGIC;
            unsigned int satx = 0;                              // sum of (clamped ^ unclamped): non-zero iff something saturated
            if (Y355_DIAG12 && first) stamp();
            char *stg = smem + (STATIC ? 1 : (sl ^ 1)) * SLABB;
            const int halo = p.out_halo;
            constexpr int OTW = POOL ? TW / 2 : TW;
            const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
            const int oy0 = POOL ? (y0 >> 1) : y0, ox0 = POOL ? (x0 >> 1) : x0;
            int8_t *outb = p.out + (size_t)b * (Ho + 2 * halo) * (Wo + 2 * halo) * p.cstride + nb * BN;
            if constexpr (DIRECT) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < RPM; ++r) {
                        const int srow = POOL ? (wm * MT + m) * 4 + g : (wm * MT + m) * 16 + 4 * g + r;
                        unsigned int w = 0;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            int v;
                            if constexpr (POOL) {
                                const v4i a = acc[m][t];
                                v = max(max(a[0], a[1]), max(a[2], a[3]));
                            } else {
                                v = acc[m][t][r];
                            }
                            const int qq = y355_requant_fast(v, bias[t], rq);
                            const int q = y355_clamp8<int>(qq);
                            nsat += (srow < OROWS && oy0 + srow / OTW < Ho && ox0 + srow % OTW < Wo && q != qq) ? 1u : 0u;   // rows of an edge tile beyond the map are not outputs
                            w |= (unsigned int)(q & 0xff) << (8 * t);
                        }
                        const int oy = oy0 + srow / OTW, ox = ox0 + srow % OTW;
                        int8_t *dst = outb + ((size_t)(oy + halo) * (Wo + 2 * halo) + ox + halo) * p.cstride + ncol;
                        if (!(srow < OROWS && oy < Ho && ox < Wo)) dst = p.sink + tid * 4;   // keeps the store count static
                        *(unsigned int *)dst = w;
                        __builtin_amdgcn_sched_barrier(0);       // one row at a time: short live ranges
                    }
            } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the last chunk's reads have returned
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                if (ps > 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's copy-out reads
                    __builtin_amdgcn_s_barrier();
                }
                // pass ps stages m-tiles [ps*MH, (ps+1)*MH) of EVERY wave row (all waves requantise in
                // both passes); local row = (wm*MH + m - ps*MH) * RPMT + (row inside the m-tile)
                constexpr int MH = MT / NPASS;
                constexpr int RPMT = POOL ? 4 : 16;                 // staged rows per m-tile
                // FPE: the hot pass stages UNCLAMPED low bytes (the LeakyReLU's max writes its byte straight into the packed word:
                // SDWA, no med3 / pack) and tracks the two branches; when one left [-127, 127] (rare) this wave stages the pass
                // again, clamped, before the barrier -- same rows, no store: the counted waits see the same operations
                auto stage = [&](auto clampc) {
                    constexpr bool CL = decltype(clampc)::value;
#pragma unroll
                    for (int mm = 0; mm < MH; ++mm) {
                        const int m = ps * MH + mm;
#pragma unroll
                        for (int r = 0; r < RPM; ++r) {
                            const int lrow = POOL ? (wm * MH + mm) * 4 + g : (wm * MH + mm) * 16 + 4 * g + r;
                            unsigned int w = 0;
                            float yq[NT], ypos[NT], yneg[NT];
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                int v;
                                if constexpr (POOL) {
                                    const v4i a = acc[m][t];
                                    v = max(max(a[0], a[1]), max(a[2], a[3]));
                                } else {
                                    v = acc[m][t][r];
                                }
                                if constexpr (FPE) {
                                    if constexpr (CL) asm volatile("" : "+v"(v));      // recomputed here: nothing of the hot pass stays live
                                    yq[t] = requantf(v, t, ypos[t], yneg[t]);
                                } else {
                                    const int qq = requant(v, t);
                                    const int q = y355_clamp8<int>(qq);
                                    satx += (unsigned int)(q ^ qq);     // v_xad_u32; the exact count is taken below, rarely
                                    w |= (unsigned int)(q & 0xff) << (8 * t);
                                }
                            }
                            if constexpr (FPE) {
                                static_assert(!FPE || NT == 4, "the fp32 epilogue packs four channels per lane and row");
                                if constexpr (CL) {
                                    float yc[NT];
#pragma unroll
                                    for (int t = 0; t < NT; ++t) yc[t] = __builtin_amdgcn_fmed3f(rvmax(ypos[t], yneg[t]), RQLO, RQHI);
                                    w = rpack4(yc[0], yc[1], yc[2], yc[3]);
                                } else {
                                    ymx = rvmax3(rvmax3(ymx, ypos[0], ypos[1]), ypos[2], ypos[3]);
                                    ymn = rvmin3(rvmin3(ymn, yneg[0], yneg[1]), yneg[2], yneg[3]);
                                    rmax_to_byte<0>(w, ypos[0], yneg[0]);
                                    rmax_to_byte<1>(w, ypos[1], yneg[1]);
                                    rmax_to_byte<2>(w, ypos[2], yneg[2]);
                                    rmax_to_byte<3>(w, ypos[3], yneg[3]);
                                }
                            }
                            *(unsigned int *)(stg + lrow * SSTR + ncol) = w;
                            if constexpr (FPE) __builtin_amdgcn_sched_barrier(0);   // one row at a time: short live ranges (no spill)
                        }
                    }
                };
                stage(std::false_type{});
                if constexpr (FPE) {
                    if (__builtin_amdgcn_ballot_w64(ymx > RQHI || ymn < RQLO) != 0ull) stage(std::true_type{});
                }
                if (Y355_DIAG12 && first) stamp();                   // requantised and staged (this wave)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (Y355_DIAG12 && first) stamp();                   // everybody's rows are staged
#pragma unroll
                for (int j = 0; j < NITP; ++j) {
                    const int it = min(tid + j * NTHR, RP * CG - 1);      // tail clamps: duplicates rewrite the same bytes
                    const int rl = it / CG, cg = it % CG;
                    const int row = ((rl / (MH * RPMT)) * MT + ps * MH) * RPMT + rl % (MH * RPMT);
                    const int oy = oy0 + row / OTW, ox = ox0 + row % OTW;
                    const v4i v = *(const v4i *)(stg + rl * SSTR + cg * 16);
                    int8_t *dst = outb + ((size_t)(oy + halo) * (Wo + 2 * halo) + ox + halo) * p.cstride + cg * 16;
                    if (!(row < OROWS && oy < Ho && ox < Wo)) dst = p.sink + tid * 16;
                    *(v4i *)dst = v;
                }
            }
            if constexpr (FPE) satx = (ymx > RQHI || ymn < RQLO) ? 1u : 0u;
            if (satx) {                                         // cold: count the clamped outputs of real rows exactly
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < RPM; ++r) {
                        const int srow = POOL ? (wm * MT + m) * 4 + g : (wm * MT + m) * 16 + 4 * g + r;
                        // a real output: inside the tile AND inside the map (round 5: the rows / columns of an edge tile beyond the
                        // map -- computed from the halo and from whatever lies behind it -- used to be counted when they clamped:
                        // 22 against the oracle's 20 on a 7 x 4 map, tests/test_gpu_parity.py::test_small_map_saturation_counts)
                        const bool real = srow < OROWS && oy0 + srow / OTW < Ho && ox0 + srow % OTW < Wo;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            int v;
                            if constexpr (POOL) {
                                const v4i a = acc[m][t];
                                v = max(max(a[0], a[1]), max(a[2], a[3]));
                            } else {
                                v = acc[m][t][r];
                            }
                            if constexpr (FPE) {
                                asm volatile("" : "+v"(v));        // recompute here: do not keep the hot pass's 96 values alive for this branch
                                float ypc, ync;
                                (void)requantf(v, t, ypc, ync);
                                const float y = rvmax(ypc, ync);
                                nsat += (real && (y > RQHI || y < RQLO)) ? 1u : 0u;
                            } else {
                                const int qq = requant(v, t);
                                nsat += (real && y355_clamp8<int>(qq) != qq) ? 1u : 0u;
                            }
                        }
                    }
            }
        }
        }
        if (Y355_DIAG12 && first) stamp();
        pstamp(first ? 4 : 8);
        first = false;
        if (!more) break;
        tile = ntile;
        b = b2; y0 = y2; x0 = x2; nb = nb2;
    }
    rwait_vmcnt<0>();       // retire the prefetches before the wave ends
    pstamp(5);
    if (Y355_DIAG12) { nstamp = 28; stamp(); }
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
}

// ------------------------------------------------------------------------------------------
template <int CIN, int BN, int TH, int TW, bool POOL, int WM, int WN, int PF, bool ROLL, bool DIRECT, bool FPE>
struct ConvInstR {
    static constexpr int PWL = (TW + 2 + 7) / 8 * 8;
    static constexpr int SLABB = ((TH + 2) * PWL * 64 + 1023) / 1024 * 1024;
    static constexpr int WB = (BN / 16) * 1024;
    static constexpr size_t LDS = 2 * (size_t)SLABB + (size_t)(PF + 2) * WB + 1024 + (FPE ? 1024 : 0);
    static int prepare() {
        return (int)hipFuncSetAttribute((const void *)conv3x3_i8_ring_kernel<CIN, BN, TH, TW, POOL, WM, WN, PF, ROLL, DIRECT, FPE>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    static bool launch(const ConvParams &p_in, hipStream_t s) {
        ConvParams p = p_in;
        p.tiles_x = (p.W + TW - 1) / TW;
        p.tiles_y = (p.H + TH - 1) / TH;
        const int total = p.tiles_x * p.tiles_y * p.nblk * p.B;
        {   // items that read one image's input and are few enough to sit on one XCD together: 2 or 4 (see `decode`)
            const int per_image = p.tiles_x * p.tiles_y * p.nblk;
            p.xcd_share_log2 = !Y355_RING_XCD_SHARE ? 0 : (per_image == 8 && WM * WN == 4) ? 3 : per_image == 4 ? 2 : (per_image == 2 || p.nblk == 2) ? 1 : 0;
        }
        constexpr int PER_CU = (WM * WN == 4) ? 2 : 1;                     // four-wave instantiations: two workgroups share a CU
        int grid = y355_cu_count() * PER_CU;                               // one persistent 8-wave workgroup per CU
        if (p.grid_limit > 0 && p.grid_limit * PER_CU < grid) grid = p.grid_limit * PER_CU;   // fewer, each walking more tiles (throughput mode)
        if (grid > total) grid = total;
        if (p.ev_start && p.ev_stop) {
            hipEvent_t e0 = (hipEvent_t)p.ev_start, e1 = (hipEvent_t)p.ev_stop;
            p.ev_start = p.ev_stop = nullptr;
            hipExtLaunchKernelGGL((conv3x3_i8_ring_kernel<CIN, BN, TH, TW, POOL, WM, WN, PF, ROLL, DIRECT, FPE>), dim3(grid), dim3(WM * WN * 64), LDS, s,
                                  e0, e1, 0, p, total);
        } else {
            hipLaunchKernelGGL((conv3x3_i8_ring_kernel<CIN, BN, TH, TW, POOL, WM, WN, PF, ROLL, DIRECT, FPE>), dim3(grid), dim3(WM * WN * 64), LDS, s, p, total);
        }
        return true;
    }
};

// must mirror the tile table of conv3x3.hip (same packing: BN, WN and NT are shared)
template <bool ROLL, bool DIRECT, bool FPE>
struct RSet {
    using C3_2 = ConvInstR<64, 64, 26, 26, true, 8, 1, Y355_RING_PF, ROLL, DIRECT, FPE>;
    using C4_1 = ConvInstR<64, 128, 13, 26, false, 4, 2, Y355_RING_PF, ROLL, DIRECT, FPE>;
    using C4_2 = ConvInstR<128, 64, 26, 26, true, 8, 1, Y355_RING_PF, ROLL, DIRECT, FPE>;
    using C5 = ConvInstR<128, 128, 13, 26, false, 4, 2, Y355_RING_PF, ROLL, DIRECT, FPE>;
#if Y355_RING_C67_SMALL
    using C67 = ConvInstR<256, 128, 13, 13, false, 2, 2, Y355_RING_PF, ROLL, DIRECT, FPE>;   // experiment: two 4-wave workgroups per CU
#else
    using C67 = ConvInstR<256, 128, 13, 26, false, 4, 2, Y355_RING_PF, ROLL, DIRECT, FPE>;
#endif
    using PRED = ConvInstR<256, 64, 13, 13, false, 8, 1, Y355_RING_PF, ROLL, DIRECT, FPE>;
    // pred in the throughput mode (several handles share the GPU: p.grid_limit > 0): 13 x 26 tiles = 128 work items at B = 64, same
    // weight packing.  A launch then holds 128 CUs for ~15 us instead of 256 for ~12 (the launch is mostly start-up and drain,
    // profiles/r03_notes.md): three handles 278.3 k -> 280.9 k img/s over nine interleaved runs (whole maps, 64 items: no better)
    using PRED_W = ConvInstR<256, 64, 13, 26, false, 8, 1, Y355_RING_PF, ROLL, DIRECT, FPE>;
    static int prepare() {
        int e = C3_2::prepare();
        if (!e) e = C4_1::prepare();
        if (!e) e = C4_2::prepare();
        if (!e) e = C5::prepare();
        if (!e) e = C67::prepare();
        if (!e) e = PRED::prepare();
        if (!e) e = PRED_W::prepare();
        return e;
    }
    static bool launch(int kid, const ConvParams &p, hipStream_t s) {
        switch (kid) {
        case Y355_K_CONV3_2: return C3_2::launch(p, s);
        case Y355_K_CONV4_1: return C4_1::launch(p, s);
        case Y355_K_CONV4_2: return C4_2::launch(p, s);
        case Y355_K_CONV5: return C5::launch(p, s);
        case Y355_K_CONV67: return C67::launch(p, s);
        case Y355_K_PRED: return p.grid_limit > 0 ? PRED_W::launch(p, s) : PRED::launch(p, s);
        default: return false;
        }
    }
};

int y355_prepare_conv_ring(void) {
    const int e = RSet<false, false, false>::prepare();
    return e ? e : RSet<false, false, true>::prepare();
}

bool y355_launch_conv_ring(int kid, const ConvParams &p, hipStream_t s) {
    if ((p.mode & 0xff) != 0 || p.rq.wide || p.guard) return false;   // those go to conv3x3.hip
    // ROLL = false (chunk loop unrolled) and staged epilogue: measured best of the four combinations (144.5 k img/s vs 140-142 k,
    // one stream, B = 64, round 1); round 3's operand-swapped form with 16-byte stores straight from the accumulators
    // (scratch/ring_experiments/conv3x3_ring_r3_swap.hip) shortens the epilogue by 0.9 us and changes nothing end to end
    // fp32 epilogue where the host can prove it exact (header of the kernel): no accumulator / left requant shift, a right shift
    // of at most 17 bits, the reference's slope
#ifndef Y355_RING_NO_FPE
    if (p.rq.shl == 0 && p.rq.sh_l == 0 && p.rq.sh_r <= 17 && p.rq.neg_mul == 1 && p.cstride >= 4 && p.cstride <= 256)
        return RSet<false, false, true>::launch(kid, p, s);
#endif
    return RSet<false, false, false>::launch(kid, p, s);
}
