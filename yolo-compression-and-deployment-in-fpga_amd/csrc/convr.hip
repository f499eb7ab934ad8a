// yolo355 -- 3x3 convolutions of the generic nets (csrc/net.hip) on the LDS-DMA ring discipline of conv3x3_ring.hip, in both
// arithmetic types of convg.hip: bf16 x bf16 -> fp32 for the fp32 model families (utils.modules.Conv2d / Conv_BN_LeakyReLU,
// models/slim_yolo_v2.py:386-622, backbone/darknet.py:12-22) and int8 x int8 -> int32 with the general-slope integer
// epilogue for the quantized YOLOv3tiny (models/tiny_yolo_v3.py:9-273).  VERDICT r2 item 6.
//
// convg8_kernel (convg.hip) stages the activations through registers into LDS and reads every B fragment straight from
// global memory, two k-steps ahead: 16-32 KB per k-step and CU through the vector-memory pipe, every wave with the same
// output-channel range loading the same fragments again, MFMA pipe 27 % busy (profiles/r02_notes.md).  Here, as in the int8
// ring kernel: persistent 8-wave workgroups; 64-byte chunks of the input patch in a 2-slot LDS ring filled by LDS-DMA, one
// 1 KiB piece per wave and k-step; weights in a 7-slot LDS ring, five k-steps in flight, loaded ONCE per workgroup; counted
// vmcnt + one barrier per k-step; B fragments of step s + 1 read under the MFMAs of step s.  A k-step is 64 bytes of one
// tap: 32 bf16 channels (v_mfma_f32_16x16x32_bf16) or 64 int8 channels (v_mfma_i32_16x16x64_i8).  Same fragment order as convg.hip (y355_convg_pack with this kernel's
// BN / WN / NT), same MFMA sequence per output element: results are bit-identical to convg8_kernel's.
// Epilogue: bf16 -- bias + LeakyReLU slope in fp32 on the accumulators, 8-byte bf16 (or 16-byte fp32, prediction maps) stores
// of a lane's four adjacent channels straight from registers (16 lanes = one 128-byte line of a pixel); int8 -- convg.hip's
// integer pipeline (32-bit form where the host proved it fits, else 64-bit), 4-byte stores, saturation counted; 2x2 max first
// on pooled tiles; the store count per tile is static (rows outside the map go to a sink) so that the next tile's counted waits hold.
#include "y355_common.h"
#include <mutex>
#include <hip/hip_ext.h>
#include <type_traits>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ void fglds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
__device__ __forceinline__ void fwait_vmcnt_dyn(int n) {
#define FW_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 63 ? 63 : n)) {
        FW_CASE(0) FW_CASE(1) FW_CASE(2) FW_CASE(3) FW_CASE(4) FW_CASE(5) FW_CASE(6) FW_CASE(7) FW_CASE(8) FW_CASE(9)
        FW_CASE(10) FW_CASE(11) FW_CASE(12) FW_CASE(13) FW_CASE(14) FW_CASE(15) FW_CASE(16) FW_CASE(17) FW_CASE(18)
        FW_CASE(19) FW_CASE(20) FW_CASE(21) FW_CASE(22) FW_CASE(23) FW_CASE(24) FW_CASE(25) FW_CASE(26) FW_CASE(27)
        FW_CASE(28) FW_CASE(29) FW_CASE(30) FW_CASE(31) FW_CASE(32) FW_CASE(33) FW_CASE(34) FW_CASE(35) FW_CASE(36)
        FW_CASE(37) FW_CASE(38) FW_CASE(39) FW_CASE(40) FW_CASE(41) FW_CASE(42) FW_CASE(43) FW_CASE(44) FW_CASE(45)
        FW_CASE(46) FW_CASE(47) FW_CASE(48) FW_CASE(49) FW_CASE(50) FW_CASE(51) FW_CASE(52) FW_CASE(53) FW_CASE(54)
        FW_CASE(55) FW_CASE(56) FW_CASE(57) FW_CASE(58) FW_CASE(59) FW_CASE(60) FW_CASE(61) FW_CASE(62) FW_CASE(63)
    }
#undef FW_CASE
}
// slab pieces issued in steps lo..hi (step u issues one when 1 <= (u mod 9) <= ppw); negative steps are the previous tile's
constexpr int fring_sp(int lo, int hi, int ppw, bool prev) {
    int n = 0;
    for (int u = lo; u <= hi; ++u) {
        if (u < 0 && !prev) continue;
        const int t = ((u % 9) + 9) % 9;
        if (t >= 1 && t <= ppw) ++n;
    }
    return n;
}
constexpr int PF = 4;        // ring of PF + 2 = 6 weight slots: divides the k-steps of every multi-chunk layer here (STATIC below)
}  // namespace

// CINB = bytes per input pixel (a multiple of 64), BN output channels per workgroup
template <bool BF, int CINB, int BN, int TH, int TW, bool POOL, int WM, int WN, bool NARROW>
__global__ __launch_bounds__(WM * WN * 64, 1) void convr_kernel(const ConvGParams p, const int total_tiles, char *sink) {
    using ACC = typename std::conditional<BF, v4f, v4i>::type;
    constexpr int NW = WM * WN;
    constexpr int NCH = CINB / 64, SPC = 9, KS = NCH * SPC;
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
    constexpr int NIT = POOL ? MT : MT * 4;                        // output stores per thread per tile (static)
    static_assert(CINB % 64 == 0 && NT == 4 && NW == 8, "64-byte chunks, four n-tiles per wave, eight waves");
    // STATIC (conv3x3_ring.hip): ring and slab slots are compile-time constants of the unrolled loop when the ring's length divides
    // the tile's k-steps and the chunk count is even; the k-step's read / MFMA order is pinned below
    constexpr bool STATIC = KS % WSLOTS == 0 && NCH % 2 == 0;
    static_assert(PPW <= 8, "slab pieces go out at t = 1..PPW");
    static_assert((NW * 16) % PWL == 0 && NPIX % 16 == 0, "slab pieces step by whole patch rows");
    static_assert(!POOL || (TH % 2 == 0 && TW % 2 == 0), "pooled tiles are even");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;

    constexpr int PSTEP = NW * 16 / PWL;
    const int pix0 = wave * 16 + (lane >> 2);
    const int ppy0 = pix0 / PWL, ppx0 = min(pix0 % PWL, PW - 1);     // pitch padding re-reads column PW-1
    const int pwithin = ((lane & 3) ^ ((lane >> 3) & 3)) << 4;
    auto decode = [&](int tile, int &b, int &y0, int &x0, int &nb) {
        // the work items that read one input (the n-blocks of a tile, the tiles of an image) go to workgroups of ONE XCD: the
        // permutation of conv3x3_ring.hip's walk (see there), share = 2^xcd_share_log2 set by the launcher
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
        const char *inb = p.in + (size_t)b * (H + 2) * (W + 2) * CINB + c * 64;
        const int gy = min(y0 + ppy0 + j * PSTEP, H + 1), gx = min(x0 + ppx0, W + 1);
        const char *src = inb + ((size_t)gy * (W + 2) + gx) * CINB + pwithin;   // pad pieces read a valid row too
        char *dst = (q < NPIECE) ? smem + slot * SLABB + q * 1024 : smem + OFF_DUMMY;
        fglds16(src, dst);
    };
    auto issue_w = [&](int nb, int ks, int slot) {
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int f = wave + NW * j;
            const bool ok = f < NFR;
            const char *src = p.w + ((size_t)(nb * KS + ks) * NFR + (ok ? f : 0)) * 1024 + lane * 16;
            char *dst = ok ? smem + OFF_W + slot * WB + f * 1024 : smem + OFF_DUMMY;
            fglds16(src, dst);
        }
    };
    auto wrap = [](int s) { return s >= WSLOTS ? s - WSLOTS : s; };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int b, y0, x0, nb;
    decode(tile, b, y0, x0, nb);
    int sl = 0;                                                // slab slot of the current chunk
    int wq = 0;                                                // ring slot of W(s) at step s
    // ---- prologue: slab 0 whole, then W(0) .. W(PF)
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_slab_piece(b, y0, x0, 0, 0, j);
#pragma unroll
    for (int k = 0; k <= PF; ++k) {
        if (k < KS) issue_w(nb, k, k);
        else issue_w(nb, k - KS, k);                           // KS > PF for every layer here; keeps counts static
    }
    const float slope = p.slope;
    const RequantG rq = p.rq;
    const int halo = p.out_halo;
    unsigned int nsat = 0;
    bool first = true;

    for (;;) {
        int ntile = tile + gridDim.x;
        const bool more = ntile < total_tiles;
        if (!more) ntile = tile;                               // keep the operation counts static
        int b2, y2, x2, nb2;
        decode(ntile, b2, y2, x2, nb2);
        // per-lane A-fragment bases, per tile from an opaque copy of the lane id (not live across the epilogue)
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

        ACC acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (BF) acc[m][t] = (v4f){0.f, 0.f, 0.f, 0.f};
                else acc[m][t] = (v4i){0, 0, 0, 0};
            }

        // ---- pre-phase: publish W(0) (and slab 0) and read W(0)'s B fragments (vmcnt is in issue order: younger than W(0)
        // are W(1..PF), the slab pieces issued with them, and the previous tile's NIT output stores)
        v4i bfb[2][NT];
        v4i afp[2];
        if (first) fwait_vmcnt_dyn(PF * WPW);
        else fwait_vmcnt_dyn(PF * WPW + fring_sp(-PF, -1, PPW, true) + NIT);
        __builtin_amdgcn_s_barrier();
        {
            const char *wb0 = smem + OFF_W + (STATIC ? 0 : wq) * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) bfb[0][tt] = *(const v4i *)(wb0 + tt * 1024);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int slc = STATIC ? (c & 1) : sl;
            const int soff = slc * SLABB;
            const bool lastc = (c + 1 == NCH);
#pragma unroll
            for (int t = 0; t < SPC; ++t) {
                const int s_idx = c * SPC + t;
                // W(s+1) has landed (own pieces) -> barrier -> everybody's has; at a chunk's first step the slab too
                {
                    constexpr int n_slab = (9 - PPW) * WPW;
                    const int lo = s_idx - PF + 1, hi = s_idx - 1;
                    int n_first = (PF - 1) * WPW + fring_sp(lo, hi, PPW, false);
                    int n_later = (PF - 1) * WPW + fring_sp(lo, hi, PPW, true) + (s_idx < PF ? NIT : 0);
                    if (t == 0) {
                        if (c > 0 && n_slab < n_first) n_first = n_slab;
                        const int n_slab2 = n_slab + (c == 0 ? NIT : 0);
                        if (n_slab2 < n_later) n_later = n_slab2;
                    }
                    if (n_first == n_later) fwait_vmcnt_dyn(n_later);
                    else if (first) fwait_vmcnt_dyn(n_first);
                    else fwait_vmcnt_dyn(n_later);
                }
                __builtin_amdgcn_s_barrier();
                // refill: one slab piece (t = 1..PPW) into the slot that died two barriers ago, W(s+1+PF) into the ring slot
                // read in step s-2
                const int wqs = STATIC ? s_idx % WSLOTS : wq;
                {
                    if (t >= 1 && t <= PPW)
                        issue_slab_piece(lastc ? b2 : b, lastc ? y2 : y0, lastc ? x2 : x0, lastc ? 0 : c + 1, slc ^ 1, t - 1);
                    const int ksn = s_idx + 1 + PF;
                    const bool nxt = ksn >= KS;
                    issue_w(nxt ? nb2 : nb, nxt ? ksn - KS : ksn, wrap(wqs + PF + 1));
                }
                const int ko = (t / 3) * PWL * 64;
                const int acol = t % 3;
                const int cur = s_idx & 1;
                if constexpr (!STATIC) wq = wrap(wq + 1);
                v4i af[MT];
                if (t == 0) {
                    af[0] = *(const v4i *)(smem + abase[0][acol] + soff + ko);
                    if constexpr (MT > 1) af[1] = *(const v4i *)(smem + abase[1][acol] + soff + ko);
                } else {
                    af[0] = afp[0];
                    if constexpr (MT > 1) af[1] = afp[1];
                }
                auto mf = [&](int m) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        if constexpr (BF)
                            acc[m][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[m]), __builtin_bit_cast(v8bf, bfb[cur][tt]),
                                                                                 acc[m][tt], 0, 0, 0);
                        else
                            acc[m][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[m], bfb[cur][tt], acc[m][tt], 0, 0, 0);
                    }
                };
                if constexpr (STATIC) {
                    // pinned order (conv3x3_ring.hip): this step's remaining A fragments, the MFMAs of the two m-tiles whose
                    // fragments are already here, then the next step's B fragments and first A fragments two reads per m-tile.
                    // Plain scheduling fences between the groups: the sched_group_barrier pipelines of conv3x3_ring.hip take the
                    // compiler tens of minutes on this file's 72- and 144-step loops.
#pragma unroll
                    for (int m = 2; m < MT; ++m) af[m] = *(const v4i *)(smem + abase[m][acol] + soff + ko);
                    mf(0);
                    if constexpr (MT > 1) mf(1);
                    __builtin_amdgcn_sched_barrier(0);
                    const char *wbn = smem + OFF_W + wrap(wqs + 1) * WB + (wn * NT) * 1024 + lane * 16;
                    const int ko2 = ((t + 1) / 3) * PWL * 64;
                    const int acol2 = (t + 1) % 3;
                    constexpr int RDPM = MT > 2 ? (NT + 2 + MT - 3) / (MT - 2) : 0;   // reads per m-tile behind the first two
#pragma unroll
                    for (int m = 2; m < MT; ++m) {
                        // this m-tile's share of the next step's list of reads: B0 .. B3, A0, A1
#pragma unroll
                        for (int k = RDPM * (m - 2); k < RDPM * (m - 2) + RDPM; ++k) {
                            if (k < NT) {
                                if (s_idx + 1 < KS) bfb[cur ^ 1][k] = *(const v4i *)(wbn + k * 1024);
                            } else if (k - NT < 2 && k - NT < MT && t + 1 < SPC) {
                                afp[k - NT] = *(const v4i *)(smem + abase[k - NT][acol2] + soff + ko2);
                            }
                        }
                        mf(m);
                        __builtin_amdgcn_sched_barrier(0);
                    }

                    if constexpr (MT <= 2) {                          // no m-tile behind the first two: the reads go out here
                        if (s_idx + 1 < KS) {
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) bfb[cur ^ 1][tt] = *(const v4i *)(wbn + tt * 1024);
                        }
                        if (t + 1 < SPC) {
                            afp[0] = *(const v4i *)(smem + abase[0][acol2] + soff + ko2);
                            if constexpr (MT > 1) afp[1] = *(const v4i *)(smem + abase[1][acol2] + soff + ko2);
                        }
                    }
                } else {
                    if (s_idx + 1 < KS) {                        // B fragments of step s+1, under this step's MFMAs
                        const char *wbn = smem + OFF_W + wrap(wqs + 1) * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) bfb[cur ^ 1][tt] = *(const v4i *)(wbn + tt * 1024);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (m + 2 < MT) af[m + 2] = *(const v4i *)(smem + abase[m + 2][acol] + soff + ko);
                        if (m == MT - 1 && t + 1 < SPC) {        // next step's first A fragments (same slab)
                            const int ko2 = ((t + 1) / 3) * PWL * 64;
                            const int acol2 = (t + 1) % 3;
                            afp[0] = *(const v4i *)(smem + abase[0][acol2] + soff + ko2);
                            if constexpr (MT > 1) afp[1] = *(const v4i *)(smem + abase[1][acol2] + soff + ko2);
                        }
                        mf(m);
                    }
                }
            }
            if constexpr (!STATIC) sl ^= 1;
        }

        // ---- epilogue straight from the registers (convg.hip's arithmetic, expression for expression)
        {
            int li_e = li, g_e = g;
            asm volatile("" : "+v"(li_e), "+v"(g_e));
            const int nlane = nb * BN + wn * (NT * 16) + li_e * NT;   // first of this lane's NT channels
            const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
            char *outb = p.out + (size_t)b * (Ho + 2 * halo) * (Wo + 2 * halo) * p.out_pb + p.out_off;
            char *snk = sink + tid * 16;
            float biasf[NT];
            long long biasw[NT];
            int biasn[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (BF) { biasf[t] = p.bias_f[nlane + t]; biasw[t] = 0; }
                else { biasw[t] = p.bias_w[nlane + t]; biasf[t] = 0.f; }
                biasn[t] = (int)biasw[t];
            }
            Requant rqn{};
            rqn.shl = rq.shl; rqn.sh = rq.sh; rqn.lk = rq.lk; rqn.neg_mul = rq.neg_mul; rqn.split = rq.split;
            auto finish = [&](const float (&vf)[NT], const int (&vi)[NT], bool valid, int oy, int ox) {
                char *dst = outb + ((size_t)(oy + halo) * (Wo + 2 * halo) + ox + halo) * p.out_pb;
                if constexpr (BF) {
                    float y[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float x = vf[t] + biasf[t];
                        y[t] = x >= 0.f ? x : x * slope;
                    }
                    if (p.out_f32) {
                        dst = valid ? dst + (size_t)nlane * 4 : snk;
                        *(v4f *)dst = (v4f){y[0], y[1], y[2], y[3]};
                    } else {
                        dst = valid ? dst + (size_t)nlane * 2 : snk;
                        unsigned short hh[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) hh[t] = __builtin_bit_cast(unsigned short, (__bf16)y[t]);
                        uint2 u;
                        u.x = (unsigned int)hh[0] | ((unsigned int)hh[1] << 16);
                        u.y = (unsigned int)hh[2] | ((unsigned int)hh[3] << 16);
                        *(uint2 *)dst = u;
                    }
                } else {
                    int q[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (NARROW) {
                            const int qq = y355_requant_gen32(vi[t], biasn[t], rqn);
                            q[t] = y355_clamp8<int>(qq);
                            nsat += (valid && q[t] != qq) ? 1u : 0u;
                        } else {
                            static_assert(BF || NARROW, "the 64-bit epilogue lives in convg.hip only");
                            q[t] = 0;
                        }
                    }
                    dst = valid ? dst + nlane : snk;
                    *(unsigned int *)dst = (unsigned int)((q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24));
                }
            };
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (POOL) {
                    const int w = (wm * MT + m) * 4 + g_e;          // monotone epilogue: pool the raw accumulators first
                    const int wy = w / (TW / 2), wx = w % (TW / 2);
                    const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
                    const bool valid = (w * 4 < BM) && oy < Ho && ox < Wo;
                    float vf[NT];
                    int vi[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const ACC a = acc[m][t];
                        if constexpr (BF) { vf[t] = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])); vi[t] = 0; }
                        else { vi[t] = max(max(a[0], a[1]), max(a[2], a[3])); vf[t] = 0.f; }
                    }
                    finish(vf, vi, valid, oy, ox);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = (wm * MT + m) * 16 + 4 * g_e + r;
                        const int oy = y0 + row / TW, ox = x0 + row % TW;
                        const bool valid = row < BM && oy < Ho && ox < Wo;
                        float vf[NT];
                        int vi[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            if constexpr (BF) { vf[t] = acc[m][t][r]; vi[t] = 0; }
                            else { vi[t] = acc[m][t][r]; vf[t] = 0.f; }
                        }
                        finish(vf, vi, valid, oy, ox);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        first = false;
        if (!more) break;
        tile = ntile;
        b = b2; y0 = y2; x0 = x2; nb = nb2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // retire the prefetches before the wave ends
    if constexpr (!BF) {
        if (nsat && p.ctr) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
    }
}

// ------------------------------------------------------------------------------------------
// 1x1 convolutions of the int8 nets (conv_1x1_2, pred_1, pred_2 of YOLOv3tiny, models/tiny_yolo_v3.py:31-39): convg.hip stages a
// 3x3-haloed patch per tile for them (26-36 us per launch for 5-13 MMAC per image).  Here a workgroup keeps the fragments of ONE
// block of 64 output channels in LDS (all k-steps), a wave takes 16 consecutive pixels of the batch at a time, reads their
// channels straight from global memory as A fragments (all k-steps in flight together) and stores four adjacent channels of
// four pixels per lane.  Fragment and k order are convg.hip's, the epilogue is its 32-bit general-slope form: bit-identical.
template <int MAXKS>
__global__ __launch_bounds__(256) void pw_i8_kernel(const ConvGParams p, const int npix, const int ks_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int nb = blockIdx.x % p.nblk, wg = blockIdx.x / p.nblk, nwg = gridDim.x / p.nblk;
    const char *wsrc = p.w + (size_t)nb * ks_n * 4 * 1024;
    for (int i = tid; i < ks_n * 4 * 64; i += 256) *(v4i *)(smem + i * 16) = *(const v4i *)(wsrc + (size_t)i * 16);
    __syncthreads();
    const int H = p.H, W = p.W, HW = H * W;
    int biasn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) biasn[t] = (int)p.bias_w[nb * 64 + li * 4 + t];
    Requant rqn{};
    rqn.shl = p.rq.shl; rqn.sh = p.rq.sh; rqn.lk = p.rq.lk; rqn.neg_mul = p.rq.neg_mul; rqn.split = p.rq.split;
    unsigned int nsat = 0;
    const int ngroups = (npix + 15) / 16;
    for (int grp = wg * 4 + wave; grp < ngroups; grp += nwg * 4) {
        const int pi = min(grp * 16 + li, npix - 1);
        const int b = pi / HW, r = pi - b * HW, y = r / W, x = r - y * W;
        const char *src = p.in + (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * p.in_pb + g * 16;
        v4i a[MAXKS];
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < ks_n) a[ks] = *(const v4i *)(src + ks * 64);
        v4i acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < ks_n) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const v4i bq = *(const v4i *)(smem + ((ks * 4 + t) * 64 + lane) * 16);
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks], bq, acc[t], 0, 0, 0);
                }
            }
        }
        // row 4 g + rr of the group = pixel, column li = channels 64 nb + 4 li .. + 3 (n-tile t = channel 4 li + t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int po = grp * 16 + 4 * g + rr;
            if (po >= npix) continue;
            const int bo = po / HW, ro = po - bo * HW, yo = ro / W, xo = ro - yo * W;
            char *dst = p.out + (((size_t)bo * (H + 2 * p.out_halo) + yo + p.out_halo) * (W + 2 * p.out_halo) + xo + p.out_halo) * p.out_pb + p.out_off +
                        nb * 64 + li * 4;
            int q[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int qq = y355_requant_gen32(acc[t][rr], biasn[t], rqn);
                q[t] = y355_clamp8<int>(qq);
                nsat += q[t] != qq ? 1u : 0u;
            }
            *(unsigned int *)dst = (unsigned int)((q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24));
        }
    }
    if (nsat && p.ctr) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
}

// p.w = y355_convg_pack order for (bn 64, wn 1, nt 4), taps 1, p.nblk = cout_pad / 64; false = not launched
bool y355_launch_pw_i8(const ConvGParams &p, hipStream_t s) {
    if (p.taps != 1 || p.res || !p.bias_w || !p.rq.narrow || p.nblk < 1 || p.in_pb % 64 || p.in_pb > 1024) return false;
    if (((p.out_off | p.out_pb) & 3) != 0) return false;
    if ((long long)p.B * p.H * p.W >= (1ll << 30)) return false;
    const int ks_n = p.in_pb / 64, npix = p.B * p.H * p.W;
    int per_nb = (npix + 63) / 64;                               // workgroups per n-block: 64 pixels per pass of its four waves
    if (per_nb * p.nblk > 2048) per_nb = 2048 / p.nblk;
    const size_t lds = (size_t)ks_n * 4 * 1024;
    if (ks_n <= 8) hipLaunchKernelGGL(pw_i8_kernel<8>, dim3(per_nb * p.nblk), dim3(256), lds, s, p, npix, ks_n);
    else hipLaunchKernelGGL(pw_i8_kernel<16>, dim3(per_nb * p.nblk), dim3(256), lds, s, p, npix, ks_n);
    return true;
}

// ------------------------------------------------------------------------------------------
namespace {
char *g_sink[16] = {};                                         // per device: where the masked rows' stores go

template <bool BF, int CINB, int BN, int TH, int TW, bool POOL, int WM, int WN>
struct ConvRInst {
    static constexpr int PWL = (TW + 2 + 7) / 8 * 8;
    static constexpr int SLABB = ((TH + 2) * PWL * 64 + 1023) / 1024 * 1024;
    static constexpr int WB = (BN / 16) * 1024;
    static constexpr size_t LDS = 2 * (size_t)SLABB + (size_t)(PF + 2) * WB + 1024;
    static_assert(LDS <= 160 * 1024, "LDS");
    static constexpr Y355ConvRInfo info() { return Y355ConvRInfo{BF ? 1 : 0, CINB, BN, TH, TW, POOL ? 1 : 0, WN, BN / 16 / WN}; }
    static int prepare() {
        return (int)hipFuncSetAttribute((const void *)convr_kernel<BF, CINB, BN, TH, TW, POOL, WM, WN, !BF>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    static bool launch(const ConvGParams &p_in, int device, hipStream_t s) {
        ConvGParams p = p_in;
        p.tiles_x = (p.W + TW - 1) / TW;
        p.tiles_y = (p.H + TH - 1) / TH;
        const int total = p.tiles_x * p.tiles_y * p.nblk * p.B;
        {
            const int per_image = p.tiles_x * p.tiles_y * p.nblk;
            p.xcd_share_log2 = per_image == 4 ? 2 : (per_image == 2 || p.nblk == 2) ? 1 : 0;
        }
        const int cus = y355_cu_count(); int grid = total < cus ? total : cus;                 // one persistent workgroup per CU
        if (p.grid_limit > 0 && p.grid_limit < grid) grid = p.grid_limit;     // fewer, each walking more tiles (throughput mode)
        // int8: only the 32-bit epilogue is built here; a layer that needs the 64-bit one (rq.narrow == 0: 96 accumulators plus
        // 64-bit temporaries spill 1.2 KB per lane and the launch is 2.4x slower than convg8's, profiles/r03_notes.md) stays on convg.hip
        if (!BF && !p.rq.narrow) return false;
        hipLaunchKernelGGL((convr_kernel<BF, CINB, BN, TH, TW, POOL, WM, WN, !BF>), dim3(grid), dim3(WM * WN * 64), LDS, s, p, total, g_sink[device]);
        return true;
    }
};
//                         BF   CINB  BN  TH  TW  POOL  WM WN     SlimYOLOv2 fp32 (bf16: 2 bytes per channel)
using R0 = ConvRInst<true, 64, 64, 13, 26, false, 8, 1>;     // 32 ->  64                     conv3_1
using R1 = ConvRInst<true, 128, 64, 26, 26, true, 8, 1>;     // 64 ->  64, pooled             conv3_2
using R2 = ConvRInst<true, 128, 128, 13, 26, false, 4, 2>;   // 64 -> 128                     conv4_1
using R3 = ConvRInst<true, 256, 64, 26, 26, true, 8, 1>;     // 128 -> 128, pooled            conv4_2
using R4 = ConvRInst<true, 256, 128, 13, 26, false, 4, 2>;   // 128 -> 256                    conv5
using R5 = ConvRInst<true, 512, 128, 13, 26, false, 4, 2>;   // 256 -> 256                    conv6, conv7
using R6 = ConvRInst<true, 512, 64, 13, 13, false, 8, 1>;    // 256 -> <= 64 (fp32 out)       pred
//                                                              YOLOv3tiny int8 (1 byte per channel)
// (a layer that needs the 64-bit epilogue stays on convg8: in this kernel 96 accumulators + 64-bit temporaries spill)
using I0 = ConvRInst<false, 64, 64, 26, 26, true, 8, 1>;     // 64 -> 128, pooled             conv_4
using I1 = ConvRInst<false, 256, 256, 13, 13, false, 2, 4>;  // 256 -> 512 on 13 x 13 maps    extra_conv_2 (conv_6 when it fits 32 bits)
using I2 = ConvRInst<false, 128, 128, 13, 26, false, 4, 2>;  // 128 -> 256                    conv_5 (when it fits 32 bits)
using I3 = ConvRInst<false, 384, 128, 13, 26, false, 4, 2>;  // 384 -> 256                    conv_set_1
using I4 = ConvRInst<false, 512, 256, 13, 13, false, 2, 4>;  // 512 -> 1024 on 13 x 13 maps   conv_7
using I5 = ConvRInst<false, 1024, 128, 13, 13, false, 4, 2>; // 1024 -> 256 on 13 x 13 maps   conv_set_2
constexpr int NR = 13;
constexpr Y355ConvRInfo g_info[NR] = {R0::info(), R1::info(), R2::info(), R3::info(), R4::info(), R5::info(), I0::info(), I1::info(), R6::info(), I2::info(),
                                      I3::info(), I4::info(), I5::info()};
}  // namespace

int y355_prepare_convr(int device) {
    if (device < 0 || device >= 16) return 1;
    {
        // one sink per device for the life of the process, allocated once whichever handle gets here first (two
        // threads creating nets on one device raced on the pointer; the buffer only ever receives masked rows' stores)
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!g_sink[device]) {
            char *ptr = nullptr;
            if (hipMalloc((void **)&ptr, 16384) != hipSuccess) return 1;
            g_sink[device] = ptr;
        }
    }
    int e = R0::prepare();
    if (!e) e = R1::prepare();
    if (!e) e = R2::prepare();
    if (!e) e = R3::prepare();
    if (!e) e = R4::prepare();
    if (!e) e = R5::prepare();
    if (!e) e = I0::prepare();
    if (!e) e = I1::prepare();
    if (!e) e = R6::prepare();
    if (!e) e = I2::prepare();
    if (!e) e = I3::prepare();
    if (!e) e = I4::prepare();
    if (!e) e = I5::prepare();
    return e;
}

const Y355ConvRInfo *y355_convr_info(int rid) { return rid >= 0 && rid < NR ? &g_info[rid] : nullptr; }

// ring instantiation for a 3x3 / stride-1 layer with in_pb bytes per input pixel and cout_pad output channels (-1: none)
int y355_convr_select(int bf, int in_pb, int cout_pad, int pool, int H, int W) {
    if (pool && ((H | W) & 1)) return -1;
    for (int i = 0; i < NR; ++i)
        if (g_info[i].bf == (bf ? 1 : 0) && g_info[i].cinb == in_pb && g_info[i].pool == pool && cout_pad % g_info[i].bn == 0 &&
            H >= g_info[i].th && W >= g_info[i].tw) return i;
    return -1;
}

// p.w = y355_convg_pack order with (bn, wn, nt) of y355_convr_info(rid); false = not launched (the caller runs convg.hip)
bool y355_launch_convr(int rid, const ConvGParams &p, int device, hipStream_t s) {
    if (rid < 0 || rid >= NR || device < 0 || device >= 16 || !g_sink[device]) return false;
    const Y355ConvRInfo &ri = g_info[rid];
    if (p.taps != 9 || p.res || p.in_pb != ri.cinb) return false;
    if (ri.bf ? !p.bias_f : !p.bias_w) return false;
    if (((p.out_off | p.out_pb) & (ri.bf ? (p.out_f32 ? 15 : 7) : 3)) != 0) return false;
    switch (rid) {
    case 0: return R0::launch(p, device, s);
    case 1: return R1::launch(p, device, s);
    case 2: return R2::launch(p, device, s);
    case 3: return R3::launch(p, device, s);
    case 4: return R4::launch(p, device, s);
    case 5: return R5::launch(p, device, s);
    case 6: return I0::launch(p, device, s);
    case 7: return I1::launch(p, device, s);
    case 8: return R6::launch(p, device, s);
    case 9: return I2::launch(p, device, s);
    case 10: return I3::launch(p, device, s);
    case 11: return I4::launch(p, device, s);
    case 12: return I5::launch(p, device, s);
    default: return false;
    }
}
