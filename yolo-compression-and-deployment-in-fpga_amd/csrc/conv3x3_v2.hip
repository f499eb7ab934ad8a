// yolo355 -- fused int8 3x3 convolution, production variant (persistent + LDS-DMA rings).
//
// Same math, tile geometry, weight packing and epilogue as conv3x3.hip (which stays as the
// statistics / 64-bit-epilogue / generic-shape kernel); what changes is how bytes move:
//   * workgroups are persistent (grid = a few per CU) and walk tiles with stride gridDim.x;
//   * the input patch is cut into 64-channel chunks; a 2-slot LDS ring of chunks is filled by
//     LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip and no
//     staging VALU) one chunk ahead of the MFMAs -- across tile boundaries too;
//   * weights come through a 4-slot LDS ring of k-steps (one tap of one chunk, BN/16 1-KiB
//     B fragments), also by LDS-DMA, two k-steps ahead; for the two thin layers (K = 144 / 288)
//     the whole layer's fragments stay resident in LDS instead;
//   * one raw s_barrier per k-step behind a COUNTED s_waitcnt vmcnt(N) (never 0 inside a tile
//     except its first two steps, which also drain the previous tile's stores).
// LDS pixel rows are unpadded (16 / 32 / 64 bytes).  For 64-byte rows the four 16-byte chunks
// of pixel p are stored at chunk ^ ((p >> 1) & 3): with that XOR the ds_read_b128 of an MFMA
// A fragment (16 pixel rows x 4 k-groups) is bank-conflict free for contiguous pixels and for
// 2x2 pooling-window order (4 LDS cycles instead of 8).  The DMA applies the XOR on its SOURCE
// address, the MFMA side on its read address (LDS-DMA destinations are lane-linear).
#include "y355_common.h"
#include <cstdlib>

// -DY355_DIAG=1 builds the ablation switches (ConvParams.mode bits 8..) and the s_memtime
// stamps into the kernel; the production build has neither (they fragment the k-step into
// basic blocks and stop the scheduler from overlapping LDS reads with the MFMAs).
#ifndef Y355_DIAG
#define Y355_DIAG 0
#endif
#ifndef Y355_C32_WRES
#define Y355_C32_WRES 0          // 1: conv3_2 on the resident-weight kernel of this file instead of conv3x3_ring.hip (measured: 37.0 vs 35.0 us, not used)
#endif
#ifndef Y355_SLAB_AUX
#define Y355_SLAB_AUX 2
#endif

// LDS row pitch in pixels.  64-byte pixels: a multiple of 8, so the chunk XOR depends on the patch
// column only.  32- and 16-byte pixels: the pitch that makes the A-fragment ds_read_b128 conflict-free
// under gfx950's 4 x 16 lane grouping (scratch/bank_sim.py: 6 -> 4 and 8 -> 4.9 LDS cycles per read):
// pitch = 2 (mod 8) for 32-byte pixels, pitch = 8 (mod 16) for 16-byte pixels.
constexpr int y355_lds_pitch(int cc, int pw) {
    if (cc == 64) return (pw + 7) / 8 * 8;
    if (cc == 32) { int p = pw; while (p % 8 != 2) ++p; return p; }
    int p = pw; while (p % 16 != 8) ++p; return p;
}

template <int CIN>
struct KGeom2 {
    static constexpr int CC = CIN < 64 ? CIN : 64;                // channels per chunk
    static constexpr int NCH = CIN / CC;
    static constexpr int STRIDE = CC;                             // LDS bytes per pixel (no padding)
    static constexpr int SPC = (CC == 16) ? 3 : (CC == 32) ? 5 : 9;   // k-steps per chunk
    static constexpr int KS = NCH * SPC;
};

__device__ __forceinline__ void glds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
// non-temporal form (aux = 2) for the activation slabs: read once per tile, they should not push
// the layer's weights (re-read by every tile) out of the XCD's L2
__device__ __forceinline__ void glds16_nt(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, Y355_SLAB_AUX);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NT>
__device__ __forceinline__ void store_bytes2(int8_t *dst, const int (&q)[NT]) {
    if constexpr (NT == 1) {
        *dst = (int8_t)q[0];
    } else if constexpr (NT == 2) {
        *(unsigned short *)dst = (unsigned short)((q[0] & 0xff) | ((q[1] & 0xff) << 8));
    } else if constexpr (NT == 4) {
        *(unsigned int *)dst = (unsigned int)((q[0] & 0xff) | ((q[1] & 0xff) << 8) |
                                              ((q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24));
    } else {
        static_assert(NT == 8, "NT");
        uint2 v;
        v.x = (unsigned int)((q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24));
        v.y = (unsigned int)((q[4] & 0xff) | ((q[5] & 0xff) << 8) | ((q[6] & 0xff) << 16) | ((unsigned)(q[7] & 0xff) << 24));
        *(uint2 *)dst = v;
    }
}

template <int CIN, int BN, int TH, int TW, bool POOL, int WM, int WN, bool WRES, int MINB = 1>
__global__ __launch_bounds__(WM * WN * 64, MINB) void conv3x3_i8_v2_kernel(const ConvParams p, const int total_tiles) {
    constexpr int NW = WM * WN;                                    // waves per workgroup (4 or 8)
    constexpr int NTHR = NW * 64;
    using G = KGeom2<CIN>;
    constexpr int CC = G::CC, NCH = G::NCH, STRIDE = G::STRIDE, SPC = G::SPC, KS = G::KS;
    constexpr int PW = TW + 2, PH = TH + 2;
    // LDS row pitch in pixels: a multiple of 8 for 64-byte pixels, so that the chunk XOR depends
    // on the patch column only and every tap stays an immediate offset
    constexpr int PWL = y355_lds_pitch(CC, PW);
    constexpr int NPIX = PH * PWL;
    constexpr int BM = TH * TW;
    constexpr int MT_TOT = (BM + 15) / 16;
    constexpr int MT = (MT_TOT + WM - 1) / WM;
    constexpr int NT = BN / 16 / WN;
    constexpr int SLABB = (NPIX * STRIDE + 1023) / 1024 * 1024;   // bytes of one chunk slot
    constexpr int NPIECE = SLABB / 1024;
    constexpr int PPW = (NPIECE + NW - 1) / NW;                    // slab pieces per wave
    constexpr int WB = (BN / 16) * 1024;                           // weight bytes per k-step
    constexpr int NFR = BN / 16;                                   // fragments per k-step
    constexpr int WPW = (NFR + NW - 1) / NW;                       // weight pieces per wave per k-step
    constexpr int WSLOTS = WRES ? KS : 4;
    constexpr int OFF_W = 2 * SLABB;
    constexpr int OFF_DUMMY = OFF_W + WSLOTS * WB;                 // 1 KiB sink for padding pieces
    // int8 output tile, row-major.  Resident-weight kernels of 64-channel layers (two 57 KiB slab slots + the layer's
    // weights) have no room for a staging buffer of their own: they stage through the slab slot of the tile that just
    // finished (one extra barrier per tile: every wave's last A-fragment read must have returned first)
    constexpr bool STG_ALIAS = WRES && CC == 64;
    constexpr int OFF_STG = OFF_DUMMY + 1024;
    // staged rows (pixels / windows): POOL ? MT * WM * 4 : MT * WM * 16 -- sized by the launcher (ConvInst2::SROWS)
    constexpr int SSTR = BN + 16;                                  // bytes per staged row (+pad)
    constexpr int OROWS = POOL ? BM / 4 : BM;                      // real rows of a tile
    constexpr int NIT = (OROWS * (BN / 16) + NTHR - 1) / NTHR;     // output stores per thread per tile (static)
    constexpr int cap63 = 63;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(WRES ? NCH == 1 : true, "resident weights only for single-chunk layers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;

    // ---- tile-independent per-lane geometry
    int abase[MT][(CC == 64) ? 3 : 1];       // CC==64: one swizzled base per tap column dx
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int row = (wm * MT + m) * 16 + li;
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
        if constexpr (CC == 64) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                abase[m][dx] = (oy * PWL + ox + dx) * 64 + ((g ^ (((ox + dx) >> 1) & 3)) << 4);
        } else {
            abase[m][0] = (oy * PWL + ox) * STRIDE;
        }
    }
    int kofs[(CC < 64) ? SPC : 1];
    if constexpr (CC == 16) {
#pragma unroll
        for (int t = 0; t < SPC; ++t) {
            const int tap = min(4 * t + g, 8);
            kofs[t] = ((tap / 3) * PWL + tap % 3) * STRIDE;
        }
    } else if constexpr (CC == 32) {
#pragma unroll
        for (int t = 0; t < SPC; ++t) {
            const int tap = min(2 * t + (g >> 1), 8);
            kofs[t] = ((tap / 3) * PWL + tap % 3) * STRIDE + (g & 1) * 16;
        }
    } else {
        kofs[0] = 0;
    }
    // slab DMA pieces of this wave: piece q = wave + NW*j covers LDS bytes [q*1024, +1024);
    // lane l owns 16 of them: pixel = o / STRIDE, byte `within` of that pixel's chunk
    int ppix[PPW];      // (py << 16) | px, or -1 for pad / out-of-patch lanes
    int pwithin[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = wave + NW * j;
        const int o = q * 1024 + lane * 16;
        const int pix = o / STRIDE;
        int within = o - pix * STRIDE;
        if constexpr (CC == 64) within ^= ((pix >> 1) & 3) << 4;     // LDS slot s holds chunk s ^ swz(pix)
        const bool ok = (q < NPIECE) && (pix < NPIX);
        ppix[j] = ok ? (((pix / PWL) << 16) | min(pix % PWL, PW - 1)) : -1;    // pitch padding re-reads column PW-1
        pwithin[j] = within;
    }

    auto decode = [&](int tile, int &b, int &y0, int &x0, int &nb) {
        nb = tile % p.nblk;
        tile /= p.nblk;
        x0 = (tile % p.tiles_x) * TW;
        tile /= p.tiles_x;
        y0 = (tile % p.tiles_y) * TH;
        b = tile / p.tiles_y;
    };
    const int dbg = Y355_DIAG ? (p.mode >> 8) : 0;
    auto issue_slab = [&](int b, int y0, int x0, int c, int slot) {
        if (dbg & 2) return;
        const int8_t *inb = p.in + (size_t)b * (H + 2) * (W + 2) * CIN + c * CC;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int q = wave + NW * j;
            const int8_t *src = p.in;                         // dummy for pad lanes
            if (ppix[j] >= 0) {
                const int gy = min(y0 + (ppix[j] >> 16), H + 1), gx = min(x0 + (ppix[j] & 0xffff), W + 1);
                src = inb + ((size_t)gy * (W + 2) + gx) * CIN + pwithin[j];
            }
            char *dst = (q < NPIECE) ? smem + slot * SLABB + q * 1024 : smem + OFF_DUMMY;
            glds16_nt(src, dst);
        }
    };
    auto issue_w = [&](int nb, int ks, int slot) {
        if (dbg & 1) return;
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int f = wave + NW * j;                       // fragment of this k-step
            const bool ok = f < NFR;
            const int8_t *src = p.w + ((size_t)(nb * KS + ks) * NFR + (ok ? f : 0)) * 1024 + lane * 16;
            char *dst = ok ? smem + OFF_W + slot * WB + f * 1024 : smem + OFF_DUMMY;
            glds16(src, dst);
        }
    };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (Y355_DIAG) {
            if (p.stamps && tid == 0 && nstamp < 32) p.stamps[(size_t)blockIdx.x * 32 + nstamp++] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp();
    int b, y0, x0, nb;
    decode(tile, b, y0, x0, nb);
    int sl = 0;                                                // slab slot of the current chunk
    int wq = 0;                                                // weight ring slot of the current k-step
    // ---- prologue
    issue_slab(b, y0, x0, 0, 0);
    if constexpr (WRES) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) issue_w(nb, ks, ks);
    } else {
        issue_w(nb, 0, 0);
        issue_w(nb, 1, 1);
        issue_w(nb, 2, 2);
    }

    if (dbg & 8) { wait_vmcnt<0>(); return; }
    const Requant rq = p.rq;
    unsigned int nsat = 0;
    bool first = true;
    stamp();

    for (;;) {
        int ntile = tile + gridDim.x;
        const bool more = ntile < total_tiles;
        if (!more) ntile = tile;                               // keep the DMA counts static
        int b2, y2, x2, nb2;
        decode(ntile, b2, y2, x2, nb2);

        v4i acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (v4i){0, 0, 0, 0};

        // Ring kernels keep the B fragments of a k-step in registers ONE PHASE EARLY: step s multiplies
        // with the fragments it read (from LDS) during step s-1 and reads those of step s+1 while its
        // MFMAs run, so nothing waits for LDS right behind a barrier.  The barrier of step s therefore
        // publishes W(s+1), and the DMA runs three steps ahead (W(s+3) refills the slot read in step
        // s-2: two barriers after its last read, the LDS-DMA WAR rule).  bfb[s & 1] is static because
        // both loops are unrolled; a tile starts with a "pre" phase that publishes and reads W(0).
        v4i bfb[2][NT];
        v4i afp[2];                                              // A fragments 0, 1 of the next step (same chunk)
        if constexpr (!WRES) {
            if (first) wait_vmcnt<2 * WPW>();
            else wait_vmcnt<(2 * WPW + NIT < cap63 ? 2 * WPW + NIT : cap63)>();
            __builtin_amdgcn_s_barrier();
            const char *wb0 = smem + OFF_W + wq * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) bfb[0][tt] = *(const v4i *)(wb0 + tt * 1024);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool lastc = (c == NCH - 1);
            int aaddr[MT][(CC == 64) ? 3 : 1];                   // slot base + row base (bytes); taps are immediates
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int dx = 0; dx < ((CC == 64) ? 3 : 1); ++dx) aaddr[m][dx] = sl * SLABB + abase[m][dx];
#pragma unroll
            for (int t = 0; t < SPC; ++t) {
                const bool fine = Y355_DIAG && p.stamps && (p.mode >> 16) && c == 1;
                if (fine) stamp();
                // vmcnt is in-order: "at most N younger operations may still fly".  Ring kernels wait
                // for W(s+1); younger than it are W(s+2), a slab issued one step earlier, and -- at a
                // tile's first steps -- the NIT output stores of the previous tile (always issued, see
                // the copy-out), so finished tiles do not stall the ring.
                if constexpr (WRES) {
                    if (c == 0 && t == 0) {
                        if (first) wait_vmcnt<0>();
                        else wait_vmcnt<(NIT < cap63 ? NIT : cap63)>();
                    }
                } else {
                    if (c == 0 && t <= 1 && !first) wait_vmcnt<(NIT + WPW < cap63 ? NIT + WPW : cap63)>();
                    else if (t == 2) wait_vmcnt<WPW + PPW>();   // W(s+2) and the slab issued at t=1 may fly
                    else wait_vmcnt<WPW>();
                }
                if (fine) stamp();
                if (!WRES || t == 0) __builtin_amdgcn_s_barrier();
                if (Y355_DIAG && (fine || (c == 0 && t == 0 && !(p.mode >> 16)))) stamp();
                // -- refill the rings.  WAR rule for LDS-DMA (cdna_hip_programming.md, "Read a staged
                // buffer one phase AFTER..."): a slot may be restaged TWO barriers after its last
                // ds_read was issued (every wave has then executed a whole phase, whose own lgkmcnt
                // waits retire the older reads), or one barrier after when an lgkmcnt(0) sat in front of
                // it.  The next slab is issued at t = 1 (its slot was last read at t = 8 of the previous
                // chunk).  Resident-weight kernels have one barrier per tile, right after the epilogue's
                // lgkmcnt(0) + barrier, so their slab goes out at t = 0.
                if (t == (WRES ? 0 : 1)) {
                    if (!lastc) issue_slab(b, y0, x0, c + 1, sl ^ 1);
                    else issue_slab(b2, y2, x2, 0, sl ^ 1);
                }
                constexpr int dummy_guard = 0;
                (void)dummy_guard;
                const int s_idx = c * SPC + t;                  // k-step of the tile (compile-time: both loops unrolled)
                if constexpr (!WRES) {
                    const int ks3 = s_idx + 3;
                    if (ks3 < KS) issue_w(nb, ks3, (wq + 3) & 3);
                    else issue_w(nb2, ks3 - KS, (wq + 3) & 3);
                }
                int ko;
                if constexpr (CC < 64) ko = kofs[t];
                else ko = (t / 3) * PWL * 64;                   // row offset of the tap; its column picks the base
                const int acol = (CC == 64) ? (t % 3) : 0;
                v4i bres[NT];                                    // resident-weight kernels read B in the step
                if constexpr (WRES) {
                    const char *wb = smem + OFF_W + t * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) bres[tt] = *(const v4i *)(wb + tt * 1024);
                }
                const int cur = s_idx & 1;
                // A fragments 0, 1: prefetched by the previous step of the chunk, else read now
                v4i af[MT];
                if (WRES || t == 0) {
                    af[0] = *(const v4i *)(smem + aaddr[0][acol] + ko);
                    if constexpr (MT > 1) af[1] = *(const v4i *)(smem + aaddr[1][acol] + ko);
                } else {
                    af[0] = afp[0];
                    if constexpr (MT > 1) af[1] = afp[1];
                }
                if constexpr (!WRES) {
                    // B fragments of step s+1 (published by this step's barrier), under this step's MFMAs
                    if (s_idx + 1 < KS) {
                        const char *wbn = smem + OFF_W + ((wq + 1) & 3) * WB + (wn * NT) * 1024 + lane * 16;
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) bfb[cur ^ 1][tt] = *(const v4i *)(wbn + tt * 1024);
                    }
                    wq = (wq + 1) & 3;
                }
                if (!(dbg & 4)) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (m + 2 < MT) af[m + 2] = *(const v4i *)(smem + aaddr[m + 2][acol] + ko);
                    if constexpr (!WRES) {
                        if (m == MT - 1 && t + 1 < SPC) {        // next step's first A fragments (same slab)
                            const int ko2 = (CC < 64) ? kofs[t + 1 < SPC ? t + 1 : t] : ((t + 1) / 3) * PWL * 64;
                            const int acol2 = (CC == 64) ? ((t + 1) % 3) : 0;
                            afp[0] = *(const v4i *)(smem + aaddr[0][acol2] + ko2);
                            if constexpr (MT > 1) afp[1] = *(const v4i *)(smem + aaddr[1][acol2] + ko2);
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        if constexpr (WRES)
                            acc[m][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[m], bres[tt], acc[m][tt], 0, 0, 0);
                        else
                            acc[m][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[m], bfb[cur][tt], acc[m][tt], 0, 0, 0);
                    }
                    if constexpr (WRES) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read ...
                        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // ... then this row's MFMAs
                    }
                }
                }
            }
            sl ^= 1;
        }

        stamp();
        // ---- epilogue of this tile: integer pipeline of conv3x3.hip (32-bit path) -> int8 tile in
        //      LDS (each lane packs its NT adjacent channels) -> barrier -> 16-byte coalesced stores
        if (!(dbg & 16)) {
            const int ncol = wn * (NT * 16) + li * NT;           // first channel of this lane in the tile
            int bias[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bias[t] = p.bias_t[nb * BN + ncol + t];
            char *stg = STG_ALIAS ? smem + (sl ^ 1) * SLABB : smem + OFF_STG;       // `sl` already points at the next tile's slot
            if constexpr (STG_ALIAS) {
                static_assert(!STG_ALIAS || (POOL ? MT * WM * 4 : MT * WM * 16) * SSTR <= SLABB, "the staged tile fits a slab slot");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // sh_l (a left requant shift; sh_r = 0 then) folded into the accumulator shift and the bias;
            // saturation is detected with one v_xad per output and counted exactly only when it happened
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
            unsigned int satx = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (POOL) {
                    const int srow = (wm * MT + m) * 4 + g;      // pooling window of the tile
                    int q[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const v4i a = acc[m][t];
                        const int vmax = max(max(a[0], a[1]), max(a[2], a[3]));
                        const int qq = requant(vmax, t);
                        q[t] = y355_clamp8<int>(qq);
                        satx += (unsigned int)(q[t] ^ qq);
                    }
                    store_bytes2<NT>((int8_t *)stg + srow * SSTR + ncol, q);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int srow = (wm * MT + m) * 16 + 4 * g + r;
                        int q[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int qq = requant(acc[m][t][r], t);
                            q[t] = y355_clamp8<int>(qq);
                            satx += (unsigned int)(q[t] ^ qq);
                        }
                        store_bytes2<NT>((int8_t *)stg + srow * SSTR + ncol, q);
                    }
                }
            }
            if (satx) {                                          // cold: exact count over the tile's real rows
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < (POOL ? 1 : 4); ++r) {
                        const int srow = POOL ? (wm * MT + m) * 4 + g : (wm * MT + m) * 16 + 4 * g + r;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            int v;
                            if constexpr (POOL) {
                                const v4i a = acc[m][t];
                                v = max(max(a[0], a[1]), max(a[2], a[3]));
                            } else {
                                v = acc[m][t][r];
                            }
                            const int qq = requant(v, t);
                            nsat += (srow < OROWS && y355_clamp8<int>(qq) != qq) ? 1u : 0u;
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp();
            __builtin_amdgcn_s_barrier();
            // copy-out: item = (row, 16-byte channel group); rows outside the image are dropped
            const int halo = p.out_halo;
            constexpr int CG = BN / 16;
            constexpr int OTW = POOL ? TW / 2 : TW;
            const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
            const int oy0 = POOL ? (y0 >> 1) : y0, ox0 = POOL ? (x0 >> 1) : x0;
            int8_t *outb = p.out + (size_t)b * (Ho + 2 * halo) * (Wo + 2 * halo) * p.cstride + nb * BN;
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int it = min(tid + j * NTHR, OROWS * CG - 1);   // tail clamps: duplicates rewrite the same bytes
                const int row = it / CG, cg = it % CG;
                const int oy = oy0 + row / OTW, ox = ox0 + row % OTW;
                const v4i v = *(const v4i *)(stg + row * SSTR + cg * 16);
                int8_t *dst = outb + ((size_t)(oy + halo) * (Wo + 2 * halo) + ox + halo) * p.cstride + cg * 16;
                if (!(oy < Ho && ox < Wo)) dst = p.sink + tid * 16;  // outside the image: a scratch line
                *(v4i *)dst = v;
            }
        }
        first = false;
        stamp();
        if (!more) break;
        tile = ntile;
        b = b2; y0 = y2; x0 = x2; nb = nb2;
    }
    wait_vmcnt<0>();       // retire the padding DMAs before the wave ends
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
}

// ------------------------------------------------------------------------------------------
template <int CIN, int BN, int TH, int TW, bool POOL, int WM, int WN, bool WRES, int MINB = 1>
struct ConvInst2 {
    using G = KGeom2<CIN>;
    static constexpr int PWL = y355_lds_pitch(G::CC, TW + 2);
    static constexpr int SLABB = ((TH + 2) * PWL * G::STRIDE + 1023) / 1024 * 1024;
    static constexpr int WB = (BN / 16) * 1024;
    static constexpr int MTT = (TH * TW + 15) / 16;
    static constexpr int SROWS = POOL ? ((MTT + WM - 1) / WM) * WM * 4 : ((MTT + WM - 1) / WM) * WM * 16;
    static constexpr bool STG_ALIAS = WRES && G::CC == 64;
    static constexpr size_t LDS = 2 * (size_t)SLABB + (size_t)(WRES ? G::KS : 4) * WB + 1024 + (STG_ALIAS ? 0 : (size_t)SROWS * (BN + 16));
    static size_t lds_launch() {
#ifdef Y355_EXPERIMENTS
        static const bool solo = getenv("Y355_V2_SOLO") != nullptr;       // experiment: one workgroup per CU
        return (solo && LDS < 84 * 1024) ? 84 * 1024 : LDS;
#else
        return LDS;
#endif
    }
    static int prepare() {
        return (int)hipFuncSetAttribute((const void *)conv3x3_i8_v2_kernel<CIN, BN, TH, TW, POOL, WM, WN, WRES, MINB>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch());
    }
    static bool launch(const ConvParams &p_in, hipStream_t s) {
        if (WRES && p_in.nblk != 1) return false;  // resident weights = one n-block
        ConvParams p = p_in;
        p.ev_start = p.ev_stop = nullptr;
        const int total = p.tiles_x * p.tiles_y * p.nblk * p.B;
        // persistent grid = what is actually co-resident: 8-wave workgroups take > 128 VGPRs per wave, so
        // only one fits a CU whatever its LDS (a 512-workgroup grid would run as two sequential rounds)
        int per_cu = (int)((160 * 1024) / LDS);
        per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
        if (WM * WN == 8 && MINB < 4) per_cu = 1;          // (MINB = waves per SIMD the kernel was compiled for: 4 = two 8-wave workgroups per CU)
        int grid = 256 * per_cu;
        if (grid > total) grid = total;
        Y355_LAUNCH((conv3x3_i8_v2_kernel<CIN, BN, TH, TW, POOL, WM, WN, WRES, MINB>), dim3(grid), dim3(WM * WN * 64), lds_launch(), s,
                    p_in.ev_start, p_in.ev_stop, p, total);
        return true;
    }
};

// must mirror the tile table of conv3x3.hip (same packing: BN, WN and NT are shared)
using V2_CONV2 = ConvInst2<16, 32, 16, 52, true, 4, 1, true>;
#ifndef Y355_C31_OCC
#define Y355_C31_OCC 1
#endif
using V2_CONV3_1 = ConvInst2<32, 64, Y355_C31_TH, Y355_C31_TW, false, Y355_C31_WM, Y355_C31_WN, true, Y355_C31_OCC>;
#if Y355_C32_WRES
using V2_CONV3_2 = ConvInst2<64, 64, 26, 26, true, 8, 1, true>;      // weights (36 KiB) resident: no per-step barrier, no weight re-streaming
#else
using V2_CONV3_2 = ConvInst2<64, 64, 26, 26, true, 8, 1, false>;
#endif
using V2_CONV4_1 = ConvInst2<64, 128, 13, 26, false, 4, 2, false>;
using V2_CONV4_2 = ConvInst2<128, 64, 26, 26, true, 8, 1, false>;
// conv5..7 at batch 64: 256 work items either way, but a 13x26 strip x 128 channels fetches 36 % fewer
// bytes per CU than a 13x13 tile x 256 channels (the weight stream is what these layers wait for)
using V2_CONV5 = ConvInst2<128, 128, 13, 26, false, 4, 2, false>;
using V2_CONV67 = ConvInst2<256, 128, 13, 26, false, 4, 2, false>;
using V2_PRED = ConvInst2<256, 64, 13, 13, false, 8, 1, false>;

// The production library uses this file for the two thin layers only (conv2, conv3_1: resident weights); the deep layers
// run conv3x3_ring.hip.  Their instantiations here are the fallbacks of the experiment builds (Y355_NO_RING_MASK); two of
// them spill registers, which the counted waits do not allow (check_kernels.py), so they are not even compiled by default.
int y355_prepare_conv_v2(void) {
    int e = V2_CONV2::prepare();
    if (!e) e = V2_CONV3_1::prepare();
#if Y355_C32_WRES && !defined(Y355_EXPERIMENTS)
    if (!e) e = V2_CONV3_2::prepare();
#endif
#ifdef Y355_EXPERIMENTS
    if (!e) e = V2_CONV3_2::prepare();
    if (!e) e = V2_CONV4_1::prepare();
    if (!e) e = V2_CONV4_2::prepare();
    if (!e) e = V2_CONV5::prepare();
    if (!e) e = V2_CONV67::prepare();
    if (!e) e = V2_PRED::prepare();
#endif
    return e;
}

bool y355_conv_v2_preferred(int kid) {
#if Y355_C32_WRES
    if (kid == Y355_K_CONV3_2) return true;
#endif
    return kid == Y355_K_CONV2 || kid == Y355_K_CONV3_1;
}

bool y355_launch_conv_v2(int kid, const ConvParams &p, hipStream_t s) {
    if ((p.mode & 0xff) != 0 || p.rq.wide || p.guard) return false;   // those go to conv3x3.hip
    switch (kid) {
    case Y355_K_CONV2: return V2_CONV2::launch(p, s);
    case Y355_K_CONV3_1: return V2_CONV3_1::launch(p, s);
#if Y355_C32_WRES && !defined(Y355_EXPERIMENTS)
    case Y355_K_CONV3_2: return V2_CONV3_2::launch(p, s);
#endif
#ifdef Y355_EXPERIMENTS
    case Y355_K_CONV3_2: return V2_CONV3_2::launch(p, s);
    case Y355_K_CONV4_1: return V2_CONV4_1::launch(p, s);
    case Y355_K_CONV4_2: return V2_CONV4_2::launch(p, s);
    case Y355_K_CONV5: return V2_CONV5::launch(p, s);
    case Y355_K_CONV67: return V2_CONV67::launch(p, s);
    case Y355_K_PRED: return V2_PRED::launch(p, s);
#endif
    default: return false;
    }
}
