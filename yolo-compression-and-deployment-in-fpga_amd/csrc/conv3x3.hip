// yolo355 -- fused int8 3x3/pad1 convolution for gfx950 (MI355X).
//
// Replaces, for one layer, the reference sequence
//   a_tracker.quantize_activation -> Conv2d_fuse (nn.Conv2d + LeakyReLU(0.125)) -> [MaxPool2d(2,2)]
// of models/slim_yolo_v2.py:220-319 (utils/modules.py:20-29) with ONE kernel working on
// integers: implicit-GEMM on v_mfma_i32_16x16x64_i8, bias + leaky + power-of-two
// requantisation (round-half-even) + 2x2 max-pool in the epilogue.
//
// Mapping (DESIGN.md "conv3x3 kernel"):
//   * one workgroup (4 waves) = TH x TW output pixels of one image x BN output channels;
//   * the (TH+2) x (TW+2) x CIN input patch is staged ONCE into LDS (pixel-major rows of
//     CIN+16 bytes: the 16-byte pad makes the per-tap ds_read_b128 conflict-free); the nine
//     taps are nine constant LDS offsets -> im2col never exists in memory;
//   * GEMM rows = pixels (for pooled layers ordered as 2x2 windows so the four accumulator
//     registers of a lane ARE one pooling window), GEMM columns = output channels;
//   * weights are pre-packed on the host in MFMA B-fragment order, one 1 KiB fragment per
//     (k-step, n-tile); each wave streams exactly the fragments it consumes straight from L2
//     into VGPRs (coalesced dwordx4, one k-step ahead) -- no LDS traffic and no barrier in
//     the K loop.
#include "y355_common.h"
#include <type_traits>

template <int CIN>
struct KGeom {
    static constexpr int KS = (CIN == 16) ? 3 : (CIN == 32) ? 5 : 9 * (CIN / 64);
    static constexpr int STRIDE = (CIN == 16) ? 16 : CIN + 16;
    static constexpr int CPP = CIN / 16;
};

template <int NT>
__device__ __forceinline__ void store_bytes(int8_t *dst, const int (&q)[NT]) {
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

template <int CIN, int BN, int TH, int TW, bool POOL, int WM, int WN, bool STATS, bool WIDE>
__global__ __launch_bounds__(256) void conv3x3_i8_kernel(const ConvParams p) {
    using G = KGeom<CIN>;
    constexpr int PW = TW + 2, PH = TH + 2, NPIX = PH * PW;
    constexpr int STRIDE = G::STRIDE, CPP = G::CPP, KS = G::KS;
    constexpr int BM = TH * TW;
    constexpr int MT_TOT = (BM + 15) / 16;
    constexpr int MT = (MT_TOT + WM - 1) / WM;
    constexpr int NT = BN / 16 / WN;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(!POOL || (TH % 2 == 0 && TW % 2 == 0), "pooled tiles are even");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    int bid = y355_xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid % p.nblk;
    bid /= p.nblk;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int H = p.H, W = p.W;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the input patch (all CIN channels) into LDS, 16 B per thread per step
    {
        const int8_t *inb = p.in + (size_t)b * (H + 2) * (W + 2) * CIN;
        constexpr int ITEMS = NPIX * CPP;
        constexpr int BATCH = 8;
        // the tail is clamped, not branched: duplicate items rewrite the same bytes
        for (int it0 = tid; it0 < ITEMS; it0 += 256 * BATCH) {
            v4i v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int it = min(it0 + u * 256, ITEMS - 1);
                const int pix = it / CPP, c = it % CPP;
                const int py = pix / PW, px = pix % PW;
                const int gy = min(y0 + py, H + 1), gx = min(x0 + px, W + 1);
                v[u] = *(const v4i *)(inb + ((size_t)gy * (W + 2) + gx) * CIN + c * 16);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int it = min(it0 + u * 256, ITEMS - 1);
                const int pix = it / CPP, c = it % CPP;
                *(v4i *)(smem + pix * STRIDE + c * 16) = v[u];
            }
        }
    }

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;

    // ---- per-lane LDS base of each m-tile row (the tile's output pixel, tap (0,0))
    int abase[MT];
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
        abase[m] = (oy * PW + ox) * STRIDE + (CIN >= 64 ? g * 16 : 0);
    }
    // per-lane k-step offsets for the thin layers (tap depends on the lane's k-group)
    int kofs[(CIN < 64) ? KS : 1];
    if constexpr (CIN == 16) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tap = min(4 * ks + g, 8);
            kofs[ks] = ((tap / 3) * PW + tap % 3) * STRIDE;
        }
    } else if constexpr (CIN == 32) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tap = min(2 * ks + (g >> 1), 8);
            kofs[ks] = ((tap / 3) * PW + tap % 3) * STRIDE + (g & 1) * 16;
        }
    } else {
        kofs[0] = 0;
    }

    v4i acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = (v4i){0, 0, 0, 0};

    const int8_t *wp = p.w + ((size_t)(nb * KS) * WN + wn) * NT * 1024 + lane * 16;
    constexpr size_t WSTEP = (size_t)WN * NT * 1024;
    v4i bcur[NT], bnext[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bcur[t] = *(const v4i *)(wp + t * 1024);

    __syncthreads();

    auto kstep = [&](int ks, int ko) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bnext[t] = *(const v4i *)(wp + (size_t)min(ks + 1, KS - 1) * WSTEP + t * 1024);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const v4i a = *(const v4i *)(smem + abase[m] + ko);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bcur[t], acc[m][t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) bcur[t] = bnext[t];
    };

    if constexpr (CIN < 64) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kstep(ks, kofs[ks]);
    } else {
        constexpr int NCH = CIN / 64;
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) kstep(ch * 9 + tap, ((tap / 3) * PW + tap % 3) * STRIDE + ch * 64);
        }
    }

    // ---- epilogue
    const Requant rq = p.rq;
    const int nlane = nb * BN + wn * (NT * 16) + li * NT;
    using T = typename std::conditional<WIDE, long long, int>::type;
    using U = typename UnsignedOf<T>::type;
    T bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (WIDE) bias[t] = p.bias_w[nlane + t];
        else bias[t] = p.bias_t[nlane + t];
    }

    U amax = 0;
    unsigned int nsat = 0, nguard = 0;
    const U gthr = (!p.guard || rq.guard_log2 >= (WIDE ? 63 : 31)) ? ~(U)0 : ((U)1 << rq.guard_log2);

    if constexpr (STATS) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (wm * MT + m) * 16 + 4 * g + r;
                int oy, ox;
                if constexpr (POOL) {
                    const int w = row >> 2;
                    oy = 2 * (w / (TW / 2)) + ((row & 3) >> 1);
                    ox = 2 * (w % (TW / 2)) + (row & 1);
                } else {
                    oy = row / TW;
                    ox = row % TW;
                }
                const bool valid = (row < BM) && (y0 + oy < H) && (x0 + ox < W);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const T tp = y355_pre<T>(acc[m][t][r], bias[t], rq);
                    amax = max(amax, valid ? y355_uabs<T>(tp) : (U)0);
                    if (p.raw && valid)
                        p.raw[(((size_t)b * H + y0 + oy) * W + x0 + ox) * p.cstride + nlane + t] = (long long)tp;
                }
            }
        }
        const unsigned long long wmax = y355_wave_max_u64((unsigned long long)amax);
        if (lane == 0) atomicMax(&p.ctr->absmax, wmax);
    } else if constexpr (POOL) {
        const int halo = p.out_halo;
        const int Ho = H >> 1, Wo = W >> 1;
        int8_t *outb = p.out + (size_t)b * (Ho + 2 * halo) * (Wo + 2 * halo) * p.cstride;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int w = (wm * MT + m) * 4 + g;
            const int wy = w / (TW / 2), wx = w % (TW / 2);
            const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
            const bool valid = (w * 4 < BM) && oy < Ho && ox < Wo;
            int q[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const v4i a = acc[m][t];
                const int vmax = max(max(a[0], a[1]), max(a[2], a[3]));
                const T tp = y355_pre<T>(vmax, bias[t], rq);
                const int vmin = min(min(a[0], a[1]), min(a[2], a[3]));
                const T tn = y355_pre<T>(vmin, bias[t], rq);
                nguard += (valid && max(y355_uabs<T>(tp), y355_uabs<T>(tn)) >= gthr) ? 1u : 0u;
                const T qq = y355_rne_shift<T>(tp, rq.sh);
                q[t] = y355_clamp8<T>(qq);
                nsat += (valid && (T)q[t] != qq) ? 1u : 0u;
            }
            if (valid)
                store_bytes<NT>(outb + ((size_t)(oy + halo) * (Wo + 2 * halo) + ox + halo) * p.cstride + nlane, q);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const int halo = p.out_halo;
        int8_t *outb = p.out + (size_t)b * (H + 2 * halo) * (W + 2 * halo) * p.cstride;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (wm * MT + m) * 16 + 4 * g + r;
                const int oy = row / TW, ox = row % TW;
                const int gy = y0 + oy, gx = x0 + ox;
                const bool valid = row < BM && gy < H && gx < W;
                int q[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const T tp = y355_pre<T>(acc[m][t][r], bias[t], rq);
                    nguard += (valid && y355_uabs<T>(tp) >= gthr) ? 1u : 0u;
                    const T qq = y355_rne_shift<T>(tp, rq.sh);
                    q[t] = y355_clamp8<T>(qq);
                    nsat += (valid && (T)q[t] != qq) ? 1u : 0u;
                }
                if (valid)
                    store_bytes<NT>(outb + ((size_t)(gy + halo) * (W + 2 * halo) + gx + halo) * p.cstride + nlane, q);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one m-tile's temporaries live at a time
        }
    }
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
    if (nguard) atomicAdd(&p.ctr->guard, (unsigned long long)nguard);
}

// ---------------------------------------------------------------------------------- host side
template <int CIN, int BN, int TH, int TW, bool POOL, int WM, int WN>
struct ConvInst {
    static constexpr size_t LDS = (size_t)(TH + 2) * (TW + 2) * KGeom<CIN>::STRIDE + 64;
    template <bool ST, bool WD>
    static void go(const ConvParams &p, int nblocks, hipStream_t s) {
        hipLaunchKernelGGL((conv3x3_i8_kernel<CIN, BN, TH, TW, POOL, WM, WN, ST, WD>), dim3(nblocks), dim3(256), LDS, s, p);
    }
    static void launch(const ConvParams &p, int nblocks, hipStream_t s) {
        if (p.rq.wide) { if (p.mode == 1) go<true, true>(p, nblocks, s); else go<false, true>(p, nblocks, s); }
        else { if (p.mode == 1) go<true, false>(p, nblocks, s); else go<false, false>(p, nblocks, s); }
    }
    template <bool ST, bool WD>
    static int prep1(void) {
        return (int)hipFuncSetAttribute((const void *)conv3x3_i8_kernel<CIN, BN, TH, TW, POOL, WM, WN, ST, WD>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    static int prepare(void) {
        int e = prep1<true, true>();
        if (!e) e = prep1<false, true>();
        if (!e) e = prep1<true, false>();
        if (!e) e = prep1<false, false>();
        return e;
    }
    static constexpr ConvKernelInfo info() {
        return ConvKernelInfo{CIN, BN, TH, TW, POOL ? 1 : 0, WM, WN, BN / 16 / WN, KGeom<CIN>::KS, LDS, &launch, &prepare};
    }
};

// conv3_1's tile (y355_common.h: the weight packing depends on WN)
static const ConvKernelInfo g_kernels[Y355_K_COUNT] = {
    //        CIN  BN  TH  TW  POOL  WM WN      tuned for 416x416
    ConvInst<16, 32, 16, 52, true, 4, 1>::info(),     // conv2    208x208
    ConvInst<32, 64, Y355_C31_TH, Y355_C31_TW, false, Y355_C31_GWM, Y355_C31_WN>::info(),    // conv3_1  104x104
    ConvInst<64, 64, 26, 26, true, 4, 1>::info(),     // conv3_2  104x104
    ConvInst<64, 128, 13, 26, false, 2, 2>::info(),   // conv4_1  52x52
    ConvInst<128, 64, 26, 26, true, 4, 1>::info(),    // conv4_2  52x52
    ConvInst<128, 128, 13, 26, false, 2, 2>::info(),  // conv5    26x26  (13x26 strip x 128 channels)
    ConvInst<256, 128, 13, 26, false, 2, 2>::info(),  // conv6/7  26x26
    ConvInst<256, 64, 13, 13, false, 4, 1>::info(),   // pred     26x26
    // generic small-tile variants (operator-level API, any shape)
    ConvInst<16, 64, 8, 16, false, 4, 1>::info(),
    ConvInst<32, 64, 8, 16, false, 4, 1>::info(),
    ConvInst<64, 64, 8, 16, false, 4, 1>::info(),
    ConvInst<128, 64, 8, 16, false, 4, 1>::info(),
    ConvInst<256, 64, 8, 16, false, 4, 1>::info(),
    ConvInst<16, 64, 8, 16, true, 4, 1>::info(),
    ConvInst<32, 64, 8, 16, true, 4, 1>::info(),
    ConvInst<64, 64, 8, 16, true, 4, 1>::info(),
    ConvInst<128, 64, 8, 16, true, 4, 1>::info(),
    ConvInst<256, 64, 8, 16, true, 4, 1>::info(),
};

const ConvKernelInfo *y355_conv_kernel(int id) {
    return (id >= 0 && id < Y355_K_COUNT) ? &g_kernels[id] : nullptr;
}

size_t y355_packed_bytes(const ConvKernelInfo &ki, int cout_pad) {
    return (size_t)(cout_pad / ki.bn) * ki.ks * ki.wn * ki.nt * 1024;
}

// B-fragment order: frag(nb, ks, wn, t) = ((nb*KS + ks)*WN + wn)*NT + t, 1 KiB each;
// inside a fragment lane l = (g = l>>4, j = l&15) holds 16 consecutive k of output channel
//   n = nb*BN + wn*NT*16 + j*NT + t      (so that a lane's NT outputs are adjacent bytes)
// with k -> (tap, cin) as the kernel's A side walks the patch (64-channel chunk major).
void y355_pack_weights(const ConvKernelInfo &ki, const int8_t *q_w, int cout, int cin, int cout_pad, int8_t *dst) {
    const int CIN = ki.cin, KS = ki.ks, NT = ki.nt, WN = ki.wn, BN = ki.bn;
    const int nblk = cout_pad / BN;
    for (int nb = 0; nb < nblk; ++nb)
        for (int ks = 0; ks < KS; ++ks)
            for (int wn = 0; wn < WN; ++wn)
                for (int t = 0; t < NT; ++t) {
                    int8_t *f = dst + ((((size_t)nb * KS + ks) * WN + wn) * NT + t) * 1024;
                    for (int l = 0; l < 64; ++l) {
                        const int g = l >> 4, j = l & 15;
                        const int n = nb * BN + wn * NT * 16 + j * NT + t;
                        for (int kk = 0; kk < 16; ++kk) {
                            int tap, ci;
                            if (CIN == 16) { tap = 4 * ks + g; ci = kk; }
                            else if (CIN == 32) { tap = 2 * ks + (g >> 1); ci = 16 * (g & 1) + kk; }
                            else { tap = ks % 9; ci = 64 * (ks / 9) + 16 * g + kk; }
                            int8_t v = 0;
                            if (tap < 9 && n < cout && ci < cin) v = q_w[((size_t)n * cin + ci) * 9 + tap];
                            f[l * 16 + kk] = v;
                        }
                    }
                }
}
