// yolo355 -- generic chunked-K implicit-GEMM convolution (3x3/pad 1 or 1x1; stride 1, or 2 for 3x3) for gfx950,
// in two arithmetic types:
//   bf16 : BN-folded fp32 models run as bf16 x bf16 -> fp32 on v_mfma_f32_16x16x32_bf16
//          (SlimYOLOv2.forward, models/slim_yolo_v2.py:549-622; utils.modules.Conv2d :6-18)
//   int8 : the integer pipeline of conv3x3.hip for layers it cannot hold (more than 256 input
//          channels, 1x1 kernels, LeakyReLU slopes that are not a power of two:
//          models/tiny_yolo_v3.py:9-273, backbone/darknet.py:211-255)
//
// Same mapping as conv3x3.hip -- GEMM rows = output pixels of a TH x TW tile (2x2 pooling windows
// in adjacent rows), columns = output channels, nine taps = nine constant LDS offsets -- with the
// input patch staged in CHUNKS of CHB bytes per pixel, so the LDS slab is independent of the
// number of input channels.  A k-step is always 64 bytes of one pixel: 32 bf16 or 64 int8
// channels; byte-wise the A/B fragments of the two MFMA shapes are identical (lane (g, j)
// holds 16 bytes of row/column j at k-offset 16 g), which is why one kernel serves both.
#include "y355_common.h"
#include <cstring>
#include <type_traits>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int NT>
__device__ __forceinline__ void store_bf16(char *dst, const float (&v)[NT]) {
    unsigned short h[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) h[t] = __builtin_bit_cast(unsigned short, (__bf16)v[t]);
    if constexpr (NT == 1) {
        *(unsigned short *)dst = h[0];
    } else if constexpr (NT == 2) {
        *(unsigned int *)dst = (unsigned int)h[0] | ((unsigned int)h[1] << 16);
    } else if constexpr (NT == 4) {
        uint2 u;
        u.x = (unsigned int)h[0] | ((unsigned int)h[1] << 16);
        u.y = (unsigned int)h[2] | ((unsigned int)h[3] << 16);
        *(uint2 *)dst = u;
    } else {
        static_assert(NT == 8, "NT");
        uint4 u;
        u.x = (unsigned int)h[0] | ((unsigned int)h[1] << 16);
        u.y = (unsigned int)h[2] | ((unsigned int)h[3] << 16);
        u.z = (unsigned int)h[4] | ((unsigned int)h[5] << 16);
        u.w = (unsigned int)h[6] | ((unsigned int)h[7] << 16);
        *(uint4 *)dst = u;
    }
}

template <int NT>
__device__ __forceinline__ void store_i8(char *dst, const int (&q)[NT]) {
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {
        if constexpr (NT >= 4) {
            *(unsigned int *)(dst + t0) = (unsigned int)((q[t0] & 0xff) | ((q[t0 + 1] & 0xff) << 8) |
                                                         ((q[t0 + 2] & 0xff) << 16) | ((unsigned)(q[t0 + 3] & 0xff) << 24));
        }
    }
    if constexpr (NT == 2) *(unsigned short *)dst = (unsigned short)((q[0] & 0xff) | ((q[1] & 0xff) << 8));
    if constexpr (NT == 1) *dst = (char)q[0];
}

// integer epilogue with a general LeakyReLU slope neg_mul / 2^lk (DESIGN.md "requantisation"):
//   t = acc * 2^shl + bias;  t' = t >= 0 ? t * 2^lk : t * neg_mul;  q = clamp(RNE(t' * 2^-sh))
__device__ __forceinline__ long long requant_g(int acc, long long bias, const RequantG &rq) {
    long long t = (long long)acc * (1ll << rq.shl) + bias;
    t = t >= 0 ? t * (1ll << rq.lk) : t * (long long)rq.neg_mul;
    return y355_rne_shift<long long>(t, rq.sh);
}

// ---- four waves per workgroup (round 1): one tile per workgroup, stage -> barrier -> k-steps -> barrier per chunk.  Kept for
// the thin layers and the small / stride-2 tiles, where its 2-3 workgroups per CU overlap each other's phases.
template <bool BF, int CHB, int BN, int TH, int TW, bool POOL, int WM, int WN, int S, bool NARROW = false>
__global__ __launch_bounds__(256) void convg_kernel(const ConvGParams p) {
    constexpr bool THIN = (CHB == 32);           // 32 B per pixel: a k-step covers two taps
    // input patch of a TH x TW output tile: S*(T-1)+3 pixels a side (stride S, 3x3, pad 1)
    constexpr int PW = S * (TW - 1) + 3, PH = S * (TH - 1) + 3, NPIX = PH * PW;
    static_assert(S == 1 || (S == 2 && !POOL && !THIN), "stride 2: plain 64-byte-chunk tiles only");
    constexpr int STRIDE = CHB + 16;             // 16-byte pad: conflict-free ds_read_b128 across pixels
    constexpr int CPP = CHB / 16;
    constexpr int SUB = THIN ? 1 : CHB / 64;     // k-steps per tap and chunk
    constexpr int BM = TH * TW;
    constexpr int MT_TOT = (BM + 15) / 16;
    constexpr int MT = (MT_TOT + WM - 1) / WM;
    constexpr int NT = BN / 16 / WN;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(!POOL || (TH % 2 == 0 && TW % 2 == 0), "pooled tiles are even");
    using ACC = typename std::conditional<BF, v4f, v4i>::type;
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
    const int taps = p.taps;
    const int kpc = THIN ? 5 : SUB * taps;       // k-steps per chunk
    const int KS = kpc * p.nchunks;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;

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
        abase[m] = (S * oy * PW + S * ox) * STRIDE + (THIN ? 0 : g * 16);
    }
    int kofs[THIN ? 5 : 1];
    if constexpr (THIN) {
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = min(2 * ks + (g >> 1), 8);
            kofs[ks] = ((tap / 3) * PW + tap % 3) * STRIDE + (g & 1) * 16;
        }
    } else {
        kofs[0] = 0;
    }

    ACC acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (BF) acc[m][t] = (v4f){0.f, 0.f, 0.f, 0.f};
            else acc[m][t] = (v4i){0, 0, 0, 0};
        }

    const char *wp = p.w + ((size_t)(nb * KS) * WN + wn) * NT * 1024 + lane * 16;
    constexpr size_t WSTEP = (size_t)WN * NT * 1024;
    v4i bcur[NT], bnext[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bcur[t] = *(const v4i *)(wp + t * 1024);
    int ksg = 0;

    auto kstep = [&](int ko) {
        const int nx = min(ksg + 1, KS - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) bnext[t] = *(const v4i *)(wp + (size_t)nx * WSTEP + t * 1024);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const v4i a = *(const v4i *)(smem + abase[m] + ko);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (BF)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, bcur[t]),
                                                                        acc[m][t], 0, 0, 0);
                else
                    acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bcur[t], acc[m][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) bcur[t] = bnext[t];
        ++ksg;
    };

    const char *inb = p.in + (size_t)b * (H + 2) * (W + 2) * p.in_pb;
    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch) __syncthreads();
        // ---- stage CHB bytes of every patch pixel, 16 B per thread per step (tail clamped)
        {
            constexpr int ITEMS = NPIX * CPP;
            constexpr int BATCH = 8;
            const char *src0 = inb + ch * CHB;
            for (int it0 = tid; it0 < ITEMS; it0 += 256 * BATCH) {
                v4i v[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int it = min(it0 + u * 256, ITEMS - 1);
                    const int pix = it / CPP, c = it % CPP;
                    const int py = pix / PW, px = pix % PW;
                    const int gy = min(S * y0 + py, H + 1), gx = min(S * x0 + px, W + 1);
                    v[u] = *(const v4i *)(src0 + ((size_t)gy * (W + 2) + gx) * p.in_pb + c * 16);
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int it = min(it0 + u * 256, ITEMS - 1);
                    const int pix = it / CPP, c = it % CPP;
                    *(v4i *)(smem + pix * STRIDE + c * 16) = v[u];
                }
            }
        }
        __syncthreads();
        if constexpr (THIN) {
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) kstep(kofs[ks]);
        } else {
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                if (taps == 9) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) kstep(((tap / 3) * PW + tap % 3) * STRIDE + sub * 64);
                } else {
                    kstep((PW + 1) * STRIDE + sub * 64);
                }
            }
        }
    }

    // ---- epilogue
    const int nlane = nb * BN + wn * (NT * 16) + li * NT;       // first of this lane's NT channels
    const int halo = p.out_halo;
    const int Ho = POOL ? (H >> 1) : (S == 2 ? (H + 1) >> 1 : H), Wo = POOL ? (W >> 1) : (S == 2 ? (W + 1) >> 1 : W);
    char *outb = p.out + (size_t)b * (Ho + 2 * halo) * (Wo + 2 * halo) * p.out_pb + p.out_off;
    const char *resb = p.res ? p.res + (size_t)b * (Ho + 2) * (Wo + 2) * p.res_pb + p.res_off : nullptr;
    unsigned int nsat = 0;

    float biasf[NT];
    long long biasw[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (BF) { biasf[t] = p.bias_f[nlane + t]; biasw[t] = 0; }
        else { biasw[t] = p.bias_w[nlane + t]; biasf[t] = 0.f; }
    }
    int biasn[NT];                                         // NARROW: the biases fit 32 bits too
#pragma unroll
    for (int t = 0; t < NT; ++t) biasn[t] = (int)biasw[t];
    const float slope = p.slope;
    const RequantG rq = p.rq;
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
            if (resb && valid) {                        // the residual is a bf16 activation: exact in fp32
                const unsigned short *r = (const unsigned short *)(resb + ((size_t)(oy + 1) * (Wo + 2) + ox + 1) * p.res_pb) + nlane;
#pragma unroll
                for (int t = 0; t < NT; ++t) y[t] += __uint_as_float((unsigned int)r[t] << 16);
            }
            if (valid) {
                if (p.out_f32) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) ((float *)dst)[nlane + t] = y[t];
                } else {
                    store_bf16<NT>(dst + (size_t)nlane * 2, y);
                }
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
                    const long long qq = requant_g(vi[t], biasw[t], rq);
                    q[t] = y355_clamp8<long long>(qq);
                    nsat += (valid && (long long)q[t] != qq) ? 1u : 0u;
                }
            }
            if (valid) store_i8<NT>(dst + nlane, q);
        }
    };

#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if constexpr (POOL) {
            // monotone epilogue: pool the raw accumulators first
            const int w = (wm * MT + m) * 4 + g;
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
                const int row = (wm * MT + m) * 16 + 4 * g + r;
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
    if constexpr (!BF) {
        if (nsat && p.ctr) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
    }
}


// ---- eight waves per workgroup (round 2)
// One tile per workgroup, eight waves (96 accumulators per wave instead of 176: two waves per SIMD).  The input patch is staged
// in chunks of CHB bytes per pixel through TWO LDS slabs: the global loads of chunk c + 1 are issued before the k-steps of
// chunk c and written to the other slab after them, one barrier per chunk.  B fragments come straight from global memory
// (L2-resident), two k-steps ahead.  Measured (B = 64, slim fp32 / B = 128, tiny int8): 59.9 -> 63.8 k and 50.8 -> 63.1 k img/s.
// Persistent workgroups with the next tile's first chunk prefetched across the epilogue were built too and were SLOWER
// (58.2 k / 49.2 k): the extra live state spills, and one 8-wave workgroup per CU has nothing to overlap its epilogue with.
// What bounds this kernel now is the B path: 16-32 KB of fragments per k-step per CU through the vector-memory pipe
// (profiles/r02_notes.md); the int8 ring kernels avoid exactly that with LDS-DMA weight rings.
template <bool BF, int CHB, int BN, int TH, int TW, bool POOL, int WM, int WN, int S, bool NARROW = false>
__global__ __launch_bounds__(WM * WN * 64) void convg8_kernel(const ConvGParams p, const int total) {
    constexpr int NTHR = WM * WN * 64;
    constexpr bool THIN = (CHB == 32);           // 32 B per pixel: a k-step covers two taps
    // input patch of a TH x TW output tile: S*(T-1)+3 pixels a side (stride S, 3x3, pad 1)
    constexpr int PW = S * (TW - 1) + 3, PH = S * (TH - 1) + 3, NPIX = PH * PW;
    static_assert(S == 1 || (S == 2 && !POOL && !THIN), "stride 2: plain 64-byte-chunk tiles only");
    constexpr int STRIDE = CHB + 16;             // 16-byte pad: conflict-free ds_read_b128 across pixels
    constexpr int CPP = CHB / 16;
    constexpr int SUB = THIN ? 1 : CHB / 64;     // k-steps per tap and chunk
    constexpr int BM = TH * TW;
    constexpr int MT_TOT = (BM + 15) / 16;
    constexpr int MT = (MT_TOT + WM - 1) / WM;
    constexpr int NT = BN / 16 / WN;
    static_assert(WM * WN == 8, "8 waves");
    constexpr int SLAB = (NPIX * STRIDE + 15) / 16 * 16;
    static_assert(!POOL || (TH % 2 == 0 && TW % 2 == 0), "pooled tiles are even");
    using ACC = typename std::conditional<BF, v4f, v4i>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int H = p.H, W = p.W;
    const int taps = p.taps;
    const int kpc = THIN ? 5 : SUB * taps;       // k-steps per chunk
    const int KS = kpc * p.nchunks;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, g = lane >> 4;

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
        abase[m] = (S * oy * PW + S * ox) * STRIDE + (THIN ? 0 : g * 16);
    }
    int kofs[THIN ? 5 : 1];
    if constexpr (THIN) {
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = min(2 * ks + (g >> 1), 8);
            kofs[ks] = ((tap / 3) * PW + tap % 3) * STRIDE + (g & 1) * 16;
        }
    } else {
        kofs[0] = 0;
    }

    int bid = y355_xcd_remap(blockIdx.x, gridDim.x);
    if (bid >= total) return;
    const int nb = bid % p.nblk;
    bid /= p.nblk;
    const int x0 = (bid % p.tiles_x) * TW;
    bid /= p.tiles_x;
    const int y0 = (bid % p.tiles_y) * TH;
    const int b = bid / p.tiles_y;

    // ---- staging: CHB bytes of every patch pixel, 16 B per thread per item (tail clamped: duplicates rewrite the same
    // bytes).  The item -> address arithmetic is recomputed per chunk on purpose: hoisted out of the loops it costs three
    // registers per item for the whole kernel.
    constexpr int ITEMS = NPIX * CPP;
    constexpr int NIT = (ITEMS + NTHR - 1) / NTHR;
    v4i stg[NIT];
    auto stage_load = [&](int bb, int yy, int xx, int ch) {
        const char *src0 = p.in + (size_t)bb * (H + 2) * (W + 2) * p.in_pb + ch * CHB;
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = min(tl + u * NTHR, ITEMS - 1);
            const int pix = it / CPP, c = it % CPP;
            const int py = pix / PW, px = pix % PW;
            const int gy = min(S * yy + py, H + 1), gx = min(S * xx + px, W + 1);
            stg[u] = *(const v4i *)(src0 + ((size_t)gy * (W + 2) + gx) * p.in_pb + c * 16);
        }
    };
    auto stage_store = [&](char *dst) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = min(tl + u * NTHR, ITEMS - 1);
            const int pix = it / CPP, c = it % CPP;
            *(v4i *)(dst + pix * STRIDE + c * 16) = stg[u];
        }
    };
    constexpr size_t WSTEP = (size_t)WN * NT * 1024;
    constexpr bool DEEP = true;                  // B fragments two k-steps ahead
    v4i b0[NT], b1[NT];
    auto load_b01 = [&](const char *wp) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            b0[t] = *(const v4i *)(wp + t * 1024);
            if constexpr (DEEP) b1[t] = *(const v4i *)(wp + (size_t)min(1, KS - 1) * WSTEP + t * 1024);
            else b1[t] = b0[t];
        }
    };

    stage_load(b, y0, x0, 0);
    const char *wp = p.w + ((size_t)(nb * KS) * WN + wn) * NT * 1024 + lane * 16;
    load_b01(wp);
    stage_store(smem);
    __syncthreads();
    int cur = 0;                                 // slab holding the chunk about to be consumed
    unsigned int nsat = 0;
    const float slope = p.slope;
    const RequantG rq = p.rq;
    const int halo = p.out_halo;
    const int Ho = POOL ? (H >> 1) : (S == 2 ? (H + 1) >> 1 : H), Wo = POOL ? (W >> 1) : (S == 2 ? (W + 1) >> 1 : W);

    {
        ACC acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (BF) acc[m][t] = (v4f){0.f, 0.f, 0.f, 0.f};
                else acc[m][t] = (v4i){0, 0, 0, 0};
            }
        int ksg = 0;
        const char *slab = smem;
        auto kstep = [&](int ko) {
            const int nx = min(ksg + (DEEP ? 2 : 1), KS - 1);
            v4i bn[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = *(const v4i *)(wp + (size_t)nx * WSTEP + t * 1024);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const v4i a = *(const v4i *)(slab + abase[m] + ko);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (BF)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b0[t]),
                                                                            acc[m][t], 0, 0, 0);
                    else
                        acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b0[t], acc[m][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (DEEP) { b0[t] = b1[t]; b1[t] = bn[t]; }
                else b0[t] = bn[t];
            }
            ++ksg;
        };

        for (int ch = 0; ch < p.nchunks; ++ch) {
            const bool more = ch + 1 < p.nchunks;
            slab = smem + cur * SLAB;
            if (more) stage_load(b, y0, x0, ch + 1);       // in flight under this chunk's k-steps
            if constexpr (THIN) {
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) kstep(kofs[ks]);
            } else {
#pragma unroll 1
                for (int sub = 0; sub < SUB; ++sub) {
                    if (taps == 9) {
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) kstep(((tap / 3) * PW + tap % 3) * STRIDE + sub * 64);
                    } else {
                        kstep((PW + 1) * STRIDE + sub * 64);
                    }
                }
            }
            if (more) {
                stage_store(smem + (cur ^ 1) * SLAB);
                __syncthreads();
                cur ^= 1;
            }
        }
        // ---- epilogue
        const int nlane = nb * BN + wn * (NT * 16) + li * NT;       // first of this lane's NT channels
        char *outb = p.out + (size_t)b * (Ho + 2 * halo) * (Wo + 2 * halo) * p.out_pb + p.out_off;
        const char *resb = p.res ? p.res + (size_t)b * (Ho + 2) * (Wo + 2) * p.res_pb + p.res_off : nullptr;

        float biasf[NT];
        long long biasw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (BF) { biasf[t] = p.bias_f[nlane + t]; biasw[t] = 0; }
            else { biasw[t] = p.bias_w[nlane + t]; biasf[t] = 0.f; }
        }
        int biasn[NT];                                     // NARROW: the biases fit 32 bits too
#pragma unroll
        for (int t = 0; t < NT; ++t) biasn[t] = (int)biasw[t];
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
                if (resb && valid) {                        // the residual is a bf16 activation: exact in fp32
                    const unsigned short *r = (const unsigned short *)(resb + ((size_t)(oy + 1) * (Wo + 2) + ox + 1) * p.res_pb) + nlane;
#pragma unroll
                    for (int t = 0; t < NT; ++t) y[t] += __uint_as_float((unsigned int)r[t] << 16);
                }
                if (valid) {
                    if (p.out_f32) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) ((float *)dst)[nlane + t] = y[t];
                    } else {
                        store_bf16<NT>(dst + (size_t)nlane * 2, y);
                    }
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
                        const long long qq = requant_g(vi[t], biasw[t], rq);
                        q[t] = y355_clamp8<long long>(qq);
                        nsat += (valid && (long long)q[t] != qq) ? 1u : 0u;
                    }
                }
                if (valid) store_i8<NT>(dst + nlane, q);
            }
        };

#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if constexpr (POOL) {
                // monotone epilogue: pool the raw accumulators first
                const int w = (wm * MT + m) * 4 + g;
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
                    const int row = (wm * MT + m) * 16 + 4 * g + r;
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
    if constexpr (!BF) {
        if (nsat && p.ctr) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
    }
}

// ------------------------------------------------------------------------------------------
template <bool BF, int CHB, int BN, int TH, int TW, bool POOL, int WM, int WN, int S = 1>
struct ConvGInst {
    static constexpr size_t SLAB = ((size_t)(S * (TH - 1) + 3) * (S * (TW - 1) + 3) * (CHB + 16) + 15) / 16 * 16;
    static constexpr bool EIGHT = (WM * WN == 8);
    static constexpr size_t LDS = EIGHT ? 2 * SLAB : SLAB;
    static void launch(const ConvGParams &p, int nblocks, hipStream_t s) {
        if constexpr (EIGHT) {
            // one tile per workgroup (persistent workgroups measured slower: slim fp32, B = 64, 58.2 k vs 63.8 k img/s)
            if (!BF && p.rq.narrow)
                hipLaunchKernelGGL((convg8_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S, !BF>), dim3(nblocks), dim3(512),
                                   p.nchunks > 1 ? 2 * SLAB : SLAB, s, p, nblocks);
            else
                hipLaunchKernelGGL((convg8_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S>), dim3(nblocks), dim3(512),
                                   p.nchunks > 1 ? 2 * SLAB : SLAB, s, p, nblocks);
        } else {
            if (!BF && p.rq.narrow)
                hipLaunchKernelGGL((convg_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S, !BF>), dim3(nblocks), dim3(256), SLAB, s, p);
            else
                hipLaunchKernelGGL((convg_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S>), dim3(nblocks), dim3(256), SLAB, s, p);
        }
    }
    static int prepare() {
        const void *fn;
        if constexpr (EIGHT) {
            fn = (const void *)convg8_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S>;
            if constexpr (!BF) {
                if (int e = (int)hipFuncSetAttribute((const void *)convg8_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S, !BF>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS)) return e;
            }
        } else {
            fn = (const void *)convg_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S>;
            if constexpr (!BF) {
                if (int e = (int)hipFuncSetAttribute((const void *)convg_kernel<BF, CHB, BN, TH, TW, POOL, WM, WN, S, !BF>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS)) return e;
            }
        }
        return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    static constexpr ConvGInfo info() {
        return ConvGInfo{BF ? 1 : 0, CHB, BN, TH, TW, POOL ? 1 : 0, WM, WN, BN / 16 / WN, S, LDS, &launch, &prepare};
    }
};

#define CONVG_SET(BF)                                                                                   \
    ConvGInst<BF, 32, 32, 16, 52, true, 4, 1>::info(),     /* 0 thin, pooled     (conv2)            */ \
    ConvGInst<BF, 32, 64, 13, 26, false, 2, 2>::info(),    /* 1 thin                                */ \
    ConvGInst<BF, 64, 64, 13, 26, false, 4, 2>::info(),    /* 2 64 B chunks      (conv3_1)          */ \
    ConvGInst<BF, 64, 64, 26, 26, true, 8, 1>::info(),     /* 3 64 B, pooled     (conv3_2, conv4_2) */ \
    ConvGInst<BF, 128, 128, 13, 26, false, 4, 2>::info(),  /* 4 128 B chunks     (conv4_1)          */ \
    ConvGInst<BF, 256, 256, 13, 13, false, 2, 4>::info(),  /* 5 256 B chunks     (conv5..7)         */ \
    ConvGInst<BF, 256, 64, 13, 13, false, 4, 1>::info(),   /* 6 256 B, few couts (pred)             */ \
    ConvGInst<BF, 64, 64, 8, 16, false, 4, 1>::info(),     /* 7 small tiles, any shape              */ \
    ConvGInst<BF, 64, 64, 8, 16, true, 4, 1>::info(),      /* 8 small tiles, pooled                 */ \
    ConvGInst<BF, 64, 64, 8, 16, false, 4, 1, 2>::info(),  /* 9 stride 2 (darknet53 down-sampling)  */ \
    ConvGInst<BF, 128, 64, 13, 13, false, 4, 2>::info()    /* 10 13x13 maps, 64 couts per workgroup: fills the chip at small batches */

static const ConvGInfo g_convg[2][Y355_G_COUNT] = {{CONVG_SET(false)}, {CONVG_SET(true)}};

const ConvGInfo *y355_convg_kernel(int bf, int id) {
    return (id >= 0 && id < Y355_G_COUNT) ? &g_convg[bf ? 1 : 0][id] : nullptr;
}

int y355_prepare_convg(void) {
    for (int bf = 0; bf < 2; ++bf)
        for (int i = 0; i < Y355_G_COUNT; ++i)
            if (int e = g_convg[bf][i].prepare()) return e;
    return 0;
}

// Pick the instantiation for a layer: `in_pb` bytes per input pixel (multiple of 32), real output
// channels `cout`, pooled or not, on an H x W map.
int y355_convg_select(int in_pb, int cout, int pool, int H, int W, int stride, int batch_hint) {
    if (stride == 2) return (in_pb % 64 == 0 && !pool) ? 9 : -1;
    const bool small = (H < 13 || W < 13);
    if (in_pb == 32) return pool ? 0 : 1;
    if (pool) return small ? 8 : 3;
    if (small) return 7;
    if (cout <= 64) return (in_pb % 256 == 0) ? 6 : 2;
    if (cout <= 128) return (in_pb % 128 == 0) ? 4 : 2;
    if (in_pb % 256 == 0) {
        // 256 output channels per workgroup on 13x13 tiles: too few workgroups for 256 CUs on small maps at small batches
        const long wgs = (long)((H + 12) / 13) * ((W + 12) / 13) * ((cout + 255) / 256) * (batch_hint > 0 ? batch_hint : 1 << 20);
        return wgs >= 192 ? 5 : 10;
    }
    return (in_pb % 128 == 0) ? 4 : 2;
}

int y355_convg_ksteps(const ConvGInfo &ki, int in_pb, int taps) {
    return ki.chb == 32 ? 5 : (in_pb / 64) * taps;
}

size_t y355_convg_packed_bytes(const ConvGInfo &ki, int in_pb, int taps, int cout_pad) {
    return (size_t)(cout_pad / ki.bn) * y355_convg_ksteps(ki, in_pb, taps) * ki.wn * ki.nt * 1024;
}

// B-fragment order as conv3x3.hip (y355_pack_weights): frag(nb, ks, wn, t), lane (g, j) holds the 16
// bytes at k-offset 16 g of output channel n = nb*BN + wn*NT*16 + j*NT + t.  k-steps run
// chunk-major: ks = (chunk * SUB + sub) * taps + tap covers input bytes [64*(chunk*SUB+sub), +64).
// `w` is [cout][cin][k][k] (k = 1 or 3), fp32 (bf16 nets: rounded to nearest-even here) or int8.
void y355_convg_pack(const ConvGInfo &ki, const float *w_f, const int8_t *w_q, int cout, int cin, int ksize,
                     int in_pb, int cout_pad, char *dst) {
    const int taps = ksize * ksize;
    const int es = ki.bf ? 2 : 1, epg = 16 / es;          // element size, elements per lane
    const int KS = y355_convg_ksteps(ki, in_pb, taps), NT = ki.nt, WN = ki.wn, BN = ki.bn;
    const int nblk = cout_pad / BN;
    for (int nb = 0; nb < nblk; ++nb)
        for (int ks = 0; ks < KS; ++ks)
            for (int wn = 0; wn < WN; ++wn)
                for (int t = 0; t < NT; ++t) {
                    char *f = dst + ((((size_t)nb * KS + ks) * WN + wn) * NT + t) * 1024;
                    for (int l = 0; l < 64; ++l) {
                        const int g = l >> 4, j = l & 15;
                        const int n = nb * BN + wn * NT * 16 + j * NT + t;
                        for (int e = 0; e < epg; ++e) {
                            int tap, ci;
                            if (ki.chb == 32) { tap = 2 * ks + (g >> 1); ci = (g & 1) * epg + e; }
                            else { tap = ks % taps; ci = (ks / taps) * (64 / es) + g * epg + e; }
                            const bool ok = tap < taps && n < cout && ci < cin;
                            const size_t wi = ((size_t)n * cin + ci) * taps + tap;
                            if (ki.bf) {
                                const float v = ok ? w_f[wi] : 0.f;
                                unsigned int u;
                                memcpy(&u, &v, 4);
                                u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;      // RNE (weights are finite)
                                const unsigned short h = (unsigned short)u;
                                memcpy(f + l * 16 + e * 2, &h, 2);
                            } else {
                                f[l * 16 + e] = ok ? (char)w_q[wi] : 0;
                            }
                        }
                    }
                }
}

// ------------------------------------------------------------------------------------------
// First layer of the bf16 nets: fp32 NCHW -> bf16 -> conv3x3(3 -> 16) + bias + LeakyReLU + 2x2 max
// pool -> bf16 NHWC16 with halo (SlimYOLOv2.conv1 + pool1, models/slim_yolo_v2.py:551-552).
// LDS patch of 8-byte pixels (r, g, b, 0); K = 3 filter rows x 2 pixels x 4 per MFMA, two MFMAs
// (pixel columns 0-1, then column 2) per 16 pixels x 16 channels.
template <int TW>
__global__ __launch_bounds__(256) void conv1_bf16_kernel(const Conv1FParams p) {
    constexpr int TH = 16;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int NW = (TH / 2) * (TW / 2);
    constexpr int MT_TOT = TH * TW / 16;
    __shared__ __attribute__((aligned(16))) uint2 patch[PH * PW + 8];
    __shared__ __attribute__((aligned(16))) unsigned short otile[NW * 16];

    const int tid = threadIdx.x;
    int bid = y355_xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int H = p.H, W = p.W;
    const int y0 = ty * TH, x0 = tx * TW;
    const float *xb = p.x + (size_t)b * 3 * H * W;
    const size_t plane = (size_t)H * W;
    for (int it0 = tid; it0 < PH * PW; it0 += 256 * 4) {
        float v[4][3];
        bool inside[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = min(it0 + u * 256, PH * PW - 1);
            const int py = it / PW, px = it % PW;
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            inside[u] = (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W);
            const size_t o = (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = xb[c * plane + o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = min(it0 + u * 256, PH * PW - 1);
            unsigned short h[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) h[c] = inside[u] ? __builtin_bit_cast(unsigned short, (__bf16)v[u][c]) : (unsigned short)0;
            uint2 w;
            w.x = (unsigned int)h[0] | ((unsigned int)h[1] << 16);
            w.y = (unsigned int)h[2];
            patch[it] = w;
        }
    }
    if (tid < 8) patch[PH * PW + tid] = make_uint2(0u, 0u);
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const v4i bw0 = *(const v4i *)(p.w + lane * 16);
    const v4i bw1 = *(const v4i *)(p.w + 1024 + lane * 16);
    const float bias = p.bias[li];
    const float slope = p.slope;
    const int Ho = H >> 1, Wo = W >> 1;
    for (int mt = wave; mt < MT_TOT; mt += 4) {
        const int row = mt * 16 + li;
        const int w = row >> 2, r = row & 3;
        const int oy = 2 * (w / (TW / 2)) + (r >> 1);
        const int ox = 2 * (w % (TW / 2)) + (r & 1);
        const uint2 *src = patch + (oy + min(g, 2)) * PW + ox;
        const uint2 s0 = src[0], s1 = src[1], s2 = src[2], s3 = src[3];
        const v4i a0 = {(int)s0.x, (int)s0.y, (int)s1.x, (int)s1.y};
        const v4i a1 = {(int)s2.x, (int)s2.y, (int)s3.x, (int)s3.y};
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a0), __builtin_bit_cast(v8bf, bw0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a1), __builtin_bit_cast(v8bf, bw1), acc, 0, 0, 0);
        const int wo = mt * 4 + g;
        const float x = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])) + bias;
        const float y = x >= 0.f ? x : x * slope;
        otile[wo * 16 + li] = __builtin_bit_cast(unsigned short, (__bf16)y);
    }
    __syncthreads();
    char *outb = p.out + (size_t)b * (Ho + 2) * (Wo + 2) * 32;
    for (int it = tid; it < NW * 2; it += 256) {
        const int w = it >> 1, hf = it & 1;
        const int wy = w / (TW / 2), wx = w % (TW / 2);
        const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
        if (oy < Ho && ox < Wo)
            *(v4i *)(outb + ((size_t)(oy + 1) * (Wo + 2) + ox + 1) * 32 + hf * 16) = *(const v4i *)((const char *)otile + w * 32 + hf * 16);
    }
}

static int conv1f_tw(int W) { return (W % 104 == 0) ? 104 : 32; }

void y355_conv1f_tiles(int H, int W, int *tx, int *ty) {
    const int tw = conv1f_tw(W);
    *tx = (W + tw - 1) / tw;
    *ty = (H + 15) / 16;
}

void y355_launch_conv1f(const Conv1FParams &p, hipStream_t s) {
    const int n = p.tiles_x * p.tiles_y * p.B;
    if (conv1f_tw(p.W) == 104) hipLaunchKernelGGL((conv1_bf16_kernel<104>), dim3(n), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv1_bf16_kernel<32>), dim3(n), dim3(256), 0, s, p);
}

// two B fragments: lane (g = filter row, j = cout) holds k = 4 d' + c, d' = 0,1 -> pixel column
// d = 2 f + d' of fragment f (column 3 and colour 3 are zero)
void y355_pack_conv1f(const float *w /*[16][3][3][3]*/, char *dst /*2048*/) {
    for (int f = 0; f < 2; ++f)
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, j = l & 15;
            for (int e = 0; e < 8; ++e) {
                const int d = 2 * f + (e >> 2), c = e & 3;
                float v = 0.f;
                if (g < 3 && d < 3 && c < 3) v = w[((j * 3 + c) * 3 + g) * 3 + d];
                unsigned int u;
                memcpy(&u, &v, 4);
                u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
                const unsigned short h = (unsigned short)u;
                memcpy(dst + f * 1024 + l * 16 + e * 2, &h, 2);
            }
        }
}
