// yolo355 -- the thin 3x3 layers of the bf16 nets with the WEIGHTS IN REGISTERS (round 6): the bf16
// form of convpx.hip for SlimYOLOv2's conv3_1 (32 -> 64), conv3_2 + pool3 (64 -> 64) and conv4_1 (64 -> 128)
// (models/slim_yolo_v2.py:403-412, forward :561-575, BatchNorm folded on the host).
//
// convr.hip runs these layers on the deep layers' ring discipline (weights streamed through an LDS ring, a barrier per k-step):
// 52 / 55 / 39 us at B = 64 for 10 / 20 / 10 us of MFMA work -- their K is 9 x 32 .. 9 x 64 channels, too shallow for a k-loop to
// amortise a tile's start-up.  convpx.hip's construction carries over unchanged because v_mfma_f32_16x16x32_bf16 takes the same 16
// operand bytes per lane as the int8 MFMA: a bf16 pixel of C channels is an "int8 pixel" of 2 C bytes for the LDS-DMA row ring, the
// swizzles and every address; only the k-step holds 32 channels instead of 64.  So, as there:
//   * the weights are the MFMA's A operand (rows = output channels); a wave keeps ALL fragments of its block of 16 NTN output
//     channels in registers for the whole launch (36 fragments = 144 VGPRs in all three instantiations);
//   * the pixels are the B operand (16 consecutive pixels, or 16 consecutive 2x2 pooling windows whose 4x4 neighbourhood is read
//     once and fed to the window's four conv outputs), read from a rolling ring of whole padded input rows in LDS that LDS-DMA fills
//     one chunk ahead, behind a counted vmcnt;
//   * a lane ends up with 4 NTN consecutive channels of ONE pixel: 16- or 32-byte stores straight from the registers.
// What differs is the epilogue, which is the bf16 nets' (convg.hip / convr.hip): fp32 accumulator (the bias rides in as the MFMAs' C
// operand) -> [2x2 max] -> LeakyReLU(slope) -> bf16 (RNE) -- no saturation, no cold pass, no counters.
// Parity: tolerance against the fp32 oracle (tests/test_fp32_models.py); the accumulation order differs from convr's in where the bias
// enters, so the two routes are not bit-identical and are compared within 2 bf16 ulps (tests/test_fp32_models.py::test_convpxb_...).
#include "y355_common.h"
#include <cstring>
#include <type_traits>
#ifndef PXB_R31
#define PXB_R31 11               // groups per wave and chunk of the unpooled layers
#define PXB_R41 6
#endif

namespace {
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bglds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
__device__ __forceinline__ unsigned int pk2(float a, float b) {       // (a, b) -> two bf16 (RNE), a in the low half
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    const v2bf v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}

// PXB = bytes per input pixel (2 x channels): 64 or 128
template <int PXB, int NTN, int NCB, bool POOL>
struct PbGeom {
    static constexpr int PPP = 1024 / PXB;                       // pixels per 1 KiB DMA piece
    static constexpr int KPP = PXB / 64;                         // k-steps (32 channels) per tap
    static constexpr int KS = 9 * KPP;
    static constexpr int CPB = 16 * NTN;                         // output channels per block (per wave)
    static constexpr int COUT = CPB * NCB;
    static constexpr int NFRAG = NCB * KS * NTN;
    static_assert(PXB == 64 || PXB == 128, "input channels 32 or 64");
};
struct PbArgs {
    int total_groups, ngi, cg, pwl, logr, ppg;                   // as convpx.hip's PxArgs
};
struct PbChunk { int b, g0, g1, lo, hi; };

__device__ __forceinline__ void bwait_vmcnt(int n) {
#define BW_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        BW_CASE(0) BW_CASE(1) BW_CASE(2) BW_CASE(3) BW_CASE(4) BW_CASE(5) BW_CASE(6) BW_CASE(7) BW_CASE(8) BW_CASE(9)
        BW_CASE(10) BW_CASE(11) BW_CASE(12) BW_CASE(13) BW_CASE(14) BW_CASE(15) BW_CASE(16) BW_CASE(17) BW_CASE(18) BW_CASE(19)
        BW_CASE(20) BW_CASE(21) BW_CASE(22) BW_CASE(23) BW_CASE(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef BW_CASE
}
}  // namespace

template <int PXB, int NTN, int NCB, bool POOL, int NW>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void convpxb_kernel(const ConvGParams p, const PbArgs a) {
    using G = PbGeom<PXB, NTN, NCB, POOL>;
    constexpr int PPP = G::PPP, KPP = G::KPP, KS = G::KS, CPB = G::CPB;
    constexpr int NPS = NW / NCB;                        // pixel streams: waves that share a channel block
    constexpr int NV = POOL ? 4 : 1;                     // conv outputs per column (pooling window)
    constexpr int SPG = NTN / 2;                         // 16-byte stores per lane and group
    static_assert(NW % NCB == 0 && (NTN == 4 || NTN == 2), "geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // ring of 2^logr rows of PWL * PXB bytes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % NCB, ps = wave / NCB;
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;
    const int PW = W + 2, PWL = a.pwl;
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const int npw = Ho * Wo;
    const int rowb = PWL * PXB;

    // ---- weights of this wave's channel block: A fragments [k-step][n-tile], registers for the whole launch
    v4i wf[KS][NTN];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int n = 0; n < NTN; ++n) wf[ks][n] = *(const v4i *)(p.w + ((size_t)(cb * KS + ks) * NTN + n) * 1024 + lane * 16);
    // accumulator register r of n-tile n of lane group g = channel cb * CPB + 4 NTN g + 4 n + r; the bias is the MFMAs' C operand
    v4f cin[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) cin[n] = *(const v4f *)(p.bias_f + cb * CPB + 4 * NTN * g + 4 * n);
    float slope = p.slope;
    asm volatile("" : "+v"(slope));                      // a VGPR operand for the epilogue's multiply (front.hip)
    const float invWo = 1.0f / (float)Wo;

    // ---- this workgroup's share of the batch's groups, walked in chunks of <= cg groups that stay inside one image
    const int G_ = gridDim.x;
    const int gbeg = (int)((long long)a.total_groups * blockIdx.x / G_), gend = (int)((long long)a.total_groups * (blockIdx.x + 1) / G_);
    if (gbeg >= gend) return;
    constexpr int MUL = POOL ? 2 : 1;
    const int R = 1 << a.logr, RM = R - 1;
    auto chunk_at = [&](int gg) {
        PbChunk c;
        c.b = gg / a.ngi;
        c.g0 = gg - c.b * a.ngi;
        c.g1 = min(min(c.g0 + a.cg, a.ngi), c.g0 + (gend - gg));
        const int ya = (16 * c.g0) / Wo, yb = (min(16 * c.g1, npw) - 1) / Wo;
        c.lo = c.b * (H + 2) + MUL * ya;
        c.hi = c.b * (H + 2) + MUL * yb + (POOL ? 4 : 3);
        return c;
    };
    // LDS-DMA of whole padded rows into ring slots row & RM, 1 KiB pieces of PPP pixels (convpx.hip: same swizzles)
    constexpr int CPX = PXB / 16;
    const int dpx = lane / CPX, dch = lane % CPX;
    const int ppr = PWL / PPP;
    const int RS = ppr < NW ? NW / ppr : 1;
    const int CPW = ppr < NW ? 1 : (ppr + NW - 1) / NW;
    const int pc0 = ppr < NW ? wave % ppr : wave;
    const int rr0 = ppr < NW ? (wave / ppr < RS ? wave / ppr : (1 << 28)) : 0;
    auto lane_off = [&](int pc) {
        const int col = pc * PPP + dpx;
        int sch = dch;
        if constexpr (CPX == 4) sch ^= ((col >> 2) & 1) << 1;
        if constexpr (CPX == 8) sch ^= ((col >> 1) & 3) << 1;
        return min(col, PW - 1) * PXB + 16 * sch;
    };
    const int goff0 = lane_off(pc0);
    auto issue_pieces = [&](int r0, int nrows, int &rr, int count) {
        int done = 0;
        for (; rr < nrows && done < count; rr += RS) {
            const int row = r0 + rr;
            const char *src = p.in + (size_t)row * (size_t)(PW * PXB);
            char *dst = smem + (row & RM) * rowb;
            bglds16(src + goff0, dst + pc0 * 1024);
            ++done;
            for (int j = 1; j < CPW; ++j) {
                const int pc = pc0 + j * NW;
                if (pc < ppr) {
                    bglds16(src + lane_off(pc), dst + pc * 1024);
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

    PbChunk ch = chunk_at(gbeg);
    issue_rows(ch.lo, ch.hi);
    int loaded = ch.hi;
    int gg = gbeg, nstores = -1;                         // stores this wave issued behind its last DMA piece (-1: wait for everything)
    for (;;) {
        bwait_vmcnt(nstores);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (loaded < ch.hi) {                            // rows that could not be issued ahead (the ring was full: image boundaries)
            issue_rows(max(loaded, ch.lo), ch.hi);
            loaded = ch.hi;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const int gnext = gg + (ch.g1 - ch.g0);
        const bool more = gnext < gend;
        PbChunk nx = ch;
        if (more) nx = chunk_at(gnext);
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
        char *outb = p.out + (((size_t)ch.b * (Ho + 2) + 1) * (Wo + 2) + 1) * (size_t)p.out_pb + p.out_off + cb * CPB * 2;    // wave-uniform
        const int rbase = ch.b * (H + 2);

        auto locate = [&](int grp, int &oy, int &ox) {
            const int pc = min(grp * 16 + li, npw - 1);            // padding lanes of an image's last group repeat its last pixel
            oy = (int)(((float)pc + 0.5f) * invWo);                // pc / Wo (exact: pc < 2^16)
            ox = pc - oy * Wo;
        };
        auto mfma = [&](const v4i &wa, const v4i &b, const v4f &c) {
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wa), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
        };
        auto issue = [&](int grp, v4f (&acc)[NV][NTN]) {
            int oy, ox;
            locate(grp, oy, ox);
            const int ar = rbase + MUL * oy, x0 = MUL * ox;
            constexpr int NC = POOL ? 4 : 3;
            int xoff[NC], roff[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int x = x0 + c;
                int sch = g;
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
                if constexpr (KPP == 1) {                          // every read of the group, then its MFMAs (one exposed LDS wait per group)
                    v4i bq[9];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) bq[tap] = *(const v4i *)(smem + roff[tap / 3] + xoff[tap % 3]);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                        for (int n = 0; n < NTN; ++n) acc[0][n] = mfma(wf[tap][n], bq[tap], acc[0][n]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 9 * NTN, 0);
                } else {                                           // a filter row's reads, then its MFMAs (18 operands do not fit beside 144 VGPRs of weights)
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty) {
                        v4i bq[3][KPP];
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                            for (int h = 0; h < KPP; ++h) bq[tx][h] = *(const v4i *)(smem + roff[ty] + (xoff[tx] ^ (h << 6)));
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                            for (int h = 0; h < KPP; ++h)
#pragma unroll
                                for (int n = 0; n < NTN; ++n) acc[0][n] = mfma(wf[(ty * 3 + tx) * KPP + h][n], bq[tx][h], acc[0][n]);
                    }
                }
            } else {
                // neighbourhood row r feeds conv output (dy, dx) with filter tap (r - dy, c - dx)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v4i bq[4][KPP];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int h = 0; h < KPP; ++h) bq[c][h] = *(const v4i *)(smem + roff[r] + (xoff[c] ^ (h << 6)));
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
                                        acc[2 * dy + dx][n] = mfma(wf[(ty * 3 + tx) * KPP + h][n], bq[c][h], acc[2 * dy + dx][n]);
                            }
                    }
                }
            }
        };
        auto finish = [&](int grp, const v4f (&acc)[NV][NTN]) {
            int oy, ox;
            locate(grp, oy, ox);
            unsigned int word[NTN][2];
#pragma unroll
            for (int n = 0; n < NTN; ++n) {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[0][n][r];
                    if constexpr (POOL) t = fmaxf(fmaxf(t, acc[1][n][r]), fmaxf(acc[2][n][r], acc[3][n][r]));   // the max commutes with the monotone LeakyReLU
                    y[r] = fmaxf(t, t * slope);                    // LeakyReLU for 0 <= slope <= 1 (launcher); slope 1: identity
                }
                word[n][0] = pk2(y[0], y[1]);
                word[n][1] = pk2(y[2], y[3]);
            }
            // unconditional: the padding lanes rewrite the image's last pixel with the same bytes, and the number of stores a
            // wave has in flight stays a function of its group count (the counted wait above)
            char *dst = outb + (size_t)(oy * (Wo + 2) + ox) * (size_t)p.out_pb + 8 * NTN * g;
            *(v4i *)dst = (v4i){(int)word[0][0], (int)word[0][1], (int)word[1][0], (int)word[1][1]};
            if constexpr (NTN == 4) *(v4i *)(dst + 16) = (v4i){(int)word[2][0], (int)word[2][1], (int)word[3][0], (int)word[3][1]};
        };
        nstores = 0;
        {
            v4f acc[NV][NTN];
#pragma unroll 1
            for (int grp = ch.g0 + ps; grp < ch.g1; grp += NPS) {
                issue(grp, acc);
                prefetch(a.ppg);
                finish(grp, acc);
                nstores += SPG;
            }
            prefetch(1 << 30);
        }
        if (!more) break;
        gg = gnext;
        ch = nx;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
namespace {
__host__ inline unsigned short f2bf(float f) {                    // RNE, as the device's conversion
    unsigned int u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// A fragments: fragment ((cb * KS + ks) * NTN + n), lane (i = l & 15, g = l >> 4), 8 bf16:
//   row i = output channel cb * CPB + 4 NTN (i >> 2) + 4 n + (i & 3);  k = tap ks / KPP, input channels 32 (ks % KPP) + 8 g .. + 7
template <int PXB, int NTN, int NCB, bool POOL>
void pb_pack(const float *w, int cout, int cin, char *dst) {
    using G = PbGeom<PXB, NTN, NCB, POOL>;
    memset(dst, 0, (size_t)G::NFRAG * 1024);
    unsigned short *d = (unsigned short *)dst;
    for (int cb = 0; cb < NCB; ++cb)
        for (int ks = 0; ks < G::KS; ++ks)
            for (int n = 0; n < NTN; ++n)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4;
                    const int ch = cb * G::CPB + 4 * NTN * (i >> 2) + 4 * n + (i & 3);
                    const int tap = ks / G::KPP, c0 = 32 * (ks % G::KPP) + 8 * g;
                    if (ch >= cout) continue;
                    for (int kk = 0; kk < 8; ++kk)
                        if (c0 + kk < cin)
                            d[((((size_t)cb * G::KS + ks) * NTN + n) * 1024 + l * 16) / 2 + kk] = f2bf(w[((size_t)ch * cin + c0 + kk) * 9 + tap]);
                }
}

template <int PXB, int NTN, int NCB, bool POOL, int NW>
struct PbInst {
    using G = PbGeom<PXB, NTN, NCB, POOL>;
    static constexpr int NPS = NW / NCB;
    static PbArgs args(const ConvGParams &p, int rounds) {
        PbArgs a;
        const int Ho = POOL ? p.H / 2 : p.H, Wo = POOL ? p.W / 2 : p.W;
        a.ngi = (Ho * Wo + 15) / 16;
        a.total_groups = a.ngi * p.B;
        a.cg = rounds * NPS;
        a.pwl = (p.W + 2 + G::PPP - 1) / G::PPP * G::PPP;
        const int rows2 = (2 * 16 * a.cg + Wo - 1) / Wo + 1;
        const int need = (POOL ? 2 : 1) * rows2 + 2;
        a.logr = 2;
        while ((1 << a.logr) < need) ++a.logr;
        const int newrows = (POOL ? 2 : 1) * ((16 * a.cg + Wo - 1) / Wo + 1);
        const int ppr = a.pwl / G::PPP, rs = ppr < NW ? NW / ppr : 1, cpw = ppr < NW ? 1 : (ppr + NW - 1) / NW;
        const int per_wave = (newrows + rs - 1) / rs * cpw;
        a.ppg = (per_wave + rounds - 1) / rounds;
        return a;
    }
    static size_t lds_bytes(const PbArgs &a) { return ((size_t)a.pwl * PXB) << a.logr; }
    static bool launch(const ConvGParams &p, int rounds, hipStream_t s) {
        const int Wo = POOL ? p.W / 2 : p.W;
        if (p.in_pb != PXB || !p.out_halo || p.out_f32 || p.res || p.taps != 9 || !p.bias_f || p.W < 16 || Wo < 1) return false;
        if (POOL && ((p.H | p.W) & 1)) return false;
        if (!(p.slope >= 0.f && p.slope <= 1.f)) return false;                 // the epilogue takes max(t, slope * t)
        if (((p.out_pb | p.out_off) & 15) || p.out_pb < p.out_off + G::COUT * 2) return false;
        if ((long long)p.B * (p.H + 2) * (p.W + 2) * PXB >= (1ll << 31)) return false;      // 32-bit row arithmetic
        // the largest chunk (groups per pixel stream: `rounds` at most) whose ring of whole padded rows fits LDS -- a bf16 row is twice
        // an int8 row, so the shapes of this network take about half of convpx.hip's chunk; a map too wide for any chunk is declined
        PbArgs a = args(p, rounds);
        while (lds_bytes(a) > 160 * 1024 && rounds > 1) a = args(p, --rounds);
        if (lds_bytes(a) > 160 * 1024) return false;
        int grid = y355_cu_count();                                // one NW-wave workgroup per CU
        if (p.grid_limit > 0 && p.grid_limit < grid) grid = p.grid_limit;
        if (grid > a.total_groups) grid = a.total_groups;
        hipLaunchKernelGGL((convpxb_kernel<PXB, NTN, NCB, POOL, NW>), dim3(grid), dim3(NW * 64), lds_bytes(a), s, p, a);
        return true;
    }
    static int prepare() {
        return (int)hipFuncSetAttribute((const void *)convpxb_kernel<PXB, NTN, NCB, POOL, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
};
//                    PXB NTN NCB POOL  NW        SlimYOLOv2 on bf16
using PB_C3_1 = PbInst<64, 4, 1, false, 8>;       // 32 -> 64:           eight pixel streams
using PB_C3_2 = PbInst<128, 2, 2, true, 8>;       // 64 -> 64, pooled:   two 32-channel blocks x four window streams
using PB_C4_1 = PbInst<128, 2, 4, false, 8>;      // 64 -> 128:          four 32-channel blocks x two pixel streams
}  // namespace

int y355_prepare_convpxb(void) {
    int e = PB_C3_1::prepare();
    if (!e) e = PB_C3_2::prepare();
    if (!e) e = PB_C4_1::prepare();
    return e;
}

// id of the instantiation for a 3x3 / stride 1 layer with in_pb bytes per input pixel, cout output channels and a fused 2x2 pool
// (or none): -1 = none.  Packed weights: y355_convpxb_packed_bytes / y355_convpxb_pack (w fp32 [cout][cin][3][3]).
int y355_convpxb_select(int in_pb, int cout, int pool) {
    if (in_pb == 64 && cout == 64 && !pool) return 0;
    if (in_pb == 128 && cout == 64 && pool) return 1;
    if (in_pb == 128 && cout == 128 && !pool) return 2;
    return -1;
}
size_t y355_convpxb_packed_bytes(int id) {
    switch (id) {
    case 0: return (size_t)PB_C3_1::G::NFRAG * 1024;
    case 1: return (size_t)PB_C3_2::G::NFRAG * 1024;
    case 2: return (size_t)PB_C4_1::G::NFRAG * 1024;
    default: return 0;
    }
}
bool y355_convpxb_pack(int id, const float *w, int cout, int cin, char *dst) {
    switch (id) {
    case 0: if (cin > 32 || cout != 64) return false; pb_pack<64, 4, 1, false>(w, cout, cin, dst); return true;
    case 1: if (cin > 64 || cout != 64) return false; pb_pack<128, 2, 2, true>(w, cout, cin, dst); return true;
    case 2: if (cin > 64 || cout != 128) return false; pb_pack<128, 2, 4, false>(w, cout, cin, dst); return true;
    default: return false;
    }
}
// false = not available for this launch: the caller runs convr.hip / convg.hip.  p.w = y355_convpxb_pack layout.
bool y355_launch_convpxb(int id, const ConvGParams &p, hipStream_t s) {
    switch (id) {
    case 0: return PB_C3_1::launch(p, PXB_R31, s);
    case 1: return PB_C3_2::launch(p, 2, s);
    case 2: return PB_C4_1::launch(p, PXB_R41, s);
    default: return false;
    }
}
