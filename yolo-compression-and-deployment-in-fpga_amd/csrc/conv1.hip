// yolo355 -- first layer: fp32 NCHW input -> int8 quantise -> conv3x3(3->16) -> bias ->
// LeakyReLU(0.125) -> requantise -> 2x2 max-pool -> int8 NHWC16 (with zero halo), one kernel.
//
// Replaces models/slim_yolo_v2.py:218-231 (a_tracker_in.quantize_activation, conv1,
// a_tracker1.quantize_activation, pool1) and the FPGA driver's first_conv /
// pixel_norm_quantize (c_embedding/yolo_forward.c:57-85, 269-418).
//
// The layer is HBM-bound (2.08 MB of fp32 per image in, 0.69 MB out, 74.8 MMAC): the fp32
// planes are read coalesced along W, quantised once into an LDS patch of 4-byte pixels
// (r,g,b,0) and the 27-deep dot products run on the matrix cores: one v_mfma_i32_16x16x64_i8
// per 16 pixels x 16 channels with K = 3 filter rows x (4 pixels x 4 bytes), the unused K
// slots multiplied by zero weights.  GEMM rows are ordered as 2x2 pooling windows so the
// pool is a max over the lane's four accumulators BEFORE the (monotone) epilogue.
#include "y355_common.h"
#include <type_traits>

template <int TW, bool WIDE>
__global__ __launch_bounds__(256) void conv1_kernel(const Conv1Params p) {
    using T = typename std::conditional<WIDE, long long, int>::type;
    using U = typename UnsignedOf<T>::type;
    constexpr int TH = 16;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int NW = (TH / 2) * (TW / 2);        // pooling windows per tile
    constexpr int MT_TOT = TH * TW / 16;
    __shared__ __attribute__((aligned(16))) unsigned int patch[PH * PW + 8];
    __shared__ __attribute__((aligned(16))) unsigned char otile[NW * 16];

    const int tid = threadIdx.x;
    int bid = y355_xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int H = p.H, W = p.W;
    const int y0 = ty * TH, x0 = tx * TW;
    const float sc = p.in_scale;
    unsigned int nsat_in = 0;

    // ---- quantise the (TH+2)x(TW+2) input patch: q = clamp(rne(x * 2^sa0))  (:33-35)
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
            const int py = it / PW, px = it % PW;
            const bool own = inside[u] && (it0 + u * 256 < PH * PW) && py >= 1 && py <= TH && px >= 1 && px <= TW;
            unsigned int w = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r = rintf(v[u][c] * sc);
                const float rc = fminf(fmaxf(r, -127.f), 127.f);
                nsat_in += (own && rc != r) ? 1u : 0u;
                const int q = inside[u] ? (int)rc : 0;
                w |= (unsigned int)(q & 0xff) << (8 * c);
            }
            patch[it] = w;
        }
    }
    if (tid < 8) patch[PH * PW + tid] = 0;
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const v4i bw = *(const v4i *)(p.w + lane * 16);
    T bias;
    if constexpr (WIDE) bias = p.bias_w[li];
    else bias = p.bias_t[li];
    const Requant rq = p.rq;
    const U gthr = (!p.guard || rq.guard_log2 >= (WIDE ? 63 : 31)) ? ~(U)0 : ((U)1 << rq.guard_log2);
    const int Ho = H >> 1, Wo = W >> 1;
    U amax = 0;
    unsigned int nsat = 0, nguard = 0;

    for (int mt = wave; mt < MT_TOT; mt += 4) {
        const int row = mt * 16 + li;
        const int w = row >> 2, r = row & 3;
        const int oy = 2 * (w / (TW / 2)) + (r >> 1);
        const int ox = 2 * (w % (TW / 2)) + (r & 1);
        const unsigned int *src = patch + (oy + min(g, 2)) * PW + ox;
        v4i a;
        a[0] = (int)src[0];
        a[1] = (int)src[1];
        a[2] = (int)src[2];
        a[3] = (int)src[3];
        v4i acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bw, acc, 0, 0, 0);
        // lane (g, li) now holds the four positions of window mt*4+g for channel li
        const int wo = mt * 4 + g;
        const int wy = wo / (TW / 2), wx = wo % (TW / 2);
        const bool valid = ((y0 >> 1) + wy < Ho) && ((x0 >> 1) + wx < Wo);
        const int vmax = max(max(acc[0], acc[1]), max(acc[2], acc[3]));
        const int vmin = min(min(acc[0], acc[1]), min(acc[2], acc[3]));
        const T tp = y355_pre<T>(vmax, bias, rq);
        const T tn = y355_pre<T>(vmin, bias, rq);
        const U am = max(y355_uabs<T>(tp), y355_uabs<T>(tn));
        amax = max(amax, valid ? am : (U)0);
        nguard += (valid && am >= gthr) ? 1u : 0u;
        const T qq = y355_rne_shift<T>(tp, rq.sh);
        const int q = y355_clamp8<T>(qq);
        nsat += (valid && (T)q != qq) ? 1u : 0u;
        otile[wo * 16 + li] = (unsigned char)(q & 0xff);
    }

    if (p.mode == 1) {
        const unsigned long long wmax = y355_wave_max_u64((unsigned long long)amax);
        if (lane == 0) atomicMax(&p.ctr->absmax, wmax);
        return;
    }
    __syncthreads();
    const int opb = p.out_pb ? p.out_pb : 16;
    int8_t *outb = p.out + (size_t)b * (Ho + 2) * (Wo + 2) * opb;
    for (int w = tid; w < NW; w += 256) {
        const int wy = w / (TW / 2), wx = w % (TW / 2);
        const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
        if (oy < Ho && ox < Wo)
            *(v4i *)(outb + ((size_t)(oy + 1) * (Wo + 2) + ox + 1) * opb) = *(const v4i *)(otile + w * 16);
    }
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
    if (nguard) atomicAdd(&p.ctr->guard, (unsigned long long)nguard);
    if (nsat_in) atomicAdd(&p.ctr->in_sat, (unsigned long long)nsat_in);
}

// ---- production variant: same arithmetic (32-bit epilogue, no statistics / guard), organised so
// that nothing in the hot loops needs an integer division: waves quantise whole patch rows (lanes
// along x) and own whole rows of pooling windows, so LDS addresses advance by constants.
// U8: the input is the camera frame itself (uint8 HWC BGR); BaseTransform's (u/255 - mean)/std, the
// BGR->RGB swap and the HWC->CHW permute (data/__init__.py:30-56, test.py:79) happen in the load, with
// the reference's fp32 operations in the reference's order, so the quantised pixels are identical.
template <int TW, bool U8, bool GEN = false>
__global__ __launch_bounds__(256) void conv1_fast_kernel(const Conv1Params p) {
    constexpr int TH = 16;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int WPR = TW / 2;                    // pooling windows per window-row
    constexpr int NWR = TH / 2;                    // window rows per tile
    constexpr int NW = NWR * WPR;
    constexpr int MPR = WPR / 4;                   // m-tiles (4 windows) per window row
    static_assert(WPR % 4 == 0, "window rows split into whole m-tiles");
    __shared__ __attribute__((aligned(16))) unsigned int patch[PH * PW + 8];
    __shared__ __attribute__((aligned(16))) unsigned char otile[NW * 16];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = y355_xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int H = p.H, W = p.W;
    const int y0 = ty * TH, x0 = tx * TW;
    const float sc = p.in_scale;
    unsigned int nsat_in = 0;

    // ---- quantise the patch: wave w takes patch rows w, w+4, ...; lanes run along x
    const float *xb = U8 ? nullptr : p.x + (size_t)b * 3 * H * W;
    const size_t plane = (size_t)H * W;
    constexpr int XP = (PW + 63) / 64;             // passes along a row
    constexpr int RPW = (PH + 3) / 4;              // patch rows per wave
    if constexpr (U8) {
        // normalise + quantise is a function of the byte: one 256-entry table per channel, built by the
        // workgroup with the reference's fp32 operations in the reference's order ((u/255 - mean)/std,
        // data/__init__.py:43-45; round(x * 2^sa), slim_yolo_v2.py:35), bit 8 = "was clamped"
        __shared__ unsigned short lut[3 * 256];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = (float)tid;
            t /= 255.0f;
            t -= p.nmean[c];
            t /= p.nstd[c];
            const float r = rintf(t * sc);
            const float rc = fminf(fmaxf(r, -127.f), 127.f);
            lut[c * 256 + tid] = (unsigned short)(((int)rc & 0xff) | (rc != r ? 0x100 : 0));
        }
        const uint8_t *fb = p.x_u8 + (size_t)b * H * W * 3;
        unsigned char raw[RPW][XP][3];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int py = wave + 4 * k;
            const int gy = y0 + py - 1;
            const size_t ro = (size_t)min(max(gy, 0), H - 1) * W;
#pragma unroll
            for (int xp = 0; xp < XP; ++xp) {
                const int gx = x0 + xp * 64 + lane - 1;
                const size_t o = (ro + min(max(gx, 0), W - 1)) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) raw[k][xp][c] = fb[o + (2 - c)];      // RGB channel c = BGR byte 2 - c
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int py = wave + 4 * k;
            const int gy = y0 + py - 1;
            const bool rowin = gy >= 0 && gy < H && py < PH;
#pragma unroll
            for (int xp = 0; xp < XP; ++xp) {
                const int px = xp * 64 + lane;
                const int gx = x0 + px - 1;
                const bool inside = rowin && gx >= 0 && gx < W;
                const bool own = inside && py >= 1 && py <= TH && px >= 1 && px <= TW;
                unsigned int w = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned int e = lut[c * 256 + raw[k][xp][c]];
                    nsat_in += (own && (e & 0x100u)) ? 1u : 0u;
                    w |= (inside ? (e & 0xffu) : 0u) << (8 * c);
                }
                if (px < PW && py < PH) patch[py * PW + px] = w;
            }
        }
    } else {
    // all of a wave's rows are loaded before any is quantised: RPW * XP * 3 loads in flight
    float v[RPW][XP][3];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int py = wave + 4 * k;
        const int gy = y0 + py - 1;
        const size_t ro = (size_t)min(max(gy, 0), H - 1) * W;
#pragma unroll
        for (int xp = 0; xp < XP; ++xp) {
            const int gx = x0 + xp * 64 + lane - 1;
            const size_t o = ro + min(max(gx, 0), W - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[k][xp][c] = xb[c * plane + o];
        }
    }
    // 5 VALU ops per value: v_mul, v_rndne, v_med3 (clamp), v_max |r| (saturation DETECTED; counted exactly below only
    // when it happened), v_cvt_i32_f32 with SDWA writing byte c of the pixel word.  Pixels outside the image are
    // zeroed with one select per pixel.
    float satm = 0.f;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int py = wave + 4 * k;
        const int gy = y0 + py - 1;
        const bool rowin = gy >= 0 && gy < H && py < PH;
#pragma unroll
        for (int xp = 0; xp < XP; ++xp) {
            const int px = xp * 64 + lane;
            const int gx = x0 + px - 1;
            const bool inside = rowin && gx >= 0 && gx < W;
            unsigned int w = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r = rintf(v[k][xp][c] * sc);
                const float rc = __builtin_amdgcn_fmed3f(r, -127.f, 127.f);
                satm = fmaxf(satm, fabsf(r));
                if (c == 0) asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(rc));
                else if (c == 1) asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(rc));
                else asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(rc));
            }
            if (px < PW && py < PH) patch[py * PW + px] = inside ? w : 0u;
        }
    }
    if (!(satm <= 127.f)) {                          // cold (also taken for NaN): the exact count over the tile's own pixels
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int py = wave + 4 * k;
            const int gy = y0 + py - 1;
            const bool rowin = gy >= 0 && gy < H && py < PH;
#pragma unroll
            for (int xp = 0; xp < XP; ++xp) {
                const int px = xp * 64 + lane;
                const int gx = x0 + px - 1;
                const bool own = rowin && gx >= 0 && gx < W && py >= 1 && py <= TH && px >= 1 && px <= TW;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float r = rintf(v[k][xp][c] * sc);
                    nsat_in += (own && fminf(fmaxf(r, -127.f), 127.f) != r) ? 1u : 0u;
                }
            }
        }
    }
    }
    if (tid < 8) patch[PH * PW + tid] = 0;
    __syncthreads();

    const int li = lane & 15, g = lane >> 4;
    const v4i bw = *(const v4i *)(p.w + lane * 16);
    const int bias = p.bias_t[li];
    const Requant rq = p.rq;
    const int Ho = H >> 1, Wo = W >> 1;
    unsigned int nsat = 0;
    // sh_l (a left requant shift; sh_r = 0 then) folded into the accumulator shift and the bias; saturation is
    // DETECTED with one v_xad per output and COUNTED exactly by a second, cold pass over the tile only when it happened
    const int shl2 = rq.shl + rq.sh_l;
    const int bias2 = bias << rq.sh_l;
    auto requant = [&](int v) {
        if constexpr (GEN) {
            return y355_requant_gen32(v, bias, rq);        // slope neg_mul / 2^lk (y355_net: LeakyReLU(0.1) = 205 / 2048)
        } else {
            int x = (v << shl2) + bias2;
            x = max(x, x << rq.lk);
            const int rb = (int)__builtin_amdgcn_ubfe((unsigned int)x, (unsigned int)rq.sh_r, (unsigned int)rq.bw);
            return (x + rq.hm1 + rb) >> rq.sh_r;
        }
    };
    // lane geometry inside an m-tile: window li>>2, position li&3 -> pixel (r>>1, 2*(li>>2) + (r&1))
    const int r4 = li & 3;
    const int lbase = ((r4 >> 1) + min(g, 2)) * PW + 2 * (li >> 2) + (r4 & 1);
    unsigned int satx = 0;
#pragma unroll 1
    for (int wy = wave; wy < NWR; wy += 4) {
        const unsigned int *src = patch + lbase + wy * 2 * PW;
        unsigned char *dst = otile + (wy * WPR + g) * 16 + li;
#pragma unroll
        for (int mt = 0; mt < MPR; ++mt) {
            v4i a;
            a[0] = (int)src[mt * 8 + 0];
            a[1] = (int)src[mt * 8 + 1];
            a[2] = (int)src[mt * 8 + 2];
            a[3] = (int)src[mt * 8 + 3];
            v4i acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bw, acc, 0, 0, 0);
            const int vmax = max(max(acc[0], acc[1]), max(acc[2], acc[3]));
            const int qq = requant(vmax);
            const int q = y355_clamp8<int>(qq);
            satx += (unsigned int)(q ^ qq);
            dst[mt * 64] = (unsigned char)(q & 0xff);
        }
    }
    if (__builtin_amdgcn_ballot_w64(satx != 0) != 0ull) {        // wave-uniform: the recount re-runs MFMAs (all lanes feed them)
#pragma unroll 1
        for (int wy = wave; wy < NWR; wy += 4) {
            const unsigned int *src = patch + lbase + wy * 2 * PW;
#pragma unroll 1
            for (int mt = 0; mt < MPR; ++mt) {
                v4i a;
                a[0] = (int)src[mt * 8 + 0];
                a[1] = (int)src[mt * 8 + 1];
                a[2] = (int)src[mt * 8 + 2];
                a[3] = (int)src[mt * 8 + 3];
                v4i acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bw, acc, 0, 0, 0);
                const int qq = requant(max(max(acc[0], acc[1]), max(acc[2], acc[3])));
                const int wx = mt * 4 + g;
                nsat += (y355_clamp8<int>(qq) != qq && (y0 >> 1) + wy < Ho && (x0 >> 1) + wx < Wo) ? 1u : 0u;
            }
        }
    }
    __syncthreads();
    const int opb = p.out_pb ? p.out_pb : 16;                // bytes per output pixel (wider buffers: the rest stays as it is)
    int8_t *outb = p.out + (size_t)b * (Ho + 2) * (Wo + 2) * opb;
    for (int w = tid; w < NW; w += 256) {
        const int wy = w / WPR, wx = w % WPR;
        const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
        if (oy < Ho && ox < Wo)
            *(v4i *)(outb + ((size_t)(oy + 1) * (Wo + 2) + ox + 1) * opb) = *(const v4i *)(otile + w * 16);
    }
    if (nsat) atomicAdd(&p.ctr->sat, (unsigned long long)nsat);
    if (nsat_in) atomicAdd(&p.ctr->in_sat, (unsigned long long)nsat_in);
}

static int conv1_tw(int W) { return (W % 104 == 0) ? 104 : 32; }

void y355_conv1_tiles(int H, int W, int *tx, int *ty) {
    const int tw = conv1_tw(W);
    *tx = (W + tw - 1) / tw;
    *ty = (H + 15) / 16;
}

void y355_launch_conv1(const Conv1Params &p, hipStream_t s) {
    const int n = p.tiles_x * p.tiles_y * p.B;
    const bool big = conv1_tw(p.W) == 104;
    if (p.mode == 0 && !p.guard && p.rq.gen32 && p.x && p.bias_t) {       // y355_net's first layer, general slope in 32 bits
        if (big) hipLaunchKernelGGL((conv1_fast_kernel<104, false, true>), dim3(n), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv1_fast_kernel<32, false, true>), dim3(n), dim3(256), 0, s, p);
        return;
    }
    if (p.mode == 0 && !p.rq.wide && !p.guard && !p.out_pb) {
        if (p.x) {
            if (big) hipLaunchKernelGGL((conv1_fast_kernel<104, false>), dim3(n), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv1_fast_kernel<32, false>), dim3(n), dim3(256), 0, s, p);
        } else {
            if (big) hipLaunchKernelGGL((conv1_fast_kernel<104, true>), dim3(n), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv1_fast_kernel<32, true>), dim3(n), dim3(256), 0, s, p);
        }
        return;
    }
    if (p.rq.wide) {
        if (big) hipLaunchKernelGGL((conv1_kernel<104, true>), dim3(n), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv1_kernel<32, true>), dim3(n), dim3(256), 0, s, p);
    } else {
        if (big) hipLaunchKernelGGL((conv1_kernel<104, false>), dim3(n), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv1_kernel<32, false>), dim3(n), dim3(256), 0, s, p);
    }
}

// B fragment of the single k-step: lane (g = filter row, j = cout) holds k = 4*d + c for
// pixel column d (0..3, d = 3 unused) and colour c (0..3, c = 3 unused); g = 3 unused.
void y355_pack_conv1(const int8_t *q_w, int8_t *dst) {
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, j = l & 15;
        for (int kk = 0; kk < 16; ++kk) {
            const int d = kk >> 2, c = kk & 3;
            int8_t v = 0;
            if (g < 3 && d < 3 && c < 3) v = q_w[((j * 3 + c) * 3 + g) * 3 + d];
            dst[l * 16 + kk] = v;
        }
    }
}

// ---- max |x| of the network input (tracker 0, slim_yolo_v2.py:22) ------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float *x, size_t n, unsigned int *out) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *(const float4 *)(x + i);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        } else {
            for (size_t k = i; k < n; ++k) m = fmaxf(m, fabsf(x[k]));
        }
    }
    unsigned int u = y355_wave_max_u32(__float_as_uint(m));
    if ((threadIdx.x & 63) == 0) atomicMax(out, u);
}

void y355_launch_absmax(const float *x, size_t n, unsigned int *out_bits, hipStream_t s) {
    int blocks = (int)((n / 4 + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, x, n, out_bits);
}

// ---- BaseTransform on the GPU for the paths that take fp32 (statistics / 64-bit epilogue / guard) ------
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t *frames, float *x, int B, int H, int W, float m0, float m1,
                                                           float m2, float s0, float s1, float s2) {
    const size_t n = (size_t)B * H * W;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / ((size_t)H * W), r = i % ((size_t)H * W);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = (float)frames[i * 3 + (2 - c)];
            t /= 255.0f;
            t -= mean[c];
            t /= sd[c];
            x[(b * 3 + c) * (size_t)H * W + r] = t;
        }
    }
}

void y355_launch_normalize_u8(const uint8_t *frames, float *x, int B, int H, int W, const float *mean_rgb, const float *std_rgb,
                              hipStream_t s) {
    const size_t n = (size_t)B * H * W;
    int blocks = (int)((n + 255) / 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(blocks), dim3(256), 0, s, frames, x, B, H, W, mean_rgb[0], mean_rgb[1], mean_rgb[2],
                       std_rgb[0], std_rgb[1], std_rgb[2]);
}
