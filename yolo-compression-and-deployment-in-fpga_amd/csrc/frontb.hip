// yolo355 -- fused front end of the bf16 nets (round 4): fp32 NCHW input -> bf16 -> conv(3 -> 16) + bias + LeakyReLU + 2x2 max
// pool -> conv(16 -> 32) + bias + LeakyReLU + 2x2 max pool -> bf16 NHWC32 with halo, ONE kernel.
//
// Replaces SlimYOLOv2.conv1 + pool1 + conv2 + pool2 (models/slim_yolo_v2.py:549-575, BatchNorm folded on the host) and
// DarkNet_Light.conv_1 + maxpool_1 + conv_2 + maxpool_2 of YOLOv3tiny (backbone/darknet.py:216-220): the two launches
// conv1_bf16_kernel + convg_kernel<true, 32, 32, 16, 52, pool> of csrc/convg.hip (69 + 64 us at B = 64, the 16-channel map
// written to HBM and read back).
//
// Same construction as the int8 front end (front.hip): the 2x2 pooling window is the unit of work, its four conv outputs are
// MFMAs over ONE neighbourhood operand with four shifted weight fragments, the WEIGHTS are the A operand (rows = output
// channels) and the windows the B operand, so a lane holds four channels of one window and the pool is an element-wise max.
// bf16 doubles the bytes of K: conv1's 4x4-pixel x 4-channel neighbourhood is 64 elements = two v_mfma_f32_16x16x32_bf16
// (neighbourhood rows 0-1, then 2-3), conv2's neighbourhood row is 4 pixels x 16 channels = two MFMAs (pixels 0-1, 2-3).
// Arithmetic = the two-launch path's: operands rounded to bf16 (RNE), fp32 accumulation, + bias, LeakyReLU, pool, rounded to
// bf16 between the layers; only the order of the fp32 accumulation differs (tolerances of tests/test_fp32_models.py).
//
// 512 threads, two workgroups per CU (56.6 KB of LDS each: input patch of 8-byte pixels, conv1's pooled 16-channel map, the
// staged output aliased onto the patch).  conv2's weight fragments stay in registers for the whole launch (waves 0-3 own
// output channels {8 g + r}, waves 4-7 {8 g + 4 + r}: 12 fragments each); conv1's eight fragments are read from LDS.
#include "y355_common.h"
#include <cstring>

namespace {
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int TOY = 13, TOX = 13;                    // pooled conv2 outputs per tile
constexpr int P1H = 2 * TOY + 2, P1W = 2 * TOX + 2;  // pooled conv1 tile with its halo
constexpr int PH0 = 2 * P1H + 2;                     // input patch rows (= columns used)
constexpr int P0 = 64;                               // patch pitch in 8-byte pixels
constexpr int P1P = P1W;                             // p1 pitch in 32-byte pixels
constexpr int P1ROWS = P1H + 2;                      // slack rows: the clamped padding windows of C2 stay inside
constexpr int NW1 = P1H * P1W, NG1 = NW1 / 16;       // 784 conv1 windows = 49 groups of 16
constexpr int NW2 = TOY * TOX, NG2 = (NW2 + 15) / 16;   // 169 conv2 windows = 11 groups
constexpr int NTHR = 512, NWAVE = 8;
constexpr int W1_BYTES = 8 * 1024, W2_BYTES = 24 * 1024;
static_assert(NW1 % 16 == 0 && PH0 <= 4 * NWAVE * 2 && P0 >= PH0 + 2, "front geometry");

__device__ __forceinline__ void fb_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned int row_next(unsigned int v) {        // lane i of each row of 16 lanes gets lane i + 1's value
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
}
// bare instructions: hipcc canonicalises (quiets) both operands of fmaxf chains and turns `m >= 0 ? m : m * s` into a compare + select
// (v_cndmask on vcc: 22 cycles back to back, profiles/r04_notes.md section 9)
__device__ __forceinline__ float fb_max(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned int pk_bf16(float a, float b) {       // (a, b) -> two bf16 (RNE), a in the low half
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    const v2bf v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}
}  // namespace

__global__ __launch_bounds__(512, 4) void frontb_kernel(const FrontBParams p, const int total_tiles) {
    __shared__ __attribute__((aligned(16))) uint2 patch[PH0 * P0];        // bf16 (r, g, b, 0) pixels; later the staged output tile
    __shared__ __attribute__((aligned(16))) char p1[P1ROWS * P1P * 32];   // conv1's pooled map, 16 bf16 channels per pixel
    __shared__ __attribute__((aligned(16))) char wl[W1_BYTES + 64];       // conv1's fragments and biases
    char *stg = (char *)patch;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;
    const int Hp = H >> 1, Wp = W >> 1, Ho = H >> 2, Wo = W >> 2;
    const size_t plane = (size_t)H * W;
    const float slope1 = p.slope1, slope2 = p.slope2;      // 0 <= slope <= 1 (y355_launch_frontb's callers): LeakyReLU(m) = max(m, m * slope)

    int tile = y355_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;
    const int G_ = gridDim.x;
    // ---- once per workgroup: conv1's fragments into LDS, this wave's conv2 fragments and biases into registers
    *(v4i *)(wl + tid * 16) = *(const v4i *)(p.wf + tid * 16);
    if (tid < 4) *(v4i *)(wl + W1_BYTES + 16 * tid) = *(const v4i *)(p.bias1 + 4 * tid);
    const int npass = wave >> 2;                           // this wave's half of conv2's output channels: 8 g + 4 npass + r
    v4i w2[3][2][2];                                       // [filter row][dx][pixel pair]
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                w2[ky][dx][h] = *(const v4i *)(p.wf + W1_BYTES + ((((npass * 3 + ky) * 2 + dx) * 2 + h) * 1024) + lane * 16);
    const float4 b2q = *(const float4 *)(p.bias2 + 8 * g + 4 * npass);
    const float b2[4] = {b2q.x, b2q.y, b2q.z, b2q.w};
    fb_barrier();

    // Q: wave-item q = wave + 8 k covers patch rows 4 q .. 4 q + 3; lane = 16 * (row in the item) + 4-pixel group j (j = 15 idle)
    const int qr0 = 4 * wave + g, qj = li;

    // tile -> (tx, ty, b) once; the next tile of this workgroup is one grid further: three carries (step_* = the grid size in the
    // tile index's mixed radix, from the launcher) instead of three integer divisions by run-time values per tile (front.hip, round 6)
    int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    for (;; tile += G_) {
        int li_ = li, g_ = g, tid_ = tid;
        asm volatile("" : "+v"(li_), "+v"(g_), "+v"(tid_));          // per-tile copies: derived addresses are not hoisted out of the loop
        const int y0p = 4 * TOY * ty - 3, x0p = 4 * TOX * tx - 4;
        const bool border = ty == 0 || tx == 0 || ty == p.tiles_y - 1 || tx == p.tiles_x - 1;

        // ---- Q: fp32 planes -> bf16 pixels (r, g, b, 0) in the LDS patch.  Patch column L holds global column x0p + 1 + L (the
        // neighbourhood of every conv1 window then starts on a 16-byte boundary): the first pixel of a lane's aligned 4-pixel
        // group belongs to the previous lane's 32 bytes
        {
            float4 vf[2][3];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = qr0 + 32 * k;
                const int gy = min(max(y0p + r, 0), H - 1);
                const int gx = min(max(x0p + 4 * qj, 0), W - 4);
                const float *xb = p.x + (size_t)b * 3 * plane + (size_t)gy * W + gx;
#pragma unroll
                for (int c = 0; c < 3; ++c) vf[k][c] = *(const float4 *)(xb + c * plane);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = qr0 + 32 * k;
                const int gy = y0p + r, gx = x0p + 4 * qj;
                unsigned int lo[4], hi[4];
                const float xr[4] = {vf[k][0].x, vf[k][0].y, vf[k][0].z, vf[k][0].w};
                const float xg[4] = {vf[k][1].x, vf[k][1].y, vf[k][1].z, vf[k][1].w};
                const float xbv[4] = {vf[k][2].x, vf[k][2].y, vf[k][2].z, vf[k][2].w};
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    lo[px] = pk_bf16(xr[px], xg[px]);
                    hi[px] = pk_bf16(xbv[px], 0.f);
                }
                if (border) {                                 // wave-uniform: interior tiles (36 of 64) have no padding to select
                    const bool zero = !((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W);          // conv1's zero padding
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        lo[px] = zero ? 0u : lo[px];
                        hi[px] = zero ? 0u : hi[px];
                    }
                }
                const v4i a = {(int)lo[1], (int)hi[1], (int)lo[2], (int)hi[2]};
                const v4i c = {(int)lo[3], (int)hi[3], (int)row_next(lo[0]), (int)row_next(hi[0])};
                if (r < PH0 && qj < 15) {
                    *(v4i *)(patch + r * P0 + 4 * qj) = a;
                    *(v4i *)(patch + r * P0 + 4 * qj + 2) = c;
                }
            }
        }
        fb_barrier();                                       // B1: patch complete

        // ---- C1: conv1 + bias + LeakyReLU + pool1 -> p1.  Group = 16 consecutive windows of the 28 x 28 window grid; lane
        // (li, g): window li; MFMA h takes neighbourhood rows 2 h, 2 h + 1: lane group g = (row g >> 1, pixel pair g & 1)
        {
            const float4 b1q = *(const float4 *)(wl + W1_BYTES + 16 * g_);
            const float b1[4] = {b1q.x, b1q.y, b1q.z, b1q.w};
            const int gyp0 = 2 * TOY * ty - 1, gxp0 = 2 * TOX * tx - 1;
#pragma unroll 1
            for (int grp = wave; grp < NG1; grp += NWAVE) {
                const int w = grp * 16 + li_;
                const int py = (w * 2341) >> 16;              // w / 28 for w < 784
                const int px = w - py * P1W;
                const uint2 *src = patch + (2 * py + (g_ >> 1)) * P0 + 2 * px + 2 * (g_ & 1);
                const v4i q0 = *(const v4i *)src, q1 = *(const v4i *)(src + 2 * P0);
                v4f acc[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const v4i wa = *(const v4i *)(wl + (2 * v) * 1024 + lane * 16), wb = *(const v4i *)(wl + (2 * v + 1) * 1024 + lane * 16);
                    acc[v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wa), __builtin_bit_cast(v8bf, q0), (v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc[v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wb), __builtin_bit_cast(v8bf, q1), acc[v], 0, 0, 0);
                }
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (the pool stays C++: an inline-asm read of an MFMA result gets no hazard wait states from the compiler -- the first
                    // version of this line read accumulators the matrix pipe had not written yet, and two forwards of one input differed)
                    const float m = fmaxf(fmaxf(acc[0][r], acc[1][r]), fmaxf(acc[2][r], acc[3][r])) + b1[r];
                    y[r] = fb_max(m, m * slope1);
                }
                uint2 o = make_uint2(pk_bf16(y[0], y[1]), pk_bf16(y[2], y[3]));
                if (border) {                                 // windows outside the image: conv2's zero padding
                    const bool inimg = (unsigned)(gyp0 + py) < (unsigned)Hp && (unsigned)(gxp0 + px) < (unsigned)Wp;
                    if (!inimg) o = make_uint2(0u, 0u);
                }
                *(uint2 *)(p1 + (py * P1P + px) * 32 + 8 * g_) = o;
            }
        }
        fb_barrier();                                       // B2: p1 complete (and the patch is dead: stg may be written)

        // ---- C2: conv2 + bias + LeakyReLU + pool2 -> staged bf16 tile.  Group = 16 windows of the 13 x 13 grid (the last
        // group's padding slots repeat window 168); neighbourhood row t, MFMA h = pixels 2 h, 2 h + 1: lane group g = (pixel
        // g >> 1, channels 8 (g & 1) ..)
        {
#pragma unroll 1
            for (int grp = wave & 3; grp < NG2; grp += 4) {
                const int wraw = grp * 16 + li_;
                const int w = min(wraw, NW2 - 1);
                const int wy = (w * 5042) >> 16;              // w / 13 for w < 169
                const int wx = w - wy * TOX;
                const char *src = p1 + ((2 * wy) * P1P + 2 * wx + (g_ >> 1)) * 32 + 16 * (g_ & 1);
                v4f acc[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) acc[dy][dx] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const v4i q0 = *(const v4i *)(src + t * P1P * 32), q1 = *(const v4i *)(src + t * P1P * 32 + 64);
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int ky = t - dy;
                        if (ky < 0 || ky > 2) continue;
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            acc[dy][dx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, w2[ky][dx][0]), __builtin_bit_cast(v8bf, q0), acc[dy][dx], 0, 0, 0);
                            acc[dy][dx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, w2[ky][dx][1]), __builtin_bit_cast(v8bf, q1), acc[dy][dx], 0, 0, 0);
                        }
                    }
                }
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m = fmaxf(fmaxf(acc[0][0][r], acc[0][1][r]), fmaxf(acc[1][0][r], acc[1][1][r])) + b2[r];
                    y[r] = fb_max(m, m * slope2);
                }
                *(uint2 *)(stg + wraw * 64 + 16 * g_ + 8 * npass) = make_uint2(pk_bf16(y[0], y[1]), pk_bf16(y[2], y[3]));
            }
        }
        fb_barrier();                                       // B3: staged tile complete

        // ---- OUT: NHWC (32 bf16 channels at the head of a pixel of out_pb bytes) with halo, 16 bytes per thread and item
        {
            char *outb = p.out + (((size_t)b * (Ho + 2) + TOY * ty + 1) * (Wo + 2) + TOX * tx + 1) * p.out_pb;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int item = tid_ + NTHR * k;
                const int wdw = item >> 2;
                const int row = (wdw * 5042) >> 16, col = wdw - row * TOX;
                if (item < NW2 * 4 && TOY * ty + row < Ho && TOX * tx + col < Wo)
                    *(v4i *)(outb + ((size_t)row * (Wo + 2) + col) * p.out_pb + (item & 3) * 16) = *(const v4i *)(stg + item * 16);
            }
        }
        if (tile + G_ >= total_tiles) break;
        tx += p.step_x;
        if (tx >= p.tiles_x) { tx -= p.tiles_x; ++ty; }
        ty += p.step_y;
        if (ty >= p.tiles_y) { ty -= p.tiles_y; ++b; }
        b += p.step_b;
        fb_barrier();                                       // B4: the staged tile (= the patch) has been read out
    }
}

void y355_frontb_tiles(int H, int W, int *tx, int *ty) {
    *tx = (W / 4 + TOX - 1) / TOX;
    *ty = (H / 4 + TOY - 1) / TOY;
}

static unsigned short bf16_rne(float v) {
    unsigned int u;
    memcpy(&u, &v, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    return (unsigned short)u;
}

// Weight fragments of the bf16 front end (32 KiB): MFMA A operands, lane (i = l & 15: accumulator row, g = l >> 4), 8 bf16.
//   conv1, variant v = 2 dy + dx (offset of the conv output inside the pooling window), half h, at (2 v + h) * 1024:
//     row i = output channel i; element e of lane group g: neighbourhood row 2 h + (g >> 1), pixel 2 (g & 1) + (e >> 2), colour e & 3
//   conv2, fragment (((n * 3 + ky) * 2 + dx) * 2 + h) at 8192 + ... * 1024:
//     row i = output channel 8 (i >> 2) + 4 n + (i & 3); element e of lane group g: pixel 2 h + (g >> 1), input channel 8 (g & 1) + e
// A null tensor leaves its part untouched.
void y355_pack_frontb(const float *w1 /*[16][3][3][3]*/, const float *w2 /*[32][16][3][3]*/, char *dst /*32768*/) {
    if (w1) {
        memset(dst, 0, W1_BYTES);
        for (int v = 0; v < 4; ++v)
            for (int h = 0; h < 2; ++h)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, g = l >> 4;
                    for (int e = 0; e < 8; ++e) {
                        const int ky = 2 * h + (g >> 1) - (v >> 1), kx = 2 * (g & 1) + (e >> 2) - (v & 1), c = e & 3;
                        if (ky < 0 || ky > 2 || kx < 0 || kx > 2 || c > 2) continue;
                        const unsigned short hb = bf16_rne(w1[((i * 3 + c) * 3 + ky) * 3 + kx]);
                        memcpy(dst + (2 * v + h) * 1024 + l * 16 + e * 2, &hb, 2);
                    }
                }
    }
    if (w2) {
        memset(dst + W1_BYTES, 0, W2_BYTES);
        for (int n = 0; n < 2; ++n)
            for (int ky = 0; ky < 3; ++ky)
                for (int dx = 0; dx < 2; ++dx)
                    for (int h = 0; h < 2; ++h)
                        for (int l = 0; l < 64; ++l) {
                            const int i = l & 15, g = l >> 4;
                            const int ch = 8 * (i >> 2) + 4 * n + (i & 3), kx = 2 * h + (g >> 1) - dx;
                            if (kx < 0 || kx > 2) continue;
                            for (int e = 0; e < 8; ++e) {
                                const int ci = 8 * (g & 1) + e;
                                const unsigned short hb = bf16_rne(w2[((ch * 16 + ci) * 3 + ky) * 3 + kx]);
                                memcpy(dst + W1_BYTES + ((((n * 3 + ky) * 2 + dx) * 2 + h) * 1024) + l * 16 + e * 2, &hb, 2);
                            }
                        }
    }
}

void y355_launch_frontb(const FrontBParams &p, hipStream_t s) {
    const int total = p.tiles_x * p.tiles_y * p.B;
    int grid = y355_cu_count() * 2;                                   // two persistent workgroups per CU
    if (grid > total) grid = total;
    FrontBParams q = p;
    q.step_x = grid % p.tiles_x;                           // the walk's stride (one grid) in the tile index's mixed radix
    q.step_y = (grid / p.tiles_x) % p.tiles_y;
    q.step_b = grid / (p.tiles_x * p.tiles_y);
    hipLaunchKernelGGL(frontb_kernel, dim3(grid), dim3(NTHR), 0, s, q, total);
}
