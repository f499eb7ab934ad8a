#include <hip/hip_runtime.h>
// v_max_f32 with an SDWA byte destination: does byte k of d receive bits [7:0] of the fp32 result, the rest preserved?
__global__ void kmax(const float *x, unsigned int *out) {
    float a = x[threadIdx.x], b = x[threadIdx.x + 64], c = x[threadIdx.x + 128];
    const float M = 12582912.0f;
    float ya = a + M, yb = b + M, yc = c + M;          // bit patterns 0x4B400000 + value
    unsigned int w = 0xdeadbeefu;
    asm volatile("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(ya), "v"(yb));
    asm volatile("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(yb), "v"(yc));
    asm volatile("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(ya), "v"(yc));
    out[threadIdx.x] = w;
}
__global__ void k(const float *x, unsigned int *out) {
    float a = x[threadIdx.x], b = x[threadIdx.x + 64], c = x[threadIdx.x + 128];
    unsigned int w = 0;
    asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(a));
    asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(b));
    asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(c));
    out[threadIdx.x] = w;
}
int main() {
    float *x; unsigned int *o; hipMalloc(&x, 192 * 4); hipMalloc(&o, 256);
    float h[192]; for (int i = 0; i < 192; ++i) h[i] = (float)((i * 37) % 255 - 127);
    hipMemcpy(x, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, o);
    unsigned int r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        unsigned int want = ((unsigned)(int)h[i] & 0xff) | (((unsigned)(int)h[i + 64] & 0xff) << 8) | (((unsigned)(int)h[i + 128] & 0xff) << 16);
        bad += want != r[i];
    }
    printf("sdwa cvt pack: %d mismatches, sample %08x\n", bad, r[5]);
    kmax<<<1, 64>>>(x, o);
    hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int i = 0; i < 64; ++i) {
        auto mx = [](float p, float q) { return (unsigned)(int)(p > q ? p : q) & 0xffu; };
        unsigned int want = mx(h[i], h[i + 64]) | (mx(h[i + 64], h[i + 128]) << 8) | (mx(h[i], h[i + 128]) << 16) | 0xde000000u;
        bad2 += want != r[i];
    }
    printf("sdwa v_max_f32 byte pack: %d mismatches, sample %08x\n", bad2, r[5]);
    return (bad | bad2) != 0;
}
