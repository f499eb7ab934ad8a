#include <hip/hip_runtime.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i* a, const v4i* b, v4f* c, unsigned short* o, const float* f){
  v4f acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[threadIdx.x]), __builtin_bit_cast(v8bf, b[threadIdx.x]), acc, 0,0,0);
  c[threadIdx.x]=acc;
  __bf16 h = (__bf16)f[threadIdx.x];
  o[threadIdx.x] = __builtin_bit_cast(unsigned short, h);
}
