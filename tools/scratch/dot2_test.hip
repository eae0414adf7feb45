#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* o) {
  int t = threadIdx.x;
  float acc = 1.0f;
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a[t]), __builtin_bit_cast(h2, b[t]), acc, false);
  o[t] = acc;
  float acc2 = 1.0f;
  acc2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a[t]), __builtin_bit_cast(b2, b[t]), acc2, false);
  o[64 + t] = acc2;
  h2 x = __builtin_bit_cast(h2, a[t]), y = __builtin_bit_cast(h2, b[t]);
  o[128 + t] = 1.0f + (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1];
}
int main() {
  uint32_t ha[64], hb[64];
  for (int i = 0; i < 64; ++i) {
    _Float16 v0 = (_Float16)(0.5f * i), v1 = (_Float16)(-1.25f + i), w0 = (_Float16)2.0f, w1 = (_Float16)(0.25f * i);
    uint16_t r0, r1, s0, s1;
    __builtin_memcpy(&r0, &v0, 2); __builtin_memcpy(&r1, &v1, 2); __builtin_memcpy(&s0, &w0, 2); __builtin_memcpy(&s1, &w1, 2);
    ha[i] = r0 | ((uint32_t)r1 << 16); hb[i] = s0 | ((uint32_t)s1 << 16);
  }
  uint32_t *da, *db; float* dout; float ho[192];
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 192 * 4);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
  hipMemcpy(ho, dout, 192 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) printf("i=%d fdot2=%g  scalar=%g  (bf16-interp=%g)\n", i, ho[i], ho[128 + i], ho[64 + i]);
  return 0;
}
