// VALU issue rates on gfx950 (wave64): cycles per instruction per SIMD for the multiply-add forms a depthwise convolution can use.
// One workgroup of W waves per SIMD x 4 SIMDs on ONE CU; 16 independent accumulators per form; s_memtime around 64 x 256 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

template <int FORM> __global__ void k(float* out, unsigned long long* cyc, float seed) {
  float a[16]; f2_t a2[16];
  half2_t h = {(_Float16)seed, (_Float16)(seed * 0.5f)}, w = {(_Float16)0.25f, (_Float16)0.125f};
  unsigned hu = __builtin_bit_cast(unsigned, h), wu = __builtin_bit_cast(unsigned, w);
  for (int i = 0; i < 16; ++i) { a[i] = seed * i; a2[i] = f2_t{seed, seed * i}; }
  float fs = seed, fw = 0.5f;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (FORM == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(fs), "v"(fw));
        if (FORM == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(hu), "v"(wu));
        if (FORM == 2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(hu), "v"(wu));
        if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(a2[(i + 1) & 15]), "v"(a2[(i + 2) & 15]));
        if (FORM == 4) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(hu));
        if (FORM == 5) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(hu), "v"(wu));
        if (FORM == 6) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(hu), "v"(wu));
        if (FORM == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(hu), "v"(wu), "v"(0x05040100));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + a2[i][0] + a2[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1;
  if ((threadIdx.x & 63) == 0) cyc[16 + (threadIdx.x >> 6)] = t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 512);
  const char* names[] = {"v_fma_f32", "v_fma_mix_f32", "v_dot2_f32_f16", "v_pk_fma_f32", "v_cvt_f32_f16", "v_pk_fma_f16", "v_dot2c_f32_f16", "v_perm_b32"};
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int f = 0; f < 8; ++f) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (f) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
          case 7: hipLaunchKernelGGL(k<7>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 1.5f); break;
        }
        hipDeviceSynchronize();
      }
      unsigned long long cc[32]; hipMemcpy(cc, cyc, 256, hipMemcpyDeviceToHost);
      unsigned long long tmax = 0, tmin = ~0ull; for (int i = 0; i < 4 * wps; ++i) { if (cc[i] > tmax) tmax = cc[i]; if (cc[16 + i] < tmin) tmin = cc[16 + i]; }
      unsigned long long c = tmax - tmin;
      // s_memtime counts at a constant 100 MHz on gfx9? readcyclecounter = s_memtime (shader clock domain on gfx950) -> report raw ticks per instruction per SIMD
      printf("%d wave(s)/SIMD  %-18s %8.3f ticks per wave-instruction, %8.3f per SIMD-instruction slot\n", wps, names[f], (double)c / (64.0 * 256), (double)c / (64.0 * 256 * wps));
    }
  }
  return 0;
}
