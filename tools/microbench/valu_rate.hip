// How many cycles does one wave64 fp32 VALU instruction occupy a gfx950 SIMD for, as a function
// of waves per SIMD?  (planning input for the fused kernels: are they VALU-issue-bound?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void chain(float* out, int iters, float b, float c) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {          // fma
      a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
      a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
    } else if (KIND == 1) {   // separate mul + add (what -ffp-contract=off code looks like)
      a0 = a0 * b; a1 = a1 + c; a2 = a2 * b; a3 = a3 + c; a4 = a4 * b; a5 = a5 + c; a6 = a6 * b; a7 = a7 + c;
    } else {                  // compare + select
      a0 = a0 > c ? a0 : b + a0; a1 = a1 > c ? a1 : b + a1; a2 = a2 > c ? a2 : b + a2; a3 = a3 > c ? a3 : b + a3;
      a4 = a4 * b; a5 = a5 * b; a6 = a6 * b; a7 = a7 * b;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 32 * 64 * sizeof(float) * 2);
  const int iters = 20000;
  for (int kind = 0; kind < 3; ++kind)
    for (int wps : {1, 2, 4, 8}) {
      const int threads = 256;                      // 4 waves = 1 wave per SIMD per block
      const int blocks = 256 * wps;                 // blocks per CU = waves per SIMD
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (kind == 1) hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (kind == 2) hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      float cyc;
      hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
      const double instr_per_simd = (double)iters * 8 * wps;        // wave-instructions per SIMD
      printf("kind %d waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD  (clock64 ticks/instr for one wave: %.2f)\n",
             kind, wps, ms, ms * 1e6 / instr_per_simd, cyc / (iters * 8.0));
    }
  return 0;
}
