// LDS operation cost on gfx950 by kind: plain 4-byte reads / writes, read-modify-write by three instructions, and the
// float atomic add - what the backward's scatter of SSIM partials can be built from.
//   hipcc --offload-arch=gfx950 -O3 -w -o lds_rate tools/microbench/lds_rate.hip && ./lds_rate > profiles/r03/lds_rate.txt
// 256-thread blocks (one wave per SIMD), blocks per CU = waves per SIMD; every lane works on its own word (+ a rotating
// offset): conflict-free.  Reported: shader cycles per wave-instruction PER CU (the LDS is shared by the CU's 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>

enum Kind { READ32 = 0, WRITE32, RMW3, ATOMIC_F32, ATOMIC_F32_SAME9, ATOMIC_U32, KINDS };
static const char* kind_name[KINDS] = {"ds_read_b32", "ds_write_b32", "ds_read_b32 + v_add_f32 + ds_write_b32", "ds_add_f32 (distinct words)",
                                       "ds_add_f32 (9 lanes per word)", "ds_add_u32 (distinct words)"};
constexpr int UNROLL = 32;

template <int KIND>
__global__ __launch_bounds__(256) void lds_kernel(float* out, unsigned long long* clk, int iters) {
  __shared__ float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = i * 0.5f;
  __syncthreads();
  const int lane = threadIdx.x;
  float acc = 0.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int idx = (lane + 256 * (u & 7) + it) & 4095;
      if (KIND == READ32) acc += buf[idx];
      else if (KIND == WRITE32) buf[idx] = acc + u;
      else if (KIND == RMW3) buf[idx] = buf[idx] + 1.0f;
      else if (KIND == ATOMIC_F32) atomicAdd(&buf[idx], 1.0f);
      else if (KIND == ATOMIC_F32_SAME9) atomicAdd(&buf[((lane / 9) + 256 * (u & 7) + it) & 4095], 1.0f);
      else if (KIND == ATOMIC_U32) atomicAdd(reinterpret_cast<unsigned*>(&buf[idx]), 1u);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc + buf[lane];
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
static void run(float* out, unsigned long long* clk) {
  const int iters = 2000;
  const double per = KIND == RMW3 ? 2.0 : 1.0;     // LDS instructions per element
  for (int wps : {1, 2, 4}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double instr_per_cu = (double)iters * UNROLL * per * wps * 4;      // 4 waves per block
    const double ns = ms * 1e6 / instr_per_cu;
    printf("%-42s waves/SIMD %d: %8.3f ms  %.3f ns per LDS wave-instruction per CU  clock %.2f GHz -> %.2f cycles  (per element: %.1f cycles per CU)\n",
           kind_name[KIND], wps, ms, ns, ghz, ns * ghz, ns * ghz * per);
  }
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  hipMalloc(&clk, 64);
  run<READ32>(out, clk);
  run<WRITE32>(out, clk);
  run<RMW3>(out, clk);
  run<ATOMIC_F32>(out, clk);
  run<ATOMIC_F32_SAME9>(out, clk);
  run<ATOMIC_U32>(out, clk);
  return 0;
}
