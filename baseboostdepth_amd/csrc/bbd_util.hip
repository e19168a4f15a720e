// bbd_util.hip - measurement aid: the on-box stream-copy ceiling bench.py quotes beside the 8 TB/s specification
// ("also record an on-box measured stream-copy ceiling and quote both fractions").  A plain float4 copy:
// one 16-byte load and one 16-byte store per thread per step, grid-stride, enough workgroups to fill 256 CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"

namespace {
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int bbd_stream_copy(const float* src, float* dst, long n_floats, void* stream) {
  if (!src || !dst || n_floats <= 0 || (n_floats & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return BBD_E_BADARG;
  const long n4 = n_floats / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;          // 32 workgroups per CU, each thread a few KB in flight over its loop
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
