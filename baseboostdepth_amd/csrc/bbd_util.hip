// bbd_util.hip - measurement aid: the on-box stream-copy ceiling bench.py quotes beside the 8 TB/s specification
// ("also record an on-box measured stream-copy ceiling and quote both fractions").  A plain float4 copy:
// U independent 16-byte loads per thread in flight, then their U stores, grid-stride over chunks of 256 * U float4,
// 32 workgroups per CU.  bench.py times U = 1, 4, 8 and quotes the best.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
  const long chunk = 256L * U, stride = (long)gridDim.x * chunk;
  for (long base = (long)blockIdx.x * chunk + threadIdx.x; base < n4; base += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + 256L * u < n4) v[u] = __builtin_nontemporal_load(src + base + 256L * u);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + 256L * u < n4) __builtin_nontemporal_store(v[u], dst + base + 256L * u);
  }
}
// Pose-pass input of the pooled step: row r of out [R, 2, chw] = the image pair (pool[idx_a[r]], pool[idx_b[r]]), each
// texel (v - sub) * mul - the encoder's input normalisation (networks/resnet_encoder.py:83: (x - 0.45) / 0.225, which
// PyTorch-ROCm evaluates as a subtraction and a multiplication by the float reciprocal, two roundings) folded into the
// gather: one pass instead of two index_select, a cat, a sub and a mul over [R, 6, H, W].
__global__ __launch_bounds__(256) void gather_pairs_kernel(const f4* __restrict__ pool, const int32_t* __restrict__ idx_a,
                                                            const int32_t* __restrict__ idx_b, f4* __restrict__ out, long chw4,
                                                            float sub, float mul) {
  const int slot = blockIdx.y;                                // 2 * r + {0: first image, 1: second image}
  const int32_t row = (slot & 1) ? idx_b[slot >> 1] : idx_a[slot >> 1];
  const f4* src = pool + (long)row * chw4;
  f4* dst = out + (long)slot * chw4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chw4; i += (long)gridDim.x * 256) {
    f4 v = src[i];
    v.x = (v.x - sub) * mul; v.y = (v.y - sub) * mul; v.z = (v.z - sub) * mul; v.w = (v.w - sub) * mul;
    dst[i] = v;
  }
}
}  // namespace

extern "C" int bbd_gather_pairs(const float* pool, const int32_t* idx_a, const int32_t* idx_b, float* out, int R, long chw,
                                double sub, double mul, void* stream) {
  if (!pool || !idx_a || !idx_b || !out || R <= 0 || R > 32767 || chw <= 0 || (chw & 3) || ((uintptr_t)pool & 15) ||
      ((uintptr_t)out & 15))
    return BBD_E_BADARG;
  const long chw4 = chw / 4;
  long bx = (chw4 + 255) / 256;
  if (bx > 64) bx = 64;                                       // 64 x 2R workgroups: the launch fills the chip from R = 8 on
  hipLaunchKernelGGL(gather_pairs_kernel, dim3((unsigned)bx, (unsigned)(2 * R)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f4*>(pool), idx_a, idx_b, reinterpret_cast<f4*>(out), chw4, (float)sub, (float)mul);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int bbd_stream_copy(const float* src, float* dst, long n_floats, int unroll, void* stream) {
  if (!src || !dst || n_floats <= 0 || (n_floats & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return BBD_E_BADARG;
  const long n4 = n_floats / 4;
  const int U = unroll >= 8 ? 8 : (unroll >= 4 ? 4 : (unroll >= 2 ? 2 : 1));
  long blocks = (n4 + 256L * U - 1) / (256L * U);
  if (blocks > 256 * 32) blocks = 256 * 32;
  const f4* s4 = reinterpret_cast<const f4*>(src);
  f4* d4 = reinterpret_cast<f4*>(dst);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 g((unsigned)blocks), b(256);
  if (U == 8) hipLaunchKernelGGL(stream_copy_kernel<8>, g, b, 0, st, s4, d4, n4);
  else if (U == 4) hipLaunchKernelGGL(stream_copy_kernel<4>, g, b, 0, st, s4, d4, n4);
  else if (U == 2) hipLaunchKernelGGL(stream_copy_kernel<2>, g, b, 0, st, s4, d4, n4);
  else hipLaunchKernelGGL(stream_copy_kernel<1>, g, b, 0, st, s4, d4, n4);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
