// FETCH_SIZE calibration on gfx950 (MI355X_MICROARCH.md, HBM: the counter reports half the bytes of a 16-byte-
// per-lane streaming read; "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Three kernels stream the SAME 1 GiB buffer (4x the Infinity Cache) once, with 4-, 8- and
// 16-byte loads per lane, plus the hot path's own pattern: 8-byte loads at a per-lane 2-D gather address
// (two texels of a row, rows `pitch` floats apart).  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- ./fetch_calib
// and divide the known bytes (printed) by FETCH_SIZE * 1024.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ __launch_bounds__(256) void stream_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = in[i];
    acc += reinterpret_cast<const float*>(&v)[0];
  }
  if (acc == 12345.678f) out[0] = acc;
}

// the warp kernels' gather: lane -> (row y, column x) of a [rows, pitch] float image, one 8-byte load of the
// texel pair (x, x+1); consecutive lanes take consecutive x (as adjacent target pixels do), every element of
// the image is covered exactly once per pair
__global__ __launch_bounds__(256) void gather8_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                      int pitch) {
  float acc = 0.0f;
  const int pairs = pitch / 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * pairs; i += (size_t)gridDim.x * 256) {
    const int y = (int)(i / pairs), x = (int)(i % pairs) * 2;
    const float2 v = *reinterpret_cast<const float2*>(in + (size_t)y * pitch + x);
    acc += v.x + v.y;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float *buf, *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, bytes);
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_kernel<float>, grid, block, 0, 0, buf, out, bytes / 4);
    hipLaunchKernelGGL(stream_kernel<float2>, grid, block, 0, 0, reinterpret_cast<const float2*>(buf), out, bytes / 8);
    hipLaunchKernelGGL(stream_kernel<float4>, grid, block, 0, 0, reinterpret_cast<const float4*>(buf), out, bytes / 16);
    hipLaunchKernelGGL(gather8_kernel, grid, block, 0, 0, buf, out, (int)(bytes / 4 / 640), 640);
  }
  (void)hipDeviceSynchronize();
  printf("known bytes per launch: %zu (stream kernels), %zu (gather8: %d rows x 640 floats)\n", bytes,
         (size_t)(bytes / 4 / 640) * 640 * 4, (int)(bytes / 4 / 640));
  return 0;
}
