// bbd_vit.hip - gfx950 kernels for the MonoViT encoder (BASELINE configs[4]): depth-wise convolution on
// TOKEN-layout activations.
//
// MPViT's position encodings (reference networksvit/mpvit.py:240-330) are depth-wise k x k convolutions
// (k = 3, 5, 7) applied to the token matrix viewed as an image.  This build keeps activations as
// [B, N, C] row-major tokens = the NHWC image [B, H, W, C]; MIOpen has no tuned depth-wise path for that
// layout (it ran naive kernels, and its weight gradient - a CK batched GEMM - took 1.7 ms per call, 54 % of
// the whole MonoViT step).  These are HBM/L2-streaming kernels: lanes run along the channel axis, so every
// tap is one coalesced row segment; a thread produces 4 consecutive pixels along W and slides its k-wide
// window through registers (k*(k+3) loads for 4 outputs instead of 4*k*k).
//
// Inputs / outputs may be channel slices of wider token rows (`row` = floats between consecutive tokens):
// v is read in place from the packed qkv activation, the head groups of ConvRelPosEnc write their slices of
// one output, and no split / cat copies exist.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"

namespace {

constexpr int NT = 256;
constexpr int PW = 4;     // output pixels per thread along W

// y[b,h,w,c] = bias[c] + add_in * x[b,h,w,c] + sum_{i,j} wt[c,i,j] * x[b,h+i-p,w+j-p,c]      (zero padding)
// FLIP reads the taps mirrored: the same kernel is the data gradient (x := dy, no bias).
template <int K, bool FLIP>
__global__ __launch_bounds__(NT) void dwconv_tokens_kernel(const float* __restrict__ x, int x_row,
                                                           const float* __restrict__ wt, const float* __restrict__ bias,
                                                           float* __restrict__ y, int y_row, int B, int H, int W, int C,
                                                           int add_in) {
  constexpr int P = K / 2;
  const int wq = (W + PW - 1) / PW;
  const long total = (long)B * H * wq * C;
  const long id = (long)blockIdx.x * NT + threadIdx.x;
  if (id >= total) return;
  const int c = (int)(id % C);
  long t = id / C;
  const int w0 = (int)(t % wq) * PW;
  t /= wq;
  const int h = (int)(t % H);
  const int b = (int)(t / H);
  float wk[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wk[i] = wt[c * K * K + (FLIP ? K * K - 1 - i : i)];
  float acc[PW];
  const float b0 = bias ? bias[c] : 0.0f;
#pragma unroll
  for (int q = 0; q < PW; ++q) acc[q] = b0;
  const float* xb = x + (long)b * H * W * x_row + c;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int hy = h + i - P;
    if (hy < 0 || hy >= H) continue;
    const float* xr = xb + (long)hy * W * x_row;
    float win[PW + K - 1];
#pragma unroll
    for (int j = 0; j < PW + K - 1; ++j) {
      const int wx = w0 + j - P;
      win[j] = (wx >= 0 && wx < W) ? xr[(long)wx * x_row] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < PW; ++q)
#pragma unroll
      for (int j = 0; j < K; ++j) acc[q] = fmaf(wk[i * K + j], win[q + j], acc[q]);
    if (add_in && i == P) {
#pragma unroll
      for (int q = 0; q < PW; ++q) acc[q] += win[q + P];
    }
  }
  float* yo = y + ((long)(b * H + h) * W + w0) * y_row + c;
#pragma unroll
  for (int q = 0; q < PW; ++q)
    if (w0 + q < W) yo[(long)q * y_row] = acc[q];
}

// Weight / bias gradient, stage 1: one thread per (image row (b,h), channel) accumulates its k*k + 1 partial
// sums over the W pixels of that row; stage 2 adds the B*H partials of every (channel, tap) in fixed order.
template <int K>
__global__ __launch_bounds__(NT) void dwconv_tokens_wgrad_kernel(const float* __restrict__ x, int x_row,
                                                                 const float* __restrict__ dy, int dy_row,
                                                                 float* __restrict__ partial, int B, int H, int W, int C) {
  constexpr int P = K / 2;
  const long id = (long)blockIdx.x * NT + threadIdx.x;
  if (id >= (long)B * H * C) return;
  const int c = (int)(id % C);
  const int bh = (int)(id / C);
  const int h = bh % H, b = bh / H;
  float acc[K * K + 1];
#pragma unroll
  for (int i = 0; i <= K * K; ++i) acc[i] = 0.0f;
  const float* xb = x + (long)b * H * W * x_row + c;
  const float* dr = dy + (long)bh * W * dy_row + c;
  for (int w = 0; w < W; ++w) {
    const float g = dr[(long)w * dy_row];
    acc[K * K] += g;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int hy = h + i - P;
      if (hy < 0 || hy >= H) continue;
      const float* xr = xb + (long)hy * W * x_row;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int wx = w + j - P;
        if (wx >= 0 && wx < W) acc[i * K + j] = fmaf(g, xr[(long)wx * x_row], acc[i * K + j]);
      }
    }
  }
  float* po = partial + ((long)bh * C + c) * (K * K + 1);
#pragma unroll
  for (int i = 0; i <= K * K; ++i) po[i] = acc[i];
}

__global__ __launch_bounds__(NT) void dwconv_tokens_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                        float* __restrict__ dw, float* __restrict__ dbias,
                                                                        int rows, int C, int kk) {
  const int id = blockIdx.x * NT + threadIdx.x;
  if (id >= C * (kk + 1)) return;
  const int c = id / (kk + 1), t = id - c * (kk + 1);
  double s = 0.0;
  for (int r = 0; r < rows; ++r) s += (double)partial[((long)r * C + c) * (kk + 1) + t];
  if (t < kk) dw[c * kk + t] = (float)s;
  else if (dbias) dbias[c] = (float)s;
}

int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int bbd_dwconv_tokens_fwd(const float* x, int x_row, const float* weight, const float* bias, float* y, int y_row,
                          int B, int H, int W, int C, int k, int add_input, int flip, void* stream) {
  if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || x_row < C || y_row < C) return BBD_E_BADARG;
  if (k != 3 && k != 5 && k != 7) return BBD_E_BADARG;
  const long total = (long)B * H * ((W + PW - 1) / PW) * C;
  const dim3 grid((unsigned)((total + NT - 1) / NT));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define BBD_DW(K, F) hipLaunchKernelGGL((dwconv_tokens_kernel<K, F>), grid, dim3(NT), 0, st, x, x_row, weight, bias, y, y_row, B, H, W, C, add_input)
  if (k == 3) { if (flip) BBD_DW(3, true); else BBD_DW(3, false); }
  else if (k == 5) { if (flip) BBD_DW(5, true); else BBD_DW(5, false); }
  else { if (flip) BBD_DW(7, true); else BBD_DW(7, false); }
#undef BBD_DW
  return launch_status();
}

int bbd_dwconv_tokens_wgrad(const float* x, int x_row, const float* grad_y, int gy_row, float* partial,
                            float* grad_weight, float* grad_bias, int B, int H, int W, int C, int k, void* stream) {
  if (!x || !grad_y || !partial || !grad_weight || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BBD_E_BADARG;
  if (k != 3 && k != 5 && k != 7) return BBD_E_BADARG;
  const long total = (long)B * H * C;
  const dim3 grid((unsigned)((total + NT - 1) / NT));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (k == 3) hipLaunchKernelGGL(dwconv_tokens_wgrad_kernel<3>, grid, dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial, B, H, W, C);
  else if (k == 5) hipLaunchKernelGGL(dwconv_tokens_wgrad_kernel<5>, grid, dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial, B, H, W, C);
  else hipLaunchKernelGGL(dwconv_tokens_wgrad_kernel<7>, grid, dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial, B, H, W, C);
  const int n = C * (k * k + 1);
  hipLaunchKernelGGL(dwconv_tokens_wgrad_reduce_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, partial,
                     grad_weight, grad_bias, B * H, C, k * k);
  return launch_status();
}

}  // extern "C"
