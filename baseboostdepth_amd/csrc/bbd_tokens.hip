// bbd_tokens.hip - residual + stochastic depth + LayerNorm over token-layout activations [rows = B*N, C], one pass
// each way (MonoViT's MHCABlock, reference networksvit/mpvit.py:397-440: `x = x + drop_path(branch); z = norm(x)`).
//
// As eager ops that line is a broadcast multiply, an add and a LayerNorm forward (7 passes over [rows, C]), and on the
// way back LayerNorm's input gradient, its two-kernel gamma/beta gradient, an add and a multiply (10 passes).  Here:
//   forward   y = x + branch * mask[b];  z = (y - mean) * rstd * w + bias          reads x, branch; writes y, z (+ 8 B/row)
//   backward  g = gy + LayerNorm'(gz);  gx = g;  gbranch = g * mask[b];  gw, gbias   reads gz, gy, y; writes gx, gbranch
// HBM-streaming kernels: a row is held in registers by LPR lanes (16 / 32 / 64: rows of 64, 128 or more channels; four
// channels per lane and 16-byte accesses), statistics by butterfly over those lanes, 64 / LPR rows per wave.  The
// column sums of the parameter gradients stay in registers over all rows of a workgroup and leave as one partial row
// per workgroup; a second tiny kernel adds the partial rows in a fixed order (deterministic, no atomics).
// `branch` / `mask` / `gy` may be null (plain LayerNorm, no stochastic depth, no direct gradient); `weight == null`
// skips the normalisation (residual only).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_PARTIAL_ROWS = 1024;

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the LPR lanes that hold one row; every one of them gets the total.  16 lanes: four DPP steps (lane pairs,
// quad halves, then the mirror images inside 8 and 16 lanes - the operands are already uniform below each step);
// 32 / 64 lanes add one / two cross-row exchanges.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
  v += dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp<0x141>(v);     // row_half_mirror
  v += dpp<0x140>(v);     // row_mirror
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPR, int VPL>
__global__ __launch_bounds__(NT) void token_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ branch,
                                                          const float* __restrict__ mask, const float* __restrict__ weight,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ z, float* __restrict__ stats, int rows, int N,
                                                          int C, float eps) {
  constexpr int RPW = 64 / LPR;                     // rows per wave
  const int lane = threadIdx.x & 63, sub = lane % LPR, rw = lane / LPR;
  const int wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  float4 w4[VPL], b4[VPL];
  if (weight != nullptr) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c4 = sub + i * LPR;
      w4[i] = c4 < C4 ? ld4(weight + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      b4[i] = c4 < C4 ? ld4(bias + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float inv_c = 1.0f / (float)C;
  for (int r0 = (blockIdx.x * (NT / 64) + wave) * RPW; r0 < rows; r0 += gridDim.x * (NT / 64) * RPW) {
    const int r = r0 + rw;
    const bool live = r < rows;
    const float m = (live && mask != nullptr) ? mask[r / N] : 1.0f;
    float4 v[VPL];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c4 = sub + i * LPR;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && c4 < C4) {
        const size_t o = (size_t)r * C + 4 * c4;
        v[i] = ld4(x + o);
        if (branch != nullptr) {
          const float4 q = ld4(branch + o);
          v[i].x += q.x * m; v[i].y += q.y * m; v[i].z += q.z * m; v[i].w += q.w * m;
          st4(y + o, v[i]);
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    if (weight == nullptr) continue;
    const float mean = row_sum<LPR>(s) * inv_c;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c4 = sub + i * LPR;
      if (c4 < C4) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = rsqrtf(row_sum<LPR>(q) * inv_c + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c4 = sub + i * LPR;
      if (live && c4 < C4) {
        float4 o4;
        o4.x = (v[i].x - mean) * rstd * w4[i].x + b4[i].x;
        o4.y = (v[i].y - mean) * rstd * w4[i].y + b4[i].y;
        o4.z = (v[i].z - mean) * rstd * w4[i].z + b4[i].z;
        o4.w = (v[i].w - mean) * rstd * w4[i].w + b4[i].w;
        st4(z + (size_t)r * C + 4 * c4, o4);
      }
    }
    if (live && sub == 0) {
      stats[2 * (size_t)r] = mean;
      stats[2 * (size_t)r + 1] = rstd;
    }
  }
}

template <int LPR, int VPL>
__global__ __launch_bounds__(NT) void token_ln_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ gy,
                                                          const float* __restrict__ y, const float* __restrict__ stats,
                                                          const float* __restrict__ weight, const float* __restrict__ mask,
                                                          float* __restrict__ gx, float* __restrict__ gbranch,
                                                          float* __restrict__ partial, int rows, int N, int C) {
  constexpr int RPW = 64 / LPR;
  __shared__ float4 s_acc[NT / 64][2][VPL * LPR];
  const int lane = threadIdx.x & 63, sub = lane % LPR, rw = lane / LPR;
  const int wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  float4 w4[VPL], aw[VPL], ab[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c4 = sub + i * LPR;
    w4[i] = c4 < C4 ? ld4(weight + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    aw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_c = 1.0f / (float)C;
  for (int r0 = (blockIdx.x * (NT / 64) + wave) * RPW; r0 < rows; r0 += gridDim.x * (NT / 64) * RPW) {
    const int r = r0 + rw;
    const bool live = r < rows;
    const float mean = live ? stats[2 * (size_t)r] : 0.0f, rstd = live ? stats[2 * (size_t)r + 1] : 0.0f;
    const float m = (live && mask != nullptr) ? mask[r / N] : 1.0f;
    float4 g[VPL], xh[VPL];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c4 = sub + i * LPR;
      g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && c4 < C4) {
        const size_t o = (size_t)r * C + 4 * c4;
        const float4 gq = ld4(gz + o), yq = ld4(y + o);
        xh[i].x = (yq.x - mean) * rstd; xh[i].y = (yq.y - mean) * rstd;
        xh[i].z = (yq.z - mean) * rstd; xh[i].w = (yq.w - mean) * rstd;
        aw[i].x += gq.x * xh[i].x; aw[i].y += gq.y * xh[i].y; aw[i].z += gq.z * xh[i].z; aw[i].w += gq.w * xh[i].w;
        ab[i].x += gq.x; ab[i].y += gq.y; ab[i].z += gq.z; ab[i].w += gq.w;
        g[i].x = gq.x * w4[i].x; g[i].y = gq.y * w4[i].y; g[i].z = gq.z * w4[i].z; g[i].w = gq.w * w4[i].w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
    }
    s1 = row_sum<LPR>(s1) * inv_c;
    s2 = row_sum<LPR>(s2) * inv_c;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c4 = sub + i * LPR;
      if (live && c4 < C4) {
        const size_t o = (size_t)r * C + 4 * c4;
        float4 t;
        t.x = rstd * (g[i].x - s1 - xh[i].x * s2);
        t.y = rstd * (g[i].y - s1 - xh[i].y * s2);
        t.z = rstd * (g[i].z - s1 - xh[i].z * s2);
        t.w = rstd * (g[i].w - s1 - xh[i].w * s2);
        if (gy != nullptr) {
          const float4 d = ld4(gy + o);
          t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
        }
        st4(gx + o, t);
        if (gbranch != nullptr) st4(gbranch + o, make_float4(t.x * m, t.y * m, t.z * m, t.w * m));
      }
    }
  }
  // column sums of this workgroup's rows: over the row groups of a wave (butterfly), then over the waves (LDS, fixed order)
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      aw[i].x += __shfl_xor(aw[i].x, off, 64); aw[i].y += __shfl_xor(aw[i].y, off, 64);
      aw[i].z += __shfl_xor(aw[i].z, off, 64); aw[i].w += __shfl_xor(aw[i].w, off, 64);
      ab[i].x += __shfl_xor(ab[i].x, off, 64); ab[i].y += __shfl_xor(ab[i].y, off, 64);
      ab[i].z += __shfl_xor(ab[i].z, off, 64); ab[i].w += __shfl_xor(ab[i].w, off, 64);
    }
    if (rw == 0) {
      s_acc[wave][0][i * LPR + sub] = aw[i];
      s_acc[wave][1][i * LPR + sub] = ab[i];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * VPL * LPR; j += NT) {
    const int which = j / (VPL * LPR), c4 = j % (VPL * LPR);
    if (c4 >= C4) continue;
    float4 t = s_acc[0][which][c4];
#pragma unroll
    for (int w8 = 1; w8 < NT / 64; ++w8) {
      const float4 u = s_acc[w8][which][c4];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    st4(partial + ((size_t)blockIdx.x * 2 + which) * C + 4 * c4, t);
  }
}

// gw[c] = sum over partial rows p of partial[p][0][c], gb[c] likewise from [p][1][c]: a block owns 64 columns, its 16
// row lanes take every 16th partial row (independent loads in flight), LDS combine in a fixed order
constexpr int PG_ROWS = 16;
__global__ __launch_bounds__(64 * PG_ROWS) void token_ln_param_grad_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                                          float* __restrict__ gb, int prow, int C) {
  __shared__ float sh[PG_ROWS][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  float s0 = 0.0f, s1 = 0.0f;
  if (j < 2 * C) {
    int p = rl;
    for (; p + PG_ROWS < prow; p += 2 * PG_ROWS) {
      s0 += partial[(size_t)p * 2 * C + j];
      s1 += partial[(size_t)(p + PG_ROWS) * 2 * C + j];
    }
    if (p < prow) s0 += partial[(size_t)p * 2 * C + j];
  }
  sh[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && j < 2 * C) {
    float s = sh[0][cl];
#pragma unroll
    for (int k = 1; k < PG_ROWS; ++k) s += sh[k][cl];
    if (j < C) gw[j] = s; else gb[j - C] = s;
  }
}

// Column sums of a [rows, C] matrix (the bias gradient of a token-parallel Linear layer): a thread owns one float4 column,
// CL column lanes x 256/CL row lanes per workgroup, four independent rows in flight per thread; the row lanes combine
// through LDS in a fixed order and the workgroup leaves one partial row; colsum_final adds the partial rows.
template <int CL>
__global__ __launch_bounds__(NT) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, long rows,
                                                            int C) {
  constexpr int RLN = NT / CL;
  __shared__ float4 sh[RLN][CL];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c4 = blockIdx.x * CL + cl, C4 = C >> 2;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (c4 < C4) {
    const long stride = (long)gridDim.y * RLN;
    long r = (long)blockIdx.y * RLN + rl;
    for (; r + 3 * stride < rows; r += 4 * stride) {
      const float4 v0 = ld4(x + r * C + 4 * c4), v1 = ld4(x + (r + stride) * C + 4 * c4);
      const float4 v2 = ld4(x + (r + 2 * stride) * C + 4 * c4), v3 = ld4(x + (r + 3 * stride) * C + 4 * c4);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; r < rows; r += stride) {
      const float4 v0 = ld4(x + r * C + 4 * c4);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  sh[rl][cl] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                           (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (rl != 0 || c4 >= C4) return;
  float4 t = sh[0][cl];
#pragma unroll
  for (int q = 1; q < RLN; ++q) {
    const float4 u = sh[q][cl];
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  st4(partial + (size_t)blockIdx.y * C + 4 * c4, t);
}

__global__ __launch_bounds__(64 * PG_ROWS) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                   int prow, int C) {
  __shared__ float sh[PG_ROWS][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  float s0 = 0.0f, s1 = 0.0f;
  if (j < C) {
    int p = rl;
    for (; p + PG_ROWS < prow; p += 2 * PG_ROWS) {
      s0 += partial[(size_t)p * C + j];
      s1 += partial[(size_t)(p + PG_ROWS) * C + j];
    }
    if (p < prow) s0 += partial[(size_t)p * C + j];
  }
  sh[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && j < C) {
    float s = sh[0][cl];
#pragma unroll
    for (int k = 1; k < PG_ROWS; ++k) s += sh[k][cl];
    out[j] = s;
  }
}

int colsum_rows(long rows, int C) {
  const int C4 = C >> 2, cl = C4 <= 16 ? 16 : (C4 <= 32 ? 32 : 64);
  const int tiles = (C4 + cl - 1) / cl, rln = NT / cl;
  long want = (1024 + tiles - 1) / tiles;                  // ~1 024 workgroups in all
  const long most = (rows + 4L * rln - 1) / (4L * rln);    // at least four rows per thread
  if (want > most) want = most;
  if (want > 256) want = 256;
  return (int)(want < 1 ? 1 : want);
}

int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int lanes_per_row(int C) { return C <= 64 ? 16 : (C <= 128 ? 32 : 64); }

int fwd_grid(int rows, int C) {
  const int rpw = 64 / lanes_per_row(C);
  const long wgs = ((long)rows + (NT / 64) * rpw - 1) / ((NT / 64) * rpw);
  return (int)(wgs < 4096 ? wgs : 4096);
}
int bwd_grid(int rows, int C) {
  // one partial row per workgroup: 1 024 keep the big stage-1 activations (24 MB per tensor) HBM-bound, 256 are enough for
  // the rest and keep the fixed-order column sum of the partial rows short
  const int g = fwd_grid(rows, C), cap = ((long)rows * C > (4L << 20)) ? MAX_PARTIAL_ROWS : 256;
  return g < cap ? g : cap;
}

}  // namespace

extern "C" {

int bbd_token_ln_supported(int C) { return C > 0 && (C % 4) == 0 && C <= 1024; }

long bbd_token_ln_scratch_floats(int rows, int C) { return (long)bwd_grid(rows, C) * 2 * C; }

long bbd_colsum_scratch_floats(long rows, int C) { return (long)colsum_rows(rows, C) * C; }

int bbd_colsum(const float* x, float* partial, float* out, long rows, int C, void* stream) {
  if (!x || !partial || !out || rows < 0 || C <= 0 || (C % 4) != 0) return BBD_E_BADARG;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int C4 = C >> 2, prow = colsum_rows(rows, C);
  if (C4 <= 16) hipLaunchKernelGGL(colsum_partial_kernel<16>, dim3((unsigned)((C4 + 15) / 16), (unsigned)prow), dim3(NT), 0, st, x, partial, rows, C);
  else if (C4 <= 32) hipLaunchKernelGGL(colsum_partial_kernel<32>, dim3((unsigned)((C4 + 31) / 32), (unsigned)prow), dim3(NT), 0, st, x, partial, rows, C);
  else hipLaunchKernelGGL(colsum_partial_kernel<64>, dim3((unsigned)((C4 + 63) / 64), (unsigned)prow), dim3(NT), 0, st, x, partial, rows, C);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64 * PG_ROWS), 0, st, partial, out, prow, C);
  return launch_status();
}

#define BBD_TOKEN_DISPATCH(KERNEL, GRID, ...)                                                                         \
  do {                                                                                                                \
    const hipStream_t st = static_cast<hipStream_t>(stream);                                                          \
    if (C <= 64) hipLaunchKernelGGL((KERNEL<16, 1>), dim3(GRID), dim3(NT), 0, st, __VA_ARGS__);                        \
    else if (C <= 128) hipLaunchKernelGGL((KERNEL<32, 1>), dim3(GRID), dim3(NT), 0, st, __VA_ARGS__);                  \
    else if (C <= 256) hipLaunchKernelGGL((KERNEL<64, 1>), dim3(GRID), dim3(NT), 0, st, __VA_ARGS__);                  \
    else if (C <= 512) hipLaunchKernelGGL((KERNEL<64, 2>), dim3(GRID), dim3(NT), 0, st, __VA_ARGS__);                  \
    else hipLaunchKernelGGL((KERNEL<64, 4>), dim3(GRID), dim3(NT), 0, st, __VA_ARGS__);                                \
  } while (0)

int bbd_token_ln_fwd(const float* x, const float* branch, const float* mask, const float* weight, const float* bias,
                     float* y, float* z, float* stats, int rows, int N, int C, double eps, void* stream) {
  if (!x || rows < 0 || N <= 0 || !bbd_token_ln_supported(C)) return BBD_E_BADARG;
  if ((branch && !y) || (weight && (!bias || !z || !stats)) || (!branch && !weight)) return BBD_E_BADARG;
  if (rows == 0) return 0;
  BBD_TOKEN_DISPATCH(token_ln_fwd_kernel, (unsigned)fwd_grid(rows, C), x, branch, mask, weight, bias, y, z, stats, rows, N, C,
                     (float)eps);
  return launch_status();
}

int bbd_token_ln_bwd(const float* grad_z, const float* grad_y, const float* y, const float* stats, const float* weight,
                     const float* mask, float* grad_x, float* grad_branch, float* partial, float* grad_weight,
                     float* grad_bias, int rows, int N, int C, void* stream) {
  if (!grad_z || !y || !stats || !weight || !grad_x || !partial || !grad_weight || !grad_bias) return BBD_E_BADARG;
  if (rows < 0 || N <= 0 || !bbd_token_ln_supported(C)) return BBD_E_BADARG;
  if (rows == 0) return 0;
  const int grid = bwd_grid(rows, C);
  BBD_TOKEN_DISPATCH(token_ln_bwd_kernel, (unsigned)grid, grad_z, grad_y, y, stats, weight, mask, grad_x, grad_branch, partial,
                     rows, N, C);
  hipLaunchKernelGGL(token_ln_param_grad_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(64 * PG_ROWS), 0,
                     static_cast<hipStream_t>(stream), partial, grad_weight, grad_bias, grid, C);
  return launch_status();
}

}  // extern "C"
