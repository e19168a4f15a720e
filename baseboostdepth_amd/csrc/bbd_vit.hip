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

// y[b,h,w,c] = bias[c] + (add_in & 1) * x[b,h,w,c] + sum_{i,j} wt[c,i,j] * x[b,h+i-p,w+j-p,c]      (zero padding)
// (add_in & 2: added to what y already holds - a data gradient landing in a slice that has another contribution)
// FLIP reads the taps mirrored: the same kernel is the data gradient (x := dy, no bias).
template <int K, bool FLIP>
__device__ __forceinline__ void dwconv_tokens_body(const float* __restrict__ x, int x_row, const float* __restrict__ wt,
                                                   const float* __restrict__ bias, float* __restrict__ y, int y_row, int B,
                                                   int H, int W, int C, int add_in, long block) {
  constexpr int P = K / 2;
  const int wq = (W + PW - 1) / PW;
  const long total = (long)B * H * wq * C;
  const long id = block * NT + threadIdx.x;
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
    if ((add_in & 1) && i == P) {
#pragma unroll
      for (int q = 0; q < PW; ++q) acc[q] += win[q + P];
    }
  }
  float* yo = y + ((long)(b * H + h) * W + w0) * y_row + c;
#pragma unroll
  for (int q = 0; q < PW; ++q)
    if (w0 + q < W) yo[(long)q * y_row] = (add_in & 2) ? yo[(long)q * y_row] + acc[q] : acc[q];     // bit 1: accumulate into y
}

template <int K, bool FLIP>
__global__ __launch_bounds__(NT) void dwconv_tokens_kernel(const float* __restrict__ x, int x_row,
                                                           const float* __restrict__ wt, const float* __restrict__ bias,
                                                           float* __restrict__ y, int y_row, int B, int H, int W, int C,
                                                           int add_in) {
  dwconv_tokens_body<K, FLIP>(x, x_row, wt, bias, y, y_row, B, H, W, C, add_in, (long)blockIdx.x);
}

// Several channel groups with their own window size in ONE launch (MPViT's ConvRelPosEnc: head groups with windows 3 / 5 /
// 7): a workgroup belongs to one group (block ranges of the 1-D grid), so there is no divergence inside a wave.
constexpr int DW_MAX_GROUPS = 4;
struct DwGroups {
  int n;
  int c0[DW_MAX_GROUPS], cn[DW_MAX_GROUPS], k[DW_MAX_GROUPS];
  const float* w[DW_MAX_GROUPS];
  const float* b[DW_MAX_GROUPS];
  float* gw[DW_MAX_GROUPS];
  float* gb[DW_MAX_GROUPS];
  unsigned block0[DW_MAX_GROUPS + 1];      // first block of every group in the launch's x dimension
  unsigned fblock0[DW_MAX_GROUPS + 1];     // the same for the weight gradient's column-sum launch
  long part0[DW_MAX_GROUPS];               // weight gradient: float offset of the group's partial rows
};
__device__ __forceinline__ int dw_group_of(const DwGroups& g, unsigned block) {
  int gi = 0;
#pragma unroll
  for (int i = 1; i < DW_MAX_GROUPS; ++i)
    if (i < g.n && block >= g.block0[i]) gi = i;
  return gi;
}

template <bool FLIP>
__global__ __launch_bounds__(NT) void dwconv_tokens_groups_kernel(const float* __restrict__ x, int x_row, float* __restrict__ y,
                                                                  int y_row, DwGroups g, int B, int H, int W, int add_in) {
  const int gi = dw_group_of(g, blockIdx.x);
  const long block = (long)(blockIdx.x - g.block0[gi]);
  const float* xs = x + g.c0[gi];
  float* ys = y + g.c0[gi];
  switch (g.k[gi]) {
    case 3: dwconv_tokens_body<3, FLIP>(xs, x_row, g.w[gi], g.b[gi], ys, y_row, B, H, W, g.cn[gi], add_in, block); break;
    case 5: dwconv_tokens_body<5, FLIP>(xs, x_row, g.w[gi], g.b[gi], ys, y_row, B, H, W, g.cn[gi], add_in, block); break;
    default: dwconv_tokens_body<7, FLIP>(xs, x_row, g.w[gi], g.b[gi], ys, y_row, B, H, W, g.cn[gi], add_in, block); break;
  }
}

// Weight / bias gradient.  A workgroup = 64 channels x 4 segment lanes: a thread walks row segments of WCH pixels
// (seg = blockIdx.y * 4 + lane, stride 4 * gridDim.y), sliding the k-wide window of x through registers, and keeps its
// k*k + 1 sums in registers over ALL its segments; the four segment lanes combine through LDS in a fixed order and the
// workgroup writes ONE partial row - the partial matrix is [gridDim.y <= 256, C*(k*k+1)] instead of one row per
// segment run (round 2: 147 MB for MPViT-small's 7x7 layer, plus a column-sum stage to shrink it).  Second launch: the
// column sums of those rows in fp64, 16 row lanes per column, fixed order - deterministic.
constexpr int WCH = 8;
constexpr int WG_SEG_LANES = NT / 64;
constexpr int WGRAD_MAX_ROWS = 256;
template <int K>
__device__ __forceinline__ void dwconv_tokens_wgrad_body(const float* __restrict__ x, int x_row, const float* __restrict__ dy,
                                                         int dy_row, float* __restrict__ partial, int B, int H, int W, int C,
                                                         int tile, float* sh_raw) {
  constexpr int P = K / 2;
  float (*sh)[K * K + 1][64] = reinterpret_cast<float (*)[K * K + 1][64]>(sh_raw);      // [WG_SEG_LANES - 1][K*K+1][64]
  const int wseg = (W + WCH - 1) / WCH;
  const long segs = (long)B * H * wseg;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = tile * 64 + cl;
  const bool live = c < C;
  float acc[K * K + 1];
#pragma unroll
  for (int i = 0; i <= K * K; ++i) acc[i] = 0.0f;
  if (live)
  for (long seg = (long)blockIdx.y * WG_SEG_LANES + rl; seg < segs; seg += (long)gridDim.y * WG_SEG_LANES) {
  const int w0 = (int)(seg % wseg) * WCH;
  const int bh = (int)(seg / wseg);
  const int h = bh % H, b = bh / H;
  float g[WCH];
  const float* dr = dy + ((long)bh * W + w0) * dy_row + c;
#pragma unroll
  for (int q = 0; q < WCH; ++q) {
    g[q] = (w0 + q < W) ? dr[(long)q * dy_row] : 0.0f;
    acc[K * K] += g[q];
  }
  const float* xb = x + (long)b * H * W * x_row + c;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int hy = h + i - P;
    if (hy < 0 || hy >= H) continue;
    const float* xr = xb + (long)hy * W * x_row;
    float win[WCH + K - 1];
#pragma unroll
    for (int j = 0; j < WCH + K - 1; ++j) {
      const int wx = w0 + j - P;
      win[j] = (wx >= 0 && wx < W) ? xr[(long)wx * x_row] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
      for (int q = 0; q < WCH; ++q) acc[i * K + j] = fmaf(g[q], win[q + j], acc[i * K + j]);
  }
  }
  if (rl > 0) {
#pragma unroll
    for (int i = 0; i <= K * K; ++i) sh[rl - 1][i][cl] = acc[i];
  }
  __syncthreads();
  if (rl != 0 || !live) return;
  float* po = partial + ((long)blockIdx.y * C + c) * (K * K + 1);
#pragma unroll
  for (int i = 0; i <= K * K; ++i) {
    float t = acc[i];
#pragma unroll
    for (int q = 0; q < WG_SEG_LANES - 1; ++q) t += sh[q][i][cl];
    po[i] = t;
  }
}

template <int K>
__global__ __launch_bounds__(NT) void dwconv_tokens_wgrad_kernel(const float* __restrict__ x, int x_row,
                                                                 const float* __restrict__ dy, int dy_row,
                                                                 float* __restrict__ partial, int B, int H, int W, int C) {
  __shared__ float sh[(WG_SEG_LANES - 1) * (K * K + 1) * 64];
  dwconv_tokens_wgrad_body<K>(x, x_row, dy, dy_row, partial, B, H, W, C, (int)blockIdx.x, sh);
}

__global__ __launch_bounds__(NT) void dwconv_tokens_wgrad_groups_kernel(const float* __restrict__ x, int x_row,
                                                                        const float* __restrict__ dy, int dy_row,
                                                                        float* __restrict__ partial, DwGroups g, int B, int H,
                                                                        int W) {
  __shared__ float sh[(WG_SEG_LANES - 1) * (7 * 7 + 1) * 64];
  const int gi = dw_group_of(g, blockIdx.x);
  const int tile = (int)(blockIdx.x - g.block0[gi]);
  const float* xs = x + g.c0[gi];
  const float* ds = dy + g.c0[gi];
  float* ps = partial + g.part0[gi];
  switch (g.k[gi]) {
    case 3: dwconv_tokens_wgrad_body<3>(xs, x_row, ds, dy_row, ps, B, H, W, g.cn[gi], tile, sh); break;
    case 5: dwconv_tokens_wgrad_body<5>(xs, x_row, ds, dy_row, ps, B, H, W, g.cn[gi], tile, sh); break;
    default: dwconv_tokens_wgrad_body<7>(xs, x_row, ds, dy_row, ps, B, H, W, g.cn[gi], tile, sh); break;
  }
}

// column sums with WRL row lanes per column: block = 64 columns x WRL rows, fixed-order combine through LDS
constexpr int RL = NT / 64;        // row lanes of the 256-thread combine kernels below
constexpr int WRL = 16;            // row lanes of the weight-gradient column sum (1 024-thread blocks)
__device__ __forceinline__ void dwconv_tokens_wgrad_final_body(const float* __restrict__ partial, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int rows, int C, int kk,
                                                               int accumulate, int block) {
  __shared__ double sh[WRL][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int id = block * 64 + cl;
  const bool live = id < C * (kk + 1);
  double s = 0.0;
  if (live)
    for (int r = rl; r < rows; r += WRL) s += (double)partial[(long)r * C * (kk + 1) + id];
  sh[rl][cl] = s;
  __syncthreads();
  if (!live || rl != 0) return;
  double t = sh[0][cl];
#pragma unroll
  for (int q = 1; q < WRL; ++q) t += sh[q][cl];
  const int c = id / (kk + 1), tap = id - c * (kk + 1);
  // `accumulate`: added to what the gradient buffers hold (a parameter shared by several layers: the layers' launches
  // run one after the other on one stream, in backward order, so the sum has a fixed order)
  if (tap < kk) dw[c * kk + tap] = (accumulate ? dw[c * kk + tap] : 0.0f) + (float)t;
  else if (dbias) dbias[c] = (accumulate ? dbias[c] : 0.0f) + (float)t;
}

__global__ __launch_bounds__(64 * WRL) void dwconv_tokens_wgrad_final_kernel(const float* __restrict__ partial,
                                                                           float* __restrict__ dw, float* __restrict__ dbias,
                                                                           int rows, int C, int kk, int accumulate) {
  dwconv_tokens_wgrad_final_body(partial, dw, dbias, rows, C, kk, accumulate, (int)blockIdx.x);
}

__global__ __launch_bounds__(64 * WRL) void dwconv_tokens_wgrad_final_groups_kernel(const float* __restrict__ partial, DwGroups g,
                                                                                  int rows, int accumulate) {
  int gi = 0;
#pragma unroll
  for (int i = 1; i < DW_MAX_GROUPS; ++i)
    if (i < g.n && blockIdx.x >= g.fblock0[i]) gi = i;
  dwconv_tokens_wgrad_final_body(partial + g.part0[gi], g.gw[gi], g.gb[gi], rows, g.cn[gi], g.k[gi] * g.k[gi], accumulate,
                                 (int)(blockIdx.x - g.fblock0[gi]));
}

// partial rows of the weight-gradient launch: enough workgroups to fill the chip (~768 with the channel tiles), at most
// WGRAD_MAX_ROWS, never more than there are segment quads
static int wgrad_rows(long segs, int C) {
  const int tiles = (C + 63) / 64;
  long want = (768 + tiles - 1) / tiles;
  const long quads = (segs + WG_SEG_LANES - 1) / WG_SEG_LANES;
  if (want > quads) want = quads;
  if (want > WGRAD_MAX_ROWS) want = WGRAD_MAX_ROWS;
  return (int)(want < 1 ? 1 : want);
}

// ------------------------------------------------------------------------------------------------------
// Factorised attention of MPViT (reference networksvit/mpvit.py:333-394), token layout.
//   qkv [B, N, 3, h, Ch] (the Linear's output, read in place: row = 3C floats, q | k | v thirds)
//   ctxs[b,h,kc,vc] = scale * sum_n softmax_N(k)[b,n,h,kc] * v[b,n,h,vc]                (h * Ch * Ch per image)
//   out[b,n,h,vc]   = sum_kc q[b,n,h,kc] * ctxs[b,h,kc,vc]  +  q[b,n,h,vc] * convv[b,n,h,vc]
// The reference evaluates this as softmax (a [B,h,N,Ch] kernel over the strided N axis: 0.9 ms per call on
// MI355X), two einsum GEMMs with Ch = 8..36, and three element-wise launches; here: column statistics of k
// (online max / sum per token segment), the [Ch x Ch] contexts as per-segment partial sums combined in fixed
// order, and one token-parallel kernel for the output - all reading the packed qkv rows coalesced along C.
// ------------------------------------------------------------------------------------------------------
constexpr int FA_TOK = 16;      // tokens staged per iteration of the context kernel
constexpr int FA_MAXO = 48;     // context entries per thread (C * Ch <= 256 * 48); instantiated for 2 / 8 / 24 / 48

// per-(image, token segment) online softmax statistics of k over the tokens of the segment
__global__ __launch_bounds__(NT) void fa_kstats_kernel(const float* __restrict__ qkv, float* __restrict__ pm,
                                                       float* __restrict__ ps, int N, int C, int seg_tokens) {
  const int seg = blockIdx.x, b = blockIdx.y, nseg = gridDim.x;
  const int n0 = seg * seg_tokens, n1 = min(N, n0 + seg_tokens);
  const float* kb = qkv + ((long)b * N) * 3 * C + C;
  for (int c = threadIdx.x; c < C; c += NT) {
    float m = -INFINITY, sum = 0.0f;
    for (int n = n0; n < n1; ++n) {
      const float v = kb[(long)n * 3 * C + c];
      const float mn = fmaxf(m, v);
      sum = sum * __expf(m - mn) + __expf(v - mn);
      m = mn;
    }
    pm[((long)b * nseg + seg) * C + c] = m;
    ps[((long)b * nseg + seg) * C + c] = sum;
  }
}

__global__ __launch_bounds__(NT) void fa_kstats_combine_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                               float* __restrict__ kmax, float* __restrict__ krsum,
                                                               int nseg, int C, int total) {
  // 64 (sample, channel) columns x RL segment lanes per workgroup (one thread per column looping over ~90 segments in
  // 3-14 workgroups took 30 us per call); fixed-order combine through LDS
  __shared__ float sh[RL][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int id = blockIdx.x * 64 + cl;
  const bool live = id < total;
  const int b = live ? id / C : 0, c = live ? id - b * C : 0;
  float m = -INFINITY;
  if (live)
    for (int s = rl; s < nseg; s += RL) m = fmaxf(m, pm[((long)b * nseg + s) * C + c]);
  sh[rl][cl] = m;
  __syncthreads();
  m = sh[0][cl];
#pragma unroll
  for (int q = 1; q < RL; ++q) m = fmaxf(m, sh[q][cl]);
  __syncthreads();
  float sum = 0.0f;
  if (live)
    for (int s = rl; s < nseg; s += RL) {
      const float sm = ps[((long)b * nseg + s) * C + c];
      if (sm > 0.0f) sum += sm * __expf(pm[((long)b * nseg + s) * C + c] - m);
    }
  sh[rl][cl] = sum;
  __syncthreads();
  if (!live || rl != 0) return;
  sum = sh[0][cl];
#pragma unroll
  for (int q = 1; q < RL; ++q) sum += sh[q][cl];
  kmax[id] = m;
  krsum[id] = 1.0f / sum;
}
// (token, channel) of element e = threadIdx.x + k * NT of a [tokens, C] tile without a division per element: one division
// at the start, then NT is added in radix C.
struct TokChan {
  int tt, c, q, r;
  __device__ __forceinline__ TokChan(int C) {
    q = NT / C; r = NT - q * C;
    tt = (int)threadIdx.x / C; c = (int)threadIdx.x - tt * C;
  }
  __device__ __forceinline__ void next(int C) {
    c += r; tt += q;
    if (c >= C) { c -= C; tt += 1; }
  }
};

// partial[b, seg, o] = sum over the segment's tokens of A[n, hk(o)] * Bm[n, hv(o)],  o = (h, kc, vc) flattened.
// SOFTMAX: A = exp(k - kmax) * krsum (the softmax over tokens), Bm = v      -> forward contexts
// else   : A = q,                                               Bm = dout   -> their gradient
template <bool SOFTMAX, int MAXO>
__global__ __launch_bounds__(NT) void fa_context_kernel(const float* __restrict__ qkv, const float* __restrict__ bm_src,
                                                        int bm_row, const float* __restrict__ kmax,
                                                        const float* __restrict__ krsum, float* __restrict__ partial,
                                                        int N, int C, int Ch, int seg_tokens) {
  extern __shared__ float fa_lds[];
  float* s_a = fa_lds;                  // [FA_TOK][C]
  float* s_b = fa_lds + FA_TOK * C;     // [FA_TOK][C]
  const int seg = blockIdx.x, b = blockIdx.y, nseg = gridDim.x;
  const int n0 = seg * seg_tokens, n1 = min(N, n0 + seg_tokens);
  const int nout = C * Ch;
  float acc[MAXO];
  int ia[MAXO], ib[MAXO];
  {
    // output o = threadIdx.x + i * NT = (h, kc, vc): the thread's first output by two divisions, the next ones by adding
    // NT in mixed radix (Ch, Ch) - the unrolled form with two runtime divisions per output spent more instructions here
    // (2 x 48 divisions per thread) than in the token loop of a short segment
    const int q1 = NT / Ch, r1 = NT - q1 * Ch;          // NT = q1 * Ch + r1
    const int q2 = q1 / Ch, r2 = q1 - q2 * Ch;          // q1 = q2 * Ch + r2
    int hk0 = threadIdx.x / Ch;
    int vc = threadIdx.x - hk0 * Ch;
    int h = hk0 / Ch;
    int kc = hk0 - h * Ch;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      const bool on = threadIdx.x + i * NT < nout;
      ia[i] = on ? h * Ch + kc : 0;
      ib[i] = on ? h * Ch + vc : 0;
      acc[i] = 0.0f;
      vc += r1;
      kc += r2;
      if (vc >= Ch) { vc -= Ch; kc += 1; }
      h += q2;
      if (kc >= Ch) { kc -= Ch; h += 1; }
    }
  }
  const float* row0 = qkv + ((long)b * N) * 3 * C;
  const float* bsrc = bm_src + ((long)b * N) * bm_row;
  for (int t0 = n0; t0 < n1; t0 += FA_TOK) {
    const int tn = min(FA_TOK, n1 - t0);
    __syncthreads();
    TokChan tc(C);
    for (int e = threadIdx.x; e < tn * C; e += NT, tc.next(C)) {
      const int tt = tc.tt, c = tc.c;
      const long n = t0 + tt;
      float av;
      if (SOFTMAX) av = __expf(row0[n * 3 * C + C + c] - kmax[b * C + c]) * krsum[b * C + c];
      else av = row0[n * 3 * C + c];
      s_a[e] = av;
      s_b[e] = bsrc[n * bm_row + c];
    }
    __syncthreads();
    for (int tt = 0; tt < tn; ++tt) {
#pragma unroll
      for (int i = 0; i < MAXO; ++i) acc[i] = fmaf(s_a[tt * C + ia[i]], s_b[tt * C + ib[i]], acc[i]);
    }
  }
  float* po = partial + ((long)b * nseg + seg) * nout;
#pragma unroll
  for (int i = 0; i < MAXO; ++i) {
    const int o = threadIdx.x + i * NT;
    if (o < nout) po[o] = acc[i];
  }
}

// The same partial sums with a thread per context ROW (h, kc): the row's a-value is one LDS read per token and the head's
// Ch b-values arrive as 16-byte reads (each head's slice padded to CHP = a multiple of 4 floats), so a token costs
// 1 + CHP / 4 LDS reads for Ch multiply-adds - the entry-per-thread form above needs two reads for each - and the Ch
// sums of a row live in registers without index arrays.  Block = C rounded up to whole waves (<= 512 threads).
template <bool SOFTMAX, int CHP>
__global__ __launch_bounds__(512) void fa_context_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ bm_src,
                                                              int bm_row, const float* __restrict__ kmax,
                                                              const float* __restrict__ krsum, float* __restrict__ partial,
                                                              int N, int C, int Ch, int seg_tokens) {
  extern __shared__ float fa_lds[];
  const int heads = C / Ch;
  float* s_a = fa_lds;                               // [FA_TOK][C]
  float* s_b = fa_lds + FA_TOK * C;                  // [FA_TOK][heads][CHP]
  const int seg = blockIdx.x, b = blockIdx.y, nseg = gridDim.x;
  const int n0 = seg * seg_tokens, n1 = min(N, n0 + seg_tokens);
  const int nt = (int)blockDim.x;
  const int hk = threadIdx.x;                        // this thread's row (idle beyond C)
  const int h = hk < C ? hk / Ch : 0;
  float acc[CHP];
#pragma unroll
  for (int i = 0; i < CHP; ++i) acc[i] = 0.0f;
  const float* row0 = qkv + ((long)b * N) * 3 * C;
  const float* bsrc = bm_src + ((long)b * N) * bm_row;
  // (token, channel) walker for the staging loops: blockDim.x added in radix C, channel -> (head, vc) by a second walker
  const int q = nt / C, r = nt - q * C;
  for (int t0 = n0; t0 < n1; t0 += FA_TOK) {
    const int tn = min(FA_TOK, n1 - t0);
    __syncthreads();
    int tt = (int)threadIdx.x / C, c = (int)threadIdx.x - tt * C;
    for (int e = threadIdx.x; e < tn * C; e += nt) {
      const long n = t0 + tt;
      float av;
      if (SOFTMAX) av = __expf(row0[n * 3 * C + C + c] - kmax[b * C + c]) * krsum[b * C + c];
      else av = row0[n * 3 * C + c];
      s_a[e] = av;
      const int ch = c / Ch;                          // (one division per staged element; 16 tokens per pass)
      s_b[(tt * heads + ch) * CHP + (c - ch * Ch)] = bsrc[n * bm_row + c];
      c += r; tt += q;
      if (c >= C) { c -= C; tt += 1; }
    }
    __syncthreads();
    if (hk < C) {
      for (int t = 0; t < tn; ++t) {
        const float av = s_a[t * C + hk];
        const float4* bp = reinterpret_cast<const float4*>(s_b + (t * heads + h) * CHP);
#pragma unroll
        for (int i = 0; i < CHP / 4; ++i) {
          const float4 bv = bp[i];
          acc[4 * i + 0] = fmaf(av, bv.x, acc[4 * i + 0]);
          acc[4 * i + 1] = fmaf(av, bv.y, acc[4 * i + 1]);
          acc[4 * i + 2] = fmaf(av, bv.z, acc[4 * i + 2]);
          acc[4 * i + 3] = fmaf(av, bv.w, acc[4 * i + 3]);
        }
      }
    }
  }
  if (hk < C) {
    float* po = partial + ((long)b * nseg + seg) * C * Ch + (long)hk * Ch;
#pragma unroll
    for (int i = 0; i < CHP; ++i)
      if (i < Ch) po[i] = acc[i];
  }
}

// ctx[b, o] = scale * sum_seg partial[b, seg, o]   (fixed order)
__global__ __launch_bounds__(NT) void fa_context_reduce_kernel(const float* __restrict__ partial, float* __restrict__ ctx,
                                                               int nseg, int nout, int total, float scale) {
  __shared__ float sh[RL][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int id = blockIdx.x * 64 + cl;
  const bool live = id < total;
  const int b = live ? id / nout : 0, o = live ? id - b * nout : 0;
  float s = 0.0f;
  if (live)
    for (int g = rl; g < nseg; g += RL) s += partial[((long)b * nseg + g) * nout + o];
  sh[rl][cl] = s;
  __syncthreads();
  if (!live || rl != 0) return;
  s = sh[0][cl];
#pragma unroll
  for (int q = 1; q < RL; ++q) s += sh[q][cl];
  ctx[id] = s * scale;
}

// out[b,n,c=(h,vc)] = sum_kc q[b,n,h,kc] * ctxs[b,h,kc,vc] + q[b,n,c] * convv[b,n,c]
__global__ __launch_bounds__(NT) void fa_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ ctxs,
                                                      const float* __restrict__ convv, float* __restrict__ out, int N,
                                                      int C, int Ch, int tok_per_block) {
  extern __shared__ float fa_lds[];
  float* s_ctx = fa_lds;                           // [C][Ch]  (entry (h,kc,vc) at (h*Ch+kc)*Ch + vc)
  float* s_q = fa_lds + C * Ch;                    // [tok_per_block][C]
  int* s_head = reinterpret_cast<int*>(s_q + tok_per_block * C);     // [C] head of a channel (one division per channel)
  const int b = blockIdx.y, n0 = blockIdx.x * tok_per_block, tn = min(tok_per_block, N - n0);
  for (int e = threadIdx.x; e < C * Ch; e += NT) s_ctx[e] = ctxs[(long)b * C * Ch + e];
  for (int c = threadIdx.x; c < C; c += NT) s_head[c] = c / Ch;
  const float* row0 = qkv + ((long)b * N + n0) * 3 * C;
  {
    TokChan tc(C);
    for (int e = threadIdx.x; e < tn * C; e += NT, tc.next(C)) s_q[e] = row0[(long)tc.tt * 3 * C + tc.c];
  }
  __syncthreads();
  TokChan tc(C);
  for (int e = threadIdx.x; e < tn * C; e += NT, tc.next(C)) {
    const int tt = tc.tt, c = tc.c;
    const int h = s_head[c], vc = c - h * Ch;
    const float* qh = s_q + tt * C + h * Ch;
    const float* cx = s_ctx + (h * Ch) * Ch + vc;
    float acc = 0.0f;
    for (int kc = 0; kc < Ch; ++kc) acc = fmaf(qh[kc], cx[kc * Ch], acc);
    const long o = ((long)b * N + n0 + tt) * C + c;
    out[o] = acc + s_q[e] * convv[o];
  }
}

// Token-parallel backward: dq | dk | dv into dqkv [B,N,3C] and dconvv [B,N,C].
//   dq[n,h,x] = sum_vc dout[n,h,vc] ctxs[h,x,vc] + dout[n,h,x] convv[n,h,x]
//   dv[n,h,x] = sum_kc p[n,h,kc] D[h,kc,x]              D = scale * sum_n q (x) dout  (d loss / d context)
//   dk[n,h,x] = p[n,h,x] (sum_vc v[n,h,vc] D[h,x,vc] - r[h,x]),   r[h,x] = sum_vc ctxs[h,x,vc] D[h,x,vc] / scale
__global__ __launch_bounds__(NT) void fa_bwd_token_kernel(const float* __restrict__ qkv, const float* __restrict__ ctxs,
                                                          const float* __restrict__ dctx, const float* __restrict__ convv,
                                                          const float* __restrict__ dout, const float* __restrict__ kmax,
                                                          const float* __restrict__ krsum, float* __restrict__ dqkv,
                                                          float* __restrict__ dconvv, int N, int C, int Ch,
                                                          int tok_per_block, float inv_scale) {
  extern __shared__ float fa_lds[];
  const int Chp = Ch | 1;                          // odd row stride: row- and column-wise reads both conflict-free
  float* s_ctx = fa_lds;                           // [C][Chp]
  float* s_d = s_ctx + C * Chp;                    // [C][Chp]
  float* s_r = s_d + C * Chp;                      // [C]
  float* s_p = s_r + C;                            // [tok][C] softmax(k)
  float* s_v = s_p + tok_per_block * C;            // [tok][C]
  float* s_g = s_v + tok_per_block * C;            // [tok][C] dout
  int* s_head = reinterpret_cast<int*>(s_g + tok_per_block * C);     // [C] head of a channel
  const int b = blockIdx.y, n0 = blockIdx.x * tok_per_block, tn = min(tok_per_block, N - n0);
  {
    // (row hk, column vc) of context entry e = threadIdx.x + k * NT: NT added in radix Ch
    const int q = NT / Ch, r = NT - q * Ch;
    int hk = (int)threadIdx.x / Ch, vc = (int)threadIdx.x - hk * Ch;
    for (int e = threadIdx.x; e < C * Ch; e += NT) {
      s_ctx[hk * Chp + vc] = ctxs[(long)b * C * Ch + e];
      s_d[hk * Chp + vc] = dctx[(long)b * C * Ch + e];
      vc += r; hk += q;
      if (vc >= Ch) { vc -= Ch; hk += 1; }
    }
  }
  for (int c = threadIdx.x; c < C; c += NT) s_head[c] = c / Ch;
  const float* row0 = qkv + ((long)b * N + n0) * 3 * C;
  {
    TokChan tc(C);
    for (int e = threadIdx.x; e < tn * C; e += NT, tc.next(C)) {
      const int tt = tc.tt, c = tc.c;
      s_p[e] = __expf(row0[(long)tt * 3 * C + C + c] - kmax[b * C + c]) * krsum[b * C + c];
      s_v[e] = row0[(long)tt * 3 * C + 2 * C + c];
      s_g[e] = dout[((long)b * N + n0 + tt) * C + c];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NT) {
    float r = 0.0f;
    for (int vc = 0; vc < Ch; ++vc) r = fmaf(s_ctx[c * Chp + vc], s_d[c * Chp + vc], r);
    s_r[c] = r * inv_scale;
  }
  __syncthreads();
  TokChan tc(C);
  for (int e = threadIdx.x; e < tn * C; e += NT, tc.next(C)) {
    const int tt = tc.tt, c = tc.c;
    const int h = s_head[c], x = c - h * Ch;
    const float* gh = s_g + tt * C + h * Ch;
    const float* ph = s_p + tt * C + h * Ch;
    const float* vh = s_v + tt * C + h * Ch;
    float aq = 0.0f, av = 0.0f, ap = 0.0f;
    for (int j = 0; j < Ch; ++j) {
      aq = fmaf(gh[j], s_ctx[c * Chp + j], aq);               // row x of ctxs
      ap = fmaf(vh[j], s_d[c * Chp + j], ap);                 // row x of D
      av = fmaf(ph[j], s_d[(h * Ch + j) * Chp + x], av);      // column x of D
    }
    const long tok = (long)b * N + n0 + tt;
    const float q = row0[(long)tt * 3 * C + c];
    const float cv = convv[tok * C + c];
    float* o = dqkv + tok * 3 * C + c;
    o[0] = aq + s_g[e] * cv;
    o[C] = s_p[e] * (ap - s_r[c]);
    o[2 * C] = av;
    dconvv[tok * C + c] = s_g[e] * q;
  }
}

template <bool SOFTMAX>
void fa_launch_context(dim3 grid, size_t lds, hipStream_t st, const float* qkv, const float* bsrc, int brow,
                              const float* kmax, const float* krsum, float* part, int N, int C, int Ch, int seg_tokens) {
  // row form: heads of up to 48 channels, up to 512 rows; the padded b tile must not contain stale values that matter:
  // the pad lanes of a head only feed accumulators that are never stored
  // (narrow token rows - C <= 128, one or two waves of rows per block - stay with the entry-per-thread form: measured
  // 19-25 us there against 22-51 us for the row form; wide rows 16-23 us against 23-38 us)
  if (Ch <= 48 && C <= 512 && C > 128) {
    const int chp = (Ch + 3) & ~3, heads = C / Ch;
    const size_t lds2 = (size_t)FA_TOK * (C + heads * chp) * sizeof(float);
    const dim3 block((unsigned)((C + 63) / 64 * 64));
#define BBD_FAR(P) hipLaunchKernelGGL((fa_context_rows_kernel<SOFTMAX, P>), grid, block, lds2, st, qkv, bsrc, brow, kmax, krsum, part, N, C, Ch, seg_tokens)
    switch (chp) {
      case 4: BBD_FAR(4); break;    case 8: BBD_FAR(8); break;    case 12: BBD_FAR(12); break;  case 16: BBD_FAR(16); break;
      case 20: BBD_FAR(20); break;  case 24: BBD_FAR(24); break;  case 28: BBD_FAR(28); break;  case 32: BBD_FAR(32); break;
      case 36: BBD_FAR(36); break;  case 40: BBD_FAR(40); break;  case 44: BBD_FAR(44); break;  default: BBD_FAR(48); break;
    }
#undef BBD_FAR
    return;
  }
  const int nacc = (C * Ch + NT - 1) / NT;
#define BBD_FA(M) hipLaunchKernelGGL((fa_context_kernel<SOFTMAX, M>), grid, dim3(NT), lds, st, qkv, bsrc, brow, kmax, krsum, part, N, C, Ch, seg_tokens)
  if (nacc <= 2) BBD_FA(2);
  else if (nacc <= 8) BBD_FA(8);
  else if (nacc <= 24) BBD_FA(24);
  else BBD_FA(48);
#undef BBD_FA
}

/* segments of tokens: enough workgroups to fill the chip, at least FA_TOK tokens each */
int fa_segments(int B, int N) {
  int nseg = (1024 + B - 1) / B;
  const int max_seg = (N + FA_TOK - 1) / FA_TOK;
  if (nseg > max_seg) nseg = max_seg;
  return nseg < 1 ? 1 : nseg;
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

long bbd_dwconv_tokens_wgrad_scratch_floats(int B, int H, int W, int C, int k) {
  const long segs = (long)B * H * ((W + WCH - 1) / WCH);
  return (long)wgrad_rows(segs, C) * C * (k * k + 1);
}

int bbd_dwconv_tokens_wgrad(const float* x, int x_row, const float* grad_y, int gy_row, float* partial,
                            float* grad_weight, float* grad_bias, int B, int H, int W, int C, int k, int accumulate,
                            void* stream) {
  if (!x || !grad_y || !partial || !grad_weight || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BBD_E_BADARG;
  if (k != 3 && k != 5 && k != 7) return BBD_E_BADARG;
  const long segs = (long)B * H * ((W + WCH - 1) / WCH);
  const int rows = wgrad_rows(segs, C);
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)rows);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (k == 3) hipLaunchKernelGGL(dwconv_tokens_wgrad_kernel<3>, grid, dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial, B, H, W, C);
  else if (k == 5) hipLaunchKernelGGL(dwconv_tokens_wgrad_kernel<5>, grid, dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial, B, H, W, C);
  else hipLaunchKernelGGL(dwconv_tokens_wgrad_kernel<7>, grid, dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial, B, H, W, C);
  const int cols = C * (k * k + 1);
  hipLaunchKernelGGL(dwconv_tokens_wgrad_final_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64 * WRL), 0, st, partial,
                     grad_weight, grad_bias, rows, C, k * k, accumulate);
  return launch_status();
}

// ---- several channel groups in one launch ----------------------------------------------------------------------------
static int fill_groups(DwGroups* g, int n, const int32_t* c0, const int32_t* cn, const int32_t* k, const void* const* w,
                       const void* const* b) {
  if (n < 1 || n > DW_MAX_GROUPS || !c0 || !cn || !k || !w) return 1;
  g->n = n;
  for (int i = 0; i < DW_MAX_GROUPS; ++i) {
    const bool on = i < n;
    g->c0[i] = on ? c0[i] : 0; g->cn[i] = on ? cn[i] : 0; g->k[i] = on ? k[i] : 3;
    g->w[i] = on ? static_cast<const float*>(w[i]) : nullptr;
    g->b[i] = (on && b) ? static_cast<const float*>(b[i]) : nullptr;
    g->gw[i] = nullptr; g->gb[i] = nullptr; g->part0[i] = 0;
    if (on && (g->cn[i] <= 0 || !g->w[i] || (g->k[i] != 3 && g->k[i] != 5 && g->k[i] != 7))) return 1;
  }
  return 0;
}

int bbd_dwconv_tokens_groups_fwd(const float* x, int x_row, float* y, int y_row, int n_groups, const int32_t* c0,
                                 const int32_t* cn, const int32_t* k, const void* const* weights, const void* const* biases,
                                 int B, int H, int W, int add_input, int flip, void* stream) {
  DwGroups g;
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || fill_groups(&g, n_groups, c0, cn, k, weights, biases)) return BBD_E_BADARG;
  unsigned at = 0;
  for (int i = 0; i < n_groups; ++i) {
    if (c0[i] + cn[i] > x_row || c0[i] + cn[i] > y_row) return BBD_E_BADARG;
    g.block0[i] = at;
    const long total = (long)B * H * ((W + PW - 1) / PW) * cn[i];
    at += (unsigned)((total + NT - 1) / NT);
  }
  for (int i = n_groups; i <= DW_MAX_GROUPS; ++i) g.block0[i] = at;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (flip) hipLaunchKernelGGL(dwconv_tokens_groups_kernel<true>, dim3(at), dim3(NT), 0, st, x, x_row, y, y_row, g, B, H, W, add_input);
  else hipLaunchKernelGGL(dwconv_tokens_groups_kernel<false>, dim3(at), dim3(NT), 0, st, x, x_row, y, y_row, g, B, H, W, add_input);
  return launch_status();
}

static int groups_wgrad_rows(long segs, int n, const int32_t* cn) {
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (cn[i] + 63) / 64;
  return wgrad_rows(segs, tiles * 64);
}

long bbd_dwconv_tokens_groups_wgrad_scratch_floats(int B, int H, int W, int n_groups, const int32_t* cn, const int32_t* k) {
  if (n_groups < 1 || n_groups > DW_MAX_GROUPS || !cn || !k) return 0;
  const long segs = (long)B * H * ((W + WCH - 1) / WCH);
  const int rows = groups_wgrad_rows(segs, n_groups, cn);
  long total = 0;
  for (int i = 0; i < n_groups; ++i) total += (long)rows * cn[i] * (k[i] * k[i] + 1);
  return total;
}

int bbd_dwconv_tokens_groups_wgrad(const float* x, int x_row, const float* grad_y, int gy_row, float* partial, int n_groups,
                                   const int32_t* c0, const int32_t* cn, const int32_t* k, const void* const* grad_weights,
                                   const void* const* grad_biases, int B, int H, int W, int accumulate, void* stream) {
  DwGroups g;
  if (!x || !grad_y || !partial || B <= 0 || H <= 0 || W <= 0 || fill_groups(&g, n_groups, c0, cn, k, grad_weights, grad_biases))
    return BBD_E_BADARG;
  const long segs = (long)B * H * ((W + WCH - 1) / WCH);
  const int rows = groups_wgrad_rows(segs, n_groups, cn);
  unsigned at = 0, fat = 0;
  long part = 0;
  for (int i = 0; i < n_groups; ++i) {
    g.gw[i] = const_cast<float*>(g.w[i]);          // fill_groups parked the gradient pointers in w / b
    g.gb[i] = const_cast<float*>(g.b[i]);
    g.w[i] = nullptr; g.b[i] = nullptr;
    g.block0[i] = at; g.fblock0[i] = fat; g.part0[i] = part;
    at += (unsigned)((cn[i] + 63) / 64);
    fat += (unsigned)((cn[i] * (k[i] * k[i] + 1) + 63) / 64);
    part += (long)rows * cn[i] * (k[i] * k[i] + 1);
  }
  for (int i = n_groups; i <= DW_MAX_GROUPS; ++i) { g.block0[i] = at; g.fblock0[i] = fat; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dwconv_tokens_wgrad_groups_kernel, dim3(at, (unsigned)rows), dim3(NT), 0, st, x, x_row, grad_y, gy_row, partial,
                     g, B, H, W);
  hipLaunchKernelGGL(dwconv_tokens_wgrad_final_groups_kernel, dim3(fat), dim3(64 * WRL), 0, st, partial, g, rows, accumulate);
  return launch_status();
}

int bbd_factor_att_segments(int B, int N) { return fa_segments(B, N); }
int bbd_factor_att_supported(int C, int Ch) {
  return C > 0 && Ch > 0 && C % Ch == 0 && (long)C * Ch <= (long)NT * FA_MAXO && C <= 1024;
}

int bbd_factor_att_fwd(const float* qkv, const float* convv, float* kmax, float* krsum, float* ctxs, float* scratch,
                       float* out, int B, int N, int C, int Ch, double scale, void* stream) {
  if (!qkv || !convv || !kmax || !krsum || !ctxs || !scratch || !out || B <= 0 || N <= 0) return BBD_E_BADARG;
  if (!bbd_factor_att_supported(C, Ch)) return BBD_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nseg = fa_segments(B, N), seg_tokens = (N + nseg - 1) / nseg, nout = C * Ch;
  float* pm = scratch;                            // [B, nseg, C]
  float* ps = pm + (long)B * nseg * C;            // [B, nseg, C]
  float* part = ps + (long)B * nseg * C;          // [B, nseg, C*Ch]
  hipLaunchKernelGGL(fa_kstats_kernel, dim3(nseg, B), dim3(NT), 0, st, qkv, pm, ps, N, C, seg_tokens);
  hipLaunchKernelGGL(fa_kstats_combine_kernel, dim3((unsigned)((B * C + 63) / 64)), dim3(NT), 0, st, pm, ps, kmax,
                     krsum, nseg, C, B * C);
  fa_launch_context<true>(dim3(nseg, B), (size_t)2 * FA_TOK * C * sizeof(float), st, qkv, qkv + 2 * C, 3 * C, kmax, krsum,
                          part, N, C, Ch, seg_tokens);
  hipLaunchKernelGGL(fa_context_reduce_kernel, dim3((unsigned)(((long)B * nout + 63) / 64)), dim3(NT), 0, st, part,
                     ctxs, nseg, nout, B * nout, (float)scale);
  const int tpb = 8;
  hipLaunchKernelGGL(fa_apply_kernel, dim3((unsigned)((N + tpb - 1) / tpb), B), dim3(NT),
                     (size_t)(nout + tpb * C + C) * sizeof(float), st, qkv, ctxs, convv, out, N, C, Ch, tpb);
  return launch_status();
}

long bbd_factor_att_scratch_floats(int B, int N, int C, int Ch) {
  const long nseg = fa_segments(B, N);
  return (long)B * nseg * (2L * C + (long)C * Ch);
}

int bbd_factor_att_bwd(const float* qkv, const float* convv, const float* kmax, const float* krsum, const float* ctxs,
                       const float* grad_out, float* dctx, float* scratch, float* grad_qkv, float* grad_convv, int B,
                       int N, int C, int Ch, double scale, void* stream) {
  if (!qkv || !convv || !kmax || !krsum || !ctxs || !grad_out || !dctx || !scratch || !grad_qkv || !grad_convv)
    return BBD_E_BADARG;
  if (B <= 0 || N <= 0 || !bbd_factor_att_supported(C, Ch)) return BBD_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nseg = fa_segments(B, N), seg_tokens = (N + nseg - 1) / nseg, nout = C * Ch;
  float* part = scratch + 2L * B * nseg * C;
  fa_launch_context<false>(dim3(nseg, B), (size_t)2 * FA_TOK * C * sizeof(float), st, qkv, grad_out, C, kmax, krsum, part,
                           N, C, Ch, seg_tokens);
  hipLaunchKernelGGL(fa_context_reduce_kernel, dim3((unsigned)(((long)B * nout + 63) / 64)), dim3(NT), 0, st, part,
                     dctx, nseg, nout, B * nout, (float)scale);
  const int tpb = 8, Chp = Ch | 1;
  const size_t lds = (size_t)(2 * C * Chp + C + 3 * tpb * C + C) * sizeof(float);
  if (lds > 64 * 1024) {      // up to 114 KB for C = 288, Ch = 36: above the default dynamic-LDS limit
    static size_t granted = 0;
    if (lds > granted) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fa_bwd_token_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return BBD_E_BADARG;
      granted = lds;
    }
  }
  hipLaunchKernelGGL(fa_bwd_token_kernel, dim3((unsigned)((N + tpb - 1) / tpb), B), dim3(NT), lds, st, qkv, ctxs, dctx,
                     convv, grad_out, kmax, krsum, grad_qkv, grad_convv, N, C, Ch, tpb, (float)(1.0 / scale));
  return launch_status();
}

}  // extern "C"
