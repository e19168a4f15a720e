// bbd_kernels.hip - hand-written gfx950 (CDNA4, wave64) kernels of the photometric
// reprojection hot path and their C-ABI launchers (include/bbd_hip.h).
//
// Layout / tiling (DESIGN.md "Kernels"):
//   * one workgroup = 256 threads = 4 waves = one 64x16 pixel tile of one (scale, sample);
//   * each thread owns a horizontal strip of 4 pixels (16-byte global stores, 16-byte LDS
//     window reads); 16 lanes cover a 256-byte tile row, 4 rows per wave;
//   * target and warped tiles (+1 px reflected halo) are staged in LDS as planar fp32 with a
//     row stride of 68 floats, so every strip's window read is 16-byte aligned and a
//     16-lane group reads one contiguous 256-byte bank row (conflict-free);
//   * the candidate loop is block-uniform (candidates are per sample), so the candidate
//     descriptor and the 3x4 projection sit in SGPRs;
//   * the 1-D grid is ordered sample-major, then scale, then tile: the four scales of a
//     sample re-read the same source/target images while they are L2 / Infinity-Cache hot.
//
// Arithmetic lives in bbd_math.h and is shared with the host port used by the CPU tests.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"
#include "bbd_math.h"

namespace {

constexpr int TW = 64;   // tile width  (pixels)
constexpr int TH = 16;   // tile height (pixels)
constexpr int NT = 256;  // threads per workgroup
constexpr int PPT = 4;   // pixels per thread (horizontal strip)
constexpr int SPR = TW / PPT;  // strips per tile row = 16

// forward staging: (TH+2) x (TW+2) cells, row stride LS
constexpr int LS = TW + 4;
constexpr int LH = TH + 2;
constexpr int LW = TW + 2;
constexpr int FPLANE = LH * LS;

// backward staging: x/y region (TH+4) x (TW+4), coefficient region (TH+2) x (TW+2)
constexpr int BS = TW + 8;          // 72
constexpr int BH = TH + 4;
constexpr int BW = TW + 4;
constexpr int BPLANE = BH * BS;
constexpr int CS = TW + 4;          // 68
constexpr int CH = TH + 2;
constexpr int CW = TW + 2;
constexpr int CPLANE = CH * CS;

struct FramePtrs {
  const float* base[BBD_MAX_FRAME_SLOTS];
};

constexpr int KIND_MASK = 0xff;
constexpr int FLAG_NO_POSE_GRAD = 0x100;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over the 256 threads of the block; result valid in thread 0.  Fixed order -> deterministic.
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_red[wv] = v;
  __syncthreads();
  float r = 0.0f;
  if (threadIdx.x == 0) r = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  __syncthreads();
  return r;
}

struct TileCoord {
  int tx0, ty0;   // image coordinates of the tile's first pixel
  int tile;       // tile index inside the image
};

__device__ __forceinline__ TileCoord decode_tile(int t, int W) {
  const int tiles_x = (W + TW - 1) / TW;
  TileCoord c;
  c.tile = t;
  c.ty0 = (t / tiles_x) * TH;
  c.tx0 = (t % tiles_x) * TW;
  return c;
}

// Stage one [3,H,W] image tile (+1 reflected halo) into planar LDS.
__device__ __forceinline__ void stage_image_tile(const float* img, int H, int W, int tx0, int ty0,
                                                 float (*s)[FPLANE]) {
  const size_t plane = (size_t)H * W;
  for (int i = threadIdx.x; i < LH * LW; i += NT) {
    const int r = i / LW, c = i - r * LW;
    const int yy = bbd_reflect(ty0 + r - 1, H), xx = bbd_reflect(tx0 + c - 1, W);
    const float* p = img + (size_t)yy * W + xx;
    s[0][r * LS + c] = p[0];
    s[1][r * LS + c] = p[plane];
    s[2][r * LS + c] = p[2 * plane];
  }
}

// 3 rows x 6 columns window of one plane for a strip (row ly, first padded column lx0).
__device__ __forceinline__ void load_window(const float* plane, int ly, int lx0, float win[3][6]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* p = plane + (ly + r) * LS + lx0;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float2 b = *reinterpret_cast<const float2*>(p + 4);
    win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w;
    win[r][4] = b.x; win[r][5] = b.y;
  }
}

// Target-window statistics for the strip's 4 pixels, all 3 channels.
__device__ __forceinline__ void strip_ystats(const float (*sy)[FPLANE], int ly, int lx0,
                                             float mu_y[3][PPT], float sg_y[3][PPT]) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float y[3][6];
    load_window(sy[ch], ly, lx0, y);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float s = 0.0f, ss = 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = y[r][j + c];
          s += v;
          ss += v * v;
        }
      bbd_ystats(s, ss, &mu_y[ch][j], &sg_y[ch][j]);
    }
  }
}

// Photometric loss of the strip's 4 pixels given the staged prediction (sx) and target (sy).
__device__ __forceinline__ void strip_loss(const float (*sx)[FPLANE], const float (*sy)[FPLANE],
                                           int ly, int lx0, const float mu_y[3][PPT],
                                           const float sg_y[3][PPT], int no_ssim, float out[PPT]) {
  float ssim[PPT][3], l1[PPT][3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float x[3][6], y[3][6];
    load_window(sx[ch], ly, lx0, x);
    load_window(sy[ch], ly, lx0, y);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float s = 0.0f, ss = 0.0f, sxy = 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = x[r][j + c];
          s += v;
          ss += v * v;
          sxy += v * y[r][j + c];
        }
      ssim[j][ch] = no_ssim ? 0.0f : bbd_ssim(s, ss, sxy, mu_y[ch][j], sg_y[ch][j]);
      l1[j][ch] = fabsf(y[1][j + 1] - x[1][j + 1]);
    }
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) out[j] = bbd_combine(ssim[j], l1[j], no_ssim);
}

// ------------------------------------------------------------------------------------------
// Identity loss (trainer.py:501-508): SSIM+L1 between an un-warped source and the target.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void identity_loss_kernel(FramePtrs frames, const float* __restrict__ target,
                                                           const int32_t* __restrict__ items,
                                                           float* __restrict__ ident, int H, int W,
                                                           int ntiles, int no_ssim) {
  __shared__ __attribute__((aligned(16))) float s_y[3][FPLANE];
  __shared__ __attribute__((aligned(16))) float s_x[3][FPLANE];
  const int item = blockIdx.x / ntiles;
  const TileCoord tc = decode_tile(blockIdx.x - item * ntiles, W);
  const int b = items[item * 4 + 0], slot = items[item * 4 + 1], row = items[item * 4 + 2];
  const size_t img = (size_t)3 * H * W;
  stage_image_tile(target + (size_t)b * img, H, W, tc.tx0, tc.ty0, s_y);
  stage_image_tile(frames.base[slot] + (size_t)row * img, H, W, tc.tx0, tc.ty0, s_x);
  __syncthreads();
  const int ly = threadIdx.x / SPR, lx0 = (threadIdx.x % SPR) * PPT;
  float mu_y[3][PPT], sg_y[3][PPT], loss[PPT];
  strip_ystats(s_y, ly, lx0, mu_y, sg_y);
  strip_loss(s_x, s_y, ly, lx0, mu_y, sg_y, no_ssim, loss);
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  if (yy < H) {
    float* o = ident + (size_t)item * H * W + (size_t)yy * W + xx;
    if (xx + PPT <= W && (W & 3) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(loss[0], loss[1], loss[2], loss[3]);
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j)
        if (xx + j < W) o[j] = loss[j];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused forward: warp + SSIM/L1 + min/arg-min over the candidate list.
// ------------------------------------------------------------------------------------------
struct FwdArgs {
  FramePtrs frames;
  const float* target;
  const float* depth;
  const float* proj;
  const float* ident;
  const float* noise;
  const bbd_cand_t* cand;
  const int32_t* ncand;
  float* min_loss;
  uint8_t* argmin;
  float* partial;
  float* warped;
  int S, B, NP, H, W, ntiles, no_ssim;
};

// Warp one source image into the staged tile (with halo) for projection row `proj`.
template <int ROWS, int COLS, int STRIDE, int HALO, int PLANE>
__device__ __forceinline__ void warp_into_lds(const float* __restrict__ src, const float* __restrict__ depth,
                                              const float* __restrict__ proj, int H, int W, int tx0, int ty0,
                                              float (*s)[PLANE], float* __restrict__ warped_out) {
  const size_t plane = (size_t)H * W;
  float pj[21];
  bbd_make_proj(proj, pj);
  for (int i = threadIdx.x; i < ROWS * COLS; i += NT) {
    const int r = i / COLS, c = i - r * COLS;
    const int py = ty0 + r - HALO, px = tx0 + c - HALO;
    const int yy = bbd_reflect(py, H), xx = bbd_reflect(px, W);
    const float d = depth[(size_t)yy * W + xx];
    BbdSample sm;
    bbd_project(pj, xx, yy, d, H, W, &sm);
    BbdTaps t;
    bbd_taps(sm.ix, sm.iy, &t);
    float val[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float v[4];
      bbd_fetch4(src + ch * plane, H, W, &t, v);
      val[ch] = bbd_bilerp(v, &t);
      s[ch][r * STRIDE + c] = val[ch];
    }
    if (warped_out != nullptr && py == yy && px == xx && r >= HALO && r < ROWS - HALO && c >= HALO &&
        c < COLS - HALO) {
      float* o = warped_out + (size_t)yy * W + xx;
      o[0] = val[0];
      o[plane] = val[1];
      o[2 * plane] = val[2];
    }
  }
}

__global__ __launch_bounds__(NT) void warp_ssim_min_fwd_kernel(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_y[3][FPLANE];
  __shared__ __attribute__((aligned(16))) float s_x[3][FPLANE];
  __shared__ float s_red[4];
  const int H = a.H, W = a.W;
  // grid order: sample-major, then scale, then tile
  int bid = blockIdx.x;
  const int b = bid / (a.S * a.ntiles);
  bid -= b * a.S * a.ntiles;
  const int s = bid / a.ntiles;
  const TileCoord tc = decode_tile(bid - s * a.ntiles, W);
  const size_t hw = (size_t)H * W, img = 3 * hw;
  const size_t sb = (size_t)s * a.B + b;

  stage_image_tile(a.target + (size_t)b * img, H, W, tc.tx0, tc.ty0, s_y);
  __syncthreads();
  const int ly = threadIdx.x / SPR, lx0 = (threadIdx.x % SPR) * PPT;
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  const bool row_ok = yy < H;
  const bool vec_ok = row_ok && (xx + PPT <= W) && ((W & 3) == 0);
  const size_t pix = (size_t)yy * W + xx;

  float mu_y[3][PPT], sg_y[3][PPT];
  strip_ystats(s_y, ly, lx0, mu_y, sg_y);

  float nz[PPT] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (a.noise != nullptr && row_ok) {
    const float* np_ = a.noise + (size_t)b * hw + pix;
    if (vec_ok) {
      const float4 v = *reinterpret_cast<const float4*>(np_);
      nz[0] = v.x; nz[1] = v.y; nz[2] = v.z; nz[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j)
        if (xx + j < W) nz[j] = np_[j];
    }
  }

  float best[PPT];
  int arg[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) { best[j] = INFINITY; arg[j] = 0; }

  const int nc = a.ncand[b];
  const float* depth = a.depth + sb * hw;
  for (int c = 0; c < nc; ++c) {
    const bbd_cand_t cd = a.cand[b * BBD_MAX_CAND + c];
    float loss[PPT];
    if ((cd.kind & KIND_MASK) == BBD_KIND_WARP) {
      const float* src = a.frames.base[cd.slot] + (size_t)cd.row * img;
      float* wout = a.warped ? a.warped + ((size_t)s * a.NP + cd.pose) * img : nullptr;
      warp_into_lds<LH, LW, LS, 1, FPLANE>(src, depth, a.proj + (size_t)cd.pose * BBD_POSE_STRIDE, H, W,
                                           tc.tx0, tc.ty0, s_x, wout);
      __syncthreads();
      strip_loss(s_x, s_y, ly, lx0, mu_y, sg_y, a.no_ssim, loss);
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) loss[j] = 0.0f;
      if (row_ok) {
        const float* ip = a.ident + (size_t)cd.row * hw + pix;
        if (vec_ok) {
          const float4 v = *reinterpret_cast<const float4*>(ip);
          loss[0] = v.x + nz[0]; loss[1] = v.y + nz[1]; loss[2] = v.z + nz[2]; loss[3] = v.w + nz[3];
        } else {
#pragma unroll
          for (int j = 0; j < PPT; ++j)
            if (xx + j < W) loss[j] = ip[j] + nz[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) bbd_min_update(loss[j], c, &best[j], &arg[j]);
  }

  float tsum = 0.0f;
  if (row_ok) {
    float* mo = a.min_loss + sb * hw + pix;
    uint8_t* ao = a.argmin + sb * hw + pix;
    if (vec_ok) {
      *reinterpret_cast<float4*>(mo) = make_float4(best[0], best[1], best[2], best[3]);
      *reinterpret_cast<uint32_t*>(ao) =
          (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
      tsum = ((best[0] + best[1]) + best[2]) + best[3];
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j)
        if (xx + j < W) {
          mo[j] = best[j];
          ao[j] = (uint8_t)arg[j];
          tsum += best[j];
        }
    }
  }
  const float total = block_sum(tsum, s_red);
  if (threadIdx.x == 0) a.partial[sb * a.ntiles + tc.tile] = total;
}

// ------------------------------------------------------------------------------------------
// Fused backward.
// ------------------------------------------------------------------------------------------
struct BwdArgs {
  FramePtrs frames;
  const float* target;
  const float* depth;
  const float* proj;
  const bbd_cand_t* cand;
  const int32_t* ncand;
  const uint8_t* argmin;
  const float* gscale;
  float* grad_depth;
  float* grad_proj;
  int S, B, NP, H, W, ntiles, no_ssim;
};

__global__ __launch_bounds__(NT) void warp_ssim_min_bwd_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_y[3][BPLANE];
  __shared__ __attribute__((aligned(16))) float s_x[3][BPLANE];
  __shared__ __attribute__((aligned(16))) float s_cf[3][CPLANE];  // A, B, C of the current channel
  __shared__ uint8_t s_arg[CH * CW];
  __shared__ float s_red[4];
  const int H = a.H, W = a.W;
  int bid = blockIdx.x;
  const int b = bid / (a.S * a.ntiles);
  bid -= b * a.S * a.ntiles;
  const int s = bid / a.ntiles;
  const TileCoord tc = decode_tile(bid - s * a.ntiles, W);
  const size_t hw = (size_t)H * W, img = 3 * hw;
  const size_t sb = (size_t)s * a.B + b;
  const float* depth = a.depth + sb * hw;
  const float g = a.gscale[s];
  const float w_ssim = a.no_ssim ? 0.0f : g * 0.85f / 3.0f;
  const float w_l1 = a.no_ssim ? g / 3.0f : g * 0.15f / 3.0f;

  // arg-min ids of the (TH+2)x(TW+2) region of loss pixels that see this tile's texels
  for (int i = threadIdx.x; i < CH * CW; i += NT) {
    const int r = i / CW, c = i - r * CW;
    const int py = tc.ty0 + r - 1, px = tc.tx0 + c - 1;
    s_arg[i] = (py >= 0 && py < H && px >= 0 && px < W) ? a.argmin[sb * hw + (size_t)py * W + px] : 255;
  }
  // target over the (TH+4)x(TW+4) region (reflected)
  {
    const float* tg = a.target + (size_t)b * img;
    for (int i = threadIdx.x; i < BH * BW; i += NT) {
      const int r = i / BW, c = i - r * BW;
      const int yy = bbd_reflect(tc.ty0 + r - 2, H);
      const int xx = bbd_reflect(tc.tx0 + c - 2, W);
      const float* p = tg + (size_t)yy * W + xx;
      s_y[0][r * BS + c] = p[0];
      s_y[1][r * BS + c] = p[hw];
      s_y[2][r * BS + c] = p[2 * hw];
    }
  }
  __syncthreads();

  const int ly = threadIdx.x / SPR, lx0 = (threadIdx.x % SPR) * PPT;
  const int qy = tc.ty0 + ly, qx0 = tc.tx0 + lx0;
  float gdepth[PPT] = {0.0f, 0.0f, 0.0f, 0.0f};

  const int nc = a.ncand[b];
  for (int c = 0; c < nc; ++c) {
    const bbd_cand_t cd = a.cand[b * BBD_MAX_CAND + c];
    if ((cd.kind & KIND_MASK) != BBD_KIND_WARP) continue;
    float* gp_out = a.grad_proj + (((size_t)s * a.NP + cd.pose) * a.ntiles + tc.tile) * 12;
    int mine = 0;
    for (int i = threadIdx.x; i < CH * CW; i += NT) mine |= (s_arg[i] == c);
    if (!__syncthreads_or(mine)) {
      if (threadIdx.x < 12) gp_out[threadIdx.x] = 0.0f;
      continue;
    }
    const float* src = a.frames.base[cd.slot] + (size_t)cd.row * img;
    const float* proj = a.proj + (size_t)cd.pose * BBD_POSE_STRIDE;
    // Staging cell (r,c) holds the warped value AT the reflected image pixel, exactly what
    // ReflectionPad2d would have copied there; cells two steps outside the image are only
    // read by windows of loss pixels that do not exist (s_arg == 255) and are never used.
    {
      float pj[21];
      bbd_make_proj(proj, pj);
      for (int i = threadIdx.x; i < BH * BW; i += NT) {
        const int r = i / BW, cc = i - r * BW;
        const int yy = bbd_reflect(tc.ty0 + r - 2, H);
        const int xx = bbd_reflect(tc.tx0 + cc - 2, W);
        BbdSample sm;
        bbd_project(pj, xx, yy, depth[(size_t)yy * W + xx], H, W, &sm);
        BbdTaps t;
        bbd_taps(sm.ix, sm.iy, &t);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          float v[4];
          bbd_fetch4(src + ch * hw, H, W, &t, v);
          s_x[ch][r * BS + cc] = bbd_bilerp(v, &t);
        }
      }
    }
    __syncthreads();

    float gx[3][PPT];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      // (1) SSIM partials at every loss pixel of the (TH+2)x(TW+2) region won by candidate c
      if (!a.no_ssim) {
        for (int i = threadIdx.x; i < CH * CW; i += NT) {
          const int r = i / CW, cc = i - r * CW;
          float A = 0.0f, Bc = 0.0f, Cc = 0.0f;
          if (s_arg[i] == c) {
            // window of loss pixel p=(r,cc) in the x/y region: rows r..r+2, cols cc..cc+2.
            // Window texels outside the image are the reflection about the image border,
            // which the staging already applied.
            float sx_ = 0.0f, sxx = 0.0f, sxy = 0.0f, sy_ = 0.0f, syy = 0.0f;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
              for (int dc = 0; dc < 3; ++dc) {
                const float xv = s_x[ch][(r + dr) * BS + cc + dc];
                const float yv = s_y[ch][(r + dr) * BS + cc + dc];
                sx_ += xv; sxx += xv * xv; sxy += xv * yv; sy_ += yv; syy += yv * yv;
              }
            float mu_y, sg_y;
            bbd_ystats(sy_, syy, &mu_y, &sg_y);
            bbd_ssim_grad(sx_, sxx, sxy, mu_y, sg_y, &A, &Bc, &Cc);
            A *= w_ssim; Bc *= w_ssim; Cc *= w_ssim;
          }
          s_cf[0][r * CS + cc] = A;
          s_cf[1][r * CS + cc] = Bc;
          s_cf[2][r * CS + cc] = Cc;
        }
        __syncthreads();
      }
      // (2) gather the adjoint of reflect-pad + 3x3 mean at this thread's 4 texels
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int qx = qx0 + j;
        float acc = 0.0f;
        const float xq = s_x[ch][(ly + 2) * BS + lx0 + j + 2];
        const float yq = s_y[ch][(ly + 2) * BS + lx0 + j + 2];
        if (qy < H && qx < W) {
          if (!a.no_ssim) {
            float SA = 0.0f, SB = 0.0f, SC = 0.0f;
#pragma unroll
            for (int dr = -1; dr <= 1; ++dr) {
              const int py = qy + dr;
              if (py < 0 || py >= H) continue;
              const int my = bbd_reflect_mult(qy, py, H);
#pragma unroll
              for (int dc = -1; dc <= 1; ++dc) {
                const int px = qx + dc;
                if (px < 0 || px >= W) continue;
                const float m = (float)(my * bbd_reflect_mult(qx, px, W));
                const int ci = (ly + 1 + dr) * CS + lx0 + j + 1 + dc;
                SA += m * s_cf[0][ci];
                SB += m * s_cf[1][ci];
                SC += m * s_cf[2][ci];
              }
            }
            acc = (SA + xq * SB + yq * SC) * (1.0f / 9.0f);
          }
          if (s_arg[(ly + 1) * CW + lx0 + j + 1] == c) {
            const float df = xq - yq;
            acc += w_l1 * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
          }
        }
        gx[ch][j] = acc;
      }
      if (!a.no_ssim) __syncthreads();
    }

    // (3) texel gradient -> sampling coordinates -> depth and P
    float gP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gP[k] = 0.0f;
    if (qy < H) {
      float pj[21];
      bbd_make_proj(proj, pj);
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int qx = qx0 + j;
        if (qx >= W) continue;
        if (gx[0][j] == 0.0f && gx[1][j] == 0.0f && gx[2][j] == 0.0f) continue;
        BbdSample sm;
        bbd_project(pj, qx, qy, depth[(size_t)qy * W + qx], H, W, &sm);
        BbdTaps t;
        bbd_taps(sm.ix, sm.iy, &t);
        float gix = 0.0f, giy = 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          float v[4];
          bbd_fetch4(src + ch * hw, H, W, &t, v);
          bbd_bilerp_grad(v, &t, gx[ch][j], &gix, &giy);
        }
        float gd, gp1[12];
        bbd_project_grad(pj, &sm, gix, giy, &gd, gp1);
        gdepth[j] += gd;
#pragma unroll
        for (int k = 0; k < 12; ++k) gP[k] += gp1[k];
      }
    }
    if (cd.kind & FLAG_NO_POSE_GRAD) {
      if (threadIdx.x < 12) gp_out[threadIdx.x] = 0.0f;
      __syncthreads();
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const float tot = block_sum(gP[k], s_red);
        if (threadIdx.x == 0) gp_out[k] = tot;
      }
    }
  }

  if (qy < H) {
    float* o = a.grad_depth + sb * hw + (size_t)qy * W + qx0;
    if (qx0 + PPT <= W && (W & 3) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(gdepth[0], gdepth[1], gdepth[2], gdepth[3]);
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j)
        if (qx0 + j < W) o[j] = gdepth[j];
    }
  }
}

// ------------------------------------------------------------------------------------------
// disp -> depth (bilinear upsample + reciprocal affine), forward and adjoint.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void disp_to_depth_fwd_kernel(const float* __restrict__ disp,
                                                               float* __restrict__ depth, int B, int h, int w,
                                                               int H, int W, float lo, float span) {
  const size_t n = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int b = (int)(i / ((size_t)W * H));
    const float* d = disp + (size_t)b * h * w;
    float v;
    if (h == H && w == W) {
      v = d[(size_t)y * w + x];
    } else {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      bbd_up_src(y, h, H, &y0, &y1, &ly0, &ly1);
      bbd_up_src(x, w, W, &x0, &x1, &lx0, &lx1);
      v = bbd_up_blend(d[(size_t)y0 * w + x0], d[(size_t)y0 * w + x1], d[(size_t)y1 * w + x0],
                       d[(size_t)y1 * w + x1], ly0, ly1, lx0, lx1, (H + W) <= 128);
    }
    depth[i] = 1.0f / (lo + span * v);
  }
}

__global__ __launch_bounds__(NT) void disp_to_depth_bwd_kernel(const float* __restrict__ disp,
                                                               const float* __restrict__ gdepth,
                                                               float* __restrict__ gdisp, int B, int h, int w,
                                                               int H, int W, float lo, float span) {
  const size_t n = (size_t)B * h * w;
  const int fy = (H + h - 1) / h, fx = (W + w - 1) / w;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int b = (int)(i / ((size_t)w * h));
    const float* d = disp + (size_t)b * h * w;
    const float* g = gdepth + (size_t)b * H * W;
    float acc = 0.0f;
    if (h == H && w == W) {
      const float sc = lo + span * d[(size_t)y * w + x];
      acc = g[(size_t)y * W + x] * (-span / (sc * sc));
    } else {
      const int oy_lo = max(0, (y - 1) * fy), oy_hi = min(H - 1, (y + 2) * fy);
      const int ox_lo = max(0, (x - 1) * fx), ox_hi = min(W - 1, (x + 2) * fx);
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly0, ly1;
        bbd_up_src(oy, h, H, &y0, &y1, &ly0, &ly1);
        const float wy = (y0 == y ? ly0 : 0.0f) + (y1 == y ? ly1 : 0.0f);
        if (wy == 0.0f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          int x0, x1;
          float lx0, lx1;
          bbd_up_src(ox, w, W, &x0, &x1, &lx0, &lx1);
          const float wx = (x0 == x ? lx0 : 0.0f) + (x1 == x ? lx1 : 0.0f);
          if (wx == 0.0f) continue;
          // recompute the upsampled disparity at (oy, ox) for d depth / d disp_up
          const float sc = lo + span * bbd_up_blend(d[(size_t)y0 * w + x0], d[(size_t)y0 * w + x1],
                                                    d[(size_t)y1 * w + x0], d[(size_t)y1 * w + x1], ly0, ly1,
                                                    lx0, lx1, (H + W) <= 128);
          acc += g[(size_t)oy * W + ox] * (-span / (sc * sc)) * wy * wx;
        }
      }
    }
    gdisp[i] = acc;
  }
}


// ------------------------------------------------------------------------------------------
// Stand-alone layer kernels (layers.BackprojectDepth / Project3D / SSIM API surface).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void backproject_kernel(const float* __restrict__ depth,
                                                         const float* __restrict__ inv_K,
                                                         float* __restrict__ points, int n, int H, int W) {
  const size_t hw = (size_t)H * W, total = (size_t)n * hw;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
    const int b = (int)(i / hw);
    const size_t p = i - (size_t)b * hw;
    const float fx = (float)(p % W), fy = (float)(p / W);
    const float* k = inv_K + (size_t)b * 16;
    const float d = depth[i];
    float* o = points + (size_t)b * 4 * hw + p;
    o[0] = d * bbd_dot3_hom(k, fx, fy);
    o[hw] = d * bbd_dot3_hom(k + 4, fx, fy);
    o[2 * hw] = d * bbd_dot3_hom(k + 8, fx, fy);
    o[3 * hw] = 1.0f;
  }
}

__global__ __launch_bounds__(NT) void project3d_kernel(const float* __restrict__ points,
                                                       const float* __restrict__ K, const float* __restrict__ T,
                                                       float* __restrict__ grid, int n, int H, int W, float eps) {
  const size_t hw = (size_t)H * W, total = (size_t)n * hw;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
    const int b = (int)(i / hw);
    const size_t p = i - (size_t)b * hw;
    const float* k = K + (size_t)b * 16;
    const float* t = T + (size_t)b * 16;
    float row[28], P[21];
#pragma unroll
    for (int r = 0; r < 12; ++r) row[r] = k[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) row[12 + r] = t[r];
    {
      const float* Kp = row;
      const float* Tp = row + 12;
#pragma unroll
      for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          float acc = Kp[i2 * 4 + 0] * Tp[j2];
          acc = acc + Kp[i2 * 4 + 1] * Tp[4 + j2];
          acc = acc + Kp[i2 * 4 + 2] * Tp[8 + j2];
          acc = acc + Kp[i2 * 4 + 3] * Tp[12 + j2];
          P[i2 * 4 + j2] = acc;
        }
    }
    const float* q = points + (size_t)b * 4 * hw + p;
    const float X = q[0], Y = q[hw], Z = q[2 * hw], Wc = q[3 * hw];
    const float qx = fmaf(P[3], Wc, fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)));
    const float qy = fmaf(P[7], Wc, fmaf(P[6], Z, fmaf(P[5], Y, P[4] * X)));
    const float qz = fmaf(P[11], Wc, fmaf(P[10], Z, fmaf(P[9], Y, P[8] * X)));
    const float zi = qz + eps;
    grid[i * 2 + 0] = ((qx / zi) / (float)(W - 1) - 0.5f) * 2.0f;
    grid[i * 2 + 1] = ((qy / zi) / (float)(H - 1) - 0.5f) * 2.0f;
  }
}

__global__ __launch_bounds__(NT) void ssim_map_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ out, int H, int W, int ntiles) {
  __shared__ __attribute__((aligned(16))) float s_y[3][FPLANE];
  __shared__ __attribute__((aligned(16))) float s_x[3][FPLANE];
  const int item = blockIdx.x / ntiles;
  const TileCoord tc = decode_tile(blockIdx.x - item * ntiles, W);
  const size_t hw = (size_t)H * W, img = 3 * hw;
  stage_image_tile(y + (size_t)item * img, H, W, tc.tx0, tc.ty0, s_y);
  stage_image_tile(x + (size_t)item * img, H, W, tc.tx0, tc.ty0, s_x);
  __syncthreads();
  const int ly = threadIdx.x / SPR, lx0 = (threadIdx.x % SPR) * PPT;
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  float mu_y[3][PPT], sg_y[3][PPT];
  strip_ystats(s_y, ly, lx0, mu_y, sg_y);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float xv[3][6], yv[3][6];
    load_window(s_x[ch], ly, lx0, xv);
    load_window(s_y[ch], ly, lx0, yv);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float sx = 0.0f, sxx = 0.0f, sxy = 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = xv[r][j + c];
          sx += v; sxx += v * v; sxy += v * yv[r][j + c];
        }
      if (yy < H && xx + j < W)
        out[(size_t)item * img + ch * hw + (size_t)yy * W + xx + j] = bbd_ssim(sx, sxx, sxy, mu_y[ch][j], sg_y[ch][j]);
    }
  }
}

int fill_frames(const void* const* frames, FramePtrs* out) {
  if (frames == nullptr) return BBD_E_BADARG;
  for (int i = 0; i < BBD_MAX_FRAME_SLOTS; ++i) out->base[i] = static_cast<const float*>(frames[i]);
  return 0;
}

int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int bbd_abi_version(void) { return BBD_ABI_VERSION; }
int bbd_tile_w(void) { return TW; }
int bbd_tile_h(void) { return TH; }
int bbd_num_tiles(int H, int W) { return ((H + TH - 1) / TH) * ((W + TW - 1) / TW); }

int bbd_identity_loss_fwd(const void* const* frames, const float* target, const int32_t* items, int NI,
                          float* ident, int H, int W, int no_ssim, void* stream) {
  if (!target || !items || !ident || NI < 0 || H < 3 || W < 3) return BBD_E_BADARG;
  if (NI == 0) return 0;
  FramePtrs fp;
  if (fill_frames(frames, &fp)) return BBD_E_BADARG;
  const int ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(identity_loss_kernel, dim3((unsigned)(NI * ntiles)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), fp, target, items, ident, H, W, ntiles, no_ssim);
  return launch_status();
}

int bbd_warp_ssim_min_fwd(const void* const* frames, const float* target, const float* depth, const float* proj,
                          const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                          float* min_loss, uint8_t* argmin, float* partial, float* warped, int S, int B, int NP,
                          int H, int W, int no_ssim, void* stream) {
  if (!target || !depth || !cand || !ncand || !min_loss || !argmin || !partial) return BBD_E_BADARG;
  if (S <= 0 || B <= 0 || H < 3 || W < 3 || NP < 0) return BBD_E_BADARG;
  FwdArgs a;
  if (fill_frames(frames, &a.frames)) return BBD_E_BADARG;
  a.target = target; a.depth = depth; a.proj = proj; a.ident = ident; a.noise = noise;
  a.cand = cand; a.ncand = ncand; a.min_loss = min_loss; a.argmin = argmin; a.partial = partial;
  a.warped = warped; a.S = S; a.B = B; a.NP = NP; a.H = H; a.W = W; a.no_ssim = no_ssim;
  a.ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(warp_ssim_min_fwd_kernel, dim3((unsigned)(S * B * a.ntiles)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), a);
  return launch_status();
}

int bbd_warp_ssim_min_bwd(const void* const* frames, const float* target, const float* depth, const float* proj,
                          const bbd_cand_t* cand, const int32_t* ncand, const uint8_t* argmin, const float* gscale,
                          float* grad_depth, float* grad_proj, int S, int B, int NP, int H, int W, int no_ssim,
                          void* stream) {
  if (!target || !depth || !cand || !ncand || !argmin || !gscale || !grad_depth || !grad_proj) return BBD_E_BADARG;
  if (S <= 0 || B <= 0 || H < 3 || W < 3 || NP < 0) return BBD_E_BADARG;
  BwdArgs a;
  if (fill_frames(frames, &a.frames)) return BBD_E_BADARG;
  a.target = target; a.depth = depth; a.proj = proj; a.cand = cand; a.ncand = ncand; a.argmin = argmin;
  a.gscale = gscale; a.grad_depth = grad_depth; a.grad_proj = grad_proj;
  a.S = S; a.B = B; a.NP = NP; a.H = H; a.W = W; a.no_ssim = no_ssim;
  a.ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(warp_ssim_min_bwd_kernel, dim3((unsigned)(S * B * a.ntiles)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), a);
  return launch_status();
}

int bbd_disp_to_depth_fwd(const float* disp, float* depth, int B, int h, int w, int H, int W, double min_depth,
                          double max_depth, void* stream) {
  if (!disp || !depth || B <= 0 || h <= 0 || w <= 0 || H < h || W < w) return BBD_E_BADARG;
  // layers.py:18-20 evaluates these in Python doubles before they meet the fp32 tensor
  const float lo = (float)(1.0 / max_depth), span = (float)(1.0 / min_depth - 1.0 / max_depth);
  const size_t n = (size_t)B * H * W;
  const unsigned grid = (unsigned)((n + NT - 1) / NT < 4096 ? (n + NT - 1) / NT : 4096);
  hipLaunchKernelGGL(disp_to_depth_fwd_kernel, dim3(grid), dim3(NT), 0, static_cast<hipStream_t>(stream), disp,
                     depth, B, h, w, H, W, lo, span);
  return launch_status();
}

int bbd_disp_to_depth_bwd(const float* disp, const float* grad_depth, float* grad_disp, int B, int h, int w, int H,
                          int W, double min_depth, double max_depth, void* stream) {
  if (!disp || !grad_depth || !grad_disp || B <= 0 || h <= 0 || w <= 0 || H < h || W < w) return BBD_E_BADARG;
  const float lo = (float)(1.0 / max_depth), span = (float)(1.0 / min_depth - 1.0 / max_depth);
  const size_t n = (size_t)B * h * w;
  const unsigned grid = (unsigned)((n + NT - 1) / NT < 4096 ? (n + NT - 1) / NT : 4096);
  hipLaunchKernelGGL(disp_to_depth_bwd_kernel, dim3(grid), dim3(NT), 0, static_cast<hipStream_t>(stream), disp,
                     grad_depth, grad_disp, B, h, w, H, W, lo, span);
  return launch_status();
}

int bbd_backproject_fwd(const float* depth, const float* inv_K, float* points, int n, int H, int W, void* stream) {
  if (!depth || !inv_K || !points || n <= 0 || H <= 0 || W <= 0) return BBD_E_BADARG;
  const size_t tot = (size_t)n * H * W;
  const unsigned grid = (unsigned)((tot + NT - 1) / NT < 4096 ? (tot + NT - 1) / NT : 4096);
  hipLaunchKernelGGL(backproject_kernel, dim3(grid), dim3(NT), 0, static_cast<hipStream_t>(stream), depth, inv_K,
                     points, n, H, W);
  return launch_status();
}

int bbd_project3d_fwd(const float* points, const float* K, const float* T, float* grid_out, int n, int H, int W,
                      double eps, void* stream) {
  if (!points || !K || !T || !grid_out || n <= 0 || H < 2 || W < 2) return BBD_E_BADARG;
  const size_t tot = (size_t)n * H * W;
  const unsigned grid = (unsigned)((tot + NT - 1) / NT < 4096 ? (tot + NT - 1) / NT : 4096);
  hipLaunchKernelGGL(project3d_kernel, dim3(grid), dim3(NT), 0, static_cast<hipStream_t>(stream), points, K, T,
                     grid_out, n, H, W, (float)eps);
  return launch_status();
}

int bbd_ssim_fwd(const float* x, const float* y, float* out, int n, int H, int W, void* stream) {
  if (!x || !y || !out || n <= 0 || H < 3 || W < 3) return BBD_E_BADARG;
  const int ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(ssim_map_kernel, dim3((unsigned)(n * ntiles)), dim3(NT), 0, static_cast<hipStream_t>(stream),
                     x, y, out, H, W, ntiles);
  return launch_status();
}

}  // extern "C"
