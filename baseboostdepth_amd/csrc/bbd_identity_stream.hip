// bbd_identity_stream.hip - the identity pre-pass (trainer.py:501-508: compute_reprojection_loss of every UN-warped source
// against the target) as a streaming stencil: no LDS, no barrier.
//
// Round 5 measured two streaming forms slower than the tiled, LDS-staged kernel (profiles/r05/identity_forms.txt) and named
// what they lacked; this is that form (round 6):
//   * ALIGNED 8-byte loads: a lane owns the two columns (c, c + 1), c even, of a 128-column band - one float2 per row,
//     channel and image, no lane overlaps another (round 5's 16-byte loads were 4-byte aligned and overlapped pairwise);
//   * the two HALO columns c - 1 and c + 2 come from the neighbour lanes over the DPP crossbar (wave_shr:1 / wave_shl:1), so a
//     band yields 126 output columns (its first and last loaded column are halo only) and bands advance by 126 columns - still
//     8-byte aligned.  Reflection at the image's left / right edge costs nothing: the load address is clamped to [0, W - 2],
//     and the clamped pair's elements ARE the reflected columns (col -1 -> col 1, col W -> col W - 2; W even);
//   * the target's window statistics are computed once per row and shared by TWO identity candidates of the sample marched
//     together (MD2: frames -1 and +1 are one march; the boosted recipe's three to six candidates are pairs);
//   * the row window rotates by loop unrolling (four register rows, four steps with permuted roles: three rows are evaluated
//     while the fourth receives the next image row), not by moves;
//   * the two pixels of a lane are the two halves of packed registers: the 3x3 sums (the bulk of the arithmetic) are
//     v_pk_add_f32 / v_pk_mul_f32, in the reference's row-major order per component - the same bits as the tiled form
//     (tests/test_gpu_parity.py::test_identity_pass_forms_agree_on_ragged_sizes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"
#include "bbd_math.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

struct FramePtrsS {
  const float* base[BBD_MAX_FRAME_SLOTS];
};

constexpr int SBAND = 126;           // output columns per wave (128 loaded)

__device__ __forceinline__ int uload(const int32_t* p) { return __builtin_amdgcn_readfirstlane(*p); }

// value of the lane below / above on the DPP crossbar (lane 0 / lane 63 get their own value: they are halo lanes)
__device__ __forceinline__ float from_lane_below(float v) {        // lane i <- lane i - 1   (wave_shr:1)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v) {        // lane i <- lane i + 1   (wave_shl:1)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

struct Row3 {                        // one image row of a lane: its own two columns, three channels
  v2f c[3];
};

__device__ __forceinline__ Row3 load_row(const float* __restrict__ img, int hw, int off) {
  Row3 r;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) r.c[ch] = *reinterpret_cast<const v2f*>(img + ch * hw + off);
  return r;
}

// the three column pairs of a row for the lane's two pixels: (c-1, c), (c, c+1), (c+1, c+2)
struct Taps {
  v2f p[3];
};
__device__ __forceinline__ Taps taps(const v2f own) {
  Taps t;
  t.p[0].x = from_lane_below(own.y);
  t.p[0].y = own.x;
  t.p[1] = own;
  t.p[2].x = own.y;
  t.p[2].y = from_lane_above(own.x);
  return t;
}

// ---- the per-pixel SSIM / L1 tail on the lane's two pixels as the two halves of packed registers: the SAME operations, in
// the same order, per component as bbd_ystats / bbd_ssim_nd / bbd_div / bbd_ssim_from_ratio / bbd_combine (bbd_math.h)
__device__ __forceinline__ v2f pk(float a) { v2f r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f div9_2(v2f x) {           // bbd_div9
  const v2f r = pk(1.0f / 9.0f);
  const v2f q = x * r;
  return pfma(pfma(pk(-9.0f), q, x), r, q);
}
__device__ __forceinline__ v2f div3_2(v2f x) {           // bbd_div3
  const v2f r = pk(1.0f / 3.0f);
  const v2f q = x * r;
  return pfma(pfma(pk(-3.0f), q, x), r, q);
}
__device__ __forceinline__ v2f div_2(v2f n, v2f d) {     // bbd_div: the refined-reciprocal sequence inside the exponent window
  if (bbd_exp_ok3(n.x, d.x, d.x) && bbd_exp_ok3(n.y, d.y, d.y)) {
    v2f r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    r = pfma(pfma(-d, r, pk(1.0f)), r, r);               // bbd_rcp_refined
    v2f q = n * r;                                       // bbd_div_with
    q = pfma(pfma(-d, q, n), r, q);
    return pfma(pfma(-d, q, n), r, q);
  }
  v2f q;
  q.x = bbd_div(n.x, d.x);
  q.y = bbd_div(n.y, d.y);
  return q;
}
__device__ __forceinline__ v2f clamp01_2(v2f v) {        // torch.clamp(., 0, 1): NaN stays NaN
  v2f o;
  o.x = v.x < 0.0f ? 0.0f : (v.x > 1.0f ? 1.0f : v.x);
  o.y = v.y < 0.0f ? 0.0f : (v.y > 1.0f ? 1.0f : v.y);
  return o;
}

// One output row: window rows (top, mid, bot) of the target and of NS sources -> the lane's two loss values per source.
template <int NS>
__device__ __forceinline__ void eval_row(const Row3& ty, const Row3& my, const Row3& by, const Row3 (&tx)[NS],
                                         const Row3 (&mx)[NS], const Row3 (&bx)[NS], int no_ssim, v2f (&out)[NS]) {
  v2f ssim[NS][3], l1[NS][3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const Taps yt[3] = {taps(ty.c[ch]), taps(my.c[ch]), taps(by.c[ch])};
    v2f sy = {0.0f, 0.0f}, syy = {0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        sy = sy + yt[r].p[c];
        syy = syy + yt[r].p[c] * yt[r].p[c];
      }
    const v2f mu_y = div9_2(sy);                                       // bbd_ystats
    const v2f myy = mu_y * mu_y;
    const v2f sg_y = div9_2(syy) - myy;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const Taps xt[3] = {taps(tx[s].c[ch]), taps(mx[s].c[ch]), taps(bx[s].c[ch])};
      v2f sx = {0.0f, 0.0f}, sxx = {0.0f, 0.0f}, sxy = {0.0f, 0.0f};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const v2f v = xt[r].p[c];
          sx = sx + v;
          sxx = sxx + v * v;
          sxy = sxy + v * yt[r].p[c];
        }
      if (!no_ssim) {
        const v2f mu_x = div9_2(sx);                                   // bbd_ssim_nd
        const v2f mxx = mu_x * mu_x, mxy = mu_x * mu_y;
        const v2f sig_x = div9_2(sxx) - mxx;
        const v2f sig_xy = div9_2(sxy) - mxy;
        const v2f n = pfma(mxy, pk(2.0f), pk(BBD_C1)) * pfma(sig_xy, pk(2.0f), pk(BBD_C2));
        const v2f d = (mxx + myy + pk(BBD_C1)) * (sig_x + sg_y + pk(BBD_C2));
        const v2f q = div_2(n, d);
        ssim[s][ch] = clamp01_2((pk(1.0f) - q) / pk(2.0f));            // bbd_ssim_from_ratio
      } else {
        ssim[s][ch] = pk(0.0f);
      }
      const v2f df = my.c[ch] - mx[s].c[ch];
      l1[s][ch].x = fabsf(df.x);
      l1[s][ch].y = fabsf(df.y);
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {                                       // bbd_combine
    const v2f l1m = div3_2(l1[s][0] + l1[s][1] + l1[s][2]);
    if (no_ssim) { out[s] = l1m; continue; }
    const v2f sm = div3_2(ssim[s][0] + ssim[s][1] + ssim[s][2]);
    out[s] = pk(0.85f) * sm + pk(0.15f) * l1m;
  }
}

template <int NS>
__device__ __forceinline__ void march(const float* __restrict__ tg, const float* const (&sr)[NS], float* const (&out)[NS],
                                      int H, int W, int col, int y0, int rows, bool st0, bool st1, int no_ssim) {
  const int hw = H * W;
  // FOUR register rows: step k evaluates rows (k, k + 1, k + 2) mod 4 while the image row after them travels into row
  // (k + 3) mod 4 - the window rotates through the four unrolled steps, nothing is moved
  Row3 wy[4], wx[4][NS];
  auto fetch = [&](int slot, int y) {      // image row y (reflected; beyond the image: a valid row nobody uses) -> register row
    const int off = bbd_reflect(y < H + 1 ? y : H, H) * W + col;
    wy[slot] = load_row(tg, hw, off);
#pragma unroll
    for (int s = 0; s < NS; ++s) wx[slot][s] = load_row(sr[s], hw, off);
  };
  auto emit = [&](int y, const v2f (&o)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float* p = out[s] + (size_t)y * W + col;
      if (st0 && st1) *reinterpret_cast<v2f*>(p) = o[s];
      else if (st0) p[0] = o[s].x;
      else if (st1) p[1] = o[s].y;
    }
  };
  fetch(0, y0 - 1);
  fetch(1, y0);
  fetch(2, y0 + 1);
  for (int i = 0; i < rows; i += 4) {
    v2f o[NS];
    fetch(3, y0 + i + 2);
    eval_row<NS>(wy[0], wy[1], wy[2], wx[0], wx[1], wx[2], no_ssim, o);
    emit(y0 + i, o);
    if (i + 1 >= rows) break;
    fetch(0, y0 + i + 3);
    eval_row<NS>(wy[1], wy[2], wy[3], wx[1], wx[2], wx[3], no_ssim, o);
    emit(y0 + i + 1, o);
    if (i + 2 >= rows) break;
    fetch(1, y0 + i + 4);
    eval_row<NS>(wy[2], wy[3], wy[0], wx[2], wx[3], wx[0], no_ssim, o);
    emit(y0 + i + 2, o);
    if (i + 3 >= rows) break;
    fetch(2, y0 + i + 5);
    eval_row<NS>(wy[3], wy[0], wy[1], wx[3], wx[0], wx[1], no_ssim, o);
    emit(y0 + i + 3, o);
  }
}

// grid: blocks of 4 waves; a wave = (group, pair of the group's items, row chunk, band)
__global__ __launch_bounds__(256) void identity_stream_kernel(FramePtrsS frames, const float* __restrict__ target,
                                                              const int32_t* __restrict__ items,
                                                              const int32_t* __restrict__ group_off, float* __restrict__ ident,
                                                              int H, int W, int nbands, int nchunks, int rows_per_wave,
                                                              int max_pairs, int waves_per_group, int no_ssim) {
  const int wave = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  const int grp = wave / waves_per_group;
  int rem = wave - grp * waves_per_group;
  const int pair = rem / (nbands * nchunks);
  rem -= pair * nbands * nchunks;
  const int chunk = rem / nbands, band = rem - chunk * nbands;
  const int i0 = uload(group_off + grp), i1 = uload(group_off + grp + 1);
  const int ia = i0 + 2 * pair;
  if (pair >= max_pairs || ia >= i1) return;
  const bool two = ia + 1 < i1;
  const int b = uload(items + ia * 4 + 0);
  const int hw = H * W;
  const size_t img = (size_t)3 * hw;
  const float* tg = target + (size_t)b * img;
  const float* s0 = frames.base[uload(items + ia * 4 + 1)] + (size_t)uload(items + ia * 4 + 2) * img;
  const int lane = (int)threadIdx.x & 63;
  const int c = SBAND * band - 2 + 2 * lane;                          // the lane's first column (even)
  const int col = c < 0 ? 0 : (c > W - 2 ? W - 2 : c);                // clamped load column: its pair holds the reflections
  const bool st0 = lane >= 1 && c >= 0 && c < W;                      // pixel c needs column c - 1 from the lane below
  const bool st1 = lane <= 62 && c + 1 >= 0 && c + 1 < W;             // pixel c + 1 needs column c + 2 from the lane above
  const int y0 = chunk * rows_per_wave;
  const int rows = (H - y0) < rows_per_wave ? (H - y0) : rows_per_wave;
  if (rows <= 0) return;
  if (two) {
    const float* s1 = frames.base[uload(items + (ia + 1) * 4 + 1)] + (size_t)uload(items + (ia + 1) * 4 + 2) * img;
    const float* const sr[2] = {s0, s1};
    float* const out[2] = {ident + (size_t)ia * hw, ident + (size_t)(ia + 1) * hw};
    march<2>(tg, sr, out, H, W, col, y0, rows, st0, st1, no_ssim);
  } else {
    const float* const sr[1] = {s0};
    float* const out[1] = {ident + (size_t)ia * hw};
    march<1>(tg, sr, out, H, W, col, y0, rows, st0, st1, no_ssim);
  }
}

}  // namespace

extern "C" int bbd_identity_loss_stream_supported(int H, int W) { return (H >= 3 && W >= 4 && (W & 1) == 0) ? 1 : 0; }

extern "C" int bbd_identity_loss_stream_fwd(const void* const* frames, const float* target, const int32_t* items,
                                            const int32_t* group_off, int G, int max_items, float* ident, int H, int W,
                                            int no_ssim, int rows_per_wave, void* stream) {
  if (!frames || !target || !items || !group_off || !ident || G < 0 || max_items < 0) return BBD_E_BADARG;
  if (!bbd_identity_loss_stream_supported(H, W)) return BBD_E_BADARG;
  if (G == 0 || max_items == 0) return 0;
  FramePtrsS fp;
  for (int i = 0; i < BBD_MAX_FRAME_SLOTS; ++i) fp.base[i] = static_cast<const float*>(frames[i]);
  if (rows_per_wave <= 0) rows_per_wave = 8;
  rows_per_wave = (rows_per_wave + 3) / 4 * 4;
  const int nbands = (W + 1 + SBAND - 1) / SBAND;                     // output columns -1 .. W - 1 in steps of 126
  const int nchunks = (H + rows_per_wave - 1) / rows_per_wave;
  const int max_pairs = (max_items + 1) / 2;
  const int waves_per_group = max_pairs * nbands * nchunks;
  const long long waves = (long long)G * waves_per_group;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  hipLaunchKernelGGL(identity_stream_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), fp, target, items,
                     group_off, ident, H, W, nbands, nchunks, rows_per_wave, max_pairs, waves_per_group, no_ssim);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
