// bbd_eval.hip - validation-time depth metrics on the device.
//
// One launch scores a batch of predictions against ragged ground-truth depth maps:
//   trainer.py:572-617 (KITTI branch of Trainer.compute_depth_losses)    - flags 0
//   evaluate_depth.py:244-297 (+ compute_errors :57-69, KITTI branch)      - BBD_EVAL_PRED_IS_DISP |
//                                                                           BBD_EVAL_MEDIAN_MIDPOINT
// The reference does this per image with ~25 eager ops, a boolean gather and two sorts
// (torch.median / np.median).  Here one 1024-thread workgroup owns one image: the prediction is
// resampled on the fly at every ground-truth pixel of the crop window (never materialised), both
// medians come from an exact 3-level radix select on the float bit patterns (11+11+10 bits, integer
// LDS histograms => deterministic), and the seven error sums are accumulated in fp64 in one more
// sweep.  4 sweeps over <= 0.47 M pixels, ~2 MB of L2-resident reads per image; a validation set is
// scored n images per launch (n workgroups).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"
#include "bbd_math.h"

namespace {

constexpr int ET = 1024;           // threads per image
constexpr int EW = ET / 64;        // waves
constexpr int NBIN = 2048;

struct EvalArgs {
  const float* pred;       // [n,h,w] depth (or disparity with BBD_EVAL_PRED_IS_DISP)
  const float* gt;         // ragged ground truth, image i at gt + offset_i
  const int32_t* desc;     // [n, BBD_EVAL_DESC] offset_lo, offset_hi, GH, GW, r0, r1, c0, c1
  float* out;              // [n, BBD_EVAL_OUT]
  int h, w;
  float min_depth, max_depth, clamp_lo, clamp_hi, scale_factor;
  int flags;
};

__device__ __forceinline__ uint32_t order_key(float v) {   // monotone float -> uint
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Prediction at ground-truth pixel (y, x) of a GH x GW map.
__device__ __forceinline__ float resample(const EvalArgs& a, const float* img, int y, int x, int GH, int GW) {
  if (a.flags & BBD_EVAL_PRED_IS_DISP) {
    // cv2.resize(pred_disp, (gt_width, gt_height)) - INTER_LINEAR on float32 (evaluate_depth.py:248):
    // half-pixel centres, coordinate in double -> float, edge taps collapse to weight 0, horizontal
    // pass then vertical pass, products and sums rounded separately; then pred_depth = 1 / pred_disp.
    const double sx_ = (double)a.w / (double)GW, sy_ = (double)a.h / (double)GH;
    float fx = (float)(((double)x + 0.5) * sx_ - 0.5);
    float fy = (float)(((double)y + 0.5) * sy_ - 0.5);
    int ix = (int)floorf(fx), iy = (int)floorf(fy);
    fx -= (float)ix;
    fy -= (float)iy;
    if (ix < 0) { ix = 0; fx = 0.0f; }
    if (ix >= a.w - 1) { ix = a.w - 1; fx = 0.0f; }
    if (iy < 0) { iy = 0; fy = 0.0f; }
    if (iy >= a.h - 1) { iy = a.h - 1; fy = 0.0f; }
    const int ix1 = ix < a.w - 1 ? ix + 1 : ix, iy1 = iy < a.h - 1 ? iy + 1 : iy;
    const float* r0 = img + (size_t)iy * a.w;
    const float* r1 = img + (size_t)iy1 * a.w;
    const float top = r0[ix] * (1.0f - fx) + r0[ix1] * fx;
    const float bot = r1[ix] * (1.0f - fx) + r1[ix1] * fx;
    const float d = top * (1.0f - fy) + bot * fy;
    return (1.0f / d) * a.scale_factor;            // evaluate_depth.py:252, :275
  }
  // F.interpolate(depth_pred, [gt_h, gt_w], bilinear, align_corners=False) then clamp (trainer.py:599)
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  bbd_up_src(y, a.h, GH, &y0, &y1, &ly0, &ly1);
  bbd_up_src(x, a.w, GW, &x0, &x1, &lx0, &lx1);
  const float* r0 = img + (size_t)y0 * a.w;
  const float* r1 = img + (size_t)y1 * a.w;
  float v = bbd_up_blend(r0[x0], r0[x1], r1[x0], r1[x1], ly0, ly1, lx0, lx1, GH + GW <= 128);
  v = v < a.clamp_lo ? a.clamp_lo : v;             // torch.clamp: NaN propagates
  v = v > a.clamp_hi ? a.clamp_hi : v;
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(ET) void depth_metrics_kernel(EvalArgs a) {
  __shared__ uint32_t hist[4][NBIN];
  __shared__ uint32_t q_prefix[4], q_rank[4], s_count;
  __shared__ double red[EW][8];

  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* d = a.desc + (size_t)img * BBD_EVAL_DESC;
  const size_t off = (size_t)(uint32_t)d[0] | ((size_t)(uint32_t)d[1] << 32);
  const int GH = d[2], GW = d[3], r0 = d[4], r1 = d[5], c0 = d[6], c1 = d[7];
  const float* gt = a.gt + off;
  const float* pr = a.pred + (size_t)img * a.h * a.w;
  float* out = a.out + (size_t)img * BBD_EVAL_OUT;
  const int ww = c1 - c0, npx = (r1 - r0) * ww;

  if (tid < 4) { q_prefix[tid] = 0u; q_rank[tid] = 0u; }

  // ---- exact medians: 3-level radix select, 4 queries (gt lower/upper, pred lower/upper) ----
  for (int level = 0; level < 3; ++level) {
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const int prev_shift = level == 1 ? 21 : 10;
    const uint32_t mask = level == 2 ? 1023u : 2047u;
    for (int i = tid; i < 4 * NBIN; i += ET) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t p0 = q_prefix[0], p1 = q_prefix[1], p2 = q_prefix[2], p3 = q_prefix[3];
    for (int i = tid; i < npx; i += ET) {
      const int y = r0 + i / ww, x = c0 + i % ww;
      const float g = gt[(size_t)y * GW + x];
      if (!(g > a.min_depth && g < a.max_depth)) continue;
      const uint32_t kg = order_key(g), kp = order_key(resample(a, pr, y, x, GH, GW));
      if (level == 0) {
        atomicAdd(&hist[0][kg >> 21], 1u);
        atomicAdd(&hist[2][kp >> 21], 1u);
      } else {
        if ((kg >> prev_shift) == p0) atomicAdd(&hist[0][(kg >> shift) & mask], 1u);
        if ((kg >> prev_shift) == p1) atomicAdd(&hist[1][(kg >> shift) & mask], 1u);
        if ((kp >> prev_shift) == p2) atomicAdd(&hist[2][(kp >> shift) & mask], 1u);
        if ((kp >> prev_shift) == p3) atomicAdd(&hist[3][(kp >> shift) & mask], 1u);
      }
    }
    __syncthreads();
    if (wave < 4) {                     // wave q resolves query q
      const int q = wave;
      const uint32_t* hq = hist[level == 0 ? (q & 2) : q];
      const int per = NBIN / 64;
      uint32_t mine = 0;
      for (int j = 0; j < per; ++j) mine += hq[lane * per + j];
      uint32_t incl = mine;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const uint32_t total = __shfl(incl, 63, 64);
      uint32_t rank = q_rank[q];
      if (level == 0) {
        rank = (q & 1) ? total / 2u : (total ? (total - 1u) / 2u : 0u);
        if (q == 0 && lane == 0) s_count = total;
      }
      const uint32_t before = incl - mine;
      if (total > 0 && rank >= before && rank < incl) {
        uint32_t cum = before;
        int b = lane * per;
        for (; b < lane * per + per; ++b) {
          const uint32_t c = hq[b];
          if (rank < cum + c) break;
          cum += c;
        }
        q_prefix[q] = (level == 0 ? 0u : (q_prefix[q] << (level == 2 ? 10 : 11))) | (uint32_t)b;
        q_rank[q] = rank - cum;
      }
    }
    __syncthreads();
  }

  const uint32_t count = s_count;
  if (count == 0) {                     // reference: median of an empty tensor raises; here: NaN row
    if (tid < BBD_EVAL_OUT) out[tid] = tid == 10 ? 0.0f : __uint_as_float(0x7fc00000u);
    return;
  }
  float med_gt = key_value(q_prefix[0]), med_pr = key_value(q_prefix[2]);
  if (a.flags & BBD_EVAL_MEDIAN_MIDPOINT) {      // np.median: mean of the two middle values
    med_gt = (med_gt + key_value(q_prefix[1])) / 2.0f;
    med_pr = (med_pr + key_value(q_prefix[3])) / 2.0f;
  }
  const float ratio = (a.flags & BBD_EVAL_NO_MEDIAN_SCALING) ? 1.0f : med_gt / med_pr;

  // ---- the seven metrics (layers.py:271-286 / evaluate_depth.py:57-72) ----
  double s_abs = 0, s_sq = 0, s_d2 = 0, s_l2 = 0, s_a1 = 0, s_a2 = 0, s_a3 = 0;
  for (int i = tid; i < npx; i += ET) {
    const int y = r0 + i / ww, x = c0 + i % ww;
    const float g = gt[(size_t)y * GW + x];
    if (!(g > a.min_depth && g < a.max_depth)) continue;
    float p = resample(a, pr, y, x, GH, GW);
    if (!(a.flags & BBD_EVAL_NO_MEDIAN_SCALING)) p *= ratio;
    p = p < a.min_depth ? a.min_depth : p;
    p = p > a.max_depth ? a.max_depth : p;
    const float t0 = g / p, t1 = p / g;
    const float th = t0 > t1 ? t0 : t1;
    s_a1 += th < 1.25f ? 1.0 : 0.0;
    s_a2 += th < 1.5625f ? 1.0 : 0.0;             // 1.25 ** 2
    s_a3 += th < 1.953125f ? 1.0 : 0.0;           // 1.25 ** 3
    const float df = g - p;
    const float d2 = df * df;
    const float dl = logf(g) - logf(p);
    s_d2 += (double)d2;
    s_l2 += (double)(dl * dl);
    s_abs += (double)(fabsf(df) / g);
    s_sq += (double)(d2 / g);
  }
  double v[7] = {s_abs, s_sq, s_d2, s_l2, s_a1, s_a2, s_a3};
  for (int k = 0; k < 7; ++k) {
    const double r = wave_sum(v[k]);
    if (lane == 0) red[wave][k] = r;
  }
  __syncthreads();
  if (tid < 7) {
    double s = 0;
    for (int wv = 0; wv < EW; ++wv) s += red[wv][tid];
    double m = s / (double)count;
    if (tid == 2 || tid == 3) m = sqrt(m);
    out[tid] = (float)m;
  }
  if (tid == 7) out[7] = ratio;
  if (tid == 8) out[8] = med_gt;
  if (tid == 9) out[9] = med_pr;
  if (tid == 10) out[10] = (float)count;
  if (tid == 11) out[11] = 0.0f;
}

}  // namespace

extern "C" int bbd_depth_metrics(const float* pred, const float* gt, const int32_t* desc, float* out, int n, int h,
                                 int w, double min_depth, double max_depth, double clamp_lo, double clamp_hi,
                                 double scale_factor, int flags, void* stream) {
  if (!pred || !gt || !desc || !out || n <= 0 || h < 1 || w < 1) return BBD_E_BADARG;
  EvalArgs a;
  a.pred = pred; a.gt = gt; a.desc = desc; a.out = out; a.h = h; a.w = w;
  a.min_depth = (float)min_depth; a.max_depth = (float)max_depth;
  a.clamp_lo = (float)clamp_lo; a.clamp_hi = (float)clamp_hi;
  a.scale_factor = (float)scale_factor; a.flags = flags;
  hipLaunchKernelGGL(depth_metrics_kernel, dim3((unsigned)n), dim3(ET), 0, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
