// bbd_nn.hip - fused training-mode BatchNorm2d (+ residual add) (+ ReLU), forward and backward.
//
// The encoders are torchvision-layout ResNets (networks/resnet_encoder.py:12-91): every convolution is
// followed by BatchNorm2d, then either ReLU or "+ identity, ReLU".  In eager PyTorch-ROCm that is one
// MIOpen batch-norm launch plus separate add / clamp / threshold kernels, each a full round trip over
// the activation (profiles/r01/bench_md2_one_steady_step_v6.csv: 2.6 ms BN + ~1.3 ms add/ReLU per step).
// Here the tail of a block is two launches each way over NCHW fp32:
//
//   forward : stats   per channel sum / sum of squares (fp64 partials, fixed-order combine)
//             apply   y = relu(gamma * (x - mean) * invstd + beta + residual); running stats updated
//   backward: reduce  g = dy * (y > 0);  per channel sum(g), sum(g * xhat)
//             apply   dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));  dresidual = g;
//                     dgamma = sum(g * xhat), dbeta = sum(g)
//
// Same definition as torch.nn.functional.batch_norm(training=True): biased variance for the
// normalisation, unbiased for running_var, momentum update of the running statistics.  HBM-bound
// elementwise / reduction work: coalesced float4 accesses of each (n, c) plane, one workgroup per
// (channel, slice of the plane), deterministic (no floating-point atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_SPLIT = 64;
#ifndef BN_SMALL_ELEMS
#define BN_SMALL_ELEMS 8192     // per (channel, call group): at or below this the two passes of a direction share one launch
#endif

struct BnArgs {
  const float* x;        // [N,C,HW] conv output
  const float* res;      // residual or NULL
  const float* gamma;    // [C]
  const float* beta;     // [C]
  float* y;              // [N,C,HW]
  double* part;          // [C, split, 2] scratch
  float* mean;           // [C] saved for backward
  float* invstd;         // [C]
  float* run_mean;       // [C] or NULL
  float* run_var;        // [C] or NULL
  long long* batches;    // num_batches_tracked (int64 scalar) or NULL
  // backward
  const float* dy;
  float* dx;
  float* dres;           // or NULL
  float* dgamma;
  float* dbeta;
  int N, C, HW, split, relu;
  float eps, momentum;
  // call groups: samples [rows[g], rows[g+1]) are normalised with their own statistics, as if each group had been
  // a separate call of the layer (one batched pass of a network that the reference calls G times); blockIdx.z = g
  int G;
  int untracked;         // the last `untracked` groups are normalised like the others but leave the running statistics alone
  // device-resident group table (bbd_bn_act_grouped_dev_*): rows[0..G] then, at [BBD_BN_MAX_GROUPS + 1], the number of
  // tracked groups - a launch whose arguments do not depend on the batch signature, so that ONE captured step graph
  // serves every ordering with the same padded row count.  Groups may be empty there.  NULL: the by-value table below
  const int32_t* rows_dev;
  int rows[BBD_BN_MAX_GROUPS + 1];
};

__device__ __forceinline__ int group_row(const BnArgs& a, int g) { return a.rows_dev ? a.rows_dev[g] : a.rows[g]; }
__device__ __forceinline__ int tracked_groups(const BnArgs& a) {
  return a.rows_dev ? a.rows_dev[BBD_BN_MAX_GROUPS + 1] : a.G - a.untracked;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// number of plane slices for a batch of N images (host: grid size for the largest group; device: each group's own, so
// that a group is reduced exactly as a separate call on that sub-batch would reduce it)
__host__ __device__ inline int pick_split(int N, int HW) {
  long long per = (long long)N * HW;
  int s = (int)((per + 4095) / 4096);
  if (s < 1) s = 1;
  if (s > MAX_SPLIT) s = MAX_SPLIT;
  const int max_by_plane = (HW + 3) / 4;     // at least one float4 of the plane per slice
  return s < max_by_plane ? s : max_by_plane;
}

// slice k of `split` of the HW plane: [lo, hi), multiples of 4 when HW % 4 == 0
__device__ __forceinline__ void plane_slice(int HW, int split, int k, int* lo, int* hi) {
  int len = (HW + split - 1) / split;
  len = (len + 3) & ~3;
  *lo = k * len;
  *hi = min(HW, *lo + len);
}

// The four passes below are written per (channel c, call group grp, plane slice k).  Large activations launch one
// workgroup per slice and pass the per-slice sums through `part` (two launches per direction); small ones (a whole
// (channel, group) fits one workgroup comfortably) run both passes in ONE launch, the workgroup looping over the same
// slices and adding their block totals in the same order - identical numbers, half the launches, no partial round trip.
// Thread layout: 64 lanes sweep the slice, the 4 waves take every 4th image - a slice is only ~100 float4 long, so 256
// lanes along it left most of them idle with one dependent load per image in flight.

// forward statistics of slice k: block totals of x and x^2
__device__ __forceinline__ void stats_slice(const BnArgs& a, int c, int grp, int split, int k, double* sh, double* ts,
                                            double* tss) {
  int lo, hi;
  plane_slice(a.HW, split, k, &lo, &hi);
  float s = 0.0f, ss = 0.0f;
  double ds = 0.0, dss = 0.0;
  const bool vec = (a.HW & 3) == 0;
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  for (int n = group_row(a, grp) + sub; n < group_row(a, grp + 1); n += NT / 64) {
    const float* p = a.x + ((size_t)n * a.C + c) * a.HW;
    if (vec) {
      for (int i = lo + 4 * lane; i < hi; i += 4 * 64) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        s += (v.x + v.y) + (v.z + v.w);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    } else {
      for (int i = lo + lane; i < hi; i += 64) {
        const float v = p[i];
        s += v;
        ss += v * v;
      }
    }
    ds += (double)s; dss += (double)ss;     // short fp32 runs, fp64 across images
    s = 0.0f; ss = 0.0f;
  }
  *ts = block_sum(ds, sh);
  *tss = block_sum(dss, sh);
}

// mean / invstd of one (channel, group) from its totals
__device__ __forceinline__ void group_moments(const BnArgs& a, int grp, double ts, double tss, double* mean, double* var) {
  const double cnt = (double)(group_row(a, grp + 1) - group_row(a, grp)) * (double)a.HW;
  if (!(cnt > 0.0)) { *mean = 0.0; *var = 0.0; return; }      // an empty group of a device-resident table
  *mean = ts / cnt;
  double v = tss / cnt - (*mean) * (*mean);
  *var = v > 0.0 ? v : 0.0;
}
// running statistics: one momentum update per group, in group order = the order of the calls replaced.
// totals(q, &s, &ss) yields group q's sums.
template <typename Totals>
__device__ __forceinline__ void update_running(const BnArgs& a, int c, Totals totals) {
  if (a.run_mean) {
    float rm = a.run_mean[c], rv = a.run_var[c];
    const int tracked = tracked_groups(a);
    for (int q = 0; q < tracked; ++q) {
      double qs, qss, qmean, qvar;
      totals(q, &qs, &qss);
      group_moments(a, q, qs, qss, &qmean, &qvar);
      const double qcnt = (double)(group_row(a, q + 1) - group_row(a, q)) * (double)a.HW;
      const double unbiased = qcnt > 1.0 ? qvar * qcnt / (qcnt - 1.0) : qvar;
      rm = (1.0f - a.momentum) * rm + a.momentum * (float)qmean;
      rv = (1.0f - a.momentum) * rv + a.momentum * (float)unbiased;
    }
    a.run_mean[c] = rm;
    a.run_var[c] = rv;
  }
  if (a.batches && c == 0) *a.batches += tracked_groups(a);
}

// forward apply of slice k
__device__ __forceinline__ void apply_slice(const BnArgs& a, int c, int grp, int split, int k, float mean, float scale,
                                            float shift) {
  int lo, hi;
  plane_slice(a.HW, split, k, &lo, &hi);
  const bool vec = (a.HW & 3) == 0;
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  for (int n = group_row(a, grp) + sub; n < group_row(a, grp + 1); n += NT / 64) {
    const size_t base = ((size_t)n * a.C + c) * a.HW;
    if (vec) {
      for (int i = lo + 4 * lane; i < hi; i += 4 * 64) {
        const float4 v = *reinterpret_cast<const float4*>(a.x + base + i);
        float4 o;
        o.x = (v.x - mean) * scale + shift; o.y = (v.y - mean) * scale + shift;
        o.z = (v.z - mean) * scale + shift; o.w = (v.w - mean) * scale + shift;
        if (a.res) {
          const float4 r = *reinterpret_cast<const float4*>(a.res + base + i);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (a.relu) {
          o.x = o.x > 0.0f ? o.x : 0.0f; o.y = o.y > 0.0f ? o.y : 0.0f;
          o.z = o.z > 0.0f ? o.z : 0.0f; o.w = o.w > 0.0f ? o.w : 0.0f;
        }
        *reinterpret_cast<float4*>(a.y + base + i) = o;
      }
    } else {
      for (int i = lo + lane; i < hi; i += 64) {
        float o = (a.x[base + i] - mean) * scale + shift;
        if (a.res) o += a.res[base + i];
        if (a.relu) o = o > 0.0f ? o : 0.0f;
        a.y[base + i] = o;
      }
    }
  }
}

// backward reduction of slice k: block totals of g = dy * (y > 0) and g * xhat
__device__ __forceinline__ void bwd_reduce_slice(const BnArgs& a, int c, int grp, int split, int k, double* sh, double* tg,
                                                 double* tgx) {
  int lo, hi;
  plane_slice(a.HW, split, k, &lo, &hi);
  const float mean = a.mean[(size_t)grp * a.C + c], invstd = a.invstd[(size_t)grp * a.C + c];
  // ReLU mask without the saved output (no residual in the forward): apply_slice's own expression on x
  const bool remask = a.relu && a.y == nullptr;
  const float scale = remask ? a.gamma[c] * invstd : 0.0f, shift = remask ? a.beta[c] : 0.0f;
  double dg = 0.0, dgx = 0.0;
  const bool vec = (a.HW & 3) == 0;
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  for (int n = group_row(a, grp) + sub; n < group_row(a, grp + 1); n += NT / 64) {
    const size_t base = ((size_t)n * a.C + c) * a.HW;
    float sg = 0.0f, sgx = 0.0f;
    if (vec) {
      for (int i = lo + 4 * lane; i < hi; i += 4 * 64) {
        float4 g = *reinterpret_cast<const float4*>(a.dy + base + i);
        const float4 x = *reinterpret_cast<const float4*>(a.x + base + i);
        if (a.relu) {
          float4 y;
          if (remask) {
            y.x = (x.x - mean) * scale + shift; y.y = (x.y - mean) * scale + shift;
            y.z = (x.z - mean) * scale + shift; y.w = (x.w - mean) * scale + shift;
          } else {
            y = *reinterpret_cast<const float4*>(a.y + base + i);
          }
          g.x = y.x > 0.0f ? g.x : 0.0f; g.y = y.y > 0.0f ? g.y : 0.0f;
          g.z = y.z > 0.0f ? g.z : 0.0f; g.w = y.w > 0.0f ? g.w : 0.0f;
        }
        sg += (g.x + g.y) + (g.z + g.w);
        sgx += (g.x * ((x.x - mean) * invstd) + g.y * ((x.y - mean) * invstd)) +
               (g.z * ((x.z - mean) * invstd) + g.w * ((x.w - mean) * invstd));
      }
    } else {
      for (int i = lo + lane; i < hi; i += 64) {
        float g = a.dy[base + i];
        if (a.relu && !((remask ? (a.x[base + i] - mean) * scale + shift : a.y[base + i]) > 0.0f)) g = 0.0f;
        sg += g;
        sgx += g * ((a.x[base + i] - mean) * invstd);
      }
    }
    dg += (double)sg;
    dgx += (double)sgx;
  }
  *tg = block_sum(dg, sh);
  *tgx = block_sum(dgx, sh);
}

// backward apply of slice k
__device__ __forceinline__ void bwd_apply_slice(const BnArgs& a, int c, int grp, int split, int k, float mg, float mgx,
                                                float kk) {
  const float mean = a.mean[(size_t)grp * a.C + c], invstd = a.invstd[(size_t)grp * a.C + c];
  const bool remask = a.relu && a.y == nullptr;
  const float scale = remask ? a.gamma[c] * invstd : 0.0f, shift = remask ? a.beta[c] : 0.0f;
  int lo, hi;
  plane_slice(a.HW, split, k, &lo, &hi);
  const bool vec = (a.HW & 3) == 0;
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  for (int n = group_row(a, grp) + sub; n < group_row(a, grp + 1); n += NT / 64) {
    const size_t base = ((size_t)n * a.C + c) * a.HW;
    if (vec) {
      for (int i = lo + 4 * lane; i < hi; i += 4 * 64) {
        float4 g = *reinterpret_cast<const float4*>(a.dy + base + i);
        const float4 x = *reinterpret_cast<const float4*>(a.x + base + i);
        if (a.relu) {
          float4 y;
          if (remask) {
            y.x = (x.x - mean) * scale + shift; y.y = (x.y - mean) * scale + shift;
            y.z = (x.z - mean) * scale + shift; y.w = (x.w - mean) * scale + shift;
          } else {
            y = *reinterpret_cast<const float4*>(a.y + base + i);
          }
          g.x = y.x > 0.0f ? g.x : 0.0f; g.y = y.y > 0.0f ? g.y : 0.0f;
          g.z = y.z > 0.0f ? g.z : 0.0f; g.w = y.w > 0.0f ? g.w : 0.0f;
        }
        if (a.dres) *reinterpret_cast<float4*>(a.dres + base + i) = g;
        float4 o;
        o.x = kk * (g.x - mg - ((x.x - mean) * invstd) * mgx);
        o.y = kk * (g.y - mg - ((x.y - mean) * invstd) * mgx);
        o.z = kk * (g.z - mg - ((x.z - mean) * invstd) * mgx);
        o.w = kk * (g.w - mg - ((x.w - mean) * invstd) * mgx);
        *reinterpret_cast<float4*>(a.dx + base + i) = o;
      }
    } else {
      for (int i = lo + lane; i < hi; i += 64) {
        float g = a.dy[base + i];
        if (a.relu && !((remask ? (a.x[base + i] - mean) * scale + shift : a.y[base + i]) > 0.0f)) g = 0.0f;
        if (a.dres) a.dres[base + i] = g;
        a.dx[base + i] = kk * (g - mg - ((a.x[base + i] - mean) * invstd) * mgx);
      }
    }
  }
}

// per-(channel, group) sums of the two partial columns written by the two-launch reduction kernels
__device__ __forceinline__ void group_totals(const BnArgs& a, int c, int grp, double* t0, double* t1) {
  double s0 = 0.0, s1 = 0.0;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  for (int k = 0; k < split; ++k) {
    s0 += a.part[(((size_t)c * a.G + grp) * a.split + k) * 2];
    s1 += a.part[(((size_t)c * a.G + grp) * a.split + k) * 2 + 1];
  }
  *t0 = s0; *t1 = s1;
}

// ---- two launches per direction: grid (slices of the largest group, C, G)
__global__ __launch_bounds__(NT) void bn_stats_kernel(BnArgs a) {
  __shared__ double sh[4];
  const int c = blockIdx.y, grp = blockIdx.z;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  if ((int)blockIdx.x >= split) return;      // the grid is sized for the largest group
  double ts, tss;
  stats_slice(a, c, grp, split, blockIdx.x, sh, &ts, &tss);
  if (threadIdx.x == 0) {
    a.part[(((size_t)c * a.G + grp) * a.split + blockIdx.x) * 2] = ts;
    a.part[(((size_t)c * a.G + grp) * a.split + blockIdx.x) * 2 + 1] = tss;
  }
}

__global__ __launch_bounds__(NT) void bn_apply_kernel(BnArgs a) {
  __shared__ float s_mean, s_scale, s_shift;
  const int c = blockIdx.y, grp = blockIdx.z;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  if ((int)blockIdx.x >= split) return;
  if (threadIdx.x == 0) {
    double ts, tss, mean, var;
    group_totals(a, c, grp, &ts, &tss);
    group_moments(a, grp, ts, tss, &mean, &var);
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    s_mean = (float)mean;
    s_scale = a.gamma[c] * invstd;
    s_shift = a.beta[c];
    if (blockIdx.x == 0) {
      a.mean[(size_t)grp * a.C + c] = (float)mean;
      a.invstd[(size_t)grp * a.C + c] = invstd;
      if (grp == 0) update_running(a, c, [&](int q, double* s0, double* s1) { group_totals(a, c, q, s0, s1); });
    }
  }
  __syncthreads();
  apply_slice(a, c, grp, split, blockIdx.x, s_mean, s_scale, s_shift);
}

__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(BnArgs a) {
  __shared__ double sh[4];
  const int c = blockIdx.y, grp = blockIdx.z;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  if ((int)blockIdx.x >= split) return;
  double tg, tgx;
  bwd_reduce_slice(a, c, grp, split, blockIdx.x, sh, &tg, &tgx);
  if (threadIdx.x == 0) {
    a.part[(((size_t)c * a.G + grp) * a.split + blockIdx.x) * 2] = tg;
    a.part[(((size_t)c * a.G + grp) * a.split + blockIdx.x) * 2 + 1] = tgx;
  }
}

__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(BnArgs a) {
  __shared__ float s_k[3];
  const int c = blockIdx.y, grp = blockIdx.z;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  if ((int)blockIdx.x >= split) return;
  if (threadIdx.x == 0) {
    double tg, tgx;
    group_totals(a, c, grp, &tg, &tgx);
    const double cnt = (double)(group_row(a, grp + 1) - group_row(a, grp)) * (double)a.HW;
    s_k[0] = (float)(tg / cnt);
    s_k[1] = (float)(tgx / cnt);
    s_k[2] = a.gamma[c] * a.invstd[(size_t)grp * a.C + c];
    if (blockIdx.x == 0 && grp == 0) {      // parameter gradients: sums over every group, in group order
      double sg = tg, sgx = tgx;
      for (int q = 1; q < a.G; ++q) {
        double qg, qgx;
        group_totals(a, c, q, &qg, &qgx);
        sg += qg; sgx += qgx;
      }
      a.dgamma[c] = (float)sgx;
      a.dbeta[c] = (float)sg;
    }
  }
  __syncthreads();
  bwd_apply_slice(a, c, grp, split, blockIdx.x, s_k[0], s_k[1], s_k[2]);
}

// ---- one launch per direction for small activations: grid (1, C, G), the workgroup owns its whole (channel, group).
// `part` then holds one (total, total) pair per (channel, group): [C][G][2].  With more than one group the
// cross-group results (running statistics; parameter gradients) come from a C-thread follow-up launch.
__global__ __launch_bounds__(NT) void bn_fwd_small_kernel(BnArgs a) {
  __shared__ double sh[4];
  __shared__ float s_mean, s_scale, s_shift;
  const int c = blockIdx.y, grp = blockIdx.z;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  double ts = 0.0, tss = 0.0;
  for (int k = 0; k < split; ++k) {          // the same slices, added in the same order, as the two-launch form
    double ks, kss;
    stats_slice(a, c, grp, split, k, sh, &ks, &kss);
    ts += ks; tss += kss;
  }
  if (threadIdx.x == 0) {
    double mean, var;
    group_moments(a, grp, ts, tss, &mean, &var);
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    s_mean = (float)mean;
    s_scale = a.gamma[c] * invstd;
    s_shift = a.beta[c];
    a.mean[(size_t)grp * a.C + c] = (float)mean;
    a.invstd[(size_t)grp * a.C + c] = invstd;
    if (a.G == 1) {
      update_running(a, c, [&](int, double* s0, double* s1) { *s0 = ts; *s1 = tss; });
    } else {
      a.part[((size_t)c * a.G + grp) * 2] = ts;
      a.part[((size_t)c * a.G + grp) * 2 + 1] = tss;
    }
  }
  __syncthreads();
  for (int k = 0; k < split; ++k) apply_slice(a, c, grp, split, k, s_mean, s_scale, s_shift);
}
__global__ __launch_bounds__(64) void bn_running_small_kernel(BnArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.C) return;
  update_running(a, c, [&](int q, double* s0, double* s1) {
    *s0 = a.part[((size_t)c * a.G + q) * 2];
    *s1 = a.part[((size_t)c * a.G + q) * 2 + 1];
  });
}

__global__ __launch_bounds__(NT) void bn_bwd_small_kernel(BnArgs a) {
  __shared__ double sh[4];
  __shared__ float s_k[3];
  const int c = blockIdx.y, grp = blockIdx.z;
  const int split = pick_split(group_row(a, grp + 1) - group_row(a, grp), a.HW);
  double tg = 0.0, tgx = 0.0;
  for (int k = 0; k < split; ++k) {
    double kg, kgx;
    bwd_reduce_slice(a, c, grp, split, k, sh, &kg, &kgx);
    tg += kg; tgx += kgx;
  }
  if (threadIdx.x == 0) {
    const double cnt = (double)(group_row(a, grp + 1) - group_row(a, grp)) * (double)a.HW;
    s_k[0] = (float)(tg / cnt);
    s_k[1] = (float)(tgx / cnt);
    s_k[2] = a.gamma[c] * a.invstd[(size_t)grp * a.C + c];
    if (a.G == 1) {
      a.dgamma[c] = (float)tgx;
      a.dbeta[c] = (float)tg;
    } else {
      a.part[((size_t)c * a.G + grp) * 2] = tg;
      a.part[((size_t)c * a.G + grp) * 2 + 1] = tgx;
    }
  }
  __syncthreads();
  for (int k = 0; k < split; ++k) bwd_apply_slice(a, c, grp, split, k, s_k[0], s_k[1], s_k[2]);
}
__global__ __launch_bounds__(64) void bn_param_grad_small_kernel(BnArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.C) return;
  double sg = 0.0, sgx = 0.0;
  for (int q = 0; q < a.G; ++q) {            // group order, like bn_bwd_apply_kernel
    sg += a.part[((size_t)c * a.G + q) * 2];
    sgx += a.part[((size_t)c * a.G + q) * 2 + 1];
  }
  a.dgamma[c] = (float)sgx;
  a.dbeta[c] = (float)sg;
}

// ---------------------------------------------------------------------------------------------
// ReflectionPad2d(1) (layers.py:118-133, every decoder convolution) and MaxPool2d(3, 2, 1)
// (networks/resnet_encoder.py: the torchvision stem), forward and gather-form backward: no atomics,
// deterministic, one coalesced pass.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect1(int i, int n) {   // index into the un-padded axis for padded index i
  i -= 1;
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// Row-tiled launches: a workgroup of 64 x 4 threads owns 4 rows of one plane (blockIdx.y = plane), the
// 64 lanes of a wave sweep the row, so every access is a coalesced run with no index division.
constexpr int RW = 64, RR = 4;

__global__ __launch_bounds__(RW * RR) void reflect_pad1_fwd_kernel(const float* __restrict__ in,
                                                                   float* __restrict__ out, int H, int W) {
  const int PH = H + 2, PW = W + 2;
  const int py = blockIdx.x * RR + threadIdx.y;
  if (py >= PH) return;
  const size_t pl = blockIdx.y;
  const float* src = in + (pl * H + reflect1(py, H)) * W;
  float* dst = out + (pl * PH + py) * PW;
  for (int px = threadIdx.x; px < PW; px += RW) dst[px] = src[reflect1(px, W)];
}

// grad_in[y][x] = sum of grad_out over every padded position that reads (y, x)
__global__ __launch_bounds__(RW * RR) void reflect_pad1_bwd_kernel(const float* __restrict__ gout,
                                                                   float* __restrict__ gin, int H, int W) {
  const int PH = H + 2, PW = W + 2;
  const int y = blockIdx.x * RR + threadIdx.y;
  if (y >= H) return;
  const size_t pl = blockIdx.y;
  const float* g = gout + pl * PH * PW;
  float* dst = gin + (pl * H + y) * W;
  int ys[3], ny = 0;                       // padded rows that map onto y (wave-uniform)
  ys[ny++] = y + 1;
  if (y == 1) ys[ny++] = 0;
  if (y == H - 2) ys[ny++] = PH - 1;
  for (int x = threadIdx.x; x < W; x += RW) {
    float acc = 0.0f;
    for (int a = 0; a < ny; ++a) {
      const float* row = g + (size_t)ys[a] * PW;
      acc += row[x + 1];
      if (x == 1) acc += row[0];
      if (x == W - 2) acc += row[PW - 1];
    }
    dst[x] = acc;
  }
}

// ---- decoder glue: nearest x2 up-sampling + skip concatenation + ReflectionPad2d(1) as ONE pass -------------
// (reference networks/depth_decoder.py:44-50: `upsample(x)`, `torch.cat([x, skip], 1)`, then the next ConvBlock's
// ReflectionPad2d(1), layers.py:118-133 - three full-tensor round trips in eager mode).  out[n, c, py, px] for the
// padded (H+2)x(W+2) image of the concatenation; H = 2h, W = 2w.
__global__ __launch_bounds__(RW * RR) void upcat_pad_fwd_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                                float* __restrict__ out, int C1, int C2, int h, int w) {
  const int H = 2 * h, W = 2 * w, PH = H + 2, PW = W + 2;
  const int py = blockIdx.x * RR + threadIdx.y;
  if (py >= PH) return;
  const int pl = blockIdx.y, n = pl / (C1 + C2), c = pl - n * (C1 + C2);
  const int yy = reflect1(py, H);
  float* dst = out + ((size_t)pl * PH + py) * PW;
  if (c < C1) {
    const float* src = x + (((size_t)n * C1 + c) * h + (yy >> 1)) * w;
    for (int px = threadIdx.x; px < PW; px += RW) dst[px] = src[reflect1(px, W) >> 1];
  } else {
    const float* src = skip + (((size_t)n * C2 + (c - C1)) * H + yy) * W;
    for (int px = threadIdx.x; px < PW; px += RW) dst[px] = src[reflect1(px, W)];
  }
}

// sum of grad_out over every padded position that reads image position (y, x)  (1, 2, 3, 4, 6 or 9 terms)
__device__ __forceinline__ float pad1_adjoint(const float* __restrict__ g, int y, int x, int H, int W) {
  const int PW = W + 2;
  int rows[3], nr = 0;
  rows[nr++] = y + 1;
  if (y == 1) rows[nr++] = 0;
  if (y == H - 2) rows[nr++] = H + 1;
  float acc = 0.0f;
  for (int a = 0; a < nr; ++a) {
    const float* row = g + (size_t)rows[a] * PW;
    acc += row[x + 1];
    if (x == 1) acc += row[0];
    if (x == W - 2) acc += row[PW - 1];
  }
  return acc;
}

__global__ __launch_bounds__(RW * RR) void upcat_pad_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gx,
                                                                float* __restrict__ gskip, int C1, int C2, int h, int w) {
  const int H = 2 * h, W = 2 * w, PH = H + 2, PW = W + 2;
  const int pl = blockIdx.y, n = pl / (C1 + C2), c = pl - n * (C1 + C2);
  const float* g = gout + (size_t)pl * PH * PW;
  if (c < C1) {                                  // low-resolution rows: each output sums its 2x2 up-sampled block
    const int i = blockIdx.x * RR + threadIdx.y;
    if (i >= h) return;
    float* dst = gx + (((size_t)n * C1 + c) * h + i) * w;
    for (int j = threadIdx.x; j < w; j += RW)
      dst[j] = (pad1_adjoint(g, 2 * i, 2 * j, H, W) + pad1_adjoint(g, 2 * i, 2 * j + 1, H, W)) +
               (pad1_adjoint(g, 2 * i + 1, 2 * j, H, W) + pad1_adjoint(g, 2 * i + 1, 2 * j + 1, H, W));
  } else {
    for (int y = blockIdx.x * RR + threadIdx.y; y < H; y += gridDim.x * RR) {
      float* dst = gskip + (((size_t)n * C2 + (c - C1)) * H + y) * W;
      for (int xx = threadIdx.x; xx < W; xx += RW) dst[xx] = pad1_adjoint(g, y, xx, H, W);
    }
  }
}

// ---- decoder glue: bias + ELU in place on the convolution's output, and its backward with the bias gradient ------
// (layers.ConvBlock = Conv3x3 + ELU, layers.py:103-115; eager PyTorch-ROCm launches conv, a bias add and ELU forward,
// and ELU-backward plus a bias reduction backward).  ELU(alpha = 1): y = v > 0 ? v : expm1(v); dy/dv = y > 0 ? 1 : y + 1.
__global__ __launch_bounds__(NT) void bias_elu_fwd_kernel(float* __restrict__ y, const float* __restrict__ bias, int C, int HW,
                                                          size_t total) {
  for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * NT * 4) {
    const int c = (int)((i / HW) % C);            // HW % 4 == 0: the four elements share a channel
    const float b = bias[c];
    float4 v = *reinterpret_cast<float4*>(y + i);
    v.x += b; v.y += b; v.z += b; v.w += b;
    v.x = v.x > 0.0f ? v.x : expm1f(v.x); v.y = v.y > 0.0f ? v.y : expm1f(v.y);
    v.z = v.z > 0.0f ? v.z : expm1f(v.z); v.w = v.w > 0.0f ? v.w : expm1f(v.w);
    *reinterpret_cast<float4*>(y + i) = v;
  }
}

__global__ __launch_bounds__(NT) void bias_elu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                          float* __restrict__ gx, double* __restrict__ part, int N, int C,
                                                          int HW, int split) {
  __shared__ double sh[4];
  const int c = blockIdx.y;
  int len = (HW + split - 1) / split;
  len = (len + 3) & ~3;
  const int lo = blockIdx.x * len, hi = min(HW, lo + len);
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * HW;
    float s = 0.0f;
    for (int i = lo + 4 * (int)threadIdx.x; i < hi; i += 4 * NT) {
      const float4 o = *reinterpret_cast<const float4*>(y + base + i);
      float4 g = *reinterpret_cast<const float4*>(gy + base + i);
      g.x = o.x > 0.0f ? g.x : g.x * (o.x + 1.0f); g.y = o.y > 0.0f ? g.y : g.y * (o.y + 1.0f);
      g.z = o.z > 0.0f ? g.z : g.z * (o.z + 1.0f); g.w = o.w > 0.0f ? g.w : g.w * (o.w + 1.0f);
      *reinterpret_cast<float4*>(gx + base + i) = g;
      s += (g.x + g.y) + (g.z + g.w);
    }
    acc += (double)s;
  }
  const double t = block_sum(acc, sh);
  if (threadIdx.x == 0) part[(size_t)c * split + blockIdx.x] = t;
}

__global__ __launch_bounds__(64) void bias_grad_final_kernel(const double* __restrict__ part, float* __restrict__ gbias, int C,
                                                             int split) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int k = 0; k < split; ++k) s += part[(size_t)c * split + k];
  gbias[c] = (float)s;
}

// MaxPool2d(kernel 3, stride 2, padding 1): ATen's scan order and tie / NaN rule (first maximum in
// row-major window order; a NaN replaces the running maximum).  `code` = window position 0..8.
__global__ __launch_bounds__(RW * RR) void maxpool3s2_fwd_kernel(const float* __restrict__ in,
                                                                 float* __restrict__ out, uint8_t* __restrict__ code,
                                                                 int H, int W, int OH, int OW) {
  const int oy = blockIdx.x * RR + threadIdx.y;
  if (oy >= OH) return;
  const size_t pl = blockIdx.y;
  const float* p = in + pl * H * W;
  for (int ox = threadIdx.x; ox < OW; ox += RW) {
    float best = -INFINITY;
    int bc = -1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * oy - 1 + dy;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * ox - 1 + dx;
        if (x < 0 || x >= W) continue;
        const float v = p[(size_t)y * W + x];
        if (bc < 0 || v > best || v != v) { best = v; bc = dy * 3 + dx; }
      }
    }
    out[(pl * OH + oy) * OW + ox] = best;
    code[(pl * OH + oy) * OW + ox] = (uint8_t)bc;
  }
}

// Backward, one thread per output window (oy, ox): it owns the 2x2 input block whose top-left pixel is
// the window centre (2oy, 2ox) and gathers from the <= 4 windows that overlap the block, so every code /
// gradient value is read once per neighbour and the stores are two coalesced float2 rows.
__global__ __launch_bounds__(RW * RR) void maxpool3s2_bwd_kernel(const float* __restrict__ gout,
                                                                 const uint8_t* __restrict__ code,
                                                                 float* __restrict__ gin, int H, int W, int OH,
                                                                 int OW) {
  const int oy = blockIdx.x * RR + threadIdx.y;
  if (oy >= OH) return;
  const size_t pl = blockIdx.y;
  const float* g = gout + pl * OH * OW;
  const uint8_t* cd = code + pl * OH * OW;
  float* dst = gin + pl * H * W;
  const bool down = oy + 1 < OH;
  const int y0 = 2 * oy, y1 = 2 * oy + 1;
  for (int ox = threadIdx.x; ox < OW; ox += RW) {
    const bool right = ox + 1 < OW;
    const size_t i00 = (size_t)oy * OW + ox;
    const int c00 = cd[i00];
    const float g00 = g[i00];
    const int c01 = right ? cd[i00 + 1] : -1;
    const float g01 = right ? g[i00 + 1] : 0.0f;
    const int c10 = down ? cd[i00 + OW] : -1;
    const float g10 = down ? g[i00 + OW] : 0.0f;
    const int c11 = (down && right) ? cd[i00 + OW + 1] : -1;
    const float g11 = (down && right) ? g[i00 + OW + 1] : 0.0f;
    // window (oy,ox) covers rows 2oy-1..2oy+1: the block's pixels sit at window positions 4,5 / 7,8
    const float a = c00 == 4 ? g00 : 0.0f;                                            // (y0, x0)
    const float b = (c00 == 5 ? g00 : 0.0f) + (c01 == 3 ? g01 : 0.0f);                // (y0, x0+1)
    const float c = (c00 == 7 ? g00 : 0.0f) + (c10 == 1 ? g10 : 0.0f);                // (y0+1, x0)
    const float d = ((c00 == 8 ? g00 : 0.0f) + (c01 == 6 ? g01 : 0.0f)) +
                    ((c10 == 2 ? g10 : 0.0f) + (c11 == 0 ? g11 : 0.0f));              // (y0+1, x0+1)
    const int x0 = 2 * ox;
    dst[(size_t)y0 * W + x0] = a;
    if (x0 + 1 < W) dst[(size_t)y0 * W + x0 + 1] = b;
    if (y1 < H) {
      dst[(size_t)y1 * W + x0] = c;
      if (x0 + 1 < W) dst[(size_t)y1 * W + x0 + 1] = d;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Disparity head of the depth decoder: Conv3x3(C -> 1) = ReflectionPad2d(1) + Conv2d(C, 1, 3) + bias
// (layers.py:118-133, networks/depth_decoder.py:38-39, C = 16/32/64/128 at 192x640 .. 24x80).  With one
// output channel the layer is a pure streaming reduction over C input planes; MIOpen runs it at
// ~2.4 TFLOP/s (0.54 ms for C=16 at full resolution, forward+backward).  Here: one pass over x each way,
// reflection handled by index (no padded copy), weights in LDS, deterministic weight-gradient partials.
// ---------------------------------------------------------------------------------------------
constexpr int DC_MAXC = 256;
constexpr int DC_CHUNKS = 64;

// 3 rows x 6 columns (px0-1 .. px0+4, reflected at the image border) of one plane around a 4-pixel strip
__device__ __forceinline__ void load_strip_window(const float* __restrict__ plane, int r0, int r1, int r2, int px0,
                                                  int cl, int cr, float win[3][6]) {
  const int rows[3] = {r0, r1, r2};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float4 m = *reinterpret_cast<const float4*>(plane + rows[r] + px0);
    win[r][0] = plane[rows[r] + cl];
    win[r][1] = m.x; win[r][2] = m.y; win[r][3] = m.z; win[r][4] = m.w;
    win[r][5] = plane[rows[r] + cr];
  }
}

// One thread = 4 adjacent output pixels (W % 4 == 0: the strips and their 16-byte loads stay aligned);
// per channel 3 x (float4 + 2 scalars) loads feed 36 multiply-adds.  Other widths: one pixel per thread.
__global__ __launch_bounds__(NT) void dispconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int C, int H, int W) {
  __shared__ float sw[DC_MAXC * 9];
  for (int i = threadIdx.x; i < C * 9; i += NT) sw[i] = w[i];
  __syncthreads();
  const int hw = H * W;
  const size_t n = blockIdx.y;
  const float b = bias ? bias[0] : 0.0f;
  if ((W & 3) == 0) {
    const int strips = hw >> 2;
    for (int t = blockIdx.x * NT + threadIdx.x; t < strips; t += gridDim.x * NT) {
      const int p = t << 2;
      const int py = p / W, px0 = p - py * W;
      const int r0 = reflect1(py, H) * W, r1 = py * W, r2 = reflect1(py + 2, H) * W;
      const int cl = reflect1(px0, W), cr = reflect1(px0 + 5, W);
      float acc[4] = {b, b, b, b};
      const float* xc = x + n * C * hw;
      for (int c = 0; c < C; ++c, xc += hw) {
        const float* k = sw + c * 9;
        float win[3][6];
        load_strip_window(xc, r0, r1, r2, px0, cl, cr, win);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] += (k[0] * win[0][j] + k[1] * win[0][j + 1] + k[2] * win[0][j + 2]) +
                    (k[3] * win[1][j] + k[4] * win[1][j + 1] + k[5] * win[1][j + 2]) +
                    (k[6] * win[2][j] + k[7] * win[2][j + 1] + k[8] * win[2][j + 2]);
      }
      *reinterpret_cast<float4*>(y + n * hw + p) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    return;
  }
  for (int p = blockIdx.x * NT + threadIdx.x; p < hw; p += gridDim.x * NT) {
    const int py = p / W, px = p - py * W;
    const int r0 = reflect1(py, H) * W, r1 = py * W, r2 = reflect1(py + 2, H) * W;
    const int c0 = reflect1(px, W), c2 = reflect1(px + 2, W);
    float acc = b;
    const float* xc = x + n * C * hw;
    for (int c = 0; c < C; ++c, xc += hw) {
      const float* k = sw + c * 9;
      acc += (k[0] * xc[r0 + c0] + k[1] * xc[r0 + px] + k[2] * xc[r0 + c2]) +
             (k[3] * xc[r1 + c0] + k[4] * xc[r1 + px] + k[5] * xc[r1 + c2]) +
             (k[6] * xc[r2 + c0] + k[7] * xc[r2 + px] + k[8] * xc[r2 + c2]);
    }
    y[n * hw + p] = acc;
  }
}

// grad wrt x:  gx[n,c,q] = sum_d w[c][d] * G_d(q),  G_d(q) = sum over the padded positions (u,v) that
// read q of g[n, u - dy, v - dx] (zero outside the image) - G_d is shared by all channels.
__global__ __launch_bounds__(NT) void dispconv_bwd_data_kernel(const float* __restrict__ gy,
                                                               const float* __restrict__ w, float* __restrict__ gx,
                                                               int C, int H, int W) {
  __shared__ float sw[DC_MAXC * 9];
  for (int i = threadIdx.x; i < C * 9; i += NT) sw[i] = w[i];
  __syncthreads();
  const int hw = H * W;
  const size_t n = blockIdx.y;
  const float* g = gy + n * hw;
  for (int q = blockIdx.x * NT + threadIdx.x; q < hw; q += gridDim.x * NT) {
    const int qy = q / W, qx = q - qy * W;
    int us[3], vs[3], nu = 0, nv = 0;          // padded coordinates (0..H+1, 0..W+1) mapping onto (qy, qx)
    us[nu++] = qy + 1;
    if (qy == 1) us[nu++] = 0;
    if (qy == H - 2) us[nu++] = H + 1;
    vs[nv++] = qx + 1;
    if (qx == 1) vs[nv++] = 0;
    if (qx == W - 2) vs[nv++] = W + 1;
    float G[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) G[d] = 0.0f;
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nv; ++b) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int py = us[a] - dy;           // output row whose window row dy sits on padded row us[a]
          if (py < 0 || py >= H) continue;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int px = vs[b] - dx;
            if (px < 0 || px >= W) continue;
            G[dy * 3 + dx] += g[py * W + px];
          }
        }
      }
    float* o = gx + n * C * hw + q;
    for (int c = 0; c < C; ++c, o += hw) {
      const float* k = sw + c * 9;
      *o = ((k[0] * G[0] + k[1] * G[1] + k[2] * G[2]) + (k[3] * G[3] + k[4] * G[4] + k[5] * G[5])) +
           (k[6] * G[6] + k[7] * G[7] + k[8] * G[8]);
    }
  }
}

// grad wrt w (and bias): block (chunk, c) accumulates its slice of the N*H*W pixels; fp64 partials.
__global__ __launch_bounds__(NT) void dispconv_bwd_weight_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ gy,
                                                                 double* __restrict__ part, int N, int C, int H,
                                                                 int W) {
  __shared__ double sh[4];
  const int c = blockIdx.y, hw = H * W;
  float acc[10];
#pragma unroll
  for (int d = 0; d < 10; ++d) acc[d] = 0.0f;
  if ((W & 3) == 0) {
    const long long strips = ((long long)N * hw) >> 2;
    for (long long t = (long long)blockIdx.x * NT + threadIdx.x; t < strips; t += (long long)gridDim.x * NT) {
      const long long i = t << 2;
      const int n = (int)(i / hw), p = (int)(i - (long long)n * hw);
      const int py = p / W, px0 = p - py * W;
      const float4 g4 = *reinterpret_cast<const float4*>(gy + i);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
      const float* xc = x + ((size_t)n * C + c) * hw;
      float win[3][6];
      load_strip_window(xc, reflect1(py, H) * W, py * W, reflect1(py + 2, H) * W, px0, reflect1(px0, W),
                        reflect1(px0 + 5, W), win);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int d = 0; d < 3; ++d) acc[r * 3 + d] += g[j] * win[r][j + d];
        acc[9] += g[j];
      }
    }
  } else {
    const long long total = (long long)N * hw;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
      const int n = (int)(i / hw), p = (int)(i - (long long)n * hw);
      const int py = p / W, px = p - py * W;
      const float g = gy[i];
      const float* xc = x + ((size_t)n * C + c) * hw;
      const int r0 = reflect1(py, H) * W, r1 = py * W, r2 = reflect1(py + 2, H) * W;
      const int c0 = reflect1(px, W), c2 = reflect1(px + 2, W);
      acc[0] += g * xc[r0 + c0]; acc[1] += g * xc[r0 + px]; acc[2] += g * xc[r0 + c2];
      acc[3] += g * xc[r1 + c0]; acc[4] += g * xc[r1 + px]; acc[5] += g * xc[r1 + c2];
      acc[6] += g * xc[r2 + c0]; acc[7] += g * xc[r2 + px]; acc[8] += g * xc[r2 + c2];
      acc[9] += g;
    }
  }
#pragma unroll
  for (int d = 0; d < 10; ++d) {
    const double t = block_sum((double)acc[d], sh);
    if (threadIdx.x == 0) part[((size_t)blockIdx.x * C + c) * 10 + d] = t;
  }
}

// one wave per output: lanes = chunks (DC_CHUNKS == 64), fixed-order butterfly
__global__ __launch_bounds__(64) void dispconv_bwd_weight_final_kernel(const double* __restrict__ part,
                                                                      float* __restrict__ gw, float* __restrict__ gb,
                                                                      int C) {
  const int i = blockIdx.x;                    // 0 .. C*9 (the last one is the bias)
  const int c = i < C * 9 ? i / 9 : 0, d = i < C * 9 ? i - c * 9 : 9;
  double v = part[((size_t)threadIdx.x * C + c) * 10 + d];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if (threadIdx.x == 0) {
    if (i < C * 9) gw[i] = (float)v;
    else if (gb) gb[0] = (float)v;
  }
}

int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int bbd_bn_scratch_doubles(int N, int C, int HW) { return C * pick_split(N, HW) * 2; }
int bbd_bn_grouped_scratch_doubles(int max_group_rows, int G, int C, int HW) {
  return G * C * pick_split(max_group_rows, HW) * 2;
}

namespace {
// copies the group table into the launch arguments; returns the largest group (0 = invalid table)
int fill_groups(const int32_t* group_rows, int G, int N, BnArgs* a) {
  if (!group_rows || G < 1 || G > BBD_BN_MAX_GROUPS || group_rows[0] != 0 || group_rows[G] != N) return 0;
  int biggest = 0;
  for (int g = 0; g < G; ++g) {
    const int n = group_rows[g + 1] - group_rows[g];
    if (n <= 0) return 0;
    biggest = n > biggest ? n : biggest;
  }
  a->G = G;
  for (int g = 0; g <= G; ++g) a->rows[g] = group_rows[g];
  return biggest;
}
}  // namespace

namespace {
int launch_bn_fwd(BnArgs& a, int biggest, bool running, hipStream_t st) {
  a.split = pick_split(biggest, a.HW);
  if ((long long)biggest * a.HW <= BN_SMALL_ELEMS) {       // one launch (+ a C-thread one for the cross-group results)
    hipLaunchKernelGGL(bn_fwd_small_kernel, dim3(1, (unsigned)a.C, (unsigned)a.G), dim3(NT), 0, st, a);
    if (a.G > 1 && running)
      hipLaunchKernelGGL(bn_running_small_kernel, dim3((unsigned)((a.C + 63) / 64)), dim3(64), 0, st, a);
    return status();
  }
  const dim3 grid((unsigned)a.split, (unsigned)a.C, (unsigned)a.G);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(NT), 0, st, a);
  hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(NT), 0, st, a);
  return status();
}
int launch_bn_bwd(BnArgs& a, int biggest, hipStream_t st) {
  a.split = pick_split(biggest, a.HW);
  if ((long long)biggest * a.HW <= BN_SMALL_ELEMS) {
    hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(1, (unsigned)a.C, (unsigned)a.G), dim3(NT), 0, st, a);
    if (a.G > 1) hipLaunchKernelGGL(bn_param_grad_small_kernel, dim3((unsigned)((a.C + 63) / 64)), dim3(64), 0, st, a);
    return status();
  }
  const dim3 grid((unsigned)a.split, (unsigned)a.C, (unsigned)a.G);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(NT), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(NT), 0, st, a);
  return status();
}
void fill_fwd(BnArgs& a, const float* x, const float* residual, const float* gamma, const float* beta, float* y,
              float* save_mean, float* save_invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
              double* scratch, int N, int C, int HW, double eps, double momentum, int relu) {
  a.x = x; a.res = residual; a.gamma = gamma; a.beta = beta; a.y = y; a.part = scratch; a.mean = save_mean;
  a.invstd = save_invstd; a.run_mean = running_mean; a.run_var = running_var; a.batches = num_batches_tracked; a.N = N; a.C = C; a.HW = HW;
  a.relu = relu; a.eps = (float)eps; a.momentum = (float)momentum;
}
void fill_bwd(BnArgs& a, const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
              const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual, float* grad_gamma,
              float* grad_beta, double* scratch, int N, int C, int HW, int relu) {
  a.x = x; a.y = const_cast<float*>(y); a.dy = grad_y; a.gamma = gamma; a.beta = beta; a.mean = const_cast<float*>(save_mean);
  a.invstd = const_cast<float*>(save_invstd); a.dx = grad_x; a.dres = grad_residual; a.dgamma = grad_gamma;
  a.dbeta = grad_beta; a.part = scratch; a.N = N; a.C = C; a.HW = HW; a.relu = relu;
}
}  // namespace

int bbd_bn_act_grouped_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                           float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                           long long* num_batches_tracked, double* scratch, const int32_t* group_rows, int G,
                           int untracked_groups, int N, int C, int HW, double eps, double momentum, int relu, void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !scratch || N <= 0 || C <= 0 || HW <= 0)
    return BBD_E_BADARG;
  if (untracked_groups < 0 || untracked_groups >= G) return BBD_E_BADARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return BBD_E_BADARG;
  BnArgs a = {};
  const int biggest = fill_groups(group_rows, G, N, &a);
  if (!biggest) return BBD_E_BADARG;
  fill_fwd(a, x, residual, gamma, beta, y, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, scratch,
           N, C, HW, eps, momentum, relu);
  a.untracked = untracked_groups;
  return launch_bn_fwd(a, biggest, running_mean || num_batches_tracked, static_cast<hipStream_t>(stream));
}

int bbd_bn_act_grouped_bwd(const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual,
                           float* grad_gamma, float* grad_beta, double* scratch, const int32_t* group_rows, int G, int N,
                           int C, int HW, int relu, void* stream) {
  if (!x || !grad_y || !gamma || !save_mean || !save_invstd || !grad_x || !grad_gamma || !grad_beta || !scratch ||
      N <= 0 || C <= 0 || HW <= 0 || (relu && !y && !beta))
    return BBD_E_BADARG;
  BnArgs a = {};
  const int biggest = fill_groups(group_rows, G, N, &a);
  if (!biggest) return BBD_E_BADARG;
  fill_bwd(a, x, y, grad_y, gamma, beta, save_mean, save_invstd, grad_x, grad_residual, grad_gamma, grad_beta, scratch, N, C,
           HW, relu);
  return launch_bn_bwd(a, biggest, static_cast<hipStream_t>(stream));
}

// Device-resident group table (ABI 7): the same launches with NOTHING of the batch signature in their arguments.
int bbd_bn_act_grouped_dev_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                               float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                               long long* num_batches_tracked, double* scratch, const int32_t* group_table, int G,
                               int max_group_rows, int N, int C, int HW, double eps, double momentum, int relu,
                               void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !scratch || !group_table || N <= 0 || C <= 0 || HW <= 0)
    return BBD_E_BADARG;
  if (G < 1 || G > BBD_BN_MAX_GROUPS || max_group_rows < 1 || max_group_rows > N) return BBD_E_BADARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return BBD_E_BADARG;
  BnArgs a = {};
  fill_fwd(a, x, residual, gamma, beta, y, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, scratch,
           N, C, HW, eps, momentum, relu);
  a.G = G; a.rows_dev = group_table;
  return launch_bn_fwd(a, max_group_rows, running_mean || num_batches_tracked, static_cast<hipStream_t>(stream));
}

int bbd_bn_act_grouped_dev_bwd(const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual,
                               float* grad_gamma, float* grad_beta, double* scratch, const int32_t* group_table, int G,
                               int max_group_rows, int N, int C, int HW, int relu, void* stream) {
  if (!x || !grad_y || !gamma || !save_mean || !save_invstd || !grad_x || !grad_gamma || !grad_beta || !scratch ||
      !group_table || N <= 0 || C <= 0 || HW <= 0 || (relu && !y && !beta))
    return BBD_E_BADARG;
  if (G < 1 || G > BBD_BN_MAX_GROUPS || max_group_rows < 1 || max_group_rows > N) return BBD_E_BADARG;
  BnArgs a = {};
  fill_bwd(a, x, y, grad_y, gamma, beta, save_mean, save_invstd, grad_x, grad_residual, grad_gamma, grad_beta, scratch, N, C,
           HW, relu);
  a.G = G; a.rows_dev = group_table;
  return launch_bn_bwd(a, max_group_rows, static_cast<hipStream_t>(stream));
}

int bbd_bn_act_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                   float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                   long long* num_batches_tracked, double* scratch, int N, int C, int HW, double eps, double momentum,
                   int relu, void* stream) {
  const int32_t rows[2] = {0, N};
  return bbd_bn_act_grouped_fwd(x, residual, gamma, beta, y, save_mean, save_invstd, running_mean, running_var,
                                num_batches_tracked, scratch, rows, 1, 0, N, C, HW, eps, momentum, relu, stream);
}

int bbd_bn_act_bwd(const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                   double* scratch, int N, int C, int HW, int relu, void* stream) {
  const int32_t rows[2] = {0, N};
  return bbd_bn_act_grouped_bwd(x, y, grad_y, gamma, beta, save_mean, save_invstd, grad_x, grad_residual, grad_gamma,
                                grad_beta, scratch, rows, 1, N, C, HW, relu, stream);
}

int bbd_reflect_pad1_fwd(const float* in, float* out, int planes, int H, int W, void* stream) {
  if (!in || !out || planes <= 0 || planes > 65535 || H < 2 || W < 2) return BBD_E_BADARG;
  hipLaunchKernelGGL(reflect_pad1_fwd_kernel, dim3((unsigned)((H + 2 + RR - 1) / RR), (unsigned)planes), dim3(RW, RR), 0,
                     static_cast<hipStream_t>(stream), in, out, H, W);
  return status();
}

int bbd_reflect_pad1_bwd(const float* grad_out, float* grad_in, int planes, int H, int W, void* stream) {
  if (!grad_out || !grad_in || planes <= 0 || planes > 65535 || H < 2 || W < 2) return BBD_E_BADARG;
  hipLaunchKernelGGL(reflect_pad1_bwd_kernel, dim3((unsigned)((H + RR - 1) / RR), (unsigned)planes), dim3(RW, RR), 0,
                     static_cast<hipStream_t>(stream), grad_out, grad_in, H, W);
  return status();
}

int bbd_upcat_pad1_fwd(const float* x, const float* skip, float* out, int N, int C1, int C2, int h, int w, void* stream) {
  if (!x || !out || N <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !skip) || h < 1 || w < 1) return BBD_E_BADARG;
  if ((long)N * (C1 + C2) > 65535) return BBD_E_BADARG;
  hipLaunchKernelGGL(upcat_pad_fwd_kernel, dim3((unsigned)((2 * h + 2 + RR - 1) / RR), (unsigned)(N * (C1 + C2))), dim3(RW, RR),
                     0, static_cast<hipStream_t>(stream), x, skip, out, C1, C2, h, w);
  return status();
}

int bbd_upcat_pad1_bwd(const float* grad_out, float* grad_x, float* grad_skip, int N, int C1, int C2, int h, int w,
                       void* stream) {
  if (!grad_out || !grad_x || N <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !grad_skip) || h < 1 || w < 1) return BBD_E_BADARG;
  if ((long)N * (C1 + C2) > 65535) return BBD_E_BADARG;
  hipLaunchKernelGGL(upcat_pad_bwd_kernel, dim3((unsigned)((2 * h + RR - 1) / RR), (unsigned)(N * (C1 + C2))), dim3(RW, RR), 0,
                     static_cast<hipStream_t>(stream), grad_out, grad_x, grad_skip, C1, C2, h, w);
  return status();
}

int bbd_bias_elu_scratch_doubles(int N, int C, int HW) { return C * pick_split(N, HW); }

int bbd_bias_elu_fwd(float* y, const float* bias, int N, int C, int HW, void* stream) {
  if (!y || !bias || N <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return BBD_E_BADARG;
  const size_t total = (size_t)N * C * HW;
  const size_t blocks = (total / 4 + NT - 1) / NT;
  hipLaunchKernelGGL(bias_elu_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), y, bias, C, HW, total);
  return status();
}

int bbd_bias_elu_bwd(const float* y, const float* grad_y, float* grad_x, float* grad_bias, double* scratch, int N, int C,
                     int HW, void* stream) {
  if (!y || !grad_y || !grad_x || !grad_bias || !scratch || N <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return BBD_E_BADARG;
  const int split = pick_split(N, HW);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bias_elu_bwd_kernel, dim3((unsigned)split, (unsigned)C), dim3(NT), 0, st, y, grad_y, grad_x, scratch, N, C,
                     HW, split);
  hipLaunchKernelGGL(bias_grad_final_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, st, scratch, grad_bias, C, split);
  return status();
}

int bbd_maxpool3s2_fwd(const float* in, float* out, uint8_t* code, int planes, int H, int W, void* stream) {
  if (!in || !out || !code || planes <= 0 || planes > 65535 || H < 1 || W < 1) return BBD_E_BADARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;          // floor((H + 2 - 3) / 2) + 1
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((unsigned)((OH + RR - 1) / RR), (unsigned)planes), dim3(RW, RR), 0,
                     static_cast<hipStream_t>(stream), in, out, code, H, W, OH, OW);
  return status();
}

int bbd_maxpool3s2_bwd(const float* grad_out, const uint8_t* code, float* grad_in, int planes, int H, int W,
                       void* stream) {
  if (!grad_out || !code || !grad_in || planes <= 0 || planes > 65535 || H < 1 || W < 1) return BBD_E_BADARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)((OH + RR - 1) / RR), (unsigned)planes), dim3(RW, RR), 0,
                     static_cast<hipStream_t>(stream), grad_out, code, grad_in, H, W, OH, OW);
  return status();
}

int bbd_dispconv_scratch_doubles(int C) { return DC_CHUNKS * C * 10; }

int bbd_dispconv_fwd(const float* x, const float* weight, const float* bias, float* y, int N, int C, int H, int W,
                     void* stream) {
  if (!x || !weight || !y || N <= 0 || C <= 0 || C > DC_MAXC || H < 2 || W < 2) return BBD_E_BADARG;
  const int hw = H * W;
  const int work = (W & 3) == 0 ? hw / 4 : hw;
  const unsigned gx = (unsigned)((work + NT - 1) / NT);
  hipLaunchKernelGGL(dispconv_fwd_kernel, dim3(gx, (unsigned)N), dim3(NT), 0, static_cast<hipStream_t>(stream), x,
                     weight, bias, y, C, H, W);
  return status();
}

int bbd_dispconv_bwd(const float* x, const float* weight, const float* grad_y, float* grad_x, float* grad_weight,
                     float* grad_bias, double* scratch, int N, int C, int H, int W, void* stream) {
  if (!x || !weight || !grad_y || !scratch || N <= 0 || C <= 0 || C > DC_MAXC || H < 2 || W < 2) return BBD_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int hw = H * W;
  if (grad_x)
    hipLaunchKernelGGL(dispconv_bwd_data_kernel, dim3((unsigned)((hw + NT - 1) / NT), (unsigned)N), dim3(NT), 0, st,
                       grad_y, weight, grad_x, C, H, W);
  if (grad_weight) {
    hipLaunchKernelGGL(dispconv_bwd_weight_kernel, dim3((unsigned)DC_CHUNKS, (unsigned)C), dim3(NT), 0, st, x, grad_y,
                       scratch, N, C, H, W);
    static_assert(DC_CHUNKS == 64, "the final reduction maps chunks onto the 64 lanes of a wave");
    hipLaunchKernelGGL(dispconv_bwd_weight_final_kernel, dim3((unsigned)(C * 9 + 1)), dim3(64), 0, st, scratch,
                       grad_weight, grad_bias, C);
  }
  return status();
}

}  // extern "C"
