// bbd_image.hip - the loader's image pipeline on the device.
//
// The reference resizes / jitters / converts every frame of every sample on CPU workers through
// torchvision + Pillow (datasets/mono_dataset.py:186-205) and then stacks the per-item tensors in
// Trainer.custom_collate (trainer.py:867-886).  Here the host only decodes JPEGs; the decoded uint8
// HWC frames are uploaded once and everything else is byte/integer kernels driven by job tables
// ("index-table form": each job names its source frame and the row of the collated batch tensor it
// writes, so there is no per-item tensor and no stacking copy):
//
//   bbd_resample_h_u8 / bbd_resample_v_u8   one pass of Pillow's 8-bit ImagingResample (LANCZOS here;
//                                           any separable filter - the host supplies the fixed-point
//                                           coefficient table), optional left-right flip folded into
//                                           the horizontal read
//   bbd_color_jitter_u8                     torchvision ColorJitter (brightness / contrast / saturation
//                                           / hue in a per-image random order) + ToTensor, two kernels
//                                           (the contrast op needs the image-wide mean of the image as
//                                           it is at that point of the sequence)
//   bbd_u8_to_float_chw                     ToTensor: uint8 HWC -> fp32 CHW / 255
//
// All arithmetic is in bbd_image_math.h and is bit-exact against Pillow.  These kernels are byte
// movers: per 1242x375 -> 640x192 frame the two passes read 1.40 + 0.72 MB and write 0.72 + 0.37 MB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"
#include "bbd_image_math.h"

namespace {

constexpr int NT = 256;
constexpr int RH = 8;    // input rows per workgroup, horizontal pass
constexpr int RV = 8;    // output rows per workgroup, vertical pass

struct ResampleJob {      // BBD_RESAMPLE_JOB int32 fields
  int32_t src_lo, src_hi, dst_lo, dst_hi;   // byte offsets into src / dst
  int32_t in_h, in_w, out_size, ksize;
  int32_t coef_off, bounds_off, flags, pad;
};
static_assert(sizeof(ResampleJob) == BBD_RESAMPLE_JOB * 4, "job layout");

__device__ __forceinline__ size_t off64(int32_t lo, int32_t hi) {
  return (size_t)(uint32_t)lo | ((size_t)(uint32_t)hi << 32);
}

// out[row][x][c] = clip8(2^21 + sum_j k[x][j] * in[row][xmin + j][c]); flip mirrors the source columns.
// A workgroup stages RH source rows in LDS with aligned dword loads (rows start at arbitrary byte
// addresses: 1242*3 is not a multiple of 4), then every thread produces its output columns for all RH
// rows, reading each filter tap once.  Rows wider than the LDS budget take the direct-from-L2 path.
constexpr int H_LDS_ROW = 6160;   // bytes per staged row: up to 2048 px * 3 + alignment slack
template <int C>
__global__ __launch_bounds__(NT) void resample_h_kernel(const uint8_t* src, uint8_t* dst, const ResampleJob* jobs,
                                                        const int32_t* coef, const int32_t* bounds) {
  __shared__ __attribute__((aligned(16))) uint8_t s_rows[RH][H_LDS_ROW];
  const ResampleJob jb = jobs[blockIdx.y];
  const int row0 = blockIdx.x * RH;
  if (row0 >= jb.in_h) return;
  const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
  uint8_t* out = dst + off64(jb.dst_lo, jb.dst_hi);
  const int out_w = jb.out_size, in_w = jb.in_w;
  const int flip = jb.flags & BBD_RESAMPLE_FLIP;
  const int rows = jb.in_h - row0 < RH ? jb.in_h - row0 : RH;
  const int rowbytes = in_w * C;
  const bool staged = rowbytes + 8 <= H_LDS_ROW;
  int mis[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) mis[r] = 0;
  if (staged) {
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      if (r >= rows) break;
      const uint8_t* irow = in + (size_t)(row0 + r) * rowbytes;
      mis[r] = (int)((uintptr_t)irow & 3u);
      const uint32_t* w = reinterpret_cast<const uint32_t*>(irow - mis[r]);
      const int nw = (rowbytes + mis[r]) >> 2;            // whole dwords inside the row; tail by bytes
      uint32_t* d = reinterpret_cast<uint32_t*>(s_rows[r]);
      for (int i = threadIdx.x; i < nw; i += NT) d[i] = w[i];
      const int tail = (rowbytes + mis[r]) & 3;
      if ((int)threadIdx.x < tail) s_rows[r][4 * nw + threadIdx.x] = irow[4 * nw - mis[r] + threadIdx.x];
    }
    __syncthreads();
  }
  for (int x = threadIdx.x; x < out_w; x += NT) {
    const int xmin = bounds[jb.bounds_off + 2 * x], xmax = bounds[jb.bounds_off + 2 * x + 1];
    const int32_t* k = coef + jb.coef_off + (size_t)x * jb.ksize;
    int32_t acc[RH][C];
#pragma unroll
    for (int r = 0; r < RH; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[r][c] = 1 << (BBD_RESAMPLE_PRECISION - 1);
    for (int j = 0; j < xmax; ++j) {
      const int kj = k[j];
      const int sx = (flip ? in_w - 1 - (xmin + j) : xmin + j) * C;
      if (staged) {
#pragma unroll
        for (int r = 0; r < RH; ++r)
#pragma unroll
          for (int c = 0; c < C; ++c) acc[r][c] += (int32_t)s_rows[r][mis[r] + sx + c] * kj;
      } else {
#pragma unroll
        for (int r = 0; r < RH; ++r)
          if (r < rows)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[r][c] += (int32_t)in[(size_t)(row0 + r) * rowbytes + sx + c] * kj;
      }
    }
#pragma unroll
    for (int r = 0; r < RH; ++r)
      if (r < rows)
#pragma unroll
        for (int c = 0; c < C; ++c) out[((size_t)(row0 + r) * out_w + x) * C + c] = bbd_img_clip8(acc[r][c]);
  }
}

// out[y][b] = clip8(2^21 + sum_j k[y][j] * in[ymin + j][b]) over byte columns b of a row (W*C bytes).
// Each thread owns 4 adjacent byte columns: one dword load per tap when the rows are dword-aligned
// (640*3 bytes per row and dword-aligned image offsets - the loader's case), byte loads otherwise.
__global__ __launch_bounds__(NT) void resample_v_kernel(const uint8_t* src, uint8_t* dst, const ResampleJob* jobs,
                                                        const int32_t* coef, const int32_t* bounds, int C) {
  const ResampleJob jb = jobs[blockIdx.z];
  const int y0 = blockIdx.y * RV;
  const int rowbytes = jb.in_w * C;
  const int b = (blockIdx.x * NT + threadIdx.x) * 4;
  if (y0 >= jb.out_size || b >= rowbytes) return;
  const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
  uint8_t* out = dst + off64(jb.dst_lo, jb.dst_hi);
  const int ys = jb.out_size - y0 < RV ? jb.out_size - y0 : RV;
  const bool words = ((rowbytes & 3) == 0) && (((uintptr_t)in & 3u) == 0) && (((uintptr_t)out & 3u) == 0);
  const int nb = rowbytes - b < 4 ? rowbytes - b : 4;
  for (int r = 0; r < ys; ++r) {
    const int y = y0 + r;
    const int ymin = bounds[jb.bounds_off + 2 * y], ymax = bounds[jb.bounds_off + 2 * y + 1];
    const int32_t* k = coef + jb.coef_off + (size_t)y * jb.ksize;
    int32_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = 1 << (BBD_RESAMPLE_PRECISION - 1);
    if (words) {
      for (int j = 0; j < ymax; ++j) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(in + (size_t)(ymin + j) * rowbytes + b);
        const int kj = k[j];
        acc[0] += (int32_t)(v & 255u) * kj;
        acc[1] += (int32_t)((v >> 8) & 255u) * kj;
        acc[2] += (int32_t)((v >> 16) & 255u) * kj;
        acc[3] += (int32_t)(v >> 24) * kj;
      }
      const uint32_t o = (uint32_t)bbd_img_clip8(acc[0]) | ((uint32_t)bbd_img_clip8(acc[1]) << 8) |
                         ((uint32_t)bbd_img_clip8(acc[2]) << 16) | ((uint32_t)bbd_img_clip8(acc[3]) << 24);
      *reinterpret_cast<uint32_t*>(out + (size_t)y * rowbytes + b) = o;
    } else {
      for (int j = 0; j < ymax; ++j) {
        const int kj = k[j];
        for (int i = 0; i < nb; ++i) acc[i] += (int32_t)in[(size_t)(ymin + j) * rowbytes + b + i] * kj;
      }
      for (int i = 0; i < nb; ++i) out[(size_t)y * rowbytes + b + i] = bbd_img_clip8(acc[i]);
    }
  }
}

struct JitterJob {        // BBD_JITTER_JOB int32 fields
  int32_t src_lo, src_hi;          // byte offset of the uint8 HWC image in src
  int32_t dst_lo, dst_hi;          // float offset of the [3,H,W] destination in dst
  int32_t op[4];                   // BBD_JIT_* in application order, -1 = skip
  int32_t factor_bits[4];          // float bits: factor per slot (hue slot: the uint8 offset as int)
};
static_assert(sizeof(JitterJob) == BBD_JITTER_JOB * 4, "job layout");

__device__ __forceinline__ void apply_ops(const JitterJob& jb, int first, int last, uint8_t mean_l, uint8_t* r,
                                          uint8_t* g, uint8_t* b) {
  for (int i = first; i < last; ++i) {
    const int op = jb.op[i];
    if (op < 0) continue;
    bbd_img_jitter_op(op, __int_as_float(jb.factor_bits[i]), jb.factor_bits[i], mean_l, r, g, b);
  }
}

__device__ __forceinline__ int contrast_slot(const JitterJob& jb) {
  for (int i = 0; i < 4; ++i)
    if (jb.op[i] == BBD_JIT_CONTRAST) return i;
  return 4;
}

// Pass 1: ops before the contrast op; image-wide sum of L of that intermediate image (integer atomics
// => deterministic).  Images without a contrast op skip this pass entirely.
__global__ __launch_bounds__(NT) void jitter_sum_kernel(const uint8_t* src, const JitterJob* jobs, uint32_t* lsum,
                                                        int npx) {
  const JitterJob jb = jobs[blockIdx.y];
  const int cs = contrast_slot(jb);
  if (cs == 4) return;
  const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
  uint32_t local = 0;
  for (int p = blockIdx.x * NT + threadIdx.x; p < npx; p += gridDim.x * NT) {
    uint8_t r = in[3 * (size_t)p], g = in[3 * (size_t)p + 1], b = in[3 * (size_t)p + 2];
    apply_ops(jb, 0, cs, 0, &r, &g, &b);
    local += bbd_img_luma(r, g, b);
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&lsum[blockIdx.y], local);
}

// Pass 2: the whole sequence with the now-known mean level, then ToTensor (x / 255, HWC -> CHW).
__global__ __launch_bounds__(NT) void jitter_apply_kernel(const uint8_t* src, float* dst, const JitterJob* jobs,
                                                          const uint32_t* lsum, int npx) {
  const JitterJob jb = jobs[blockIdx.y];
  const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
  float* out = dst + off64(jb.dst_lo, jb.dst_hi);
  const int cs = contrast_slot(jb);
  const uint8_t mean_l = cs == 4 ? 0 : bbd_img_mean_level(lsum[blockIdx.y], (uint64_t)npx);
  for (int p = blockIdx.x * NT + threadIdx.x; p < npx; p += gridDim.x * NT) {
    uint8_t r = in[3 * (size_t)p], g = in[3 * (size_t)p + 1], b = in[3 * (size_t)p + 2];
    apply_ops(jb, 0, 4, mean_l, &r, &g, &b);
    out[p] = (float)r / 255.0f;                       // torchvision to_tensor: .div(255)
    out[(size_t)npx + p] = (float)g / 255.0f;
    out[2 * (size_t)npx + p] = (float)b / 255.0f;
  }
}

__global__ __launch_bounds__(NT) void to_float_kernel(const uint8_t* src, float* dst, const int32_t* jobs, int npx) {
  const int32_t* jb = jobs + (size_t)blockIdx.y * BBD_CONVERT_JOB;
  const uint8_t* in = src + off64(jb[0], jb[1]);
  float* out = dst + off64(jb[2], jb[3]);
  for (int p = blockIdx.x * NT + threadIdx.x; p < npx; p += gridDim.x * NT) {
    out[p] = (float)in[3 * (size_t)p] / 255.0f;
    out[(size_t)npx + p] = (float)in[3 * (size_t)p + 1] / 255.0f;
    out[2 * (size_t)npx + p] = (float)in[3 * (size_t)p + 2] / 255.0f;
  }
}

int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int bbd_resample_h_u8(const uint8_t* src, uint8_t* dst, const int32_t* jobs, int n_jobs, int max_in_h,
                      const int32_t* coef, const int32_t* bounds, int channels, void* stream) {
  if (!src || !dst || !jobs || !coef || !bounds || n_jobs <= 0 || max_in_h <= 0 || channels <= 0) return BBD_E_BADARG;
  const dim3 grid((unsigned)((max_in_h + RH - 1) / RH), (unsigned)n_jobs);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ResampleJob* jt = reinterpret_cast<const ResampleJob*>(jobs);
  if (channels == 3) hipLaunchKernelGGL(resample_h_kernel<3>, grid, dim3(NT), 0, st, src, dst, jt, coef, bounds);
  else if (channels == 1) hipLaunchKernelGGL(resample_h_kernel<1>, grid, dim3(NT), 0, st, src, dst, jt, coef, bounds);
  else return BBD_E_BADARG;
  return status();
}

int bbd_resample_v_u8(const uint8_t* src, uint8_t* dst, const int32_t* jobs, int n_jobs, int max_out_h,
                      int max_row_bytes, const int32_t* coef, const int32_t* bounds, int channels, void* stream) {
  if (!src || !dst || !jobs || !coef || !bounds || n_jobs <= 0 || max_out_h <= 0 || max_row_bytes <= 0 || channels <= 0)
    return BBD_E_BADARG;
  hipLaunchKernelGGL(resample_v_kernel,
                     dim3((unsigned)((max_row_bytes + NT * 4 - 1) / (NT * 4)), (unsigned)((max_out_h + RV - 1) / RV),
                          (unsigned)n_jobs),
                     dim3(NT), 0, static_cast<hipStream_t>(stream), src, dst,
                     reinterpret_cast<const ResampleJob*>(jobs), coef, bounds, channels);
  return status();
}

int bbd_color_jitter_u8(const uint8_t* src, float* dst, const int32_t* jobs, int n_jobs, int H, int W,
                        uint32_t* lsum_scratch, void* stream) {
  if (!src || !dst || !jobs || !lsum_scratch || n_jobs <= 0 || H <= 0 || W <= 0) return BBD_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int npx = H * W;
  const unsigned gx = (unsigned)((npx + NT * 4 - 1) / (NT * 4));
  hipError_t e = hipMemsetAsync(lsum_scratch, 0, sizeof(uint32_t) * (size_t)n_jobs, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(jitter_sum_kernel, dim3(gx, (unsigned)n_jobs), dim3(NT), 0, st, src,
                     reinterpret_cast<const JitterJob*>(jobs), lsum_scratch, npx);
  hipLaunchKernelGGL(jitter_apply_kernel, dim3(gx, (unsigned)n_jobs), dim3(NT), 0, st, src, dst,
                     reinterpret_cast<const JitterJob*>(jobs), lsum_scratch, npx);
  return status();
}

int bbd_u8_to_float_chw(const uint8_t* src, float* dst, const int32_t* jobs, int n_jobs, int H, int W, void* stream) {
  if (!src || !dst || !jobs || n_jobs <= 0 || H <= 0 || W <= 0) return BBD_E_BADARG;
  const int npx = H * W;
  hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((npx + NT * 4 - 1) / (NT * 4)), (unsigned)n_jobs), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), src, dst, jobs, npx);
  return status();
}

}  // extern "C"
