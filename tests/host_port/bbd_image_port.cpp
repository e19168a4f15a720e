// Host port of the loader's image kernels (TEST INFRASTRUCTURE - never loaded by the product).
// Same scalar arithmetic (baseboostdepth_amd/csrc/bbd_image_math.h) and the same job-table ABI as
// bbd_image.hip, as plain loops over host memory, so the CPU test tier can pin the arithmetic against
// Pillow and exercise the Python plumbing (job tables, coefficient tables) without a GPU.
#include <cstdint>
#include <cstring>

#include "../../include/bbd_hip.h"
#include "../../baseboostdepth_amd/csrc/bbd_image_math.h"

namespace {
struct RJob { int32_t src_lo, src_hi, dst_lo, dst_hi, in_h, in_w, out_size, ksize, coef_off, bounds_off, flags, pad; };
struct JJob { int32_t src_lo, src_hi, dst_lo, dst_hi, op[4], bits[4]; };
inline size_t off64(int32_t lo, int32_t hi) { return (size_t)(uint32_t)lo | ((size_t)(uint32_t)hi << 32); }
inline float as_float(int32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
}  // namespace

extern "C" {

int hp_resample_h_u8(const uint8_t* src, uint8_t* dst, const int32_t* jobs, int n_jobs, int, const int32_t* coef,
                     const int32_t* bounds, int C) {
  for (int i = 0; i < n_jobs; ++i) {
    const RJob& jb = reinterpret_cast<const RJob*>(jobs)[i];
    const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
    uint8_t* out = dst + off64(jb.dst_lo, jb.dst_hi);
    for (int row = 0; row < jb.in_h; ++row)
      for (int x = 0; x < jb.out_size; ++x) {
        const int xmin = bounds[jb.bounds_off + 2 * x], xmax = bounds[jb.bounds_off + 2 * x + 1];
        const int32_t* k = coef + jb.coef_off + (size_t)x * jb.ksize;
        for (int c = 0; c < C; ++c) {
          int32_t acc = 1 << (BBD_RESAMPLE_PRECISION - 1);
          for (int j = 0; j < xmax; ++j) {
            const int sx = (jb.flags & BBD_RESAMPLE_FLIP) ? jb.in_w - 1 - (xmin + j) : xmin + j;
            acc += (int32_t)in[((size_t)row * jb.in_w + sx) * C + c] * k[j];
          }
          out[((size_t)row * jb.out_size + x) * C + c] = bbd_img_clip8(acc);
        }
      }
  }
  return 0;
}

int hp_resample_v_u8(const uint8_t* src, uint8_t* dst, const int32_t* jobs, int n_jobs, int, int, const int32_t* coef,
                     const int32_t* bounds, int C) {
  for (int i = 0; i < n_jobs; ++i) {
    const RJob& jb = reinterpret_cast<const RJob*>(jobs)[i];
    const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
    uint8_t* out = dst + off64(jb.dst_lo, jb.dst_hi);
    const size_t rb = (size_t)jb.in_w * C;
    for (int y = 0; y < jb.out_size; ++y) {
      const int ymin = bounds[jb.bounds_off + 2 * y], ymax = bounds[jb.bounds_off + 2 * y + 1];
      const int32_t* k = coef + jb.coef_off + (size_t)y * jb.ksize;
      for (size_t b = 0; b < rb; ++b) {
        int32_t acc = 1 << (BBD_RESAMPLE_PRECISION - 1);
        for (int j = 0; j < ymax; ++j) acc += (int32_t)in[(size_t)(ymin + j) * rb + b] * k[j];
        out[(size_t)y * rb + b] = bbd_img_clip8(acc);
      }
    }
  }
  return 0;
}

int hp_color_jitter_u8(const uint8_t* src, float* dst, const int32_t* jobs, int n_jobs, int H, int W, uint32_t*) {
  const size_t npx = (size_t)H * W;
  for (int i = 0; i < n_jobs; ++i) {
    const JJob& jb = reinterpret_cast<const JJob*>(jobs)[i];
    const uint8_t* in = src + off64(jb.src_lo, jb.src_hi);
    float* out = dst + off64(jb.dst_lo, jb.dst_hi);
    int cs = 4;
    for (int s = 0; s < 4; ++s)
      if (jb.op[s] == BBD_JIT_CONTRAST) { cs = s; break; }
    uint8_t mean_l = 0;
    if (cs < 4) {
      uint64_t sum = 0;
      for (size_t p = 0; p < npx; ++p) {
        uint8_t r = in[3 * p], g = in[3 * p + 1], b = in[3 * p + 2];
        for (int s = 0; s < cs; ++s)
          if (jb.op[s] >= 0) bbd_img_jitter_op(jb.op[s], as_float(jb.bits[s]), jb.bits[s], 0, &r, &g, &b);
        sum += bbd_img_luma(r, g, b);
      }
      mean_l = bbd_img_mean_level(sum, npx);
    }
    for (size_t p = 0; p < npx; ++p) {
      uint8_t r = in[3 * p], g = in[3 * p + 1], b = in[3 * p + 2];
      for (int s = 0; s < 4; ++s)
        if (jb.op[s] >= 0) bbd_img_jitter_op(jb.op[s], as_float(jb.bits[s]), jb.bits[s], mean_l, &r, &g, &b);
      out[p] = (float)r / 255.0f;
      out[npx + p] = (float)g / 255.0f;
      out[2 * npx + p] = (float)b / 255.0f;
    }
  }
  return 0;
}

int hp_u8_to_float_chw(const uint8_t* src, float* dst, const int32_t* jobs, int n_jobs, int H, int W) {
  const size_t npx = (size_t)H * W;
  for (int i = 0; i < n_jobs; ++i) {
    const int32_t* jb = jobs + (size_t)i * BBD_CONVERT_JOB;
    const uint8_t* in = src + off64(jb[0], jb[1]);
    float* out = dst + off64(jb[2], jb[3]);
    for (size_t p = 0; p < npx; ++p)
      for (int c = 0; c < 3; ++c) out[c * npx + p] = (float)in[3 * p + c] / 255.0f;
  }
  return 0;
}

// Exhaustive tables for pinning against Pillow: out[(a<<16|b<<8|c)*3 ..]
void hp_img_rgb2hsv_all(uint8_t* out) {
  for (uint32_t v = 0; v < (1u << 24); ++v)
    bbd_img_rgb2hsv((uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v, out + 3 * (size_t)v, out + 3 * (size_t)v + 1,
                    out + 3 * (size_t)v + 2);
}
void hp_img_hsv2rgb_all(uint8_t* out) {
  for (uint32_t v = 0; v < (1u << 24); ++v)
    bbd_img_hsv2rgb((uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v, out + 3 * (size_t)v, out + 3 * (size_t)v + 1,
                    out + 3 * (size_t)v + 2);
}
void hp_img_luma_all(uint8_t* out) {
  for (uint32_t v = 0; v < (1u << 24); ++v) out[v] = bbd_img_luma((uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v);
}
void hp_img_blend_all(float alpha, uint8_t* out) {   // out[deg*256 + img]
  for (int d = 0; d < 256; ++d)
    for (int i = 0; i < 256; ++i) out[d * 256 + i] = bbd_img_blend((uint8_t)d, (uint8_t)i, alpha);
}

}  // extern "C"
