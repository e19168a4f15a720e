/*
 * bbd_image_math.h - scalar byte/integer arithmetic of the loader's image pipeline,
 * shared by the HIP kernels (bbd_image.hip) and the test-only host port.
 *
 * The reference prepares every training image on CPU workers with Pillow through torchvision
 * (datasets/mono_dataset.py:186-205): Resize(LANCZOS) chain, ColorJitter, ToTensor.  The arithmetic
 * lives in Pillow (third-party; this image has Pillow 12.2, the reference's environment pins 8.x - the
 * 8-bit paths below are unchanged between them) and is restated here from its published C sources:
 *   Resample.c  ImagingResample 8bpc : fixed-point coefficients (22 bits), +0.5 bias, clip8
 *   Blend.c     ImagingBlend         : ImageEnhance.{Brightness,Contrast,Color}.enhance
 *   Convert.c   rgb2l (L24), rgb2hsv_row, hsv2rgb
 * Every function is pinned EXHAUSTIVELY against the installed Pillow by tests/test_image_port.py
 * (all 2^24 RGB / HSV triples, all 2^16 blend operand pairs).
 */
#ifndef BBD_IMAGE_MATH_H
#define BBD_IMAGE_MATH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BBD_IHD __host__ __device__ __forceinline__
#else
#define BBD_IHD static inline
#endif

#define BBD_RESAMPLE_PRECISION 22   /* Resample.c: PRECISION_BITS = 32 - 8 - 2 */

/* Resample.c clip8(): table lookup of (acc >> PRECISION_BITS), saturating to [0,255]. */
BBD_IHD uint8_t bbd_img_clip8(int32_t acc) {
  const int32_t v = acc >> BBD_RESAMPLE_PRECISION;   /* arithmetic shift, like the C source */
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* Convert.c rgb2l: L24(rgb) >> 16. */
BBD_IHD uint8_t bbd_img_luma(uint8_t r, uint8_t g, uint8_t b) {
  return (uint8_t)(((uint32_t)r * 19595u + (uint32_t)g * 38470u + (uint32_t)b * 7471u + 0x8000u) >> 16);
}

/* Blend.c: out = in1 + alpha * (in2 - in1), float; truncation inside [0,1], clipped extrapolation
 * outside.  ImageEnhance calls it with in1 = degenerate image, in2 = image, alpha = factor. */
BBD_IHD uint8_t bbd_img_blend(uint8_t degenerate, uint8_t image, float alpha) {
  const float prod = alpha * (float)((int)image - (int)degenerate);
  const float t = (float)(int)degenerate + prod;
  if (alpha >= 0.0f && alpha <= 1.0f) return (uint8_t)(int)t;
  if (t <= 0.0f) return 0;
  if (t >= 255.0f) return 255;
  return (uint8_t)(int)t;
}

/* Convert.c rgb2hsv_row (after colorsys.py); mixed float/double exactly as the C source. */
BBD_IHD void bbd_img_rgb2hsv(uint8_t r, uint8_t g, uint8_t b, uint8_t* oh, uint8_t* os, uint8_t* ov) {
  const uint8_t maxc = r > g ? (r > b ? r : b) : (g > b ? g : b);
  const uint8_t minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
  *ov = maxc;
  if (minc == maxc) {
    *oh = 0;
    *os = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = ((float)(maxc - r)) / cr;
  const float gc = ((float)(maxc - g)) / cr;
  const float bc = ((float)(maxc - b)) / cr;
  float h;
  if (r == maxc) h = bc - gc;
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
  *oh = (uint8_t)(ih < 0 ? 0 : (ih > 255 ? 255 : ih));
  *os = (uint8_t)(is < 0 ? 0 : (is > 255 ? 255 : is));
}

/* Convert.c hsv2rgb. */
BBD_IHD void bbd_img_hsv2rgb(uint8_t h, uint8_t s, uint8_t v, uint8_t* r, uint8_t* g, uint8_t* b) {
  if (s == 0) {
    *r = v; *g = v; *b = v;
    return;
  }
  const double hf = (double)(float)h * 6.0 / 255.0;
  const int i = (int)floor(hf);
  const float f = (float)(hf - (double)(float)i);
  const float fs = (float)((double)(float)s / 255.0);
  const double vd = (double)(float)v;
  int p = (int)floor(vd * (1.0 - (double)fs) + 0.5);                            /* C round(), x >= 0 */
  int q = (int)floor(vd * (1.0 - (double)fs * (double)f) + 0.5);
  int t = (int)floor(vd * (1.0 - (double)fs * (1.0 - (double)f)) + 0.5);
  p = p < 0 ? 0 : (p > 255 ? 255 : p);
  q = q < 0 ? 0 : (q > 255 ? 255 : q);
  t = t < 0 ? 0 : (t > 255 ? 255 : t);
  const uint8_t up = (uint8_t)p, uq = (uint8_t)q, ut = (uint8_t)t;
  switch (i % 6) {
    case 0: *r = v;  *g = ut; *b = up; break;
    case 1: *r = uq; *g = v;  *b = up; break;
    case 2: *r = up; *g = v;  *b = ut; break;
    case 3: *r = up; *g = uq; *b = v;  break;
    case 4: *r = ut; *g = up; *b = v;  break;
    default: *r = v; *g = up; *b = uq; break;
  }
}

/* torchvision ColorJitter ops on one RGB pixel (functional_pil.py, torchvision 0.9):
 *   0 brightness : blend(black, img, f)          2 saturation : blend(L(img), img, f)
 *   1 contrast   : blend(mean_L, img, f)          3 hue        : HSV, h += uint8(f * 255), back
 * `param` is f for ops 0-2 and the uint8 hue offset for op 3; `mean_l` the image-wide
 * int(mean(L) + 0.5) for op 1. */
#define BBD_JIT_BRIGHTNESS 0
#define BBD_JIT_CONTRAST 1
#define BBD_JIT_SATURATION 2
#define BBD_JIT_HUE 3

BBD_IHD void bbd_img_jitter_op(int op, float factor, int hue_off, uint8_t mean_l, uint8_t* r, uint8_t* g, uint8_t* b) {
  if (op == BBD_JIT_BRIGHTNESS) {
    *r = bbd_img_blend(0, *r, factor);
    *g = bbd_img_blend(0, *g, factor);
    *b = bbd_img_blend(0, *b, factor);
  } else if (op == BBD_JIT_CONTRAST) {
    *r = bbd_img_blend(mean_l, *r, factor);
    *g = bbd_img_blend(mean_l, *g, factor);
    *b = bbd_img_blend(mean_l, *b, factor);
  } else if (op == BBD_JIT_SATURATION) {
    const uint8_t l = bbd_img_luma(*r, *g, *b);
    *r = bbd_img_blend(l, *r, factor);
    *g = bbd_img_blend(l, *g, factor);
    *b = bbd_img_blend(l, *b, factor);
  } else if (op == BBD_JIT_HUE) {
    uint8_t h, s, v;
    bbd_img_rgb2hsv(*r, *g, *b, &h, &s, &v);
    h = (uint8_t)(h + (uint8_t)hue_off);          /* numpy uint8 wrap-around add */
    bbd_img_hsv2rgb(h, s, v, r, g, b);
  }
}

/* ImageStat.Stat(L).mean[0] -> int(mean + 0.5)  (ImageEnhance.Contrast.__init__) */
BBD_IHD uint8_t bbd_img_mean_level(uint64_t sum_l, uint64_t count) {
  return (uint8_t)(int)((double)sum_l / (double)count + 0.5);
}

#endif /* BBD_IMAGE_MATH_H */
