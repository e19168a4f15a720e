/*
 * bbd_math.h - per-pixel arithmetic of the photometric hot path, shared by the HIP kernels
 * (bbd_kernels.hip) and by the host-side port that the CPU tests build with g++
 * (tests/host_port/).  Everything here is scalar fp32 and is written to round exactly like
 * the reference's CPU op sequence (PyTorch eager: one rounding per elementwise op, no FMA
 * across ops), so compile with -ffp-contract=off; explicit fmaf() calls are intentional.
 *
 * Reference formulas: SURVEY.md Appendix A (layers.py:136-249, trainer.py:477-486 and the
 * ATen grid_sampler_2d / avg_pool2d / reflection_pad2d CPU kernels it calls).
 */
#ifndef BBD_MATH_H
#define BBD_MATH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BBD_HD __host__ __device__ __forceinline__
#else
#define BBD_HD static inline
#endif

#define BBD_C1 9.999999747378752e-05f   /* float(0.01**2) */
#define BBD_C2 0.0008999999845400453f   /* float(0.03**2) */
#define BBD_EPS 1.0000000116860974e-07f /* Project3D eps */

/* Correctly rounded x/9 and x/3 (avg_pool2d divides the window sum by 9; mean over the 3
 * colour channels divides by 3).  q + fma(-d,q,x)*r is the correctly rounded quotient when
 * r = RN(1/d) (Markstein); checked exhaustively against IEEE division in tests. */
BBD_HD float bbd_div9(float x) {
  const float r = 1.0f / 9.0f;
  float q = x * r;
  return fmaf(fmaf(-9.0f, q, x), r, q);
}
BBD_HD float bbd_div3(float x) {
  const float r = 1.0f / 3.0f;
  float q = x * r;
  return fmaf(fmaf(-3.0f, q, x), r, q);
}

/* ReflectionPad2d(1) index map, then clamped so that partial tiles never read out of range. */
BBD_HD int bbd_reflect(int i, int n) {
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - 2 - i : i;
  i = i < 0 ? 0 : i;
  return i > n - 1 ? n - 1 : i;
}

/* ---------------------------------------------------------------- IEEE division, cheaper
 * PyTorch's CPU ops divide with correctly rounded IEEE division, so the kernels must too.  hipcc's
 * expansion of `a / b` is v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas, v_div_fixup; the scale /
 * fixup instructions only matter for extreme exponents.  For operands of moderate exponent the
 * same Newton step + two residual corrections give the identical correctly rounded quotient, and
 * quotients that share a denominator share the refined reciprocal.  Lanes outside the safe
 * exponent window (or non-finite) take the compiler's full sequence.  On the host: plain `/`.
 * Checked against `/` on 2^28 random operand pairs by tests/test_gpu_parity.py. */
#if defined(__HIP_DEVICE_COMPILE__)
BBD_HD int bbd_exp_ok(float x) {           /* 2^-57 <= |x| < 2^58: any quotient of two such values,
                                              and every intermediate, stays a normal number */
  const unsigned e = (__float_as_uint(x) >> 23) & 0xffu;
  return e >= 70u && e <= 184u;
}
BBD_HD float bbd_rcp_refined(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
BBD_HD float bbd_div_with(float n, float d, float r) {
  float q = n * r;
  q = fmaf(fmaf(-d, q, n), r, q);
  return fmaf(fmaf(-d, q, n), r, q);
}
/* all three magnitudes inside the window with two 3-input integer min/max (a zero numerator simply
 * takes the exact fallback) */
BBD_HD int bbd_exp_ok3(float a, float b, float c) {
  const unsigned ia = __float_as_uint(a) & 0x7fffffffu, ib = __float_as_uint(b) & 0x7fffffffu,
                 ic = __float_as_uint(c) & 0x7fffffffu;
  const unsigned mx = max(max(ia, ib), ic), mn = min(min(ia, ib), ic);
  return mn >= (70u << 23) && mx < (185u << 23);
}
BBD_HD float bbd_div(float n, float d) {
  if (bbd_exp_ok3(n, d, d)) return bbd_div_with(n, d, bbd_rcp_refined(d));
  return n / d;
}
BBD_HD void bbd_div2(float n0, float n1, float d, float* q0, float* q1) {
  if (bbd_exp_ok3(n0, n1, d)) {
    const float r = bbd_rcp_refined(d);
    *q0 = bbd_div_with(n0, d, r);
    *q1 = bbd_div_with(n1, d, r);
  } else {
    *q0 = n0 / d;
    *q1 = n1 / d;
  }
}
/* division by a launch constant d whose reciprocal rd = RN(1/d) was computed on the host */
BBD_HD float bbd_div_const(float n, float d, float rd) {
  if (bbd_exp_ok(n) || n == 0.0f) return bbd_div_with(n, d, rd);
  return n / d;
}
BBD_HD float bbd_rcp_approx(float d) { return bbd_rcp_refined(d); }   /* <= 1 ulp; backward only */
/* The same refined-reciprocal sequences WITHOUT the exponent-window test: bit-identical to bbd_div2 /
 * bbd_div_const whenever those take their fast path (every operand of moderate exponent, i.e. every pixel whose
 * sampling coordinate is not astronomically large or exactly zero).  Used by the BACKWARD's warp recompute,
 * which must select the same texels as the forward: where the forward fell back to the full IEEE sequence the
 * coordinate is far outside the image or degenerate and clamps to the same border either way. */
BBD_HD void bbd_div2_unguarded(float n0, float n1, float d, float* q0, float* q1) {
  const float r = bbd_rcp_refined(d);
  *q0 = bbd_div_with(n0, d, r);
  *q1 = bbd_div_with(n1, d, r);
}
BBD_HD float bbd_div_const_unguarded(float n, float d, float rd) { return bbd_div_with(n, d, rd); }
#else
BBD_HD void bbd_div2_unguarded(float n0, float n1, float d, float* q0, float* q1) { *q0 = n0 / d; *q1 = n1 / d; }
BBD_HD float bbd_div_const_unguarded(float n, float d, float rd) { (void)rd; return n / d; }
BBD_HD float bbd_div(float n, float d) { return n / d; }
BBD_HD void bbd_div2(float n0, float n1, float d, float* q0, float* q1) { *q0 = n0 / d; *q1 = n1 / d; }
BBD_HD float bbd_div_const(float n, float d, float rd) { (void)rd; return n / d; }
BBD_HD float bbd_rcp_approx(float d) { return 1.0f / d; }
#endif

/* image size and the constants derived from it on the host */
struct BbdDims {
  int H, W;
  float hm1, wm1;    /* (float)(H-1), (float)(W-1) */
  float rh, rw;      /* RN(1/hm1), RN(1/wm1) */
};
BBD_HD BbdDims bbd_dims(int H, int W) {
  BbdDims d;
  d.H = H; d.W = W;
  d.hm1 = (float)(H - 1); d.wm1 = (float)(W - 1);
  d.rh = 1.0f / d.hm1; d.rw = 1.0f / d.wm1;
  return d;
}

/* Two-operation expressions of the reference whose second operation is an exact scaling by 2 or 1/2: one FMA gives
 * the same bits, because scaling by a power of two commutes with rounding (no overflow / subnormal in range):
 *   (n - 0.5) * 2      == RN(2 n - 1)      = fma(n, 2, -1)        layers.py:193
 *   (g + 1) / 2        == RN(g / 2 + 1/2)  = fma(g, 0.5, 0.5)     ATen grid_sampler_unnormalize (align_corners)
 *   2 * a * b + C      == RN(2 (a b) + C)  = fma(RN(a b), 2, C)   layers.py:244 (2 a is exact, so (2a) b == 2 (a b))
 * Checked bit for bit against the reference's golden vectors on both tiers. */
BBD_HD float bbd_norm_to_grid(float n) { return fmaf(n, 2.0f, -1.0f); }
BBD_HD float bbd_grid_to_unit(float g) { return fmaf(g, 0.5f, 0.5f); }

/* ---------------------------------------------------------------- projection (A2, A3, A4) */
struct BbdSample {
  float ix, iy;      /* clamped source coordinates in pixels */
  float u, v;        /* q.x/(z+eps), q.y/(z+eps) */
  float zi;          /* z + eps */
  float cx, cy, cz;  /* inv_K[:3,:3] . (x, y, 1) */
  float X, Y, Z;     /* depth * c */
  int clipx, clipy;  /* border clamp active -> zero coordinate gradient */
};

/* One row of the pose table (BBD_POSE_STRIDE floats): K[:3,:] (12) | T (16) | inv_K[:3,:3] (9) | pad.
 * Expands it to proj[21] = P (3x4, = (K@T)[:3,:]) followed by inv_K[:3,:3].
 * torch.matmul of two 4x4 matrices on the reference's CPU path accumulates
 * ((0 + a0*b0) + a1*b1) + ... with one rounding per operation (measured). */
BBD_HD void bbd_make_proj(const float* row, float proj[21]) {
  const float* K = row;
  const float* T = row + 12;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = K[i * 4 + 0] * T[j];
      acc = acc + K[i * 4 + 1] * T[4 + j];
      acc = acc + K[i * 4 + 2] * T[8 + j];
      acc = acc + K[i * 4 + 3] * T[12 + j];
      proj[i * 4 + j] = acc;
    }
  for (int i = 0; i < 9; ++i) proj[12 + i] = row[28 + i];
}

/* 3-term / 4-term dot products of the batched [3,k]x[k,N] matmuls (layers.py:163, :185): the
 * reference's CPU bmm accumulates with FMA in k order, acc = a0*b0; acc = fma(ak, bk, acc)
 * (measured bit-exact against the golden vectors). */
BBD_HD float bbd_dot3_hom(const float* a, float x, float y) {   /* a . (x, y, 1) */
  return fmaf(a[2], 1.0f, fmaf(a[1], y, a[0] * x));
}
BBD_HD float bbd_dot4_hom(const float* a, float x, float y, float z) {   /* a . (x, y, z, 1) */
  return fmaf(a[3], 1.0f, fmaf(a[2], z, fmaf(a[1], y, a[0] * x)));
}

/* Camera-space point of pixel (xx, yy) at `depth` (layers.py:163-164  cam = depth * (inv_K[:3,:3] @ [x,y,1])): depends on
 * the pixel and on inv_K only, not on the candidate's pose. */
BBD_HD void bbd_backproject(const float* iK, int xx, int yy, float depth, BbdSample* o) {
  const float fx = (float)xx, fy = (float)yy;
  o->cx = bbd_dot3_hom(iK, fx, fy);
  o->cy = bbd_dot3_hom(iK + 3, fx, fy);
  o->cz = bbd_dot3_hom(iK + 6, fx, fy);
  o->X = depth * o->cx;
  o->Y = depth * o->cy;
  o->Z = depth * o->cz;
}

/* Projection of the camera-space point already in o->X/Y/Z through P (3x4 row-major) to clamped source coordinates. */
BBD_HD void bbd_project_point(const float* P, const BbdDims& dm, BbdSample* o) {
  /* layers.py:185  q = P @ [X,Y,Z,1] */
  const float qx = bbd_dot4_hom(P, o->X, o->Y, o->Z);
  const float qy = bbd_dot4_hom(P + 4, o->X, o->Y, o->Z);
  const float qz = bbd_dot4_hom(P + 8, o->X, o->Y, o->Z);
  o->zi = qz + BBD_EPS;                       /* layers.py:188 */
  const float wm1 = dm.wm1, hm1 = dm.hm1;
#if defined(__HIP_DEVICE_COMPILE__)
  /* All four IEEE divisions through the refined-reciprocal sequence first, ONE validity test for the lot, and the
   * full IEEE sequence only for lanes outside the exponent window (identical results: both forms are correctly
   * rounded where the fast one is valid) - one rarely-taken branch per pixel instead of three with inlined
   * v_div_scale/fmas/fixup fallbacks. */
  bbd_div2_unguarded(qx, qy, o->zi, &o->u, &o->v);
  float nu = bbd_div_const_unguarded(o->u, wm1, dm.rw), nv = bbd_div_const_unguarded(o->v, hm1, dm.rh);
  const int ok = bbd_exp_ok3(qx, qy, o->zi) & (bbd_exp_ok(o->u) | (o->u == 0.0f)) & (bbd_exp_ok(o->v) | (o->v == 0.0f));
  if (!ok) {
    o->u = qx / o->zi;
    o->v = qy / o->zi;
    nu = o->u / wm1;
    nv = o->v / hm1;
  }
  const float gx = bbd_norm_to_grid(nu);
  const float gy = bbd_norm_to_grid(nv);
#else
  bbd_div2(qx, qy, o->zi, &o->u, &o->v);
  /* layers.py:191-193 normalise, then ATen grid_sampler unnormalise (align_corners=True) */
  const float gx = bbd_norm_to_grid(bbd_div_const(o->u, wm1, dm.rw));
  const float gy = bbd_norm_to_grid(bbd_div_const(o->v, hm1, dm.rh));
#endif
  float ix = bbd_grid_to_unit(gx) * wm1;
  float iy = bbd_grid_to_unit(gy) * hm1;
  /* border padding: clip_coordinates; gradient is zeroed when the clamp is active */
  o->clipx = !(ix > 0.0f && ix < wm1);
  o->clipy = !(iy > 0.0f && iy < hm1);
  ix = ix > 0.0f ? ix : 0.0f;   /* max(ix, 0): NaN -> 0 like std::max(NaN,0) argument order */
  iy = iy > 0.0f ? iy : 0.0f;
  o->ix = ix < wm1 ? ix : wm1;
  o->iy = iy < hm1 ? iy : hm1;
}

/* proj = 21 floats: P (3x4 row-major), inv_K[:3,:3] (row-major). */
BBD_HD void bbd_project(const float* proj, int xx, int yy, float depth, const BbdDims& dm, BbdSample* o) {
  bbd_backproject(proj + 12, xx, yy, depth, o);
  bbd_project_point(proj, dm, o);
}

/* bbd_project for the backward's warp recompute: the same operation sequence with the unguarded divisions. */
BBD_HD void bbd_project_bwd(const float* proj, int xx, int yy, float depth, const BbdDims& dm, BbdSample* o) {
  const float* P = proj;
  const float* iK = proj + 12;
  const float fx = (float)xx, fy = (float)yy;
  o->cx = bbd_dot3_hom(iK, fx, fy);
  o->cy = bbd_dot3_hom(iK + 3, fx, fy);
  o->cz = bbd_dot3_hom(iK + 6, fx, fy);
  o->X = depth * o->cx;
  o->Y = depth * o->cy;
  o->Z = depth * o->cz;
  const float qx = bbd_dot4_hom(P, o->X, o->Y, o->Z);
  const float qy = bbd_dot4_hom(P + 4, o->X, o->Y, o->Z);
  const float qz = bbd_dot4_hom(P + 8, o->X, o->Y, o->Z);
  o->zi = qz + BBD_EPS;
  bbd_div2_unguarded(qx, qy, o->zi, &o->u, &o->v);
  const float wm1 = dm.wm1, hm1 = dm.hm1;
  const float gx = bbd_norm_to_grid(bbd_div_const_unguarded(o->u, wm1, dm.rw));
  const float gy = bbd_norm_to_grid(bbd_div_const_unguarded(o->v, hm1, dm.rh));
  float ix = bbd_grid_to_unit(gx) * wm1;
  float iy = bbd_grid_to_unit(gy) * hm1;
  o->clipx = !(ix > 0.0f && ix < wm1);
  o->clipy = !(iy > 0.0f && iy < hm1);
  ix = ix > 0.0f ? ix : 0.0f;
  iy = iy > 0.0f ? iy : 0.0f;
  o->ix = ix < wm1 ? ix : wm1;
  o->iy = iy < hm1 ? iy : hm1;
}

/* Backward only: the smooth factors of the projection chain rule (no texel selection depends on them,
 * so the refined-reciprocal division is enough); the clamp is already folded into the stored coordinate
 * derivatives, hence clipx = clipy = 0 here. */
BBD_HD void bbd_sample_smooth(const float* proj, int xx, int yy, float depth, BbdSample* o) {
  const float* P = proj;
  const float* iK = proj + 12;
  const float fx = (float)xx, fy = (float)yy;
  o->cx = bbd_dot3_hom(iK, fx, fy);
  o->cy = bbd_dot3_hom(iK + 3, fx, fy);
  o->cz = bbd_dot3_hom(iK + 6, fx, fy);
  o->X = depth * o->cx;
  o->Y = depth * o->cy;
  o->Z = depth * o->cz;
  const float qx = bbd_dot4_hom(P, o->X, o->Y, o->Z);
  const float qy = bbd_dot4_hom(P + 4, o->X, o->Y, o->Z);
  o->zi = bbd_dot4_hom(P + 8, o->X, o->Y, o->Z) + BBD_EPS;
  const float rz = bbd_rcp_approx(o->zi);
  o->u = qx * rz;
  o->v = qy * rz;
  o->ix = o->u;
  o->iy = o->v;
  o->clipx = 0;
  o->clipy = 0;
}

struct BbdTaps {
  int i0, i1;        /* element offsets (inside one channel plane) of the north and south texel pairs */
  float w, e, n, s;  /* distances: w = ix-x0, e = 1-w, n = iy-y0, s = 1-n */
};

/* Tap geometry for clamped coordinates 0 <= ix <= W-1, 0 <= iy <= H-1.  The east / south taps fall
 * outside the image only when ix == W-1 / iy == H-1 exactly, where their weight is 0 and ATen masks
 * them to 0.  Instead of masking values: at ix == W-1 the pair is taken one texel to the left with
 * w = 1 (so nw = s*0, ne = s*1 multiplies the same texel by the same weight), and at iy == H-1
 * the south row index is clamped (weights n*e = n*w = 0).  Adding 0*finite terms in the FMA chain
 * leaves the result bit-identical to the masked form. */
BBD_HD void bbd_taps(float ix, float iy, const BbdDims& dm, BbdTaps* t) {
  const float x0f = floorf(ix), y0f = floorf(iy);
  int x0 = (int)x0f;
  const int y0 = (int)y0f;
  t->w = ix - x0f;
  if (x0 > dm.W - 2) { x0 = dm.W - 2; t->w = 1.0f; }
  t->e = 1.0f - t->w;
  t->n = iy - y0f;
  t->s = 1.0f - t->n;
  const int y1 = y0 + 1 < dm.H ? y0 + 1 : dm.H - 1;
  t->i0 = y0 * dm.W + x0;
  t->i1 = y1 * dm.W + x0;
}

/* The four texel values (nw, ne, sw, se) of one channel plane. */
BBD_HD void bbd_fetch4(const float* plane, const BbdTaps* t, float v[4]) {
  /* (SGPR-base + 32-bit-offset addressing measured neutral to slower: profiles/r02/gather_saddr_variants.txt) */
  const float* r0 = plane + t->i0;
  const float* r1 = plane + t->i1;
  v[0] = r0[0]; v[1] = r0[1]; v[2] = r1[0]; v[3] = r1[1];
}

/* ATen grid_sampler_2d (CPU, bilinear): nw*a + ne*b + sw*c + se*d evaluated as an FMA chain in
 * that order - measured bit-exact against the reference's golden vectors. */
BBD_HD float bbd_bilerp(const float v[4], const BbdTaps* t) {
  const float nw = t->s * t->e, ne = t->s * t->w, sw = t->n * t->e, se = t->n * t->w;
  return fmaf(v[3], se, fmaf(v[2], sw, fmaf(v[1], ne, v[0] * nw)));
}

/* ---------------------------------------------------------------- SSIM + L1 (A5) */
/* statistics of the target window (computed once per pixel, reused by every candidate) */
BBD_HD void bbd_ystats(float sy, float syy, float* mu_y, float* sig_y) {
  *mu_y = bbd_div9(sy);
  *sig_y = bbd_div9(syy) - (*mu_y) * (*mu_y);
}

/* numerator and denominator of the SSIM ratio (layers.py:241-246), then the clamp of (1 - n/d)/2 */
BBD_HD void bbd_ssim_nd(float sx, float sxx, float sxy, float mu_y, float sig_y, float* n, float* d) {
  const float mu_x = bbd_div9(sx);
  const float mxx = mu_x * mu_x, mxy = mu_x * mu_y;
  const float sig_x = bbd_div9(sxx) - mxx;
  const float sig_xy = bbd_div9(sxy) - mxy;
  *n = fmaf(mxy, 2.0f, BBD_C1) * fmaf(sig_xy, 2.0f, BBD_C2);      /* (2 mu_x mu_y + C1)(2 sig_xy + C2), see above */
  *d = (mxx + mu_y * mu_y + BBD_C1) * (sig_x + sig_y + BBD_C2);
}
BBD_HD float bbd_ssim_from_ratio(float q) {
  const float v = (1.0f - q) / 2.0f;
  return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);   /* NaN stays NaN like torch.clamp */
}
BBD_HD float bbd_ssim(float sx, float sxx, float sxy, float mu_y, float sig_y) {
  float n, d;
  bbd_ssim_nd(sx, sxx, sxy, mu_y, sig_y, &n, &d);
  return bbd_ssim_from_ratio(bbd_div(n, d));
}
/* 0.85 * mean_c(ssim) + 0.15 * mean_c(|y - x|)   (trainer.py:477-486) */
BBD_HD float bbd_combine(const float ssim[3], const float l1[3], int no_ssim) {
  const float l1m = bbd_div3(l1[0] + l1[1] + l1[2]);
  if (no_ssim) return l1m;
  const float sm = bbd_div3(ssim[0] + ssim[1] + ssim[2]);
  return 0.85f * sm + 0.15f * l1m;
}

/* torch.min(dim) update: strictly smaller wins (first index on ties), NaN wins and sticks. */
BBD_HD void bbd_min_update(float cand, int id, float* best, int* arg) {
  const int take = (cand < *best) || ((cand != cand) && (*best == *best));
  *best = take ? cand : *best;
  *arg = take ? id : *arg;
}

/* d ssim_loss / d{mu_x, E[x^2], E[xy]} at one pixel: dv/dx_i = (A + B*x_i + C*y_i) / 9 for each
 * of the nine window texels (reflection included).  Zero outside the clamp's open range. */
BBD_HD void bbd_ssim_grad(float sx, float sxx, float sxy, float mu_y, float sig_y,
                          float* A, float* B, float* C) {
  const float mu_x = bbd_div9(sx);
  const float sig_x = bbd_div9(sxx) - mu_x * mu_x;
  const float sig_xy = bbd_div9(sxy) - mu_x * mu_y;
  const float n1 = 2.0f * mu_x * mu_y + BBD_C1, n2 = 2.0f * sig_xy + BBD_C2;
  const float d1 = mu_x * mu_x + mu_y * mu_y + BBD_C1, d2 = sig_x + sig_y + BBD_C2;
  const float n = n1 * n2, d = d1 * d2;
  const float rd = bbd_rcp_approx(d);
  /* gradient arithmetic: the <= 1 ulp reciprocal is enough here (the forward's IEEE division fixes the VALUE of the
   * loss; this only decides on which side of the clamp a pixel sits, and differs from it on a set of measure zero) */
  const float v = (1.0f - n * rd) / 2.0f;
  if (!(v >= 0.0f && v <= 1.0f)) {   /* clamp passes gradient on [0,1] only */
    *A = 0.0f; *B = 0.0f; *C = 0.0f;
    return;
  }
  const float dvdn = -0.5f * rd;
  const float dvdd = 0.5f * n * rd * rd;
  *A = dvdn * (2.0f * mu_y * (n2 - n1)) + dvdd * (2.0f * mu_x * (d2 - d1));
  *B = 2.0f * (dvdd * d1);
  *C = dvdn * (2.0f * n1);
}

/* Adjoint multiplicity of ReflectionPad2d(1)+AvgPool(3,1) in one dimension: how many of the
 * three window offsets of output pixel p land on input pixel q (|p-q| <= 1 assumed). */
BBD_HD int bbd_reflect_mult(int q, int p, int n) {
  int m = 1;
  if (q == 1 && p == 0) m += 1;
  if (q == n - 2 && p == n - 1) m += 1;
  return m;
}

/* Coordinate gradient of the bilinear sample of one channel (ATen grid_sampler_2d backward,
 * CPU form):  d out/d ix = (ne-nw)*s + (se-sw)*n ;  d out/d iy = (sw-nw)*e + (se-ne)*w. */
BBD_HD void bbd_bilerp_grad(const float v[4], const BbdTaps* t, float g, float* gix, float* giy) {
  *gix += g * ((v[1] - v[0]) * t->s + (v[3] - v[2]) * t->n);
  *giy += g * ((v[2] - v[0]) * t->e + (v[3] - v[1]) * t->w);
}

/* Chain (gix, giy) back to depth and the 12 entries of P. */
BBD_HD void bbd_project_grad(const float* proj, const BbdSample* sm, float gix, float giy,
                             float* gdepth, float gP[12]) {
  const float* P = proj;
  const float du = sm->clipx ? 0.0f : gix;
  const float dv = sm->clipy ? 0.0f : giy;
  const float rz = bbd_rcp_approx(sm->zi);
  const float gq0 = du * rz, gq1 = dv * rz;
  const float gq2 = -(du * sm->u + dv * sm->v) * rz;
  const float gX = gq0 * P[0] + gq1 * P[4] + gq2 * P[8];
  const float gY = gq0 * P[1] + gq1 * P[5] + gq2 * P[9];
  const float gZ = gq0 * P[2] + gq1 * P[6] + gq2 * P[10];
  *gdepth = gX * sm->cx + gY * sm->cy + gZ * sm->cz;
  gP[0] = gq0 * sm->X; gP[1] = gq0 * sm->Y; gP[2] = gq0 * sm->Z; gP[3] = gq0;
  gP[4] = gq1 * sm->X; gP[5] = gq1 * sm->Y; gP[6] = gq1 * sm->Z; gP[7] = gq1;
  gP[8] = gq2 * sm->X; gP[9] = gq2 * sm->Y; gP[10] = gq2 * sm->Z; gP[11] = gq2;
}

/* ---------------------------------------------------------------- disp -> depth (A1) */
/* upsample_bilinear2d, align_corners=False: source index and lambda for output index o. */
BBD_HD void bbd_up_src(int o, int in_size, int out_size, int* i0, int* i1, float* l0, float* l1) {
  const float scale = (float)in_size / (float)out_size;
  float src = scale * ((float)o + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  int i = (int)src;
  i = i < in_size - 1 ? i : in_size - 1;
  *i0 = i;
  *i1 = i + (i < in_size - 1 ? 1 : 0);
  *l1 = src - (float)i;
  *l0 = 1.0f - *l1;
}

/* The 2x2 blend of upsample_bilinear2d as the reference's CPU path rounds it.  ATen picks one of
 * two kernels by OUTPUT size (UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d: H + W <= 128):
 *   small outputs : four products of the separable weights accumulated as
 *                   w01*b, fma(w00,a), fma(w10,c), fma(w11,d);
 *   otherwise     : rows first, top = fma(lx0,a, lx1*b), bot likewise, out = fma(ly0,top, ly1*bot).
 * Both orderings were identified by exhaustive search against the live reference and reproduce
 * it bit for bit (tests/test_host_port_parity.py). */
BBD_HD float bbd_up_blend(float a, float b, float c, float d, float ly0, float ly1, float lx0, float lx1,
                          int small_output) {
  if (small_output) {
    float acc = (ly0 * lx1) * b;
    acc = fmaf(ly0 * lx0, a, acc);
    acc = fmaf(ly1 * lx0, c, acc);
    return fmaf(ly1 * lx1, d, acc);
  }
  const float top = fmaf(lx0, a, lx1 * b);
  const float bot = fmaf(lx0, c, lx1 * d);
  return fmaf(ly0, top, ly1 * bot);
}

#endif /* BBD_MATH_H */
