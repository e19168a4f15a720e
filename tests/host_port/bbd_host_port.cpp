// Host-side execution of the product's per-pixel arithmetic (baseboostdepth_amd/csrc/bbd_math.h).
//
// TEST-ONLY: built with g++ by tests/host_port.py so that the CPU test tier (no GPU in the build
// container) can check the exact math the HIP kernels run - projection, bilinear taps, SSIM/L1,
// arg-min, and the hand-derived backward - against the oracle before any GPU time is spent.
// The product never loads this library; it has the same C signatures as include/bbd_hip.h
// (prefix hp_, host pointers, no stream) but uses plain loops: the backward is written in
// SCATTER form, independently of the kernels' gather form.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../baseboostdepth_amd/csrc/bbd_math.h"
#include "../../include/bbd_hip.h"

namespace {

inline float texel(const float* img, int ch, int y, int x, int H, int W) {
  return img[((size_t)ch * H + bbd_reflect(y, H)) * W + bbd_reflect(x, W)];
}

// photometric loss at (y,x) between planar images px (prediction) and py (target)
float photometric(const float* px, const float* py, int y, int x, int H, int W, int no_ssim) {
  float ssim[3], l1[3];
  for (int ch = 0; ch < 3; ++ch) {
    float sx = 0, sxx = 0, sxy = 0, sy = 0, syy = 0;
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        const float xv = texel(px, ch, y + dr, x + dc, H, W), yv = texel(py, ch, y + dr, x + dc, H, W);
        sx += xv; sxx += xv * xv; sxy += xv * yv; sy += yv; syy += yv * yv;
      }
    float mu_y, sg_y;
    bbd_ystats(sy, syy, &mu_y, &sg_y);
    ssim[ch] = no_ssim ? 0.0f : bbd_ssim(sx, sxx, sxy, mu_y, sg_y);
    l1[ch] = fabsf(py[((size_t)ch * H + y) * W + x] - px[((size_t)ch * H + y) * W + x]);
  }
  return bbd_combine(ssim, l1, no_ssim);
}

void warp_image(const float* src, const float* depth, const float* proj, int H, int W, float* out) {
  const BbdDims dm = bbd_dims(H, W);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      BbdSample sm;
      bbd_project(proj, x, y, depth[(size_t)y * W + x], dm, &sm);
      BbdTaps t;
      bbd_taps(sm.ix, sm.iy, dm, &t);
      for (int ch = 0; ch < 3; ++ch) {
        float v[4];
        bbd_fetch4(src + (size_t)ch * H * W, &t, v);
        out[((size_t)ch * H + y) * W + x] = bbd_bilerp(v, &t);
      }
    }
}

}  // namespace

extern "C" {

int hp_pose_expand(const float* pose, float* proj, int NP) {
  for (int i = 0; i < NP; ++i) {
    float out[21];
    bbd_make_proj(pose + (size_t)i * BBD_POSE_STRIDE, out);
    for (int k = 0; k < 21; ++k) proj[(size_t)i * BBD_PROJ_STRIDE + k] = out[k];
    for (int k = 21; k < BBD_PROJ_STRIDE; ++k) proj[(size_t)i * BBD_PROJ_STRIDE + k] = 0.0f;
  }
  return 0;
}

int hp_identity_loss_fwd(const void* const* frames, const float* target, const int32_t* items, int NI,
                         float* ident, int H, int W, int no_ssim) {
  const size_t img = (size_t)3 * H * W;
  for (int i = 0; i < NI; ++i) {
    const float* tg = target + (size_t)items[i * 4] * img;
    const float* src = static_cast<const float*>(frames[items[i * 4 + 1]]) + (size_t)items[i * 4 + 2] * img;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) ident[((size_t)i * H + y) * W + x] = photometric(src, tg, y, x, H, W, no_ssim);
  }
  return 0;
}

int hp_identity_loss_grouped_fwd(const void* const* frames, const float* target, const int32_t* items,
                                 const int32_t* group_off, int G, float* ident, int H, int W, int no_ssim) {
  return hp_identity_loss_fwd(frames, target, items, group_off[G], ident, H, W, no_ssim);
}

int hp_warp_ssim_min_fwd(const void* const* frames, const float* target, const float* depth, const float* proj,
                         const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                         float* min_loss, uint8_t* argmin, float* partial, float* warped, int S, int B, int NP,
                         int H, int W, int no_ssim) {
  const size_t hw = (size_t)H * W, img = 3 * hw;
  std::vector<float> wbuf(img);
  for (int s = 0; s < S; ++s)
    for (int b = 0; b < B; ++b) {
      const size_t sb = (size_t)s * B + b;
      float* best = min_loss + sb * hw;
      uint8_t* arg = argmin + sb * hw;
      std::vector<int> argi(hw, 0);
      for (size_t p = 0; p < hw; ++p) best[p] = INFINITY;
      for (int c = 0; c < ncand[b]; ++c) {
        const bbd_cand_t cd = cand[b * BBD_MAX_CAND + c];
        if ((cd.kind & 0xff) == BBD_KIND_WARP) {
          const float* src = static_cast<const float*>(frames[cd.slot]) + (size_t)cd.row * img;
          warp_image(src, depth + sb * hw, proj + (size_t)cd.pose * BBD_PROJ_STRIDE, H, W, wbuf.data());
          if (warped) memcpy(warped + ((size_t)s * NP + cd.pose) * img, wbuf.data(), img * sizeof(float));
          for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
              const float l = photometric(wbuf.data(), target + (size_t)b * img, y, x, H, W, no_ssim);
              bbd_min_update(l, c, &best[(size_t)y * W + x], &argi[(size_t)y * W + x]);
            }
        } else {
          for (size_t p = 0; p < hw; ++p) {
            const float l = ident[(size_t)cd.row * hw + p] + (noise ? noise[(size_t)b * hw + p] : 0.0f);
            bbd_min_update(l, c, &best[p], &argi[p]);
          }
        }
      }
      double tot = 0;
      for (size_t p = 0; p < hw; ++p) { arg[p] = (uint8_t)argi[p]; tot += best[p]; }
      if (partial) partial[sb] = (float)tot;   // host port: one "tile" per (scale, sample)
    }
  return 0;
}

// grad_proj here is [S,NP,12] (already reduced over the image).
int hp_warp_ssim_min_bwd(const void* const* frames, const float* target, const float* depth, const float* proj,
                         const bbd_cand_t* cand, const int32_t* ncand, const uint8_t* argmin, const float* gscale,
                         float* grad_depth, float* grad_proj, int S, int B, int NP, int H, int W, int no_ssim) {
  const size_t hw = (size_t)H * W, img = 3 * hw;
  std::vector<float> wbuf(img), gx(img);
  memset(grad_depth, 0, sizeof(float) * S * B * hw);
  memset(grad_proj, 0, sizeof(float) * S * NP * 12);
  for (int s = 0; s < S; ++s)
    for (int b = 0; b < B; ++b) {
      const size_t sb = (size_t)s * B + b;
      const float g = gscale[s];
      const float w_ssim = no_ssim ? 0.0f : g * 0.85f / 3.0f;
      const float w_l1 = no_ssim ? g / 3.0f : g * 0.15f / 3.0f;
      const float* tg = target + (size_t)b * img;
      const float* dep = depth + sb * hw;
      for (int c = 0; c < ncand[b]; ++c) {
        const bbd_cand_t cd = cand[b * BBD_MAX_CAND + c];
        if ((cd.kind & 0xff) != BBD_KIND_WARP) continue;
        const float* src = static_cast<const float*>(frames[cd.slot]) + (size_t)cd.row * img;
        const float* pj = proj + (size_t)cd.pose * BBD_PROJ_STRIDE;
        warp_image(src, dep, pj, H, W, wbuf.data());
        std::fill(gx.begin(), gx.end(), 0.0f);
        // scatter d loss_p / d warped texels for every pixel p this candidate won
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            if (argmin[sb * hw + (size_t)y * W + x] != c) continue;
            for (int ch = 0; ch < 3; ++ch) {
              const float* wx = wbuf.data();
              if (!no_ssim) {
                float sx = 0, sxx = 0, sxy = 0, sy = 0, syy = 0;
                for (int dr = -1; dr <= 1; ++dr)
                  for (int dc = -1; dc <= 1; ++dc) {
                    const float xv = texel(wx, ch, y + dr, x + dc, H, W), yv = texel(tg, ch, y + dr, x + dc, H, W);
                    sx += xv; sxx += xv * xv; sxy += xv * yv; sy += yv; syy += yv * yv;
                  }
                float mu_y, sg_y, A, Bc, Cc;
                bbd_ystats(sy, syy, &mu_y, &sg_y);
                bbd_ssim_grad(sx, sxx, sxy, mu_y, sg_y, &A, &Bc, &Cc);
                for (int dr = -1; dr <= 1; ++dr)
                  for (int dc = -1; dc <= 1; ++dc) {
                    const int ry = bbd_reflect(y + dr, H), rx = bbd_reflect(x + dc, W);
                    const size_t qi = ((size_t)ch * H + ry) * W + rx;
                    gx[qi] += w_ssim * (A + Bc * wx[qi] + Cc * tg[qi]) * (1.0f / 9.0f);
                  }
              }
              const size_t qi = ((size_t)ch * H + y) * W + x;
              const float df = wx[qi] - tg[qi];
              gx[qi] += w_l1 * (df > 0 ? 1.0f : (df < 0 ? -1.0f : 0.0f));
            }
          }
        // chain to depth and P
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            BbdSample sm;
            const BbdDims dm = bbd_dims(H, W);
            bbd_project(pj, x, y, dep[(size_t)y * W + x], dm, &sm);
            BbdTaps t;
            bbd_taps(sm.ix, sm.iy, dm, &t);
            float gix = 0, giy = 0;
            for (int ch = 0; ch < 3; ++ch) {
              float v[4];
              bbd_fetch4(src + (size_t)ch * hw, &t, v);
              bbd_bilerp_grad(v, &t, gx[((size_t)ch * H + y) * W + x], &gix, &giy);
            }
            float gd, gp[12];
            bbd_project_grad(pj, &sm, gix, giy, &gd, gp);
            grad_depth[sb * hw + (size_t)y * W + x] += gd;
            if (!(cd.kind & BBD_FLAG_NO_POSE_GRAD))
              for (int k = 0; k < 12; ++k) grad_proj[((size_t)s * NP + cd.pose) * 12 + k] += gp[k];
          }
      }
    }
  return 0;
}

int hp_disp_to_depth_fwd(const float* disp, float* depth, int B, int h, int w, int H, int W, double min_depth,
                         double max_depth) {
  const float lo = (float)(1.0 / max_depth), span = (float)(1.0 / min_depth - 1.0 / max_depth);
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
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
        depth[((size_t)b * H + y) * W + x] = 1.0f / (lo + span * v);
      }
  return 0;
}

// scatter-form adjoint (the kernel uses a gather)
int hp_disp_to_depth_bwd(const float* disp, const float* /*depth*/, const float* grad_depth, float* grad_disp, int B,
                         int h, int w, int H, int W, double min_depth, double max_depth) {
  const float lo = (float)(1.0 / max_depth), span = (float)(1.0 / min_depth - 1.0 / max_depth);
  memset(grad_disp, 0, sizeof(float) * B * h * w);
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const float* d = disp + (size_t)b * h * w;
        float* gd = grad_disp + (size_t)b * h * w;
        int y0 = y, y1 = y, x0 = x, x1 = x;
        float ly0 = 1, ly1 = 0, lx0 = 1, lx1 = 0;
        if (!(h == H && w == W)) {
          bbd_up_src(y, h, H, &y0, &y1, &ly0, &ly1);
          bbd_up_src(x, w, W, &x0, &x1, &lx0, &lx1);
        }
        const float sc = lo + span * bbd_up_blend(d[(size_t)y0 * w + x0], d[(size_t)y0 * w + x1],
                                                  d[(size_t)y1 * w + x0], d[(size_t)y1 * w + x1], ly0, ly1, lx0, lx1, (H + W) <= 128);
        const float gv = grad_depth[((size_t)b * H + y) * W + x] * (-span / (sc * sc));
        gd[(size_t)y0 * w + x0] += gv * ly0 * lx0;
        gd[(size_t)y0 * w + x1] += gv * ly0 * lx1;
        gd[(size_t)y1 * w + x0] += gv * ly1 * lx0;
        gd[(size_t)y1 * w + x1] += gv * ly1 * lx1;
      }
  return 0;
}

// ---- disparity-mode forms: composed from the pieces above -------------------------------
static void hp_depth_planes(const void* const* disp, const int32_t* disp_hw, double min_depth, double max_depth, int S,
                            int B, int H, int W, float* depth) {
  for (int s = 0; s < S; ++s)
    hp_disp_to_depth_fwd(static_cast<const float*>(disp[s]), depth + (size_t)s * B * H * W, B, disp_hw[2 * s],
                         disp_hw[2 * s + 1], H, W, min_depth, max_depth);
}

int hp_warp_ssim_min_disp_fwd(const void* const* frames, const float* target, const void* const* disp,
                              const int32_t* disp_hw, double min_depth, double max_depth, const float* proj,
                              const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                              const int32_t* /*work_items: a launch-order choice of the GPU kernels*/,
                              float* min_loss, uint8_t* argmin, float* partial, float* warped, float* depth_out,
                              int S, int B, int NP, int H, int W, int no_ssim) {
  std::vector<float> tmp;
  float* depth = depth_out;
  if (!depth) { tmp.resize((size_t)S * B * H * W); depth = tmp.data(); }
  hp_depth_planes(disp, disp_hw, min_depth, max_depth, S, B, H, W, depth);
  return hp_warp_ssim_min_fwd(frames, target, depth, proj, ident, noise, cand, ncand, min_loss, argmin, partial, warped,
                              S, B, NP, H, W, no_ssim);
}

int hp_warp_ssim_min_disp_bwd(const void* const* frames, const float* target, const void* const* disp,
                              const int32_t* disp_hw, double min_depth, double max_depth, const float* /*depth planes*/,
                              const float* proj,
                              const bbd_cand_t* cand, const int32_t* ncand, const int32_t* /*work_items*/,
                              const uint8_t* argmin, const float* gscale,
                              float* grad_up, float* grad_proj, int S, int B, int NP, int H, int W, int no_ssim) {
  std::vector<float> depth((size_t)S * B * H * W);
  hp_depth_planes(disp, disp_hw, min_depth, max_depth, S, B, H, W, depth.data());
  const int rc = hp_warp_ssim_min_bwd(frames, target, depth.data(), proj, cand, ncand, argmin, gscale, grad_up, grad_proj,
                                      S, B, NP, H, W, no_ssim);
  const float span = (float)(1.0 / min_depth - 1.0 / max_depth);
  for (size_t i = 0; i < depth.size(); ++i) grad_up[i] *= -span * depth[i] * depth[i];
  return rc;
}

int hp_disp_upsample_adjoint(const void* const* grad_up, const int32_t* disp_hw, void* const* grad_disp, int n, int B, int H,
                             int W) {
  for (int k = 0; k < n; ++k) {
    const int h = disp_hw[2 * k], w = disp_hw[2 * k + 1];
    const float* g = static_cast<const float*>(grad_up[k]);
    float* gd = static_cast<float*>(grad_disp[k]);
    memset(gd, 0, sizeof(float) * B * h * w);
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          int y0 = y, y1 = y, x0 = x, x1 = x;
          float ly0 = 1, ly1 = 0, lx0 = 1, lx1 = 0;
          if (!(h == H && w == W)) {
            bbd_up_src(y, h, H, &y0, &y1, &ly0, &ly1);
            bbd_up_src(x, w, W, &x0, &x1, &lx0, &lx1);
          }
          const float gv = g[((size_t)b * H + y) * W + x];
          float* o = gd + (size_t)b * h * w;
          o[(size_t)y0 * w + x0] += gv * ly0 * lx0;
          o[(size_t)y0 * w + x1] += gv * ly0 * lx1;
          o[(size_t)y1 * w + x0] += gv * ly1 * lx0;
          o[(size_t)y1 * w + x1] += gv * ly1 * lx1;
        }
  }
  return 0;
}

// smoothness (plain loops; single "chunk" layout: sums [B,1,2], dots [B,1])
static float hp_edge(const float* im, int hw, int i0, int i1) {
  const float g = fabsf(im[i0] - im[i1]) + fabsf(im[i0 + hw] - im[i1 + hw]) + fabsf(im[i0 + 2 * hw] - im[i1 + 2 * hw]);
  return expf(-g * (1.0f / 3.0f));
}
static float hp_sgn(float v) { return v > 0 ? 1.0f : (v < 0 ? -1.0f : 0.0f); }

int hp_smooth_loss_fwd(const float* disp, const float* img, float* mean_disp, float* sums, int B, int h, int w) {
  const int hw = h * w;
  for (int b = 0; b < B; ++b) {
    const float* d = disp + (size_t)b * hw;
    const float* im = img + (size_t)b * 3 * hw;
    double m = 0;
    for (int i = 0; i < hw; ++i) m += d[i];
    mean_disp[b] = (float)m;                       // partial-sum layout of the ABI (one chunk)
    const float inv = 1.0f / (mean_disp[b] / (float)hw + 1e-7f);
    double ax = 0, ay = 0;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int i = y * w + x;
        if (x < w - 1) ax += fabsf(d[i] * inv - d[i + 1] * inv) * hp_edge(im, hw, i, i + 1);
        if (y < h - 1) ay += fabsf(d[i] * inv - d[i + w] * inv) * hp_edge(im, hw, i, i + w);
      }
    sums[b * 2 + 0] = (float)ax;
    sums[b * 2 + 1] = (float)ay;
  }
  return 0;
}

int hp_smooth_loss_bwd(const float* disp, const float* img, const float* mean_disp, const float* gscale,
                       float* grad, float* dots, int B, int h, int w) {
  const int hw = h * w;
  const float gxs = gscale[0] / ((float)B * h * (w - 1)), gys = gscale[0] / ((float)B * (h - 1) * w);
  for (int b = 0; b < B; ++b) {
    const float* d = disp + (size_t)b * hw;
    const float* im = img + (size_t)b * 3 * hw;
    float* go = grad + (size_t)b * hw;
    const float inv = 1.0f / (mean_disp[b] / (float)hw + 1e-7f);
    double dot = 0;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int i = y * w + x;
        const float n0 = d[i] * inv;
        float gn = 0;
        if (x < w - 1) gn += gxs * hp_sgn(n0 - d[i + 1] * inv) * hp_edge(im, hw, i, i + 1);
        if (x > 0) gn -= gxs * hp_sgn(d[i - 1] * inv - n0) * hp_edge(im, hw, i - 1, i);
        if (y < h - 1) gn += gys * hp_sgn(n0 - d[i + w] * inv) * hp_edge(im, hw, i, i + w);
        if (y > 0) gn -= gys * hp_sgn(d[i - w] * inv - n0) * hp_edge(im, hw, i - w, i);
        go[i] = gn;
        dot += (double)gn * d[i];
      }
    dots[b] = (float)dot;
    const float sub = (float)dot * inv * inv / (float)hw;
    for (int i = 0; i < hw; ++i) go[i] = go[i] * inv - sub;
  }
  return 0;
}

// every scale of a step (bbd_smooth_loss_multi_*): the single-scale port per scale
int hp_smooth_loss_multi_fwd(const void* const* disp, const void* const* img, const int32_t* hw, float* mean_disp, float* sums,
                             int S, int B) {
  for (int s = 0; s < S; ++s) {
    const int rc = hp_smooth_loss_fwd(static_cast<const float*>(disp[s]), static_cast<const float*>(img[s]),
                                      mean_disp + (size_t)s * B, sums + (size_t)s * B * 2, B, hw[2 * s], hw[2 * s + 1]);
    if (rc) return rc;
  }
  return 0;
}
int hp_smooth_loss_multi_bwd(const void* const* disp, const void* const* img, const int32_t* hw, const float* mean_disp,
                             const float* gscale, void* const* grad_disp, float* dots, int S, int B) {
  for (int s = 0; s < S; ++s) {
    const int rc = hp_smooth_loss_bwd(static_cast<const float*>(disp[s]), static_cast<const float*>(img[s]),
                                      mean_disp + (size_t)s * B, gscale + s, static_cast<float*>(grad_disp[s]),
                                      dots + (size_t)s * B, B, hw[2 * s], hw[2 * s + 1]);
    if (rc) return rc;
  }
  return 0;
}

// exhaustive-ish check helpers for the constant divisions
int hp_check_div(uint32_t start, uint32_t count, uint32_t stride) {
  int bad = 0;
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t bits = start + k * stride;
    float x;
    memcpy(&x, &bits, 4);
    if (!(x == x) || fabsf(x) > 1e30f || (fabsf(x) < 1e-30f && x != 0.0f)) continue;
    if (bbd_div9(x) != x / 9.0f) ++bad;
    if (bbd_div3(x) != x / 3.0f) ++bad;
  }
  return bad;
}

}  // extern "C"
