// bbd_kernels.hip - hand-written gfx950 (CDNA4, wave64) kernels of the photometric
// reprojection hot path and their C-ABI launchers (include/bbd_hip.h).
//
// Layout / tiling (DESIGN.md "Kernels"):
//   * one workgroup = 256 threads = 4 waves = one 64x16 pixel tile of one (scale, sample);
//   * each thread owns a horizontal strip of 4 pixels (16-byte global stores, 16-byte LDS
//     window reads); 16 lanes cover a 256-byte tile row, 4 rows per wave;
//   * target and warped tiles (+ reflected halo) are staged in LDS as planar fp32.  Row strides are
//     = 4 (mod 64) floats and the strip a lane owns is ROTATED by its row index, which makes every
//     16-byte window read land on 64 distinct banks inside each ds_read_b128 lane group
//     ({0-3,12-15,20-27} ...: MI355X_MICROARCH.md, LDS) - conflict-free without padding to 128;
//   * the candidate loop is block-uniform (candidates are per sample), so the candidate
//     descriptor and the 3x4 projection sit in SGPRs;
//   * the 1-D grid is ordered sample-major, then scale, then tile: the four scales of a
//     sample re-read the same source/target images while they are L2 / Infinity-Cache hot;
//   * all index arithmetic is 32-bit against uniform (SGPR) base pointers.
//
// Arithmetic lives in bbd_math.h and is shared with the host port used by the CPU tests.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/bbd_hip.h"
#include "bbd_math.h"

#ifndef BBD_WARP_BATCH
#define BBD_WARP_BATCH 3
#endif
#ifndef BBD_PAIR_PROJECT_FWD
#define BBD_PAIR_PROJECT_FWD 0
#endif
#ifndef BBD_PAIR_PROJECT_BWD
#define BBD_PAIR_PROJECT_BWD 1
#endif
#ifndef BBD_BWD_WARP_BATCH
#define BBD_BWD_WARP_BATCH 2
#endif

namespace {

constexpr int TW = 64;   // tile width  (pixels)
constexpr int TH = 16;   // tile height (pixels)
constexpr int NT = 256;  // threads per workgroup
constexpr int PPT = 4;   // pixels per thread (horizontal strip)
constexpr int SPR = TW / PPT;  // strips per tile row = 16

// forward staging: (TH+2) x (TW+2) cells, row stride LS (windows read 8 floats from a strip start)
constexpr int LS = TW + 4;          // 68
constexpr int LH = TH + 2;
constexpr int LW = TW + 2;
constexpr int FPLANE = LH * LS;

// backward staging: x/y region (TH+4) x (TW+4) cells, coefficient region (TH+2) x (TW+2) cells
constexpr int BS = TW + 4;          // 68: the last strip's second 16-byte read runs 4 floats into
                                    // the next row (or the slack below) - values never used
constexpr int BH = TH + 4;
constexpr int BW = TW + 4;
constexpr int BPLANE = BH * BS;
constexpr int CS = TW + 4;          // 68
constexpr int CH = TH + 2;
constexpr int CW = TW + 2;
constexpr int CPLANE = CH * CS;
constexpr int CSTRIPS = (CW + PPT - 1) / PPT;   // 17 strips of 4 loss pixels per coefficient row

struct FramePtrs {
  const float* base[BBD_MAX_FRAME_SLOTS];
};

constexpr int KIND_MASK = 0xff;
constexpr int FLAG_NO_POSE_GRAD = 0x100;

// ---- wave64 reductions on the DPP crossbar (no LDS traffic); result valid in lane 63 ----------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_sum63(float v) {
  float a = v + dpp0<0x111, 0xf, 0xf>(v);   // row_shr:1
  a += dpp0<0x112, 0xf, 0xf>(v);            // row_shr:2
  a += dpp0<0x113, 0xf, 0xf>(v);            // row_shr:3
  a += dpp0<0x114, 0xf, 0xe>(a);            // row_shr:4, banks 1-3
  a += dpp0<0x118, 0xf, 0xc>(a);            // row_shr:8, banks 2-3
  a += dpp0<0x142, 0xa, 0xf>(a);            // row_bcast:15 into rows 1,3
  a += dpp0<0x143, 0xc, 0xf>(a);            // row_bcast:31 into rows 2,3
  return a;
}

// bitwise OR over the wave on the same crossbar steps; result valid in lane 63
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ unsigned dpp0u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ unsigned wave_or63(unsigned v) {
  unsigned a = v | dpp0u<0x111, 0xf, 0xf>(v);
  a |= dpp0u<0x112, 0xf, 0xf>(v);
  a |= dpp0u<0x113, 0xf, 0xf>(v);
  a |= dpp0u<0x114, 0xf, 0xe>(a);
  a |= dpp0u<0x118, 0xf, 0xc>(a);
  a |= dpp0u<0x142, 0xa, 0xf>(a);
  a |= dpp0u<0x143, 0xc, 0xf>(a);
  return a;
}

// Twelve wave totals at once (the 3x4 pose-gradient of a candidate): a reduce-scatter instead of twelve full
// reductions.  v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even 16-lane rows of TWO registers,
// so one swap + one add folds a pair of values through a butterfly level and leaves each half (row) holding a
// different value: 12 values -> 6 -> 3 registers whose four rows hold four different values summed over the rows,
// then the 16-lane row sums on the DPP crossbar.  ~50 instructions instead of ~170; fixed order => deterministic.
// Result: out[k] in lane 16*r + 15 = total of value 4*k + {0, 2, 1, 3}[r].
__device__ __forceinline__ float swap_add32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void wave_sum12(const float v[12], float out[3]) {
  float w[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) w[i] = swap_add32(v[2 * i], v[2 * i + 1]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float u = swap_add16(w[2 * k], w[2 * k + 1]);
    float a = u + dpp0<0x111, 0xf, 0xf>(u);   // row_shr:1
    a += dpp0<0x112, 0xf, 0xf>(u);            // row_shr:2
    a += dpp0<0x113, 0xf, 0xf>(u);            // row_shr:3
    a += dpp0<0x114, 0xf, 0xe>(a);            // row_shr:4, banks 1-3
    a += dpp0<0x118, 0xf, 0xc>(a);            // row_shr:8, banks 2-3
    out[k] = a;
  }
}
// which of the 12 values lane `lane` (one of 15, 31, 47, 63) holds in out[k]
__device__ __forceinline__ int wave_sum12_index(int k, int lane) {
  const int r = lane >> 4;
  return 4 * k + ((r & 1) << 1) + (r >> 1);
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

// XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, Workgroup dispatch):
// blocks b and b + 8 share an XCD and its 4 MiB L2.  The work items of these kernels are ordered sample-major -> scale
// -> tile, and neighbouring tiles (and the scales of a sample) gather from the same source-image rows, so the items are
// re-dealt such that each XCD walks ONE contiguous range of them: its L2 then serves the halo / parallax overlap of
// neighbouring tiles instead of the Infinity Cache.  Bijective for every grid size (cdna_hip_programming.md T1); a pure
// speed choice - any placement computes the same result.
__device__ __forceinline__ int xcd_work_item(int bid, int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Slab order of the fused launches (round 4): a host-built table `work[blockIdx.x]` = (sample, scale, tile, tile origin),
// bbd_fused_work_items below.  The image's tiles are cut into 8 slabs of consecutive tiles, one per XCD, and XCD x walks:
// for every target sample (in the caller's order: most candidates first) - for every scale - the tiles of slab x.  So
// (1) all scales of a sample's slab run on ONE XCD back to back and share its L2 (they read the same source / target
// region; the plain range split of xcd_work_item could not be used with several scales, it gave whole scales of
// different cost to different XCDs), (2) every XCD gets the same work whatever the samples cost, and (3) inside an XCD
// the expensive samples start first, so the launch's tail is made of cheap workgroups (boosted batches mix 8-, 14- and
// 18-candidate samples: mono_dataset.py:87-109).  A table instead of arithmetic: decoding a block index takes 3-5
// scalar integer divisions (v_rcp + v_readfirstlane round trips, serial, in front of the first load of a workgroup that
// lives ~50 k cycles) - measured 2.7 % of the MD2 backward (profiles/r04/work_order_ab.txt); the table is one s_load_dwordx2.
struct WorkItem {
  int b, s, tile, tx0, ty0;
};
constexpr int WORK_B_BITS = 12, WORK_S_BITS = 3;       // word 0 = b | s << 12 | tile << 15; word 1 = tx0 | ty0 << 16
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p);
__device__ __forceinline__ WorkItem load_work_item(const int32_t* work) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  const v2i w = uniform_load(reinterpret_cast<const v2i*>(work) + blockIdx.x);
  WorkItem it;
  it.b = w.x & ((1 << WORK_B_BITS) - 1);
  it.s = (w.x >> WORK_B_BITS) & ((1 << WORK_S_BITS) - 1);
  it.tile = (int)((unsigned)w.x >> (WORK_B_BITS + WORK_S_BITS));
  it.tx0 = w.y & 0xffff;
  it.ty0 = (int)((unsigned)w.y >> 16);
  return it;
}

// Strip owned by a thread: row ly, first tile-local column lx0 (rotated by the row, see header).
__device__ __forceinline__ void strip_of_thread(int* ly, int* lx0) {
  const int r = threadIdx.x / SPR;
  *ly = r;
  *lx0 = (((int)threadIdx.x - r) & (SPR - 1)) * PPT;
}

// ---- staging cells --------------------------------------------------------------------------
// A staged region of ROWS x COLS cells (cell (0,0) = image pixel (ty0-HALO, tx0-HALO), reflected at
// the image border) is filled by the 256 threads, N = ceil(ROWS*COLS/256) cells per thread.  The
// cell -> thread assignment is fixed for the life of the workgroup, so each thread derives its
// cells ONCE (no index math in the candidate loop) and keeps their depth in registers.  Threads
// past the end of the region duplicate the last cell (same value to the same LDS word), which
// keeps the loops branch-free so that all loads of a phase are in flight together.
template <int ROWS, int COLS, int STRIDE, int HALO, int NTH = NT>
struct Cells {
  static constexpr int N = (ROWS * COLS + NTH - 1) / NTH;
  int lds[N];    // r * STRIDE + c
  int xy[N];     // yy << 16 | xx of the (reflected) image pixel
  int ownbits;   // bit k: cell k is an un-reflected interior pixel this tile owns (warped / depth output,
                 // sample-derivative planes).  Two words per cell live across the candidate loop, not four:
                 // the pixel offset and the own-pixel index are re-derived where they are needed (rarely).

  __device__ __forceinline__ void init(int H, int W, int tx0, int ty0) {
    ownbits = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      int i = k * NTH + (int)threadIdx.x;
      i = i < ROWS * COLS ? i : ROWS * COLS - 1;
      const int r = i / COLS, c = i - r * COLS;
      const int py = ty0 + r - HALO, px = tx0 + c - HALO;
      const int yy = bbd_reflect(py, H), xx = bbd_reflect(px, W);
      lds[k] = r * STRIDE + c;
      xy[k] = (yy << 16) | xx;
      if (py == yy && px == xx && r >= HALO && r < ROWS - HALO && c >= HALO && c < COLS - HALO) ownbits |= 1 << k;
    }
  }
  __device__ __forceinline__ int pix(int k, int W) const { return (xy[k] >> 16) * W + (xy[k] & 0xffff); }
  __device__ __forceinline__ bool own(int k) const { return (ownbits >> k) & 1; }
};

// Block-uniform table reads (candidate descriptors, pose-table rows, per-scale scalars): read-only for the whole
// launch, so they are loaded through the constant address space.  hipcc then selects scalar loads (s_load_dword*,
// values in SGPRs); through a plain pointer the same reads that follow a barrier or a store became per-lane
// global_load + v_readfirstlane with s_waitcnt vmcnt(0) - three dependent vector-memory round trips per candidate
// and the 21 pose values in VGPRs (profiles/r02/phase_stamps_final.txt: 6.3k + 4.3k of 64k ticks per workgroup).
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p) {
  typedef const __attribute__((address_space(4))) T* const_ptr;
  return *(const_ptr)p;
}
__device__ __forceinline__ bbd_cand_t load_cand(const bbd_cand_t* p) {     // one s_load_dwordx4
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i w = uniform_load(reinterpret_cast<const v4i*>(p));
  bbd_cand_t c;
  c.kind = w.x; c.slot = w.y; c.row = w.z; c.pose = w.w;
  return c;
}

template <typename CellsT, int PLANE>
__device__ __forceinline__ void stage_image(const float* __restrict__ img, int hw, int hw_w, const CellsT& cl,
                                            float (*s)[PLANE]) {
  float v[CellsT::N][3];
#pragma unroll
  for (int k = 0; k < CellsT::N; ++k) {
    const int px = cl.pix(k, hw_w);
    v[k][0] = img[px];
    v[k][1] = img[px + hw];
    v[k][2] = img[px + 2 * hw];
  }
#pragma unroll
  for (int k = 0; k < CellsT::N; ++k) {
    s[0][cl.lds[k]] = v[k][0];
    s[1][cl.lds[k]] = v[k][1];
    s[2][cl.lds[k]] = v[k][2];
  }
}

// Where a (scale, sample)'s depth comes from: a full-resolution depth plane (bbd_disp_to_depth_fwd's output), or
// - SURVEY 8f-1 - the decoder's low-resolution disparity itself: F.interpolate(bilinear, align_corners=False)
// + layers.disp_to_depth (trainer.py:455-461, layers.py:13-22) evaluated per staged cell with the same
// functions (bbd_up_src / bbd_up_blend, IEEE reciprocal) as the stand-alone kernel, so depth is bit-identical.
constexpr int MAX_SCALES = 4;
struct DispSrc {
  const float* disp[MAX_SCALES];   // per scale [B,1,h,w]; all NULL = depth-plane mode
  int h[MAX_SCALES], w[MAX_SCALES];
  float lo, span;                  // 1/max_depth, 1/min_depth - 1/max_depth
  int grad_wrt_disp;               // backward hands back d loss / d (up-sampled disparity) instead of d loss / d depth
};
struct DepthSrc {
  const float* depth;              // [H,W] plane of this (scale, sample), or nullptr
  const float* disp;               // [h,w] plane of this (scale, sample)
  int h, w, small;
  float lo, span;
};
__device__ __forceinline__ DepthSrc depth_source(const float* depth_planes, const DispSrc& ds, int s, int b, size_t sb,
                                                 int H, int W) {
  DepthSrc d;
  d.depth = depth_planes ? depth_planes + sb * (size_t)H * W : nullptr;
  d.h = ds.h[s]; d.w = ds.w[s];
  d.disp = depth_planes ? nullptr : ds.disp[s] + (size_t)b * d.h * d.w;
  d.small = (H + W) <= 128;
  d.lo = ds.lo; d.span = ds.span;
  return d;
}
// Depth of N pixels in two halves.  issue() starts every global load with no branch between them, so they travel together
// and the caller's other set-up work runs under them; finish() turns the loaded values into depth.  (The earlier
// per-pixel form chose between "depth plane / same-size disparity / 2x2 up-sampling" with a branch per pixel: each branch
// region ended in s_waitcnt vmcnt(0), i.e. one full memory round trip per staged cell, five in a row in each kernel's
// set-up - profiles/r04/phase_stamps_*.txt.)  PLANE = read a full-resolution depth plane; otherwise the decoder's
// disparity map through the four-tap form: for a same-size map the taps' lambdas are exactly (1, 0) and the value is
// the first tap itself (selected, not blended, so that it stays an exact copy).
template <int N, bool PLANE>
struct DepthFetch {
  static constexpr int TAPS = PLANE ? 1 : 4, NL = PLANE ? 1 : N;
  float v[N][TAPS];
  float ly1[NL], lx1[NL];
  __device__ __forceinline__ void issue(const DepthSrc& d, int k, int yy, int xx, int H, int W) {
    if constexpr (PLANE) {
      v[k][0] = d.depth[yy * W + xx];
    } else {
      int y0, y1, x0, x1;
      float ly0, lx0;
      bbd_up_src(yy, d.h, H, &y0, &y1, &ly0, &ly1[k]);
      bbd_up_src(xx, d.w, W, &x0, &x1, &lx0, &lx1[k]);
      v[k][0] = d.disp[y0 * d.w + x0];
      v[k][1] = d.disp[y0 * d.w + x1];
      v[k][2] = d.disp[y1 * d.w + x0];
      v[k][3] = d.disp[y1 * d.w + x1];
    }
  }
  __device__ __forceinline__ float finish(const DepthSrc& d, int k, int H, int W) const {
    if constexpr (PLANE) {
      return v[k][0];
    } else {
      float val = v[k][0];
      if (!(d.h == H && d.w == W))      // (bbd_up_src: l0 = 1 - l1)
        val = bbd_up_blend(v[k][0], v[k][1], v[k][2], v[k][3], 1.0f - ly1[k], ly1[k], 1.0f - lx1[k], lx1[k], d.small);
      return 1.0f / (d.lo + d.span * val);
    }
  }
};

// Warp one source image into the staged region for one pose-table row (after the packed helpers below).
// ---- packed fp32 (two values per lane) ------------------------------------------------------------------------
// Plain fp32 vector instructions issue once per ~4 cycles per SIMD on gfx950 at any occupancy; v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 carry two fp32 values per lane at the same rate (profiles/r03/valu_rate.txt).  Each
// component is an IEEE fp32 operation, so a packed evaluation of the SAME operation sequence gives the same bits.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f pk1(float a) { v2f r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// bbd_project / bbd_project_bwd for TWO staged cells at once (components .x / .y), operation for operation: only what the
// warp phases consume comes out (clamped coordinates + clamp flags).  GUARDED = the forward's form (a component whose
// operands leave the exponent window of the refined-reciprocal divisions is redone with the scalar function).
template <bool GUARDED>
__device__ __forceinline__ void project_pair(const float* pj, int xy0, int xy1, float d0, float d1, const BbdDims& dm,
                                             BbdSample* s0, BbdSample* s1) {
#if defined(__HIP_DEVICE_COMPILE__)
  const v2f fx = pk2((float)(xy0 & 0xffff), (float)(xy1 & 0xffff)), fy = pk2((float)(xy0 >> 16), (float)(xy1 >> 16));
  const v2f dep = pk2(d0, d1), one = pk1(1.0f);
  const float* iK = pj + 12;
  // bbd_dot3_hom: fma(a2, 1, fma(a1, y, a0 * x))
  const v2f cx = pk_fma(pk1(iK[2]), one, pk_fma(pk1(iK[1]), fy, pk1(iK[0]) * fx));
  const v2f cy = pk_fma(pk1(iK[5]), one, pk_fma(pk1(iK[4]), fy, pk1(iK[3]) * fx));
  const v2f cz = pk_fma(pk1(iK[8]), one, pk_fma(pk1(iK[7]), fy, pk1(iK[6]) * fx));
  const v2f X = dep * cx, Y = dep * cy, Z = dep * cz;
  // bbd_dot4_hom: fma(a3, 1, fma(a2, z, fma(a1, y, a0 * x)))
  const v2f qx = pk_fma(pk1(pj[3]), one, pk_fma(pk1(pj[2]), Z, pk_fma(pk1(pj[1]), Y, pk1(pj[0]) * X)));
  const v2f qy = pk_fma(pk1(pj[7]), one, pk_fma(pk1(pj[6]), Z, pk_fma(pk1(pj[5]), Y, pk1(pj[4]) * X)));
  const v2f qz = pk_fma(pk1(pj[11]), one, pk_fma(pk1(pj[10]), Z, pk_fma(pk1(pj[9]), Y, pk1(pj[8]) * X)));
  const v2f zi = qz + pk1(BBD_EPS);
  v2f r0;
  r0.x = __builtin_amdgcn_rcpf(zi.x);
  r0.y = __builtin_amdgcn_rcpf(zi.y);
  const v2f r = pk_fma(pk_fma(-zi, r0, one), r0, r0);                    // bbd_rcp_refined
  v2f u = qx * r, v = qy * r;                                            // bbd_div_with x 2
  u = pk_fma(pk_fma(-zi, u, qx), r, u);
  u = pk_fma(pk_fma(-zi, u, qx), r, u);
  v = pk_fma(pk_fma(-zi, v, qy), r, v);
  v = pk_fma(pk_fma(-zi, v, qy), r, v);
  const v2f wm1 = pk1(dm.wm1), hm1 = pk1(dm.hm1), rw = pk1(dm.rw), rh = pk1(dm.rh);
  v2f nu = u * rw, nv = v * rh;                                          // bbd_div_const (divisor known on the host)
  nu = pk_fma(pk_fma(-wm1, nu, u), rw, nu);
  nu = pk_fma(pk_fma(-wm1, nu, u), rw, nu);
  nv = pk_fma(pk_fma(-hm1, nv, v), rh, nv);
  nv = pk_fma(pk_fma(-hm1, nv, v), rh, nv);
  const v2f gx = pk_fma(nu, pk1(2.0f), pk1(-1.0f)), gy = pk_fma(nv, pk1(2.0f), pk1(-1.0f));     // bbd_norm_to_grid
  v2f ix = pk_fma(gx, pk1(0.5f), pk1(0.5f)) * wm1, iy = pk_fma(gy, pk1(0.5f), pk1(0.5f)) * hm1; // bbd_grid_to_unit
  BbdSample* so[2] = {s0, s1};
  const float ixs[2] = {ix.x, ix.y}, iys[2] = {iy.x, iy.y};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float fxi = ixs[c], fyi = iys[c];
    so[c]->clipx = !(fxi > 0.0f && fxi < dm.wm1);
    so[c]->clipy = !(fyi > 0.0f && fyi < dm.hm1);
    fxi = fxi > 0.0f ? fxi : 0.0f;
    fyi = fyi > 0.0f ? fyi : 0.0f;
    so[c]->ix = fxi < dm.wm1 ? fxi : dm.wm1;
    so[c]->iy = fyi < dm.hm1 ? fyi : dm.hm1;
  }
  if (GUARDED) {
    const float qxs[2] = {qx.x, qx.y}, qys[2] = {qy.x, qy.y}, zis[2] = {zi.x, zi.y}, us[2] = {u.x, u.y}, vs[2] = {v.x, v.y};
    const int xys[2] = {xy0, xy1};
    const float ds_[2] = {d0, d1};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ok = bbd_exp_ok3(qxs[c], qys[c], zis[c]) & (bbd_exp_ok(us[c]) | (us[c] == 0.0f)) &
                     (bbd_exp_ok(vs[c]) | (vs[c] == 0.0f));
      if (!ok) bbd_project(pj, xys[c] & 0xffff, xys[c] >> 16, ds_[c], dm, so[c]);
    }
  }
#else
  bbd_project(pj, xy0 & 0xffff, xy0 >> 16, d0, dm, s0);
  bbd_project(pj, xy1 & 0xffff, xy1 >> 16, d1, dm, s1);
#endif
}

struct NoWork {
  __device__ __forceinline__ void operator()() const {}
};
// XS = element stride of the staged planes (2: the backward interleaves (x, y) pairs, see its kernel).
// `under_gathers`: work of the caller that does not depend on the warp, run once after the first batch's gathers have been
// issued and before their values are consumed (the wave would otherwise only wait there).
// OWNREG > 0 (the backward): the first OWNREG cells of a thread are its own pixels and their d warped / d (ix, iy) stay in its
// registers (dvr[cell][0..2] = d/d ix per channel, [3..5] = d/d iy; zero where the border clamp is active), so the
// sample-gradient phase needs neither gathers nor LDS planes.
template <int BATCH, typename CellsT, int PLANE, bool BWD = false, int XS = 1, typename Work = NoWork, int OWNREG = 0>
__device__ __forceinline__ void warp_into_lds(const float* __restrict__ src, const float (&d)[CellsT::N],
                                              const float (&pj)[21], const BbdDims dm, int hw,
                                              const CellsT& cl, float (*s)[PLANE],
                                              float* __restrict__ warped_out, Work under_gathers = Work(),
                                              float (*dvr)[6] = nullptr) {
  // pj = P (3x4) | inv_K[:3,:3] of this candidate: block-uniform loads from the projection table by the caller, so the
  // 21 values live in SGPRs (bbd_pose_expand formed P once, with the reference's rounding order)
  // Cells are processed in batches: project + tap geometry for the whole batch first, then all of
  // its gathers are in flight together (6 x 8-byte loads per cell), then the blends.  The batch
  // size trades loads in flight against VGPRs (occupancy): measured best at 3 cells for the forward
  // (3 waves/SIMD) and 2 for the backward (2 waves/SIMD); BBD_WARP_BATCH / BBD_BWD_WARP_BATCH override.
#pragma unroll
  for (int k0 = 0; k0 < CellsT::N; k0 += BATCH) {
    BbdTaps t[BATCH];
    int clip[BATCH];
    BbdSample smp[BATCH];
    {
      // two cells per packed projection, a last odd one by the scalar function.  Measured (profiles/r03/pair_project_ab.txt):
      // pays in the backward (unguarded divisions); in the forward the per-component validity tests and the register
      // pairs cost more than the packing saves (+9 % in the training step) - scalar there.
      constexpr bool PAIRS = BWD ? (BBD_PAIR_PROJECT_BWD != 0) : (BBD_PAIR_PROJECT_FWD != 0);
      constexpr int NPAIR = PAIRS ? BATCH / 2 : 0;
#pragma unroll
      for (int kk = 0; kk < 2 * NPAIR; kk += 2) {
        const int ka = k0 + kk < CellsT::N ? k0 + kk : CellsT::N - 1, kb = k0 + kk + 1 < CellsT::N ? k0 + kk + 1 : CellsT::N - 1;
        project_pair<!BWD>(pj, cl.xy[ka], cl.xy[kb], d[ka], d[kb], dm, &smp[kk], &smp[kk + 1]);
      }
#pragma unroll
      for (int kk = 2 * NPAIR; kk < BATCH; ++kk) {
        const int k = k0 + kk < CellsT::N ? k0 + kk : CellsT::N - 1;
        if (BWD) bbd_project_bwd(pj, cl.xy[k] & 0xffff, cl.xy[k] >> 16, d[k], dm, &smp[kk]);
        else bbd_project(pj, cl.xy[k] & 0xffff, cl.xy[k] >> 16, d[k], dm, &smp[kk]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < BATCH; ++kk) {
      const int k = k0 + kk < CellsT::N ? k0 + kk : CellsT::N - 1;
      const BbdSample sm = smp[kk];
      (void)k;
      bbd_taps(sm.ix, sm.iy, dm, &t[kk]);
      clip[kk] = sm.clipx | (sm.clipy << 1);
    }
    float v[BATCH][3][4];
#pragma unroll
    for (int kk = 0; kk < BATCH; ++kk)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        bbd_fetch4(src + ch * hw, &t[kk], v[kk][ch]);
      }
    if (k0 == 0) under_gathers();
#pragma unroll
    for (int kk = 0; kk < BATCH; ++kk) {
      if (k0 + kk >= CellsT::N) break;
      const int k = k0 + kk;
      float val[3], dvx[3], dvy[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        if (BWD) {
          // the backward needs the forward's TEXELS (the coordinates above are its bits), not its blend rounding:
          // x = top + n (bot - top) with top / bot the two horizontal lerps shares every difference with
          // d x / d ix = d_top + n (d_bot - d_top) and d x / d iy = bot - top: 8 operations per channel instead of 20
          const float* vv = v[kk][ch];
          const float dt = vv[1] - vv[0], db = vv[3] - vv[2];
          const float top = fmaf(t[kk].w, dt, vv[0]), bot = fmaf(t[kk].w, db, vv[2]);
          const float bt = bot - top;
          val[ch] = fmaf(t[kk].n, bt, top);
          dvx[ch] = fmaf(t[kk].n, db - dt, dt);
          dvy[ch] = bt;
        } else {
          val[ch] = bbd_bilerp(v[kk][ch], &t[kk]);
        }
        s[ch][XS * cl.lds[k]] = val[ch];
      }
      if (warped_out != nullptr && cl.own(k)) {
        float* o = warped_out + cl.pix(k, dm.W);
        o[0] = val[0];
        o[hw] = val[1];
        o[2 * hw] = val[2];
      }
      if (OWNREG > 0) {
        if (k < OWNREG) {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            dvr[k < OWNREG ? k : 0][ch] = (clip[kk] & 1) ? 0.0f : dvx[ch];
            dvr[k < OWNREG ? k : 0][3 + ch] = (clip[kk] & 2) ? 0.0f : dvy[ch];
          }
        }
      }
    }
  }
}

// 3 rows x 8 columns window (6 used) of one plane starting at row r0, column c0 (c0 % 4 == 0).
template <int STRIDE>
__device__ __forceinline__ void load_window(const float* plane, int r0, int c0, float win[3][8]) {
  static_assert(STRIDE % 4 == 0, "rows must stay 16-byte aligned");
  // index in float4 units so the compiler keeps the 16-byte alignment and emits ds_read_b128.  hipcc narrows these
  // loads to the 6 columns that are used and re-pairs them (ds_read_b128 + ds_read2_b32 / ds_read_b64); forcing whole
  // 16-byte reads changed no LDS counter and was 8 % slower (profiles/r02/lds_window_ab_*.txt).
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* p4 = reinterpret_cast<const v4f*>(plane) + (r0 * (STRIDE / 4) + (c0 >> 2));
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const v4f a = p4[r * (STRIDE / 4)];
    const v4f b = p4[r * (STRIDE / 4) + 1];
    win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w;
    win[r][4] = b.x; win[r][5] = b.y; win[r][6] = b.z; win[r][7] = b.w;
  }
}

// Target-window statistics for the strip's 4 pixels, all 3 channels.
__device__ __forceinline__ void strip_ystats(const float (*sy)[FPLANE], int ly, int lx0,
                                             float mu_y[3][PPT], float sg_y[3][PPT]) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float y[3][8];
    load_window<LS>(sy[ch], ly, lx0, y);
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
// RESTAT: the target window's statistics are re-derived here from the target tile (its window is loaded for the x y sums
// anyway: 18 more adds and 9 more multiplications per pixel and channel) instead of being held in 24 registers across the
// candidate loop - the forward's low-register form, see warp_ssim_min_fwd_kernel.
template <bool RESTAT = false>
__device__ __forceinline__ void strip_loss(const float (*sx)[FPLANE], const float (*sy)[FPLANE],
                                           int ly, int lx0, const float mu_y[3][PPT],
                                           const float sg_y[3][PPT], int no_ssim, float out[PPT]) {
  float ssim[PPT][3], l1[PPT][3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float x[3][8], y[3][8];
    load_window<LS>(sx[ch], ly, lx0, x);
    load_window<LS>(sy[ch], ly, lx0, y);
    float nn[PPT], dd[PPT], qq[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float s = 0.0f, ss = 0.0f, sxy = 0.0f, ty = 0.0f, tyy = 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = x[r][j + c];
          s += v;
          ss += v * v;
          sxy += v * y[r][j + c];
          if (RESTAT) {         // (the same row-major sums as strip_ystats: the same bits)
            ty += y[r][j + c];
            tyy += y[r][j + c] * y[r][j + c];
          }
        }
      float my, gy;
      if (RESTAT) bbd_ystats(ty, tyy, &my, &gy);
      else { my = mu_y[ch][j]; gy = sg_y[ch][j]; }
      bbd_ssim_nd(s, ss, sxy, my, gy, &nn[j], &dd[j]);
      l1[j][ch] = fabsf(y[1][j + 1] - x[1][j + 1]);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) qq[j] = bbd_div(nn[j], dd[j]);   // (one validity branch for all four: slower, profiles/r02/guard_variants.txt)
#pragma unroll
    for (int j = 0; j < PPT; ++j) ssim[j][ch] = no_ssim ? 0.0f : bbd_ssim_from_ratio(qq[j]);
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) out[j] = bbd_combine(ssim[j], l1[j], no_ssim);
}

__device__ __forceinline__ void store_strip(float* o, int xx, int W, bool vec_ok, const float v[PPT]) {
  if (vec_ok) {
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if (xx + j < W) o[j] = v[j];
  }
}

__device__ __forceinline__ void load_strip(const float* p, int xx, int W, bool vec_ok, float v[PPT]) {
  if (vec_ok) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int j = 0; j < PPT; ++j) v[j] = (xx + j < W) ? p[j] : 0.0f;
  }
}

// ------------------------------------------------------------------------------------------
// Identity loss (trainer.py:501-508): SSIM+L1 between an un-warped source and the target.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void identity_loss_kernel(FramePtrs frames, const float* __restrict__ target,
                                                           const int32_t* __restrict__ items,
                                                           float* __restrict__ ident, int H, int W,
                                                           int ntiles, int no_ssim, int remap) {
  __shared__ __attribute__((aligned(16))) float s_y[3][FPLANE];
  __shared__ __attribute__((aligned(16))) float s_x[3][FPLANE];
  const int wid = remap ? xcd_work_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int item = wid / ntiles;
  const TileCoord tc = decode_tile(wid - item * ntiles, W);
  const int b = uniform_load(items + item * 4 + 0), slot = uniform_load(items + item * 4 + 1), row = uniform_load(items + item * 4 + 2);
  const int hw = H * W;
  const size_t img = (size_t)3 * hw;
  Cells<LH, LW, LS, 1> cl;
  cl.init(H, W, tc.tx0, tc.ty0);
  stage_image(target + (size_t)b * img, hw, W, cl, s_y);
  stage_image(frames.base[slot] + (size_t)row * img, hw, W, cl, s_x);
  __syncthreads();
  int ly, lx0;
  strip_of_thread(&ly, &lx0);
  float mu_y[3][PPT], sg_y[3][PPT], loss[PPT];
  strip_ystats(s_y, ly, lx0, mu_y, sg_y);
  strip_loss(s_x, s_y, ly, lx0, mu_y, sg_y, no_ssim, loss);
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  if (yy < H) store_strip(ident + (size_t)item * hw + yy * W + xx, xx, W, (xx + PPT <= W) && ((W & 3) == 0), loss);
}

// Grouped form (the training path): one workgroup per (group, tile) walks the group's items - the identity candidates of
// ONE target sample (2 for MD2, up to 6 for the boosted recipe).  The target tile, its cell table and its window
// statistics are set up once per group instead of once per item, and item i + 1's source texels travel (15 coalesced
// loads per thread, held in registers) while item i's SSIM runs; the staged source is double-buffered, one barrier per item.
__global__ __launch_bounds__(NT) void identity_loss_grouped_kernel(FramePtrs frames, const float* __restrict__ target,
                                                                   const int32_t* __restrict__ items,
                                                                   const int32_t* __restrict__ group_off,
                                                                   float* __restrict__ ident, int H, int W, int ntiles,
                                                                   int no_ssim, int remap) {
  __shared__ __attribute__((aligned(16))) float s_y[3][FPLANE];
  __shared__ __attribute__((aligned(16))) float s_x[2][3][FPLANE];
  const int wid = remap ? xcd_work_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int grp = wid / ntiles;
  const TileCoord tc = decode_tile(wid - grp * ntiles, W);
  const int i0 = uniform_load(group_off + grp), i1 = uniform_load(group_off + grp + 1);
  if (i0 >= i1) return;
  const int b = uniform_load(items + i0 * 4 + 0);
  const int hw = H * W;
  const size_t img = (size_t)3 * hw;
  typedef Cells<LH, LW, LS, 1> CellsI;
  CellsI cl;
  cl.init(H, W, tc.tx0, tc.ty0);
  float v[CellsI::N][3];
  auto fetch = [&](int item) {
    const int slot = uniform_load(items + item * 4 + 1), row = uniform_load(items + item * 4 + 2);
    const float* src = frames.base[slot] + (size_t)row * img;
#pragma unroll
    for (int k = 0; k < CellsI::N; ++k) {
      const int px = cl.pix(k, W);
      v[k][0] = src[px];
      v[k][1] = src[px + hw];
      v[k][2] = src[px + 2 * hw];
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int k = 0; k < CellsI::N; ++k) {
      s_x[buf][0][cl.lds[k]] = v[k][0];
      s_x[buf][1][cl.lds[k]] = v[k][1];
      s_x[buf][2][cl.lds[k]] = v[k][2];
    }
  };
  fetch(i0);
  stage_image(target + (size_t)b * img, hw, W, cl, s_y);
  stage(0);
  __syncthreads();
  int ly, lx0;
  strip_of_thread(&ly, &lx0);
  float mu_y[3][PPT], sg_y[3][PPT];
  strip_ystats(s_y, ly, lx0, mu_y, sg_y);
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  const bool vec_ok = (xx + PPT <= W) && ((W & 3) == 0);
  for (int item = i0; item < i1; ++item) {
    const int buf = (item - i0) & 1;
    if (item + 1 < i1) fetch(item + 1);
    float loss[PPT];
    strip_loss(s_x[buf], s_y, ly, lx0, mu_y, sg_y, no_ssim, loss);
    if (yy < H) store_strip(ident + (size_t)item * hw + yy * W + xx, xx, W, vec_ok, loss);
    if (item + 1 < i1) {
      stage(buf ^ 1);
      __syncthreads();
    }
  }
}

// torch.min(dim) as an order-free update: smaller wins, on equal values the smaller id, a NaN wins over numbers and
// among NaNs the smaller id (= "first index", "NaN wins and sticks" of the sequential form bbd_min_update)
__device__ __forceinline__ void min_update_any_order(float cand, int id, float* best, int* arg) {
  const bool cn = cand != cand, bn = *best != *best;
  const bool take = (cand < *best) || ((cand == *best || (cn && bn)) && id < *arg) || (cn && !bn);
  *best = take ? cand : *best;
  *arg = take ? id : *arg;
}

// Visiting order of a sample's candidates: ascending id, except that a warp candidate is followed at once by its pairing
// partner (bits 16-23 of `kind`: the error-induced warp of the same source frame).  The pair samples almost the same
// texels, so the second candidate's gathers hit the lines the first brought in (L1 / this XCD's L2) instead of going
// back to the fabric.  Uniform (scalar) bookkeeping; results do not depend on the order (min_update_any_order).
struct CandOrder {
  unsigned todo;
  int forced;
  __device__ __forceinline__ void init(int nc) { todo = nc >= 32 ? 0xffffffffu : ((1u << nc) - 1u); forced = -1; }
  __device__ __forceinline__ bool more() const { return todo != 0u || forced >= 0; }
  __device__ __forceinline__ int next(const bbd_cand_t* tab, bbd_cand_t* cd) {
    int c;
    if (forced >= 0) { c = forced; forced = -1; }
    else { c = __builtin_ctz(todo); todo &= todo - 1u; }
    *cd = load_cand(tab + c);
    pair_up(*cd);
    return c;
  }
  __device__ __forceinline__ void pair_up(const bbd_cand_t& cd) {
    if ((cd.kind & KIND_MASK) == BBD_KIND_WARP) {
      const int hint = ((cd.kind >> 16) & 0xff) - 1;
      if (hint >= 0 && hint < 32 && ((todo >> hint) & 1u)) { forced = hint; todo &= ~(1u << hint); }
    }
  }
};

// ------------------------------------------------------------------------------------------
// Fused forward: warp + SSIM/L1 + min/arg-min over the candidate list.
// ------------------------------------------------------------------------------------------
// Diagnostic build only (-DBBD_STAMPS): wave 0 of every workgroup records s_memtime at phase
// boundaries into a buffer registered with bbd_debug_set_stamps(); never part of the shipped library.
#ifdef BBD_STAMPS
static unsigned long long* g_stamps_host_ptr = nullptr;
#define BBD_STAMP(k)                                                                         \
  do {                                                                                       \
    if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 32 + (k)] = clock64(); \
  } while (0)
// wall-clock (100 MHz, one counter for the whole chip) forms: where a workgroup starts / ends inside the launch, and a
// plain value (e.g. how many candidates it processed) - tools/stamps_timeline.py
#define BBD_STAMP_RT(k)                                                                      \
  do {                                                                                       \
    if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define BBD_STAMP_VAL(k, v)                                                                  \
  do {                                                                                       \
    if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 32 + (k)] = (unsigned long long)(v); \
  } while (0)
#elif defined(BBD_MARKS)
// static-profile build (tools/isa_phases.py): every stamp site leaves a comment in the assembly, so the instruction mix can
// be summed per phase without running anything
#define BBD_STAMP(k) asm volatile("; bbd_mark " #k)
#define BBD_STAMP_RT(k) do { } while (0)
#define BBD_STAMP_VAL(k, v) do { } while (0)
#else
#define BBD_STAMP(k) do { } while (0)
#define BBD_STAMP_RT(k) do { } while (0)
#define BBD_STAMP_VAL(k, v) do { } while (0)
#endif

struct FwdArgs {
  unsigned long long* stamps;
  FramePtrs frames;
  const float* target;
  const float* depth;
  const float* pose;
  const float* ident;
  const float* noise;
  const bbd_cand_t* cand;
  const int32_t* ncand;
  const int32_t* work;     // optional [S*B*ntiles][2]: bbd_fused_work_items' table (NULL: grid order, decoded here)
  float* min_loss;
  uint8_t* argmin;
  float* partial;
  float* warped;
  float* depth_out;     // optional [S,B,H,W]: the depth this launch used (outputs[("depth",0,s)] of the reference)
  DispSrc ds;
  BbdDims dm;
  int S, B, NP, ntiles, no_ssim, remap;
};

// Three forms of the forward, chosen per launch by launch_fused_fwd (round 4; VERDICT r3 item 1c measured INSIDE the training
// step, profiles/r04/forward_four_waves_instep.txt - round 3 had measured four-wave forms with scattered gathers only and
// found nothing):
//   FWD_DOUBLE   three waves per SIMD (<= 168 VGPRs: 146 used), the warped tile double-buffered (44 KB of LDS): candidate
//                c+1 is warped into the other buffer while slower waves still read candidate c, one barrier per candidate.
//                Single-scale launches with many candidates (the boosted recipe from epoch 10 on): 1 440 workgroups are
//                1.9 rounds of 768 slots but only 1.4 of 1 024 - the fourth wave loses to the launch's quantisation there
//                (epoch-15 draw +3.4 % with four waves, uniform m = 7 -3.4 %: the mixed batches are the real distribution).
//   FWD_RESTAT   four waves per SIMD, single buffer (29 KB, a second barrier per candidate), the target window's statistics
//                re-derived per candidate inside strip_loss instead of held in 24 registers: 119 VGPRs, no spill.  Few warp
//                candidates per sample (MD2, MonoViT): MD2 forward -3.6 %; +3 % at 12 warp candidates.
//   FWD_HELD     four waves per SIMD, single buffer, statistics held: 17 registers spilled at the 128 line (none in the warp
//                phase).  Many candidates AND several scales (the early curriculum, trimin5): -7.8 %.
constexpr int FWD_DOUBLE = 0, FWD_RESTAT = 1, FWD_HELD = 2;
template <bool PLANE, int FORM>
__global__ __launch_bounds__(NT, FORM == FWD_DOUBLE ? 3 : 4) void warp_ssim_min_fwd_kernel(FwdArgs a) {
  constexpr bool RESTAT = FORM == FWD_RESTAT;
  __shared__ __attribute__((aligned(16))) float s_y[3][FPLANE];
  __shared__ __attribute__((aligned(16))) float s_xx[FORM == FWD_DOUBLE ? 2 : 1][3][FPLANE];
  __shared__ float s_red[4];
  int buf = 0;
  const BbdDims dm = a.dm;
  const int H = dm.H, W = dm.W, hw = H * W;
  // grid order: sample-major, then scale, then tile.  (A form in which one workgroup per (sample, tile) walked the scales
  // itself - cell tables, staged target tile and window statistics set up once - measured neutral inside the training
  // step and 15 % slower on the micro-benchmark, and its loop structure alone cost this kernel 14 VGPRs and 19 spilled
  // SGPRs: removed, profiles/r03/fwd_scale_loop_ab.txt.)
  int b, s;
  TileCoord tc;
  if (a.work != nullptr) {
    const WorkItem it = load_work_item(a.work);
    b = it.b; s = it.s;
    tc.tile = it.tile; tc.tx0 = it.tx0; tc.ty0 = it.ty0;
  } else {
    int bid = a.remap ? xcd_work_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    b = bid / (a.S * a.ntiles);
    bid -= b * a.S * a.ntiles;
    s = bid / a.ntiles;
    tc = decode_tile(bid - s * a.ntiles, W);
  }
  const size_t img = (size_t)3 * hw;
  const size_t sb = (size_t)s * a.B + b;

  BBD_STAMP_RT(30);
  BBD_STAMP(0);
  // set-up: every global load is issued before anything waits - the sample's candidate tables, the target tile, the
  // depth sources of the staged cells
  typedef Cells<LH, LW, LS, 1> CellsF;
  const int nc = uniform_load(a.ncand + b);
  CellsF cl;
  cl.init(H, W, tc.tx0, tc.ty0);
  float tv[CellsF::N][3];
  {
    const float* tg = a.target + (size_t)b * img;
#pragma unroll
    for (int k = 0; k < CellsF::N; ++k) {
      const int px = cl.pix(k, W);
      tv[k][0] = tg[px];
      tv[k][1] = tg[px + hw];
      tv[k][2] = tg[px + 2 * hw];
    }
  }
  const DepthSrc dsrc = depth_source(a.depth, a.ds, s, b, sb, H, W);
  DepthFetch<CellsF::N, PLANE> dfetch;
#pragma unroll
  for (int k = 0; k < CellsF::N; ++k) dfetch.issue(dsrc, k, cl.xy[k] >> 16, cl.xy[k] & 0xffff, H, W);
#pragma unroll
  for (int k = 0; k < CellsF::N; ++k) {
    s_y[0][cl.lds[k]] = tv[k][0];
    s_y[1][cl.lds[k]] = tv[k][1];
    s_y[2][cl.lds[k]] = tv[k][2];
  }
  BBD_STAMP(1);
  __syncthreads();
  BBD_STAMP(2);
  int ly, lx0;
  strip_of_thread(&ly, &lx0);
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  const bool row_ok = yy < H;
  const bool vec_ok = (xx + PPT <= W) && ((W & 3) == 0);
  const int pix = yy * W + xx;

  float mu_y[3][PPT], sg_y[3][PPT];
  if (!RESTAT) strip_ystats(s_y, ly, lx0, mu_y, sg_y);

  float dcell[CellsF::N];
#pragma unroll
  for (int k = 0; k < CellsF::N; ++k) dcell[k] = dfetch.finish(dsrc, k, H, W);
  if (a.depth_out != nullptr) {
#pragma unroll
    for (int k = 0; k < CellsF::N; ++k)
      if (cl.own(k)) a.depth_out[sb * hw + cl.pix(k, W)] = dcell[k];
  }

  // (register budget: this kernel sat on the 168-VGPR line of 3 waves per SIMD, 154 now - the four arg-min ids share one
  // word and the identity noise is fetched where an identity candidate needs it, not held across the loop)
  float best[PPT];
  unsigned argw = 0u;
#pragma unroll
  for (int j = 0; j < PPT; ++j) best[j] = INFINITY;
  BBD_STAMP(3);

  // pass 1: the warp candidates (a frame's two warps back to back); identity candidates are only noted
  CandOrder order;
  order.init(nc);
  unsigned idents = 0u;
  int visit = 0;
  while (order.more()) {
    bbd_cand_t cd;
    const int c = order.next(a.cand + b * BBD_MAX_CAND, &cd);
    if ((cd.kind & KIND_MASK) != BBD_KIND_WARP) {
      idents |= 1u << c;
      continue;
    }
    const int vs = visit++;
    (void)vs;
    float loss[PPT];
    const float* src = a.frames.base[cd.slot] + (size_t)cd.row * img;
    float* wout = a.warped ? a.warped + ((size_t)s * a.NP + cd.pose) * img : nullptr;
    float pj[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) pj[i] = uniform_load(a.pose + (size_t)cd.pose * BBD_PROJ_STRIDE + i);
    BBD_STAMP(4 + 4 * (vs & 3));
    warp_into_lds<BBD_WARP_BATCH, CellsF, FPLANE>(src, dcell, pj, dm, hw, cl, s_xx[buf], wout);
    BBD_STAMP(5 + 4 * (vs & 3));
    __syncthreads();
    BBD_STAMP(6 + 4 * (vs & 3));
    strip_loss<RESTAT>(s_xx[buf], s_y, ly, lx0, mu_y, sg_y, a.no_ssim, loss);
    BBD_STAMP(7 + 4 * (vs & 3));
    if (FORM == FWD_DOUBLE) buf ^= 1;
    else __syncthreads();     // (single buffer: the next candidate's warp phase overwrites the tile)
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      int aj = (int)((argw >> (8 * j)) & 0xffu);
      min_update_any_order(loss[j], c, &best[j], &aj);
      argw = (argw & ~(0xffu << (8 * j))) | ((unsigned)aj << (8 * j));
    }
  }
  // pass 2: the identity candidates (trainer.py:501-508, :518-523: identity map + the sample's noise).  The running minimum
  // is order-free, so they can all come last, and then the next candidate's strip travels while this one is compared -
  // inside the loop above each of them was a global round trip of its own.
  if (idents != 0u) {
    float nz[PPT] = {0.0f, 0.0f, 0.0f, 0.0f}, cur[PPT] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row_ok && a.noise != nullptr) load_strip(a.noise + (size_t)b * hw + pix, xx, W, vec_ok, nz);
    int c = __builtin_ctz(idents);
    idents &= idents - 1u;
    if (row_ok) load_strip(a.ident + (size_t)uniform_load(&a.cand[b * BBD_MAX_CAND + c].row) * hw + pix, xx, W, vec_ok, cur);
    while (true) {
      const int cn = idents != 0u ? __builtin_ctz(idents) : -1;
      float nxt[PPT] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (cn >= 0 && row_ok) load_strip(a.ident + (size_t)uniform_load(&a.cand[b * BBD_MAX_CAND + cn].row) * hw + pix, xx, W, vec_ok, nxt);
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        int aj = (int)((argw >> (8 * j)) & 0xffu);
        const float lj = a.noise != nullptr ? cur[j] + nz[j] : cur[j];
        min_update_any_order(row_ok ? lj : 0.0f, c, &best[j], &aj);
        argw = (argw & ~(0xffu << (8 * j))) | ((unsigned)aj << (8 * j));
      }
      if (cn < 0) break;
      idents &= idents - 1u;
      c = cn;
#pragma unroll
      for (int j = 0; j < PPT; ++j) cur[j] = nxt[j];
    }
  }

  float tsum = 0.0f;
  if (row_ok) {
    store_strip(a.min_loss + sb * hw + pix, xx, W, vec_ok, best);
    uint8_t* ao = a.argmin + sb * hw + pix;
    if (vec_ok) {
      *reinterpret_cast<uint32_t*>(ao) = argw;
      tsum = ((best[0] + best[1]) + best[2]) + best[3];
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j)
        if (xx + j < W) {
          ao[j] = (uint8_t)(argw >> (8 * j));
          tsum += best[j];
        }
    }
  }
  // deterministic per-tile sum: DPP wave reduction, then the four wave totals in fixed order
  const float wsum = wave_sum63(tsum);
  if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) a.partial[sb * a.ntiles + tc.tile] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  BBD_STAMP(20);
  BBD_STAMP_RT(31);
}

// ------------------------------------------------------------------------------------------
// Fused backward.
// ------------------------------------------------------------------------------------------
struct BwdArgs {
  unsigned long long* stamps;
  FramePtrs frames;
  const float* target;
  const float* depth;
  const float* pose;
  const bbd_cand_t* cand;
  const int32_t* ncand;
  const int32_t* work;
  const uint8_t* argmin;
  const float* gscale;
  float* grad_depth;    // depth-plane mode: d loss / d depth; disparity mode: d loss / d up-sampled disparity
  float* grad_proj;
  DispSrc ds;
  BbdDims dm;
  int S, B, NP, ntiles, no_ssim, remap;
};

// ------------------------------------------------------------------------------------------
// Fused backward: the same LDS images and phases as the round-1 wide form on a 32x16-pixel tile, 256 threads = 4 waves, a
// 2-pixel strip per thread (round 2: 128 VGPRs and 38-41 KB of LDS, so FOUR independent workgroups = 16 waves per CU are
// resident: the warp and sample-gradient phases wait on gather latency, which only other resident waves can hide, and a
// barrier of one workgroup no longer idles half the CU; profiles/r02/bwd_variants.txt).  The halo'd warp region is 1.41x
// the tile.  2-pixel windows are 8-byte aligned ds_read_b64.
// ------------------------------------------------------------------------------------------
constexpr int TW2 = 32;              // backward tile width (pixels); height stays TH
constexpr int NT2 = 256;
constexpr int PPT2 = 2;
constexpr int SPR2 = TW2 / PPT2;     // 16 strips per tile row
constexpr int BS2 = TW2 + 4;         // 36: row stride of the x / y regions (TH+4) x (TW2+4)
constexpr int BW2 = TW2 + 4;
constexpr int BPLANE2 = BH * BS2;
constexpr int CW2 = TW2 + 2;         // the coefficient region is (TH+2) x (TW2+2) loss pixels: the tile and a one-pixel ring

__device__ __forceinline__ TileCoord decode_tile2(int t, int W) {
  const int tiles_x = (W + TW2 - 1) / TW2;
  TileCoord c;
  c.tile = t;
  c.ty0 = (t / tiles_x) * TH;
  c.tx0 = (t % tiles_x) * TW2;
  return c;
}

template <int STRIDE>
__device__ __forceinline__ void load_window4(const float* plane, int r0, int c0, float win[3][4]) {
  static_assert(STRIDE % 2 == 0, "rows must stay 8-byte aligned");
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f* p2 = reinterpret_cast<const v2f*>(plane) + (r0 * (STRIDE / 2) + (c0 >> 1));
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const v2f a = p2[r * (STRIDE / 2)];
    const v2f b = p2[r * (STRIDE / 2) + 1];
    win[r][0] = a.x; win[r][1] = a.y; win[r][2] = b.x; win[r][3] = b.y;
  }
}

#ifndef BBD_BWD2_WGS
#define BBD_BWD2_WGS 4   // waves per SIMD (HIP: second __launch_bounds__ argument): <= 128 VGPRs (120 used).
                         // History (profiles/r02): with the heavier mid-round kernel 3, 4 and 5 waves per SIMD ran within 1 %
                         // (bwd_variants_narrow_tile.txt) and 4 cost 15 spilled registers (scratch doubled the launch's HBM
                         // bytes); after the instruction diet (unguarded recompute, reduce-scatter, no SLP packing) the
                         // kernel fits 128 VGPRs and 4 waves per SIMD are 11 % faster than 3 (occupancy_after_diet.txt)
#endif
#ifndef BBD_BWD2_WARP_BATCH
#define BBD_BWD2_WARP_BATCH 3
#endif
#ifndef BBD_BWD_PRESENT_ONLY
#define BBD_BWD_PRESENT_ONLY 1   // round 5: for samples with more than four candidates the loop visits only the candidates
                                 // present in the tile (profiles/r05/bwd_skip_paths.txt)
#endif


// Round 4, the nine-plane form (the only one since the end of that round; the per-channel form of rounds 2-4 is kept as
// tools/experiments/per_channel_backward.hip.txt): (1) the coefficient planes of all three colour channels sit in LDS at
// once, so a candidate needs 3 barriers instead of 7; (2) the winners' SSIM partials are ONE walk over (winner, channel)
// items - a candidate of the boosted recipe wins a twelfth of a tile, and a walk per channel kept a fifth of the lanes busy
// for three rounds; (3) to keep four workgroups per CU in LDS (40 816 B) the six d warped / d (ix, iy) planes left LDS: a
// thread's first two staged cells are its own strip (CellsBwd) and their tap differences stay in its registers.  In the
// training step: MD2 backward -3.5 %, boosted m = 7 -12 %, epoch-15 draw -11 % (profiles/r04/setup_variants.txt).  125
// VGPRs, no spill - the first build of this form held the border tiles' reflection multiplicities across the channel loop
// and spilled 10 registers, which hid its gain for MD2.
constexpr int CS9 = TW2 + 2;         // 34: row stride of the coefficient region - exactly its width
constexpr int CPLANE9 = CH * CS9;

// The nine-plane form's staged cells: (TH+4) x (TW2+4) = 720 cells on 256 threads.  Cells 0 and 1 of a thread are the two
// pixels of its OWN strip, cell 2 is one of the 208 cells of the two-cell frame around the tile's pixels (threads 208..255
// repeat the last one: same value to the same LDS word).
struct CellsBwd {
  static constexpr int N = 3;
  int lds[N];
  int xy[N];
  __device__ __forceinline__ void init(int H, int W, int tx0, int ty0) {
    const int t = (int)threadIdx.x;
    int r[N], c[N];
    r[0] = r[1] = t / SPR2 + 2;
    c[0] = (t % SPR2) * PPT2 + 2;
    c[1] = c[0] + 1;
    const int h = t < 207 ? t : 207;
    const int top = h / BW2;                              // rows 0, 1 (h < 72) and 18, 19 (72 <= h < 144)
    const int side = (h - 144) >> 1;                      // columns 0, 1 (144 <= h < 176) and 34, 35: 16 rows each
    r[2] = h < 144 ? (top < 2 ? top : top + TH) : 2 + (side & (TH - 1));
    c[2] = h < 144 ? h - top * BW2 : (h < 176 ? (h & 1) : TW2 + 2 + (h & 1));
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int yy = bbd_reflect(ty0 + r[k] - 2, H), xx = bbd_reflect(tx0 + c[k] - 2, W);
      lds[k] = r[k] * BS2 + c[k];
      xy[k] = (yy << 16) | xx;
    }
  }
  __device__ __forceinline__ int pix(int k, int W) const { return (xy[k] >> 16) * W + (xy[k] & 0xffff); }
  __device__ __forceinline__ bool own(int) const { return false; }          // (forward-only uses of the shared warp code)
};

template <bool PLANE>
__global__ __launch_bounds__(NT2, BBD_BWD2_WGS) void warp_ssim_min_bwd9_kernel(BwdArgs a) {
  // warped (x) and target (y) texels of a staged cell sit side by side - planes of (x, y) pairs: every window read of
  // the coefficient phase is one 8-byte load that lands in a register pair, and its five running sums become three
  // packed operations per tap ((sx, sy) += (x, y); (sxx, syy) += (x, y)^2) plus the scalar sxy
  __shared__ __attribute__((aligned(16))) float s_xybuf[3 * 2 * BPLANE2 + 16];
  // coefficient planes of ALL THREE colour channels ([3 * ch + {A, B, C}])
  __shared__ __attribute__((aligned(16))) float s_cf[9][CPLANE9];
  __shared__ uint16_t s_list[CH * CW2];
  __shared__ float s_red[NT2 / 64][12];
  __shared__ unsigned s_present[NT2 / 64];
  __shared__ int s_count;
  float (*s_xy)[2 * BPLANE2] = reinterpret_cast<float (*)[2 * BPLANE2]>(s_xybuf);      // [ch][2 * cell + {0: x, 1: y}]
  const BbdDims dm = a.dm;
  const int H = dm.H, W = dm.W, hw = H * W;
  int b, s;
  TileCoord tc;
  if (a.work != nullptr) {
    const WorkItem it = load_work_item(a.work);
    b = it.b; s = it.s;
    tc.tile = it.tile; tc.tx0 = it.tx0; tc.ty0 = it.ty0;
  } else {
    int bid = a.remap ? xcd_work_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    b = bid / (a.S * a.ntiles);
    bid -= b * a.S * a.ntiles;
    s = bid / a.ntiles;
    tc = decode_tile2(bid - s * a.ntiles, W);
  }
  const size_t img = (size_t)3 * hw;
  const size_t sb = (size_t)s * a.B + b;
  const DepthSrc dsrc = depth_source(a.depth, a.ds, s, b, sb, H, W);
  const uint8_t* am = a.argmin + sb * hw;
  const float g = uniform_load(a.gscale + s);
  const float w_ssim = a.no_ssim ? 0.0f : g * 0.85f / 3.0f;
  const float w_l1 = a.no_ssim ? g / 3.0f : g * 0.15f / 3.0f;

  BBD_STAMP_RT(30);
  BBD_STAMP(0);
  // setup: issue every global load first, then the LDS work that does not depend on them
  const int nc = uniform_load(a.ncand + b);
  constexpr int NP_CELLS = (CH * CW2 + NT2 - 1) / NT2;
  static_assert(NP_CELLS <= 4, "arg-min ids of the loss pixels are packed four to a word");
  int pcell[NP_CELLS];
  unsigned pargw = 0u;                 // byte k = arg-min id of loss pixel k (255 = outside the image)
#pragma unroll
  for (int k = 0; k < NP_CELLS; ++k) {
    const int i = k * NT2 + (int)threadIdx.x;
    const int r = i / CW2, c = i - r * CW2;
    const int py = tc.ty0 + r - 1, px = tc.tx0 + c - 1;
    const bool in = i < CH * CW2 && py >= 0 && py < H && px >= 0 && px < W;
    pcell[k] = r * CS9 + c;
    pargw |= (in ? (unsigned)am[py * W + px] : 255u) << (8 * k);
  }
#define BBD_PARG(k) ((pargw >> (8 * (k))) & 0xffu)
  BBD_STAMP(21);
  typedef CellsBwd CellsB;
  CellsB cl;
  cl.init(H, W, tc.tx0, tc.ty0);
  float tcell[CellsB::N][3];
  {
    const float* tg = a.target + (size_t)b * img;
#pragma unroll
    for (int k = 0; k < CellsB::N; ++k) {
      const int px = cl.pix(k, W);
      tcell[k][0] = tg[px];
      tcell[k][1] = tg[px + hw];
      tcell[k][2] = tg[px + 2 * hw];
    }
  }
  const int ly = (int)threadIdx.x / SPR2, lx0 = ((int)threadIdx.x % SPR2) * PPT2;
  const int qy = tc.ty0 + ly, qx0 = tc.tx0 + lx0;
  const bool q_row_ok = qy < H;
  const bool q_vec_ok = (qx0 + PPT2 <= W) && ((W & 1) == 0);
  // depth of the halo'd cells (only needed to project them) and of the strip's own pixels: one batch of loads (pixels
  // beyond the image read a clamped address, their value is never used)
  DepthFetch<CellsB::N, PLANE> dfetch;
#pragma unroll
  for (int k = 0; k < CellsB::N; ++k) dfetch.issue(dsrc, k, cl.xy[k] >> 16, cl.xy[k] & 0xffff, H, W);

  BBD_STAMP(22);
  unsigned qargw = 0u;                 // byte j = arg-min id of own pixel j
#pragma unroll
  for (int j = 0; j < PPT2; ++j) qargw |= ((q_row_ok && qx0 + j < W) ? (unsigned)am[qy * W + qx0 + j] : 255u) << (8 * j);
  const bool interior = tc.tx0 >= 2 && tc.tx0 + TW2 + 2 <= W && tc.ty0 >= 2 && tc.ty0 + TH + 2 <= H;
  float gdepth[PPT2] = {0.0f, 0.0f};

  BBD_STAMP(23);
  if (threadIdx.x == 0) s_count = 0;
  static_assert((9 * CPLANE9) % 4 == 0, "coefficient planes are cleared 16 bytes at a time");
  for (int i = threadIdx.x; i < 9 * CPLANE9 / 4; i += NT2)
    reinterpret_cast<float4*>(&s_cf[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  BBD_STAMP(24);
  {
    // which candidates won a pixel in or next to this tile: OR over the wave on the DPP crossbar, one word per wave - no
    // LDS atomic and no barrier of its own (round 3 zeroed a shared word, synchronised, then OR-ed into it atomically)
    unsigned mine = 0u;
#pragma unroll
    for (int k = 0; k < NP_CELLS; ++k)
      if (BBD_PARG(k) != 255u) mine |= 1u << BBD_PARG(k);
    const unsigned wave_mine = wave_or63(mine);
    if ((threadIdx.x & 63) == 63) s_present[threadIdx.x >> 6] = wave_mine;
  }
  BBD_STAMP(26);
#pragma unroll
  for (int k = 0; k < CellsB::N; ++k) {
    s_xy[0][2 * cl.lds[k] + 1] = tcell[k][0];
    s_xy[1][2 * cl.lds[k] + 1] = tcell[k][1];
    s_xy[2][2 * cl.lds[k] + 1] = tcell[k][2];
  }
  float dcell[CellsB::N], qdepth[PPT2];
#pragma unroll
  for (int k = 0; k < CellsB::N; ++k) dcell[k] = dfetch.finish(dsrc, k, H, W);
#pragma unroll
  for (int j = 0; j < PPT2; ++j) qdepth[j] = (q_row_ok && qx0 + j < W) ? dcell[j] : 1.0f;      // (cells 0, 1 = own strip)
  BBD_STAMP(1);
  __syncthreads();
  BBD_STAMP(2);
  unsigned present = 0u;
#pragma unroll
  for (int w8 = 0; w8 < NT2 / 64; ++w8) present |= s_present[w8];

  int prev = -1;
  CandOrder order;                       // a frame's true-pose and error-induced warps back to back (cache locality)
  order.init(nc);
#if BBD_BWD_PRESENT_ONLY
  // Samples with many candidates (the boosted recipe: 8 / 14 / 18): only the candidates that won a pixel in or next to this
  // tile are visited.  The others' pose-gradient partials are zeros, written here by one lane per candidate - in parallel,
  // instead of one scalar descriptor load (a serial round trip) + branch per absent candidate in the loop below: a trained
  // network leaves 2-6 of the 12 warp candidates alive in a tile (-2.6 % there, profiles/r05/bwd_skip_paths.txt).  MD2's
  // four candidates keep the plain loop (the block below cost it 0.9 %, present_ab.txt); block-uniform choice.
  if (nc > 4) {
    if ((int)threadIdx.x < nc && !((present >> threadIdx.x) & 1u)) {
      const bbd_cand_t* cp = a.cand + b * BBD_MAX_CAND + threadIdx.x;
      const int kind = cp->kind, pose = cp->pose;
      if ((kind & KIND_MASK) == BBD_KIND_WARP) {
        float* z = a.grad_proj + (((size_t)s * a.NP + pose) * a.ntiles + tc.tile) * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) z[k] = 0.0f;
      }
    }
    order.todo &= present;
  }
#endif
  while (order.more()) {
    bbd_cand_t cd;
    const int c = order.next(a.cand + b * BBD_MAX_CAND, &cd);
    if ((cd.kind & KIND_MASK) != BBD_KIND_WARP) continue;
    float* gp_out = a.grad_proj + (((size_t)s * a.NP + cd.pose) * a.ntiles + tc.tile) * 12;
    if (!((present >> c) & 1u)) {
      if (threadIdx.x < 12) gp_out[threadIdx.x] = 0.0f;
      continue;
    }
    const float* src = a.frames.base[cd.slot] + (size_t)cd.row * img;
    float pj[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) pj[i] = uniform_load(a.pose + (size_t)cd.pose * BBD_PROJ_STRIDE + i);

    // ---- phase W; the winners' list of this candidate and the clearing of the previous one's coefficient entries are
    // LDS work that does not depend on the warp: done while the first batch of gathers is in flight
    const int prev_c = prev;
    auto lists = [&]() {
      if (!a.no_ssim) {
#pragma unroll
        for (int k = 0; k < NP_CELLS; ++k) {
          if ((int)BBD_PARG(k) == prev_c) {
#pragma unroll
            for (int pl = 0; pl < 9; ++pl) s_cf[pl][pcell[k]] = 0.0f;
          }
          if (BBD_PARG(k) == (unsigned)c) s_list[atomicAdd(&s_count, 1)] = (uint16_t)pcell[k];
        }
      }
    };
    prev = c;
    float dvr[PPT2][6];
    BBD_STAMP(4 + 8 * (c & 1));
    warp_into_lds<BBD_BWD2_WARP_BATCH, CellsB, 2 * BPLANE2, true, 2, decltype(lists), PPT2>(src, dcell, pj, dm, hw, cl, s_xy,
                                                                                            nullptr, lists, dvr);
    BBD_STAMP(5 + 8 * (c & 1));
    __syncthreads();
    BBD_STAMP(6 + 8 * (c & 1));

    // ---- phase C: ONE walk over (winner, colour channel) items - with a twelfth of a tile's pixels per candidate (the boosted
    // recipe) a walk per channel kept a fifth of the lanes busy for three rounds; 3 x nwin items fill the lanes in one
    const int nwin = a.no_ssim ? 0 : s_count;
    for (int item = threadIdx.x; item < 3 * nwin; item += NT2) {
      const int ch = (item >= nwin) + (item >= 2 * nwin);
      const int cell = s_list[item - ch * nwin];
      const int pr = cell / CS9, pc = cell - pr * CS9;
      v2f s1 = pk1(0.0f), s2 = pk1(0.0f);       // (sum x, sum y), (sum x^2, sum y^2)
      float sxy = 0.0f;
      const v2f* xyp = reinterpret_cast<const v2f*>(&s_xy[ch][0]);
#pragma unroll
      for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
          const v2f xy = xyp[(pr + dr) * BS2 + pc + dc];
          s1 = s1 + xy;
          s2 = s2 + xy * xy;
          sxy += xy.x * xy.y;
        }
      float mu_y, sg_y, A, Bc, Cc;
      bbd_ystats(s1.y, s2.y, &mu_y, &sg_y);
      bbd_ssim_grad(s1.x, s2.x, sxy, mu_y, sg_y, &A, &Bc, &Cc);
      float* cf = &s_cf[3 * ch][cell];
      cf[0] = A * w_ssim;
      cf[CPLANE9] = Bc * w_ssim;
      cf[2 * CPLANE9] = Cc * w_ssim;
    }
    BBD_STAMP(7 + 8 * (c & 1));
    __syncthreads();
    BBD_STAMP(8 + 8 * (c & 1));
    if (threadIdx.x == 0) s_count = 0;

    // ---- phase G: adjoint of reflect-pad + 3x3 mean at this thread's 2 texels, channel by channel (no barrier between)
    float gx[3][PPT2];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float4 xy2 = *reinterpret_cast<const float4*>(&s_xy[ch][2 * ((ly + 2) * BS2 + lx0 + 2)]);   // (x0, y0, x1, y1)
      const float xqv[PPT2] = {xy2.x, xy2.z}, yqv[PPT2] = {xy2.y, xy2.w};
      float S3[3][PPT2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < PPT2; ++j) S3[pl][j] = 0.0f;
      if (!a.no_ssim) {
        if (interior) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            // the 3x4 window arrives as register pairs (two 8-byte reads per row): column sums on pairs
            const v2f* p2 = reinterpret_cast<const v2f*>(s_cf[3 * ch + pl]) + (ly * (CS9 / 2) + (lx0 >> 1));
            const v2f colA = (p2[0] + p2[CS9 / 2]) + p2[CS9];                  // columns 0, 1
            const v2f colB = (p2[1] + p2[CS9 / 2 + 1]) + p2[CS9 + 1];          // columns 2, 3
            const float mid = colA.y + colB.x;
            S3[pl][0] = colA.x + mid;
            S3[pl][1] = mid + colB.y;
          }
        } else {
          float wy[3], wx[PPT2][3];
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int py = qy + d - 1;
            wy[d] = (py >= 0 && py < H) ? (float)bbd_reflect_mult(qy, py, H) : 0.0f;
#pragma unroll
            for (int j = 0; j < PPT2; ++j) {
              const int px = qx0 + j + d - 1;
              wx[j][d] = (px >= 0 && px < W) ? (float)bbd_reflect_mult(qx0 + j, px, W) : 0.0f;
            }
          }
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            float cw[3][4];
            load_window4<CS9>(s_cf[3 * ch + pl], ly, lx0, cw);
#pragma unroll
            for (int j = 0; j < PPT2; ++j)
#pragma unroll
              for (int dr = 0; dr < 3; ++dr) {
                float r = 0.0f;
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) r = fmaf(wx[j][dc], cw[dr][j + dc], r);
                S3[pl][j] = fmaf(wy[dr], r, S3[pl][j]);
              }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PPT2; ++j) {
        const float xq = xqv[j], yq = yqv[j];
        float acc = (S3[0][j] + xq * S3[1][j] + yq * S3[2][j]) * (1.0f / 9.0f);
        if (((qargw >> (8 * j)) & 0xffu) == (unsigned)c) {
          const float df = xq - yq;
          acc += w_l1 * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
        }
        gx[ch][j] = (q_row_ok && qx0 + j < W) ? acc : 0.0f;
      }
    }

    BBD_STAMP(9 + 8 * (c & 1));
    // texel gradient -> sampling coordinates -> depth and P; the strip's two pixels as the two halves of packed
    // registers (bbd_sample_smooth + bbd_project_grad, operation for operation: same arithmetic, half the instructions)
    float gP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gP[k] = 0.0f;
    if (q_row_ok) {
      // (a fresh scalar load instead of 21 SGPRs held across the C / G phases)
#pragma unroll
      for (int i = 0; i < 21; ++i) pj[i] = uniform_load(a.pose + (size_t)cd.pose * BBD_PROJ_STRIDE + i);
      v2f dxy[6];
#pragma unroll
      for (int pl = 0; pl < 6; ++pl) dxy[pl] = pk2(dvr[0][pl], dvr[1][pl]);
      const v2f g0 = pk2(gx[0][0], gx[0][1]), g1 = pk2(gx[1][0], gx[1][1]), g2 = pk2(gx[2][0], gx[2][1]);   // (0 beyond W)
      const v2f gix = g0 * dxy[0] + g1 * dxy[1] + g2 * dxy[2];
      const v2f giy = g0 * dxy[3] + g1 * dxy[4] + g2 * dxy[5];
      const float* iK = pj + 12;
      const v2f fx = pk2((float)qx0, (float)(qx0 + 1)), fy = pk1((float)qy), dep = pk2(qdepth[0], qdepth[1]);
      const v2f cx = pk_fma(pk1(iK[1]), fy, pk1(iK[0]) * fx) + pk1(iK[2]);
      const v2f cy = pk_fma(pk1(iK[4]), fy, pk1(iK[3]) * fx) + pk1(iK[5]);
      const v2f cz = pk_fma(pk1(iK[7]), fy, pk1(iK[6]) * fx) + pk1(iK[8]);
      const v2f X = dep * cx, Y = dep * cy, Z = dep * cz;
      const v2f qx_ = pk_fma(pk1(pj[2]), Z, pk_fma(pk1(pj[1]), Y, pk1(pj[0]) * X)) + pk1(pj[3]);
      const v2f qy_ = pk_fma(pk1(pj[6]), Z, pk_fma(pk1(pj[5]), Y, pk1(pj[4]) * X)) + pk1(pj[7]);
      const v2f zi = pk_fma(pk1(pj[10]), Z, pk_fma(pk1(pj[9]), Y, pk1(pj[8]) * X)) + pk1(pj[11] + BBD_EPS);
      v2f r0;
      r0.x = __builtin_amdgcn_rcpf(zi.x);
      r0.y = __builtin_amdgcn_rcpf(zi.y);
      const v2f rz = pk_fma(pk_fma(-zi, r0, pk1(1.0f)), r0, r0);
      const v2f u = qx_ * rz, v = qy_ * rz;
      const v2f gq0 = gix * rz, gq1 = giy * rz;
      const v2f gq2 = -(gix * u + giy * v) * rz;
      const v2f gX = gq0 * pk1(pj[0]) + gq1 * pk1(pj[4]) + gq2 * pk1(pj[8]);
      const v2f gY = gq0 * pk1(pj[1]) + gq1 * pk1(pj[5]) + gq2 * pk1(pj[9]);
      const v2f gZ = gq0 * pk1(pj[2]) + gq1 * pk1(pj[6]) + gq2 * pk1(pj[10]);
      const v2f gd = gX * cx + gY * cy + gZ * cz;
      gdepth[0] += gd.x;
      gdepth[1] += gd.y;
      const v2f gq[3] = {gq0, gq1, gq2};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const v2f a = gq[r] * X, b2 = gq[r] * Y, c2 = gq[r] * Z;
        gP[4 * r + 0] = a.x + a.y;
        gP[4 * r + 1] = b2.x + b2.y;
        gP[4 * r + 2] = c2.x + c2.y;
        gP[4 * r + 3] = gq[r].x + gq[r].y;
      }
    }
    if (cd.kind & FLAG_NO_POSE_GRAD) {
      if (threadIdx.x < 12) gp_out[threadIdx.x] = 0.0f;
    } else {
      float tot[3];
      wave_sum12(gP, tot);
      if ((threadIdx.x & 15) == 15) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 3; ++k) s_red[threadIdx.x >> 6][wave_sum12_index(k, lane)] = tot[k];
      }
    }
    BBD_STAMP(10 + 8 * (c & 1));
    __syncthreads();
    BBD_STAMP(11 + 8 * (c & 1));
    if (!(cd.kind & FLAG_NO_POSE_GRAD) && threadIdx.x < 12) {
      float t = s_red[0][threadIdx.x];
#pragma unroll
      for (int w8 = 1; w8 < NT2 / 64; ++w8) t += s_red[w8][threadIdx.x];
      gp_out[threadIdx.x] = t;
    }
  }

  if (a.ds.grad_wrt_disp) {
    // disparity mode: hand back d loss / d (up-sampled disparity) = d loss / d depth * (-span * depth^2); the
    // bilinear adjoint onto the low-resolution map is bbd_disp_upsample_adjoint (one launch for all scales)
#pragma unroll
    for (int j = 0; j < PPT2; ++j) gdepth[j] *= -dsrc.span * qdepth[j] * qdepth[j];
  }
  if (q_row_ok) {
    float* o = a.grad_depth + sb * hw + qy * W + qx0;
    if (q_vec_ok) {
      *reinterpret_cast<float2*>(o) = make_float2(gdepth[0], gdepth[1]);
    } else {
#pragma unroll
      for (int j = 0; j < PPT2; ++j)
        if (qx0 + j < W) o[j] = gdepth[j];
    }
  }
  BBD_STAMP(20);
  BBD_STAMP_RT(31);
  BBD_STAMP_VAL(29, __builtin_popcount(present));
#undef BBD_PARG
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

// Gather-form adjoint.  Q lanes (1, 4, 16 or 64 by up-sampling factor) share one low-resolution pixel:
// each takes every Q-th of the candidate output pixels and the partial sums are combined with a
// fixed xor-butterfly, so the result is deterministic and small scales still fill the chip.
template <int Q>
__global__ __launch_bounds__(NT) void disp_to_depth_bwd_kernel(const float* __restrict__ disp,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ gdepth,
                                                               float* __restrict__ gdisp, int B, int h, int w,
                                                               int H, int W, float lo, float span) {
  const size_t n = (size_t)B * h * w;
  const size_t gid = ((size_t)blockIdx.x * NT + threadIdx.x) / Q;
  const int sub = threadIdx.x % Q;
  const bool live = gid < n;
  const size_t i = live ? gid : n - 1;
  const int x = (int)(i % w);
  const int y = (int)((i / w) % h);
  const int b = (int)(i / ((size_t)w * h));
  const float* d = disp + (size_t)b * h * w;
  const float* g = gdepth + (size_t)b * H * W;
  const float* dep = depth ? depth + (size_t)b * H * W : nullptr;
  float acc = 0.0f;
  if (h == H && w == W) {
    const float sc = lo + span * d[(size_t)y * w + x];
    acc = g[(size_t)y * W + x] * (-span / (sc * sc));
  } else {
    // output rows/cols whose source index lies in [y-1, y+1): (o+0.5)*h/H - 0.5 in that range,
    // widened by one on each side (the exact membership test is repeated inside the loop)
    const float sy_ = (float)H / (float)h, sx_ = (float)W / (float)w;
    const int oy_lo = max(0, (int)floorf(((float)y - 0.5f) * sy_ - 0.5f) - 1);
    const int oy_hi = min(H - 1, (int)ceilf(((float)y + 1.5f) * sy_ - 0.5f) + 1);
    const int ox_lo = max(0, (int)floorf(((float)x - 0.5f) * sx_ - 0.5f) - 1);
    const int ox_hi = min(W - 1, (int)ceilf(((float)x + 1.5f) * sx_ - 0.5f) + 1);
    const int nx = ox_hi - ox_lo + 1, ncand = (oy_hi - oy_lo + 1) * nx;
    const bool small = (H + W) <= 128;
    for (int k = sub; k < ncand; k += Q) {
      const int oy = oy_lo + k / nx, ox = ox_lo + k % nx;
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      bbd_up_src(oy, h, H, &y0, &y1, &ly0, &ly1);
      bbd_up_src(ox, w, W, &x0, &x1, &lx0, &lx1);
      const float wy = (y0 == y ? ly0 : 0.0f) + (y1 == y ? ly1 : 0.0f);
      const float wx = (x0 == x ? lx0 : 0.0f) + (x1 == x ? lx1 : 0.0f);
      if (wy == 0.0f || wx == 0.0f) continue;
      if (dep != nullptr) {
        // d depth / d disp_up = -span / scaled^2 = -span * depth^2 with the forward's saved depth
        const float dp = dep[(size_t)oy * W + ox];
        acc += g[(size_t)oy * W + ox] * (-span * dp * dp) * wy * wx;
      } else {
        const float sc = lo + span * bbd_up_blend(d[(size_t)y0 * w + x0], d[(size_t)y0 * w + x1],
                                                  d[(size_t)y1 * w + x0], d[(size_t)y1 * w + x1], ly0, ly1, lx0,
                                                  lx1, small);
        acc += g[(size_t)oy * W + ox] * (-span / (sc * sc)) * wy * wx;
      }
    }
  }
#pragma unroll
  for (int off = Q / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (live && sub == 0) gdisp[i] = acc;
}

// Bilinear up-sampling adjoint for ALL reduced scales of a step in one launch (disparity mode of the fused
// backward): grad_up [S,B,H,W] = d loss / d (up-sampled disparity)  ->  grad_disp_s [B,h_s,w_s].  Deterministic.
struct AdjointArgs {
  const float* gup[MAX_SCALES];     // [B,H,W] plane stack of the scale
  float* gdisp[MAX_SCALES];
  int h[MAX_SCALES], w[MAX_SCALES], q[MAX_SCALES];
  int block0[MAX_SCALES + 1];       // first workgroup of each scale
  int n, B, H, W;
};

// The high-resolution indices whose bilinear stencil can touch low-resolution index i (a superset: the weight
// decides), for an up-sampling factor s = out / in.
__device__ __forceinline__ void adjoint_window(int i, float s, int out_size, int* lo, int* hi) {
  *lo = max(0, (int)floorf(((float)i - 0.5f) * s - 0.5f) - 1);
  *hi = min(out_size - 1, (int)ceilf(((float)i + 1.5f) * s - 0.5f) + 1);
}
// weight of high-resolution index o on low-resolution index i (bbd_up_src = the forward's source index / lambdas)
__device__ __forceinline__ float adjoint_weight(int o, int i, int in_size, int out_size) {
  int i0, i1;
  float l0, l1;
  bbd_up_src(o, in_size, out_size, &i0, &i1, &l0, &l1);
  return (i0 == i ? l0 : 0.0f) + (i1 == i ? l1 : 0.0f);
}

// Separable form (the shipped one): the weight of (oy, ox) on (y, x) is wy(oy, y) * wx(ox, x), so a workgroup
// owning an 8x32 low-resolution tile tabulates both weight sets once (<= AJ_TAPS per index), reduces the tile's
// high-resolution rows horizontally into LDS and then vertically: (8 f + 5) * 32 * (2 f + 5) + 256 * (2 f + 5)
// multiply-adds per tile instead of 256 * (2 f + 5)^2 stencil evaluations with two divisions each
// (81 -> see profiles/r02 for the MD2 launch).  Up-sampling factors up to 8 (the reference's scales 1-3).
constexpr int AJ_TY = 8, AJ_TX = 32;
constexpr int AJ_TAPS = 24;                        // >= 2 f + 5 for f <= 8 (+ slack for non-integer factors)
constexpr int AJ_WS = AJ_TAPS + 1;                 // odd row stride of the weight tables: conflict-free columns
constexpr int AJ_ROWS = (AJ_TY + 1) * 8 + 8;       // high-resolution rows under one tile, f <= 8
__global__ __launch_bounds__(NT) void upsample_adjoint_tiled_kernel(AdjointArgs a) {
  __shared__ float s_wx[AJ_TX * AJ_WS], s_wy[AJ_TY * AJ_WS];
  __shared__ int s_xlo[AJ_TX], s_nx[AJ_TX], s_ylo[AJ_TY], s_ny[AJ_TY];
  __shared__ float s_tmp[AJ_ROWS][AJ_TX];
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.block0[k + 1]) ++k;
  const int h = a.h[k], w = a.w[k], H = a.H, W = a.W;
  const int tiles_x = (w + AJ_TX - 1) / AJ_TX, tiles_y = (h + AJ_TY - 1) / AJ_TY;
  int bid = (int)blockIdx.x - a.block0[k];
  const int b = bid / (tiles_x * tiles_y);
  bid -= b * tiles_x * tiles_y;
  const int ty0 = (bid / tiles_x) * AJ_TY, tx0 = (bid % tiles_x) * AJ_TX;
  const float sy_ = (float)H / (float)h, sx_ = (float)W / (float)w;
  const float* g = a.gup[k] + (size_t)b * H * W;
  const int t = (int)threadIdx.x;

  for (int i = t; i < AJ_TX * AJ_TAPS; i += NT) {
    const int xl = i / AJ_TAPS, tap = i - xl * AJ_TAPS, x = tx0 + xl;
    int lo, hi;
    adjoint_window(x, sx_, W, &lo, &hi);
    const bool in = x < w && lo + tap <= hi;
    s_wx[xl * AJ_WS + tap] = in ? adjoint_weight(lo + tap, x, w, W) : 0.0f;
    if (tap == 0) { s_xlo[xl] = lo; s_nx[xl] = x < w ? min(hi - lo + 1, AJ_TAPS) : 0; }
  }
  for (int i = t; i < AJ_TY * AJ_TAPS; i += NT) {
    const int yl = i / AJ_TAPS, tap = i - yl * AJ_TAPS, y = ty0 + yl;
    int lo, hi;
    adjoint_window(y, sy_, H, &lo, &hi);
    const bool in = y < h && lo + tap <= hi;
    s_wy[yl * AJ_WS + tap] = in ? adjoint_weight(lo + tap, y, h, H) : 0.0f;
    if (tap == 0) { s_ylo[yl] = lo; s_ny[yl] = y < h ? min(hi - lo + 1, AJ_TAPS) : 0; }
  }
  __syncthreads();
  const int ylast = min(AJ_TY, h - ty0) - 1;
  const int row0 = s_ylo[0], nrows = min(s_ylo[ylast] + s_ny[ylast] - row0, AJ_ROWS);

  // horizontal: s_tmp[r][xl] = sum_ox wx(ox, x) * g[row0 + r][ox]
  const int xl = t % AJ_TX;
  {
    const int xlo = s_xlo[xl], nx = s_nx[xl];
    for (int r = t / AJ_TX; r < nrows; r += NT / AJ_TX) {
      const float* row = g + (size_t)(row0 + r) * W + xlo;
      float acc = 0.0f;
      for (int tap = 0; tap < nx; ++tap) acc += s_wx[xl * AJ_WS + tap] * row[tap];
      s_tmp[r][xl] = acc;
    }
  }
  __syncthreads();
  // vertical: one low-resolution pixel per thread
  const int yl = t / AJ_TX, y = ty0 + yl, x = tx0 + xl;
  if (y < h && x < w) {
    const int r0 = s_ylo[yl] - row0, ny = s_ny[yl];
    float acc = 0.0f;
    for (int tap = 0; tap < ny && r0 + tap < nrows; ++tap) acc += s_wy[yl * AJ_WS + tap] * s_tmp[r0 + tap][xl];
    a.gdisp[k][((size_t)b * h + y) * w + x] = acc;
  }
}

// Gather form for up-sampling factors above 8 (never the reference's): Q lanes (4 / 16 / 64 by factor) per
// low-resolution pixel evaluate the whole stencil window, fixed xor-butterfly.
__global__ __launch_bounds__(NT) void upsample_adjoint_kernel(AdjointArgs a) {
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.block0[k + 1]) ++k;
  const int h = a.h[k], w = a.w[k], Q = a.q[k], H = a.H, W = a.W;
  const size_t n = (size_t)a.B * h * w;
  const size_t gid = ((size_t)(blockIdx.x - a.block0[k]) * NT + threadIdx.x) / Q;
  const int sub = threadIdx.x % Q;
  const bool live = gid < n;
  const size_t i = live ? gid : n - 1;
  const int x = (int)(i % w);
  const int y = (int)((i / w) % h);
  const int b = (int)(i / ((size_t)w * h));
  const float* g = a.gup[k] + (size_t)b * H * W;
  const float sy_ = (float)H / (float)h, sx_ = (float)W / (float)w;
  int oy_lo, oy_hi, ox_lo, ox_hi;
  adjoint_window(y, sy_, H, &oy_lo, &oy_hi);
  adjoint_window(x, sx_, W, &ox_lo, &ox_hi);
  const int nx = ox_hi - ox_lo + 1, ncand = (oy_hi - oy_lo + 1) * nx;
  float acc = 0.0f;
  for (int c = sub; c < ncand; c += Q) {
    const int oy = oy_lo + c / nx, ox = ox_lo + c % nx;
    const float wy = adjoint_weight(oy, y, h, H), wx = adjoint_weight(ox, x, w, W);
    if (wy == 0.0f || wx == 0.0f) continue;
    acc += g[(size_t)oy * W + ox] * wy * wx;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(acc, off, 64);
    if (off < Q) acc += o;
  }
  if (live && sub == 0) a.gdisp[k][i] = acc;
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
  Cells<LH, LW, LS, 1> cl;
  cl.init(H, W, tc.tx0, tc.ty0);
  stage_image(y + (size_t)item * img, (int)hw, W, cl, s_y);
  stage_image(x + (size_t)item * img, (int)hw, W, cl, s_x);
  __syncthreads();
  int ly, lx0;
  strip_of_thread(&ly, &lx0);
  const int yy = tc.ty0 + ly, xx = tc.tx0 + lx0;
  float mu_y[3][PPT], sg_y[3][PPT];
  strip_ystats(s_y, ly, lx0, mu_y, sg_y);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float xv[3][8], yv[3][8];
    load_window<LS>(s_x[ch], ly, lx0, xv);
    load_window<LS>(s_y[ch], ly, lx0, yv);
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

// ---- backward of the stand-alone layers (closed forms of SURVEY Appendix A2-A5) -------------
// BackprojectDepth: points = depth * (inv_K[:3,:3] . (x,y,1)) | 1  ->  d depth = sum_i g_i * c_i
__global__ __launch_bounds__(NT) void backproject_bwd_kernel(const float* __restrict__ gpoints,
                                                             const float* __restrict__ inv_K,
                                                             float* __restrict__ gdepth, int n, int H, int W) {
  const size_t hw = (size_t)H * W, total = (size_t)n * hw;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
    const int b = (int)(i / hw);
    const size_t p = i - (size_t)b * hw;
    const float fx = (float)(p % W), fy = (float)(p / W);
    const float* k = inv_K + (size_t)b * 16;
    const float* g = gpoints + (size_t)b * 4 * hw + p;
    gdepth[i] = g[0] * bbd_dot3_hom(k, fx, fy) + g[hw] * bbd_dot3_hom(k + 4, fx, fy) +
                g[2 * hw] * bbd_dot3_hom(k + 8, fx, fy);
  }
}

// Project3D: grid = 2*((P.X)_{xy} / ((P.X)_z + eps) / (size-1) - 0.5).  Per pixel: d points = P^T gq;
// per sample: d P = sum_pixels gq (x) X, written as one deterministic partial per workgroup
// (gp_partial [n, gridDim.x, 12]); the caller sums them and forms d T = K[:3,:]^T dP, d K = dP T^T.
constexpr int P3D_BLOCKS = 128;
__global__ __launch_bounds__(NT) void project3d_bwd_kernel(const float* __restrict__ points,
                                                           const float* __restrict__ K, const float* __restrict__ T,
                                                           const float* __restrict__ ggrid,
                                                           float* __restrict__ gpoints,
                                                           float* __restrict__ gp_partial, int H, int W, float eps) {
  __shared__ float s_red[4][12];
  const int b = blockIdx.y;
  const size_t hw = (size_t)H * W;
  float row[28], P[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) row[r] = K[(size_t)b * 16 + r];
#pragma unroll
  for (int r = 0; r < 16; ++r) row[12 + r] = T[(size_t)b * 16 + r];
#pragma unroll
  for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
      float acc = row[i2 * 4 + 0] * row[12 + j2];
      acc = acc + row[i2 * 4 + 1] * row[12 + 4 + j2];
      acc = acc + row[i2 * 4 + 2] * row[12 + 8 + j2];
      acc = acc + row[i2 * 4 + 3] * row[12 + 12 + j2];
      P[i2 * 4 + j2] = acc;
    }
  const float sx = 2.0f / (float)(W - 1), sy = 2.0f / (float)(H - 1);
  float gP[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) gP[k] = 0.0f;
  for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < hw; p += (size_t)gridDim.x * NT) {
    const float* q = points + (size_t)b * 4 * hw + p;
    const float X[4] = {q[0], q[hw], q[2 * hw], q[3 * hw]};
    const float qx = fmaf(P[3], X[3], fmaf(P[2], X[2], fmaf(P[1], X[1], P[0] * X[0])));
    const float qy = fmaf(P[7], X[3], fmaf(P[6], X[2], fmaf(P[5], X[1], P[4] * X[0])));
    const float qz = fmaf(P[11], X[3], fmaf(P[10], X[2], fmaf(P[9], X[1], P[8] * X[0])));
    const float rz = 1.0f / (qz + eps);
    const float u = qx * rz, v = qy * rz;
    const float du = ggrid[((size_t)b * hw + p) * 2 + 0] * sx, dv = ggrid[((size_t)b * hw + p) * 2 + 1] * sy;
    const float gq[3] = {du * rz, dv * rz, -(du * u + dv * v) * rz};
    float* go = gpoints + (size_t)b * 4 * hw + p;
#pragma unroll
    for (int c = 0; c < 4; ++c) go[c * hw] = gq[0] * P[c] + gq[1] * P[4 + c] + gq[2] * P[8 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) gP[r * 4 + c] += gq[r] * X[c];
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float ws = wave_sum63(gP[k]);
    if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6][k] = ws;
  }
  __syncthreads();
  if (threadIdx.x < 12)
    gp_partial[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] =
        ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

// SSIM: d out_c(p) / d x_c(q) = (A + B x(q) + C y(q))(p) / 9 for every window slot of p that lands on q
// (ReflectionPad2d(1): border texels are hit by up to two slots per dimension).  One colour plane at a
// time: coefficient planes of the (TH+2)x(TW+2) loss pixels around the tile, weighted by the upstream
// gradient, then the 3x3 adjoint gather at the thread's own 4 texels.  The SSIM expression is symmetric
// in (x, y), so the gradient w.r.t. the second argument is this kernel with the arguments swapped.
__global__ __launch_bounds__(NT) void ssim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ gout, float* __restrict__ gx,
                                                      int H, int W, int ntiles) {
  __shared__ __attribute__((aligned(16))) float s_x[BPLANE + 8];
  __shared__ __attribute__((aligned(16))) float s_y[BPLANE + 8];
  __shared__ __attribute__((aligned(16))) float s_cf[3][CPLANE];
  const int plane = blockIdx.x / ntiles;                       // one of n*3 colour planes
  const TileCoord tc = decode_tile(blockIdx.x - plane * ntiles, W);
  const size_t hw = (size_t)H * W;
  const float* xp = x + (size_t)plane * hw;
  const float* yp = y + (size_t)plane * hw;
  const float* gp = gout + (size_t)plane * hw;
  Cells<BH, BW, BS, 2> cl;
  cl.init(H, W, tc.tx0, tc.ty0);
#pragma unroll
  for (int k = 0; k < Cells<BH, BW, BS, 2>::N; ++k) {
    s_x[cl.lds[k]] = xp[cl.pix(k, W)];
    s_y[cl.lds[k]] = yp[cl.pix(k, W)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CH * CW; i += NT) {
    const int r = i / CW, c = i - r * CW;
    const int py = tc.ty0 + r - 1, px = tc.tx0 + c - 1;
    float A = 0.0f, Bc = 0.0f, Cc = 0.0f;
    if (py >= 0 && py < H && px >= 0 && px < W) {
      float sx_ = 0.0f, sxx = 0.0f, sxy = 0.0f, sy_ = 0.0f, syy = 0.0f;
#pragma unroll
      for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
          const float xv = s_x[(r + dr) * BS + c + dc], yv = s_y[(r + dr) * BS + c + dc];
          sx_ += xv; sxx += xv * xv; sxy += xv * yv; sy_ += yv; syy += yv * yv;
        }
      float mu_y, sg_y;
      bbd_ystats(sy_, syy, &mu_y, &sg_y);
      bbd_ssim_grad(sx_, sxx, sxy, mu_y, sg_y, &A, &Bc, &Cc);
      const float g = gp[(size_t)py * W + px];
      A *= g; Bc *= g; Cc *= g;
    }
    s_cf[0][r * CS + c] = A;
    s_cf[1][r * CS + c] = Bc;
    s_cf[2][r * CS + c] = Cc;
  }
  __syncthreads();
  int ly, lx0;
  strip_of_thread(&ly, &lx0);
  const int qy = tc.ty0 + ly, qx0 = tc.tx0 + lx0;
  if (qy >= H) return;
  float wy[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int py = qy + d - 1;
    wy[d] = (py >= 0 && py < H) ? (float)bbd_reflect_mult(qy, py, H) : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int qx = qx0 + j;
    if (qx >= W) continue;
    float S[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int dc = 0; dc < 3; ++dc) {
      const int px = qx + dc - 1;
      const float wx = (px >= 0 && px < W) ? (float)bbd_reflect_mult(qx, px, W) : 0.0f;
#pragma unroll
      for (int dr = 0; dr < 3; ++dr) {
        const float wgt = wx * wy[dr];
        const int cell = (ly + dr) * CS + lx0 + j + dc;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) S[pl] = fmaf(wgt, s_cf[pl][cell], S[pl]);
      }
    }
    const float xq = s_x[(ly + 2) * BS + lx0 + j + 2], yq = s_y[(ly + 2) * BS + lx0 + j + 2];
    gx[(size_t)plane * hw + (size_t)qy * W + qx] = (S[0] + xq * S[1] + yq * S[2]) * (1.0f / 9.0f);
  }
}

// ------------------------------------------------------------------------------------------
// Pose table [NP,40] (K[:3,:] | T | inv_K[:3,:3]) -> projection table [NP,24] (P = (K@T)[:3,:] | inv_K).
// One thread per row; bbd_make_proj applies the reference CPU path's rounding order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void pose_expand_kernel(const float* __restrict__ pose, float* __restrict__ proj,
                                                         int n) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  float out[21];
  bbd_make_proj(pose + (size_t)i * BBD_POSE_STRIDE, out);
#pragma unroll
  for (int k = 0; k < 21; ++k) proj[(size_t)i * BBD_PROJ_STRIDE + k] = out[k];
  proj[(size_t)i * BBD_PROJ_STRIDE + 21] = 0.0f;
  proj[(size_t)i * BBD_PROJ_STRIDE + 22] = 0.0f;
  proj[(size_t)i * BBD_PROJ_STRIDE + 23] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// Pose matrix: (axis-angle, translation) -> 4x4, layers.transformation_from_parameters
// (layers.py:25-100) in one launch instead of ~35 element-wise launches; one thread per pose.
// ------------------------------------------------------------------------------------------
struct Rodrigues {
  float x, y, z, ca, sa, C, theta, inv;   // a = v * inv, inv = 1 / (theta + 1e-7)
  float R[9];
};
__device__ __forceinline__ Rodrigues rodrigues(const float* v) {
  Rodrigues r;
  r.theta = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  r.inv = 1.0f / (r.theta + 1e-7f);
  r.x = v[0] * r.inv; r.y = v[1] * r.inv; r.z = v[2] * r.inv;
  r.ca = cosf(r.theta); r.sa = sinf(r.theta); r.C = 1.0f - r.ca;
  const float xs = r.x * r.sa, ys = r.y * r.sa, zs = r.z * r.sa;
  const float xC = r.x * r.C, yC = r.y * r.C, zC = r.z * r.C;
  const float xyC = r.x * yC, yzC = r.y * zC, zxC = r.z * xC;
  r.R[0] = r.x * xC + r.ca; r.R[1] = xyC - zs;        r.R[2] = zxC + ys;
  r.R[3] = xyC + zs;        r.R[4] = r.y * yC + r.ca; r.R[5] = yzC - xs;
  r.R[6] = zxC - ys;        r.R[7] = yzC + xs;        r.R[8] = r.z * zC + r.ca;
  return r;
}

__global__ __launch_bounds__(NT) void pose_matrix_fwd_kernel(const float* __restrict__ aa, const float* __restrict__ tr,
                                                             float* __restrict__ M, int n, int invert,
                                                             const int32_t* __restrict__ invert_rows) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  if (invert_rows != nullptr) invert = invert_rows[i];      // one launch for the poses of both signs of a step
  const Rodrigues r = rodrigues(aa + i * 3);
  const float* t = tr + i * 3;
  float* m = M + i * 16;
  if (!invert) {                       // M = T(t) @ R
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) m[a * 4 + b] = r.R[a * 3 + b];
      m[a * 4 + 3] = t[a];
    }
  } else {                             // M = R^T @ T(-t)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) m[a * 4 + b] = r.R[b * 3 + a];
      float acc = r.R[0 * 3 + a] * (-t[0]);
      acc = acc + r.R[1 * 3 + a] * (-t[1]);
      acc = acc + r.R[2 * 3 + a] * (-t[2]);
      m[a * 4 + 3] = acc;
    }
  }
  m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
}

__global__ __launch_bounds__(NT) void pose_matrix_bwd_kernel(const float* __restrict__ aa, const float* __restrict__ tr,
                                                             const float* __restrict__ gM, float* __restrict__ gaa,
                                                             float* __restrict__ gtr, int n, int invert,
                                                             const int32_t* __restrict__ invert_rows) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  if (invert_rows != nullptr) invert = invert_rows[i];
  const float* v = aa + i * 3;
  const float* t = tr + i * 3;
  const float* g = gM + i * 16;
  const Rodrigues r = rodrigues(v);
  float G[9];                          // dL/dR
  float gt[3];
  if (!invert) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) G[a * 3 + b] = g[a * 4 + b];
      gt[a] = g[a * 4 + 3];
    }
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) G[a * 3 + b] = g[b * 4 + a];      // M[:3,:3] = R^T
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                    // M[a,3] = -sum_k R[k,a] t[k]
      gt[k] = -(g[0 * 4 + 3] * r.R[k * 3 + 0] + g[1 * 4 + 3] * r.R[k * 3 + 1] + g[2 * 4 + 3] * r.R[k * 3 + 2]);
#pragma unroll
      for (int a = 0; a < 3; ++a) G[k * 3 + a] += -g[a * 4 + 3] * t[k];
    }
  }
  const float a_[3] = {r.x, r.y, r.z};
  float quad = 0.0f, trace = G[0] + G[4] + G[8];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) quad += G[a * 3 + b] * a_[a] * a_[b];
  const float gca = trace - quad;                                    // C = 1 - ca
  const float gsa = -r.z * G[1] + r.y * G[2] + r.z * G[3] - r.x * G[5] - r.y * G[6] + r.x * G[7];
  float ga[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float sym = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) sym += (G[k * 3 + j] + G[j * 3 + k]) * a_[j];
    ga[k] = r.C * sym;
  }
  ga[0] += r.sa * (G[7] - G[5]);
  ga[1] += r.sa * (G[2] - G[6]);
  ga[2] += r.sa * (G[3] - G[1]);
  float gtheta = -gca * r.sa + gsa * r.ca;
  const float gav = ga[0] * v[0] + ga[1] * v[1] + ga[2] * v[2];
  gtheta -= gav * r.inv * r.inv;                                     // a = v / (theta + eps)
  const float s = r.theta > 0.0f ? gtheta / r.theta : 0.0f;          // d theta / d v = v / theta
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gaa[i * 3 + k] = ga[k] * r.inv + s * v[k];
    gtr[i * 3 + k] = gt[k];
  }
}

// ------------------------------------------------------------------------------------------
// Edge-aware smoothness of the mean-normalised disparity (layers.py:203-216, trainer.py:560-564).
// Deterministic: per-(sample, row-chunk) partial sums, fixed reduction order.
// ------------------------------------------------------------------------------------------
constexpr int SM_CHUNKS = 64;   // B x 64 workgroups per launch: enough to cover 256 CUs at B = 12

__device__ __forceinline__ float block_sum_all(float v, float* s_red4) {   // every thread gets the total
  const float w = wave_sum63(v);
  if ((threadIdx.x & 63) == 63) s_red4[threadIdx.x >> 6] = w;
  __syncthreads();
  const float t = ((s_red4[0] + s_red4[1]) + s_red4[2]) + s_red4[3];
  __syncthreads();
  return t;
}

// Every scale of a step in ONE launch per pass (round 4: MD2's four scales took 16 launches of 3-10 us each, and the eager
// hot path is launch-bound): grid = (B * SM_CHUNKS, S), blockIdx.y picks the scale's pointers and sizes.  The single-scale
// entry points run the same kernels with S = 1, so both forms reduce in the same fixed order.
struct SmoothArgs {
  const float* disp[MAX_SCALES];
  const float* img[MAX_SCALES];
  float* grad[MAX_SCALES];
  int h[MAX_SCALES], w[MAX_SCALES];
  float* mean;            // [S][B][SM_CHUNKS] partial sums of disp
  float* sums;            // [S][B][SM_CHUNKS][2]
  float* dots;            // [S][B][SM_CHUNKS]
  const float* gscale;    // [S]
  int B;
};

// per-(sample, chunk) partial sums of disp; the mean is finished by sample_mean() in the consumers
__global__ __launch_bounds__(NT) void smooth_mean_kernel(SmoothArgs a) {
  __shared__ float s_red4[4];
  const int s = blockIdx.y, hw = a.h[s] * a.w[s];
  const int b = blockIdx.x / SM_CHUNKS, chunk = blockIdx.x - b * SM_CHUNKS;
  const float* d = a.disp[s] + (size_t)b * hw;
  const int per = (hw + SM_CHUNKS - 1) / SM_CHUNKS;
  const int i0 = chunk * per, i1 = min(hw, i0 + per);
  float acc = 0.0f;
  for (int i = i0 + threadIdx.x; i < i1; i += NT) acc += d[i];
  const float tot = block_sum_all(acc, s_red4);
  if (threadIdx.x == 0) a.mean[(size_t)s * a.B * SM_CHUNKS + blockIdx.x] = tot;
}

__device__ __forceinline__ float sample_mean(const float* __restrict__ psum, int b, int hw) {
  float t = 0.0f;
  for (int k = 0; k < SM_CHUNKS; ++k) t += psum[b * SM_CHUNKS + k];
  return t / (float)hw;
}

__device__ __forceinline__ float edge_weight(const float* img, int hw, int i0, int i1) {
  const float g = fabsf(img[i0] - img[i1]) + fabsf(img[i0 + hw] - img[i1 + hw]) +
                  fabsf(img[i0 + 2 * hw] - img[i1 + 2 * hw]);
  return __expf(-g * (1.0f / 3.0f));
}

__global__ __launch_bounds__(NT) void smooth_fwd_kernel(SmoothArgs a) {
  __shared__ float s_red4[4];
  const int s = blockIdx.y, h = a.h[s], w = a.w[s];
  const int b = blockIdx.x / SM_CHUNKS, chunk = blockIdx.x - b * SM_CHUNKS;
  const int hw = h * w;
  const float* d = a.disp[s] + (size_t)b * hw;
  const float* im = a.img[s] + (size_t)b * 3 * hw;
  const float inv = 1.0f / (sample_mean(a.mean + (size_t)s * a.B * SM_CHUNKS, b, hw) + 1e-7f);
  const int rows = (h + SM_CHUNKS - 1) / SM_CHUNKS;
  const int y0 = chunk * rows, y1 = min(h, y0 + rows);
  float ax = 0.0f, ay = 0.0f;
  for (int i = y0 * w + threadIdx.x; i < y1 * w; i += NT) {
    const int y = i / w, x = i - y * w;
    const float n0 = d[i] * inv;
    if (x < w - 1) ax += fabsf(n0 - d[i + 1] * inv) * edge_weight(im, hw, i, i + 1);
    if (y < h - 1) ay += fabsf(n0 - d[i + w] * inv) * edge_weight(im, hw, i, i + w);
  }
  const float tx = block_sum_all(ax, s_red4);
  const float ty = block_sum_all(ay, s_red4);
  if (threadIdx.x == 0) {
    float* o = a.sums + ((size_t)s * a.B * SM_CHUNKS + blockIdx.x) * 2;
    o[0] = tx;
    o[1] = ty;
  }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// pass 1: gn = d loss / d norm (stored into grad), and per-chunk partial of sum(gn * disp)
__global__ __launch_bounds__(NT) void smooth_bwd1_kernel(SmoothArgs a) {
  __shared__ float s_red4[4];
  const int s = blockIdx.y, h = a.h[s], w = a.w[s], B = a.B;
  const int b = blockIdx.x / SM_CHUNKS, chunk = blockIdx.x - b * SM_CHUNKS;
  const int hw = h * w;
  const float* d = a.disp[s] + (size_t)b * hw;
  const float* im = a.img[s] + (size_t)b * 3 * hw;
  float* gout = a.grad[s] + (size_t)b * hw;
  const float inv = 1.0f / (sample_mean(a.mean + (size_t)s * B * SM_CHUNKS, b, hw) + 1e-7f);
  const float g = a.gscale[s];
  const float gxs = g / ((float)B * (float)h * (float)(w - 1));
  const float gys = g / ((float)B * (float)(h - 1) * (float)w);
  const int rows = (h + SM_CHUNKS - 1) / SM_CHUNKS;
  const int y0 = chunk * rows, y1 = min(h, y0 + rows);
  float dot = 0.0f;
  for (int i = y0 * w + threadIdx.x; i < y1 * w; i += NT) {
    const int y = i / w, x = i - y * w;
    const float n0 = d[i] * inv;
    float gn = 0.0f;
    if (x < w - 1) gn += gxs * sgn(n0 - d[i + 1] * inv) * edge_weight(im, hw, i, i + 1);
    if (x > 0) gn -= gxs * sgn(d[i - 1] * inv - n0) * edge_weight(im, hw, i - 1, i);
    if (y < h - 1) gn += gys * sgn(n0 - d[i + w] * inv) * edge_weight(im, hw, i, i + w);
    if (y > 0) gn -= gys * sgn(d[i - w] * inv - n0) * edge_weight(im, hw, i - w, i);
    gout[i] = gn;
    dot += gn * d[i];
  }
  const float t = block_sum_all(dot, s_red4);
  if (threadIdx.x == 0) a.dots[(size_t)s * B * SM_CHUNKS + blockIdx.x] = t;
}

// pass 2: grad_disp = gn / (m+eps) - sum(gn*disp) / (N (m+eps)^2)
__global__ __launch_bounds__(NT) void smooth_bwd2_kernel(SmoothArgs a) {
  const int s = blockIdx.y, h = a.h[s], w = a.w[s];
  const int b = blockIdx.x / SM_CHUNKS, chunk = blockIdx.x - b * SM_CHUNKS;
  const int hw = h * w;
  const float* dots = a.dots + (size_t)s * a.B * SM_CHUNKS;
  float dot = 0.0f;
  for (int k = 0; k < SM_CHUNKS; ++k) dot += dots[b * SM_CHUNKS + k];
  const float inv = 1.0f / (sample_mean(a.mean + (size_t)s * a.B * SM_CHUNKS, b, hw) + 1e-7f);
  const float sub = dot * inv * inv / (float)hw;
  float* gout = a.grad[s] + (size_t)b * hw;
  const int rows = (h + SM_CHUNKS - 1) / SM_CHUNKS;
  const int y0 = chunk * rows, y1 = min(h, y0 + rows);
  for (int i = y0 * w + threadIdx.x; i < y1 * w; i += NT) gout[i] = gout[i] * inv - sub;
}

// Self-test of the cheap exact divisions (bbd_math.h) against hipcc's IEEE `/`.
__device__ __forceinline__ uint32_t xorshift32(uint32_t& st) {
  st ^= st << 13; st ^= st >> 17; st ^= st << 5;
  return st;
}
__device__ __forceinline__ float random_float(uint32_t& st, int emin, int emax) {
  const uint32_t m = xorshift32(st) & 0x807fffffu;                       // sign + mantissa
  const uint32_t e = (uint32_t)(emin + (int)(xorshift32(st) % (uint32_t)(emax - emin + 1)));
  return __uint_as_float(m | (e << 23));
}
__global__ __launch_bounds__(NT) void selftest_div_kernel(uint32_t seed, int iters, int* __restrict__ bad) {
  uint32_t st = seed ^ (blockIdx.x * 2654435761u + threadIdx.x * 40503u + 1u);
  int mism = 0;
  const float consts[6] = {639.0f, 191.0f, 63.0f, 31.0f, 1279.0f, 383.0f};
  for (int i = 0; i < iters; ++i) {
    // operands across the projection / SSIM ranges, plus a slice of extreme exponents that must
    // take the fallback path
    const bool extreme = (i & 63) == 0;
    const float n0 = random_float(st, extreme ? 1 : 90, extreme ? 254 : 150);
    const float n1 = random_float(st, extreme ? 1 : 90, extreme ? 254 : 150);
    const float d = random_float(st, extreme ? 1 : 100, extreme ? 254 : 140);
    float q0, q1;
    bbd_div2(n0, n1, d, &q0, &q1);
    const float r0 = n0 / d, r1 = n1 / d;
    mism += (__float_as_uint(q0) != __float_as_uint(r0)) && !(q0 != q0 && r0 != r0);
    mism += (__float_as_uint(q1) != __float_as_uint(r1)) && !(q1 != q1 && r1 != r1);
    const float q2 = bbd_div(n1, n0), r2 = n1 / n0;
    mism += (__float_as_uint(q2) != __float_as_uint(r2)) && !(q2 != q2 && r2 != r2);
    const float cd = consts[i % 6];
    const float q3 = bbd_div_const(n0, cd, 1.0f / cd), r3 = n0 / cd;
    mism += (__float_as_uint(q3) != __float_as_uint(r3)) && !(q3 != q3 && r3 != r3);
    const float q4 = bbd_div9(n1), r4 = n1 / 9.0f;
    mism += (__float_as_uint(q4) != __float_as_uint(r4)) && (fabsf(n1) < 1e30f) && (fabsf(n1) > 1e-30f);
  }
  if (mism) atomicAdd(bad, mism);
}
// variant that reports per-category counts: bad[0..4] = div2.q0, div2.q1, div, div_const, div9
__global__ __launch_bounds__(NT) void selftest_div_detail_kernel(uint32_t seed, int iters, int* __restrict__ bad) {
  uint32_t st = seed ^ (blockIdx.x * 2654435761u + threadIdx.x * 40503u + 1u);
  const float consts[6] = {639.0f, 191.0f, 63.0f, 31.0f, 1279.0f, 383.0f};
  for (int i = 0; i < iters; ++i) {
    const bool extreme = (i & 63) == 0;
    const float n0 = random_float(st, extreme ? 1 : 90, extreme ? 254 : 150);
    const float n1 = random_float(st, extreme ? 1 : 90, extreme ? 254 : 150);
    const float d = random_float(st, extreme ? 1 : 100, extreme ? 254 : 140);
    float q0, q1;
    bbd_div2(n0, n1, d, &q0, &q1);
    const float r0 = n0 / d, r1 = n1 / d;
    if ((__float_as_uint(q0) != __float_as_uint(r0)) && !(q0 != q0 && r0 != r0)) atomicAdd(bad + (extreme ? 5 : 0), 1);
    if ((__float_as_uint(q1) != __float_as_uint(r1)) && !(q1 != q1 && r1 != r1)) atomicAdd(bad + (extreme ? 6 : 1), 1);
    const float q2 = bbd_div(n1, n0), r2 = n1 / n0;
    if ((__float_as_uint(q2) != __float_as_uint(r2)) && !(q2 != q2 && r2 != r2)) atomicAdd(bad + (extreme ? 7 : 2), 1);
    const float cd = consts[i % 6];
    const float q3 = bbd_div_const(n0, cd, 1.0f / cd), r3 = n0 / cd;
    if ((__float_as_uint(q3) != __float_as_uint(r3)) && !(q3 != q3 && r3 != r3)) atomicAdd(bad + (extreme ? 8 : 3), 1);
    const float q4 = bbd_div9(n1), r4 = n1 / 9.0f;
    if ((__float_as_uint(q4) != __float_as_uint(r4)) && (fabsf(n1) < 1e30f) && (fabsf(n1) > 1e-30f)) atomicAdd(bad + (extreme ? 9 : 4), 1);
  }
}

int fill_frames(const void* const* frames, FramePtrs* out) {
  if (frames == nullptr) return BBD_E_BADARG;
  for (int i = 0; i < BBD_MAX_FRAME_SLOTS; ++i) out->base[i] = static_cast<const float*>(frames[i]);
  return 0;
}

// XCD-aware work order without a table (xcd_work_item: a contiguous range of work items per XCD): the identity pre-pass
// (always: in-step fabric traffic 124 -> 65 MB and -6 % time for MD2, profiles/r04/work_order_ab.txt) and the fused launches
// when the caller passes no work-item table and there is one scale (round 3: profiles/r03/xcd_remap_ab.txt; with several
// scales a range split hands whole scales of different cost to different XCDs).  BBD_XCD_REMAP=0 / 1 forces it.
// (Both knobs are read only under BBD_EXPERIMENT=1 - the A/B scripts under tools/ and the test that forces every forward
// form set it: a stray variable in a user's environment cannot put the library into a configuration no test covers.)
int experiment_knob(const char* name) {
  const char* on = getenv("BBD_EXPERIMENT");
  if (on == nullptr || on[0] != '1') return -1;
  const char* e = getenv(name);
  return e == nullptr ? -1 : (e[0] - '0');
}
int xcd_remap_enabled(int S) {
  static const int forced = experiment_knob("BBD_XCD_REMAP");
  return forced >= 0 ? forced : (S == 1);
}
int fused_fwd_form(int S, int B, int NP, int ntiles) {
  static const int forced = experiment_knob("BBD_FWD_FORM");
  if (forced >= 0 && forced <= 2) return forced;
  if (NP <= 4 * B) return FWD_RESTAT;
  return (long)S * B * ntiles >= 4 * 1024 ? FWD_HELD : FWD_DOUBLE;
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
int bbd_num_tiles_fwd(int H, int W) {      /* tiles of the fused FORWARD launch: sizes partial [S, B, tiles] */
  return bbd_num_tiles(H, W);
}
int bbd_num_tiles_bwd(int H, int W) { return ((H + TH - 1) / TH) * ((W + TW2 - 1) / TW2); }

int bbd_fused_work_items(int B, int S, int H, int W, int backward, const int32_t* sample_order, int32_t* out) {
  if (B <= 0 || S <= 0 || H < 3 || W < 3 || !out || B > (1 << WORK_B_BITS) || S > (1 << WORK_S_BITS)) return BBD_E_BADARG;
  const int tw = backward ? TW2 : TW;
  const int tiles_x = (W + tw - 1) / tw, ntiles = tiles_x * ((H + TH - 1) / TH);
  if (ntiles >= (1 << (32 - WORK_B_BITS - WORK_S_BITS)) || W > 0xffff || H > 0xffff) return BBD_E_BADARG;
  const int n = B * S * ntiles, q = ntiles >> 3, r = ntiles & 7, bs = B * S;
  const int head = bs * r * (q + 1);            // the first r slabs hold q + 1 tiles, the others q
  if (sample_order != nullptr) {
    // a permutation of 0..B-1, nothing less: a repeated entry would run one sample's tiles twice and leave another's
    // outputs (allocated uninitialised by the caller) untouched
    uint64_t seen[(1 << WORK_B_BITS) / 64] = {0};
    for (int k = 0; k < B; ++k) {
      const int b = sample_order[k];
      if (b < 0 || b >= B || ((seen[b >> 6] >> (b & 63)) & 1)) return BBD_E_BADARG;
      seen[b >> 6] |= uint64_t(1) << (b & 63);
    }
  }
  for (int i = 0; i < n; ++i) {
    // hardware block i runs on XCD i & 7 (round-robin dispatch); XCD x walks the contiguous range [x n / 8, (x + 1) n / 8)
    // of the slab-major virtual order v: slab -> sample (caller's order) -> scale -> tile of the slab
    const int nq = n >> 3, nr = n & 7, x8 = i & 7;
    const int v = (x8 < nr ? x8 * (nq + 1) : nr * (nq + 1) + (x8 - nr) * nq) + (i >> 3);
    const int x = v < head ? v / (bs * (q + 1)) : r + (v - head) / (bs * q);
    const int t0 = x * q + (x < r ? x : r), cs = q + (x < r ? 1 : 0);
    int w = v - bs * t0;
    const int rank = w / (S * cs);
    w -= rank * S * cs;
    const int s = w / cs, tile = t0 + (w - s * cs);
    const int b = sample_order != nullptr ? sample_order[rank] : rank;
    if (b < 0 || b >= B) return BBD_E_BADARG;
    out[2 * i] = b | (s << WORK_B_BITS) | (tile << (WORK_B_BITS + WORK_S_BITS));
    out[2 * i + 1] = ((tile % tiles_x) * tw) | (((tile / tiles_x) * TH) << 16);
  }
  return 0;
}

int bbd_identity_loss_fwd(const void* const* frames, const float* target, const int32_t* items, int NI,
                          float* ident, int H, int W, int no_ssim, void* stream) {
  if (!target || !items || !ident || NI < 0 || H < 3 || W < 3) return BBD_E_BADARG;
  if (NI == 0) return 0;
  FramePtrs fp;
  if (fill_frames(frames, &fp)) return BBD_E_BADARG;
  const int ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(identity_loss_kernel, dim3((unsigned)(NI * ntiles)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), fp, target, items, ident, H, W, ntiles, no_ssim,
                     xcd_remap_enabled(1));
  return launch_status();
}

int bbd_identity_loss_grouped_fwd(const void* const* frames, const float* target, const int32_t* items,
                                  const int32_t* group_off, int G, float* ident, int H, int W, int no_ssim, void* stream);

static int fill_disp(const void* const* disp, const int32_t* disp_hw, double min_depth, double max_depth, int S, int H,
                     int W, DispSrc* ds) {
  for (int i = 0; i < MAX_SCALES; ++i) { ds->disp[i] = nullptr; ds->h[i] = H; ds->w[i] = W; }
  ds->lo = 0.0f; ds->span = 0.0f; ds->grad_wrt_disp = 0;
  if (disp == nullptr) return 0;
  if (disp_hw == nullptr || S > MAX_SCALES || min_depth <= 0.0 || max_depth <= min_depth) return BBD_E_BADARG;
  for (int i = 0; i < S; ++i) {
    ds->disp[i] = static_cast<const float*>(disp[i]);
    ds->h[i] = disp_hw[2 * i]; ds->w[i] = disp_hw[2 * i + 1];
    if (!ds->disp[i] || ds->h[i] <= 0 || ds->w[i] <= 0 || ds->h[i] > H || ds->w[i] > W) return BBD_E_BADARG;
  }
  // layers.py:18-20 evaluates these in Python doubles before they meet the fp32 tensor
  ds->lo = (float)(1.0 / max_depth);
  ds->span = (float)(1.0 / min_depth - 1.0 / max_depth);
  ds->grad_wrt_disp = 1;
  return 0;
}

static int launch_fused_fwd(const void* const* frames, const float* target, const float* depth, const void* const* disp,
                            const int32_t* disp_hw, double min_depth, double max_depth, const float* proj,
                            const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                            const int32_t* work, float* min_loss, uint8_t* argmin, float* partial, float* warped,
                            float* depth_out, int S, int B, int NP, int H, int W, int no_ssim, void* stream) {
  if (!target || (!depth && !disp) || !cand || !ncand || !min_loss || !argmin || !partial) return BBD_E_BADARG;
  if (S <= 0 || B <= 0 || H < 3 || W < 3 || NP < 0) return BBD_E_BADARG;
  FwdArgs a;
#ifdef BBD_STAMPS
  a.stamps = g_stamps_host_ptr;
#else
  a.stamps = nullptr;
#endif
  if (fill_frames(frames, &a.frames)) return BBD_E_BADARG;
  if (fill_disp(disp, disp_hw, min_depth, max_depth, S, H, W, &a.ds)) return BBD_E_BADARG;
  a.target = target; a.depth = depth; a.pose = proj; a.ident = ident; a.noise = noise;
  a.cand = cand; a.ncand = ncand; a.min_loss = min_loss; a.argmin = argmin; a.partial = partial;
  a.warped = warped; a.depth_out = depth_out; a.S = S; a.B = B; a.NP = NP; a.dm = bbd_dims(H, W); a.no_ssim = no_ssim;
  a.ntiles = bbd_num_tiles_fwd(H, W);
  a.remap = xcd_remap_enabled(S);
  a.work = work;
  // (a paired-candidate / packed-SSIM form of this kernel was built and measured slower - profiles/r03/fwdp_ab.txt; its
  // source is kept under tools/experiments/paired_packed_forward.hip.txt)
  // which form (see warp_ssim_min_fwd_kernel): few warp candidates per sample (pose-table rows per sample <= 4: MD2's 2) ->
  // RESTAT; many candidates over several scales (>= 4 rounds of the 1 024 four-wave slots) -> HELD; else the three-wave form
  const int form = fused_fwd_form(S, B, NP, a.ntiles);
  const dim3 grid((unsigned)(S * B * a.ntiles));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define BBD_LAUNCH_FWD(P, F) hipLaunchKernelGGL((warp_ssim_min_fwd_kernel<P, F>), grid, dim3(NT), 0, st, a)
  if (depth != nullptr) {
    if (form == FWD_RESTAT) BBD_LAUNCH_FWD(true, FWD_RESTAT);
    else if (form == FWD_HELD) BBD_LAUNCH_FWD(true, FWD_HELD);
    else BBD_LAUNCH_FWD(true, FWD_DOUBLE);
  } else {
    if (form == FWD_RESTAT) BBD_LAUNCH_FWD(false, FWD_RESTAT);
    else if (form == FWD_HELD) BBD_LAUNCH_FWD(false, FWD_HELD);
    else BBD_LAUNCH_FWD(false, FWD_DOUBLE);
  }
#undef BBD_LAUNCH_FWD
  return launch_status();
}

static int launch_fused_bwd(const void* const* frames, const float* target, const float* depth, const void* const* disp,
                            const int32_t* disp_hw, double min_depth, double max_depth, const float* proj,
                            const bbd_cand_t* cand, const int32_t* ncand, const int32_t* work, const uint8_t* argmin,
                            const float* gscale, float* grad_depth, float* grad_proj, int S, int B, int NP, int H, int W,
                            int no_ssim, void* stream) {
  if (!target || (!depth && !disp) || !cand || !ncand || !argmin || !gscale || !grad_depth || !grad_proj) return BBD_E_BADARG;
  if (S <= 0 || B <= 0 || H < 3 || W < 3 || NP < 0) return BBD_E_BADARG;
  BwdArgs a;
#ifdef BBD_STAMPS
  a.stamps = g_stamps_host_ptr;
#else
  a.stamps = nullptr;
#endif
  if (fill_frames(frames, &a.frames)) return BBD_E_BADARG;
  if (fill_disp(disp, disp_hw, min_depth, max_depth, S, H, W, &a.ds)) return BBD_E_BADARG;
  a.target = target; a.depth = depth; a.pose = proj; a.cand = cand; a.ncand = ncand; a.argmin = argmin;
  a.gscale = gscale; a.grad_depth = grad_depth; a.grad_proj = grad_proj;
  a.S = S; a.B = B; a.NP = NP; a.dm = bbd_dims(H, W); a.no_ssim = no_ssim;
  a.ntiles = bbd_num_tiles_bwd(H, W);
  a.remap = xcd_remap_enabled(S);
  a.work = work;
  // (a sparse-item form of this kernel - per-candidate winner lists, scatter instead of the dense phases - was built and
  // measured slower inside the training step: profiles/r03/bwd3_*.txt, tools/experiments/sparse_item_backward.hip.txt)
  const dim3 grid((unsigned)(S * B * a.ntiles));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (depth != nullptr) hipLaunchKernelGGL(warp_ssim_min_bwd9_kernel<true>, grid, dim3(NT2), 0, st, a);
  else hipLaunchKernelGGL(warp_ssim_min_bwd9_kernel<false>, grid, dim3(NT2), 0, st, a);
  return launch_status();
}

int bbd_warp_ssim_min_fwd(const void* const* frames, const float* target, const float* depth, const float* proj,
                          const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                          float* min_loss, uint8_t* argmin, float* partial, float* warped, int S, int B, int NP,
                          int H, int W, int no_ssim, void* stream) {
  if (!depth) return BBD_E_BADARG;
  return launch_fused_fwd(frames, target, depth, nullptr, nullptr, 0.0, 0.0, proj, ident, noise, cand, ncand, nullptr,
                          min_loss, argmin, partial, warped, nullptr, S, B, NP, H, W, no_ssim, stream);
}

int bbd_warp_ssim_min_bwd(const void* const* frames, const float* target, const float* depth, const float* proj,
                          const bbd_cand_t* cand, const int32_t* ncand, const uint8_t* argmin, const float* gscale,
                          float* grad_depth, float* grad_proj, int S, int B, int NP, int H, int W, int no_ssim,
                          void* stream) {
  if (!depth) return BBD_E_BADARG;
  return launch_fused_bwd(frames, target, depth, nullptr, nullptr, 0.0, 0.0, proj, cand, ncand, nullptr, argmin, gscale,
                          grad_depth, grad_proj, S, B, NP, H, W, no_ssim, stream);
}

int bbd_warp_ssim_min_disp_fwd(const void* const* frames, const float* target, const void* const* disp,
                               const int32_t* disp_hw, double min_depth, double max_depth, const float* proj,
                               const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                               const int32_t* work_items, float* min_loss, uint8_t* argmin, float* partial, float* warped,
                               float* depth_out, int S, int B, int NP, int H, int W, int no_ssim, void* stream) {
  if (!disp) return BBD_E_BADARG;
  return launch_fused_fwd(frames, target, nullptr, disp, disp_hw, min_depth, max_depth, proj, ident, noise, cand, ncand,
                          work_items, min_loss, argmin, partial, warped, depth_out, S, B, NP, H, W, no_ssim, stream);
}

int bbd_warp_ssim_min_disp_bwd(const void* const* frames, const float* target, const void* const* disp,
                               const int32_t* disp_hw, double min_depth, double max_depth, const float* depth,
                               const float* proj, const bbd_cand_t* cand, const int32_t* ncand,
                               const int32_t* work_items, const uint8_t* argmin, const float* gscale, float* grad_up,
                               float* grad_proj, int S, int B, int NP, int H, int W, int no_ssim, void* stream) {
  if (!disp) return BBD_E_BADARG;
  return launch_fused_bwd(frames, target, depth, disp, disp_hw, min_depth, max_depth, proj, cand, ncand, work_items, argmin,
                          gscale, grad_up, grad_proj, S, B, NP, H, W, no_ssim, stream);
}

int bbd_identity_loss_grouped_fwd(const void* const* frames, const float* target, const int32_t* items,
                                  const int32_t* group_off, int G, float* ident, int H, int W, int no_ssim, void* stream) {
  if (!target || !items || !group_off || !ident || G < 0 || H < 3 || W < 3) return BBD_E_BADARG;
  if (G == 0) return 0;
  FramePtrs fp;
  if (fill_frames(frames, &fp)) return BBD_E_BADARG;
  const int ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(identity_loss_grouped_kernel, dim3((unsigned)(G * ntiles)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), fp, target, items, group_off, ident, H, W, ntiles, no_ssim,
                     xcd_remap_enabled(1));
  return launch_status();
}

int bbd_disp_upsample_adjoint(const void* const* grad_up, const int32_t* disp_hw, void* const* grad_disp, int n, int B,
                              int H, int W, void* stream) {
  if (!grad_up || !disp_hw || !grad_disp || n < 0 || n > MAX_SCALES || B <= 0 || H <= 0 || W <= 0) return BBD_E_BADARG;
  if (n == 0) return 0;
  AdjointArgs a;
  a.n = n; a.B = B; a.H = H; a.W = W;
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool tiled = true;
  for (int i = 0; i < n; ++i) {
    a.gup[i] = static_cast<const float*>(grad_up[i]);
    a.gdisp[i] = static_cast<float*>(grad_disp[i]);
    a.h[i] = disp_hw[2 * i]; a.w[i] = disp_hw[2 * i + 1];
    if (!a.gup[i] || !a.gdisp[i] || a.h[i] <= 0 || a.w[i] <= 0 || a.h[i] > H || a.w[i] > W) return BBD_E_BADARG;
    if ((long long)H > 8LL * a.h[i] || (long long)W > 8LL * a.w[i]) tiled = false;
  }
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const int f = (H + a.h[i] - 1) / a.h[i];
    a.q[i] = f <= 1 ? 1 : (f <= 2 ? 4 : (f <= 4 ? 16 : 64));
    a.block0[i] = blocks;
    if (tiled) blocks += B * ((a.h[i] + AJ_TY - 1) / AJ_TY) * ((a.w[i] + AJ_TX - 1) / AJ_TX);
    else blocks += (int)(((size_t)B * a.h[i] * a.w[i] * a.q[i] + NT - 1) / NT);
  }
  a.block0[n] = blocks;
  if (tiled) hipLaunchKernelGGL(upsample_adjoint_tiled_kernel, dim3((unsigned)blocks), dim3(NT), 0, st, a);
  else hipLaunchKernelGGL(upsample_adjoint_kernel, dim3((unsigned)blocks), dim3(NT), 0, st, a);
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

int bbd_disp_to_depth_bwd(const float* disp, const float* depth, const float* grad_depth, float* grad_disp, int B,
                          int h, int w, int H, int W, double min_depth, double max_depth, void* stream) {
  if (!disp || !grad_depth || !grad_disp || B <= 0 || h <= 0 || w <= 0 || H < h || W < w) return BBD_E_BADARG;
  const float lo = (float)(1.0 / max_depth), span = (float)(1.0 / min_depth - 1.0 / max_depth);
  const size_t n = (size_t)B * h * w;
  const int f = (H + h - 1) / h;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define BBD_D2D_BWD(Q)                                                                                         \
  hipLaunchKernelGGL(disp_to_depth_bwd_kernel<Q>, dim3((unsigned)((n * Q + NT - 1) / NT)), dim3(NT), 0, st, disp, \
                     depth, grad_depth, grad_disp, B, h, w, H, W, lo, span)
  if (f <= 1) BBD_D2D_BWD(1);
  else if (f <= 2) BBD_D2D_BWD(4);
  else if (f <= 4) BBD_D2D_BWD(16);
  else BBD_D2D_BWD(64);
#undef BBD_D2D_BWD
  return launch_status();
}

int bbd_pose_expand(const float* pose, float* proj, int NP, void* stream) {
  if (!pose || !proj || NP < 0) return BBD_E_BADARG;
  if (NP == 0) return 0;
  hipLaunchKernelGGL(pose_expand_kernel, dim3((unsigned)((NP + NT - 1) / NT)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), pose, proj, NP);
  return launch_status();
}

int bbd_pose_matrix_fwd(const float* axisangle, const float* translation, float* M, int n, int invert,
                        const int32_t* invert_rows, void* stream) {
  if (!axisangle || !translation || !M || n < 0) return BBD_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(pose_matrix_fwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), axisangle, translation, M, n, invert, invert_rows);
  return launch_status();
}

int bbd_pose_matrix_bwd(const float* axisangle, const float* translation, const float* grad_M, float* grad_axisangle,
                        float* grad_translation, int n, int invert, const int32_t* invert_rows, void* stream) {
  if (!axisangle || !translation || !grad_M || !grad_axisangle || !grad_translation || n < 0) return BBD_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(pose_matrix_bwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), axisangle, translation, grad_M, grad_axisangle,
                     grad_translation, n, invert, invert_rows);
  return launch_status();
}

int bbd_smooth_chunks(void) { return SM_CHUNKS; }

static int launch_smooth_fwd(const SmoothArgs& a, int S, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(smooth_mean_kernel, dim3((unsigned)(a.B * SM_CHUNKS), (unsigned)S), dim3(NT), 0, st, a);
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3((unsigned)(a.B * SM_CHUNKS), (unsigned)S), dim3(NT), 0, st, a);
  return launch_status();
}
static int launch_smooth_bwd(const SmoothArgs& a, int S, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(smooth_bwd1_kernel, dim3((unsigned)(a.B * SM_CHUNKS), (unsigned)S), dim3(NT), 0, st, a);
  hipLaunchKernelGGL(smooth_bwd2_kernel, dim3((unsigned)(a.B * SM_CHUNKS), (unsigned)S), dim3(NT), 0, st, a);
  return launch_status();
}

int bbd_smooth_loss_fwd(const float* disp, const float* img, float* mean_disp, float* sums, int B, int h, int w,
                        void* stream) {
  if (!disp || !img || !mean_disp || !sums || B <= 0 || h < 2 || w < 2) return BBD_E_BADARG;
  SmoothArgs a = {};
  a.disp[0] = disp; a.img[0] = img; a.h[0] = h; a.w[0] = w; a.mean = mean_disp; a.sums = sums; a.B = B;
  return launch_smooth_fwd(a, 1, stream);
}

int bbd_smooth_loss_bwd(const float* disp, const float* img, const float* mean_disp, const float* gscale,
                        float* grad_disp, float* dots, int B, int h, int w, void* stream) {
  if (!disp || !img || !mean_disp || !gscale || !grad_disp || !dots || B <= 0 || h < 2 || w < 2) return BBD_E_BADARG;
  SmoothArgs a = {};
  a.disp[0] = disp; a.img[0] = img; a.grad[0] = grad_disp; a.h[0] = h; a.w[0] = w;
  a.mean = const_cast<float*>(mean_disp); a.dots = dots; a.gscale = gscale; a.B = B;
  return launch_smooth_bwd(a, 1, stream);
}

static int fill_smooth(const void* const* disp, const void* const* img, const int32_t* hw, int S, int B, SmoothArgs* a) {
  if (!disp || !img || !hw || S <= 0 || S > MAX_SCALES || B <= 0) return BBD_E_BADARG;
  for (int i = 0; i < S; ++i) {
    a->disp[i] = static_cast<const float*>(disp[i]);
    a->img[i] = static_cast<const float*>(img[i]);
    a->h[i] = hw[2 * i]; a->w[i] = hw[2 * i + 1];
    if (!a->disp[i] || !a->img[i] || a->h[i] < 2 || a->w[i] < 2) return BBD_E_BADARG;
  }
  a->B = B;
  return 0;
}

int bbd_smooth_loss_multi_fwd(const void* const* disp, const void* const* img, const int32_t* hw, float* mean_disp,
                              float* sums, int S, int B, void* stream) {
  SmoothArgs a = {};
  if (!mean_disp || !sums || fill_smooth(disp, img, hw, S, B, &a)) return BBD_E_BADARG;
  a.mean = mean_disp; a.sums = sums;
  return launch_smooth_fwd(a, S, stream);
}

int bbd_smooth_loss_multi_bwd(const void* const* disp, const void* const* img, const int32_t* hw, const float* mean_disp,
                              const float* gscale, void* const* grad_disp, float* dots, int S, int B, void* stream) {
  SmoothArgs a = {};
  if (!mean_disp || !gscale || !grad_disp || !dots || fill_smooth(disp, img, hw, S, B, &a)) return BBD_E_BADARG;
  for (int i = 0; i < S; ++i) {
    a.grad[i] = static_cast<float*>(grad_disp[i]);
    if (!a.grad[i]) return BBD_E_BADARG;
  }
  a.mean = const_cast<float*>(mean_disp); a.dots = dots; a.gscale = gscale;
  return launch_smooth_bwd(a, S, stream);
}

#ifdef BBD_STAMPS
int bbd_debug_set_stamps(void* buf) { g_stamps_host_ptr = static_cast<unsigned long long*>(buf); return 0; }
#endif

int bbd_selftest_div(int blocks, int iters, unsigned seed, int32_t* mismatches, void* stream) {
  if (!mismatches || blocks <= 0 || iters <= 0) return BBD_E_BADARG;
  if (seed & 0x80000000u)   // detail mode: mismatches[0..9] per category
    hipLaunchKernelGGL(selftest_div_detail_kernel, dim3((unsigned)blocks), dim3(NT), 0,
                       static_cast<hipStream_t>(stream), (uint32_t)seed, iters, mismatches);
  else
    hipLaunchKernelGGL(selftest_div_kernel, dim3((unsigned)blocks), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       (uint32_t)seed, iters, mismatches);
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

int bbd_backproject_bwd(const float* grad_points, const float* inv_K, float* grad_depth, int n, int H, int W,
                        void* stream) {
  if (!grad_points || !inv_K || !grad_depth || n <= 0 || H <= 0 || W <= 0) return BBD_E_BADARG;
  const size_t tot = (size_t)n * H * W;
  const unsigned grid = (unsigned)((tot + NT - 1) / NT < 4096 ? (tot + NT - 1) / NT : 4096);
  hipLaunchKernelGGL(backproject_bwd_kernel, dim3(grid), dim3(NT), 0, static_cast<hipStream_t>(stream), grad_points,
                     inv_K, grad_depth, n, H, W);
  return launch_status();
}

int bbd_project3d_bwd_blocks(void) { return P3D_BLOCKS; }

int bbd_project3d_bwd(const float* points, const float* K, const float* T, const float* grad_grid,
                      float* grad_points, float* gp_partial, int n, int H, int W, double eps, void* stream) {
  if (!points || !K || !T || !grad_grid || !grad_points || !gp_partial || n <= 0 || H < 2 || W < 2) return BBD_E_BADARG;
  hipLaunchKernelGGL(project3d_bwd_kernel, dim3(P3D_BLOCKS, (unsigned)n), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), points, K, T, grad_grid, grad_points, gp_partial, H, W, (float)eps);
  return launch_status();
}

int bbd_ssim_bwd(const float* x, const float* y, const float* grad_out, float* grad_x, int n, int H, int W,
                 void* stream) {
  if (!x || !y || !grad_out || !grad_x || n <= 0 || H < 3 || W < 3) return BBD_E_BADARG;
  const int ntiles = bbd_num_tiles(H, W);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3((unsigned)(n * 3 * ntiles)), dim3(NT), 0, static_cast<hipStream_t>(stream),
                     x, y, grad_out, grad_x, H, W, ntiles);
  return launch_status();
}

}  // extern "C"
