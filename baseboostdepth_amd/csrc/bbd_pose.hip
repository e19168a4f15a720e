// bbd_pose.hip - pose composition of the boosted recipe as one launch each way.
//
// The reference composes the poses of a step with a Python loop per frame (trainer.py:359-388, 403-405, 415-418):
// index_select of the sub-batch, a chain of 4x4 matmuls back to frame 0 (incremental mode), a detached clone whose
// translation is divided by pose_error (T_error), and - partial mode - a cat + where that swaps in the translation
// column of a direct 0 -> f pose for some rows.  Here every composed 4x4 of the step is one row of a small integer
// table (built once per batch signature on the host) and ONE kernel evaluates all of them from the pose network's
// step matrices; the backward is one kernel too (a wave per step row: its lanes take the outputs that reference it, a
// fixed butterfly sums them: deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbd_hip.h"

namespace {

constexpr int NT = 256;

struct M4 {
  float m[16];
};

__device__ __forceinline__ M4 load4(const float* p) {
  M4 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.m[i] = p[i];
  return r;
}
__device__ __forceinline__ M4 eye4() {
  M4 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return r;
}
// torch.matmul of two 4x4 matrices as the reference's CPU path rounds it: ((a0 b0 + a1 b1) + a2 b2) + a3 b3
__device__ __forceinline__ M4 mul4(const M4& a, const M4& b) {
  M4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = a.m[i * 4 + 0] * b.m[j];
      acc = acc + a.m[i * 4 + 1] * b.m[4 + j];
      acc = acc + a.m[i * 4 + 2] * b.m[8 + j];
      acc = acc + a.m[i * 4 + 3] * b.m[12 + j];
      r.m[i * 4 + j] = acc;
    }
  return r;
}
__device__ __forceinline__ M4 transpose4(const M4& a) {
  M4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r.m[i * 4 + j] = a.m[j * 4 + i];
  return r;
}

// product of the chain's steps [lo, hi)
__device__ __forceinline__ M4 chain(const float* steps, const int32_t* row, int lo, int hi) {
  M4 t = eye4();
  bool first = true;
  for (int k = lo; k < hi; ++k) {
    const M4 s = load4(steps + (size_t)row[1 + k] * 16);
    t = first ? s : mul4(t, s);          // eye @ S is S exactly
    first = false;
  }
  return t;
}

__global__ __launch_bounds__(NT) void pose_compose_fwd_kernel(const float* __restrict__ steps, const int32_t* __restrict__ table,
                                                              float* __restrict__ out, int NO, float pose_error) {
  const int o = blockIdx.x * NT + threadIdx.x;
  if (o >= NO) return;
  const int32_t* row = table + (size_t)o * BBD_COMPOSE_STRIDE;
  M4 t = chain(steps, row, 0, row[0]);
  if (row[9] & BBD_COMPOSE_REPLACE) {                      // chained rotation, direct translation column (trainer.py:416)
    const float* d = steps + (size_t)row[8] * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) t.m[a * 4 + 3] = d[a * 4 + 3];
  }
  if (row[9] & BBD_COMPOSE_ERROR) {                        // T_error: translation / pose_error (trainer.py:376-377)
#pragma unroll
    for (int a = 0; a < 3; ++a) t.m[a * 4 + 3] = t.m[a * 4 + 3] / pose_error;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[(size_t)o * 16 + i] = t.m[i];
}

// refs: for step row r the entries refs[refs_off[r] .. refs_off[r+1]) = (output row, position k in its chain; -1 = it is
// the output's `direct` pose).  One wave per step row: lane e takes the row's e-th reference (each is two chain products
// of dependent 4x4 multiplies - latency, not arithmetic), then the 16 sums are reduced over the lanes by a fixed
// butterfly, so the result does not depend on timing.
__global__ __launch_bounds__(64) void pose_compose_bwd_kernel(const float* __restrict__ steps, const int32_t* __restrict__ table,
                                                             const int32_t* __restrict__ refs_off, const int32_t* __restrict__ refs,
                                                             const float* __restrict__ gout, float* __restrict__ gsteps, int R) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const int e0 = refs_off[r], e1 = refs_off[r + 1];
  for (int e = e0 + lane; e < e1; e += 64) {
    const int o = refs[2 * e], k = refs[2 * e + 1];
    const int32_t* row = table + (size_t)o * BBD_COMPOSE_STRIDE;
    if (row[9] & BBD_COMPOSE_ERROR) continue;              // detached clone: no gradient (trainer.py:376)
    M4 G = load4(gout + (size_t)o * 16);
    if (k < 0) {                                           // this row supplied the translation column
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a * 4 + 3] += G.m[a * 4 + 3];
      continue;
    }
    if (row[9] & BBD_COMPOSE_REPLACE) {
#pragma unroll
      for (int a = 0; a < 4; ++a) G.m[a * 4 + 3] = 0.0f;   // the chained product kept its first three columns only
    }
    // T = L S_k Rr  =>  dS_k = L^T G Rr^T
    const M4 L = chain(steps, row, 0, k), Rr = chain(steps, row, k + 1, row[0]);
    const M4 d = mul4(mul4(transpose4(L), G), transpose4(Rr));
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += d.m[i];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    acc[i] = v;
  }
  if (lane < 16) {
    float v = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) v = (lane == i) ? acc[i] : v;
    gsteps[(size_t)r * 16 + lane] = v;
  }
}

int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int bbd_pose_compose_fwd(const float* steps, const int32_t* table, float* out, int NO, double pose_error, void* stream) {
  if (!steps || !table || !out || NO < 0 || pose_error == 0.0) return BBD_E_BADARG;
  if (NO == 0) return 0;
  hipLaunchKernelGGL(pose_compose_fwd_kernel, dim3((unsigned)((NO + NT - 1) / NT)), dim3(NT), 0,
                     static_cast<hipStream_t>(stream), steps, table, out, NO, (float)pose_error);
  return launch_status();
}

int bbd_pose_compose_bwd(const float* steps, const int32_t* table, const int32_t* refs_off, const int32_t* refs,
                         const float* grad_out, float* grad_steps, int R, void* stream) {
  if (!steps || !table || !refs_off || !refs || !grad_out || !grad_steps || R < 0) return BBD_E_BADARG;
  if (R == 0) return 0;
  hipLaunchKernelGGL(pose_compose_bwd_kernel, dim3((unsigned)R), dim3(64), 0,
                     static_cast<hipStream_t>(stream), steps, table, refs_off, refs, grad_out, grad_steps, R);
  return launch_status();
}

}  // extern "C"
