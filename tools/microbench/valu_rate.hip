// How long does one wave64 vector instruction occupy a gfx950 SIMD, by instruction kind and by waves per SIMD?
// (planning input for the fused kernels: their VALU floor = sum over kinds of count x cycles.)
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w -o valu_rate tools/microbench/valu_rate.hip && ./valu_rate > profiles/r03/valu_rate.txt
//
// Every kernel runs 8 independent dependency chains per lane so that one wave can issue back to back; blocks of 256
// threads put one wave on each SIMD, blocks per CU = waves per SIMD.  Reported per (kind, waves/SIMD):
//   ns per wave-instruction per SIMD (wall time / instructions issued on one SIMD),
//   shader cycles per wave-instruction per SIMD = that x the clock the chip held in THIS launch, which is measured in
//   the kernel as d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6),
//   and the cycles one wave alone sees between its own consecutive instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Kind { FMA = 0, MULADD, CMPSEL, RCP, IMUL, ADD64, CVT, FLOOR, DPP, PKFMA, PKADD, PKMUL, IADD, MOVXOR, FMA_IADD, FMA_FLOOR, MAXMIN,
            FMA_VVV, PKFMA_VVV, PKADD_VV, PKMUL_VV, FMA_VSV, FMA_VV_INLINE, FMA_VV_LITERAL, MULADD_VV, MUL_VS, ADD_V_LITERAL, KINDS };
static const char* kind_name[KINDS] = {"v_fma_f32", "v_mul_f32+v_add_f32", "v_cmp+v_cndmask(+add,mul)", "v_rcp_f32",
                                       "v_mul_lo_u32", "v_lshl_add_u64", "v_mul+v_cvt_i32_f32+v_cvt_f32_i32", "v_mul+v_floor_f32",
                                       "v_add_f32 dpp row_shr", "v_pk_fma_f32 (2 fp32 per lane)", "v_pk_add_f32", "v_pk_mul_f32",
                                       "v_add_u32", "v_xor_b32", "v_fma_f32 / v_add_u32 alternating", "v_fma_f32 / v_floor_f32 alternating",
                                       "v_max_f32 / v_min_f32", "v_fma_f32, three distinct VGPR operands", "v_pk_fma_f32, three distinct VGPR pairs",
                                       "v_pk_add_f32, two distinct VGPR pairs", "v_pk_mul_f32, two distinct VGPR pairs",
                                       "v_fma_f32 v, s, v (one SGPR operand)", "v_fma_f32 v, v, 2.0 (inline constant)",
                                       "v_fma_f32 v, v, literal (v_fmaak)", "v_mul_f32 v, v / v_add_f32 v, v", "v_mul_f32 s, v",
                                       "v_add_f32 literal, v"};
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int REPS = 16;     // 128 vector instructions per loop iteration: the loop's own s_add / s_cmp / taken branch
                             // (~28 cycles per iteration for one wave) is < 6 % of an iteration; with 8 per iteration
                             // (round 2's form) it was 47 % and every kind read ~2x too slow
template <int KIND>
__global__ __launch_bounds__(256) void chain(float* out, unsigned long long* clk, int iters, float b, float c) {
  float a[8];
  v2f pk[8];
  unsigned u[8];
  unsigned long long w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i + 1.5f; u[i] = threadIdx.x * 7u + i + 3u; w[i] = (unsigned long long)out + u[i]; pk[i] = v2f{a[i], a[i] + 0.25f}; }
  const v2f pb = {b, b * 1.0001f}, pc = {c, c + 0.125f};
  const unsigned ub = __float_as_uint(b) | 1u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == FMA) a[i] = fmaf(a[i], b, c);
      else if (KIND == MULADD) a[i] = (i & 1) ? a[i] + c : a[i] * b;
      else if (KIND == CMPSEL) a[i] = (i & 1) ? a[i] * b : (a[i] > c ? a[i] : b + a[i]);
      else if (KIND == RCP) a[i] = __builtin_amdgcn_rcpf(a[i]);
      else if (KIND == IMUL) u[i] = u[i] * (u[(i + 1) & 7] | ub);
      else if (KIND == PKFMA) pk[i] = __builtin_elementwise_fma(pk[i], pb, pc);
      else if (KIND == PKADD) pk[i] = pk[i] + pc;
      else if (KIND == PKMUL) pk[i] = pk[i] * pb;
      else if (KIND == IADD) u[i] = u[i] + u[(i + 1) & 7];            // (partner operands: nothing for the compiler to fold)
      else if (KIND == MOVXOR) u[i] = u[i] ^ u[(i + 3) & 7];
      else if (KIND == FMA_IADD) { if (i & 1) u[i] = u[i] + u[(i + 2) & 7]; else a[i] = fmaf(a[i], b, c); }
      else if (KIND == FMA_FLOOR) { if (i & 1) a[i] = floorf(a[i - 1]); else a[i] = fmaf(a[i], b, a[i + 1]); }
      else if (KIND == MAXMIN) a[i] = (i & 1) ? fmaxf(a[i], a[(i + 1) & 7]) : fminf(a[i], a[(i + 3) & 7]);
      // operands that are all per-lane registers (the kernels' case), not the broadcast constants of the kinds above
      else if (KIND == FMA_VVV) a[i] = fmaf(a[i], a[(i + 1) & 7], a[(i + 3) & 7]);
      else if (KIND == PKFMA_VVV) pk[i] = __builtin_elementwise_fma(pk[i], pk[(i + 1) & 7], pk[(i + 3) & 7]);
      else if (KIND == FMA_VSV) a[i] = fmaf(a[i], b, a[(i + 3) & 7]);
      else if (KIND == FMA_VV_INLINE) a[i] = fmaf(a[i], a[(i + 1) & 7], 2.0f);
      else if (KIND == FMA_VV_LITERAL) a[i] = fmaf(a[i], a[(i + 1) & 7], 0.1111111f);
      else if (KIND == MULADD_VV) a[i] = (i & 1) ? a[i] + a[(i + 3) & 7] : a[i] * a[(i + 1) & 7];
      else if (KIND == MUL_VS) a[i] = a[i] * b;
      else if (KIND == ADD_V_LITERAL) a[i] = a[i] + 0.1111111f;
      else if (KIND == PKADD_VV) pk[i] = pk[i] + pk[(i + 1) & 7];
      else if (KIND == PKMUL_VV) pk[i] = pk[i] * pk[(i + 3) & 7];
      else if (KIND == ADD64) w[i] = (w[i] << 2) + (unsigned long long)u[i];
      else if (KIND == CVT) a[i] = (float)(int)(a[i] * b);          // v_mul + v_cvt_i32_f32 + v_cvt_f32_i32
      else if (KIND == FLOOR) a[i] = floorf(a[i] * b);               // v_mul + v_floor
      else if (KIND == DPP)
        a[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x111, 0xf, 0xf, false));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i] + (float)(unsigned)w[i] + pk[i].x + pk[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
static void run_kind(float* out, unsigned long long* clk, int iters) {
  // wave-instructions per loop body, per chain element (what the compiler emits; check with -save-temps if a kind changes)
  const double per_elem = KIND == CMPSEL ? 2.0        // odd: 1 mul; even: add + cmp + cndmask
                          : KIND == CVT ? 3.0 : KIND == FLOOR ? 2.0 : KIND == IMUL ? 2.0 /* v_or + v_mul_lo */
                          : KIND == MOVXOR ? 110.0 / 128 : KIND == MAXMIN ? 136.0 / 128 /* counted in the .s */ : 1.0;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;           // s_memrealtime ticks at 100 MHz
    const double instr_per_simd = (double)iters * 8 * REPS * per_elem * wps;
    const double ns = ms * 1e6 / instr_per_simd;
    printf("%-28s waves/SIMD %d: %8.3f ms  %.3f ns/wave-instr/SIMD  clock %.2f GHz -> %.2f cycles/instr/SIMD   (one wave: %.2f cycles between its instructions)\n",
           kind_name[KIND], wps, ms, ns, ghz, ns * ghz, (double)h[0] / (iters * 8.0 * REPS * per_elem));
  }
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&clk, 64);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("# %s, %d CUs, clockRate %d kHz; 8 independent chains per lane, 2000 iterations x 128 instructions, 256-thread blocks (1 wave per SIMD each)\n",
         p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int iters = 2000;
  run_kind<FMA>(out, clk, iters);
  run_kind<MULADD>(out, clk, iters);
  run_kind<CMPSEL>(out, clk, iters);
  run_kind<RCP>(out, clk, iters);
  run_kind<IMUL>(out, clk, iters);
  run_kind<ADD64>(out, clk, iters);
  run_kind<CVT>(out, clk, iters);
  run_kind<FLOOR>(out, clk, iters);
  run_kind<DPP>(out, clk, iters);
  run_kind<PKFMA>(out, clk, iters);
  run_kind<PKADD>(out, clk, iters);
  run_kind<PKMUL>(out, clk, iters);
  run_kind<IADD>(out, clk, iters);
  run_kind<MOVXOR>(out, clk, iters);
  run_kind<FMA_IADD>(out, clk, iters);
  run_kind<FMA_FLOOR>(out, clk, iters);
  run_kind<MAXMIN>(out, clk, iters);
  run_kind<FMA_VVV>(out, clk, iters);
  run_kind<PKFMA_VVV>(out, clk, iters);
  run_kind<PKADD_VV>(out, clk, iters);
  run_kind<PKMUL_VV>(out, clk, iters);
  run_kind<FMA_VSV>(out, clk, iters);
  run_kind<FMA_VV_INLINE>(out, clk, iters);
  run_kind<FMA_VV_LITERAL>(out, clk, iters);
  run_kind<MULADD_VV>(out, clk, iters);
  run_kind<MUL_VS>(out, clk, iters);
  run_kind<ADD_V_LITERAL>(out, clk, iters);
  // a long run of the densest kind: the clock the chip settles at after ~2 s of back-to-back launches
  for (int rep = 0; rep < 40; ++rep) hipLaunchKernelGGL(chain<FMA>, dim3(256 * 4), dim3(256), 0, 0, out, clk, 20000, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  printf("# after 40 x 20000-iteration v_fma launches at 4 waves/SIMD: clock %.2f GHz\n", (double)h[0] / (double)h[1] * 0.1);
  return 0;
}
