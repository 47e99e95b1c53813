// Probe: do the matrix pipe (MFMA) and the VALU of a gfx950 SIMD work at the same time?
//
// VERDICT r03 "next" #4: in all three precisions of the render kernel, (matrix-pipe busy %) + (VALU issue %) is 96-98 % --
// as if the two never overlapped.  This probe answers the hardware half of that question in isolation:
//   * TWO waves on one SIMD, wave A an endless stream of independent v_mfma_f32_32x32x16_f16, wave B independent VALU
//     work (plain v_fma_f32, or the max / cvt_pk / fma_mix mix of the kernel's hi/lo split): if the SIMD overlaps them, each
//     wave runs at (almost) the speed it has alone; if not, the times add;
//   * ONE wave interleaving an MFMA with K independent VALU instructions (the "MFMA shadow"), K = 2 / 4 / 6 / 8;
//   * accumulators in VGPRs against AGPRs (inline asm pins the register class);
//   * the fp32 MFMA (v_mfma_f32_32x32x2_f32) and the 4x4x1 blend MFMA next to the same VALU stream;
//   * VALU instructions that READ an accumulator the matrix pipe is still writing (the ReLU / split of a finished layer):
//     dependent mode, the form the kernel's chunk phases have.
// Workgroup = 512 threads = 8 waves, 96 KiB of LDS so that ONE workgroup owns a CU; waves 0-3 take role A, waves 4-7 role B;
// HW_ID is logged to show which SIMD each wave landed on.  Every wave times its own loop with s_memtime.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_coissue.hip -o build/probe_coissue && build/probe_coissue
// Experiment tooling, not part of the product path.  Output committed as profiles/r04_probe_coissue.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum Mode {
  IDLE = 0,
  MFMA_V = 1,        // 32x32x16 f16, 4 independent accumulators in VGPRs
  MFMA_A = 2,        // the same, accumulators in AGPRs
  VALU_FMA = 3,      // 8 independent v_fma_f32 chains
  VALU_SPLIT = 4,    // the kernel's split mix: v_max_f32 x2, v_cvt_pk_f16_f32, v_fma_mixlo_f16, v_fma_mixhi_f16, v_pk_max_u16 (hm: v_pk_max_u16)
  MIX_V2 = 5, MIX_V4 = 6, MIX_V6 = 7, MIX_V8 = 8,   // one wave: MFMA (VGPR acc) followed by K independent v_fma_f32
  MIX_A4 = 9, MIX_A8 = 10,                          // the same with AGPR accumulators, K = 4 / 8
  MFMA_F32 = 11,     // v_mfma_f32_32x32x2_f32, VGPR accumulators
  MFMA_4X4 = 12,     // v_mfma_f32_4x4x1_16b_f32 (the gather's blend), VGPR accumulators
  DEP_V = 13,        // per step: 4 MFMAs into acc[j], then 16 VALU (v_max_f32) READING the acc the MFMAs of the PREVIOUS step wrote
  MIX_SPLIT6 = 14,   // one wave: MFMA (VGPR acc) followed by the 6-instruction split mix
  DEP_A = 15,        // DEP_V with AGPR accumulators (v_accvgpr_read in front of every VALU read)
  NMODES
};

static const char* MODE_NAME[NMODES] = {"idle", "mfma16.v", "mfma16.a", "valu.fma", "valu.split", "mix.v+2", "mix.v+4", "mix.v+6",
                                        "mix.v+8", "mix.a+4", "mix.a+8", "mfma32.v", "mfma4x4.v", "dep.v", "mix.v+split6", "dep.a"};
// (MFMA instructions, VALU instructions) one loop iteration issues
static const int MODE_MFMA[NMODES] = {0, 8, 8, 0, 0, 4, 4, 4, 4, 4, 4, 8, 8, 4, 4, 4};
static const int MODE_VALU[NMODES] = {0, 0, 0, 32, 30, 8, 16, 24, 32, 16, 32, 0, 0, 16, 24, 32};

#define MFMA16_V(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16_A(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA32_V(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))
#define MFMA4_V(acc) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
#define FMA8() do { FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5); FMA(x6); FMA(x7); } while (0)
#define FMA2() do { FMA(x0); FMA(x1); } while (0)
#define FMA4() do { FMA(x0); FMA(x1); FMA(x2); FMA(x3); } while (0)
#define FMA6() do { FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5); } while (0)
// the activation split of the f16x2 / f16f6 chunk phases, per PAIR of values: ReLU (2 x v_max), hi = cvt_pk, lo = 2 x fma_mix
// (x - hi rounded to f16), running maximum of the hi bit patterns
#define SPLIT6(p, q, h, l, mx) asm volatile(                                   \
    "v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_cvt_pk_f16_f32 %2, %0, %1\n\t" \
    "v_fma_mixlo_f16 %3, %2, -1.0, %0 op_sel_hi:[1,0,0]\n\t"                     \
    "v_fma_mixhi_f16 %3, %2, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"      \
    "v_pk_max_u16 %4, %4, %2"                                                   \
    : "+v"(p), "+v"(q), "+v"(h), "+v"(l), "+v"(mx))
#define RELU(dst, src) asm volatile("v_max_f32 %0, 0, %1" : "=v"(dst) : "v"(src))
#define RELU_A(dst, src) asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_max_f32 %0, 0, %0" : "=v"(dst) : "a"(src))

__global__ void __launch_bounds__(512) coissue_kernel(int mode_a, int mode_b, int iters, unsigned long long* cycles, unsigned* hwid,
                                                      float* sink) {
  extern __shared__ char pad[];   // 96 KiB dynamic: one workgroup per CU
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mode = (wave < 4) ? mode_a : mode_b;
  if (lane == 0 && iters < 0) pad[wave] = 1;   // keep the allocation alive
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
  float fa = 0.001f * lane, fb = 0.002f * lane;
  f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f), acc2 = (f32x16)(0.f), acc3 = (f32x16)(0.f);
  f32x4 q0 = (f32x4)(0.f), q1 = (f32x4)(0.f), q2 = (f32x4)(0.f), q3 = (f32x4)(0.f);
  float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
  float m = 0.999f, c = 0.001f;
  unsigned h0 = 0, l0 = 0, mx0 = 0, h1 = 0, l1 = 0, mx1 = 0;
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = 0.f;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  switch (mode) {
    case MFMA_V:
      for (int it = 0; it < iters; ++it) { MFMA16_V(acc0); MFMA16_V(acc1); MFMA16_V(acc2); MFMA16_V(acc3); MFMA16_V(acc0); MFMA16_V(acc1); MFMA16_V(acc2); MFMA16_V(acc3); }
      break;
    case MFMA_A:
      for (int it = 0; it < iters; ++it) { MFMA16_A(acc0); MFMA16_A(acc1); MFMA16_A(acc2); MFMA16_A(acc3); MFMA16_A(acc0); MFMA16_A(acc1); MFMA16_A(acc2); MFMA16_A(acc3); }
      break;
    case VALU_FMA:
      for (int it = 0; it < iters; ++it) { FMA8(); FMA8(); FMA8(); FMA8(); }
      break;
    case VALU_SPLIT:
      for (int it = 0; it < iters; ++it) {
        SPLIT6(x0, x1, h0, l0, mx0); SPLIT6(x2, x3, h1, l1, mx1); SPLIT6(x4, x5, h0, l0, mx0); SPLIT6(x6, x7, h1, l1, mx1);
        SPLIT6(x0, x1, h0, l0, mx0);
      }
      break;
    case MIX_V2:
      for (int it = 0; it < iters; ++it) { MFMA16_V(acc0); FMA2(); MFMA16_V(acc1); FMA2(); MFMA16_V(acc2); FMA2(); MFMA16_V(acc3); FMA2(); }
      break;
    case MIX_V4:
      for (int it = 0; it < iters; ++it) { MFMA16_V(acc0); FMA4(); MFMA16_V(acc1); FMA4(); MFMA16_V(acc2); FMA4(); MFMA16_V(acc3); FMA4(); }
      break;
    case MIX_V6:
      for (int it = 0; it < iters; ++it) { MFMA16_V(acc0); FMA6(); MFMA16_V(acc1); FMA6(); MFMA16_V(acc2); FMA6(); MFMA16_V(acc3); FMA6(); }
      break;
    case MIX_V8:
      for (int it = 0; it < iters; ++it) { MFMA16_V(acc0); FMA8(); MFMA16_V(acc1); FMA8(); MFMA16_V(acc2); FMA8(); MFMA16_V(acc3); FMA8(); }
      break;
    case MIX_A4:
      for (int it = 0; it < iters; ++it) { MFMA16_A(acc0); FMA4(); MFMA16_A(acc1); FMA4(); MFMA16_A(acc2); FMA4(); MFMA16_A(acc3); FMA4(); }
      break;
    case MIX_A8:
      for (int it = 0; it < iters; ++it) { MFMA16_A(acc0); FMA8(); MFMA16_A(acc1); FMA8(); MFMA16_A(acc2); FMA8(); MFMA16_A(acc3); FMA8(); }
      break;
    case MFMA_F32:
      for (int it = 0; it < iters; ++it) { MFMA32_V(acc0); MFMA32_V(acc1); MFMA32_V(acc2); MFMA32_V(acc3); MFMA32_V(acc0); MFMA32_V(acc1); MFMA32_V(acc2); MFMA32_V(acc3); }
      break;
    case MFMA_4X4:
      for (int it = 0; it < iters; ++it) { MFMA4_V(q0); MFMA4_V(q1); MFMA4_V(q2); MFMA4_V(q3); MFMA4_V(q0); MFMA4_V(q1); MFMA4_V(q2); MFMA4_V(q3); }
      break;
    case DEP_V:
      // the chunk-phase shape: the MFMAs of this half-step write acc0/acc1 while the VALU reads acc2/acc3 (complete: four
      // younger MFMAs have been issued since, and the pipe is in order), then the roles swap
      for (int it = 0; it < iters; it += 2) {
        MFMA16_V(acc0); MFMA16_V(acc1); MFMA16_V(acc0); MFMA16_V(acc1);
        for (int i = 0; i < 8; ++i) { RELU(r[i], acc2[i]); RELU(r[8 + i], acc3[i]); }
        MFMA16_V(acc2); MFMA16_V(acc3); MFMA16_V(acc2); MFMA16_V(acc3);
        for (int i = 0; i < 8; ++i) { RELU(r[i], acc0[i]); RELU(r[8 + i], acc1[i]); }
      }
      break;
    case DEP_A:   // the same with the accumulators in AGPRs: every value the VALU reads costs a v_accvgpr_read first
      for (int it = 0; it < iters; it += 2) {
        MFMA16_A(acc0); MFMA16_A(acc1); MFMA16_A(acc0); MFMA16_A(acc1);
        for (int i = 0; i < 8; ++i) { RELU_A(r[i], acc2[i]); RELU_A(r[8 + i], acc3[i]); }
        MFMA16_A(acc2); MFMA16_A(acc3); MFMA16_A(acc2); MFMA16_A(acc3);
        for (int i = 0; i < 8; ++i) { RELU_A(r[i], acc0[i]); RELU_A(r[8 + i], acc1[i]); }
      }
      break;
    case MIX_SPLIT6:
      for (int it = 0; it < iters; ++it) {
        MFMA16_V(acc0); SPLIT6(x0, x1, h0, l0, mx0); MFMA16_V(acc1); SPLIT6(x2, x3, h1, l1, mx1);
        MFMA16_V(acc2); SPLIT6(x4, x5, h0, l0, mx0); MFMA16_V(acc3); SPLIT6(x6, x7, h1, l1, mx1);
      }
      break;
    default: break;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results are read below: past every hazard window
  if (lane == 0) {
    cycles[blockIdx.x * 8 + wave] = t1 - t0;
    hwid[blockIdx.x * 8 + wave] = id;
  }
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(h0 ^ l0 ^ mx0 ^ h1 ^ l1 ^ mx1);
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i] + r[i];
  for (int i = 0; i < 4; ++i) s += q0[i] + q1[i] + q2[i] + q3[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

struct Result { double cyc_a, cyc_b, ms; };

static Result run(int mode_a, int mode_b, int iters, unsigned long long* d_cyc, unsigned* d_id, float* d_sink, int blocks,
                  std::vector<unsigned>* ids = nullptr) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t lds = 96 * 1024;
  for (int rep = 0; rep < 2; ++rep) {   // the second launch is the measured one
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(coissue_kernel, dim3(blocks), dim3(512), lds, 0, mode_a, mode_b, iters, d_cyc, d_id, d_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(blocks * 8);
  CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
  if (ids) { ids->resize(blocks * 8); CK(hipMemcpy(ids->data(), d_id, ids->size() * 4, hipMemcpyDeviceToHost)); }
  double a = 0, b = 0;
  for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)cyc[i * 8 + w];
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return {a / (blocks * 4.0), b / (blocks * 4.0), ms};
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 4096;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int blocks = prop.multiProcessorCount;
  printf("# probe_coissue on %s (%s), %d CUs, clock %d kHz; %d workgroups x 8 waves (one workgroup per CU), %d loop iterations\n",
         prop.name, prop.gcnArchName, blocks, prop.clockRate, blocks, iters);
  CK(hipFuncSetAttribute((const void*)coissue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  unsigned long long* d_cyc; unsigned* d_id; float* d_sink;
  CK(hipMalloc(&d_cyc, blocks * 8 * 8)); CK(hipMalloc(&d_id, blocks * 8 * 4)); CK(hipMalloc(&d_sink, 512 * 4));

  std::vector<unsigned> ids;
  run(MFMA_V, VALU_FMA, 64, d_cyc, d_id, d_sink, blocks, &ids);
  printf("# HW_ID of workgroup 0 (wave: simd_id cu_id): ");
  for (int w = 0; w < 8; ++w) printf("%d: simd %u cu %u | ", w, (ids[w] >> 4) & 3, (ids[w] >> 8) & 15);
  int paired = 0;
  for (int i = 0; i < blocks; ++i) {
    bool ok = true;
    for (int w = 0; w < 4; ++w) ok = ok && (((ids[i * 8 + w] >> 4) & 3) == ((ids[i * 8 + w + 4] >> 4) & 3)) && (((ids[i * 8 + w] >> 8) & 15) == ((ids[i * 8 + w + 4] >> 8) & 15));
    paired += ok;
  }
  printf("\n# workgroups whose waves w and w+4 share a SIMD: %d of %d\n", paired, blocks);

  struct Cfg { int a, b; const char* what; };
  const Cfg cfgs[] = {
      {MFMA_V, IDLE, "A alone: MFMA stream, VGPR accumulators"},
      {MFMA_A, IDLE, "A alone: MFMA stream, AGPR accumulators"},
      {MFMA_F32, IDLE, "A alone: fp32 MFMA stream"},
      {MFMA_4X4, IDLE, "A alone: 4x4x1 MFMA stream"},
      {IDLE, VALU_FMA, "B alone: v_fma_f32 stream"},
      {IDLE, VALU_SPLIT, "B alone: split mix (max,max,cvt_pk,mixlo,mixhi,pk_max)"},
      {MFMA_V, MFMA_V, "both waves MFMA (pipe shared: expect 2x)"},
      {VALU_FMA, VALU_FMA, "both waves VALU (port shared: expect 2x)"},
      {MFMA_V, VALU_FMA, "A MFMA (VGPR acc) || B v_fma_f32   <-- the question"},
      {MFMA_A, VALU_FMA, "A MFMA (AGPR acc) || B v_fma_f32"},
      {MFMA_V, VALU_SPLIT, "A MFMA (VGPR acc) || B split mix"},
      {MFMA_A, VALU_SPLIT, "A MFMA (AGPR acc) || B split mix"},
      {MFMA_F32, VALU_FMA, "A fp32 MFMA || B v_fma_f32"},
      {MFMA_4X4, VALU_FMA, "A 4x4x1 MFMA || B v_fma_f32"},
      {MIX_V2, IDLE, "one wave: MFMA + 2 VALU"},
      {MIX_V4, IDLE, "one wave: MFMA + 4 VALU"},
      {MIX_V6, IDLE, "one wave: MFMA + 6 VALU"},
      {MIX_V8, IDLE, "one wave: MFMA + 8 VALU"},
      {MIX_A4, IDLE, "one wave: MFMA (AGPR acc) + 4 VALU"},
      {MIX_A8, IDLE, "one wave: MFMA (AGPR acc) + 8 VALU"},
      {MIX_SPLIT6, IDLE, "one wave: MFMA + the 6-instruction split mix"},
      {MIX_V4, MIX_V4, "two waves, each MFMA + 4 VALU"},
      {MIX_V8, MIX_V8, "two waves, each MFMA + 8 VALU"},
      {MIX_SPLIT6, MIX_SPLIT6, "two waves, each MFMA + split mix (the chunk phase's instruction mix)"},
      {DEP_V, IDLE, "one wave: 4 MFMAs + 16 v_max READING the previous step's accumulators"},
      {DEP_V, DEP_V, "two waves of the same"},
      {DEP_A, IDLE, "one wave: the same with AGPR accumulators (v_accvgpr_read + v_max per value)"},
      {DEP_A, DEP_A, "two waves of the same"},
  };
  printf("\n%-14s %-14s %12s %12s %9s %9s %9s %9s  %s\n", "role A", "role B", "cyc/iter A", "cyc/iter B", "clk/MFMA", "clk/VALU", "ms",
         "GHz(A)", "what");
  for (const Cfg& c : cfgs) {
    Result r = run(c.a, c.b, iters, d_cyc, d_id, d_sink, blocks);
    double ca = r.cyc_a / iters, cb = r.cyc_b / iters;
    int nm = MODE_MFMA[c.a] + MODE_MFMA[c.b], nv = MODE_VALU[c.a] + MODE_VALU[c.b];
    double span = (c.a == IDLE ? cb : (c.b == IDLE ? ca : (ca > cb ? ca : cb)));
    double ghz = (c.a == IDLE ? r.cyc_b : r.cyc_a) / (r.ms * 1e6);
    char mf[32] = "-", vf[32] = "-";
    if (nm) snprintf(mf, sizeof mf, "%.1f", span / nm);
    if (nv) snprintf(vf, sizeof vf, "%.2f", span / nv);
    printf("%-14s %-14s %12.1f %12.1f %9s %9s %9.3f %9.2f  %s\n", MODE_NAME[c.a], MODE_NAME[c.b], ca, cb, mf, vf, r.ms, ghz, c.what);
  }
  printf("\n# reading: 'clk/MFMA' and 'clk/VALU' divide the longer of the two roles' loop time by ALL MFMA / VALU instructions both roles\n"
         "# issued per iteration on the SIMD.  Co-issue: 'A || B' rows keep the clk/MFMA of 'A alone' and the clk/VALU of 'B alone'.\n"
         "# No overlap: the loop times add.  GHz(A) = cycles of the timed loop / wall time of the launch (lower bound of the clock).\n");
  return 0;
}
