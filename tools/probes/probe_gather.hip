// Probe of the texture-addresser cost of the hoisted-latent gather (experiment tooling, not part of the product path):
// how many cycles does one 64-lane global_load_dwordx4 cost when its lanes address the channels-last hoisted map in
// different lane -> (point, 16-byte piece) patterns?  Every wave of a full-occupancy grid (8 waves per CU, like the
// render kernel) issues the loads of `gathers` bilinear gathers (4 texels x 512 bytes per point, 32 points) and folds
// them with 4 weights, so the loads cannot be dropped.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_gather.hip -o build/probe_gather && build/probe_gather
// Patterns (lane -> point p, piece c of the 32 sixteen-byte pieces of a 512-byte block):
//   0 "half"  : p = lane & 31, piece = 2*i + (lane >> 5)          (the shipped kernel: the two lanes of a point are 32 apart)
//   1 "pair"  : p = lane >> 1, piece = 2*i + (lane & 1)           (adjacent lanes share 32 contiguous bytes)
//   2 "quad"  : p = (lane >> 2) + 16*(i & 1), piece = 4*(i >> 1) + (lane & 3)   (4 adjacent lanes = 64 contiguous bytes)
//   3 "oct"   : p = (lane >> 3) + 8*(i & 3),  piece = 8*(i >> 2) + (lane & 7)   (8 adjacent lanes = one 128-byte line)
// i = 0..15 is the load index inside one texel.  Texels per point are random (worst case) or ray-like (runs of equal
// texels for neighbouring points), selected by argv[1] = run length (default 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ void __launch_bounds__(256, 2) gather_kernel(const float* __restrict__ map, const int* __restrict__ texel,
                                                        int stride, int gathers, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < gathers; ++g) {
    const int* tx = texel + ((size_t)wave * gathers + g) * 32 * 4;
    const int block = g % 6;  // which 512-byte block of the texel's 3 KiB
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int p, piece;
        if (PATTERN == 0) { p = lane & 31; piece = 2 * i + (lane >> 5); }
        else if (PATTERN == 1) { p = lane >> 1; piece = 2 * i + (lane & 1); }
        else if (PATTERN == 2) { p = (lane >> 2) + 16 * (i & 1); piece = 4 * (i >> 1) + (lane & 3); }
        else { p = (lane >> 3) + 8 * (i & 3); piece = 8 * (i >> 2) + (lane & 7); }
        const float* src = map + (size_t)tx[p * 4 + t] * stride + block * 128 + piece * 4;
        v[i] = *(const f32x4*)src;
      }
      const float w = 0.25f + 0.125f * t;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += v[i] * w;
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(acc[i]));
    }
  }
  f32x4 s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[wave] = s[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// Candidate replacement: "quad" loads (4 adjacent lanes read 64 contiguous bytes of ONE (point, lane-half) pair) and the
// bilinear accumulation AND the lane <-> register transposition done by the matrix core: v_mfma_f32_4x4x1_16B_f32 with
// A = the loaded dword of slot k, B = w_t * [lane % 4 == k], C = D = four accumulator registers.  Lane n of a quad then
// receives D[i][n] += X_{lane i}[slot n] * w_t(n): its own point's pieces, weighted with its own weight -- one fused
// multiply-add per value, like the v_fmac of the shipped gather.  check_kernel runs both forms on the same logical
// latents and writes the 64 accumulators of every lane for a bit-exact comparison.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int pos_half(int f) {  // shipped layout: logical feature f of a 128-block -> position
  const int hh = f >> 6, r = f & 63, m = r >> 4, q = (r >> 2) & 3, e = r & 3;
  return 32 * m + 8 * q + 4 * hh + e;
}
__device__ __forceinline__ int pos_quad(int f) {  // candidate layout: register r = 4e + i  <->  piece i, dword e
  const int hh = f >> 6, r = f & 63, m = r >> 4, e = (r >> 2) & 3, i = r & 3;
  return 64 * hh + 16 * m + 4 * i + e;
}
template <int S>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true);
}

template <bool QUAD>
__device__ __forceinline__ void gather_block(const float* __restrict__ blk, const int (&tx)[4], const float (&w)[4], int stride,
                                             int lane, f32x16 (&h)[4]) {
  const int hh = lane >> 5, c = lane & 3;
  if (!QUAD) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* p = blk + (size_t)tx[t] * stride + 4 * hh;
      f32x4 v[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[m][q] = *(const f32x4*)(p + 32 * m + 8 * q);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) h[m][4 * q + e] = fmaf(v[m][q][e], w[t], h[m][4 * q + e]);
#pragma unroll
      for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(h[m]));
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int o[4] = {quad_bcast<0>(tx[t]), quad_bcast<1>(tx[t]), quad_bcast<2>(tx[t]), quad_bcast<3>(tx[t])};
      f32x4 x[4][4];  // [slot s][segment g]
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) x[s][g] = *(const f32x4*)(blk + (size_t)o[s] * stride + 64 * hh + 16 * g + 4 * c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float b = c == k ? w[t] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x4 d = {h[g][4 * e], h[g][4 * e + 1], h[g][4 * e + 2], h[g][4 * e + 3]};
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(x[k][g][e], b, d, 0, 0, 0);
            h[g][4 * e] = d[0];
            h[g][4 * e + 1] = d[1];
            h[g][4 * e + 2] = d[2];
            h[g][4 * e + 3] = d[3];
          }
      }
    }
  }
}

template <bool QUAD>
__global__ void __launch_bounds__(256, 2) check_kernel(const float* __restrict__ map, const int* __restrict__ texel,
                                                       int stride, int gathers, float* __restrict__ out, int write_all) {
  const int lane = threadIdx.x & 63, j = lane & 31;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  f32x16 h[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) h[m][r] = 0.001f * (float)(lane + 16 * m + r);
  for (int g = 0; g < gathers; ++g) {
    const int* tp = texel + (((size_t)wave * gathers + g) * 32 + j) * 4;
    const int tx[4] = {tp[0], tp[1], tp[2], tp[3]};
    const float fx = 0.37f + 0.001f * (float)((j * 7 + g) % 61), fy = 0.81f - 0.002f * (float)((j * 3 + g) % 47);
    const float w[4] = {(1.f - fx) * (1.f - fy), fx * (1.f - fy), (1.f - fx) * fy, fx * fy};
    gather_block<QUAD>(map + (g % 6) * 128, tx, w, stride, lane, h);
  }
  if (write_all) {
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) out[((size_t)wave * 64 + lane) * 64 + 16 * m + r] = h[m][r];
  } else {
    float s = 0.f;
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) s += h[m][r];
    if (s == 123.456f) out[wave] = s;
  }
}

template <bool QUAD>
static float run_check(const float* map, const int* texel, int stride, int gathers, float* out, int waves, int write_all) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  check_kernel<QUAD><<<waves / 4, 256>>>(map, texel, stride, gathers, out, write_all);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  check_kernel<QUAD><<<waves / 4, 256>>>(map, texel, stride, gathers, out, write_all);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

template <int PATTERN>
static float run(const float* map, const int* texel, int stride, int gathers, float* out, int waves) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  gather_kernel<PATTERN><<<waves / 4, 256>>>(map, texel, stride, gathers, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  gather_kernel<PATTERN><<<waves / 4, 256>>>(map, texel, stride, gathers, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char** argv) {
  const int run_len = argc > 1 ? atoi(argv[1]) : 1;
  const int texels = 128 * 128, stride = 768;   // one image's hoisted map: 50 MB
  const int waves = 256 * 8 * 4, gathers = 24;  // 4 rounds of full occupancy, 24 gathers per wave
  std::vector<float> h((size_t)texels * stride);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
  std::vector<int> tx((size_t)waves * gathers * 32 * 4);
  unsigned s = 12345u;
  for (size_t q = 0; q < tx.size() / 4; ++q) {
    // ray-like coherence: `run_len` consecutive points share the same footprint
    if (q % run_len == 0) {
      s = s * 1664525u + 1013904223u;
      const int x = (s >> 8) % 127, y = (s >> 16) % 127;
      tx[4 * q] = y * 128 + x;
      tx[4 * q + 1] = y * 128 + x + 1;
      tx[4 * q + 2] = (y + 1) * 128 + x;
      tx[4 * q + 3] = (y + 1) * 128 + x + 1;
    } else {
      for (int c = 0; c < 4; ++c) tx[4 * q + c] = tx[4 * (q - 1) + c];
    }
  }
  float *map, *out;
  int* texel;
  CK(hipMalloc(&map, h.size() * 4));
  CK(hipMalloc(&out, waves * 4));
  CK(hipMalloc(&texel, tx.size() * 4));
  CK(hipMemcpy(map, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(texel, tx.data(), tx.size() * 4, hipMemcpyHostToDevice));
  const char* names[4] = {"half (shipped)", "pair", "quad", "oct"};
  float ms[4] = {run<0>(map, texel, stride, gathers, out, waves), run<1>(map, texel, stride, gathers, out, waves),
                 run<2>(map, texel, stride, gathers, out, waves), run<3>(map, texel, stride, gathers, out, waves)};
  printf("# run length %d: %d waves x %d gathers x 64 dwordx4 loads; cycles per load instruction per CU-resident wave set\n", run_len, waves, gathers);
  for (int p = 0; p < 4; ++p) {
    // each CU runs waves/256 waves in total, 8 at a time; TA time per instruction = total time / instructions per CU
    const double instr_per_cu = (double)waves / 256 * gathers * 64;
    printf("%-16s %8.3f ms   %6.1f clocks of 2.4 GHz per load instruction (CU-wide)\n", names[p], ms[p], ms[p] * 1e-3 * 2.4e9 / instr_per_cu);
  }
  // ---- shipped gather (half layout + v_fmac) against quad loads + 4x4x1 MFMA accumulate: bit-exact? faster?
  {
    std::vector<float> ha(h.size()), hb(h.size());
    for (size_t t = 0; t < (size_t)texels; ++t)
      for (int blk = 0; blk < 6; ++blk)
        for (int f = 0; f < 128; ++f) {
          unsigned u = (unsigned)(t * 768 + blk * 128 + f) * 2654435761u;
          const float val = ((float)(u >> 8) / 16777216.0f - 0.5f) * ((u & 7) == 0 ? 37.0f : 1.0f);
          const int hh = f >> 6, r = f & 63, m = r >> 4, q = (r >> 2) & 3, e = r & 3;
          ha[t * 768 + blk * 128 + 32 * m + 8 * q + 4 * hh + e] = val;                    // shipped layout
          const int e2 = (r >> 2) & 3, i2 = r & 3;
          hb[t * 768 + blk * 128 + 64 * hh + 16 * m + 4 * i2 + e2] = val;                  // candidate layout
        }
    float* map_b;
    CK(hipMalloc(&map_b, h.size() * 4));
    CK(hipMemcpy(map, ha.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(map_b, hb.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int cw = 64;
    float *oa, *ob;
    CK(hipMalloc(&oa, (size_t)cw * 64 * 64 * 4));
    CK(hipMalloc(&ob, (size_t)cw * 64 * 64 * 4));
    check_kernel<false><<<cw / 4, 256>>>(map, texel, stride, 6, oa, 1);
    check_kernel<true><<<cw / 4, 256>>>(map_b, texel, stride, 6, ob, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> ra((size_t)cw * 64 * 64), rb(ra.size());
    CK(hipMemcpy(ra.data(), oa, ra.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rb.data(), ob, rb.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    double worst = 0;
    for (size_t i = 0; i < ra.size(); ++i) {
      if (memcmp(&ra[i], &rb[i], 4)) {
        if (diff < 5) printf("  mismatch at %zu: %.9g vs %.9g\n", i, ra[i], rb[i]);
        ++diff;
        const double d = fabs((double)ra[i] - rb[i]);
        if (d > worst) worst = d;
      }
    }
    printf("# quad + MFMA accumulate vs shipped gather: %zu of %zu values differ (worst abs %.3g)\n", diff, ra.size(), worst);
    const float t_a = run_check<false>(map, texel, stride, gathers, out, waves, 0);
    const float t_b = run_check<true>(map_b, texel, stride, gathers, out, waves, 0);
    printf("shipped gather + v_fmac      %8.3f ms\nquad loads + 4x4x1 MFMA     %8.3f ms\n", t_a, t_b);
  }
  return 0;
}
