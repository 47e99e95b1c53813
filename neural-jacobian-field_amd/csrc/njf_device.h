// Device-side building blocks shared by the fused kernels (gfx950 / CDNA4 only).
//
// Register-resident MLP: a wave owns a tile of 32 points.  Lane l holds point j = l & 31 and the
// feature half hh = l >> 5.  An activation vector of width 32*MB is an array f32x16 v[MB] in the
// C/D layout of v_mfma_f32_32x32x2_f32: v[mb][r] is logical feature 16*MB*hh + 16*mb + r of point
// j.  Computing  out^T = W * in^T  with the weights as the A operand and the activations as the B
// operand makes register r of the previous layer's D tile exactly K-step r of the next layer, so
// activations never leave registers between layers; only weights stream (HBM/L2 -> LDS by
// buffer_load ... lds DMA, double buffered in 32 KiB chunks, one barrier per chunk).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ReLU as one integer max on the bit pattern (negative floats are negative ints; -0.0 -> +0.0; NaN passes through):
// avoids the canonicalising v_max_f32 x,x that fmaxf() emits in front of every v_max_f32.
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

// MFMA precision of the fused MLPs (same activation layout, same chunk sizes, different A-fragment packing):
//   PREC_F32   : v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak)
//   PREC_F16X2 : every fp32 operand x is split as x = hi + lo (two fp16), and hi*hi + hi*lo + lo*hi is
//                accumulated in fp32 by v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s peak, 3 instructions per product
//                block): the dropped lo*lo term is 2^-22 relative, i.e. fp32-class accuracy at 3/16 of the cost.
//   PREC_F16F6 : hi*hi as in PREC_F16X2; the two correction products (2^-11 of the result) in block-scaled fp6 (e2m3)
//                by v_mfma_scale_f32_32x32x64_f8f6f4 -- K = 64 in the issue time of ONE K = 16 f16 instruction (35 vs 34
//                clocks measured, profiles/r02_probe_mx.txt): 24 matrix instructions per 128x64 weight chunk instead of
//                48.  Only the 128-wide layers (MBO = 4, NKB = 2 chunks) take this form; the narrow ones stay F16X2.
//   PREC_F16   : plain fp16 products (round 5; BASELINE config 5 "fp16 MFMA fused-MLP"): weights packed ONCE as single fp16,
//                activations rounded to fp16 behind the ReLU (one v_cvt_pk_f16_f32 + one packed integer max per PAIR),
//                v_mfma_f32_32x32x16_f16 with fp32 accumulation -- issue factor 1, no hi/lo split, no correction products.
//                A reduced-precision mode with its OWN stated tolerance (2^-12 relative rounding of every operand); never the
//                default, never the headline.  A whole 128x128 layer is ONE 32 KiB chunk (half the barriers), and the hoisted
//                map it gathers from is stored in fp16 (half the gather traffic).
#define PREC_F32 0
#define PREC_F16X2 1
#define PREC_F16F6 2
#define PREC_F16 3

// Workgroup = 4 waves (one per SIMD), two workgroups resident per CU: the two waves sharing a SIMD's
// MFMA pipe belong to DIFFERENT workgroups, so one workgroup's barrier / gather / encoding phases are
// covered by the other's MFMA stream (with one 8-wave workgroup both waves of a SIMD stall together).
#define NJF_WAVES 4
#define NJF_THREADS (NJF_WAVES * 64)
#define NJF_CHUNK 8192  // floats per weight chunk (32 KiB)

// LDS carve (floats).  One dynamic array only (a second __shared__ object makes hipcc drain
// vmcnt(0) in front of every ds_read of a DMA pipeline).  Sized per kernel so that two workgroups
// fit in the CU's 160 KiB: 2 x 32 KiB weight buffers + biases (+ per-wave scratch of the PDF stage).
// NJF_ASYNC_STREAM (experiment build, needs -DNJF_WAVES=8: ONE 8-wave workgroup per CU): four weight buffers and a
// barrier-free stream -- see stream_step.
#define NJF_STREAM_BUFFERS 2
#define LDS_CTR_FLOATS 0
#define LDS_W0 0
#define LDS_W1 NJF_CHUNK
#define LDS_BIAS (NJF_STREAM_BUFFERS * NJF_CHUNK)
#define LDS_BIAS_FLOATS 2752        // density 1312 | colour 96 | Jacobian head <= 1312 (+ pad)
#define LDS_BIAS_FLOATS_PROPOSAL 1344
#define LDS_SCRATCH_PER_WAVE 528    // proposal pass: w'[<=256] | cdf[<=257] (+ pad)
#define LDS_FLOATS_RENDER (LDS_BIAS + LDS_BIAS_FLOATS + LDS_CTR_FLOATS)
#define LDS_FLOATS_PROPOSAL (LDS_BIAS + LDS_BIAS_FLOATS_PROPOSAL + NJF_WAVES * LDS_SCRATCH_PER_WAVE + LDS_CTR_FLOATS)
#define LDS_SCRATCH_PROPOSAL (LDS_BIAS + LDS_BIAS_FLOATS_PROPOSAL)

extern __shared__ __attribute__((aligned(16))) float njf_lds[];

__device__ __forceinline__ float mfma_step(float a, float b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  return 0.f;
}

// Timeline instrumentation of ONE wave (experiment builds only, -DNJF_STAMPS; tools/stamps.py): (tag << 24 | low 24
// bits of s_memtime) words, logged to a spare LDS area and copied out when the wave retires.  -DNJF_STAMPS logs a wave of the
// render kernel, -DNJF_STAMPS -DNJF_STAMPS_PROPOSAL a wave of the proposal kernel (its LDS use ends higher).  Everything the
// kernels say about it goes through the macros below, which are empty in product builds.
#ifdef NJF_STAMPS
#define NJF_STAMP_SLOTS 1024
#ifdef NJF_STAMPS_PROPOSAL
#define NJF_STAMP_BASE LDS_FLOATS_PROPOSAL
#define NJF_STAMP_LOGS_PROPOSAL true
#else
#define NJF_STAMP_BASE LDS_FLOATS_RENDER
#define NJF_STAMP_LOGS_PROPOSAL false
#endif
__device__ unsigned njf_stamp_out[NJF_STAMP_SLOTS];
#define NJF_STAMP_FIELD int stamp_i;   // next free slot of this wave's log in LDS (-1: this wave does not log)
#define NJF_STAMP_INIT(st) (st).stamp_i = -1
#define NJF_STAMP(st, tag)                                                                                   \
  do {                                                                                                       \
    if ((st).stamp_i >= 0 && (st).stamp_i < NJF_STAMP_SLOTS) {                                               \
      njf_lds[NJF_STAMP_BASE + (st).stamp_i] =                                                               \
          __uint_as_float(((unsigned)(tag) << 24) | ((unsigned)__builtin_readcyclecounter() & 0xffffffu));   \
      (st).stamp_i += 1;                                                                                     \
    }                                                                                                        \
  } while (0)
// stamps of the proposal kernel (the render kernel's are plain NJF_STAMP: the shared device functions log for whichever wave is armed)
#define NJF_STAMP_P(st, tag) do { if (NJF_STAMP_LOGS_PROPOSAL) NJF_STAMP(st, tag); } while (0)
// arm wave 1 of a mid-grid workgroup of the kernel this build logs (IS_PROPOSAL: the caller is the proposal kernel)
#define NJF_STAMP_ARM(st, IS_PROPOSAL, wave)                                                                                \
  const bool njf_stamping = ((IS_PROPOSAL) == NJF_STAMP_LOGS_PROPOSAL) && blockIdx.x == gridDim.x / 2 + 3 && (wave) == 1; \
  if (njf_stamping) (st).stamp_i = 0;                                                                                     \
  NJF_STAMP(st, 9)
#define NJF_STAMP_FLUSH(st, tag, lane)                                                                       \
  do {                                                                                                       \
    NJF_STAMP(st, tag);                                                                                      \
    if (njf_stamping)                                                                                        \
      for (int i_ = (lane); i_ < NJF_STAMP_SLOTS; i_ += 64)                                                  \
        njf_stamp_out[i_] = i_ < (st).stamp_i ? __float_as_uint(njf_lds[NJF_STAMP_BASE + i_]) : 0u;          \
  } while (0)
#define NJF_STAMP_PIN(...) asm volatile("" : __VA_ARGS__)   // keeps the stamped phase's results in front of the next stamp
#define NJF_STAMP_LDS_FLOATS NJF_STAMP_SLOTS
#else
#define NJF_STAMP_FIELD
#define NJF_STAMP_INIT(st) do {} while (0)
#define NJF_STAMP(st, tag) do {} while (0)
#define NJF_STAMP_P(st, tag) do {} while (0)
#define NJF_STAMP_ARM(st, IS_PROPOSAL, wave) do {} while (0)
#define NJF_STAMP_FLUSH(st, tag, lane) do {} while (0)
#define NJF_STAMP_PIN(...) do {} while (0)
#define NJF_STAMP_LDS_FLOATS 0
#endif

// ------------------------------------------------------------------------------------------
// Weight stream: chunks of NJF_CHUNK floats, consumed in program order, wrapping every
// `per_pass` chunks.  step() = one barrier: after it the current chunk is resident, and the DMA
// for the next one has been issued into the other buffer (whose readers all passed the barrier).
// ------------------------------------------------------------------------------------------
// GAP_AT / GAP (compile time: as run-time fields they cost the fp6-corrected kernels, which run at the register limit, 54
// spilled VGPRs): PREC_F16 networks use the first NJF_RESNET_CHUNKS_F16 chunk slots of their NJF_RESNET_CHUNKS-slot blob (the
// host-side offsets of the blobs are the same in every precision); chunks at position >= GAP_AT of a pass sit GAP slots
// further on in the blob.
struct WeightStreamState {
  __amdgpu_buffer_rsrc_t rsrc;  // the packed blob as a raw buffer (stride 0)
  int per_pass;    // chunks per tile pass
  int total;       // chunks over the whole workgroup lifetime
  int idx;         // next chunk to consume
  int in_pass;     // idx % per_pass of the chunk being prefetched
  // DMA job of the chunk being prefetched: NJF_DMA_ROUNDS rounds of 4 KiB, issued one at a time between the MFMA
  // groups of the chunk being consumed (dma_round) instead of as one burst behind the barrier: the burst -- 8 waves x
  // 8 x 1 KiB through the CU's 64 B/clk vector-memory path, all at the same moment -- stalled every wave ~1,150
  // cycles per chunk at the issue of its own loads (17 % of the render kernel, measured with s_memtime stamps).
  int dma_voff;    // this lane's byte offset inside a wave's 1 KiB share of a round (lane * 16)
  int dma_soff;    // byte offset (wave-uniform) of this wave's share of round 0 of the chunk being prefetched
  float* dma_dst;  // this wave's LDS destination of round 0
  int dma_next;    // 0 = job pending, NJF_DMA_ROUNDS = issued (or nothing to prefetch)
  NJF_STAMP_FIELD
};
template <int GAP_AT_, int GAP_>
struct WeightStreamT : WeightStreamState {
  static constexpr int GAP_AT = GAP_AT_, GAP = GAP_;
};
typedef WeightStreamT<0x7fffffff, 0> WeightStream;
#define NJF_DMA_ROUNDS (NJF_CHUNK / (NJF_THREADS * 4))

__device__ __forceinline__ void dma_issue(const WeightStreamState& st, int r) {
  // 256 threads x 16 B = 4 KiB per round.  LDS destination is wave-uniform base + lane*16 (hardware), the source is
  // buffer base + wave-uniform byte offset (SGPR) + lane*16 (VGPR).  The MUBUF form (buffer_load_dwordx4 ... lds) is
  // used instead of global_load_lds_dwordx4 on purpose: the latter is FLAT-encoded, and while a FLAT instruction that
  // may touch LDS is outstanding hipcc's wait-count pass turns EVERY s_waitcnt into lgkmcnt(0)/vmcnt(0) -- the
  // A-fragment prefetch of mma_chunk (ds_reads of unit u+1 issued before the MFMAs of unit u) then waits for the
  // loads it has just issued.  With the buffer form the waits are exact (lgkmcnt(4)).
#ifdef NJF_ABLATE_DMA  // experiment builds only: no L2 -> LDS weight traffic (weights stay whatever is in LDS; results are garbage)
  return;
#endif
  __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (__attribute__((address_space(3))) void*)(st.dma_dst + r * NJF_THREADS * 4),
                                           16, st.dma_voff, st.dma_soff + r * NJF_THREADS * 16, 0, 0);
}

// Issue the whole pending job at once (chunk shapes that do not interleave; start of the kernel).
__device__ __forceinline__ void stream_flush(WeightStreamState& st) {
  if (st.dma_next == 0) {
#pragma unroll
    for (int r = 0; r < NJF_DMA_ROUNDS; ++r) dma_issue(st, r);
    st.dma_next = NJF_DMA_ROUNDS;
  }
}

template <class ST>
__device__ __forceinline__ void dma_job(ST& st, int chunk, int buf, int wave) {
  if constexpr (ST::GAP != 0) {
    if (chunk >= ST::GAP_AT) chunk += ST::GAP;   // (wave-uniform: scalar compare + add)
  }
  st.dma_soff = chunk * (NJF_CHUNK * 4) + wave * 1024;
  st.dma_dst = njf_lds + buf * NJF_CHUNK + wave * 256;
  st.dma_next = 0;
}

template <class ST>
__device__ __forceinline__ void stream_begin(ST& st, const float* g, int per_pass, int passes, int wave,
                                             int lane) {
  // raw buffer descriptor: stride 0, num_records in bytes (range check far above any blob), dword 3 = 32-bit data format
  st.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0x7ffffff0, 0x00020000);
  st.per_pass = per_pass;
  st.total = per_pass * passes;
  st.idx = 0;
  st.in_pass = 0;
  st.dma_voff = lane * 16;
  NJF_STAMP_INIT(st);
  dma_job(st, 0, 0, wave);
  stream_flush(st);
}

// One barrier: after it the current chunk is resident and the other buffer is free (all of its readers passed the
// barrier); the job that refills it is set up here and issued by the consumer of the current chunk, round by round.
template <class ST>
__device__ __forceinline__ const float* stream_step(ST& st, int wave, int lane) {
#ifdef NJF_ABLATE_BARRIER  // experiment builds only: weights stay whatever is in LDS (results are garbage)
  return njf_lds + (st.idx++ & 1) * NJF_CHUNK;
#endif
  stream_flush(st);  // rounds the previous consumer did not issue (chunk shapes without interleaving)
  NJF_STAMP(st, 1);  // chunk's work issued
  // Every wave must have ITS share of the chunk in LDS before anyone passes the barrier.  LDS-DMA completion is
  // counted by vmcnt, and the workgroup fence of __syncthreads() does not cover it (the compiler only waits vmcnt
  // in front of this wave's own reads of the buffer): without the explicit wait another wave can read a round that
  // is still in flight -- a race the full-frame ray-sharding test caught once the rounds were issued later.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NJF_STAMP(st, 2);  // own memory operations drained
  __syncthreads();
  NJF_STAMP(st, 3);  // barrier passed
  const float* cur = njf_lds + (st.idx & 1) * NJF_CHUNK;
  st.idx += 1;
  st.in_pass += 1;
  if (st.in_pass == st.per_pass) st.in_pass = 0;
  // The last chunk of a workgroup has nothing to prefetch.  Its consumer is always a lin_out-shaped chunk (MBO = 1),
  // which only ever calls stream_flush; the interleaving consumers (mma_chunk with MBO = 4) issue their 8 rounds
  // unconditionally and are never last.
  if (st.idx < st.total) dma_job(st, st.in_pass, st.idx & 1, wave);
  return cur;
}


// hi/lo split of two fp32 values (ReLU'd on request) into elements 2p, 2p+1 of the packed B operands:
// hi = fp16(x), lo = fp16(x - hi) (the residual is exact in fp32, so lo carries the next 11 bits of x).
// The residual and its conversion are ONE v_fma_mix{lo,hi}_f16 per value -- fma(-hi, 1.0, x) evaluated in fp32 and
// rounded once to fp16, the same value because x - hi is exact -- instead of v_cvt_f32_f16 + v_sub_f32 + half a
// v_cvt_pk_f16_f32: 2.5 instead of 3.5 VALU instructions per activation value (bit-identical outputs, -1.6 % frame
// time).  hipcc does not select the mix form by itself (it folds fma(h, -1, x) back into a subtraction), hence asm;
// the asm reads only results of compiler-visible VALU instructions (v_max / v_cvt_pk), never an MFMA result directly.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool RELU>
__device__ __forceinline__ void split_pair(float x0, float x1, int p, f16x8& hi, f16x8& lo) {
  if (RELU) {
    x0 = relu_bits(x0);
    x1 = relu_bits(x1);
  }
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  hi[2 * p] = h0;
  hi[2 * p + 1] = h1;
  const f16x2 hp = {h0, h1};
  const unsigned hu = __builtin_bit_cast(unsigned, hp);
  unsigned lu;
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(hu), "v"(x0));
  asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(hu), "v"(x1));
  u32x4 lv = __builtin_bit_cast(u32x4, lo);
  lv[p] = lu;
  lo = __builtin_bit_cast(f16x8, lv);
}

typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
// max of two pairs of packed 16-bit patterns (asm: the compiler otherwise re-derives the halves from the fp32 sources
// with scalar conversions instead of using the packed register the split already produced)
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// split_pair with the packed {hi0, hi1} and {lo0, lo1} halves returned as dwords
template <bool RELU>
__device__ __forceinline__ void split_pair_u(float x0, float x1, unsigned& hu, unsigned& lu) {
  if (RELU) {
    x0 = relu_bits(x0);
    x1 = relu_bits(x1);
  }
  const f16x2 hp = {(_Float16)x0, (_Float16)x1};
  hu = __builtin_bit_cast(unsigned, hp);
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(hu), "v"(x0));
  asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(hu), "v"(x1));
}

// PREC_F16: two fp32 activations -> one dword of two fp16 (round to nearest even), ReLU'd on request on the PACKED halves:
// negative fp16 bit patterns are negative 16-bit integers, so one v_pk_max_i16 with 0 clamps both (-0.0 -> +0.0; a NaN
// with the sign bit set becomes 0, as with relu_bits).  One conversion + one max per PAIR of values.
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <bool RELU>
__device__ __forceinline__ unsigned pack_pair_f16(float x0, float x1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 xp = {x0, x1};
  const f16x2 hp = __builtin_convertvector(xp, f16x2);   // ONE v_cvt_pk_f16_f32 (two scalar casts become 2 v_cvt + v_perm)
  if constexpr (RELU) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, hp), z));
  } else {
    return __builtin_bit_cast(unsigned, hp);
  }
}

// Layout of one PREC_F16F6 weight chunk (K = 64 inputs x 128 outputs, NJF_CHUNK floats = 32 KiB, byte offsets):
//   [F6_HI   , +16 KiB) : hi fp16 A fragments [t][m][lane][8 x f16]   (t = K-step of 16, m = output block of 32)
//   [F6_P1   , + 8 KiB) : fp6 fragments, bytes 0..15 of each lane's 24  [m][w][lane][16 B]   w = 0: fp6(hi), 1: fp6(lo)
//   [F6_P2   , + 4 KiB) : fp6 fragments, bytes 16..23                   [m][w][lane][8 B]
//   [F6_SCALE, +512 B ) : E8M0 scale bytes [w][lane] as one dword per lane (byte m = output block m)
// A lane's 32 fp6 elements are its 32 K-values of the chunk in the order of the f16 path (element 8*t + i = value i of
// K-step t), so the activation side is ONE v_cvt_scalef32_pk32_fp6_f16 of the 16 registers of packed hi (or lo) halves.
#define F6_HI 0
#define F6_P1 16384
#define F6_P2 24576
#define F6_SCALE 28672


// ------------------------------------------------------------------------------------------
// out[MBO] += W[:, kb range] * in   for the (kb,q) groups stored at `wl` (LDS, packed
// [kb][q][mb][lane][e]).  RELU applies max(.,0) to the B operand on the fly.
// ------------------------------------------------------------------------------------------
template <int PREC, int MBO, int NKB, int KB0, bool RELU, int KBI, class ST>
__device__ __forceinline__ void mma_chunk(ST& st, const float* __restrict__ wl, int lane,
                                          const f32x16 (&in)[KBI], f32x16 (&out)[MBO]) {
  // DMA rounds of the next chunk ride between the MFMA groups of this one (the 128-wide layers: MBO = 4, NKB = 2, eight
  // groups): two rounds behind each of the first four groups, so the last round still has half a chunk to land before
  // the next barrier's vmcnt(0).  Every other shape issues the job up front.
  constexpr bool SPREAD = MBO == 4 && (NKB == 2 || (PREC == PREC_F16 && NKB == 4)) && NJF_DMA_ROUNDS == 8;
  if constexpr (!SPREAD) stream_flush(st);
  if constexpr (PREC == PREC_F32) {
    // hipcc schedules this fully unrolled body as groups of 4*MBO MFMAs and re-issues each group's ds_read_b128s
    // two MFMAs (128 cycles) before the registers are needed, which covers the LDS latency; a hand-pipelined
    // variant measured 1-2 % slower (round-1 A/B, tools/ablate.sh).
    const float* base = wl + lane * 4;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 a[MBO];
#pragma unroll
        for (int m = 0; m < MBO; ++m) a[m] = *(const f32x4*)(base + ((kb * 4 + q) * MBO + m) * 256);
        if constexpr (SPREAD) {
          if (kb * 4 + q < 4) {
            dma_issue(st, 2 * (kb * 4 + q));
            dma_issue(st, 2 * (kb * 4 + q) + 1);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float b = in[KB0 + kb][q * 4 + e];
          if (RELU) b = relu_bits(b);
#pragma unroll
          for (int m = 0; m < MBO; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][e], b, out[m], 0, 0, 0);
        }
      }
    }
  } else if constexpr (PREC == PREC_F16F6 && MBO == 4 && NKB == 2) {
    // 16 f16 MFMAs (hi*hi, K-steps t = 0..3 x output blocks m = 0..3) followed by 8 block-scaled fp6 MFMAs
    // (W_lo6 * x_hi6 and W_hi6 * x_lo6 per output block, K = 64 each).  The hi/lo split of K-step t+1 rides behind
    // the MFMAs of step t as in the F16X2 path; every packed hi pair is also folded into a running maximum, from
    // which the lane's power-of-two scale follows: 2^(e-2) for the hi values (largest element in [4, 8), clamped at
    // 7.5 by the conversion) and 2^(e-13) for the residuals (|lo| <= 2^-11 of its hi's binade, so <= 4 after scaling).
    // The two 32-value conversions are long single instructions (~100 clocks each, profiles/r02_probe_mx.txt) during
    // which this wave issues nothing else, so each is placed right behind a group of MFMAs that keeps the pipe fed.
    typedef __attribute__((address_space(3))) const char* lds_ptr;  // explicit LDS pointers: the asm below would hide
    const lds_ptr wb = (lds_ptr)(const char*)wl;                    // the address space from the compiler's inference
    lds_ptr p16 = wb + lane * 16;   // made opaque: all fragment addresses become base + immediate offset (otherwise
    lds_ptr p8 = wb + lane * 8;     // one loop-invariant address VGPR per fragment is kept alive)
    asm volatile("" : "+v"(p16), "+v"(p8));
    unsigned bh[4][4], bl[4][4], mx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      split_pair_u<RELU>(in[KB0][2 * p], in[KB0][2 * p + 1], bh[0][p], bl[0][p]);
      mx[p] = RELU ? bh[0][p] : (bh[0][p] & 0x7fff7fffu);
    }
#define NJF_LDS(T) const __attribute__((address_space(3))) T*
    auto hfrag = [&](int i) { return *(NJF_LDS(f16x8))(p16 + i * 1024); };                  // hi fp16 fragment (t, m): i = 4t + m
    auto f6frag = [&](int idx) {                                                           // fp6 fragment idx = 2m + w
      const i32x4 a4 = *(NJF_LDS(i32x4))(p16 + F6_P1 + idx * 1024);
      // volatile: keeps the load-store optimiser from fusing the 8-byte tails of two fragments into one ds_read2st64_b64,
      // whose four consecutive result registers then have to be moved next to each fragment's first 16 bytes (16 v_mov
      // per chunk in a stream that is VALU-issue bound)
      const i32x2 b2 = *(volatile NJF_LDS(i32x2))(p8 + F6_P2 + idx * 512);
      return i32x8{a4[0], a4[1], a4[2], a4[3], b2[0], b2[1], 0, 0};
    };
    auto op8 = [](const unsigned (&v)[4]) { return __builtin_bit_cast(f16x8, u32x4{v[0], v[1], v[2], v[3]}); };
    f16x8 a[2] = {hfrag(0), hfrag(1)};
    unsigned sb_h = 0, sb_l = 0;
    i32x8 bh6 = {0, 0, 0, 0, 0, 0, 0, 0}, wl6_0 = bh6, wl6_1 = bh6;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int u = 2 * t + half;
        f16x8 n[2] = {a[0], a[1]};
        if (u + 1 < 8) {
          n[0] = hfrag((u + 1) * 2);
          n[1] = hfrag((u + 1) * 2 + 1);
        } else {
          wl6_0 = f6frag(1);
          wl6_1 = f6frag(3);
        }
        if constexpr (SPREAD) {
          if (u < 4) {
            dma_issue(st, 2 * u);
            dma_issue(st, 2 * u + 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int m = 2 * half + mm;
#ifdef NJF_ABLATE_MFMA
          asm volatile("" :: "v"(a[mm]), "v"(op8(bh[t])));
#else
          out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mm], op8(bh[t]), out[m], 0, 0, 0);
#endif
          if (t + 1 < 4) {
            const int kb2 = (t + 1) >> 1, tt2 = (t + 1) & 1;
            split_pair_u<RELU>(in[KB0 + kb2][8 * tt2 + 2 * m], in[KB0 + kb2][8 * tt2 + 2 * m + 1], bh[t + 1][m], bl[t + 1][m]);
            mx[m] = pk_max_u16(mx[m], RELU ? bh[t + 1][m] : (bh[t + 1][m] & 0x7fff7fffu));
          }
        }
        if (u == 6) {
          // every hi value of the chunk exists now: lane scale (fp16 bit patterns of non-negative values order like
          // integers) and the fp6 image of the 32 hi values
          unsigned m2 = pk_max_u16(pk_max_u16(mx[0], mx[1]), pk_max_u16(mx[2], mx[3]));
          m2 = max(m2 & 0xffffu, m2 >> 16);
          const unsigned e5 = m2 >> 10;  // biased fp16 exponent of the maximum (0 for zero/subnormal)
          sb_h = e5 + 110u;              // E8M0 of 2^((e5 - 15) - 2)
          sb_l = e5 + 99u;               // ... and 2^-11 of it for the residuals
          const u32x16 hv = {bh[0][0], bh[0][1], bh[0][2], bh[0][3], bh[1][0], bh[1][1], bh[1][2], bh[1][3],
                             bh[2][0], bh[2][1], bh[2][2], bh[2][3], bh[3][0], bh[3][1], bh[3][2], bh[3][3]};
          const u32x6 x6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, hv), __uint_as_float(sb_h << 23));
          bh6 = i32x8{(int)x6[0], (int)x6[1], (int)x6[2], (int)x6[3], (int)x6[4], (int)x6[5], 0, 0};
        }
        a[0] = n[0];
        a[1] = n[1];
      }
    }
    const unsigned s_w0 = *(NJF_LDS(unsigned))(p8 + F6_SCALE - lane * 4);        // [w = 0][lane]: p8 = wb + 8*lane
    const unsigned s_w1 = *(NJF_LDS(unsigned))(p8 + F6_SCALE + 256 - lane * 4);  // [w = 1][lane]
#undef NJF_LDS
    i32x8 wl6_2 = f6frag(5), wl6_3 = f6frag(7);
    __builtin_amdgcn_sched_barrier(0);
#ifdef NJF_ABLATE_MFMA
#define NJF_MFMA6(o, a6, b6, sel, sa, sb) asm volatile("" :: "v"(a6), "v"(b6), "v"(sa), "v"(sb))
#else
#define NJF_MFMA6(o, a6, b6, sel, sa, sb) o = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, o, 2, 2, sel, (int)(sa), 0, (int)(sb))
#endif
    // fragment idx = 2*m + w: w = 1 (fp6 of W_lo) pairs with x_hi6, w = 0 (fp6 of W_hi) with x_lo6; scale byte m of the
    // lane's scale dword is selected by the op_sel argument
    NJF_MFMA6(out[0], wl6_0, bh6, 0, s_w1, sb_h);
    NJF_MFMA6(out[1], wl6_1, bh6, 1, s_w1, sb_h);
    i32x8 wh6_0 = f6frag(0), wh6_1 = f6frag(2);
    const u32x16 lv = {bl[0][0], bl[0][1], bl[0][2], bl[0][3], bl[1][0], bl[1][1], bl[1][2], bl[1][3],
                       bl[2][0], bl[2][1], bl[2][2], bl[2][3], bl[3][0], bl[3][1], bl[3][2], bl[3][3]};
    const u32x6 y6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, lv), __uint_as_float(sb_l << 23));
    const i32x8 bl6 = {(int)y6[0], (int)y6[1], (int)y6[2], (int)y6[3], (int)y6[4], (int)y6[5], 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    NJF_MFMA6(out[2], wl6_2, bh6, 2, s_w1, sb_h);
    NJF_MFMA6(out[3], wl6_3, bh6, 3, s_w1, sb_h);
    i32x8 wh6_2 = f6frag(4), wh6_3 = f6frag(6);
    __builtin_amdgcn_sched_barrier(0);
    NJF_MFMA6(out[0], wh6_0, bl6, 0, s_w0, sb_l);
    NJF_MFMA6(out[1], wh6_1, bl6, 1, s_w0, sb_l);
    NJF_MFMA6(out[2], wh6_2, bl6, 2, s_w0, sb_l);
    NJF_MFMA6(out[3], wh6_3, bl6, 3, s_w0, sb_l);
#undef NJF_MFMA6
    __builtin_amdgcn_sched_barrier(0);  // the matrix work of a chunk stays in front of the next chunk's barrier
  } else if constexpr (PREC == PREC_F16) {
    // packed [t][m][lane][8 x f16], t = K-step of 16 (lane half kh supplies the 8 k-values 16*KBI*kh + 8*t + i): one MFMA
    // per (K-step, output block).  Software pipeline per K-step: the MBO A fragments of step t+1 are requested from LDS
    // (and pinned there with sched_barrier) before the MFMAs of step t; the conversion of step t+1's B operand -- one
    // v_cvt_pk_f16_f32 and one packed max per pair, 8 VALU instructions per step -- is spread behind them (2 per MFMA: a
    // 32-clock MFMA hides ~5 single-issue instructions, MI355X_MICROARCH.md).
    const f16x8* base = (const f16x8*)wl + lane;
    constexpr int T = NKB * 2;
    unsigned bc[4], bn[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) bc[p] = pack_pair_f16<RELU>(in[KB0][2 * p], in[KB0][2 * p + 1]);
    auto op8 = [](const unsigned (&v)[4]) { return __builtin_bit_cast(f16x8, u32x4{v[0], v[1], v[2], v[3]}); };
    if constexpr (MBO == 4) {
      f16x8 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = base[i * 64];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f16x8 n[4] = {a[0], a[1], a[2], a[3]};
        if (t + 1 < T) {
#pragma unroll
          for (int i = 0; i < 4; ++i) n[i] = base[((t + 1) * 4 + i) * 64];
        }
        if constexpr (SPREAD) {  // the next chunk's 8 DMA rounds, two behind each of the first four steps
          if (t < 4) {
            dma_issue(st, 2 * t);
            dma_issue(st, 2 * t + 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#ifdef NJF_ABLATE_MFMA
          asm volatile("" :: "v"(a[m]), "v"(op8(bc)));
#else
          out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], op8(bc), out[m], 0, 0, 0);
#endif
          if (t + 1 < T) {
            const int kb2 = (t + 1) >> 1, tt2 = (t + 1) & 1;
            bn[m] = pack_pair_f16<RELU>(in[KB0 + kb2][8 * tt2 + 2 * m], in[KB0 + kb2][8 * tt2 + 2 * m + 1]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = n[i];
          bc[i] = bn[i];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f16x8 a[MBO];
#pragma unroll
        for (int m = 0; m < MBO; ++m) a[m] = base[(t * MBO + m) * 64];
        if (t + 1 < T) {
          const int kb2 = (t + 1) >> 1, tt2 = (t + 1) & 1;
#pragma unroll
          for (int p = 0; p < 4; ++p) bn[p] = pack_pair_f16<RELU>(in[KB0 + kb2][8 * tt2 + 2 * p], in[KB0 + kb2][8 * tt2 + 2 * p + 1]);
        }
#pragma unroll
        for (int m = 0; m < MBO; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], op8(bc), out[m], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) bc[p] = bn[p];
      }
    }
  } else {
    // packed [t][mb][hi|lo][lane][8 x f16], t = K-step of 16 (8 k-values from each lane half), same bytes as fp32.
    // Lane (j,hh) supplies its own registers 8*tt .. 8*tt+7 of block kb as the 8 k-values of step t = 2*kb + tt.
    // Software pipeline per K-step: (1) all 2*MBO A fragments of step t are requested from LDS up front,
    // (2) the hi/lo split of step t+1's B operand (VALU) is spread between the MFMA triples of step t, so LDS
    // latency and conversion work hide under the matrix pipe instead of preceding every triple.
    const f16x8* base = (const f16x8*)wl + lane;
    constexpr int T = NKB * 2;
    f16x8 bh, bl;
#pragma unroll
    for (int p = 0; p < 4; ++p) split_pair<RELU>(in[KB0][2 * p], in[KB0][2 * p + 1], p, bh, bl);
    if constexpr (MBO == 4) {
      // units u = (K-step t, pair of output blocks): the 4 A fragments of unit u+1 are requested (and pinned there
      // with sched_barrier) before the 6 MFMAs of unit u, which cover the LDS latency; 2 x 16 VGPRs of A live.
      // The hi/lo split of step t+1's B operand is spread behind the MFMA triples of step t (2 values each).
      f16x8 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = base[i * 64];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f16x8 nh = bh, nl = bl;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int u = 2 * t + half;
          f16x8 n[4] = {a[0], a[1], a[2], a[3]};
          if (u + 1 < 2 * T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) n[i] = base[((u + 1) * 4 + i) * 64];
          }
          if constexpr (SPREAD) {
            if (u < 4) {
              dma_issue(st, 2 * u);
              dma_issue(st, 2 * u + 1);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const int m = 2 * half + mm;
#ifdef NJF_ABLATE_MFMA  // experiment builds only: keep operands alive, skip the matrix work
            asm volatile("" :: "v"(a[2 * mm]), "v"(a[2 * mm + 1]), "v"(bh), "v"(bl));
#else
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * mm + 1], bh, out[m], 0, 0, 0);
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * mm], bl, out[m], 0, 0, 0);
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * mm], bh, out[m], 0, 0, 0);
#endif
            if (t + 1 < T) {
              const int kb2 = (t + 1) >> 1, tt2 = (t + 1) & 1;
              split_pair<RELU>(in[KB0 + kb2][8 * tt2 + 2 * m], in[KB0 + kb2][8 * tt2 + 2 * m + 1], m, nh, nl);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = n[i];
        }
        bh = nh;
        bl = nl;
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f16x8 ah[MBO], al[MBO];
#pragma unroll
        for (int m = 0; m < MBO; ++m) {
          ah[m] = base[((t * MBO + m) * 2 + 0) * 64];
          al[m] = base[((t * MBO + m) * 2 + 1) * 64];
        }
        f16x8 nh = bh, nl = bl;
        if (t + 1 < T) {
          const int kb2 = (t + 1) >> 1, tt2 = (t + 1) & 1;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            split_pair<RELU>(in[KB0 + kb2][8 * tt2 + 2 * p], in[KB0 + kb2][8 * tt2 + 2 * p + 1], p, nh, nl);
        }
#pragma unroll
        for (int m = 0; m < MBO; ++m) {
          out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, out[m], 0, 0, 0);
          out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, out[m], 0, 0, 0);
          out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, out[m], 0, 0, 0);
        }
        bh = nh;
        bl = nl;
      }
    }
  }
  if constexpr (SPREAD) st.dma_next = NJF_DMA_ROUNDS;
}

// acc[m][r] (+)= bias[16*MB*hh + 16*m + r]   (bias in LDS, logical order)
// ASSIGN (the accumulator starts at the bias): 16-byte LDS reads straight into the accumulator registers, no VALU work.
// Accumulate (h += b1 of a block's second layer): ONE exact-fp32 MFMA per output block, acc[m] += B_m (x) e, with the
// bias column as the A operand of a 32x32x2 product (k = 0: the 32 biases of the block's rows, k = 1: multiplied by zero)
// against B = 1 for k = 0.  bias x 1.0 is exact and is added with one rounding: bit-identical to the 16 v_add per block it
// replaces, on the matrix pipe instead of the VALU (the chunk phases of the fused kernels are issue-bound on VALU work;
// A/B in profiles/r02_ab_variants.txt).  Row i of block m holds logical feature 16*MB*hh' + 16*m + 4*(i>>3) + (i&3) with
// hh' = (i>>2)&1 (the accumulator layout of the 32x32 MFMAs: lane half hh owns rows 8*(r>>2) + 4*hh + (r&3)).
template <int MB, bool ASSIGN, int PREC = PREC_F16F6, bool SPLIT = false>
__device__ __forceinline__ void bias_init(const float* __restrict__ bl, int hh, f32x16 (&acc)[MB]) {
#ifdef NJF_ABLATE_BIAS  // experiment builds only
  if (ASSIGN)
    for (int m = 0; m < MB; ++m) acc[m] = (f32x16)(0.f);
  return;
#endif
  // the MFMA form only where the matrix pipe has the headroom (the fp6-corrected chunks); the f16x2 chunks are bound by
  // their 48 matrix instructions (proposal pass: -0.5...1 % with v_add, profiles/r02_ab_variants.txt)
  constexpr bool VALU = ASSIGN || (PREC != PREC_F16F6 && PREC != PREC_F16);
  if constexpr (!ASSIGN && PREC == PREC_F16 && SPLIT) {   // (whatever the A/B flags above say: the table holds bit patterns)
    // plain-fp16 networks: the accumulated biases are packed as {fp16 hi, fp16 lo} pairs (njf_pack_resnetfc: bias = hi + lo to
    // 22 bits) and added by ONE f16 MFMA per output block -- A = [hi, lo, 0 ...] for k = 0, 1 (lane half 0), B = [1, 1, 0 ...]:
    // hi * 1 + lo * 1 accumulated in fp32, 32 clocks instead of the 64 of the exact-fp32 form below (the chunk phases of these
    // kernels run at the matrix pipe's rate: tools/stamps.py, profiles/r05_stamps_f16.txt)
    const int i = threadIdx.x & 31;
    const float* b = bl + 16 * MB * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{hh == 0 ? 0x3c003c00u : 0u, 0u, 0u, 0u});
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const unsigned u = hh == 0 ? __float_as_uint(b[16 * m]) : 0u;
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, u32x4{u, 0u, 0u, 0u}), ones, acc[m], 0, 0, 0);
    }
  } else if constexpr (VALU) {
    const float* b = bl + 16 * MB * hh;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *(const f32x4*)(b + 16 * m + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ASSIGN) acc[m][4 * q + e] = v[e];
          else acc[m][4 * q + e] += v[e];
        }
      }
    }
  } else {
    const int i = threadIdx.x & 31;
    const float one = hh == 0 ? 1.f : 0.f;
    const float* b = bl + 16 * MB * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float a = b[16 * m];
      if (ASSIGN) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, one, (f32x16)(0.f), 0, 0, 0);
      else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, one, acc[m], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Per-point geometry: world -> context camera -> bilinear footprint in the hoisted feature map.
// Mirrors get_pixel_aligned_features (model_components/pixel_aligned_features.py:11-35) and ATen's
// grid_sampler_2d(bilinear, border, align_corners=True).  The 4-term dot products are fma chains
// in k order: that is what torch.einsum("...ij,...j->...i") lowers to on CPU, so camera-space
// coordinates match the oracle bit for bit (they feed a positional encoding that amplifies ulps).
// ------------------------------------------------------------------------------------------
struct CamCtx;
// Camera-space position of a point (the positional-encoding input) + what its bilinear footprint in the hoisted map is
// computed from.  The footprint itself (4 texel offsets + 4 weights) is NOT kept: point_footprint() recomputes it in
// front of each of the three gathers of a ResnetFC (~25 VALU instructions) instead of holding 8 registers live across
// the 22 weight chunks -- the fused kernels run at the 256-register limit and every long-lived value is a spill.
struct PointGeom {
  float xc, yc, zc;
  const CamCtx* cam;  // intrinsics (wave-uniform in the ray kernels: SGPRs)
  int hf, wf, stride;
  unsigned gofs;      // float index of the point's batch element in the hoisted map (`gz` of the gathers is the map itself:
                      // the lanes of a quad fetch each other's texels, and a tile of njf_points_forward may mix batch elements)
};
struct Footprint {
  int t00, t01, t10, t11;  // texel offsets (floats) into the feature map of this batch element
  float w00, w01, w10, w11;
};

struct CamCtx {
  float m[12];  // rows 0..2 of ctxt_w2c
  float k[9];
};

__device__ __forceinline__ void load_ctx(const float* __restrict__ w2c, const float* __restrict__ k, int b,
                                         CamCtx& c) {
#pragma unroll
  for (int i = 0; i < 12; ++i) c.m[i] = w2c[b * 16 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.k[i] = k[b * 9 + i];
}

__device__ __forceinline__ float dot4_h(const float* r, float x, float y, float z) {
  float a = r[0] * x;
  a = fmaf(r[1], y, a);
  a = fmaf(r[2], z, a);
  a = fmaf(r[3], 1.0f, a);
  return a;
}

__device__ __forceinline__ float dot3(const float* r, float x, float y, float z) {
  float a = r[0] * x;
  a = fmaf(r[1], y, a);
  a = fmaf(r[2], z, a);
  return a;
}

__device__ __forceinline__ void point_geometry(const CamCtx& c, float px, float py, float pz, int hf, int wf,
                                               int stride, unsigned gofs, PointGeom& g) {
  g.gofs = gofs;
  g.xc = dot4_h(c.m + 0, px, py, pz);
  g.yc = dot4_h(c.m + 4, px, py, pz);
  g.zc = dot4_h(c.m + 8, px, py, pz);
  g.cam = &c;
  g.hf = hf;
  g.wf = wf;
  g.stride = stride;
}

__device__ __forceinline__ void point_footprint(const PointGeom& g, Footprint& f) {
  float xc = g.xc, yc = g.yc, zc = g.zc;
  // opaque copies: without them the compiler merges the recomputations into one and keeps its 8 results alive
  asm volatile("" : "+v"(xc), "+v"(yc), "+v"(zc));
  const CamCtx& c = *g.cam;
  const int hf = g.hf, wf = g.wf, stride = g.stride;
  const float u0 = dot3(c.k + 0, xc, yc, zc);
  const float u1 = dot3(c.k + 3, xc, yc, zc);
  const float u2 = dot3(c.k + 6, xc, yc, zc);
  const float den = u2 + 1e-9f;
  // one v_rcp_f32 (1 ulp) and two multiplications: the footprint only feeds the bilinear weights and texel indices -- smooth
  // in uv, unlike the camera-space coordinates above, whose bits the positional encoding amplifies -- and it is recomputed
  // in front of each of the six gathers of a tile
  const float inv = __builtin_amdgcn_rcpf(den);
  const float u = u0 * inv, v = u1 * inv;
  const float gx = (u - 0.5f) * 2.0f, gy = (v - 0.5f) * 2.0f;
  float ix = ((gx + 1.0f) / 2.0f) * (float)(wf - 1);
  float iy = ((gy + 1.0f) / 2.0f) * (float)(hf - 1);
  ix = fminf((float)(wf - 1), fmaxf(ix, 0.0f));
  iy = fminf((float)(hf - 1), fmaxf(iy, 0.0f));
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float fx = ix - x0f, fy = iy - y0f;
  const float ex = 1.0f - fx, ey = 1.0f - fy;
  int x0 = (int)x0f, y0 = (int)y0f;
  x0 = min(max(x0, 0), wf - 1);  // also absorbs NaN coordinates (behind-camera points)
  y0 = min(max(y0, 0), hf - 1);
  const int x1 = min(x0 + 1, wf - 1), y1 = min(y0 + 1, hf - 1);
  f.t00 = (y0 * wf + x0) * stride;
  f.t01 = (y0 * wf + x1) * stride;
  f.t10 = (y1 * wf + x0) * stride;
  f.t11 = (y1 * wf + x1) * stride;
  f.w00 = ey * ex;
  f.w01 = ey * fx;
  f.w10 = fy * ex;
  f.w11 = fy * fx;
}

// h += bilerp(G)[32*MB channels starting at `gz`]: the pixel-aligned sampling of the hoisted map
// (model_components/pixel_aligned_features.py:29-33 on G = lin_z(features)).
//
// The "quad" form (F16F6 networks).
// Loads.  The 16*MB floats lane (j, hh) accumulates per texel are CONTIGUOUS in the map (njf_hoist_position, layout 1: logical
// feature f = 16*MB*hh + 16*m + (4*e + i) sits at 16*MB*hh + 16*m + 4*i + e), and they are fetched by the lane's QUAD:
// in load (slot s, segment m) the four lanes 4p..4p+3 read the four 16-byte pieces i = 0..3 of segment m of the texel of
// lane 4p+s -- 64 contiguous bytes per quad.  The texture addresser coalesces adjacent lanes, not the lane pairs 32
// apart that own a point: measured with ray-like texel coherence (tools/probes/probe_gather.hip), a 64-lane dwordx4
// load costs 45 clocks CU-wide when every lane reads its own point's piece, 28 in this form; the gather phase of the
// fused kernels is bound by exactly that rate (four waves of a workgroup reach it together, tools/stamps.py).
//
// Accumulation.  The loaded pieces sit in the wrong lanes (lane 4p+i holds piece i of lane 4p+s's texel); the matrix
// core both moves and accumulates them: v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products, one per quad) with
// A = the loaded dword, B = w_t * [lane % 4 == s], C = D = four accumulator registers gives lane n = s of the quad
// D[i] += piece_i * w_t(own) for i = 0..3 and adds exact zeros in the other three lanes.  That replaces the 64*MB
// v_fmac per gather of the round-1 form (the kernels are VALU-issue bound between the gathers) by 64*MB MFMAs of
// 8 cycles on the otherwise idle fp32 matrix pipe.  The product-sum of the MFMA is not bit-identical to v_fmac (about
// one value in five differs in the last bit); texel order t = 0..3 is kept.
//
// Pipelining.  A batch = (texel t, slot s): MB loads, 4*MB MFMAs.  DEPTH batches of loads are in flight (the registers
// `net` vacates at this point of the block); sched_barrier pins the issue order, otherwise the scheduler hoists all 16*MB
// loads (-> scratch) or serialises them.
template <int S>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned quad_bcast_n(unsigned v, int s) {  // s is a compile-time constant after unrolling
  return s == 0 ? quad_bcast<0>(v) : (s == 1 ? quad_bcast<1>(v) : (s == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
}

template <int MB, int DEPTH = 4>
__device__ __forceinline__ void add_hoisted_latent_quad(const float* __restrict__ gz, const PointGeom& g, int lane,
                                                        f32x16 (&h)[MB]) {
#ifdef NJF_ABLATE_GATHER  // experiment builds only (tools/ablate.sh)
  return;
#endif
  Footprint f;
  point_footprint(g, f);
  const int c = lane & 3;
  const unsigned lane_off = 16 * MB * (lane >> 5) + 4 * c;
  // float index of this lane's four texels in the map (gofs: its batch element); a quad exchanges them by DPP
  const unsigned gofs = g.gofs;
  const unsigned own[4] = {gofs + (unsigned)f.t00, gofs + (unsigned)f.t01, gofs + (unsigned)f.t10, gofs + (unsigned)f.t11};
  const float w[4] = {f.w00, f.w01, f.w10, f.w11};
  f32x4 x[DEPTH][MB];
  auto issue = [&](int b) {
    const float* src = gz + (size_t)(quad_bcast_n(own[b >> 2], b & 3) + lane_off);
#pragma unroll
    for (int m = 0; m < MB; ++m) x[b % DEPTH][m] = *(const f32x4*)(src + 16 * m);
  };
#pragma unroll
  for (int b = 0; b < DEPTH; ++b) issue(b);
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const float wb = c == (b & 3) ? w[b >> 2] : 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 d = {h[m][4 * e], h[m][4 * e + 1], h[m][4 * e + 2], h[m][4 * e + 3]};
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(x[b % DEPTH][m][e], wb, d, 0, 0, 0);
        h[m][4 * e] = d[0];
        h[m][4 * e + 1] = d[1];
        h[m][4 * e + 2] = d[2];
        h[m][4 * e + 3] = d[3];
      }
    if (b + DEPTH < 16) issue(b + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
  }
}


// The "half" form (every precision but F16F6): each lane fetches its own point's pieces and folds them with v_fmac.
// Within a block of 32*MB channels the map stores logical feature f = 16*MB*hh + 16*m + 4*q + e at position
// 32*m + 8*q + 4*hh + e (njf_hoist_position, layout 0), so the two lanes that own a point read ADJACENT 16-byte
// pieces in the same instruction.
template <int MB, int DEPTH = 4>
__device__ __forceinline__ void add_hoisted_latent_half(const float* __restrict__ gz, const PointGeom& g, int hh,
                                                        f32x16 (&h)[MB]) {
#ifdef NJF_ABLATE_GATHER  // experiment builds only (tools/ablate.sh)
  return;
#endif
  Footprint f;
  point_footprint(g, f);
  const float* gb = gz + (size_t)g.gofs + 4 * hh;
  const float* p[4] = {gb + f.t00, gb + f.t01, gb + f.t10, gb + f.t11};
  const float w[4] = {f.w00, f.w01, f.w10, f.w11};
  // A rolling pipeline like the quad form's: a batch = (texel t, block m) = 4 loads of 16 bytes; DEPTH batches are in
  // flight (the 64 registers `net` vacates at this point of the block), and every consumed batch is refilled at once, so
  // the addresser sees a continuous stream (the round-1 form issued a texel's 4*MB loads, waited for all of them, folded
  // them, and only then issued the next texel's; measured equal in round 2 -- this form's gather is bound by the addresser's
  // request rate, not by the exposed round trips -- and kept as the one structure both forms share).  sched_barrier pins
  // the issue order (otherwise the scheduler hoists all 16*MB loads -> scratch).
  // Accumulation order per element stays t = 0..3: bit-identical results.
  constexpr int NB = 4 * MB;
  f32x4 v[DEPTH][4];
  auto issue = [&](int b) {
    const float* src = p[b / MB] + 32 * (b % MB);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[b % DEPTH][q] = *(const f32x4*)(src + 8 * q);
  };
#pragma unroll
  for (int b = 0; b < DEPTH; ++b) issue(b);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int t = b / MB, m = b % MB;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // Texels 1..3: one v_fmac per value, written as asm so that the SLP vectoriser cannot pair them into
        // v_pk_fma_f32 -- the packed form needs every bilinear weight as a {w, w} register pair, which the
        // allocator materialises per use site and spills.  Texel 0 stays a compiler-visible fmaf: h was just
        // written by MFMAs, and the MFMA-write -> VALU-read wait states are only inserted for instructions the
        // hazard recogniser can see, never for inline asm.  Every register is therefore first read by a visible
        // instruction, and its asm updates depend on that result.  (A cheaper "touch one register per
        // accumulator block" was tried and is WRONG: the scheduler moves the other registers' asm above the
        // touch -- caught by the transformer-head golden tests.)
        if (t == 0) {
          h[m][4 * q + e] = fmaf(v[b % DEPTH][q][e], w[0], h[m][4 * q + e]);
        } else {
          float acc = h[m][4 * q + e];
          asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v[b % DEPTH][q][e]), "v"(w[t]));
          h[m][4 * q + e] = acc;
        }
      }
    asm volatile("" : "+v"(h[m]) : : "memory");  // the fmas retire into h before the batch's registers are reloaded
    if (b + DEPTH < NB) issue(b + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The fp16-map form (PREC_F16 networks): the hoisted map itself is stored in fp16 (njf_project_* with NJF_PRECISION_F16
// round the projection's fp32 result once), so a point's footprint is half the bytes and half the load instructions -- the
// gathers are bound by the texture addresser's request rate (see above).  Within a block of 32*MB channels logical feature
// f = 16*MB*hh + 16*m + 8*q + e sits at position 32*m + 16*q + 8*hh + e (njf_hoist_position, layout 2): the two lanes that
// own a point read ADJACENT 16-byte pieces (8 channels each) in the same instruction.  Each value is folded with ONE
// v_fma_mix_f32 (fp16 source, fp32 weight and accumulator): the bilinear interpolation itself is carried out in fp32.
#define NJF_F16_GATHER_DEPTH 8
template <int MB, int DEPTH = NJF_F16_GATHER_DEPTH>
__device__ __forceinline__ void add_hoisted_latent_f16(const _Float16* __restrict__ gz, const PointGeom& g, int hh,
                                                       f32x16 (&h)[MB], const Footprint* shared = nullptr) {
#ifdef NJF_ABLATE_GATHER  // experiment builds only (tools/ablate.sh)
  return;
#endif
  Footprint f;
  if (shared != nullptr) f = *shared;   // (compile-time known at every call site: TileShareF16)
  else point_footprint(g, f);
  const _Float16* gb = gz + (size_t)g.gofs + 8 * hh;
  const _Float16* p[4] = {gb + f.t00, gb + f.t01, gb + f.t10, gb + f.t11};
  const float w[4] = {f.w00, f.w01, f.w10, f.w11};
  // a batch = (texel t, block m) = 2 loads of 16 bytes (16 channels); DEPTH batches in flight, refilled as they are consumed
  constexpr int NB = 4 * MB;
  constexpr int D = DEPTH < NB ? DEPTH : NB;
  f16x8 v[D][2];
  auto issue = [&](int b) {
    const _Float16* src = p[b / MB] + 32 * (b % MB);
#pragma unroll
    for (int q = 0; q < 2; ++q) v[b % D][q] = *(const f16x8*)(src + 16 * q);
  };
#pragma unroll
  for (int b = 0; b < D; ++b) issue(b);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int t = b / MB, m = b % MB;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) h[m][8 * q + e] = fmaf((float)v[b % D][q][e], w[t], h[m][8 * q + e]);
    asm volatile("" : "+v"(h[m]) : : "memory");  // the fmas retire into h before the batch's registers are reloaded
    if (b + D < NB) issue(b + D);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// PREC_F16, inference: what the ResnetFC networks of ONE tile share.  The density network and the Jacobian head encode the SAME
// camera-space point and gather at the SAME footprint (and the three gathers of a network share it anyway).  pe: this lane's 64
// encoding values as the lin_in chunk's fp16 B operands (4 K-steps x 4 dwords): 16 registers held across the tile instead of a
// second positional encoding (~280 VALU + 60 v_sin); and within a network ONE footprint serves its three gathers (resnet_tile).
// The plain-fp16 kernels have the registers (232 of 256) and their time follows their instruction count
// (profiles/r05_ablate_f16.txt, blocks h-j); the other precisions run at the register limit and keep recomputing.
struct TileShareF16 {
  unsigned pe[16];
};

__device__ __forceinline__ void share_tile(const f32x16 (&pe)[2], TileShareF16& sh) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int p = 0; p < 4; ++p) sh.pe[4 * t + p] = pack_pair_f16<false>(pe[t >> 1][8 * (t & 1) + 2 * p], pe[t >> 1][8 * (t & 1) + 2 * p + 1]);
}

// lin_in of a PREC_F16 ResnetFC from the packed encoding: mma_chunk<PREC_F16, 4, 2, 0, false, 2> without its conversions
template <class ST>
__device__ __forceinline__ void mma_lin_in_f16_packed(ST& st, const float* __restrict__ wl, int lane, const unsigned (&pk)[16],
                                                      f32x16 (&out)[4]) {
  const f16x8* base = (const f16x8*)wl + lane;
  constexpr int T = 4;
  f16x8 a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = base[i * 64];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f16x8 n[4] = {a[0], a[1], a[2], a[3]};
    if (t + 1 < T) {
#pragma unroll
      for (int i = 0; i < 4; ++i) n[i] = base[((t + 1) * 4 + i) * 64];
    }
    dma_issue(st, 2 * t);
    dma_issue(st, 2 * t + 1);
    __builtin_amdgcn_sched_barrier(0);
    const f16x8 b = __builtin_bit_cast(f16x8, u32x4{pk[4 * t], pk[4 * t + 1], pk[4 * t + 2], pk[4 * t + 3]});
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b, out[m], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = n[i];
  }
}


// Which form a network uses follows its MFMA precision (and so does the layout its lin_z columns are packed in,
// njf_hoist_layout): the quad/MFMA form where the matrix pipe has headroom -- the fp6-corrected networks, measured
// -3.8 % on the C2 final pass -- and the half/VALU form where it is the busier pipe (F16X2: the proposal pass measured
// +5 % with the quad form; F32: the matrix pipe is the bound).
template <int MB, int PREC, int DEPTH = 4>
__device__ __forceinline__ void add_hoisted_latent(const float* __restrict__ gz, const PointGeom& g, int lane,
                                                   f32x16 (&h)[MB]) {
  if constexpr (PREC == PREC_F16) add_hoisted_latent_f16<MB>((const _Float16*)gz, g, lane >> 5, h);
  else if constexpr (PREC == PREC_F16F6) add_hoisted_latent_quad<MB, DEPTH>(gz, g, lane, h);
  else add_hoisted_latent_half<MB, DEPTH>(gz, g, lane >> 5, h);
}

// Address arithmetic on a hoisted map in ELEMENTS (fp32 maps: floats; PREC_F16 maps: halves).  The fused kernels carry map
// pointers as `const float*` whatever the element type; every offset goes through here.
template <int PREC>
__device__ __forceinline__ const float* map_at(const float* base, size_t elements) {
  if constexpr (PREC == PREC_F16) return (const float*)((const _Float16*)base + elements);
  else return base + elements;
}

// ------------------------------------------------------------------------------------------
// sin(arg) with exact range reduction (arguments reach ~3e4 rad at the top octave of the positional
// encoding; hardware v_sin_f32 and __sinf are not accurate enough there).  The reduction works in revolutions,
// arg / 2pi evaluated as a two-float product: 1/2pi = C_HI + C_LO (24 + 24 bits), u = fl(arg * C_HI), the exact
// rounding error of that product from one fma, plus arg * C_LO.  u - rint(u) is exact, so the phase is good to
// half an ulp of 0.5 (2.8e-8 revolutions, sin error 1.8e-7) -- the same as rounding an fp64 reduction to float, at
// a third of the cost (fp64 VALU runs at half rate and needs conversions both ways).  Round 3: the sine of the reduced
// phase is the hardware's v_sin_f32 (which works in revolutions) instead of a polynomial.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sin_accurate(float arg) {
  const float C_HI = 0.15915493667125702f, C_LO = 6.4206382432985265e-09f;
  const float u = arg * C_HI;
  float e = fmaf(arg, C_HI, -u);
  e = fmaf(arg, C_LO, e);
  const float y = (u - rintf(u)) + e;                    // [-0.5, 0.5] (+ rounding)
  // v_sin_f32 takes its argument in REVOLUTIONS, which is what the reduction above produces: on [-0.5, 0.5] it measures
  // 1.25e-7 max abs error against sin(2 pi y) in double (tools/probes/probe_sin.hip; the degree-13 polynomial with its
  // quadrant fold it replaces: 1.93e-7) for one quarter-rate instruction instead of 13 full-rate ones -- the positional
  // encoding is 1,200 of the render kernel's 7,700 VALU instructions per tile (profiles/r03_isa_budget.txt).
  return __builtin_amdgcn_sinf(y);
}

// Positional encoding in B-operand slot order (see njf_pack: kind 1).  Lane half hh=0 supplies
// [sin(s_{d,f}) (30) | x | y], hh=1 supplies [sin(s_{d,f} + pi/2) (30) | z | 1].
// s = fl(fl(2*pi)*x) * 2^f exactly as nerfstudio's NeRFEncoding computes it in fp32.
__device__ __forceinline__ void positional_encoding(float xc, float yc, float zc, int hh, f32x16 (&pe)[2]) {
#ifdef NJF_ABLATE_PE  // experiment builds only
  pe[0] = (f32x16)(xc);
  pe[1] = (f32x16)(yc + zc);
  return;
#endif
  const float two_pi = 6.2831855f;
  const float half_pi = hh ? 1.5707964f : 0.0f;
  const float sx[3] = {two_pi * xc, two_pi * yc, two_pi * zc};
#pragma unroll
  for (int s = 0; s < 30; ++s) {
    const int d = s / 10, f = s % 10;
    const float arg = sx[d] * (float)(1 << f) + half_pi;
    pe[s >> 4][s & 15] = sin_accurate(arg);
  }
  pe[1][14] = hh ? zc : xc;
  pe[1][15] = hh ? 1.0f : yc;
}

// 16 real spherical harmonics (degree 4) of v = 2*((d+1)/2) - 1, tiny-cuda-nn sign convention.
__device__ __forceinline__ void sh4(float dx, float dy, float dz, float (&o)[16]) {
  const float x = ((dx + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float y = ((dy + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float z = ((dz + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// ------------------------------------------------------------------------------------------
// One ResnetFC (resnet_fc.py:130-154) on a 32-point tile: 22 weight chunks.
// bias layout (LDS): [blk: fc0 (128) | fc1 (128)] x 5 | lin_out (32).
// ------------------------------------------------------------------------------------------
// Activation dump for the backward pass (training only): the ReLU'd input of every layer of the net, written in
// logical feature order as [layer][point][128] (+ the 64-slot positional encoding).  `dump` addresses layer 0 of
// this lane's point (+ 64*hh), `stride` = floats between layers; nullptr lanes (padding samples) skip the stores.
struct ActDump {
  float* act;     // [11][P][128] : r0_b = 2b, r1_b = 2b+1 (b = 0..4), r_out = 10
  float* pe;      // [P][64] slot order
  size_t stride;  // P * 128
  // ReLU masks (round 6): [11][P][4] words, this lane's two words of its point (+ 2*hh); layer stride = P * 4.  What the fused
  // backward chain reads instead of the activations themselves (njf_resnetfc_backward), nullptr = not dumped
  unsigned* mask = nullptr;
  // 16-bit training storage (round 6, opt-in): `act` addresses HALVES ([11][P][128] fp16: what the weight-gradient GEMM reads with
  // fp32 accumulation -- the reference trains on TF32 products, 10 mantissa bits as well); the stride stays a count of elements
  bool half = false;
  // ResnetFC.forward(compute_features=True) (resnet_fc.py:141-151; ABI v18): the residual stream AFTER each block, [5][P][128] fp32,
  // this lane's 64 values of block 0 (layer stride = `stride`); nullptr = not dumped.  Only the point-query kernel asks for it
  // (the 640 hidden "action features" of the flow_mlp decoder, action_decoder_flow.py:168-176)
  float* feat = nullptr;
};

// this lane's slot of layer `l` in the activation dump (elements are floats, or halves when dump.half)
__device__ __forceinline__ float* dump_layer(const ActDump& d, int l) {
  if (d.act == nullptr) return nullptr;
  return d.half ? (float*)((_Float16*)d.act + (size_t)l * d.stride) : d.act + (size_t)l * d.stride;
}

template <bool DO_RELU>
__device__ __forceinline__ void dump_vec128(float* __restrict__ dst, const f32x16 (&v)[4], unsigned* __restrict__ mask = nullptr,
                                            bool half = false) {
  if (dst == nullptr) return;
  if (half) {   // (wave-uniform) 8 stores of 16 bytes: this lane's 64 values as fp16
    _Float16* d16 = (_Float16*)dst;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_pair_f16<DO_RELU>(v[m][8 * q + 2 * e], v[m][8 * q + 2 * e + 1]);
        *(u32x4*)(d16 + 16 * m + 8 * q) = o;
      }
  } else {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = DO_RELU ? fmaxf(v[m][4 * q + e], 0.f) : v[m][4 * q + e];
      *(f32x4*)(dst + 16 * m + 4 * q) = o;
    }
  }
  if (mask != nullptr) {
    // bit 16*(m & 1) + r of word (m >> 1) = [v[m][r] > 0]: the ReLU mask of this lane's 64 features (8 bytes per lane and layer
    // instead of the 256 the backward chain used to read back: its loads were 1.05 of its 2.81 ms, profiles/r06_training_c4.json)
    unsigned w[2] = {0u, 0u};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // [x > 0] as clamp(bits(x), 0, 1): positive floats are positive integers, zeros and negatives are not (one v_med3_i32),
        // shifted into place and merged by one v_lshl_or_b32 -- two VALU instructions per value (NaNs with a clear sign bit count
        // as positive; the ReLU'd value the same lane dumps for the weight-gradient GEMM is NaN then anyway)
        const unsigned b = (unsigned)min(max(__float_as_int(v[m][r]), 0), 1);
        w[m >> 1] |= b << (16 * (m & 1) + r);
      }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *(u32x2*)mask = u32x2{w[0], w[1]};
  }
}

// SHARED (PREC_F16 inference only), bits: 1 = lin_in reads the packed encoding of `share` instead of `pe` (which is then not
// read); 2 = ONE footprint, computed here, serves the network's three gathers (8 registers held across blocks 0-2).
template <int PREC, bool DUMP = false, int SHARED = 0, class ST>
__device__ __forceinline__ void resnet_tile(ST& st, const float* __restrict__ bias,
                                            const float* __restrict__ gz, const PointGeom& g,
                                            const f32x16 (&pe)[2], int wave, int lane, f32x16 (&out)[1],
                                            ActDump dump = ActDump{nullptr, nullptr, 0}, const TileShareF16* share = nullptr) {
  static_assert(SHARED == 0 || (PREC == PREC_F16 && !DUMP), "the shared tile state exists for the plain-fp16 inference kernels");
  const int hh = lane >> 5;
  f32x16 h[4], net[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) h[m] = (f32x16)(0.f);
  if (DUMP && dump.pe != nullptr) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pe[kb][4 * q + e];
        *(f32x4*)(dump.pe + 16 * kb + 4 * q) = o;
      }
  }
  {
    const float* wl = stream_step(st, wave, lane);
    if constexpr ((SHARED & 1) != 0) mma_lin_in_f16_packed(st, wl, lane, share->pe, h);
    else mma_chunk<PREC, 4, 2, 0, false, 2>(st, wl, lane, pe, h);  // lin_in (bias folded into slot 63)
  }
  if constexpr (PREC == PREC_F16) {
    // 12 chunks: lin_in | (fc_0, fc_1) x 5, a whole 128 x 128 layer per chunk | lin_out
    static_assert(!DUMP, "the plain-fp16 mode is an inference mode (training forwards dump fp32-class activations)");
    Footprint fp_net;
    if constexpr ((SHARED & 2) != 0) point_footprint(g, fp_net);
    for (int blk = 0; blk < 5; ++blk) {
      if (blk < 3) {
        NJF_STAMP(st, 4);
        if constexpr ((SHARED & 2) != 0) add_hoisted_latent_f16<4>((const _Float16*)map_at<PREC>(gz, blk * 128), g, hh, h, &fp_net);
        else add_hoisted_latent<4, PREC>(map_at<PREC>(gz, blk * 128), g, lane, h);
        NJF_STAMP(st, 5);
      }
      const float* bl = bias + blk * 256;
      bias_init<4, true, PREC>(bl, hh, net);
      {
        const float* wl = stream_step(st, wave, lane);
        mma_chunk<PREC, 4, 4, 0, true, 4>(st, wl, lane, h, net);
      }
      if (blk >= 2) bias_init<4, false, PREC, true>(bl + 128, hh, h);   // (blocks 0, 1: folded into the next latent, njf_pack_resnetfc)
      {
        const float* wl = stream_step(st, wave, lane);
        mma_chunk<PREC, 4, 4, 0, true, 4>(st, wl, lane, net, h);
      }
    }
  } else
  for (int blk = 0; blk < 5; ++blk) {
    if (blk < 3) {
      NJF_STAMP(st, 4);  // gather begins (the stamp's own lgkmcnt(0) also ends the previous chunk's MFMA issue)
      add_hoisted_latent<4, PREC>(gz + blk * 128, g, lane, h);
      NJF_STAMP(st, 5);  // gather folded into h
    }
    if (DUMP) dump_vec128<true>(dump_layer(dump, 2 * blk), h, dump.mask ? dump.mask + (size_t)(2 * blk) * (dump.stride / 32) : nullptr, dump.half);
    const float* bl = bias + blk * 256;
    bias_init<4, true, PREC>(bl, hh, net);
    {
      {
        const float* wl = stream_step(st, wave, lane);
        mma_chunk<PREC, 4, 2, 0, true, 4>(st, wl, lane, h, net);
      }
      {
        const float* wl = stream_step(st, wave, lane);
        mma_chunk<PREC, 4, 2, 2, true, 4>(st, wl, lane, h, net);
      }
    }
    if (DUMP) dump_vec128<true>(dump_layer(dump, 2 * blk + 1), net, dump.mask ? dump.mask + (size_t)(2 * blk + 1) * (dump.stride / 32) : nullptr, dump.half);
    bias_init<4, false, PREC>(bl + 128, hh, h);
    {
      {
        const float* wl = stream_step(st, wave, lane);
        mma_chunk<PREC, 4, 2, 0, true, 4>(st, wl, lane, net, h);
      }
      {
        const float* wl = stream_step(st, wave, lane);
        mma_chunk<PREC, 4, 2, 2, true, 4>(st, wl, lane, net, h);
      }
    }
    if (DUMP && dump.feat != nullptr) dump_vec128<false>(dump.feat + (size_t)blk * dump.stride, h);
  }
  if (DUMP) dump_vec128<true>(dump_layer(dump, 10), h, dump.mask ? dump.mask + (size_t)10 * (dump.stride / 32) : nullptr, dump.half);
  bias_init<1, true, PREC>(bias + 1280, hh, out);
  {
    const float* wl = stream_step(st, wave, lane);
    mma_chunk<PREC, 1, 4, 0, true, 4>(st, wl, lane, h, out);
  }
}

// Colour-head dump for the perception-mode backward pass: `in` addresses this lane's 16 inputs ([P][32]: geo15 | 1 |
// sh16), `act` its 32 ReLU'd hidden values of layer 1 ([2][P][64], `stride` = P*64 floats to layer 2's input).
struct ColorDump {
  float* in;
  float* act;
  size_t stride;
};

template <bool DO_RELU>
__device__ __forceinline__ void dump_vec32(float* __restrict__ dst, const f32x16 (&v)[2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = DO_RELU ? fmaxf(v[m][4 * q + e], 0.f) : v[m][4 * q + e];
      *(f32x4*)(dst + 16 * m + 4 * q) = o;
    }
}

// colour head (action_decoder_jacobian.py:315-322): one chunk [L0 2048 | L1 4096 | L2 2048],
// bias (LDS): [L1 (64) | L2 (32)].  cin: hh=0 -> [geo(15), 1], hh=1 -> sh(16).
template <int PREC, bool DUMP = false, class ST>
__device__ __forceinline__ void color_tile(ST& st, const float* __restrict__ bias, const f32x16 (&cin)[1],
                                           int wave, int lane, f32x16 (&rgb)[1],
                                           ColorDump dump = ColorDump{nullptr, nullptr, 0}) {
  const int hh = lane >> 5;
  const float* wl = stream_step(st, wave, lane);
  f32x16 a[2], b[2];
  a[0] = (f32x16)(0.f);
  a[1] = (f32x16)(0.f);
  mma_chunk<PREC, 2, 1, 0, false, 1>(st, wl, lane, cin, a);
  bias_init<2, true, PREC>(bias, hh, b);
  mma_chunk<PREC, 2, 2, 0, true, 2>(st, wl + 2048, lane, a, b);
  bias_init<1, true, PREC>(bias + 64, hh, rgb);
  mma_chunk<PREC, 1, 2, 0, true, 2>(st, wl + 6144, lane, b, rgb);
  if (DUMP && dump.in != nullptr) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = cin[0][4 * q + e];
      *(f32x4*)(dump.in + 4 * q) = o;
    }
    dump_vec32<true>(dump.act, a);
    dump_vec32<true>(dump.act + dump.stride, b);
  }
}

// ------------------------------------------------------------------------------------------
// Jacobian transformer head (action_decoder_jacobian.py:418-446 + model_components/transformer.py:85-135),
// algebraically folded on the host (decoder.py::ActionDecoderJacobianTransformer.packed):
//   x0 = W_pe*pe + bilerp(G_q)                      (jacobian_query_mlp split into PE part + hoisted feature part)
//   per layer:  n = norm(x);  dots = Mqk*n + bqk    (to_q, K=to_kv(z)[:, :512], LayerNorm affine and the softmax
//               a = softmax_8(dots)                  scale folded into one 64x64 matrix; rows ordered head*8 + key)
//               x += Nov*a + bo                      (V and to_out folded into one 64x64 matrix)
//               n = norm(x);  x += W2*gelu(W1'*n + b1') + b2
//   J = Wj*x + bj
// Weight stream: 14 half-chunks of 4096 floats [query | (Mqk, Nov, W1', W2) x 3 | head] = 7 chunks.
// bias (LDS): [bqk | bo | b1' | b2] (64 each) x 3 | head (32).
// ------------------------------------------------------------------------------------------
// erf(a) without a branch: both polynomial pieces of the classic single-precision evaluation (split at |a| = 0.927734375:
// a * P(a^2) below, 1 - exp(Q(|a|)) above; each piece is within 1 ulp of erf) are computed and one is selected.  The device
// library's erff branches per element; in a tile whose 32 lanes x 32 values straddle the split both sides of all 32 branches run
// (~50 VALU instructions + the exec-mask bookkeeping per value: ~1,600 per layer -- the largest item of the head's instruction
// stream, profiles/r06_transformer_valu_budget.txt).  The exponential is the hardware's v_exp_f32 on the argument scaled by log2(e):
// the argument is in [-17, -0.8], so the scaling's rounding costs <= 1e-6 relative on a term that is <= 0.19 of the result.
__device__ __forceinline__ float erf_branchless(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float hi = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float lo = fmaf(q, a, a);
  return t > 0.927734375f ? hi : lo;
}

// The same function to 5.2e-7 ABSOLUTE (Abramowitz & Stegun 7.1.26 evaluated in fp32: 1.5e-7 of the formula + the cancellation of 1 - p e near 0) in 16
// instead of 29 VALU instructions: one v_rcp_f32, five fmas, one v_exp_f32.  GELU reads 1 + erf, so an absolute bound is a relative
// bound on its result; used by the precisions whose own product error is above it (f16x2: 4e-7 per network, f16f6: 1.5e-5, f16:
// 5e-4), never by the exact-fp32 mode.
__device__ __forceinline__ float erf_as7126(float a) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(a), 1.0f));
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f((a * a) * -1.4426950408889634f);
  return copysignf(fmaf(-p, e, 1.0f), a);
}

__device__ __forceinline__ void norm64(const f32x16 (&x)[2], f32x16 (&n)[2]) {
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[m][r];
  s += __shfl_xor(s, 32, 64);
  const float mean = s / 64.0f;
  float v = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = x[m][r] - mean;
      v = fmaf(d, d, v);
    }
  v += __shfl_xor(v, 32, 64);
  // v_rsq_f32 (1 ulp) instead of an IEEE square root and an IEEE division (~25 VALU instructions, six times per tile)
  const float rstd = __builtin_amdgcn_rsqf(v / 64.0f + 1e-5f);
  const float shift = -mean * rstd;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) n[m][r] = fmaf(x[m][r], rstd, shift);   // (x - mean) * rstd as one fma
}

// this lane's 32 values of a 64-wide tile (logical features 32*hh + 16*m + r) <-> dst / src = row + 32*hh
__device__ __forceinline__ void store_vec64(float* __restrict__ dst, const f32x16 (&v)[2]) {
  if (dst == nullptr) return;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[m][4 * q + e];
      *(f32x4*)(dst + 16 * m + 4 * q) = o;
    }
}
__device__ __forceinline__ void load_vec64(const float* __restrict__ src, bool ok, f32x16 (&v)[2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (ok) o = *(const f32x4*)(src + 16 * m + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[m][4 * q + e] = o[e];
    }
}

// `xdump` (training forward, action mode: the head's backward pass, njf_transformer_backward): this lane's slot of slice 0 of
// [4][P][64] -- the residual stream IN FRONT of each of the three layers and behind the last one; `xstride` = P * 64; nullptr = no dump
template <int PREC, class ST>
__device__ __forceinline__ void transformer_tile(ST& st, const float* __restrict__ bias,
                                                 const float* __restrict__ gq, const PointGeom& g,
                                                 const f32x16 (&pe)[2], int keys, int wave, int lane, f32x16 (&out)[1],
                                                 float* __restrict__ xdump = nullptr, size_t xstride = 0) {
  const int hh = lane >> 5;
  f32x16 x[2], n[2], t[2];
  x[0] = (f32x16)(0.f);
  x[1] = (f32x16)(0.f);
  const float* wl = stream_step(st, wave, lane);
  mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, pe, x);  // query MLP, PE part (bias in slot 63)
  add_hoisted_latent<2, PREC>(gq, g, lane, x);    // query MLP, feature part (hoisted)
  for (int l = 0; l < 3; ++l) {
    const float* bl = bias + 256 * l;
    if (xdump != nullptr) store_vec64(xdump + (size_t)l * xstride, x);
    norm64(x, n);
    bias_init<2, true, PREC>(bl, hh, t);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl + 4096, lane, n, t);  // dots[head*8 + key]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        float mx = -3.0e38f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
          if (a < keys) mx = fmaxf(mx, t[m][8 * h8 + a]);
        // exp(d) = v_exp_f32(d * log2 e), d <= 0 (1 ulp of the hardware + |d| * 6e-8 relative from the scaling, on terms that are
        // e^d of the row's largest); one v_rcp_f32 (1 ulp) and eight multiplications instead of eight IEEE divisions: the 32
        // softmax rows of a tile were ~630 VALU instructions per layer, ~190 now
        float e[8], sum = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          e[a] = (a < keys) ? __builtin_amdgcn_exp2f((t[m][8 * h8 + a] - mx) * 1.4426950408889634f) : 0.f;
          sum += e[a];
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int a = 0; a < 8; ++a) t[m][8 * h8 + a] = e[a] * inv;
      }
    }
    wl = stream_step(st, wave, lane);
    bias_init<2, false, PREC>(bl + 64, hh, x);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, t, x);  // x += to_out(attn @ V)
    norm64(x, n);
    bias_init<2, true, PREC>(bl + 128, hh, t);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl + 4096, lane, n, t);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = t[m][r];
        const float z = v * 0.70710678118654752440f;
        const float ef = PREC == PREC_F32 ? erf_branchless(z) : erf_as7126(z);
        t[m][r] = 0.5f * v * (1.0f + ef);  // exact GELU (nn.GELU default)
      }
    wl = stream_step(st, wave, lane);
    bias_init<2, false, PREC>(bl + 192, hh, x);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, t, x);  // x += FF
  }
  if (xdump != nullptr) store_vec64(xdump + 3 * xstride, x);
  bias_init<1, true, PREC>(bias + 768, hh, out);
  mma_chunk<PREC, 1, 2, 0, false, 2>(st, wl + 4096, lane, x, out);
}

// ------------------------------------------------------------------------------------------
// Backward pieces of the folded transformer head on a 32-point x 64-channel tile (njf_transformer_backward)
// ------------------------------------------------------------------------------------------
// norm64 that also returns 1 / sqrt(var + eps): what the backward of the (affine-free) normalisation needs next to n itself
__device__ __forceinline__ float norm64_rstd(const f32x16 (&x)[2], f32x16 (&n)[2]) {
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[m][r];
  s += __shfl_xor(s, 32, 64);
  const float mean = s / 64.0f;
  float v = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = x[m][r] - mean;
      v = fmaf(d, d, v);
    }
  v += __shfl_xor(v, 32, 64);
  const float rstd = __builtin_amdgcn_rsqf(v / 64.0f + 1e-5f);
  const float shift = -mean * rstd;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) n[m][r] = fmaf(x[m][r], rstd, shift);
  return rstd;
}
// n = (x - mean) * rstd over 64 channels  =>  dx = rstd * (dn - mean(dn) - n * mean(dn * n));  acc += dx
__device__ __forceinline__ void norm64_backward(const f32x16 (&dn)[2], const f32x16 (&n)[2], float rstd, f32x16 (&acc)[2]) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0 += dn[m][r];
      s1 = fmaf(dn[m][r], n[m][r], s1);
    }
  s0 += __shfl_xor(s0, 32, 64);
  s1 += __shfl_xor(s1, 32, 64);
  const float m0 = s0 / 64.0f, m1 = s1 / 64.0f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = fmaf(rstd, dn[m][r] - m0 - n[m][r] * m1, acc[m][r]);
}

// ------------------------------------------------------------------------------------------
// wave-level helpers over the 32 points of a tile (both 32-lane halves hold identical data)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_min(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive prefix sum over the 32 lanes of each half
__device__ __forceinline__ float half_scan(float v, int j) {
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const float t = __shfl_up(v, o, 32);
    if (j >= o) v += t;
  }
  return v;
}

// XCD-aware block remap: consecutive work items (neighbouring rays) stay on one XCD / one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

