// Fused HIP kernels + C ABI (include/njf_hip.h) for the NJF volumetric-rendering hot path.
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: geometry must round exactly like the ATen ops of the reference path;
//  every intended fma is written as fmaf()).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "../../include/njf_hip.h"
#include "njf_device.h"

// =============================================================================================
// error codes
// =============================================================================================
enum {
  NJF_OK = 0,
  NJF_E_NULL = -1,
  NJF_E_SHAPE = -2,
  NJF_E_ACTION_DIM = -3,
  NJF_E_SAMPLES = -4,
  NJF_E_DOUT = -5,
  NJF_E_MODE = -6,
  NJF_E_GMAP = -7,
};

extern "C" int njf_abi_version(void) { return NJF_ABI_VERSION; }
extern "C" int njf_rays_per_workgroup(void) { return NJF_WAVES; }

extern "C" const char* njf_error_string(int code) {
  switch (code) {
    case NJF_OK: return "ok";
    case NJF_E_NULL: return "required pointer is NULL";
    case NJF_E_SHAPE: return "invalid shape (negative/zero extent or overflow)";
    case NJF_E_ACTION_DIM: return "action_dim must be in [1, NJF_MAX_ACTION_DIM]";
    case NJF_E_SAMPLES: return "samples per ray must be in [1, 256] for the proposal pass / >= 1 for rendering";
    case NJF_E_DOUT: return "ResnetFC d_out must be in [1, 32]";
    case NJF_E_MODE: return "unknown mode";
    case NJF_E_GMAP: return "feature map stride/offset does not cover NJF_ZDIM channels or is not 16-byte aligned";
    default: return code > 0 ? "HIP runtime error (hipError_t)" : "unknown njf error";
  }
}

static inline bool valid_base_precision(int p) {
  return p == NJF_PRECISION_F32 || p == NJF_PRECISION_F16X2 || p == NJF_PRECISION_F16F6 || p == NJF_PRECISION_F16;
}
// `precision` of the decoder entry points may name a second precision for the Jacobian head: NJF_PRECISION_MIXED(d, j)
static inline int density_precision(int p) { return p & 15; }
static inline int jacobian_precision(int p) { return (p >> 4) ? (p >> 4) - 1 : (p & 15); }
static inline bool valid_precision(int p) {
  if (p < 0 || p > 0xff || !valid_base_precision(density_precision(p)) || !valid_base_precision(jacobian_precision(p))) return false;
  const int d = density_precision(p), j = jacobian_precision(p);
  // mixed forms: the two split-precision modes in either order (the plain-fp16 mode reads an fp16 map: never mixed)
  return d == j || (d != NJF_PRECISION_F32 && j != NJF_PRECISION_F32 && d != NJF_PRECISION_F16 && j != NJF_PRECISION_F16);
}

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NJF_OK : (int)e;
}

// =============================================================================================
// weight packing
// =============================================================================================
struct PackLayer {
  const float* w;  // [d_out, d_in] row-major (torch Linear)
  const float* b;  // [d_out] or NULL
  int d_out, d_in;
  int mb, kb;  // output / input 32-blocks
  int kind;    // 0 plain, 1 lin_in (PE slots + bias column), 2 colour layer 0 (geo|1|sh slots), 3 transposed (backward chain)
  int prec;    // NJF_PRECISION_*
  float* dst;  // kb*4*mb*256 floats (same byte count in both precisions)
  float* bdst;  // 32*mb floats (logical order, zero padded) or NULL
  int bias_form;  // 0: fp32 values; 1: each entry the bit pattern of {fp16 hi, fp16 lo} of the bias (bias = hi + lo to 22 bits):
                  //    the A operand of the plain-fp16 networks' bias MFMA (njf_device.h: bias_init); 2: zeros (the bias lives elsewhere)
};

// logical (row f, input slot k) -> source value
__device__ __forceinline__ float pack_source(const PackLayer& L, int f, int k) {
  if (f >= L.d_out) return 0.f;
  if (L.kind == 0) return k < L.d_in ? L.w[f * L.d_in + k] : 0.f;
  if (L.kind == 1) {  // slots: [sin 0..29 | x | y || cos 30..59 | z | bias]
    int ch;
    if (k < 30) ch = k;
    else if (k == 30) ch = 60;
    else if (k == 31) ch = 61;
    else if (k < 62) ch = k - 2;
    else if (k == 62) ch = 62;
    else ch = -1;
    return ch >= 0 ? L.w[f * L.d_in + ch] : L.b[f];
  }
  if (L.kind == 3) return k < L.d_in ? L.w[k * L.d_out + f] : 0.f;  // transposed layer: row f of W^T, W stored [d_in, d_out]
  // slots: [geo 0..14 | bias || sh 0..15]
  if (k < 15) return L.w[f * L.d_in + k];
  if (k == 15) return L.b[f];
  return L.w[f * L.d_in + (k - 1)];
}


__device__ __forceinline__ float pack_bias_entry(const PackLayer& L, int f) {
  const float b = (f < L.d_out && L.b) ? L.b[f] : 0.f;
  if (L.bias_form == 2) return 0.f;
  if (L.bias_form == 1) {
    const _Float16 hi = (_Float16)b;
    const _Float16 lo = (_Float16)(b - (float)hi);
    return __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16));
  }
  return b;
}

// fp6 e2m3 (1 sign, 2 exponent bits with bias 1, 3 mantissa bits; max 7.5, subnormal step 1/8), round to nearest even,
// saturating: the value set of v_mfma_scale_*_f8f6f4 with cbsz/blgp = 2.
__device__ __forceinline__ unsigned encode_e2m3(float v) {
  const unsigned sign = v < 0.f ? 32u : 0u;
  const float a = fminf(fabsf(v), 7.5f);
  unsigned code;
  if (a < 1.0f) {
    code = (unsigned)rintf(a * 8.0f);  // 0..8: 8 is the encoding of 1.0 (exponent field 1, mantissa 0)
  } else {
    int e = a >= 4.0f ? 2 : (a >= 2.0f ? 1 : 0);
    float q = rintf(a * (8.0f / (float)(1 << e)));  // 8..16
    if (q >= 16.0f) {
      q = 8.0f;
      e += 1;
    }
    code = ((unsigned)(e + 1) << 3) | ((unsigned)q - 8u);
    if (e > 2) code = 31u;
  }
  return sign | code;
}

// PREC_F16F6 form of a 128-output layer (mb = 4): one 32 KiB chunk per K-range of 64 (njf_device.h: F6_* offsets).
__device__ __forceinline__ void pack_layer_f16f6_body(const PackLayer& L) {
  const int chunks = L.kb >> 1;
  _Float16* dst16 = (_Float16*)L.dst;
  // (a) hi fp16 fragments [chunk][t][m][lane][8]
  const int n = chunks * 4 * 4 * 512;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int i8 = i & 7, lane = (i >> 3) & 63;
    int rest = i >> 9;
    const int m = rest & 3;
    rest >>= 2;
    const int t = rest & 3, c = rest >> 2;
    const int ip = lane & 31, kh = lane >> 5;
    const int f = 16 * L.mb * ((ip >> 2) & 1) + 16 * m + (ip & 3) + 4 * (ip >> 3);
    const int k = 16 * L.kb * kh + 32 * c + 8 * t + i8;
    dst16[(size_t)c * (2 * NJF_CHUNK_FLOATS) + (F6_HI >> 1) + ((t * 4 + m) * 64 + lane) * 8 + i8] = (_Float16)pack_source(L, f, k);
  }
  // (b) fp6 fragments + scales: one thread per (chunk, m, w, lane)
  const int nf = chunks * 4 * 2 * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, w = (i >> 6) & 1, m = (i >> 7) & 3, c = i >> 9;
    const int ip = lane & 31, kh = lane >> 5;
    const int f = 16 * L.mb * ((ip >> 2) & 1) + 16 * m + (ip & 3) + 4 * (ip >> 3);
    float v[32], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const float x = pack_source(L, f, 16 * L.kb * kh + 32 * c + e);
      const float hi = (float)(_Float16)x;
      v[e] = w == 0 ? hi : (float)(_Float16)(x - hi);  // the residual as the f16x2 path holds it
      amax = fmaxf(amax, fabsf(v[e]));
    }
    int ex = amax > 0.f ? ilogbf(amax) - 2 : -126;   // largest element lands in [4, 8) (clamped at 7.5)
    ex = max(ex, -126);
    const float inv = ldexpf(1.0f, -ex);
    unsigned words[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const unsigned long long code = encode_e2m3(v[e] * inv);
      const int bit = 6 * e;
      words[bit >> 5] |= (unsigned)(code << (bit & 31));
      if ((bit & 31) > 26) words[(bit >> 5) + 1] |= (unsigned)(code >> (32 - (bit & 31)));
    }
    char* base = (char*)L.dst + (size_t)c * (NJF_CHUNK_FLOATS * 4);
    unsigned* p1 = (unsigned*)(base + F6_P1 + ((2 * m + w) * 64 + lane) * 16);
    unsigned* p2 = (unsigned*)(base + F6_P2 + ((2 * m + w) * 64 + lane) * 8);
    p1[0] = words[0];
    p1[1] = words[1];
    p1[2] = words[2];
    p1[3] = words[3];
    p2[0] = words[4];
    p2[1] = words[5];
    ((unsigned char*)(base + F6_SCALE + w * 256 + lane * 4))[m] = (unsigned char)(ex + 127);
  }
  // (c) the unused tail of every chunk (the weight DMA moves whole chunks)
  const int tail = (NJF_CHUNK_FLOATS * 4 - (F6_SCALE + 512)) >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < chunks * tail; i += gridDim.x * blockDim.x)
    ((unsigned*)((char*)L.dst + (size_t)(i / tail) * (NJF_CHUNK_FLOATS * 4) + F6_SCALE + 512))[i % tail] = 0u;
  if (L.bdst != nullptr && blockIdx.x == 0) {
    for (int f = threadIdx.x; f < 32 * L.mb; f += blockDim.x) L.bdst[f] = pack_bias_entry(L, f);
  }
}

__global__ void pack_layer_f16f6_kernel(PackLayer L) { pack_layer_f16f6_body(L); }

__device__ __forceinline__ void pack_layer_body(const PackLayer& L) {
  if (L.prec == NJF_PRECISION_F32) {
    const int n = L.kb * 4 * L.mb * 256;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int e = i & 3, lane = (i >> 2) & 63;
      int rest = i >> 8;
      const int m = rest % L.mb;
      rest /= L.mb;
      const int q = rest & 3, kb = rest >> 2;
      const int ip = lane & 31, kh = lane >> 5;
      const int f = 16 * L.mb * ((ip >> 2) & 1) + 16 * m + (ip & 3) + 4 * (ip >> 3);  // logical output row
      const int k = 16 * L.kb * kh + 16 * kb + 4 * q + e;                              // logical input slot
      L.dst[i] = pack_source(L, f, k);
    }
  } else if (L.prec == NJF_PRECISION_F16) {
    // [t][m][lane][8 x f16]: ONE fp16 per weight (half the bytes of the other forms; the rest of the layer's slot is unused)
    _Float16* dst = (_Float16*)L.dst;
    const int n = L.kb * 2 * L.mb * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int i8 = i & 7, lane = (i >> 3) & 63;
      const int rest = i >> 9;
      const int m = rest % L.mb, t = rest / L.mb;
      const int ip = lane & 31, kh = lane >> 5;
      const int f = 16 * L.mb * ((ip >> 2) & 1) + 16 * m + (ip & 3) + 4 * (ip >> 3);
      const int k = 16 * L.kb * kh + 8 * t + i8;
      dst[(size_t)(t * L.mb + m) * 512 + lane * 8 + i8] = (_Float16)pack_source(L, f, k);
    }
  } else {
    // [t][m][hi|lo][lane][8 x f16]; lane half kh supplies the 8 k-values 16*KB*kh + 8*t + i of K-step t
    _Float16* dst = (_Float16*)L.dst;
    const int n = L.kb * 2 * L.mb * 512;  // (t, m, lane, i8) tuples; each writes its hi and lo plane entry
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int i8 = i & 7, lane = (i >> 3) & 63;
      const int rest = i >> 9;
      const int m = rest % L.mb, t = rest / L.mb;
      const int ip = lane & 31, kh = lane >> 5;
      const int f = 16 * L.mb * ((ip >> 2) & 1) + 16 * m + (ip & 3) + 4 * (ip >> 3);
      const int k = 16 * L.kb * kh + 8 * t + i8;
      const float v = pack_source(L, f, k);
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      const size_t o = ((size_t)(t * L.mb + m) * 2) * 512 + lane * 8 + i8;
      dst[o] = hi;
      dst[o + 512] = lo;
    }
  }
  if (L.bdst != nullptr && blockIdx.x == 0) {
    for (int f = threadIdx.x; f < 32 * L.mb; f += blockDim.x) L.bdst[f] = pack_bias_entry(L, f);
  }
}

__global__ void pack_layer_kernel(PackLayer L) { pack_layer_body(L); }

// Several layers in ONE launch (blockIdx.y = layer): a network's pack is 11-22 layers of a few microseconds each, and an optimiser
// step re-packs every network it touched -- forward blobs and the backward chain's transposed blobs (an action-mode step with the
// transformer head: 36 pack launches; a perception step: ~70).  The njf_pack_* entry points queue their layers (PackScope) and
// flush once; every layer writes its own destination, so the order among them does not matter.
#define NJF_PACK_BATCH 24
struct PackBatch {
  PackLayer l[NJF_PACK_BATCH];
  unsigned char f6[NJF_PACK_BATCH];   // 1: the fp6-corrected chunk form (pack_layer_f16f6_body)
  int count;
};

__global__ void pack_batch_kernel(PackBatch B) {
  const int k = blockIdx.y;
  if (B.f6[k]) pack_layer_f16f6_body(B.l[k]);
  else pack_layer_body(B.l[k]);
}

__global__ void fill_kernel(float* p, int n, float v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

// position of logical feature f inside its 32*MB-channel block of the hoisted map.  Two layouts, one per gather form of
// add_hoisted_latent (njf_device.h); a network's layout follows its MFMA precision:
//   0 "half" (F32, F16X2): the two lanes that own a point read adjacent 16-byte pieces
//   1 "quad" (F16F6): the 16*MB floats of a lane are contiguous, accumulator register 4*e + i <-> piece i, dword e
//   2 "half, fp16 map" (F16): as 0 with 8-channel pieces
__host__ __device__ inline int njf_hoist_layout(int precision) {
  return precision == NJF_PRECISION_F16F6 ? 1 : (precision == NJF_PRECISION_F16 ? 2 : 0);
}
__host__ __device__ inline int njf_hoist_position(int f, int mb_count, int layout) {
  const int hh = f / (16 * mb_count), r = f % (16 * mb_count);
  const int m = r >> 4;
  if (layout == 0) {
    const int q = (r >> 2) & 3, e = r & 3;
    return 32 * m + 8 * q + 4 * hh + e;
  }
  if (layout == 2) {  // fp16 map: 16-byte pieces of 8 channels, the two lane halves adjacent (add_hoisted_latent_f16)
    const int q = (r >> 3) & 1, e = r & 7;
    return 32 * m + 16 * q + 8 * hh + e;
  }
  const int e = (r >> 2) & 3, i = r & 3;
  return 16 * mb_count * hh + 16 * m + 4 * i + e;
}

// the same function for callers that write hoisted channels themselves (per-image biases, folded projections)
extern "C" int njf_hoisted_channel(int feature, int block_channels, int precision) {
  if (block_channels < 32 || (block_channels & 31) || feature < 0 || feature >= block_channels) return NJF_E_SHAPE;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  return njf_hoist_position(feature, block_channels / 32, njf_hoist_layout(precision));
}

// lin_z.{0,1,2}.weight [128,512] -> wz[k * ld + 128*i + pos(f)]  (k-major: coalesced B operand of the projection)
// fold0 / fold1 (plain-fp16 networks): fc_1.bias of blocks 0 / 1, added to the map bias of blocks 1 / 2 -- the latent of block
// k + 1 is added to h right behind block k's  h += fc_1(...) + b_1, and the bilinear weights of a footprint sum to 1, so
// bilerp(G + b_1) = bilerp(G) + b_1: those two bias additions cost nothing in the kernel
__global__ void pack_linz_kernel(const float* w0, const float* w1, const float* w2, const float* b0, const float* b1,
                                 const float* b2, float* wz, int ld, float* bz, int layout, const float* fold0,
                                 const float* fold1) {
  const int n = 3 * 128 * 512;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i % 384, k = i / 384;
    const int l = c >> 7, f = c & 127;
    const float* w = l == 0 ? w0 : (l == 1 ? w1 : w2);
    wz[(size_t)k * ld + 128 * l + njf_hoist_position(f, 4, layout)] = w[f * 512 + k];
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < 384; c += blockDim.x) {
      const int l = c >> 7, f = c & 127;
      const float extra = l == 1 ? (fold0 ? fold0[f] : 0.f) : (l == 2 ? (fold1 ? fold1[f] : 0.f) : 0.f);
      bz[128 * l + njf_hoist_position(f, 4, layout)] = (l == 0 ? b0 : (l == 1 ? b1 : b2))[f] + extra;
    }
}

// The layers an entry point packs, launched together when the scope ends (also on an early error return).  One scope per entry
// point call, on the calling thread (the C ABI is re-entrant: the queue lives in the scope object, the pointer to it is thread-local).
struct PackScope;
static thread_local PackScope* g_pack_scope = nullptr;
struct PackScope {
  PackBatch b;
  hipStream_t s;
  explicit PackScope(hipStream_t stream) : s(stream) {
    b.count = 0;
    g_pack_scope = this;
  }
  void flush() {
    if (b.count > 0) pack_batch_kernel<<<dim3(64, b.count), 256, 0, s>>>(b);
    b.count = 0;
  }
  void add(const PackLayer& L, bool f6) {
    if (b.count == NJF_PACK_BATCH) flush();
    b.l[b.count] = L;
    b.f6[b.count] = f6 ? 1 : 0;
    ++b.count;
  }
  ~PackScope() {
    flush();
    g_pack_scope = nullptr;
  }
  PackScope(const PackScope&) = delete;
  PackScope& operator=(const PackScope&) = delete;
};

static void launch_pack(const float* w, const float* b, int d_out, int d_in, int mb, int kb, int kind, int prec, float* dst,
                        float* bdst, hipStream_t s, int bias_form = 0) {
  PackLayer L{w, b, d_out, d_in, mb, kb, kind, prec, dst, bdst, bias_form};
  const int n = kb * 4 * mb * 256;
  bool f6 = false;
  if (prec == NJF_PRECISION_F16F6) {
    // the 128-wide layers take the fp6-corrected chunk form; narrow layers keep the F16X2 form (njf_device.h: mma_chunk)
    if (mb == 4 && (kb & 1) == 0) f6 = true;
    else L.prec = NJF_PRECISION_F16X2;
  }
  if (g_pack_scope != nullptr && g_pack_scope->s == s) {
    g_pack_scope->add(L, f6);
    return;
  }
  if (f6) pack_layer_f16f6_kernel<<<64, 256, 0, s>>>(L);
  else pack_layer_kernel<<<(n + 255) / 256, 256, 0, s>>>(L);
}

extern "C" int njf_pack_resnetfc_ld(const NjfResnetFcWeights* src, float* w_out, float* b_out, float* wz_out, int wz_ld,
                                    float* bz_out, int precision, void* stream) {
  if (!src || !w_out || !b_out) return NJF_E_NULL;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  const int P = precision;
  if (src->d_out < 1 || src->d_out > 32) return NJF_E_DOUT;
  if (!src->lin_in_w || !src->lin_in_b || !src->lin_out_w || !src->lin_out_b) return NJF_E_NULL;
  for (int i = 0; i < 5; ++i)
    if (!src->fc0_w[i] || !src->fc0_b[i] || !src->fc1_w[i] || !src->fc1_b[i]) return NJF_E_NULL;
  if (P == NJF_PRECISION_F16 && !wz_out) return NJF_E_NULL;  // two of its accumulated biases live in the hoisted map's bias
  hipStream_t s = (hipStream_t)stream;
  PackScope scope(s);   // the network's 12 layers: one launch
  // chunk 0: lin_in 63(+bias) -> 128
  launch_pack(src->lin_in_w, src->lin_in_b, 128, NJF_PE_DIM, 4, 2, 1, P, w_out, nullptr, s);
  // a 128 x 128 layer is two chunks (K = 64 each), one in the plain-fp16 form: NJF_RESNET_CHUNKS_F16 = 12 of the blob's 22 slots
  const int per_layer = P == NJF_PRECISION_F16 ? 1 : 2;
  for (int i = 0; i < 5; ++i) {
    float* base = w_out + (size_t)(1 + 2 * per_layer * i) * NJF_CHUNK_FLOATS;
    launch_pack(src->fc0_w[i], src->fc0_b[i], 128, 128, 4, 4, 0, P, base, b_out + 256 * i, s);
    // plain-fp16 networks: fc_1.bias of blocks 0, 1 is folded into the hoisted map (pack_linz_kernel), the others are stored
    // as {hi, lo} fp16 pairs for the bias MFMA
    const int bias_form = P == NJF_PRECISION_F16 ? (i < 2 ? 2 : 1) : 0;
    launch_pack(src->fc1_w[i], src->fc1_b[i], 128, 128, 4, 4, 0, P, base + per_layer * NJF_CHUNK_FLOATS, b_out + 256 * i + 128, s,
                bias_form);
  }
  float* last = w_out + (size_t)(1 + 10 * per_layer) * NJF_CHUNK_FLOATS;
  launch_pack(src->lin_out_w, src->lin_out_b, src->d_out, 128, 1, 4, 0, P, last, b_out + 1280, s);
  fill_kernel<<<16, 256, 0, s>>>(last + 4096, 4096, 0.f);
  if (wz_out) {
    if (!bz_out || wz_ld < 384) return NJF_E_SHAPE;
    for (int i = 0; i < 3; ++i)
      if (!src->lin_z_w[i] || !src->lin_z_b[i]) return NJF_E_NULL;
    pack_linz_kernel<<<256, 256, 0, s>>>(src->lin_z_w[0], src->lin_z_w[1], src->lin_z_w[2], src->lin_z_b[0],
                                         src->lin_z_b[1], src->lin_z_b[2], wz_out, wz_ld, bz_out, njf_hoist_layout(P),
                                         P == NJF_PRECISION_F16 ? src->fc1_b[0] : nullptr,
                                         P == NJF_PRECISION_F16 ? src->fc1_b[1] : nullptr);
  }
  scope.flush();   // (before the status query: a failed batch launch must be this call's error)
  return launch_status();
}

extern "C" int njf_pack_resnetfc(const NjfResnetFcWeights* src, float* w_out, float* b_out, float* wz_out, float* bz_out,
                                 int precision, void* stream) {
  return njf_pack_resnetfc_ld(src, w_out, b_out, wz_out, 384, bz_out, precision, stream);
}

extern "C" int njf_pack_linear(const float* w, const float* b, int d_out, int d_in, int kind, float* w_out, float* b_out,
                               int precision, void* stream) {
  if (!w || !w_out) return NJF_E_NULL;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  if (d_out < 1 || d_in < 1) return NJF_E_SHAPE;
  if (kind == 1 && (d_in != NJF_PE_DIM || !b)) return NJF_E_SHAPE;
  if (kind != 0 && kind != 1) return NJF_E_MODE;
  const int mb = (d_out + 31) / 32, kb = kind == 1 ? 2 : (d_in + 31) / 32;
  launch_pack(w, b, d_out, d_in, mb, kb, kind, precision, w_out, b_out, (hipStream_t)stream);
  return launch_status();
}

extern "C" int njf_pack_color_head(const NjfColorHeadWeights* src, float* w_out, float* b_out, int precision,
                                   void* stream) {
  if (!src || !w_out || !b_out || !src->w0 || !src->b0 || !src->w1 || !src->b1 || !src->w2 || !src->b2) return NJF_E_NULL;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  hipStream_t s = (hipStream_t)stream;
  PackScope scope(s);
  launch_pack(src->w0, src->b0, 64, 31, 2, 1, 2, precision, w_out, nullptr, s);
  launch_pack(src->w1, src->b1, 64, 64, 2, 2, 0, precision, w_out + 2048, b_out, s);
  launch_pack(src->w2, src->b2, 3, 64, 1, 2, 0, precision, w_out + 6144, b_out + 64, s);
  scope.flush();   // (before the status query: a failed batch launch must be this call's error)
  return launch_status();
}

// =============================================================================================
// feature projection  G[b,p,n] = sum_k F[b,k,p] * wz[k*ld+n] + bz[n]   (exact-fp32 MFMA kernel; split variant below)
// =============================================================================================

// Split-precision variant (PREC_F16X2): one wave = 64 texels (two A tiles) x 32*NJF_PROJ_NT channels, K swept 16
// at a time with v_mfma_f32_32x32x16_f16.  Both operands are fp32 in memory and split x = hi + lo on the fly
// (hi*hi + hi*lo + lo*hi, fp32 accumulate: same accuracy class as the fp32 kernel at 3/16 of its matrix time).
// Lane (j, kh) supplies k = 16*t + 8*kh .. +7 of texel row j (A) / output channel column j (B).
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)x[i];
    hi[i] = h;
    lo[i] = (_Float16)(x[i] - (float)h);
  }
}

// Workgroup tile 128 texels x 128 channels, K swept 16 at a time.  Per step every thread fetches 8 k-values of ONE texel
// (A) and of ONE channel (B) from global memory -- both coalesced across the threads -- splits them ONCE into fp16 hi/lo and
// writes them to LDS as ready MFMA fragments ([hi|lo][32-row block][lane][8 x f16]: conflict-free 16-byte writes and
// reads); each of the four waves then computes a 64 x 64 sub-tile from 8 fragment reads and 12 MFMAs.  Double-buffered:
// the next step's global loads are in flight during the MFMAs, one barrier per step.  (The round-1 form had every wave
// fetch and split its own operands straight from global memory: 32 load instructions per 12 MFMAs, the feature plane
// re-read once per 64 output channels -- 13 % matrix-pipe efficiency; products and their order are unchanged, results are
// bit-identical.)
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
// KS = k-values per step: 16 is shipped (C2, both maps of a frame: 0.127 -> 0.102 ms against the round-1 form; a 32-wide
// step, twice the LDS and half the barriers, measured 0.117 ms).
// OUT16: the result (bias added in fp32) is rounded ONCE to fp16 and stored as a [.., n] map of halves -- the hoisted map of
// the plain-fp16 networks (NJF_PRECISION_F16); the products themselves stay error-compensated.
template <int KS, bool OUT16 = false>
__global__ void __launch_bounds__(256, 2) project_kernel_f16x2(const float* __restrict__ feats, const float* __restrict__ wz,
                                                            const float* __restrict__ bz, int hw, int n, int ld, int K,
                                                            float* __restrict__ out) {
  constexpr int NS = KS / 16;               // 16-wide sub-steps per step
  __shared__ u32x4p s_a[2][NS][2][4][64];   // [stage][sub-step][hi|lo][32-texel block][lane]
  __shared__ u32x4p s_b[2][NS][2][4][64];   // [stage][sub-step][hi|lo][32-channel block][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  // loader role: row `lr` of the tile (a texel for A, a channel for B), k-group `lk` (8 consecutive k of a sub-step's 16)
  const int lr = tid & 127, lk = tid >> 7;
  const float* fa = feats + (size_t)b * K * hw + min(p0 + lr, hw - 1);
  const float* fb = wz + min(n0 + lr, n - 1);
  const int blk = lr >> 5, slot = lk * 32 + (lr & 31);
  // compute role: wave (wm, wn) owns texel blocks 2*wm, 2*wm+1 and channel blocks 2*wn, 2*wn+1
  const int wm = wave & 1, wn = wave >> 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[a][t] = (f32x16)(0.f);
  float xa[NS][8], xb[NS][8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int kb = k0 + 16 * u + 8 * lk;
#pragma unroll
      for (int i = 0; i < 8; ++i) xa[u][i] = fa[(size_t)(kb + i) * hw];
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[u][i] = fb[(size_t)(kb + i) * ld];
    }
  };
  fetch(0);
  int st = 0;
  for (int k0 = 0; k0 < K; k0 += KS, st ^= 1) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      f16x8 h, l;
      split8(xa[u], h, l);
      s_a[st][u][0][blk][slot] = __builtin_bit_cast(u32x4p, h);
      s_a[st][u][1][blk][slot] = __builtin_bit_cast(u32x4p, l);
      split8(xb[u], h, l);
      s_b[st][u][0][blk][slot] = __builtin_bit_cast(u32x4p, h);
      s_b[st][u][1][blk][slot] = __builtin_bit_cast(u32x4p, l);
    }
    __syncthreads();  // stage `st` complete; every wave has finished reading the other stage (it did so before this barrier)
    if (k0 + KS < K) fetch(k0 + KS);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        ah[a] = __builtin_bit_cast(f16x8, s_a[st][u][0][2 * wm + a][lane]);
        al[a] = __builtin_bit_cast(f16x8, s_a[st][u][1][2 * wm + a][lane]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bh[t] = __builtin_bit_cast(f16x8, s_b[st][u][0][2 * wn + t][lane]);
        bl[t] = __builtin_bit_cast(f16x8, s_b[st][u][1][2 * wn + t][lane]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[t], acc[a][t], 0, 0, 0);
          acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[t], acc[a][t], 0, 0, 0);
          acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[t], acc[a][t], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = n0 + 64 * wn + 32 * t + j;
    if (c >= n) continue;
    const float bias = bz ? bz[c] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = p0 + 64 * wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < hw) {
          if constexpr (OUT16) ((_Float16*)out)[((size_t)b * hw + row) * n + c] = (_Float16)(acc[a][t][r] + bias);
          else out[((size_t)b * hw + row) * n + c] = acc[a][t][r] + bias;
        }
      }
  }
}

// Exact-fp32 projection through LDS (round 5).  The form above fetches every operand of every MFMA straight from global memory
// (one 4-byte load per lane and matrix instruction): 0.27 ms per C2 map = 0.45 of the fp32-MFMA peak, replicated on every rank of a
// ray-sharded frame.  Here a workgroup owns a 128-texel x 128-channel tile, K is swept 16 at a time: every thread fetches 8 k-values
// of ONE texel (A) and of ONE channel (B) -- coalesced across the threads, the next step's loads in flight during this step's MFMAs
// -- and writes them to LDS as [k][128] lines (conflict-free both ways: a wave half reads 32 consecutive floats of one line); each
// of the four waves computes a 64 x 64 sub-tile from 4 LDS reads per 4 MFMAs.  Products and their order are the direct form's (every
// accumulator sees k = 0, 1, 2, ... through the same v_mfma_f32_32x32x2_f32 sequence): bit-identical output on six shapes
// (tools/diag/diag_project_ab.py).  Measured, one box: C2 map 0.275 -> 0.231 ms (0.53 of the peak), seven training images 1.60 ->
// 1.32 ms, 256 x 256 map 0.96 -> 0.80 ms; a 64-texel tile (twice the workgroups, six resident per CU) measured the same 0.235 ms,
// so the remaining gap to the fused kernels' 0.85 is not the tail of the launch -- per 2,048 cycles of MFMAs a wave also issues 16
// global loads, 16 LDS writes and a barrier, and its LDS reads are waited for one k-pair at a time.
__global__ void __launch_bounds__(256, 2) project_kernel_f32_lds(const float* __restrict__ feats, const float* __restrict__ wz,
                                                                const float* __restrict__ bz, int hw, int n, int ld, int K,
                                                                float* __restrict__ out) {
  constexpr int KS = 16;
  __shared__ float s_a[2][KS][128];
  __shared__ float s_b[2][KS][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const int lr = tid & 127, lk = tid >> 7;   // loader role: row lr of the tile, k-group lk (8 consecutive k of a step's 16)
  const float* fa = feats + (size_t)b * K * hw + min(p0 + lr, hw - 1);
  const float* fb = wz + min(n0 + lr, n - 1);
  const int wm = wave & 1, wn = wave >> 1;   // compute role: texel blocks 2*wm, 2*wm+1 x channel blocks 2*wn, 2*wn+1
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[a][t] = (f32x16)(0.f);
  float xa[8], xb[8];
  auto fetch = [&](int k0) {
    const int kb = k0 + 8 * lk;
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[i] = fa[(size_t)(kb + i) * hw];
#pragma unroll
    for (int i = 0; i < 8; ++i) xb[i] = fb[(size_t)(kb + i) * ld];
  };
  fetch(0);
  int st = 0;
  for (int k0 = 0; k0 < K; k0 += KS, st ^= 1) {   // K is a multiple of 16 (checked by the launchers)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s_a[st][8 * lk + i][lr] = xa[i];
      s_b[st][8 * lk + i][lr] = xb[i];
    }
    __syncthreads();  // stage `st` complete; every wave finished reading the other stage before it arrived here
    if (k0 + KS < K) fetch(k0 + KS);
#pragma unroll
    for (int kk = 0; kk < KS; kk += 2) {
      float av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = s_a[st][kk + kh][64 * wm + 32 * a + j];
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = s_b[st][kk + kh][64 * wn + 32 * t + j];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[t], acc[a][t], 0, 0, 0);
    }
  }
  // D layout: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (texel)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = n0 + 64 * wn + 32 * t + j;
    if (c >= n) continue;
    const float bias = bz ? bz[c] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = p0 + 64 * wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < hw) out[((size_t)b * hw + row) * n + c] = acc[a][t][r] + bias;
      }
  }
}

static void launch_project(const float* feats, int K, const float* wz, int ld, const float* bz, int batch, int hw, int n,
                           float* out, int precision, hipStream_t s, bool out16 = false) {
  if (precision != NJF_PRECISION_F32) {  // F16X2, F16F6, F16: both operands split on the fly
    dim3 grid((hw + 127) / 128, (n + 127) / 128, batch);
    if (out16) project_kernel_f16x2<16, true><<<grid, 256, 0, s>>>(feats, wz, bz, hw, n, ld, K, out);
    else project_kernel_f16x2<16><<<grid, 256, 0, s>>>(feats, wz, bz, hw, n, ld, K, out);
  } else {
    dim3 grid((hw + 127) / 128, (n + 127) / 128, batch);
    project_kernel_f32_lds<<<grid, 256, 0, s>>>(feats, wz, bz, hw, n, ld, K, out);
  }
}

extern "C" int njf_project_features_ld(const float* feats, const float* wz, int wz_ld, const float* bz, int batch, int hw,
                                       int n, float* out, int precision, void* stream) {
  if (!feats || !wz || !bz || !out) return NJF_E_NULL;
  if (batch < 1 || hw < 1 || n < 1 || wz_ld < n) return NJF_E_SHAPE;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  // NJF_PRECISION_F16: `out` is a map of HALVES [batch, hw, n] (include/njf_hip.h)
  launch_project(feats, 512, wz, wz_ld, bz, batch, hw, n, out, precision, (hipStream_t)stream, precision == NJF_PRECISION_F16);
  return launch_status();
}

extern "C" int njf_project_features(const float* feats, const float* wz, const float* bz, int batch, int hw, int n,
                                    float* out, int precision, void* stream) {
  return njf_project_features_ld(feats, wz, n, bz, batch, hw, n, out, precision, stream);
}

// ---------------------------------------------------------------------------------------------
// Feature-pyramid producer (encoder_resnet.py:78-86 + the lin_z hoist): the encoder's output is
// cat_l(upsample_l(latent_l)); projection and bilinear up-sampling are both linear and act on different axes, so
// G = bz + sum_l upsample_l(latent_l . Wz[rows of level l]).  Each level is projected at its OWN resolution
// (5.6x fewer FLOPs than projecting the concatenated map, which is never formed), level 0 straight into `out`;
// this kernel then adds the up-sampled coarser levels in place.  One thread = 4 channels of one texel.
// ---------------------------------------------------------------------------------------------
struct UpsampleArgs {
  const float* src[3];
  int h[3], w[3];
  int levels;
  int batch, height, width, n;
  float* out;
};

__global__ void __launch_bounds__(256) upsample_add_kernel(UpsampleArgs a) {
  const int n4 = a.n >> 2;
  const long long total = (long long)a.batch * a.height * a.width * n4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % n4) * 4;
  const long long t = i / n4;
  const int x = (int)(t % a.width), y = (int)((t / a.width) % a.height), b = (int)(t / ((long long)a.width * a.height));
  f32x4 acc = *(const f32x4*)(a.out + ((size_t)t * a.n + c));
  for (int l = 0; l < a.levels; ++l) {
    // F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5) * in / out - 0.5, clamped at 0
    const int h = a.h[l], w = a.w[l];
    const float sy = fmaxf(((float)y + 0.5f) * ((float)h / (float)a.height) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)w / (float)a.width) - 0.5f, 0.f);
    const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = sy - (float)y0, wx = sx - (float)x0;
    const float* base = a.src[l] + (size_t)b * h * w * a.n + c;
    const f32x4 v00 = *(const f32x4*)(base + ((size_t)y0 * w + x0) * a.n), v01 = *(const f32x4*)(base + ((size_t)y0 * w + x1) * a.n);
    const f32x4 v10 = *(const f32x4*)(base + ((size_t)y1 * w + x0) * a.n), v11 = *(const f32x4*)(base + ((size_t)y1 * w + x1) * a.n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float top = v00[e] * (1.0f - wx) + v01[e] * wx, bot = v10[e] * (1.0f - wx) + v11[e] * wx;
      acc[e] += top * (1.0f - wy) + bot * wy;
    }
  }
  *(f32x4*)(a.out + ((size_t)t * a.n + c)) = acc;
}

// The same sum for pyramids whose coarser levels are exact 2^-s images of level 0 (ResNet latents of an image whose sides
// divide by 32: every shape the reference trains and renders): one thread = 4 channels of a 4 x 4 block of texels.  The
// bilinear taps of a block are a 4 x 4 (s = 1), 3 x 3 (s = 2) or 2 x 2 (s >= 3) patch of the coarser level, loaded once
// into registers -- 29 tap reads per 16 texels instead of 192, which is what bound the per-texel form (13 of its 14
// 16-byte accesses per texel were taps served by L2).  Weights and the order of every sum are the per-texel form's
// (src = (dst + 0.5) / 2^s - 0.5 is exact in fp32, so floor(src) is the compile-time offset table below; clamped patch
// indices reproduce the border taps, whose weight is exactly 0 where the clamp changes the tap): bit-identical output.
template <int S>  // 1, 2, or 3 = "3 or more"
__device__ __forceinline__ void upsample_block_level(const float* __restrict__ src, int h, int w, int n, int shift, int bx, int by,
                                                     int height, int width, f32x4 (&acc)[4][4]) {
  constexpr int NC = S == 1 ? 4 : S == 2 ? 3 : 2;
  constexpr int OFF[4] = {0, S == 1 ? 1 : 0, S == 3 ? 0 : 1, S == 1 ? 2 : S == 2 ? 1 : 0};
  // floor((4 k + 0.5) / 2^s - 0.5) = (8 k + 1 - 2^s) >> (s + 1), arithmetic shift
  const int base_x = (8 * bx + 1 - (1 << shift)) >> (shift + 1), base_y = (8 * by + 1 - (1 << shift)) >> (shift + 1);
  f32x4 patch[NC][NC];
#pragma unroll
  for (int r = 0; r < NC; ++r) {
    const int yy = min(max(base_y + r, 0), h - 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int xx = min(max(base_x + c, 0), w - 1);
      patch[r][c] = *(const f32x4*)(src + ((size_t)yy * w + xx) * n);
    }
  }
  float wx[4], wy[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sx = fmaxf(((float)(4 * bx + j) + 0.5f) * ((float)w / (float)width) - 0.5f, 0.f);
    const float sy = fmaxf(((float)(4 * by + j) + 0.5f) * ((float)h / (float)height) - 0.5f, 0.f);
    wx[j] = sx - (float)min((int)sx, w - 1);
    wy[j] = sy - (float)min((int)sy, h - 1);
  }
#pragma unroll
  for (int jy = 0; jy < 4; ++jy)
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      const f32x4 v00 = patch[OFF[jy]][OFF[jx]], v01 = patch[OFF[jy]][OFF[jx] + 1];
      const f32x4 v10 = patch[OFF[jy] + 1][OFF[jx]], v11 = patch[OFF[jy] + 1][OFF[jx] + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float top = v00[e] * (1.0f - wx[jx]) + v01[e] * wx[jx], bot = v10[e] * (1.0f - wx[jx]) + v11[e] * wx[jx];
        acc[jy][jx][e] += top * (1.0f - wy[jy]) + bot * wy[jy];
      }
    }
}

struct UpsampleBlockArgs {
  UpsampleArgs u;
  int shift[3];
};

__global__ void __launch_bounds__(256) upsample_add_block_kernel(UpsampleBlockArgs p) {
  const UpsampleArgs& a = p.u;
  const int n4 = a.n >> 2, bw = a.width >> 2, bh = a.height >> 2;
  const long long total = (long long)a.batch * bh * bw * n4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % n4) * 4;
  const long long t = i / n4;
  const int bx = (int)(t % bw), by = (int)((t / bw) % bh), b = (int)(t / ((long long)bw * bh));
  float* out = a.out + (((size_t)b * a.height + 4 * by) * a.width + 4 * bx) * a.n + c;
  f32x4 acc[4][4];
#pragma unroll
  for (int jy = 0; jy < 4; ++jy)
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) acc[jy][jx] = *(const f32x4*)(out + ((size_t)jy * a.width + jx) * a.n);
  for (int l = 0; l < a.levels; ++l) {
    const float* src = a.src[l] + (size_t)b * a.h[l] * a.w[l] * a.n + c;
    if (p.shift[l] == 1) upsample_block_level<1>(src, a.h[l], a.w[l], a.n, 1, bx, by, a.height, a.width, acc);
    else if (p.shift[l] == 2) upsample_block_level<2>(src, a.h[l], a.w[l], a.n, 2, bx, by, a.height, a.width, acc);
    else upsample_block_level<3>(src, a.h[l], a.w[l], a.n, p.shift[l], bx, by, a.height, a.width, acc);
  }
#pragma unroll
  for (int jy = 0; jy < 4; ++jy)
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) *(f32x4*)(out + ((size_t)jy * a.width + jx) * a.n) = acc[jy][jx];
}

// fp32 map -> fp16 map (the pyramid route of the plain-fp16 mode: the levels are summed in fp32 and rounded once)
__global__ void __launch_bounds__(256) map_to_f16_kernel(const float* __restrict__ src, long long quads, _Float16* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= quads) return;
  const f32x4 v = *(const f32x4*)(src + 4 * i);
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  *(f16x4*)(dst + 4 * i) = o;
}

// shift s with (h << s, w << s) == (height, width), or 0 if the level is not an exact 2^-s image
static int pyramid_shift(int h, int w, int height, int width) {
  for (int s = 1; s < 16; ++s)
    if ((long long)h << s == height && (long long)w << s == width) return s;
  return 0;
}

extern "C" int njf_project_pyramid(const NjfPyramidLevel* levels, int num_levels, const float* wz, int wz_ld, const float* bz,
                                   int batch, int n, float* out, float* workspace, int precision, void* stream) {
  if (!levels || !wz || !bz || !out) return NJF_E_NULL;
  if (num_levels < 1 || num_levels > 4 || batch < 1 || n < 4 || (n & 3) || wz_ld < n) return NJF_E_SHAPE;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  if (num_levels > 1 && !workspace) return NJF_E_NULL;
  int rows = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (!levels[l].feats) return NJF_E_NULL;
    if (levels[l].channels < 16 || (levels[l].channels & 15) || levels[l].height < 1 || levels[l].width < 1) return NJF_E_SHAPE;
    rows += levels[l].channels;
  }
  if (rows != 512) return NJF_E_SHAPE;  // rows of wz = channels of the concatenated encoder output
  hipStream_t s = (hipStream_t)stream;
  UpsampleArgs u;
  u.levels = num_levels - 1;
  u.batch = batch;
  u.height = levels[0].height;
  u.width = levels[0].width;
  u.n = n;
  u.out = out;
  int row0 = 0;
  float* ws = workspace;
  // plain-fp16 map: one level is projected straight into the map of halves; a pyramid is summed in fp32 in the FIRST
  // batch * H0 * W0 * n floats of the workspace (which the caller sizes accordingly, include/njf_hip.h) and rounded once
  const bool f16map = precision == NJF_PRECISION_F16;
  float* level0 = out;
  if (f16map && num_levels > 1) {
    level0 = ws;
    ws += (size_t)batch * u.height * u.width * n;
    u.out = level0;
  }
  for (int l = 0; l < num_levels; ++l) {
    const int hw = levels[l].height * levels[l].width;
    float* dst = l == 0 ? level0 : ws;
    launch_project(levels[l].feats, levels[l].channels, wz + (size_t)row0 * wz_ld, wz_ld, l == 0 ? bz : nullptr, batch, hw, n,
                   dst, precision, s, f16map && num_levels == 1);
    if (l > 0) {
      u.src[l - 1] = ws;
      u.h[l - 1] = levels[l].height;
      u.w[l - 1] = levels[l].width;
      ws += (size_t)batch * hw * n;
    }
    row0 += levels[l].channels;
  }
  if (u.levels > 0) {
    UpsampleBlockArgs blk;
    blk.u = u;
    bool blocked = (u.height & 3) == 0 && (u.width & 3) == 0;
    for (int l = 0; l < u.levels; ++l) {
      blk.shift[l] = pyramid_shift(u.h[l], u.w[l], u.height, u.width);
      blocked = blocked && blk.shift[l] > 0;
    }
    if (blocked) {
      const long long total = (long long)batch * (u.height >> 2) * (u.width >> 2) * (n >> 2);
      upsample_add_block_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(blk);
    } else {
      const long long total = (long long)batch * u.height * u.width * (n >> 2);
      upsample_add_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(u);
    }
    if (f16map) {
      const long long quads = (long long)batch * u.height * u.width * (n >> 2);
      map_to_f16_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, s>>>(level0, quads, (_Float16*)out);
    }
  }
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// The encoder's output itself, channels-last: out[t][:] = cat_l(upsample_l(latent_l))[t] (encoder_resnet.py:78-86:
// F.interpolate(bilinear, align_corners=False) to the level-0 resolution + torch.cat) as ONE pass from the NCHW latents
// into the [B*H0*W0, 512] matrix the lin_z weight-gradient GEMM contracts against (training backward; the forward pass
// never needs it -- njf_project_pyramid).  One thread = 4 channels of one texel; the latents are small and stay in L2,
// the 512-channel rows are written once, coalesced.
// ---------------------------------------------------------------------------------------------
struct ConcatArgs {
  const float* src[4];
  int c[4], h[4], w[4];
  int c0[5];  // first output channel of each level (+ total)
  int levels, batch;
  float* out;
};

// One workgroup = 64 texels of one row x 64 output channels.  Pass 1: lane = texel, wave = 16 of the channels, so the
// four bilinear taps of every NCHW plane are read along x (coalesced); the tile is transposed through LDS (row pitch 65:
// conflict-free both ways).  Pass 2: 16 consecutive threads write the 64 channels of one texel as 256 contiguous bytes.
// (The round-1 form read the planes channel-major per thread: 16 scattered 4-byte loads per thread, 0.49 ms for the
// 235 MB matrix of the training batch; this form is bound by the write.)
__global__ void __launch_bounds__(256) upsample_concat_kernel(ConcatArgs a) {
  __shared__ float tile[64 * 65];
  const int n = a.c0[a.levels];
  const int height = a.h[0], width = a.w[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = (n + 63) >> 6;
  const int xt = blockIdx.x / groups, cg = blockIdx.x - xt * groups;
  const int y = blockIdx.y, b = blockIdx.z;
  const int x = min(xt * 64 + lane, width - 1);
  for (int g = 0; g < 4; ++g) {  // the wave's 16 channels in groups of 4 (a level holds a multiple of 4 channels)
    const int c = cg * 64 + 16 * wave + 4 * g;
    if (c >= n) break;
    int l = 0;
    while (l + 1 < a.levels && c >= a.c0[l + 1]) ++l;
    const int cl = c - a.c0[l], h = a.h[l], w = a.w[l];
    // F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5) * in / out - 0.5, clamped at 0
    const float sy = fmaxf(((float)y + 0.5f) * ((float)h / (float)height) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)w / (float)width) - 0.5f, 0.f);
    const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = sy - (float)y0, wx = sx - (float)x0;
    const size_t plane = (size_t)h * w;
    const float* base = a.src[l] + ((size_t)b * a.c[l] + cl) * plane;
    const int o00 = y0 * w + x0, o01 = y0 * w + x1, o10 = y1 * w + x0, o11 = y1 * w + x1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* p = base + e * plane;
      const float top = p[o00] * (1.0f - wx) + p[o01] * wx;
      const float bot = p[o10] * (1.0f - wx) + p[o11] * wx;
      tile[lane * 65 + 16 * wave + 4 * g + e] = top * (1.0f - wy) + bot * wy;
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 15, tr = threadIdx.x >> 4;
  const int cq = cg * 64 + 4 * q;
  if (cq >= n) return;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int tx = 16 * pass + tr, gx = xt * 64 + tx;
    if (gx >= width) continue;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = tile[tx * 65 + 4 * q + e];
    *(f32x4*)(a.out + (((size_t)b * height + y) * width + gx) * n + cq) = o;
  }
}

extern "C" int njf_upsample_concat(const NjfPyramidLevel* levels, int num_levels, int batch, float* out, void* stream) {
  if (!levels || !out) return NJF_E_NULL;
  if (num_levels < 1 || num_levels > 4 || batch < 1) return NJF_E_SHAPE;
  ConcatArgs a;
  a.levels = num_levels;
  a.batch = batch;
  a.out = out;
  a.c0[0] = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (!levels[l].feats) return NJF_E_NULL;
    if (levels[l].channels < 4 || (levels[l].channels & 3) || levels[l].height < 1 || levels[l].width < 1) return NJF_E_SHAPE;
    a.src[l] = levels[l].feats;
    a.c[l] = levels[l].channels;
    a.h[l] = levels[l].height;
    a.w[l] = levels[l].width;
    a.c0[l + 1] = a.c0[l] + levels[l].channels;
  }
  const long long tiles = (long long)((a.w[0] + 63) / 64) * ((a.c0[num_levels] + 63) / 64);
  if (tiles > 0x7fffffffLL || a.h[0] > 65535 || batch > 65535) return NJF_E_SHAPE;
  upsample_concat_kernel<<<dim3((unsigned)tiles, a.h[0], batch), 256, 0, (hipStream_t)stream>>>(a);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Adjoint of njf_upsample_concat for ONE level (the encoder tail's backward pass, encoder_resnet.py:78-86 differentiated:
// what autograd runs as slice + upsample_bilinear2d_backward per latent): grad [B*H0*W0, n] channels-last -> the gradient
// of the level's NCHW latent [B, C, h, w], channels c0 .. c0+C-1 of grad.  Gather form (no atomics, fixed summation
// order: bit-reproducible): a latent texel collects w_y * w_x * grad over the fine texels whose bilinear footprint
// contains it; the weights are recomputed with the forward pass's own formulas, so the kernel is the exact transpose of
// upsample_concat_kernel for any size ratio (the window is the support widened by one texel; weights outside are zero).
// One workgroup = 64 latent texels of one row x 64 channels: 16 lanes read 256 contiguous bytes of a fine texel's row,
// the tile is transposed through LDS and written along x (NCHW).
// ---------------------------------------------------------------------------------------------
struct ConcatBwdArgs {
  const float* grad;
  int n, H, W;     // fine map: channels, height, width
  int c0, C, h, w; // the level: first channel in grad, channels, height, width
  float* dst;      // [B, C, h, w]
};

__device__ __forceinline__ float bilinear_tap_weight(int fine, int coarse, int n_coarse, float ratio) {
  // weight of latent index `coarse` in the bilinear interpolation of fine index `fine` (align_corners=False)
  const float s = fmaxf(((float)fine + 0.5f) * ratio - 0.5f, 0.f);
  const int i0 = min((int)s, n_coarse - 1), i1 = min(i0 + 1, n_coarse - 1);
  const float t = s - (float)i0;
  return (i0 == coarse ? 1.0f - t : 0.f) + (i1 == coarse ? t : 0.f);
}

__global__ void __launch_bounds__(256) upsample_concat_backward_kernel(ConcatBwdArgs a) {
  __shared__ float tile[64 * 65];
  const int groups = (a.C + 63) >> 6;
  const int xt = blockIdx.x / groups, cg = blockIdx.x - xt * groups;
  const int yc = blockIdx.y, b = blockIdx.z;
  const int q = threadIdx.x & 15, tr = threadIdx.x >> 4;
  const int c = cg * 64 + 4 * q;
  const float ry = (float)a.h / (float)a.H, rx = (float)a.w / (float)a.W;
  const int y_lo = max(0, (int)floorf(((float)yc - 0.5f) / ry - 0.5f) - 1);
  const int y_hi = min(a.H - 1, (int)ceilf(((float)yc + 1.5f) / ry - 0.5f) + 1);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int tx = 16 * pass + tr, xc = xt * 64 + tx;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (xc < a.w && c < a.C) {
      const int x_lo = max(0, (int)floorf(((float)xc - 0.5f) / rx - 0.5f) - 1);
      const int x_hi = min(a.W - 1, (int)ceilf(((float)xc + 1.5f) / rx - 0.5f) + 1);
      for (int y = y_lo; y <= y_hi; ++y) {
        const float wy = bilinear_tap_weight(y, yc, a.h, ry);
        if (wy == 0.f) continue;
        const float* row = a.grad + (((size_t)b * a.H + y) * a.W) * a.n + a.c0 + c;
        for (int x = x_lo; x <= x_hi; ++x) {
          const float wx = bilinear_tap_weight(x, xc, a.w, rx);
          if (wx == 0.f) continue;
          const f32x4 g = *(const f32x4*)(row + (size_t)x * a.n);
          const float wgt = wy * wx;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[tx * 65 + 4 * q + e] = acc[e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = xt * 64 + lane;
  if (x >= a.w) return;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int ch = cg * 64 + 16 * wave + g;
    if (ch < a.C) a.dst[(((size_t)b * a.C + ch) * a.h + yc) * a.w + x] = tile[lane * 65 + 16 * wave + g];
  }
}

extern "C" int njf_upsample_concat_backward(const float* grad, const NjfPyramidLevel* levels, int num_levels, int batch,
                                            void* stream) {
  if (!grad || !levels) return NJF_E_NULL;
  if (num_levels < 1 || num_levels > 4 || batch < 1 || batch > 65535) return NJF_E_SHAPE;
  int n = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (!levels[l].feats) return NJF_E_NULL;
    if (levels[l].channels < 4 || (levels[l].channels & 3) || levels[l].height < 1 || levels[l].width < 1 ||
        levels[l].height > 65535)
      return NJF_E_SHAPE;
    n += levels[l].channels;
  }
  int c0 = 0;
  for (int l = 0; l < num_levels; ++l) {
    ConcatBwdArgs a{grad, n, levels[0].height, levels[0].width, c0, levels[l].channels, levels[l].height, levels[l].width,
                    const_cast<float*>(levels[l].feats)};  // the record's pointer is the OUTPUT of this entry point
    const long long tiles = (long long)((a.w + 63) / 64) * ((a.C + 63) / 64);
    if (tiles > 0x7fffffffLL) return NJF_E_SHAPE;
    upsample_concat_backward_kernel<<<dim3((unsigned)tiles, a.h, batch), 256, 0, (hipStream_t)stream>>>(a);
    c0 += levels[l].channels;
  }
  return launch_status();
}

// =============================================================================================
// batched 4x4 inverse (camera matrices: torch.inverse in rendering/geometry.py:52,64)
// =============================================================================================
// One thread per matrix: Gauss-Jordan elimination with partial pivoting on [M | I] in registers (fp64).  Replaces the
// LAPACK-style getrf/getri sequence torch.linalg.inv launches per call (6 kernels + workspace fills, ~40 us of launch
// latency for a 64-byte problem; three such calls per forward pass were a quarter of a 8,192-ray step).
__global__ void invert4x4_kernel(const float* __restrict__ m, int n, float* __restrict__ out) {
  // Evaluated in float64 and rounded once: the result is the correctly rounded inverse (to ~0.5 ulp), so it differs
  // from the reference's fp32 LAPACK inverse only by LAPACK's own rounding error.  That matters: the context pose's
  // inverse feeds the positional encoding (2*pi*512 gain), where one ulp of a matrix entry is ~1e-4 of the outputs --
  // an fp32 Gauss-Jordan (2-3 ulp off) measured 4e-4 on the optical flow of a general-pose fixture, this form 7e-5.
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a[r][c] = (double)m[i * 16 + r * 4 + c];
      a[r][4 + c] = r == c ? 1.0 : 0.0;
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int p = c;
    double best = fabs(a[c][c]);
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      const double v = fabs(a[r][c]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
#pragma unroll
    for (int r = c + 1; r < 4; ++r)  // swap rows c and p (static indices: the arrays stay in registers)
      if (p == r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double t = a[c][k];
          a[c][k] = a[r][k];
          a[r][k] = t;
        }
      }
    const double inv = 1.0 / a[c][c];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[c][k] *= inv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[r][k] = fma(-f, a[c][k], a[r][k]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) out[i * 16 + r * 4 + c] = (float)a[r][4 + c];
}

extern "C" int njf_invert_4x4(const float* matrices, int count, float* out, void* stream) {
  if (!matrices || !out) return NJF_E_NULL;
  if (count < 1) return NJF_E_SHAPE;
  invert4x4_kernel<<<(count + 63) / 64, 64, 0, (hipStream_t)stream>>>(matrices, count, out);
  return launch_status();
}

// =============================================================================================
// ray generation (rendering/geometry.py:117-134, :170-203)
// =============================================================================================
__global__ void raygen_kernel(const float* __restrict__ coords, int height, int width, const float* __restrict__ k_inv,
                              const float* __restrict__ c2w, int batch, int rays, float* __restrict__ origins,
                              float* __restrict__ directions, float* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * rays) return;
  const int b = i / rays, r = i - b * rays;
  float x, y;
  if (coords) {
    x = coords[2 * (size_t)i];
    y = coords[2 * (size_t)i + 1];
  } else {  // get_pixel_coordinates: x = (col + 0.5) / W, y = (row + 0.5) / H, row-major pixels
    const int row = r / width, col = r - row * width;
    x = ((float)col + 0.5f) / (float)width;
    y = ((float)row + 0.5f) / (float)height;
  }
  const float* ki = k_inv + b * 9;
  // einsum("cij,crj->cri", K^-1, [x, y, 1]): fma chain in j order (see point_geometry)
  float cx = fmaf(ki[2], 1.0f, fmaf(ki[1], y, ki[0] * x));
  float cy = fmaf(ki[5], 1.0f, fmaf(ki[4], y, ki[3] * x));
  float cz = fmaf(ki[8], 1.0f, fmaf(ki[7], y, ki[6] * x));
  cx *= 1.0f;  // unproject multiplies by z = 1 (geometry.py:56)
  const float nrm = sqrtf(cx * cx + cy * cy + cz * cz);
  cx /= nrm;
  cy /= nrm;
  cz /= nrm;
  const float* m = c2w + b * 16;
  const float dx = fmaf(m[3], 0.0f, fmaf(m[2], cz, fmaf(m[1], cy, m[0] * cx)));
  const float dy = fmaf(m[7], 0.0f, fmaf(m[6], cz, fmaf(m[5], cy, m[4] * cx)));
  const float dz = fmaf(m[11], 0.0f, fmaf(m[10], cz, fmaf(m[9], cy, m[8] * cx)));
  origins[3 * (size_t)i + 0] = m[3];
  origins[3 * (size_t)i + 1] = m[7];
  origins[3 * (size_t)i + 2] = m[11];
  directions[3 * (size_t)i + 0] = dx;
  directions[3 * (size_t)i + 1] = dy;
  directions[3 * (size_t)i + 2] = dz;
  if (z) z[i] = cz;
}

extern "C" int njf_generate_rays(const float* coords, int height, int width, const float* k_inv, const float* c2w,
                                 int batch, int rays, float* origins, float* directions, float* z, void* stream) {
  if (!k_inv || !c2w || !origins || !directions) return NJF_E_NULL;
  if (batch < 1 || rays < 1) return NJF_E_SHAPE;
  if (!coords && (height < 1 || width < 1 || (long long)height * width != rays)) return NJF_E_SHAPE;
  const int n = batch * rays;
  raygen_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(coords, height, width, k_inv, c2w, batch, rays, origins,
                                                                 directions, z);
  return launch_status();
}

// =============================================================================================
// shared pieces of the fused ray kernels
// =============================================================================================
// chunks of a ResnetFC a kernel streams per tile: the plain-fp16 form holds a whole 128 x 128 layer per chunk
template <int PREC>
constexpr int resnet_chunks() { return PREC == PREC_F16 ? NJF_RESNET_CHUNKS_F16 : NJF_RESNET_CHUNKS; }
// ... and where the stream skips the unused slots of the FIRST network's blob (WeightStreamT<GAP_AT, GAP>; the blobs keep
// their NJF_RESNET_CHUNKS-slot extent in every precision)
template <int PREC>
constexpr int resnet_gap_at() { return PREC == PREC_F16 ? NJF_RESNET_CHUNKS_F16 : 0x7fffffff; }
template <int PREC>
constexpr int resnet_gap() { return PREC == PREC_F16 ? NJF_RESNET_CHUNKS - NJF_RESNET_CHUNKS_F16 : 0; }

struct RayCommon {
  const float* origins;
  const float* directions;
  int rays_per_batch;
  int total_rays;
  NjfCameras cams;
  NjfFeatureMap gmap;
};

__device__ __forceinline__ void load_bias_block(const float* __restrict__ src, int n, int dst_off) {
  for (int i = threadIdx.x; i < n; i += NJF_THREADS) njf_lds[LDS_BIAS + dst_off + i] = src[i];
}

// alpha = 1 - exp(-ds), ds >= 0 (ray_samplers.py:93-95), WITHOUT the cancellation of the literal form.  In fp32
// `1 - exp(-ds)` is a multiple of 2^-24 whatever its size, so a sample in nearly empty space (ds ~ 1e-6) carries its weight
// with 5-25 % relative error -- in the reference too -- and two correct exp implementations (device libm here, Sleef in
// torch's CPU path) that round exp(-ds) to different neighbours disagree by a whole quantum.  Harmless for the composited
// pixels (sums dominated by large weights), but the ds-nerf depth loss differentiates log(w + 1e-7) (utils/loss_utils.py:
// 9-35): its upstream gradient 1 / (w + 1e-7) is LARGEST exactly on those weights, and the one-quantum disagreement was the
// 2.3-2.9e-3 deviation of every proposal-net gradient from the oracle (tools/diag/diag_perception.py, round 4; same in exact
// fp32 and split precision).  Evaluated accurately -- Taylor to x^5 below 2^-4 (truncation 1.3e-9 relative), the literal form
// above it (relative error <= 1e-6) -- the weights follow the float64 result instead of a rounding accident, and the
// gradients fall inside twice the oracle's own fp32-vs-float64 floors.
__device__ __forceinline__ float alpha_of(float ds) {
  const float series = ds * fmaf(-0.5f * ds, fmaf(-(1.0f / 3.0f) * ds, fmaf(-0.25f * ds, fmaf(-0.2f, ds, 1.0f), 1.0f), 1.0f), 1.0f);
  return ds < 0.0625f ? series : 1.0f - expf(-ds);
}

// alpha compositing weights of one 32-sample tile (ray_samplers.py:77-101), carrying the running
// optical depth across tiles.
__device__ __forceinline__ float tile_weights(float delta, float sigma, bool valid, int j, float& carry) {
  const float ds = (valid && delta > 0.f) ? delta * sigma : 0.f;
  const float incl = half_scan(ds, j);
  float excl = __shfl_up(incl, 1, 32);
  if (j == 0) excl = 0.f;
  excl += carry;
  carry += __shfl(incl, 31, 32);
  return alpha_of(ds) * expf(-excl);
}

// inverse-CDF resampling of one ray (ray_samplers.py:351-451).  w' (already annealed, +padding not
// yet applied) lives in sc[0..s_in); cdf is built in sc[256..256+s_in].  All 64 lanes cooperate.
__device__ __forceinline__ void pdf_resample_ray(float* __restrict__ sc, const float* __restrict__ bins_in, int s_in,
                                                 const float* __restrict__ u, int s_out, float* __restrict__ bins_out,
                                                 int lane, bool store) {
  float part = 0.f;
  for (int i = lane; i < s_in; i += 64) part += sc[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
  float wsum = part;
  const float pad = fmaxf(1e-5f - wsum, 0.f);
  const float padper = pad / (float)s_in;
  wsum += pad;
  // inclusive cumsum of pdf: contiguous run per lane + wave scan of run totals
  const int per = (s_in + 63) >> 6;
  const int i0 = lane * per;
  float run = 0.f;
  for (int i = 0; i < per; ++i) {
    const int s = i0 + i;
    if (s < s_in) run += (sc[s] + padper) / wsum;
  }
  float incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  float acc = incl - run;
  float* cdf = sc + 256;
  __syncthreads();  // all lanes have read sc[] for wsum before anyone overwrites (uniform control flow)
  for (int i = 0; i < per; ++i) {
    const int s = i0 + i;
    if (s < s_in) {
      acc += (sc[s] + padper) / wsum;
      cdf[s + 1] = fminf(1.0f, acc);
    }
  }
  if (lane == 0) cdf[0] = 0.f;
  __syncthreads();
  for (int k = lane; k <= s_out; k += 64) {
    const float uk = u[k];
    int lo = 0, hi = s_in + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uk) lo = mid + 1;
      else hi = mid;
    }
    const int below = min(max(lo - 1, 0), s_in), above = min(max(lo, 0), s_in);
    const float c0 = cdf[below], c1 = cdf[above];
    const float g0 = bins_in[below], g1 = bins_in[above];
    float t = (uk - c0) / (c1 - c0);
    if (t != t) t = 0.f;  // nan_to_num(nan=0); +-inf are absorbed by the clip
    t = fminf(fmaxf(t, 0.f), 1.f);
    if (store) bins_out[k] = g0 + t * (g1 - g0);
  }
}

// =============================================================================================
// proposal pass
// =============================================================================================
struct ProposalArgs {
  RayCommon rc;
  int gmap_offset;
  const float* w_pack;
  const float* b_pack;
  const float* bins_in;
  int bins_per_ray;
  int s_in;
  const float* u;
  int u_per_ray;
  int s_out;
  float anneal;
  float* bins_out;
  float* weights_out;
  float* density_out;
  NjfActivationDump dump;  // DUMP instantiation only (training forward)
};

// this lane's share of a point's backward-pass inputs: activations / encoding (both halves), footprint (half 0)
__device__ __forceinline__ ActDump point_dump(const NjfActivationDump& d, size_t pidx, size_t points, int hh,
                                              const PointGeom& g, int tex0, int texel_stride) {
  float* act = nullptr;   // this lane's 64 values of layer 0 (floats, or halves under the 16-bit training storage)
  if (d.act) act = d.act_f16 ? (float*)((_Float16*)d.act + pidx * 128 + 64 * hh) : d.act + pidx * 128 + 64 * hh;
  ActDump dump{act, d.pe + pidx * 64 + 32 * hh, points * 128, d.mask ? d.mask + pidx * 4 + 2 * hh : nullptr, d.act_f16 != 0};
  if (hh == 0 && d.foot_idx != nullptr) {
    Footprint f;
    point_footprint(g, f);
    int* fi = d.foot_idx + pidx * 4;
    fi[0] = tex0 + f.t00 / texel_stride;
    fi[1] = tex0 + f.t01 / texel_stride;
    fi[2] = tex0 + f.t10 / texel_stride;
    fi[3] = tex0 + f.t11 / texel_stride;
    float* fw = d.foot_w + pidx * 4;
    fw[0] = f.w00;
    fw[1] = f.w01;
    fw[2] = f.w10;
    fw[3] = f.w11;
  }
  return dump;
}

template <int PREC, bool DUMP = false>
__global__ void __launch_bounds__(NJF_THREADS, 2) proposal_kernel(ProposalArgs a) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ray = wg * NJF_WAVES + wave;
  const bool ray_ok = ray < a.rc.total_rays;
  const int rayc = min(ray, a.rc.total_rays - 1);
  const int b = rayc / a.rc.rays_per_batch;

  load_bias_block(a.b_pack, NJF_RESNET_B_FLOATS, 0);
  const int tiles = (a.s_in + 31) >> 5;
  WeightStream st;
  stream_begin(st, a.w_pack, resnet_chunks<PREC>(), tiles, wave, lane);
  NJF_STAMP_ARM(st, true, wave);

  CamCtx cam;
  load_ctx(a.rc.cams.ctxt_w2c, a.rc.cams.ctxt_k, b, cam);
  const float near = a.rc.cams.z_near[b], far = a.rc.cams.z_far[b];
  const float ox = a.rc.origins[3 * (size_t)rayc], oy = a.rc.origins[3 * (size_t)rayc + 1],
              oz = a.rc.origins[3 * (size_t)rayc + 2];
  const float dx = a.rc.directions[3 * (size_t)rayc], dy = a.rc.directions[3 * (size_t)rayc + 1],
              dz = a.rc.directions[3 * (size_t)rayc + 2];
  const float* gz = map_at<PREC>(a.rc.gmap.data, (size_t)b * a.rc.gmap.height * a.rc.gmap.width * a.rc.gmap.stride + a.gmap_offset);
  const float* bins = a.bins_in + (a.bins_per_ray ? (size_t)rayc * (a.s_in + 1) : 0);
  float* sc = njf_lds + LDS_SCRATCH_PROPOSAL + wave * LDS_SCRATCH_PER_WAVE;
  const float* bias = njf_lds + LDS_BIAS;

  float carry = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const int s = t * 32 + j;
    const bool valid = s < a.s_in;
    const int sc_i = min(s, a.s_in - 1);
    const float b0 = bins[sc_i], b1 = bins[sc_i + 1];
    const float start = b0 * far + (1.0f - b0) * near;
    const float end = b1 * far + (1.0f - b1) * near;
    const float se = start + end;
    const float px = ox + (dx * se) / 2.0f, py = oy + (dy * se) / 2.0f, pz = oz + (dz * se) / 2.0f;
    PointGeom g;
    point_geometry(cam, px, py, pz, a.rc.gmap.height, a.rc.gmap.width, a.rc.gmap.stride, 0u, g);
    NJF_STAMP_P(st, 10);
    f32x16 pe[2];
    positional_encoding(g.xc, g.yc, g.zc, hh, pe);
    f32x16 out[1];
    ActDump dump{nullptr, nullptr, 0};
    if (DUMP && valid && ray_ok)
      dump = point_dump(a.dump, (size_t)ray * a.s_in + s, (size_t)a.rc.total_rays * a.s_in, hh, g,
                        b * a.rc.gmap.height * a.rc.gmap.width, a.rc.gmap.stride);
    // (plain fp16: one footprint for the three gathers + lin_in from a packed encoding, as in the render kernel, measured 0.7 %
    //  SLOWER here -- profiles/r05_ablate_f16.txt block k -- and is not used)
    resnet_tile<PREC, DUMP>(st, bias, gz, g, pe, wave, lane, out, dump);
    NJF_STAMP_P(st, 14);
    const float pre = __shfl(out[0][0], j, 64);
    const float sigma = expf(pre - 1.0f);
    const float w = tile_weights(end - start, sigma, valid, j, carry);
    if (valid && hh == 0) {
      float wa = w;
      if (a.anneal != 1.0f) wa = powf(w, a.anneal);
      sc[s] = wa + 0.01f;  // histogram_padding (ray_samplers.py:375)
      if (ray_ok) {
        if (a.weights_out) a.weights_out[(size_t)ray * a.s_in + s] = w;
        if (a.density_out) a.density_out[(size_t)ray * a.s_in + s] = sigma;
      }
    }
  }
  NJF_STAMP_P(st, 13);
  __syncthreads();
  const float* u = a.u + (a.u_per_ray ? (size_t)rayc * (a.s_out + 1) : 0);
  pdf_resample_ray(sc, bins, a.s_in, u, a.s_out, a.bins_out + (size_t)rayc * (a.s_out + 1), lane, ray_ok);
  NJF_STAMP_FLUSH(st, 15, lane);
}

// =============================================================================================
// decoder evaluation of one tile, stage by stage: density net -> colour head -> Jacobian head / flow.  The callers
// composite (or store) each stage's result before the next stage starts, so that only the sample weight, the
// camera-space position and the ray constants stay live across the 22 weight chunks of the Jacobian head.
// =============================================================================================
// bias layout (LDS_BIAS): [density 1312 | colour 96 | jacobian head (MLP 1312 / transformer 800)]
// JKIND: 0 = no Jacobian head, 1 = ResnetFC head (jacobian_mlp), 2 = folded transformer head (jacobian_transformer)
// DUMP: 0 = inference, 1 = dump the Jacobian ResnetFC (action-mode training), 2 = dump the density ResnetFC and the
// colour head (perception-mode training); `dump` addresses the dumped net, `cdump` the colour head.
// SHARE (plain-fp16 inference): the stage fills `share` (packed encoding + footprint of this tile's points, TileShareF16) for
// its own network and for a ResnetFC Jacobian head that follows.
template <int PREC, int DUMP, int SHARE = 0, class ST>
__device__ __forceinline__ float density_stage(ST& st, const float* __restrict__ gz_d, const PointGeom& g, int wave,
                                               int lane, f32x16 (&geo)[1], ActDump dump, TileShareF16* share = nullptr) {
  const int j = lane & 31, hh = lane >> 5;
  f32x16 pe[2];
  positional_encoding(g.xc, g.yc, g.zc, hh, pe);
  if constexpr (SHARE != 0) {
    share_tile(pe, *share);
    resnet_tile<PREC, false, SHARE>(st, njf_lds + LDS_BIAS, gz_d, g, pe, wave, lane, geo, dump, share);
  } else {
    resnet_tile<PREC, DUMP == 2>(st, njf_lds + LDS_BIAS, gz_d, g, pe, wave, lane, geo, dump);
  }
  return expf(__shfl(geo[0][15], j, 64) - 1.0f);
}

template <int PREC, int DUMP, class ST>
__device__ __forceinline__ void color_stage(ST& st, const f32x16 (&geo)[1], float dirx, float diry, float dirz,
                                            int wave, int lane, float (&rgb)[3], ColorDump cdump) {
  const int j = lane & 31, hh = lane >> 5;
  // The harmonics depend on the ray only, so the compiler computes them once in front of the tile loop -- and, with every
  // register taken by the networks, spills all 16 and reloads them here one by one behind s_waitcnt vmcnt(0) (which also
  // waits for the weight DMA in flight): 16 serialised memory round trips per tile for ~30 VALU instructions of work.
  // Opaque copies of the direction make it a per-tile recomputation.
  asm volatile("" : "+v"(dirx), "+v"(diry), "+v"(dirz));
  float sh[16];
  sh4(dirx, diry, dirz, sh);
  f32x16 cin[1], crgb[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) cin[0][r] = hh ? sh[r] : (r < 15 ? geo[0][r] : 1.0f);
  color_tile<PREC, DUMP == 2>(st, njf_lds + LDS_BIAS + NJF_RESNET_B_FLOATS, cin, wave, lane, crgb, cdump);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = __shfl(crgb[0][c], j, 64);
    rgb[c] = 1.0f / (1.0f + expf(-x));
  }
}

template <int JKIND, int PREC, int DUMP, int SHARE = 0, class ST>
__device__ __forceinline__ void jacobian_stage(ST& st, const float* __restrict__ gz_j, const PointGeom& g,
                                               const float* __restrict__ action, int action_dim, int wave, int lane,
                                               f32x16 (&jac)[1], float (&flow)[3], ActDump dump, const TileShareF16* share = nullptr) {
  const int hh = lane >> 5;
  const float* bias = njf_lds + LDS_BIAS + NJF_RESNET_B_FLOATS + NJF_COLOR_B_FLOATS;
  if constexpr (SHARE != 0) {   // ResnetFC head in plain fp16: the density stage's packed encoding (TileShareF16)
    static_assert(JKIND == 1 && DUMP == 0, "shared tile state: ResnetFC Jacobian head, inference");
    NJF_STAMP(st, 6);
    NJF_STAMP(st, 7);
    f32x16 unused[2];
    resnet_tile<PREC, false, SHARE>(st, bias, gz_j, g, unused, wave, lane, jac, dump, share);
  } else {
  // the encoding is recomputed (~1 % of the head's time) rather than held in 32 VGPRs across density + colour
  // ... and REALLY recomputed: without the opaque copies the compiler keeps the density stage's 25 encoding values alive
  // through scratch and reloads them inside the lin_in chunk, each reload behind its own s_waitcnt vmcnt(0)
  float xc = g.xc, yc = g.yc, zc = g.zc;
  asm volatile("" : "+v"(xc), "+v"(yc), "+v"(zc) : : "memory");
  f32x16 pe[2];
  NJF_STAMP(st, 6);  // Jacobian stage begins
  positional_encoding(xc, yc, zc, hh, pe);
  NJF_STAMP_PIN("+v"(pe[0]), "+v"(pe[1]));
  NJF_STAMP(st, 7);  // encoding done
  if (JKIND == 1)
    resnet_tile<PREC, DUMP == 1>(st, bias, gz_j, g, pe, wave, lane, jac, dump);
  else {
    if (DUMP == 1 && dump.pe != nullptr) {  // the transformer head's backward pass recomputes it from pe + footprint
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
    // (training forward: dump.act addresses the head's residual-stream dump [4][P][64], render_kernel)
    transformer_tile<PREC>(st, bias, gz_j, g, pe, action_dim, wave, lane, jac, DUMP == 1 ? dump.act : nullptr, dump.stride);
  }
  }
  // flow_s = sum_a J[3a+s] * action[a]  (action_decoder_jacobian.py:128-145); this lane holds
  // logical outputs 16*hh + r.  Partial sums by phase r%3, then the two halves are combined.
  float ph[3] = {0.f, 0.f, 0.f};
  if (action) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d0 = r, d1 = 16 + r;  // logical output index for hh = 0 / 1
      const float a0 = (d0 / 3) < action_dim ? action[d0 / 3] : 0.f;
      const float a1 = (d1 / 3) < action_dim ? action[d1 / 3] : 0.f;
      ph[r % 3] = fmaf(jac[0][r], hh ? a1 : a0, ph[r % 3]);
    }
  }
  // hh=0: spatial index s = r%3 ; hh=1: s = (16+r)%3 = (r+1)%3  ->  phase (s+2)%3
  float mine[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) mine[s] = hh ? ph[(s + 2) % 3] : ph[s];
#pragma unroll
  for (int s = 0; s < 3; ++s) flow[s] = mine[s] + __shfl_xor(mine[s], 32, 64);
}

// =============================================================================================
// final pass: decoder + compositing
// =============================================================================================
struct RenderArgs {
  RayCommon rc;
  int goff_d, goff_j;
  const float* w_all;  // [density 22 chunks | colour 1 | jacobian 22] contiguous
  const float* b_d;
  const float* b_c;
  const float* b_j;
  const float* bins;
  int samples;
  NjfRenderOutputs out;
};

// sample placement of lane j of tile t: interval [start, end], its mid-point and the world-space position there
// (ray_samplers.py:104-147).  Recomputed from the bins after each network instead of being kept in registers.
struct SamplePlace {
  float delta, tm, px, py, pz;
};
// `b0`, `b1`: the sample's two bin edges, loaded once per tile and kept in two registers; the five derived values are
// recomputed where they are needed (opaque copies: otherwise the compiler keeps all five alive across the networks)
__device__ __forceinline__ void place_sample(float b0, float b1, float near, float far, float ox,
                                             float oy, float oz, float dx, float dy, float dz, SamplePlace& sp) {
  asm volatile("" : "+v"(b0), "+v"(b1));
  const float start = b0 * far + (1.0f - b0) * near;
  const float end = b1 * far + (1.0f - b1) * near;
  const float se = start + end;
  sp.delta = end - start;
  sp.tm = se / 2.0f;
  sp.px = ox + (dx * se) / 2.0f;
  sp.py = oy + (dy * se) / 2.0f;
  sp.pz = oz + (dz * se) / 2.0f;
}

// AF: composite the per-sample action features (sum_s w J, 16 more accumulators per lane) -- a compile-time switch
// because the extra live registers cost ~80 spilled VGPRs in the frames that do not ask for them
// PRECJ: MFMA precision of the Jacobian head (default: that of the density / colour networks)
template <int JKIND, int PREC, int DUMP = 0, bool AF = true, int PRECJ = PREC>
__global__ void __launch_bounds__(NJF_THREADS, 2) render_kernel(RenderArgs a) {
  constexpr bool WITH_J = JKIND != 0;
  constexpr int J_CHUNKS = JKIND == 1 ? resnet_chunks<PRECJ>() : (JKIND == 2 ? NJF_TRANSFORMER_CHUNKS : 0);
  constexpr int J_BIAS = JKIND == 1 ? NJF_RESNET_B_FLOATS : NJF_TRANSFORMER_B_FLOATS;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ray = wg * NJF_WAVES + wave;
  const bool ray_ok = ray < a.rc.total_rays;
  const int rayc = min(ray, a.rc.total_rays - 1);
  const int b = rayc / a.rc.rays_per_batch;
  const int S = a.samples;

  load_bias_block(a.b_d, NJF_RESNET_B_FLOATS, 0);
  load_bias_block(a.b_c, NJF_COLOR_B_FLOATS, NJF_RESNET_B_FLOATS);
  if (WITH_J) load_bias_block(a.b_j, J_BIAS, NJF_RESNET_B_FLOATS + NJF_COLOR_B_FLOATS);
  const int tiles = (S + 31) >> 5;
  WeightStreamT<resnet_gap_at<PREC>(), resnet_gap<PREC>()> st;
  stream_begin(st, a.w_all, resnet_chunks<PREC>() + 1 + J_CHUNKS, tiles, wave, lane);
  NJF_STAMP_ARM(st, false, wave);

  CamCtx cam;
  load_ctx(a.rc.cams.ctxt_w2c, a.rc.cams.ctxt_k, b, cam);
  const float near = a.rc.cams.z_near[b], far = a.rc.cams.z_far[b];
  const float ox = a.rc.origins[3 * (size_t)rayc], oy = a.rc.origins[3 * (size_t)rayc + 1],
              oz = a.rc.origins[3 * (size_t)rayc + 2];
  const float dx = a.rc.directions[3 * (size_t)rayc], dy = a.rc.directions[3 * (size_t)rayc + 1],
              dz = a.rc.directions[3 * (size_t)rayc + 2];
  const size_t gbase = (size_t)b * a.rc.gmap.height * a.rc.gmap.width * a.rc.gmap.stride;
  const float* gz_d = map_at<PREC>(a.rc.gmap.data, gbase + a.goff_d);
  const float* gz_j = map_at<PRECJ>(a.rc.gmap.data, gbase + a.goff_j);
  const float* bins = a.bins + (size_t)rayc * (S + 1);
  const int A = a.rc.cams.action_dim;
  const float* action = a.rc.cams.action ? a.rc.cams.action + (size_t)b * A : nullptr;

  float carry = 0.f;
  float acc_rgb[3] = {0.f, 0.f, 0.f}, acc_w = 0.f, acc_wt = 0.f;
  float acc_p[3] = {0.f, 0.f, 0.f}, acc_pw[3] = {0.f, 0.f, 0.f};
  float tmin = 3.0e38f, tmax = -3.0e38f;
  f32x16 acc_j = (f32x16)(0.f);
  const bool want_af = AF && WITH_J && a.out.action_features != nullptr;

  for (int t = 0; t < tiles; ++t) {
    const int s = t * 32 + j;
    const bool valid = s < S;
    const int sc_i = min(s, S - 1);
    const size_t si = (size_t)ray * S + s;
    const bool store = valid && ray_ok;
    const float bin0 = bins[sc_i], bin1 = bins[sc_i + 1];
    PointGeom g;
    {
      SamplePlace sp;
      place_sample(bin0, bin1, near, far, ox, oy, oz, dx, dy, dz, sp);
      point_geometry(cam, sp.px, sp.py, sp.pz, a.rc.gmap.height, a.rc.gmap.width, a.rc.gmap.stride, 0u, g);
    }
    NJF_STAMP(st, 10);  // tile begins
    ActDump dump{nullptr, nullptr, 0};
    ColorDump cdump{nullptr, nullptr, 0};
    if (DUMP != 0 && store) {
      const size_t points = (size_t)a.rc.total_rays * S;
      const NjfActivationDump d{DUMP == 1 ? a.out.jac_act : a.out.den_act, a.out.jac_pe, a.out.foot_idx, a.out.foot_w,
                                DUMP == 1 ? a.out.jac_mask : a.out.den_mask, a.out.dump_f16};
      dump = point_dump(d, si, points, hh, g, b * a.rc.gmap.height * a.rc.gmap.width, a.rc.gmap.stride);
      if constexpr (JKIND == 2 && DUMP == 1) {
        // the transformer head's backward pass (njf_transformer_backward) recomputes each layer from the residual stream in front
        // of it: jac_act is [4][P][64] here -- x before layers 0, 1, 2 and behind layer 2 -- written by transformer_tile
        dump.act = a.out.jac_act ? a.out.jac_act + si * 64 + 32 * hh : nullptr;
        dump.stride = points * 64;
        dump.mask = nullptr;
        dump.half = false;
      }
      if (DUMP == 2) cdump = ColorDump{a.out.col_in + si * 32 + 16 * hh, a.out.col_act + si * 64 + 32 * hh, points * 64};
    }
    // ---- density net -> sample weight; everything that only needs the weight is composited right away
    f32x16 geo[1];
    // plain-fp16 inference: the tile's packed encoding serves both networks, one footprint the gathers of a network (TileShareF16)
#define NJF_F16_SHARE_D 1
#define NJF_F16_SHARE_J 3
    // (the instantiations that also composite the action features, AF, have no registers to spare: 6-35 spilled VGPRs with any of it)
    constexpr int SHARE_D = (PREC == PREC_F16 && DUMP == 0 && !AF) ? NJF_F16_SHARE_D : 0;
    constexpr int SHARE_J = (SHARE_D != 0 && JKIND == 1 && PRECJ == PREC_F16) ? NJF_F16_SHARE_J : 0;
    TileShareF16 share;
    const float sigma = density_stage<PREC, DUMP, SHARE_D>(st, gz_d, g, wave, lane, geo, DUMP == 2 ? dump : ActDump{nullptr, nullptr, 0}, &share);
    float w;
    {
      SamplePlace sp;
      place_sample(bin0, bin1, near, far, ox, oy, oz, dx, dy, dz, sp);
      NJF_STAMP(st, 14);  // density net returned
      w = tile_weights(sp.delta, sigma, valid, j, carry);
      NJF_STAMP_PIN("+v"(w));
      NJF_STAMP(st, 15);  // weights done
      if (valid) {
        acc_w += w;
        acc_wt = fmaf(w, sp.tm, acc_wt);
        tmin = fminf(tmin, sp.tm);
        tmax = fmaxf(tmax, sp.tm);
        acc_p[0] = fmaf(w, sp.px, acc_p[0]);
        acc_p[1] = fmaf(w, sp.py, acc_p[1]);
        acc_p[2] = fmaf(w, sp.pz, acc_p[2]);
      }
    }
    if (store && hh == 0) {
      if (a.out.weights) a.out.weights[si] = w;
      if (a.out.density) a.out.density[si] = sigma;
    }
    // ---- colour head
    NJF_STAMP(st, 11);  // density net + weights done
    {
      float rgb[3];
      color_stage<PREC, DUMP>(st, geo, dx, dy, dz, wave, lane, rgb, cdump);
      if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc_rgb[c] = fmaf(w, rgb[c], acc_rgb[c]);
      }
      if (store && hh == 0 && a.out.color) {
        a.out.color[3 * si] = rgb[0];
        a.out.color[3 * si + 1] = rgb[1];
        a.out.color[3 * si + 2] = rgb[2];
      }
    }
    // ---- Jacobian head -> scene flow
    NJF_STAMP(st, 12);  // colour head done
    float flow[3] = {0.f, 0.f, 0.f};
    if (WITH_J) {
      f32x16 jac[1];
      jacobian_stage<JKIND, PRECJ, DUMP, SHARE_J>(st, gz_j, g, action, A, wave, lane, jac, flow, DUMP == 1 ? dump : ActDump{nullptr, nullptr, 0}, &share);
      if (valid && want_af) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_j[r] = fmaf(w, jac[0][r], acc_j[r]);
      }
      if (store) {
        if (hh == 0 && a.out.sample_flow) {
          a.out.sample_flow[3 * si] = flow[0];
          a.out.sample_flow[3 * si + 1] = flow[1];
          a.out.sample_flow[3 * si + 2] = flow[2];
        }
        if (a.out.jacobian) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = 16 * hh + r;
            if (d < 3 * A) a.out.jacobian[si * (3 * A) + d] = jac[0][r];
          }
        }
      }
    }
    if (valid) {
      SamplePlace sp;
      place_sample(bin0, bin1, near, far, ox, oy, oz, dx, dy, dz, sp);
      acc_pw[0] = fmaf(w, sp.px + flow[0], acc_pw[0]);
      acc_pw[1] = fmaf(w, sp.py + flow[1], acc_pw[1]);
      acc_pw[2] = fmaf(w, sp.pz + flow[2], acc_pw[2]);
    }
  }

  NJF_STAMP_FLUSH(st, 13, lane);  // tiles done
  // reduce over the ray's samples (32 lanes of a half; both halves hold identical per-sample data)
  acc_w = half_sum(acc_w);
  acc_wt = half_sum(acc_wt);
  tmin = half_min(tmin);
  tmax = half_max(tmax);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    acc_rgb[c] = half_sum(acc_rgb[c]);
    acc_p[c] = half_sum(acc_p[c]);
    acc_pw[c] = half_sum(acc_pw[c]);
  }
  if (want_af) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_j[r] = half_sum(acc_j[r]);
    if (ray_ok && j == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 16 * hh + r;
        if (d < 3 * A) a.out.action_features[(size_t)ray * (3 * A) + d] = acc_j[r];
      }
    }
  }
  float flow2d[2] = {0.f, 0.f};
  if (ray_ok && lane == 0) {
    if (a.out.rgb) {
      a.out.rgb[3 * (size_t)ray] = acc_rgb[0];
      a.out.rgb[3 * (size_t)ray + 1] = acc_rgb[1];
      a.out.rgb[3 * (size_t)ray + 2] = acc_rgb[2];
    }
    if (a.out.depth) a.out.depth[ray] = acc_wt / (acc_w + 1e-10f);
    if (a.out.step_minmax) {
      a.out.step_minmax[2 * (size_t)ray] = tmin;
      a.out.step_minmax[2 * (size_t)ray + 1] = tmax;
    }
    if (a.out.pos) {
      a.out.pos[3 * (size_t)ray] = acc_p[0];
      a.out.pos[3 * (size_t)ray + 1] = acc_p[1];
      a.out.pos[3 * (size_t)ray + 2] = acc_p[2];
    }
    if (a.out.pos_warped) {
      a.out.pos_warped[3 * (size_t)ray] = acc_pw[0];
      a.out.pos_warped[3 * (size_t)ray + 1] = acc_pw[1];
      a.out.pos_warped[3 * (size_t)ray + 2] = acc_pw[2];
    }
    if (a.out.flow && a.rc.cams.trgt_w2c && a.rc.cams.trgt_k) {
      // project_world_coords_to_camera (rendering/geometry.py:206-215) of both means
      const float* m = a.rc.cams.trgt_w2c + b * 16;
      const float* k = a.rc.cams.trgt_k + b * 9;
      float uv[2][2];
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const float* p = v ? acc_pw : acc_p;
        const float xc = dot4_h(m + 0, p[0], p[1], p[2]);
        const float yc = dot4_h(m + 4, p[0], p[1], p[2]);
        const float zc = dot4_h(m + 8, p[0], p[1], p[2]);
        const float u0 = dot3(k + 0, xc, yc, zc), u1 = dot3(k + 3, xc, yc, zc), u2 = dot3(k + 6, xc, yc, zc);
        uv[v][0] = u0 / (u2 + 1e-9f);
        uv[v][1] = u1 / (u2 + 1e-9f);
      }
      flow2d[0] = uv[1][0] - uv[0][0];
      flow2d[1] = uv[1][1] - uv[0][1];
      a.out.flow[2 * (size_t)ray] = flow2d[0];
      a.out.flow[2 * (size_t)ray + 1] = flow2d[1];
    }
  }
  if (a.out.frame_partials) {
    // frame-level reductions of this workgroup's four rays (include/njf_hip.h: NjfRenderOutputs.frame_partials), in a fixed
    // order
    float se_rgb = 0.f, se_flow = 0.f;
    if (ray_ok && lane == 0) {
      if (a.out.trgt_rgb && a.out.rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = acc_rgb[c] - a.out.trgt_rgb[3 * (size_t)ray + c];
          se_rgb = fmaf(d, d, se_rgb);
        }
      }
      if (a.out.trgt_flow && a.out.flow && a.rc.cams.trgt_w2c && a.rc.cams.trgt_k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float d = flow2d[c] - a.out.trgt_flow[2 * (size_t)ray + c];
          se_flow = fmaf(d, d, se_flow);
        }
      }
    }
    __syncthreads();  // every wave has consumed its last weight chunk: the weight buffers are free
    if (lane == 0) {
      njf_lds[4 * wave + 0] = ray_ok ? tmin : 3.0e38f;
      njf_lds[4 * wave + 1] = ray_ok ? tmax : -3.0e38f;
      njf_lds[4 * wave + 2] = se_rgb;
      njf_lds[4 * wave + 3] = se_flow;
    }
    __syncthreads();
    if (tid == 0) {
      float mn = njf_lds[0], mx = njf_lds[1], s0 = njf_lds[2], s1 = njf_lds[3];
#pragma unroll
      for (int w = 1; w < NJF_WAVES; ++w) {
        mn = fminf(mn, njf_lds[4 * w]);
        mx = fmaxf(mx, njf_lds[4 * w + 1]);
        s0 += njf_lds[4 * w + 2];
        s1 += njf_lds[4 * w + 3];
      }
      float* dst = a.out.frame_partials + 4 * (size_t)wg;
      dst[0] = mn;
      dst[1] = mx;
      dst[2] = s0;
      dst[3] = s1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// frame-level reductions of a (ray-sharded) render: fold the per-workgroup partials of the render kernel's epilogue, and
// assemble the frame from the all-gathered per-rank packets (include/njf_hip.h).  Fixed orders: bit-reproducible.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_frame_partials_kernel(const float* __restrict__ partials, int groups,
                                                                    float* __restrict__ out4) {
  __shared__ float red[4][256];
  float mn = 3.0e38f, mx = -3.0e38f, s0 = 0.f, s1 = 0.f;
  for (int g = threadIdx.x; g < groups; g += 256) {
    const f32x4 v = *(const f32x4*)(partials + 4 * (size_t)g);
    mn = fminf(mn, v[0]);
    mx = fmaxf(mx, v[1]);
    s0 += v[2];
    s1 += v[3];
  }
  red[0][threadIdx.x] = mn;
  red[1][threadIdx.x] = mx;
  red[2][threadIdx.x] = s0;
  red[3][threadIdx.x] = s1;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] = fminf(red[0][threadIdx.x], red[0][threadIdx.x + o]);
      red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + o]);
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
      red[3][threadIdx.x] += red[3][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) out4[threadIdx.x] = red[threadIdx.x][0];
}

extern "C" int njf_reduce_frame_partials(const float* partials, int groups, float* out4, void* stream) {
  if (!partials || !out4) return NJF_E_NULL;
  if (groups < 1) return NJF_E_SHAPE;
  reduce_frame_partials_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partials, groups, out4);
  return launch_status();
}

__global__ void __launch_bounds__(256) assemble_frame_kernel(const float* __restrict__ packets, int world, int packet_floats,
                                                             int batch, int rays, int cap, float rgb_scale, float flow_scale,
                                                             float* __restrict__ frame, float* __restrict__ scalars6) {
  // global bounds / sums from the trailing record of every packet, ranks in order (every workgroup recomputes them:
  // world <= a few dozen, and it saves a launch)
  float mn = 3.0e38f, mx = -3.0e38f, s0 = 0.f, s1 = 0.f;
  for (int k = 0; k < world; ++k) {
    const float* rec = packets + (size_t)k * packet_floats + (packet_floats - 4);
    mn = fminf(mn, rec[0]);
    mx = fmaxf(mx, rec[1]);
    s0 += rec[2];
    s1 += rec[3];
  }
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    scalars6[0] = mn;
    scalars6[1] = mx;
    scalars6[2] = s0;
    scalars6[3] = s1;
    scalars6[4] = s0 * rgb_scale;    // e.g. 1 / (B*R*3): the photometric mse of the whole frame (model_wrapper.py:119-121)
    scalars6[5] = s1 * flow_scale;   // e.g. 0.01 / (B*R*2): flow_loss (model_wrapper.py:148-160)
  }
  if (i >= (long long)batch * rays) return;
  const int b = (int)(i / rays), r = (int)(i % rays);
  // parallel.shard_bounds: the first `rem` ranks own q + 1 rays, the others q
  const int q = rays / world, rem = rays % world;
  int k, local;
  if (r < rem * (q + 1)) {
    k = r / (q + 1);
    local = r - k * (q + 1);
  } else {
    k = rem + (r - rem * (q + 1)) / max(q, 1);
    local = r - rem * (q + 1) - (k - rem) * q;
  }
  const int n_k = q + (k < rem ? 1 : 0);
  const float* pk = packets + (size_t)k * packet_floats;
  const float* rgb = pk + ((size_t)b * n_k + local) * 3;
  const float* dep = pk + (size_t)3 * batch * cap + (size_t)b * n_k + local;
  const float* flw = pk + (size_t)4 * batch * cap + ((size_t)b * n_k + local) * 2;
  float* dst = frame + (size_t)i * 6;
  dst[0] = rgb[0];
  dst[1] = rgb[1];
  dst[2] = rgb[2];
  dst[3] = fminf(fmaxf(dep[0], mn), mx);   // torch.clip(depth, steps.min(), steps.max()) with the GLOBAL bounds
  dst[4] = flw[0];
  dst[5] = flw[1];
}

extern "C" int njf_assemble_frame(const float* packets, int world, int packet_floats, int batch, int rays_per_batch,
                                  float rgb_scale, float flow_scale, float* frame, float* scalars6, void* stream) {
  if (!packets || !frame || !scalars6) return NJF_E_NULL;
  if (world < 1 || batch < 1 || rays_per_batch < 1) return NJF_E_SHAPE;
  const int cap = (rays_per_batch + world - 1) / world;
  if (packet_floats < 6 * batch * cap + 4) return NJF_E_SHAPE;
  const long long total = (long long)batch * rays_per_batch;
  assemble_frame_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(packets, world, packet_floats, batch,
                                                                                          rays_per_batch, cap, rgb_scale, flow_scale,
                                                                                          frame, scalars6);
  return launch_status();
}

// =============================================================================================
// point-list evaluation
// =============================================================================================
struct PointsArgs {
  const float* xyz;
  const float* dirs;
  int points_per_batch;
  int total_points;
  NjfCameras cams;
  NjfFeatureMap gmap;
  int goff_d, goff_j;
  const float* w_all;
  const float* b_d;
  const float* b_c;
  const float* b_j;
  float* density;
  float* color;
  float* flow;
  float* jacobian;
  float* geo;
  float* features;  // [5, P, 128] or null (FEAT instantiation only)
};

// MODE 0: proposal net (density only); 1: decoder without Jacobian head; 2: decoder + ResnetFC Jacobian head;
// 3: decoder + transformer Jacobian head.  FEAT (MODE 2): the head also stores its residual stream after every block
// (ResnetFC.forward(compute_features=True), resnet_fc.py:141-151) through the training instantiation of resnet_tile
template <int MODE, int PREC, int PRECJ = PREC, bool FEAT = false>
__global__ void __launch_bounds__(NJF_THREADS, 2) points_kernel(PointsArgs a) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = wg * NJF_WAVES + wave;
  const int p = tile * 32 + j;
  const bool ok = p < a.total_points;
  const int pc = min(p, a.total_points - 1);
  const int b = pc / a.points_per_batch;  // per lane: a tile may straddle batch elements

  load_bias_block(a.b_d, NJF_RESNET_B_FLOATS, 0);
  if (MODE >= 1) load_bias_block(a.b_c, NJF_COLOR_B_FLOATS, NJF_RESNET_B_FLOATS);
  if (MODE == 2) load_bias_block(a.b_j, NJF_RESNET_B_FLOATS, NJF_RESNET_B_FLOATS + NJF_COLOR_B_FLOATS);
  if (MODE == 3) load_bias_block(a.b_j, NJF_TRANSFORMER_B_FLOATS, NJF_RESNET_B_FLOATS + NJF_COLOR_B_FLOATS);
  WeightStreamT<resnet_gap_at<PREC>(), resnet_gap<PREC>()> st;
  stream_begin(st, a.w_all,
               MODE == 0 ? resnet_chunks<PREC>()
                         : resnet_chunks<PREC>() + 1 + (MODE == 2 ? resnet_chunks<PRECJ>() : (MODE == 3 ? NJF_TRANSFORMER_CHUNKS : 0)),
               1, wave, lane);
  CamCtx cam;
  load_ctx(a.cams.ctxt_w2c, a.cams.ctxt_k, b, cam);
  const float px = a.xyz[3 * (size_t)pc], py = a.xyz[3 * (size_t)pc + 1], pz = a.xyz[3 * (size_t)pc + 2];
  PointGeom g;
  // the batch element's offset travels with the point (PointGeom::gofs): the gathers take the map itself as base
  const unsigned gbase = (unsigned)b * (unsigned)(a.gmap.height * a.gmap.width) * (unsigned)a.gmap.stride;
  point_geometry(cam, px, py, pz, a.gmap.height, a.gmap.width, a.gmap.stride, gbase, g);
  const float* bias = njf_lds + LDS_BIAS;
  if (MODE == 0) {
    f32x16 pe[2], out[1];
    positional_encoding(g.xc, g.yc, g.zc, hh, pe);
    resnet_tile<PREC>(st, bias, map_at<PREC>(a.gmap.data, a.goff_d), g, pe, wave, lane, out);
    if (ok && hh == 0 && a.density) a.density[p] = expf(out[0][0] - 1.0f);
  } else {
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (a.dirs) {
      dx = a.dirs[3 * (size_t)pc];
      dy = a.dirs[3 * (size_t)pc + 1];
      dz = a.dirs[3 * (size_t)pc + 2];
    }
    const int A = a.cams.action_dim;
    const float* action = a.cams.action ? a.cams.action + (size_t)b * A : nullptr;
    constexpr int JK = MODE >= 2 ? MODE - 1 : 0;
    const ActDump nodump{nullptr, nullptr, 0};
    // stage by stage, each result stored before the next network starts (nothing but the point itself stays live)
    f32x16 geo[1];
    const float sigma = density_stage<PREC, 0>(st, map_at<PREC>(a.gmap.data, a.goff_d), g, wave, lane, geo, nodump);
    if (ok && hh == 0) {
      if (a.density) a.density[p] = sigma;
      if (a.geo) {
#pragma unroll
        for (int r = 0; r < 15; ++r) a.geo[15 * (size_t)p + r] = geo[0][r];
      }
    }
    {
      float rgb[3];
      color_stage<PREC, 0>(st, geo, dx, dy, dz, wave, lane, rgb, ColorDump{nullptr, nullptr, 0});
      if (ok && hh == 0 && a.color) {
        a.color[3 * (size_t)p] = rgb[0];
        a.color[3 * (size_t)p + 1] = rgb[1];
        a.color[3 * (size_t)p + 2] = rgb[2];
      }
    }
    if (JK != 0) {
      f32x16 jac[1];
      float flow[3];
      // NOTE: `action` is per lane here (tiles may straddle batch elements)
      if constexpr (FEAT) {
        static_assert(MODE == 2, "feature dump: ResnetFC head");
        ActDump fdump{nullptr, nullptr, (size_t)a.total_points * 128};
        if (ok) fdump.feat = a.features + (size_t)p * 128 + 64 * hh;
        jacobian_stage<JK, PRECJ, 1>(st, map_at<PRECJ>(a.gmap.data, a.goff_j), g, action, A, wave, lane, jac, flow, fdump);
      } else
      jacobian_stage<JK, PRECJ, 0>(st, map_at<PRECJ>(a.gmap.data, a.goff_j), g, action, A, wave, lane, jac, flow, nodump);
      if (ok) {
        if (hh == 0 && a.flow) {
          a.flow[3 * (size_t)p] = flow[0];
          a.flow[3 * (size_t)p + 1] = flow[1];
          a.flow[3 * (size_t)p + 2] = flow[2];
        }
        if (a.jacobian) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = 16 * hh + r;
            if (d < 3 * A) a.jacobian[(size_t)p * (3 * A) + d] = jac[0][r];
          }
        }
      }
    }
  }
}

// =============================================================================================
// inverse dynamics: Levenberg-Marquardt on the linearised flow, one workgroup per batch element
// =============================================================================================
// optical_flow(a) = proj(x + M a) - proj(x) with x = sum_s w x_s and M = sum_s w J_s (the composited outputs of the
// final pass; render_optical_flow, model.py:288-314).  Minimises sum_r mask_r |flow_r(a) - target_r|^2 over the command
// a -- the control loop of notebooks/real_world/2_inverse_dynamics.ipynb, which runs 100 Adam steps through
// Model.infer_optical_flow instead.  All iterations run inside ONE launch: per iteration every thread linearises its
// rays (2 x A Jacobian rows into LDS), A*A threads contract them into the normal matrix in a fixed order (no
// atomics: deterministic), thread 0 solves the damped A x A system, and the step is kept if it lowers the cost.
#define NJF_SOLVE_MAX_A 16
struct SolveArgs {
  const float* pos;     // [B,R,3]
  const float* jac;     // [B,R,3,A]
  const float* proj;    // [B,3,4]  K . inv(E)[:3]
  const float* target;  // [B,R,2]
  const float* mask;    // [B,R] or null
  const float* init;    // [B,A] or null
  int R, A, iters;
  float damping;
  float* action;        // [B,A]
};

__global__ void __launch_bounds__(256) solve_action_kernel(SolveArgs a) {
  __shared__ float rows[256][2 * NJF_SOLVE_MAX_A + 2];  // per ray: 2 Jacobian rows (A each) + 2 residuals
  __shared__ float hmat[NJF_SOLVE_MAX_A][NJF_SOLVE_MAX_A + 1];
  __shared__ float act[NJF_SOLVE_MAX_A], cand[NJF_SOLVE_MAX_A], red[256];
  __shared__ float lam, cost, cost_c;
  const int b = blockIdx.x, tid = threadIdx.x, A = a.A, R = a.R;
  const float* P = a.proj + b * 12;
  if (tid < A) act[tid] = a.init ? a.init[b * A + tid] : 0.f;
  if (tid == 0) lam = a.damping;
  __syncthreads();

  // residual (and optionally Jacobian rows) of ray r at command `cmd`; returns the squared residual
  auto eval_ray = [&](int r, const float* cmd, bool want_rows, int slot) -> float {
    const size_t ri = (size_t)b * R + r;
    const float w = a.mask ? a.mask[ri] : 1.f;
    float x[3], x0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x0[c] = a.pos[ri * 3 + c];
      float v = x0[c];
      for (int k = 0; k < A; ++k) v = fmaf(a.jac[(ri * 3 + c) * A + k], cmd[k], v);
      x[c] = v;
    }
    auto project = [&](const float* p, float& u, float& v, float& d) {
      const float hx = fmaf(P[2], p[2], fmaf(P[1], p[1], P[0] * p[0])) + P[3];
      const float hy = fmaf(P[6], p[2], fmaf(P[5], p[1], P[4] * p[0])) + P[7];
      d = fmaf(P[10], p[2], fmaf(P[9], p[1], P[8] * p[0])) + P[11] + 1e-9f;
      u = hx / d;
      v = hy / d;
    };
    float u0, v0, d0, u, v, d;
    project(x0, u0, v0, d0);
    project(x, u, v, d);
    const float r0 = ((u - u0) - a.target[ri * 2]) * w, r1 = ((v - v0) - a.target[ri * 2 + 1]) * w;
    if (want_rows) {
      // d uv / d x = (P[:2,:3] - uv (x) P[2,:3]) / depth, times M
      float gu[3], gv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gu[c] = (P[c] - u * P[8 + c]) / d;
        gv[c] = (P[4 + c] - v * P[8 + c]) / d;
      }
      for (int k = 0; k < A; ++k) {
        float ju = 0.f, jv = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float m = a.jac[(ri * 3 + c) * A + k];
          ju = fmaf(gu[c], m, ju);
          jv = fmaf(gv[c], m, jv);
        }
        rows[slot][k] = ju * w;
        rows[slot][A + k] = jv * w;
      }
      rows[slot][2 * A] = r0;
      rows[slot][2 * A + 1] = r1;
    }
    return r0 * r0 + r1 * r1;
  };
  auto block_sum = [&](float v) -> float {
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const float out = red[0];
    __syncthreads();
    return out;
  };

  for (int it = 0; it < a.iters; ++it) {
    // normal equations H = J^T J, g = J^T res, accumulated over chunks of 256 rays in a fixed order
    // entry e = hi * (A + 1) + hj of [H | g] (hj == A: the right-hand side); A * (A + 1) <= 272 entries, so a thread
    // owns entry tid and, for A = 16, entry tid + 256 as well
    const int entries = A * (A + 1);
    float hacc[2] = {0.f, 0.f}, csum = 0.f;
    for (int r0 = 0; r0 < R; r0 += 256) {
      const int r = r0 + tid;
      if (r < R) csum += eval_ray(r, act, true, tid);
      __syncthreads();
      const int n = min(256, R - r0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 256 * k;
        if (e < entries) {
          const int hi = e / (A + 1), hj = e % (A + 1);
          float acc = hacc[k];
          for (int q = 0; q < n; ++q) {
            const float bu = hj < A ? rows[q][hj] : rows[q][2 * A], bv = hj < A ? rows[q][A + hj] : rows[q][2 * A + 1];
            acc = fmaf(rows[q][hi], bu, acc);
            acc = fmaf(rows[q][A + hi], bv, acc);
          }
          hacc[k] = acc;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      if (e < entries) hmat[e / (A + 1)][e % (A + 1)] = hacc[k];
    }
    const float c_now = block_sum(csum);
    if (tid == 0) {
      cost = c_now;
      // (H + lam diag(H)) step = g, un-pivoted Gauss-Jordan (symmetric positive definite after damping)
      for (int i = 0; i < A; ++i) hmat[i][i] += lam * fmaxf(hmat[i][i], 1e-12f);
      for (int k = 0; k < A; ++k) {
        const float inv = 1.0f / hmat[k][k];
        for (int j = 0; j <= A; ++j) hmat[k][j] *= inv;
        for (int i = 0; i < A; ++i) {
          if (i == k) continue;
          const float f = hmat[i][k];
          for (int j = 0; j <= A; ++j) hmat[i][j] = fmaf(-f, hmat[k][j], hmat[i][j]);
        }
      }
      for (int i = 0; i < A; ++i) cand[i] = act[i] - hmat[i][A];
    }
    __syncthreads();
    float cs = 0.f;
    for (int r = tid; r < R; r += 256) cs += eval_ray(r, cand, false, 0);
    const float c_new = block_sum(cs);
    if (tid == 0) {
      const bool better = c_new < cost;  // NaN (a point behind the camera) compares false: step rejected
      if (better)
        for (int i = 0; i < A; ++i) act[i] = cand[i];
      lam = fminf(fmaxf(better ? lam / 3.0f : lam * 4.0f, 1e-9f), 1e9f);
    }
    __syncthreads();
  }
  if (tid < A) a.action[b * A + tid] = act[tid];
}

extern "C" int njf_solve_action(const float* mean_position, const float* jacobian, const float* projection,
                                const float* target_flow, const float* visible_mask, const float* init_action, int batch,
                                int rays, int action_dim, int iterations, float damping, float* action, void* stream) {
  if (!mean_position || !jacobian || !projection || !target_flow || !action) return NJF_E_NULL;
  if (batch < 1 || rays < 1 || iterations < 0) return NJF_E_SHAPE;
  if (action_dim < 1 || action_dim > NJF_SOLVE_MAX_A) return NJF_E_ACTION_DIM;
  SolveArgs a{mean_position, jacobian, projection, target_flow, visible_mask, init_action, rays, action_dim, iterations, damping, action};
  solve_action_kernel<<<batch, 256, 0, (hipStream_t)stream>>>(a);
  return launch_status();
}

// =============================================================================================
// backward of the pixel-aligned bilinear sampling (training)
// =============================================================================================
// out[foot_idx[p][c]] += foot_w[p][c] * grad[p]  for the four texels of every point's footprint: the input gradient of
// F.grid_sample(bilinear, border, align_corners=True) (model_components/pixel_aligned_features.py:29-33) in hoisted
// order, i.e. on the [P, channels] latent gradient BEFORE lin_z's transpose is applied per texel.  One thread per
// (run of `run` consecutive points, channel), lane = channel: every atomic instruction of a wave covers 64 consecutive
// floats of one texel row (two full 128-byte lines; global_atomic_add_f32, no CAS loop).  Points are ordered ray-major,
// so consecutive points are neighbouring samples of one ray, which mostly fall into the same texels: each footprint
// corner keeps a running sum in a register and only issues an atomic when its texel changes (2-3x fewer atomics).
__global__ void __launch_bounds__(256) scatter_footprint_kernel(const float* __restrict__ grad, long long slice_stride,
                                                                const int* __restrict__ foot_idx,
                                                                const float* __restrict__ foot_w, int points, int run,
                                                                int channels, int row, long long total,
                                                                float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long r = i / row;
  const int col = (int)(i - r * row);          // column of the [texels, slices * channels] output
  const int slice = col / channels, ch = col - slice * channels;
  const float* g = grad + (size_t)slice * slice_stride + ch;
  const long long p0 = r * run;
  const int n = (int)min((long long)run, (long long)points - p0);
  int cur[4] = {-1, -1, -1, -1};
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < n; ++s) {
    const size_t p = (size_t)(p0 + s);
    const float v = g[p * channels];
    const int4 idx = *(const int4*)(foot_idx + p * 4);
    const f32x4 w = *(const f32x4*)(foot_w + p * 4);
    const int t[4] = {idx.x, idx.y, idx.z, idx.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (t[c] != cur[c]) {  // wave-uniform: every lane of a wave works on the same points
        if (cur[c] >= 0) unsafeAtomicAdd(out + (size_t)cur[c] * row + col, acc[c]);
        cur[c] = t[c];
        acc[c] = 0.f;
      }
      acc[c] = fmaf(w[c], v, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (cur[c] >= 0) unsafeAtomicAdd(out + (size_t)cur[c] * row + col, acc[c]);
}

extern "C" int njf_scatter_footprint(const float* grad, int slices, long long slice_stride, const int* foot_idx,
                                     const float* foot_w, int points, int channels, int texels, int run_length, float* out,
                                     void* stream) {
  if (!grad || !foot_idx || !foot_w || !out) return NJF_E_NULL;
  if (points < 1 || texels < 1 || channels < 1 || run_length < 1 || slices < 1 || slices > 64) return NJF_E_SHAPE;
  if ((channels & 63) && slices > 1) return NJF_E_SHAPE;  // a wave must stay inside one slice (wave-uniform footprints)
  const long long runs = ((long long)points + run_length - 1) / run_length;
  const int row = slices * channels;
  const long long total = runs * row;
  if ((total + 255) / 256 > 0x7fffffffLL) return NJF_E_SHAPE;
  scatter_footprint_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      grad, slice_stride, foot_idx, foot_w, points, run_length, channels, row, total, out);
  return launch_status();
}

// out = residual + upstream * [act > 0]  and per-block column sums of out: one layer step of the ResnetFC backward chain
// (model_components/resnet_fc.py:69-79,130-154 differentiated: the ReLU mask, the residual add and the bias gradient
// that autograd would run as compare + multiply + add + sum kernels).  A workgroup owns `rows_per_block` consecutive
// rows; a thread owns 4 consecutive channels and walks the rows in steps of 256 / (channels / 4) (8 rows per iteration
// for 128 channels): every access is a coalesced 16-byte piece of a row.
// Column sums are reduced per workgroup in a fixed order (registers -> LDS -> one row of `partial`): deterministic; the
// caller adds the few partial rows.
__global__ void __launch_bounds__(256) relu_backward_kernel(const float* __restrict__ upstream,
                                                            const float* __restrict__ act,
                                                            const float* __restrict__ residual, int points, int quads,
                                                            int rows_per_block, float* __restrict__ out,
                                                            float* __restrict__ partial) {
  __shared__ float red[256 * 4];
  const int lanes_per_row = quads;                 // threads covering one row (channels / 4)
  const int rows_per_iter = 256 / lanes_per_row;   // rows a workgroup handles per iteration
  const int tq = threadIdx.x % lanes_per_row, tr = threadIdx.x / lanes_per_row;
  const long long row0 = (long long)blockIdx.x * rows_per_block;
  const int nrows = (int)min((long long)rows_per_block, (long long)points - row0);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  if (tr < rows_per_iter) {
    for (int r = tr; r < nrows; r += rows_per_iter) {
      const size_t o = ((size_t)(row0 + r) * quads + tq) * 4;
      const f32x4 g = *(const f32x4*)(upstream + o);
      const f32x4 a = *(const f32x4*)(act + o);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? g[e] : 0.f;
      if (residual != nullptr) {
        const f32x4 d = *(const f32x4*)(residual + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += d[e];
      }
      *(f32x4*)(out + o) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
  }
  if (partial == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = sum[e];
  __syncthreads();
  if (threadIdx.x < lanes_per_row) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rows_per_iter; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += red[(r * lanes_per_row + threadIdx.x) * 4 + e];
    *(f32x4*)(partial + ((size_t)blockIdx.x * quads + threadIdx.x) * 4) = t;
  }
}

extern "C" int njf_relu_backward(const float* upstream, const float* act, const float* residual, int points, int channels,
                                 int rows_per_block, float* out, float* partial_colsum, void* stream) {
  if (!upstream || !act || !out) return NJF_E_NULL;
  if (points < 1 || rows_per_block < 1) return NJF_E_SHAPE;
  if (channels < 4 || (channels & 3) || channels > 1024 || (256 % (channels / 4)) != 0) return NJF_E_SHAPE;
  const long long blocks = ((long long)points + rows_per_block - 1) / rows_per_block;
  if (blocks > 0x7fffffffLL) return NJF_E_SHAPE;
  relu_backward_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(upstream, act, residual, points, channels / 4,
                                                                           rows_per_block, out, partial_colsum);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Epilogue of a convolution of the FROZEN encoder trunk (encoder_resnet.py:24-89: torchvision BasicBlocks in eval mode):
// out = [relu]( batch_norm_eval(x) [+ skip] ) on NCHW fp32 tensors, in ONE pass -- what the library runs as a batch-norm
// inference kernel, an add and a ReLU (2-3 launches of ~4 us each: a single-image trunk is ~115 launches whose device time is
// the launch floor, not the bytes).  batch_norm_eval(x)[c] = (x - mean[c]) / sqrt(var[c] + eps) * gamma[c] + beta[c], evaluated
// in that order (the library's).  One thread = 4 consecutive floats of one (image, channel) plane (hw % 4 == 0), or one float.
// ---------------------------------------------------------------------------------------------
struct BnActArgs {
  const float* x;
  const float *gamma, *beta, *mean, *var;
  float eps;
  const float* skip;
  int relu, channels, hw;
  long long total;   // elements
  float* out;
};

template <int V>
__global__ void __launch_bounds__(256) bn_act_kernel(BnActArgs a) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (i >= a.total) return;
  const int c = (int)((i / a.hw) % a.channels);
  const float mean = a.mean[c], invstd = 1.0f / sqrtf(a.var[c] + a.eps), g = a.gamma[c], bta = a.beta[c];
  float v[V], sk[V];
  if constexpr (V == 4) {
    const f32x4 x4 = *(const f32x4*)(a.x + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x4[e];
    if (a.skip) {
      const f32x4 s4 = *(const f32x4*)(a.skip + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) sk[e] = s4[e];
    }
  } else {
    v[0] = a.x[i];
    if (a.skip) sk[0] = a.skip[i];
  }
#pragma unroll
  for (int e = 0; e < V; ++e) {
    float y = (v[e] - mean) * invstd * g + bta;
    if (a.skip) y += sk[e];
    v[e] = a.relu ? fmaxf(y, 0.f) : y;
  }
  if constexpr (V == 4) {
    const f32x4 o = {v[0], v[1], v[2], v[3]};
    *(f32x4*)(a.out + i) = o;
  } else {
    a.out[i] = v[0];
  }
}

extern "C" int njf_bn_act(const float* x, const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, const float* skip, int relu, int batch, int channels, int hw,
                          float* out, void* stream) {
  if (!x || !gamma || !beta || !running_mean || !running_var || !out) return NJF_E_NULL;
  if (batch < 1 || channels < 1 || hw < 1 || !(eps >= 0.f)) return NJF_E_SHAPE;
  BnActArgs a{x, gamma, beta, running_mean, running_var, eps, skip, relu != 0, channels, hw, (long long)batch * channels * hw, out};
  hipStream_t s = (hipStream_t)stream;
  if ((hw & 3) == 0) {
    const long long threads = a.total >> 2;
    if ((threads + 255) / 256 > 0x7fffffffLL) return NJF_E_SHAPE;
    bn_act_kernel<4><<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(a);
  } else {
    if ((a.total + 255) / 256 > 0x7fffffffLL) return NJF_E_SHAPE;
    bn_act_kernel<1><<<(unsigned)((a.total + 255) / 256), 256, 0, s>>>(a);
  }
  return launch_status();
}

// =============================================================================================
// backward data-gradient chain of one ResnetFC (training)
// =============================================================================================
// The reverse of resnet_tile: delta^T = W^T * upstream^T layer by layer, with the weights (transposed, packed once per
// weight update by njf_pack_resnetfc_backward) as the MFMA A operand streamed through LDS exactly like the forward pass
// and the gradient kept in the accumulator registers across all 11 layers of the net (model_components/resnet_fc.py:
// 69-79, 130-154 differentiated).  Per layer the wave reads the ReLU'd layer input the forward pass dumped (the ReLU
// mask) and writes the gradient the weight-gradient GEMM of that layer contracts with:
//   deltas[10]      = [act[10] > 0] * (W_out^T d_out)                       gradient w.r.t. h after block 4
//   deltas[2b + 1]  = [act[2b+1] > 0] * (W_fc1,b^T deltas[2b + 2])          gradient w.r.t. fc_0's output of block b
//   deltas[2b]      = deltas[2b + 2] + [act[2b] > 0] * (W_fc0,b^T deltas[2b + 1])   gradient w.r.t. h before block b
// (for b = 4 read deltas[10] where the formulas say deltas[2b + 2]).  Hence dW of the layer whose input is act[l] is
// deltas[l + 1]^T act[l] for l = 0..9, its bias gradient the column sum of deltas[l + 1], and deltas[2b] (b < 3) / deltas[0]
// are the gradients w.r.t. the hoisted latents / the lin_in output.  Exact-fp32 MFMA: gradients span many orders of
// magnitude (the split-precision forms are only exact inside fp16's range), and the work is small (0.33 MFLOP/point).
struct BackwardArgs {
  const float* d_out;   // [P, d_out_dim]
  int d_out_dim;
  const float* act;     // [11, P, 128]
  const float* w_pack;  // 21 chunks: lin_out^T | (fc_1^T, fc_0^T) of blocks 4..0
  int points;
  float* deltas;        // [11, P, 128]
  float* colsum;        // [tiles, 11, 128] per-tile column sums of deltas, or nullptr
  const unsigned* masks;  // [11, P, 4] ReLU masks the training forward dumped (then `act` is not read), or nullptr
  const float* absmax;    // device scalar max|d_out|: PREC_F16X2 (the chain runs on gradients scaled by a power of two) and deltas16
  _Float16* deltas16;     // 16-bit training storage: [11, P, 128] halves = deltas x 2^k (k from absmax as in PREC_F16X2); `deltas`
                          // is then [3, P, 128] and receives slices 0, 2, 4 only (the latent gradients the footprint scatter reads)
};

// Sum over the 32 points of a tile (lanes of one wave half) of every accumulator register, in DPP: rotate-and-add inside
// the rows of 16 lanes (afterwards every lane of a row holds the row's sum), then row_bcast:15 adds row 0 into row 1 and
// row 2 into row 3 -- lanes 16..31 / 48..63 hold the sums of the half's 64 features.  Lane 16 + 4m + q (48 + ...) then
// stores its quad: 16 predicated 16-byte stores per layer.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ void tile_colsum(const f32x16 (&acc)[4], float* __restrict__ dst, int lane, float unscale = 1.0f) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[m][4 * q + e];
        v = dpp_add<0x128, 0xf>(v);  // row_ror:8
        v = dpp_add<0x124, 0xf>(v);  // row_ror:4
        v = dpp_add<0x122, 0xf>(v);  // row_ror:2
        v = dpp_add<0x121, 0xf>(v);  // row_ror:1
        v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
        o[e] = v * unscale;
      }
      if ((lane & 31) == 16 + 4 * m + q) *(f32x4*)(dst + 16 * m + 4 * q) = o;
    }
}

// the same for a 64-wide tile (two accumulator blocks; dst = the layer's 64 sums + 32 * hh)
__device__ __forceinline__ void tile_colsum64(const f32x16 (&acc)[2], float* __restrict__ dst, int lane, float scale = 1.0f) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[m][4 * q + e];
        v = dpp_add<0x128, 0xf>(v);  // row_ror:8
        v = dpp_add<0x124, 0xf>(v);  // row_ror:4
        v = dpp_add<0x122, 0xf>(v);  // row_ror:2
        v = dpp_add<0x121, 0xf>(v);  // row_ror:1
        v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
        o[e] = v * scale;
      }
      if ((lane & 31) == 16 + 4 * m + q) *(f32x4*)(dst + 16 * m + 4 * q) = o;
    }
}

// acc = [act > 0] * acc (+ base), written to `dst`; act / dst address this lane's 64 features of its point.  `mask` (this lane's
// two words of the forward pass's ReLU-mask dump, dump_vec128) replaces the 16 loads of the activations by one 8-byte load: the
// chain needs the SIGN of an activation only, and reading the fp32 values back was 1.05 of the kernel's 2.81 ms on the C4 shard
template <bool ADD, bool SCALED = false>
__device__ __forceinline__ void mask_store(const float* __restrict__ act, const unsigned* __restrict__ mask, float* __restrict__ dst,
                                           bool ok, f32x16 (&acc)[4], const f32x16 (&base)[4], float unscale = 1.0f,
                                           _Float16* __restrict__ dst16 = nullptr, float scale16 = 1.0f) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 bits = {0u, 0u};
  if (mask != nullptr && ok) bits = *(const u32x2*)mask;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (mask == nullptr && ok) a = *(const f32x4*)(act + 16 * m + 4 * q);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v;
        if (mask != nullptr) {
          // the mask bit as an all-ones / all-zeros word (v_bfe_i32: a sign-extended 1-bit field) ANDed onto the gradient's bits
          const int keep = ((int)(bits[m >> 1] << (31 - (16 * (m & 1) + 4 * q + e)))) >> 31;
          v = __int_as_float(__float_as_int(acc[m][4 * q + e]) & keep);
        } else {
          v = a[e] > 0.f ? acc[m][4 * q + e] : 0.f;
        }
        if (ADD) v += base[m][4 * q + e];
        acc[m][4 * q + e] = v;
        o[e] = SCALED ? v * unscale : v;   // (a power of two: exact)
      }
      if (ok && dst != nullptr) *(f32x4*)(dst + 16 * m + 4 * q) = o;
    }
  if (dst16 != nullptr && ok) {   // (wave-uniform) the same values x 2^k as fp16: 8 stores of 16 bytes
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_pair_f16<false>(acc[m][8 * q + 2 * e] * scale16, acc[m][8 * q + 2 * e + 1] * scale16);
        *(u32x4*)(dst16 + 16 * m + 8 * q) = o;
      }
  }
}

template <int PREC>
__global__ void __launch_bounds__(NJF_THREADS, 2) resnetfc_backward_kernel(BackwardArgs a) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int tile = blockIdx.x * NJF_WAVES + wave;
  const int p = tile * 32 + j;
  const bool ok = p < a.points;
  const size_t pc = (size_t)min(p, a.points - 1);
  const size_t layer = (size_t)a.points * 128;
  const float* act = a.act ? a.act + pc * 128 + 64 * hh : nullptr;
  const unsigned* msk = a.masks ? a.masks + pc * 4 + 2 * hh : nullptr;
  const size_t mlayer = (size_t)a.points * 4;
  float* out = a.deltas + pc * 128 + 64 * hh;
  float* sums = a.colsum ? a.colsum + (size_t)tile * (11 * 128) + 64 * hh : nullptr;
  WeightStream st;
  stream_begin(st, a.w_pack, 21, 1, wave, lane);
  // PREC_F16X2 (opt-in; training.py: backward_precision): every product is hi*hi + hi*lo + lo*hi of fp16 halves, fp32-class (2^-22)
  // only inside fp16's range -- gradients are 1e-3 ... 1e-9.  The chain is LINEAR in d_out, so it runs on d_out * 2^k with k chosen
  // from max|d_out| (device scalar `absmax`, one reduction in front of the launch) such that the largest entry becomes 64: 2^9 of
  // head room before fp16's 65504, full 22 bits for entries down to 2^-9 of the largest and >= 11 bits down to 2^-20 of it (what
  // the smaller ones contribute to a weight gradient is below that in absolute terms); results are scaled back on the way out
  // (powers of two: exact).  The reference itself trains on TF32 products (train.py:64-65: 10 mantissa bits).
  constexpr bool SCALED = PREC != PREC_F32;
  float scale = 1.0f, unscale = 1.0f;   // of the chain's own arithmetic
  float pow2 = 1.0f;                    // 2^k of max|d_out| (what the fp16 deltas are scaled by, in either product form)
  if (SCALED || a.deltas16 != nullptr) {
    const float mx = a.absmax ? *a.absmax : 0.f;
    if (mx > 0.f && mx < 3.0e38f) {
      int e;
      frexpf(mx, &e);                      // mx = f * 2^e, f in [0.5, 1)
      const int k = max(min(6 - e, 120), -120);
      pow2 = ldexpf(1.0f, k);
      if (SCALED) {
        scale = pow2;
        unscale = ldexpf(1.0f, -k);
      }
    }
  }
  const float scale16 = SCALED ? 1.0f : pow2;   // the accumulators already carry 2^k in the split-precision form
  _Float16* out16 = a.deltas16 ? a.deltas16 + pc * 128 + 64 * hh : nullptr;
  const bool compact = a.deltas16 != nullptr;   // fp32 deltas: slices 0, 2, 4 only, stored as [3, P, 128]
  f32x16 din[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = 16 * hh + r;
    din[0][r] = (ok && d < a.d_out_dim) ? a.d_out[pc * a.d_out_dim + d] * scale : 0.f;
  }
  f32x16 delta[4], t[4], u[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) delta[m] = (f32x16)(0.f);
  {
    const float* wl = stream_step(st, wave, lane);
    mma_chunk<PREC, 4, 1, 0, false, 1>(st, wl, lane, din, delta);   // lin_out^T (first half of the chunk)
  }
  mask_store<false, SCALED>(act + 10 * layer, msk ? msk + 10 * mlayer : nullptr, compact ? nullptr : out + 10 * layer, ok, delta, delta,
                            unscale, out16 ? out16 + 10 * layer : nullptr, scale16);
  const bool live = tile * 32 < a.points;  // wave-uniform
  if (sums && live) tile_colsum(delta, sums + 10 * 128, lane, unscale);
  for (int blk = 4; blk >= 0; --blk) {
#pragma unroll
    for (int m = 0; m < 4; ++m) t[m] = (f32x16)(0.f);
    {
      const float* wl = stream_step(st, wave, lane);
      mma_chunk<PREC, 4, 2, 0, false, 4>(st, wl, lane, delta, t);
    }
    {
      const float* wl = stream_step(st, wave, lane);
      mma_chunk<PREC, 4, 2, 2, false, 4>(st, wl, lane, delta, t);
    }
    mask_store<false, SCALED>(act + (size_t)(2 * blk + 1) * layer, msk ? msk + (size_t)(2 * blk + 1) * mlayer : nullptr,
                              compact ? nullptr : out + (size_t)(2 * blk + 1) * layer, ok, t, t, unscale,
                              out16 ? out16 + (size_t)(2 * blk + 1) * layer : nullptr, scale16);
    if (sums && live) tile_colsum(t, sums + (2 * blk + 1) * 128, lane, unscale);
#pragma unroll
    for (int m = 0; m < 4; ++m) u[m] = (f32x16)(0.f);
    {
      const float* wl = stream_step(st, wave, lane);
      mma_chunk<PREC, 4, 2, 0, false, 4>(st, wl, lane, t, u);
    }
    {
      const float* wl = stream_step(st, wave, lane);
      mma_chunk<PREC, 4, 2, 2, false, 4>(st, wl, lane, t, u);
    }
    mask_store<true, SCALED>(act + (size_t)(2 * blk) * layer, msk ? msk + (size_t)(2 * blk) * mlayer : nullptr,
                             compact ? (blk < 3 ? out + (size_t)blk * layer : nullptr) : out + (size_t)(2 * blk) * layer, ok, u, delta,
                             unscale, out16 ? out16 + (size_t)(2 * blk) * layer : nullptr, scale16);
    if (sums && live) tile_colsum(u, sums + (2 * blk) * 128, lane, unscale);
#pragma unroll
    for (int m = 0; m < 4; ++m) delta[m] = u[m];
  }
}

extern "C" int njf_pack_resnetfc_backward(const NjfResnetFcWeights* src, float* w_out, int precision, void* stream) {
  if (!src || !w_out) return NJF_E_NULL;
  if (precision != NJF_PRECISION_F32 && precision != NJF_PRECISION_F16X2) return NJF_E_MODE;
  if (src->d_out < 1 || src->d_out > 32) return NJF_E_DOUT;
  if (!src->lin_out_w) return NJF_E_NULL;
  for (int i = 0; i < 5; ++i)
    if (!src->fc0_w[i] || !src->fc1_w[i]) return NJF_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int P = precision;
  PackScope scope(s);   // 11 transposed layers: one launch
  // chunk 0: lin_out^T  [128 x d_out -> 32], half a chunk; the other half is never read but is moved by the DMA
  launch_pack(src->lin_out_w, nullptr, 128, src->d_out, 4, 1, 3, P, w_out, nullptr, s);
  fill_kernel<<<16, 256, 0, s>>>(w_out + 4096, 4096, 0.f);
  for (int blk = 4, c = 1; blk >= 0; --blk, c += 4) {
    launch_pack(src->fc1_w[blk], nullptr, 128, 128, 4, 4, 3, P, w_out + (size_t)c * NJF_CHUNK_FLOATS, nullptr, s);
    launch_pack(src->fc0_w[blk], nullptr, 128, 128, 4, 4, 3, P, w_out + (size_t)(c + 2) * NJF_CHUNK_FLOATS, nullptr, s);
  }
  scope.flush();   // (before the status query: a failed batch launch must be this call's error)
  return launch_status();
}

// =============================================================================================
// Backward of the folded Jacobian transformer head (transformer_tile): the data-gradient chain of its three layers in ONE launch,
// in exact fp32 MFMA.  Per layer l (x = residual stream in front of it, dumped by the training forward; dx = gradient behind it):
//   forward again:  n = norm(x);  a = softmax_8(Mqk n + bqk);  xm = x + Nov a + bo;  n2 = norm(xm);  u = W1' n2 + b1';  h = gelu(u)
//   backward:       du  = (W2^T dx) * gelu'(u)                      dxm = dx + norm'(W1'^T du ; n2)
//                   ds  = softmax'(a ; Nov^T dxm)                   dx  = dxm + norm'(Mqk^T ds ; n)     (= gradient in front of l)
// and, like njf_resnetfc_backward, it EMITS per layer the pairs a weight gradient contracts over the points (K = points: one
// batched library GEMM on the host): X = (n, a, n2, h) and dY = (ds, dxm, du, dx) for (Mqk, Nov, W1', W2).  What autograd runs for
// the reference's parameterisation of the same head (transformer.py:38-135 recomputed in library ops: 54 ms of a 62 ms action
// step on SURVEY's C4 shard) -- the gradients of the FOLDED matrices go back to the reference's parameters through the fold's
// own autograd graph on the host (64 x 64 matrices).
// Weight blob (13 chunks, two 64 x 64 matrices each): [Wj^T | -] then for l = 2, 1, 0: [Mqk | Nov] [W1' | W2^T] [W1'^T | Nov^T]
// [Mqk^T | -];  bias blob [3][192] = (bqk | bo | b1') per layer.
// =============================================================================================

struct TransformerBwdArgs {
  const float* x;        // [4, P, 64] residual stream (NjfRenderOutputs.jac_act of a transformer-head training forward)
  const float* d_out;    // [P, d_out_dim]
  int d_out_dim, keys, points;
  const float* w_pack;   // NJF_TRANSFORMER_BACKWARD_CHUNKS chunks
  const float* b_pack;   // [3][192]
  float* wg_x;           // [12, P, 64]  X of (Mqk, Nov, W1', W2) of layer l at 4l + (0, 1, 2, 3)
  float* wg_dy;          // [12, P, 64]  dY, same order
  float* dx0;            // [P, 64] gradient w.r.t. the head's input (query MLP output)
  float* colsum;         // [tiles, 12, 64] per-tile column sums of the dY (the bias gradients), or nullptr
  // 16-bit training storage (training.py: what the weight-gradient GEMMs read): wg_x / wg_dy address HALVES, the dY are stored
  // x 2^k with k = 6 - exponent(max|d_out|) as in njf_resnetfc_backward (the caller divides the GEMM's result by 2^k)
  int half;
  const float* absmax;   // device scalar max|d_out| (half storage only)
};

// this lane's 32 values as halves (x scale): 4 stores of 16 bytes at row + 32 * hh (in halves)
__device__ __forceinline__ void store_vec64_f16(_Float16* __restrict__ dst, const f32x16 (&v)[2], float scale) {
  if (dst == nullptr) return;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_pair_f16<false>(v[m][8 * q + 2 * e] * scale, v[m][8 * q + 2 * e + 1] * scale);
      *(u32x4*)(dst + 16 * m + 8 * q) = o;
    }
}

// PREC_F16X2 (opt-in like resnetfc_backward_kernel<PREC_F16X2>; training.py: backward_precision): every product -- the layer's
// re-evaluation and the chain -- is hi*hi + hi*lo + lo*hi of fp16 halves.  The re-evaluated activations are O(1) (the forward pass
// runs the same products in the same form); the chain is linear in d_out and runs on d_out * 2^k, k = 6 - exponent(max|d_out|), the
// dY / dx0 / column sums are scaled back on the way out (fp16 pairs keep the 2^k, as in the exact form).
template <int PREC>
__global__ void __launch_bounds__(NJF_THREADS, 2) transformer_backward_kernel(TransformerBwdArgs a) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
  const int tile = blockIdx.x * NJF_WAVES + wave;
  const int p = tile * 32 + j;
  const bool ok = p < a.points;
  const size_t pc = (size_t)min(p, a.points - 1);
  const size_t slice = (size_t)a.points * 64;
  const size_t row = pc * 64 + 32 * hh;
  load_bias_block(a.b_pack, 3 * 192, 0);
  WeightStream st;
  stream_begin(st, a.w_pack, NJF_TRANSFORMER_BACKWARD_CHUNKS, 1, wave, lane);
  const float* bias = njf_lds + LDS_BIAS;
  float* const wx = !ok ? nullptr : (a.half ? (float*)((_Float16*)a.wg_x + row) : a.wg_x + row);
  float* const wy = !ok ? nullptr : (a.half ? (float*)((_Float16*)a.wg_dy + row) : a.wg_dy + row);
  constexpr bool SCALED = PREC != PREC_F32;
  float pow2 = 1.0f, unpow2 = 1.0f;   // 2^k of max|d_out| and its inverse (wave-uniform)
  if (SCALED || a.half) {
    const float mx = a.absmax ? *a.absmax : 0.f;
    if (mx > 0.f && mx < 3.0e38f) {
      int e;
      frexpf(mx, &e);
      const int k = max(min(6 - e, 120), -120);
      pow2 = ldexpf(1.0f, k);
      unpow2 = ldexpf(1.0f, -k);
    }
  }
  const bool half = a.half != 0;
  const float scale = SCALED ? pow2 : 1.0f;                                     // of the chain's own arithmetic
  const float unscale = SCALED ? unpow2 : 1.0f;                                 // what leaves the chain in fp32
  const float dy_scale = half ? (SCALED ? 1.0f : pow2) : unscale;               // of a stored dY
  // the (X, dY) pair slots: fp32, or halves under the 16-bit training storage (X as it is, dY x 2^k)
  auto put = [&](float* base, int k, const f32x16 (&v)[2], float scale) {
    if (base == nullptr) return;
    if (half) store_vec64_f16((_Float16*)base + (size_t)k * slice, v, scale);
    else if (SCALED && scale != 1.0f) {
      f32x16 w[2] = {v[0] * scale, v[1] * scale};
      store_vec64(base + (size_t)k * slice, w);
    } else store_vec64(base + (size_t)k * slice, v);
  };
  // (rows of padding lanes contribute nothing: their d_out is zero and the chain is linear in it)
  float* const sums = (a.colsum && tile * 32 < a.points) ? a.colsum + (size_t)tile * (12 * 64) + 32 * hh : nullptr;

  f32x16 din[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = 16 * hh + r;
    din[0][r] = (ok && d < a.d_out_dim) ? a.d_out[pc * a.d_out_dim + d] * scale : 0.f;
  }
  f32x16 dx[2], xin[2], n[2], t[2], n2[2], u[2];
  dx[0] = (f32x16)(0.f);
  dx[1] = (f32x16)(0.f);
  const float* wl = stream_step(st, wave, lane);
  mma_chunk<PREC, 2, 1, 0, false, 1>(st, wl, lane, din, dx);   // Wj^T: gradient behind layer 2
  for (int l = 2; l >= 0; --l) {
    const float* bl = bias + 192 * l;
    load_vec64(a.x + (size_t)l * slice + row, true, xin);
    // ---- the layer again -----------------------------------------------------------------------------------------
    const float rstd1 = norm64_rstd(xin, n);
    put(wx, 4 * l + 0, n, 1.0f);
    bias_init<2, true, PREC>(bl, hh, t);
    wl = stream_step(st, wave, lane);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, n, t);            // dots[head * 8 + key]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < a.keys) mx = fmaxf(mx, t[m][8 * h8 + k]);
        float e[8], sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          e[k] = (k < a.keys) ? __builtin_amdgcn_exp2f((t[m][8 * h8 + k] - mx) * 1.4426950408889634f) : 0.f;
          sum += e[k];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[m][8 * h8 + k] = e[k] * inv;       // a
      }
    put(wx, 4 * l + 1, t, 1.0f);
    bias_init<2, false, PREC>(bl + 64, hh, xin);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl + 4096, lane, t, xin);    // xm = x + Nov a + bo
    const float rstd2 = norm64_rstd(xin, n2);
    put(wx, 4 * l + 2, n2, 1.0f);
    bias_init<2, true, PREC>(bl + 128, hh, u);
    wl = stream_step(st, wave, lane);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, n2, u);           // u = W1' n2 + b1'
    {
      f32x16 hval[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = u[m][r];
          const float cdf = 0.5f * (1.0f + erf_branchless(v * 0.70710678118654752440f));
          hval[m][r] = v * cdf;                                             // gelu(u)
          // gelu'(u) = Phi(u) + u phi(u),  phi(u) = exp(-u^2 / 2) / sqrt(2 pi)
          u[m][r] = fmaf(v * 0.3989422804014327f, __builtin_amdgcn_exp2f(v * v * -0.7213475204444817f), cdf);
        }
      put(wx, 4 * l + 3, hval, 1.0f);
    }
    // ---- and backwards ---------------------------------------------------------------------------------------------
    put(wy, 4 * l + 3, dx, dy_scale);                                    // W2:  dY = dx
    if (sums) tile_colsum64(dx, sums + (4 * l + 3) * 64, lane, unscale);
    xin[0] = (f32x16)(0.f);
    xin[1] = (f32x16)(0.f);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl + 4096, lane, dx, xin);   // W2^T dx
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) u[m][r] *= xin[m][r];                     // du
    put(wy, 4 * l + 2, u, dy_scale);                                     // W1': dY = du
    if (sums) tile_colsum64(u, sums + (4 * l + 2) * 64, lane, unscale);
    xin[0] = (f32x16)(0.f);
    xin[1] = (f32x16)(0.f);
    wl = stream_step(st, wave, lane);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, u, xin);           // W1'^T du = dn2
    norm64_backward(xin, n2, rstd2, dx);                                     // dxm = dx + norm'(dn2)
    put(wy, 4 * l + 1, dx, dy_scale);                                    // Nov: dY = dxm
    if (sums) tile_colsum64(dx, sums + (4 * l + 1) * 64, lane, unscale);
    xin[0] = (f32x16)(0.f);
    xin[1] = (f32x16)(0.f);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl + 4096, lane, dx, xin);   // Nov^T dxm = da
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dot = fmaf(t[m][8 * h8 + k], xin[m][8 * h8 + k], dot);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[m][8 * h8 + k] *= xin[m][8 * h8 + k] - dot;   // ds = a (da - <a, da>)
      }
    put(wy, 4 * l + 0, t, dy_scale);                                     // Mqk: dY = ds
    if (sums) tile_colsum64(t, sums + (4 * l + 0) * 64, lane, unscale);
    xin[0] = (f32x16)(0.f);
    xin[1] = (f32x16)(0.f);
    wl = stream_step(st, wave, lane);
    mma_chunk<PREC, 2, 2, 0, false, 2>(st, wl, lane, t, xin);           // Mqk^T ds = dn
    norm64_backward(xin, n, rstd1, dx);                                      // gradient in front of layer l
  }
  if (ok) {
    if (SCALED) {
      dx[0] *= unscale;
      dx[1] *= unscale;
    }
    store_vec64(a.dx0 + row, dx);
  }
}

// =============================================================================================
// stand-alone sampler ops
// =============================================================================================
__global__ void __launch_bounds__(256) alpha_weights_kernel(const float* __restrict__ deltas,
                                                            const float* __restrict__ dens, int rays, int samples,
                                                            float* __restrict__ weights) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // one wave per ray, both halves duplicate
  const int lane = threadIdx.x & 63, j = lane & 31;
  if (wave >= rays) return;
  float carry = 0.f;
  for (int t = 0; t * 32 < samples; ++t) {
    const int s = t * 32 + j;
    const bool valid = s < samples;
    const size_t i = (size_t)wave * samples + min(s, samples - 1);
    const float w = tile_weights(deltas[i], dens[i], valid, j, carry);
    if (valid && lane < 32) weights[i] = w;
  }
}

extern "C" int njf_alpha_weights(const float* deltas, const float* densities, int rays, int samples, float* weights,
                                 void* stream) {
  if (!deltas || !densities || !weights) return NJF_E_NULL;
  if (rays < 1 || samples < 1) return NJF_E_SHAPE;
  alpha_weights_kernel<<<(rays + 3) / 4, 256, 0, (hipStream_t)stream>>>(deltas, densities, rays, samples, weights);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Backward of alpha compositing (training, perception mode): what autograd runs for RaySamples.get_weights
// (ray_samplers.py:77-101) + render_rgb + the un-clipped render_depth (model.py:257-279) -- where / mul / cumsum / cat /
// exp / exp / sub / mul / sum ... ~25 launches per level -- as ONE launch.  With ds_s = delta_s sigma_s (0 where delta <= 0),
// T_s = exp(-sum_{j<s} ds_j), w_s = (1 - exp(-ds_s)) T_s:
//   G_s        = g_w[s] + g_rgb . c_s + g_depth (t_s - depth) / (sum w + 1e-10)      (total derivative w.r.t. w_s)
//   dL/dds_k   = G_k T_{k+1} - sum_{s>k} G_s w_s                                      (dw_s/dds_s = T_{s+1}, dw_s/dds_k = -w_s)
//   dL/dsigma_k = delta_k dL/dds_k,   dL/dc_k = w_k g_rgb
// One wave per ray, lane = sample inside a tile of 64; a forward sweep over the tiles (prefix sums -> T, w, sum w, sum w t)
// and a reverse sweep (suffix sums of G w), both with fixed orders: bit-reproducible.
// ---------------------------------------------------------------------------------------------
struct CompositeBwdArgs {
  const float* deltas;   // [rays, S]
  const float* steps;    // [rays, S] sample mid-points t (only read with g_depth)
  const float* sigma;    // [rays, S]
  const float* color;    // [rays, S, 3] or NULL
  const float* g_w;      // [rays, S] or NULL
  const float* g_rgb;    // [rays, 3] or NULL
  const float* g_depth;  // [rays] or NULL (gradient w.r.t. the UN-clipped depth)
  int rays, samples;
  float* g_sigma;        // [rays, S]
  float* g_color;        // [rays, S, 3] or NULL
};

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#define NJF_COMPOSITE_MAX_TILES 16  // 1,024 samples per ray

__global__ void __launch_bounds__(256) composite_backward_kernel(CompositeBwdArgs a) {
  __shared__ float tile_start[4][NJF_COMPOSITE_MAX_TILES];  // per wave: optical depth in front of each 64-sample tile
  const int ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool live = ray < a.rays;
  const int S = a.samples, tiles = (S + 63) >> 6;
  const size_t base = (size_t)min(ray, a.rays - 1) * S;
  // forward sweep: the optical depth in front of every tile (accumulated front to back, like the forward pass -- taking it
  // from the ray's total by subtraction loses 1e-4 of the transmittance on opaque rays), sum w and sum w t (depth)
  float sum_w = 0.f, sum_wt = 0.f, carry = 0.f;
  const bool need_depth = a.g_depth != nullptr;
  for (int t = 0; t < tiles; ++t) {
    const int s = t * 64 + lane;
    const bool valid = s < S;
    const float delta = valid ? a.deltas[base + s] : 0.f;
    const float ds = (valid && delta > 0.f) ? delta * a.sigma[base + s] : 0.f;
    const float incl = wave_incl_scan(ds, lane);
    if (lane == 0) tile_start[wv][t] = carry;
    if (need_depth) {
      const float w = alpha_of(ds) * expf(-(carry + incl - ds));
      sum_w += valid ? w : 0.f;
      sum_wt += valid ? w * a.steps[base + s] : 0.f;
    }
    carry += __shfl(incl, 63, 64);
  }
  if (need_depth) {
    sum_w = wave_sum(sum_w);
    sum_wt = wave_sum(sum_wt);
  }
  __syncthreads();  // tile_start written by lane 0 of each wave, read by all of its lanes (uniform trip counts: S is per launch)
  if (!live) return;
  const float denom = sum_w + 1e-10f;
  const float depth = sum_wt / denom;
  const float gd = need_depth ? a.g_depth[ray] / denom : 0.f;
  float gr[3] = {0.f, 0.f, 0.f};
  if (a.g_rgb) {
    gr[0] = a.g_rgb[3 * (size_t)ray];
    gr[1] = a.g_rgb[3 * (size_t)ray + 1];
    gr[2] = a.g_rgb[3 * (size_t)ray + 2];
  }
  float behind = 0.f;   // sum of G_s w_s over the tiles already visited (samples further along the ray)
  for (int t = tiles - 1; t >= 0; --t) {
    const int s = t * 64 + lane;
    const bool valid = s < S;
    const float delta = valid ? a.deltas[base + s] : 0.f;
    const float ds = (valid && delta > 0.f) ? delta * a.sigma[base + s] : 0.f;
    const float incl = wave_incl_scan(ds, lane);
    const float before = tile_start[wv][t];             // optical depth in front of this tile
    const float t_next = expf(-(before + incl));        // T_{s+1}
    const float w = alpha_of(ds) * expf(-(before + incl - ds));
    float G = a.g_w ? (valid ? a.g_w[base + s] : 0.f) : 0.f;
    float c[3] = {0.f, 0.f, 0.f};
    if (a.color && valid) {
      c[0] = a.color[3 * (base + s)];
      c[1] = a.color[3 * (base + s) + 1];
      c[2] = a.color[3 * (base + s) + 2];
      G += gr[0] * c[0] + gr[1] * c[1] + gr[2] * c[2];
    }
    if (need_depth && valid) G += gd * (a.steps[base + s] - depth);
    const float gw = valid ? G * w : 0.f;
    // exclusive suffix sum inside the tile = tile sum - inclusive prefix sum
    const float pre = wave_incl_scan(gw, lane);
    const float tile_gw = __shfl(pre, 63, 64);
    const float suffix = behind + (tile_gw - pre);
    if (valid) {
      a.g_sigma[base + s] = delta > 0.f ? delta * (G * t_next - suffix) : 0.f;
      if (a.g_color) {
        a.g_color[3 * (base + s)] = w * gr[0];
        a.g_color[3 * (base + s) + 1] = w * gr[1];
        a.g_color[3 * (base + s) + 2] = w * gr[2];
      }
    }
    behind += tile_gw;
  }
}

extern "C" int njf_composite_backward(const float* deltas, const float* steps, const float* sigma, const float* color,
                                      const float* g_weights, const float* g_rgb, const float* g_depth, int rays, int samples,
                                      float* g_sigma, float* g_color, void* stream) {
  if (!deltas || !sigma || !g_sigma) return NJF_E_NULL;
  if (g_depth && !steps) return NJF_E_NULL;
  if ((g_rgb || g_color) && !color) return NJF_E_NULL;
  if (rays < 1 || samples < 1) return NJF_E_SHAPE;
  if (samples > 64 * NJF_COMPOSITE_MAX_TILES) return NJF_E_SAMPLES;
  CompositeBwdArgs a{deltas, steps, sigma, color, g_weights, g_rgb, g_depth, rays, samples, g_sigma, g_color};
  composite_backward_kernel<<<(rays + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
  return launch_status();
}

struct PdfArgs {
  const float* weights;
  const float* bins_in;
  int bins_per_ray, s_in;
  const float* u;
  int u_per_ray, s_out;
  float anneal;
  int rays;
  float* bins_out;
};

__global__ void __launch_bounds__(NJF_THREADS) pdf_kernel(PdfArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * NJF_WAVES + wave;
  const bool ok = ray < a.rays;
  const int rayc = min(ray, a.rays - 1);
  float* sc = njf_lds + LDS_SCRATCH_PROPOSAL + wave * LDS_SCRATCH_PER_WAVE;
  for (int s = lane; s < a.s_in; s += 64) {
    float w = a.weights[(size_t)rayc * a.s_in + s];
    if (a.anneal != 1.0f) w = powf(w, a.anneal);
    sc[s] = w + 0.01f;
  }
  __syncthreads();
  const float* bins = a.bins_in + (a.bins_per_ray ? (size_t)rayc * (a.s_in + 1) : 0);
  const float* u = a.u + (a.u_per_ray ? (size_t)rayc * (a.s_out + 1) : 0);
  pdf_resample_ray(sc, bins, a.s_in, u, a.s_out, a.bins_out + (size_t)rayc * (a.s_out + 1), lane, ok);
}

extern "C" int njf_pdf_resample(const float* weights, const float* bins_in, int bins_per_ray, int s_in, const float* u,
                                int u_per_ray, int s_out, float anneal, int rays, float* bins_out, void* stream) {
  if (!weights || !bins_in || !u || !bins_out) return NJF_E_NULL;
  if (rays < 1 || s_out < 1) return NJF_E_SHAPE;
  if (s_in < 1 || s_in > 256) return NJF_E_SAMPLES;
  PdfArgs a{weights, bins_in, bins_per_ray, s_in, u, u_per_ray, s_out, anneal, rays, bins_out};
  pdf_kernel<<<(rays + NJF_WAVES - 1) / NJF_WAVES, NJF_THREADS, LDS_FLOATS_PROPOSAL * sizeof(float), (hipStream_t)stream>>>(a);
  return launch_status();
}

// =============================================================================================
// launchers of the fused kernels
// =============================================================================================
static int check_common(const float* origins, const float* directions, int rays_per_batch, const NjfCameras* cams,
                        const NjfFeatureMap* gmap) {
  if (!origins || !directions || !cams || !gmap) return NJF_E_NULL;
  if (!cams->ctxt_w2c || !cams->ctxt_k || !cams->z_near || !cams->z_far || !gmap->data) return NJF_E_NULL;
  if (rays_per_batch < 1 || cams->batch < 1 || gmap->height < 1 || gmap->width < 1) return NJF_E_SHAPE;
  if ((long long)rays_per_batch * cams->batch > 0x7fffffffLL / 64) return NJF_E_SHAPE;
  if ((long long)gmap->height * gmap->width * gmap->stride > 0x7fffffffLL) return NJF_E_SHAPE;  // texel offsets are int32
  return NJF_OK;
}

// `precision`: of the network that reads the block (a plain-fp16 network reads a map of halves: 16-byte pieces = 8 elements)
static int check_gmap(const NjfFeatureMap* gmap, int off, int channels = NJF_ZDIM, int precision = NJF_PRECISION_F32) {
  const int mask = precision == NJF_PRECISION_F16 ? 7 : 3;
  if (off < 0 || (off & mask) || (gmap->stride & mask) || off + channels > gmap->stride) return NJF_E_GMAP;
  return NJF_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is needed once per (kernel, device), not per launch: a lock-free
// set of the function pointers already raised, per device (every fused kernel is always launched with the same LDS size).
static std::atomic<const void*> g_lds_raised[16][256];

static bool lds_attribute_known(const void* fn) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  const size_t h = ((uintptr_t)fn >> 4) & 255;
  for (int i = 0; i < 256; ++i) {
    const void* v = g_lds_raised[dev][(h + i) & 255].load(std::memory_order_acquire);
    if (v == fn) return true;
    if (v == nullptr) return false;
  }
  return false;
}

static void lds_attribute_remember(const void* fn) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
  const size_t h = ((uintptr_t)fn >> 4) & 255;
  for (int i = 0; i < 256; ++i) {
    const void* expected = nullptr;
    std::atomic<const void*>& slot = g_lds_raised[dev][(h + i) & 255];
    if (slot.compare_exchange_strong(expected, fn, std::memory_order_acq_rel) || expected == fn) return;
  }
}

#ifdef NJF_STAMPS
extern "C" int njf_debug_read_stamps(unsigned* host_out, int n) {
  if (n > NJF_STAMP_SLOTS) n = NJF_STAMP_SLOTS;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(njf_stamp_out), (size_t)n * sizeof(unsigned));
}
#endif

template <typename K, typename A>
static int launch_fused(K kernel, const A& args, int work_items, hipStream_t s, int lds_floats = LDS_FLOATS_RENDER) {
  static_assert(sizeof(A) <= 4096, "kernel args too large");
  const int grid = (work_items + NJF_WAVES - 1) / NJF_WAVES;
#ifdef NJF_STAMPS
  const size_t lds = (size_t)(lds_floats + NJF_STAMP_SLOTS) * sizeof(float);
#else
  const size_t lds = (size_t)lds_floats * sizeof(float);
#endif
  if (!lds_attribute_known((const void*)kernel)) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_attribute_remember((const void*)kernel);
  }
  kernel<<<grid, NJF_THREADS, lds, s>>>(args);
  return launch_status();
}

// run `f(std::integral_constant<int, PREC_*>)` for the MFMA precision selected at run time.  TRAIN: the dispatch of the
// training forwards (activation dumps), which do not exist in the plain-fp16 mode -- NJF_E_MODE there.
template <bool TRAIN = false, typename F>
static int with_precision(int precision, F&& f) {
#ifdef NJF_DEV_ONLY_PREC  // development builds only (static ISA checks of one precision: a third of the compile time)
  if constexpr (TRAIN && NJF_DEV_ONLY_PREC == PREC_F16) return NJF_E_MODE;
  else return f(std::integral_constant<int, NJF_DEV_ONLY_PREC>{});
#else
  if (precision == NJF_PRECISION_F16X2) return f(std::integral_constant<int, PREC_F16X2>{});
  if (precision == NJF_PRECISION_F16F6) return f(std::integral_constant<int, PREC_F16F6>{});
  if (precision == NJF_PRECISION_F16) {
    if constexpr (TRAIN) return NJF_E_MODE;
    else return f(std::integral_constant<int, PREC_F16>{});
  }
  return f(std::integral_constant<int, PREC_F32>{});
#endif
}
// ... and `f(P_density, P_jacobian)` for the decoder kernels, whose Jacobian head may run in the other split precision
template <bool TRAIN = false, typename F>
static int with_precisions(int precision, F&& f) {
  const int d = density_precision(precision), j = jacobian_precision(precision);
#ifdef NJF_DEV_ONLY_PREC
  return with_precision<TRAIN>(d, [&](auto P) { return f(P, P); });
#else
  if (d == j) return with_precision<TRAIN>(d, [&](auto P) { return f(P, P); });
  if (d == NJF_PRECISION_F16F6) return f(std::integral_constant<int, PREC_F16F6>{}, std::integral_constant<int, PREC_F16X2>{});
  return f(std::integral_constant<int, PREC_F16X2>{}, std::integral_constant<int, PREC_F16F6>{});
#endif
}
#define NJF_P decltype(P)::value
#define NJF_PJ decltype(PJ)::value

extern "C" int njf_proposal_forward(const float* origins, const float* directions, int rays_per_batch,
                                    const NjfCameras* cams, const NjfFeatureMap* gmap, int gmap_offset,
                                    const float* w_pack, const float* b_pack, const float* bins_in, int bins_per_ray,
                                    int s_in, const float* u, int u_per_ray, int s_out, float anneal, float* bins_out,
                                    float* weights_out, float* density_out, const NjfActivationDump* dump,
                                    int precision, void* stream) {
  int rc = check_common(origins, directions, rays_per_batch, cams, gmap);
  if (rc) return rc;
  if (!w_pack || !b_pack || !bins_in || !u || !bins_out) return NJF_E_NULL;
  if (s_in < 1 || s_in > 256 || s_out < 1) return NJF_E_SAMPLES;
  if (!valid_base_precision(precision)) return NJF_E_MODE;
  if ((rc = check_gmap(gmap, gmap_offset, NJF_ZDIM, precision))) return rc;
  ProposalArgs a;
  a.rc = RayCommon{origins, directions, rays_per_batch, rays_per_batch * cams->batch, *cams, *gmap};
  a.gmap_offset = gmap_offset;
  a.w_pack = w_pack;
  a.b_pack = b_pack;
  a.bins_in = bins_in;
  a.bins_per_ray = bins_per_ray;
  a.s_in = s_in;
  a.u = u;
  a.u_per_ray = u_per_ray;
  a.s_out = s_out;
  a.anneal = anneal;
  a.bins_out = bins_out;
  a.weights_out = weights_out;
  a.density_out = density_out;
  a.dump = NjfActivationDump{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (dump != nullptr && dump->act != nullptr) {  // training forward: inputs of the proposal net's backward pass
    if (!dump->pe || !dump->foot_idx || !dump->foot_w) return NJF_E_NULL;
    a.dump = *dump;
    return with_precision<true>(precision, [&](auto P) {
      return launch_fused(proposal_kernel<NJF_P, true>, a, a.rc.total_rays, (hipStream_t)stream, LDS_FLOATS_PROPOSAL);
    });
  }
  return with_precision(precision, [&](auto P) {
    return launch_fused(proposal_kernel<NJF_P>, a, a.rc.total_rays, (hipStream_t)stream, LDS_FLOATS_PROPOSAL);
  });
}

// The decoder blobs must be one allocation laid out [density | colour | jacobian] (what
// njf_pack_* write when given consecutive destinations); the launcher verifies contiguity.
static int check_jacobian(int kind, const NjfCameras* cams, const NjfFeatureMap* gmap, int goff_j, const float* w_j,
                          const float* b_j, int precision) {
  if (kind == NJF_JACOBIAN_NONE) return NJF_OK;
  if (kind != NJF_JACOBIAN_MLP && kind != NJF_JACOBIAN_TRANSFORMER) return NJF_E_MODE;
  if (!w_j || !b_j) return NJF_E_NULL;
  const int max_a = kind == NJF_JACOBIAN_MLP ? NJF_MAX_ACTION_DIM : 8;  // transformer: 8 key slots per head
  if (cams->action_dim < 1 || cams->action_dim > max_a) return NJF_E_ACTION_DIM;
  return check_gmap(gmap, goff_j, kind == NJF_JACOBIAN_MLP ? NJF_ZDIM : NJF_QDIM, jacobian_precision(precision));
}

static int check_contiguous(const float* w_d, const float* w_c, const float* w_j, bool with_j) {
  if (w_c != w_d + NJF_RESNET_W_FLOATS) return NJF_E_SHAPE;
  if (with_j && w_j != w_c + NJF_COLOR_W_FLOATS) return NJF_E_SHAPE;
  return NJF_OK;
}

extern "C" int njf_render_forward(const float* origins, const float* directions, int rays_per_batch,
                                  const NjfCameras* cams, const NjfFeatureMap* gmap, int gmap_offset_density,
                                  int gmap_offset_jacobian, int jacobian_kind, const float* w_density,
                                  const float* b_density, const float* w_color, const float* b_color,
                                  const float* w_jacobian, const float* b_jacobian, const float* bins, int samples,
                                  const NjfRenderOutputs* out, int precision, void* stream) {
  int rc = check_common(origins, directions, rays_per_batch, cams, gmap);
  if (rc) return rc;
  if (!w_density || !b_density || !w_color || !b_color || !bins || !out) return NJF_E_NULL;
  if (samples < 1) return NJF_E_SAMPLES;
  if (!valid_precision(precision)) return NJF_E_MODE;
  if ((rc = check_gmap(gmap, gmap_offset_density, NJF_ZDIM, density_precision(precision)))) return rc;
  if ((rc = check_jacobian(jacobian_kind, cams, gmap, gmap_offset_jacobian, w_jacobian, b_jacobian, precision))) return rc;
  const bool with_j = jacobian_kind != NJF_JACOBIAN_NONE;
  if ((rc = check_contiguous(w_density, w_color, w_jacobian, with_j))) return rc;
  RenderArgs a;
  a.rc = RayCommon{origins, directions, rays_per_batch, rays_per_batch * cams->batch, *cams, *gmap};
  a.goff_d = gmap_offset_density;
  a.goff_j = with_j ? gmap_offset_jacobian : gmap_offset_density;
  a.w_all = w_density;
  a.b_d = b_density;
  a.b_c = b_color;
  a.b_j = b_jacobian;
  a.bins = bins;
  a.samples = samples;
  a.out = *out;
  hipStream_t s = (hipStream_t)stream;
  const int n = a.rc.total_rays;
  // Training forwards keep the AF = true instantiations (the 16 action-feature accumulators stay allocated although no training
  // forward composites them any more, model.py::_vis_at_bins).  Round 5 built the AF = false ones to get the dump kernels off the
  // spill path -- 104-119 spilled VGPRs -> 45-68 -- and measured them SLOWER or equal on the C4 shard (1 x 8,192 rays: action step
  // 8.56 vs 8.34 ms, perception 13.63 vs 13.67 ms, two repetitions each, profiles/r05_spills_ab.txt): the spills are not what
  // these kernels wait for.  -DNJF_TRAIN_NO_AF rebuilds the A/B.
  constexpr bool TRAIN_AF = true;
  const bool training_forward = out->jac_act != nullptr || out->jac_pe != nullptr || out->den_act != nullptr;
  if (!TRAIN_AF && training_forward && out->action_features != nullptr) return NJF_E_MODE;
  if (out->jac_act != nullptr || (out->jac_pe != nullptr && out->den_act == nullptr)) {
    // action-mode training forward: dump the Jacobian head's backward-pass inputs (ResnetFC head: activations +
    // encoding + footprint; transformer head: encoding + footprint, the head is recomputed by the backward pass)
    if (jacobian_kind == NJF_JACOBIAN_NONE || out->den_act) return NJF_E_MODE;
    if (!out->jac_pe || !out->foot_idx || !out->foot_w) return NJF_E_NULL;
    if (jacobian_kind == NJF_JACOBIAN_MLP) {
      if (!out->jac_act) return NJF_E_NULL;
      return with_precisions<true>(precision, [&](auto P, auto PJ) { return launch_fused(render_kernel<1, NJF_P, 1, TRAIN_AF, NJF_PJ>, a, n, s); });
    }
    // (transformer head: jac_act is optional and has another shape, [4, P, 64]: the residual stream njf_transformer_backward reads)
    if (out->dump_f16 || out->jac_mask) return NJF_E_MODE;
    return with_precisions<true>(precision, [&](auto P, auto PJ) { return launch_fused(render_kernel<2, NJF_P, 1, TRAIN_AF, NJF_PJ>, a, n, s); });
  }
  if (out->den_act != nullptr) {  // perception-mode training forward: dump the density net and the colour head
    if (!out->jac_pe || !out->foot_idx || !out->foot_w || !out->col_in || !out->col_act) return NJF_E_NULL;
    if (jacobian_kind == NJF_JACOBIAN_NONE)
      return with_precision<true>(density_precision(precision), [&](auto P) { return launch_fused(render_kernel<0, NJF_P, 2, TRAIN_AF>, a, n, s); });
    return with_precisions<true>(precision, [&](auto P, auto PJ) {
      if (jacobian_kind == NJF_JACOBIAN_MLP) return launch_fused(render_kernel<1, NJF_P, 2, TRAIN_AF, NJF_PJ>, a, n, s);
      return launch_fused(render_kernel<2, NJF_P, 2, TRAIN_AF, NJF_PJ>, a, n, s);
    });
  }
  const bool af = with_j && out->action_features != nullptr;
  if (jacobian_kind == NJF_JACOBIAN_NONE)
    return with_precision(density_precision(precision), [&](auto P) { return launch_fused(render_kernel<0, NJF_P, 0, false>, a, n, s); });
  return with_precisions(precision, [&](auto P, auto PJ) {
    if (jacobian_kind == NJF_JACOBIAN_MLP)
      return af ? launch_fused(render_kernel<1, NJF_P, 0, true, NJF_PJ>, a, n, s) : launch_fused(render_kernel<1, NJF_P, 0, false, NJF_PJ>, a, n, s);
    return af ? launch_fused(render_kernel<2, NJF_P, 0, true, NJF_PJ>, a, n, s) : launch_fused(render_kernel<2, NJF_P, 0, false, NJF_PJ>, a, n, s);
  });
}

extern "C" int njf_points_forward(const float* xyz, const float* dirs, int points_per_batch, const NjfCameras* cams,
                                  const NjfFeatureMap* gmap, int gmap_offset_density, int gmap_offset_jacobian, int mode,
                                  int jacobian_kind, const float* w_density, const float* b_density, const float* w_color,
                                  const float* b_color, const float* w_jacobian, const float* b_jacobian, float* density,
                                  float* color, float* flow, float* jacobian, float* geo, float* features, int precision,
                                  void* stream) {
  if (!xyz || !cams || !gmap || !w_density || !b_density) return NJF_E_NULL;
  if (!cams->ctxt_w2c || !cams->ctxt_k || !gmap->data) return NJF_E_NULL;
  if (points_per_batch < 1 || cams->batch < 1) return NJF_E_SHAPE;
  if (mode != 0 && mode != 1) return NJF_E_MODE;
  if (!valid_precision(precision)) return NJF_E_MODE;
  int rc;
  if ((rc = check_gmap(gmap, gmap_offset_density, NJF_ZDIM, density_precision(precision)))) return rc;
  // a point carries the float index of its batch element in 32 bits (PointGeom::gofs): maps up to 16 GiB
  if ((long long)cams->batch * gmap->height * gmap->width * gmap->stride > 0xffffffffLL) return NJF_E_SHAPE;
  PointsArgs a;
  a.xyz = xyz;
  a.dirs = dirs;
  a.points_per_batch = points_per_batch;
  a.total_points = points_per_batch * cams->batch;
  a.cams = *cams;
  a.gmap = *gmap;
  a.goff_d = gmap_offset_density;
  a.goff_j = gmap_offset_jacobian;
  a.w_all = w_density;
  a.b_d = b_density;
  a.b_c = b_color;
  a.b_j = b_jacobian;
  a.density = density;
  a.color = color;
  a.flow = flow;
  a.jacobian = jacobian;
  a.geo = geo;
  a.features = features;
  // the per-block features exist for a ResnetFC head evaluated next to the decoder (mode 1, NJF_JACOBIAN_MLP)
  if (features && (mode != 1 || jacobian_kind != NJF_JACOBIAN_MLP)) return NJF_E_MODE;
  if (features && (long long)a.total_points * 5 * 128 > 0x7fffffffffLL) return NJF_E_SHAPE;
  const int tiles = (a.total_points + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0)
    return with_precision(density_precision(precision), [&](auto P) { return launch_fused(points_kernel<0, NJF_P>, a, tiles, s); });
  if (!w_color || !b_color) return NJF_E_NULL;
  if ((rc = check_jacobian(jacobian_kind, cams, gmap, gmap_offset_jacobian, w_jacobian, b_jacobian, precision))) return rc;
  const bool with_j = jacobian_kind != NJF_JACOBIAN_NONE;
  if ((rc = check_contiguous(w_density, w_color, w_jacobian, with_j))) return rc;
  if (!with_j)
    return with_precision(density_precision(precision), [&](auto P) { return launch_fused(points_kernel<1, NJF_P>, a, tiles, s); });
  return with_precisions(precision, [&](auto P, auto PJ) {
    if (features) {   // (the plain-fp16 mode has no training instantiation of resnet_tile: its block biases ride elsewhere)
      if constexpr (NJF_PJ == PREC_F16) return (int)NJF_E_MODE;
      else return launch_fused(points_kernel<2, NJF_P, NJF_PJ, true>, a, tiles, s);
    }
    if (jacobian_kind == NJF_JACOBIAN_MLP) return launch_fused(points_kernel<2, NJF_P, NJF_PJ>, a, tiles, s);
    return launch_fused(points_kernel<3, NJF_P, NJF_PJ>, a, tiles, s);
  });
}

extern "C" int njf_resnetfc_backward(const float* d_out, int d_out_dim, const float* activations, const float* w_backward,
                                     int points, float* deltas, float* colsum_partial, const unsigned* masks, int precision,
                                     const float* d_out_absmax, void* deltas16, void* stream) {
  if (!d_out || (!activations && !masks) || !w_backward || !deltas) return NJF_E_NULL;
  if (deltas16 && (!masks || !d_out_absmax)) return NJF_E_NULL;   // the 16-bit storage form reads masks and needs the scale
  if (points < 1 || (long long)points * 11 * 128 > 0x7fffffffffLL) return NJF_E_SHAPE;
  if (d_out_dim < 1 || d_out_dim > 32) return NJF_E_DOUT;
  BackwardArgs a{d_out, d_out_dim, activations, w_backward, points, deltas, colsum_partial, masks, d_out_absmax, (_Float16*)deltas16};
  if (precision == NJF_PRECISION_F16X2) {
    if (!d_out_absmax) return NJF_E_NULL;
    return launch_fused(resnetfc_backward_kernel<PREC_F16X2>, a, (points + 31) / 32, (hipStream_t)stream);
  }
  if (precision != NJF_PRECISION_F32) return NJF_E_MODE;
  return launch_fused(resnetfc_backward_kernel<PREC_F32>, a, (points + 31) / 32, (hipStream_t)stream);
}

extern "C" int njf_pack_transformer_backward(const float* mats, const float* biases, const float* head_w, int d_out,
                                             float* w_out, float* b_out, int precision, void* stream) {
  // mats [3][4][64][64] row-major [out][in] = (Mqk, Nov, W1', W2) per layer; biases [3][3][64] = (bqk, bo, b1') per layer;
  // head_w [d_out][64] = the output Linear.  w_out: NJF_TRANSFORMER_BACKWARD_CHUNKS chunks, b_out [3][192]
  if (!mats || !biases || !head_w || !w_out || !b_out) return NJF_E_NULL;
  if (d_out < 1 || d_out > 32) return NJF_E_DOUT;
  if (precision != NJF_PRECISION_F32 && precision != NJF_PRECISION_F16X2) return NJF_E_MODE;
  hipStream_t s = (hipStream_t)stream;
  const int P = precision;
  fill_kernel<<<64, 256, 0, s>>>(w_out, NJF_TRANSFORMER_BACKWARD_CHUNKS * NJF_CHUNK_FLOATS, 0.f);
  PackScope scope(s);   // 22 matrices: one launch, after the fill
  launch_pack(head_w, nullptr, 64, d_out, 2, 1, 3, P, w_out, nullptr, s);                      // Wj^T: 64 rows, K = d_out (<= 32)
  for (int l = 2, c = 1; l >= 0; --l, c += 4) {
    const float* m = mats + (size_t)l * 4 * 4096;
    const float* b = biases + (size_t)l * 192;
    float* base = w_out + (size_t)c * NJF_CHUNK_FLOATS;
    launch_pack(m, b, 64, 64, 2, 2, 0, P, base, b_out + 192 * l, s);                           // Mqk (+ bqk)
    launch_pack(m + 4096, b + 64, 64, 64, 2, 2, 0, P, base + 4096, b_out + 192 * l + 64, s);   // Nov (+ bo)
    launch_pack(m + 2 * 4096, b + 128, 64, 64, 2, 2, 0, P, base + NJF_CHUNK_FLOATS, b_out + 192 * l + 128, s);   // W1' (+ b1')
    launch_pack(m + 3 * 4096, nullptr, 64, 64, 2, 2, 3, P, base + NJF_CHUNK_FLOATS + 4096, nullptr, s);          // W2^T
    launch_pack(m + 2 * 4096, nullptr, 64, 64, 2, 2, 3, P, base + 2 * NJF_CHUNK_FLOATS, nullptr, s);             // W1'^T
    launch_pack(m + 4096, nullptr, 64, 64, 2, 2, 3, P, base + 2 * NJF_CHUNK_FLOATS + 4096, nullptr, s);          // Nov^T
    launch_pack(m, nullptr, 64, 64, 2, 2, 3, P, base + 3 * NJF_CHUNK_FLOATS, nullptr, s);                        // Mqk^T
  }
  scope.flush();   // (before the status query: a failed batch launch must be this call's error)
  return launch_status();
}

extern "C" int njf_transformer_backward(const float* x, const float* d_out, int d_out_dim, int keys, int points,
                                        const float* w_backward, const float* b_backward, float* wg_x, float* wg_dy,
                                        float* dx0, float* colsum_partial, int half_storage, const float* d_out_absmax,
                                        int precision, void* stream) {
  if (!x || !d_out || !w_backward || !b_backward || !wg_x || !wg_dy || !dx0) return NJF_E_NULL;
  if (points < 1 || (long long)points * 12 * 64 > 0x7fffffffffLL) return NJF_E_SHAPE;
  if (d_out_dim < 1 || d_out_dim > 32) return NJF_E_DOUT;
  if (keys < 1 || keys > 8) return NJF_E_ACTION_DIM;
  if (half_storage && !d_out_absmax) return NJF_E_NULL;
  TransformerBwdArgs a{x, d_out, d_out_dim, keys, points, w_backward, b_backward, wg_x, wg_dy, dx0, colsum_partial,
                       half_storage != 0, d_out_absmax};
  if (precision == NJF_PRECISION_F16X2) {
    if (!d_out_absmax) return NJF_E_NULL;
    return launch_fused(transformer_backward_kernel<PREC_F16X2>, a, (points + 31) / 32, (hipStream_t)stream);
  }
  if (precision != NJF_PRECISION_F32) return NJF_E_MODE;
  return launch_fused(transformer_backward_kernel<PREC_F32>, a, (points + 31) / 32, (hipStream_t)stream);
}

