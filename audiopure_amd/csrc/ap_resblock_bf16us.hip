// AP_PREC_BF16_STORE, small batches: Residual_block.forward (WaveNet.py:75-97) on the u images of ap_resblock_bf16u.hip with ONE
// 64-sample tile per workgroup, for launches that have at most one 128-sample tile per CU (one- and two-clip calls of the shipped
// length).  There the persistent kernel gives half the CUs one tile each and the launch lasts one tile's latency; half-size tiles on
// twice as many workgroups halve the work per CU (the reasoning and the measurements of ap_resblock_bf16s.hip, AP_PREC_BF16's twin).
// Results are BIT-IDENTICAL to the persistent kernel's (a clip's result must not depend on the batch it travels in):
//   * GEMM1: accumulators start from b1; the same bf16 operands (a chunk's (column, tap) operand is one 64-byte row of the image, channel
//     positions in the image's own order; the weight image packed with the same K permutation) enter the same
//     v_mfma_f32_32x32x16_bf16 sequence per output element, chunk by chunk, six k-steps each;
//   * the gate: gate_pair_u's arithmetic on the same channel pairs, bf16 RNE;
//   * GEMM2 (res_conv): accumulators start from b2, sixteen k-steps in order; u' = bf16((u + acc) sqrt(1/2) + part_t of the next layer)
//     with the three operations rounded separately (fp contract off), as in the persistent kernel's epilogue.
// tests: one- and two-clip chains against the 512-clip batch bit for bit (tests/test_gpu_bf16_store.py), kernel against kernel on random
// shapes (tools/fuzz_blocks.py, tests/test_gpu_fuzz.py).  Built for res = skip = 256 channels.
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int UC_ = 256;                  // res = skip channels
constexpr int UXS_ = 96 + 8;              // bf16 per column row of the X chunk image (3 taps x 32 channel positions; 208-byte rows)
constexpr int UGS_ = 256 + 8;             // bf16 per column row of the g image (528-byte rows)
constexpr unsigned UFR_ = 64 * 16;        // bytes of one row tile's fragment of a k-step

// tanh(a) sigmoid(b): ap_resblock_bf16u.hip's gate_pair_u, operation for operation (the results must be its results)
__device__ __forceinline__ f32x2 gate_pair_us(f32x2 a, f32x2 b) {
  const f32x2 ac = {__builtin_amdgcn_fmed3f(a[0], -16.0f, 16.0f), __builtin_amdgcn_fmed3f(a[1], -16.0f, 16.0f)};
  const f32x2 ea = ac * -2.885390081777926815f;
  const f32x2 eb = b * -1.442695040888963407f;
  const f32x2 E = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
  const f32x2 F = {__builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
  const f32x2 den = (E + 1.0f) * (F + 1.0f);
  const f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  return (1.0f - E) * r;
}

}  // namespace

// One workgroup = one (clip, 64-sample tile); 8 waves, wave w = gate channels [32 w, 32 w + 32) in GEMM1 and res rows [32 w, 32 w + 32)
// (= chunk w of the image) in GEMM2.  w1 / w2: this layer's images (pack_w1_bf16_kernel with the image's K permutation:
// [wave][chunk 8][k-step 6][row tile 2][lane][8]; pack_w2_bf16_kernel: [wave][row tile 2][k-step 16][lane][8], row tile 0 = res rows).
template <bool NOH>
__global__ __launch_bounds__(512, 4) void resblock_bf16us_kernel(const void *__restrict__ uin, void *__restrict__ uout, const float *__restrict__ ptn,
                                                                 void *__restrict__ gout, const __bf16 *__restrict__ w1, const __bf16 *__restrict__ w2,
                                                                 const float *__restrict__ b1, const float *__restrict__ b2, int L, int d, int ntiles) {
  constexpr int C = UC_;
  constexpr int NT = 64;
  constexpr int XB = NT * UXS_;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * XB + NT * UGS_];   // 26.6 KB X ring + 33.8 KB g image
  __bf16 *gim = lds + 2 * XB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / ntiles));
  const int t0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % ntiles) * NT);
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 2u;                 // a clip's image: [C / 32][L][32] bf16
  const __amdgpu_buffer_rsrc_t urs = uni_rsrc(reinterpret_cast<const char *>(uin) + (size_t)b * clip_bytes, clip_bytes);
  const __amdgpu_buffer_rsrc_t w1rs = uni_rsrc(reinterpret_cast<const char *>(w1) + (size_t)wave * (8 * 6 * 2 * UFR_), 8 * 6 * 2 * UFR_);
  const __amdgpu_buffer_rsrc_t w2rs = uni_rsrc(reinterpret_cast<const char *>(w2) + (size_t)wave * (2 * 16 * UFR_), 16 * UFR_);   // row tile 0 only
  const unsigned lane16 = (unsigned)lane * 16u;

  // ---- GEMM1 staging (pure data movement): thread = (column sj, 16-byte piece so of the 64-byte row); threads 0-255 move taps 0 and 1,
  // threads 256-511 tap 2
  const int sj = tid & 63, so = (tid >> 6) & 3, sh = tid >> 8;
  unsigned xv[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int tap = sh ? 2 : i;
    const int tp = t0 + sj + (tap - 1) * d;
    xv[i] = (tp >= 0 && tp < L && !(sh && i)) ? (unsigned)tp * 64u + (unsigned)so * 16u : 0x80000000u;   // outside the clip: zeros (WaveNet.py:26-27)
  }
  u32x4v xq[2];
  auto issue_x = [&](int ch) {
    xq[0] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(urs, xv[0], ch * L * 64, 0));
    if (!sh) xq[1] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(urs, xv[1], ch * L * 64, 0));
  };
  auto store_x = [&](__bf16 *dst) {
    __bf16 *row = dst + sj * UXS_ + 8 * so;
    if (sh) {
      *reinterpret_cast<u32x4v *>(row + 64) = xq[0];
    } else {
      *reinterpret_cast<u32x4v *>(row) = xq[0];
      *reinterpret_cast<u32x4v *>(row + 32) = xq[1];
    }
  };

  f32x16 acc[2][2];                                              // [tanh rows | sigmoid rows][column tile]; start from the conv's bias
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(b1 + rt * C + 32 * wave + 8 * q + 4 * hh);
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[rt][ct][4 * q + e] = bv[e];
    }
  auto load_a1 = [&](bf16x8(&a)[2], int step) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
      a[rt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1rs, lane16 + rt * UFR_, step * 2 * UFR_, 0));
  };
  bf16x8 a1[3][2];                                               // ring of three k-steps, requested two ahead (six per chunk: slot = ks % 3)
  load_a1(a1[0], 0);
  load_a1(a1[1], 1);
  issue_x(0);
  store_x(lds);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < 8; ch++) {
    const __bf16 *xb = lds + (ch & 1) * XB + j * UXS_ + 8 * hh;
    if (ch + 1 < 8) issue_x(ch + 1);
#pragma unroll
    for (int ks = 0; ks < 6; ks++) {
      const int nx = ch * 6 + ks + 2;
      load_a1(a1[(ks + 2) % 3], nx < 48 ? nx : 47);
      bf16x8 bq[2];
#pragma unroll
      for (int ct = 0; ct < 2; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(xb + 32 * ct * UXS_ + 16 * ks);
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks % 3][rt], bq[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch + 1 < 8) store_x(lds + ((ch + 1) & 1) * XB);
    __syncthreads();
  }

  // ---- GEMM2's first weight fragments and the residual's rows of u (chunk `wave` of the image: lane (j, hh) of column tile ct takes bytes
  // [32 s + 16 hh, + 16) of row t0 + 32 ct + j -- the values its accumulator registers 8 s .. 8 s + 7 belong to) go out under the gate
  bf16x8 a2[4];
  u32x4v pre[2][2];
  unsigned ro[2];
  if constexpr (!NOH) {
#pragma unroll
    for (int i = 0; i < 3; i++) a2[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16, i * UFR_, 0));
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const int t = t0 + 32 * ct + j;
      ro[ct] = t < L ? (unsigned)((wave * L + t) * 64 + hh * 16) : 0x80000000u;   // outside the clip: loads 0, store dropped
      pre[ct][0] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(urs, ro[ct], 0, 0));
      pre[ct][1] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(urs, ro[ct] + 32u, 0, 0));
    }
  }

  // ---- gate -> g image [column][channel] (bf16); rows rowoff(4 qq .. 4 qq + 3, hh) are channels 32 wave + 8 qq + 4 hh ..
#pragma unroll
  for (int ct = 0; ct < 2; ct++)
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      unsigned pk[2];
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 a2v = {acc[0][ct][4 * qq + e], acc[0][ct][4 * qq + e + 1]};
        const f32x2 b2v = {acc[1][ct][4 * qq + e], acc[1][ct][4 * qq + e + 1]};
        pk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(gate_pair_us(a2v, b2v), bf16x2));
      }
      *reinterpret_cast<uint2 *>(gim + (32 * ct + j) * UGS_ + 32 * wave + 8 * qq + 4 * hh) = make_uint2(pk[0], pk[1]);
    }
  __syncthreads();

  // ---- the g image leaves as whole 512-byte sample rows (the skip GEMM's operand): thread = (column tid >> 3, 64-byte part tid & 7)
  {
    const __amdgpu_buffer_rsrc_t grs = uni_rsrc(reinterpret_cast<const char *>(gout) + (size_t)b * L * 512u, (unsigned)L * 512u);
    const int col = tid >> 3, part = tid & 7;
    const int t = t0 + col;
    const unsigned off = t < L ? (unsigned)t * 512u + (unsigned)part * 64u : 0x80000000u;     // outside the clip: dropped
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4v *>(gim + col * UGS_ + 32 * part + 8 * i), grs, off + 16u * i, 0, 0);
  }
  if constexpr (NOH) {
    return;                                                      // the net's last layer: no image is written (WaveNet.py:131-135)
  } else {
    // ---- GEMM2, res_conv rows: accumulators start from b2
    f32x16 acr[2];
    f32x4 pn[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = 32 * wave + 8 * q + 4 * hh;
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(b2 + c);
      pn[q] = *reinterpret_cast<const f32x4 *>(ptn + c);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acr[0][4 * q + e] = bv[e];
        acr[1][4 * q + e] = bv[e];
      }
    }
    const __bf16 *gb = gim + j * UGS_ + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
      if (ks + 3 < 16) a2[(ks + 3) & 3] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16, (ks + 3) * UFR_, 0));
      bf16x8 bq[2];
#pragma unroll
      for (int ct = 0; ct < 2; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + 32 * ct * UGS_ + 16 * ks);
#pragma unroll
      for (int ct = 0; ct < 2; ct++) acr[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks & 3], bq[ct], acr[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- u' = bf16((u + res) sqrt(1/2) + part_t of the next layer)  (WaveNet.py:97, :84), straight from the accumulators: registers
    // 8 s .. 8 s + 7 of a lane are positions 16 s + 8 hh .. + 7 of its row -> two 16-byte stores per column tile
    const __amdgpu_buffer_rsrc_t uors = uni_rsrc(reinterpret_cast<char *>(uout) + (size_t)b * clip_bytes, clip_bytes);
    const float RS = 0.707106781186547524f;
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int s = 0; s < 2; s++) {
#pragma clang fp contract(off)                                  // the oracle's operation order: (u + acc) * rs, then + part_t, each rounded to fp32
        u32x4v o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned uw = pre[ct][s][e];
          const float u0 = __builtin_bit_cast(float, uw << 16), u1 = __builtin_bit_cast(float, uw & 0xffff0000u);
          const int r = 8 * s + 2 * e;
          const f32x4 pq = pn[r >> 2];
          const float pa = (r & 2) ? pq[2] : pq[0], pb = (r & 2) ? pq[3] : pq[1];
          const f32x2 v2 = {(u0 + acr[ct][r]) * RS + pa, (u1 + acr[ct][r + 1]) * RS + pb};
          o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
        }
        // (offset step in the VGPR, soffset = 0: see the note on 16-byte buffer stores in ap_resblock_bf16p.hip)
        __builtin_amdgcn_raw_buffer_store_b128(o, uors, ro[ct] + (unsigned)(32 * s), 0, 0);
      }
  }
}

#ifdef AP_TOOLS
static int g_no_bf16us = 0;     // 1: small launches stay on the persistent kernel; 2: every launch without factors on this one (A/B, bit identity)
#else
static constexpr int g_no_bf16us = 0;
#endif

// at most one 128-sample tile per CU: the launches whose duration is one tile's latency on the persistent kernel
bool resblock_bf16us_serves(const ap_ctx *ctx, int B, int L) {
  if (ctx->cfg.precision != AP_PREC_BF16_STORE || ctx->C != UC_ || ctx->S != UC_ || g_no_bf16us == 1) return false;
  if ((size_t)L * 512 >= ((size_t)1 << 31)) return false;
  return g_no_bf16us == 2 || (long long)B * ((L + 127) / 128) <= 256;
}

int launch_resblock_bf16us(ap_ctx *ctx, int layer, const void *uin, const float *pt_next, void *uout, void *gout, int B, int L, hipStream_t st) {
  const int C = UC_;
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + C) * C;
  const __bf16 *w1 = (const __bf16 *)ctx->w1p_bf + (size_t)layer * n1, *w2 = (const __bf16 *)ctx->w2p_bf + (size_t)layer * n2;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C, *b2 = ctx->b2 + (size_t)layer * 2 * C;
  const int nt = (L + 63) / 64;
  if ((long long)B * nt >= (1ll << 31)) { set_error("AP_PREC_BF16_STORE: too many tiles"); return -22; }
  if (uout)
    resblock_bf16us_kernel<false><<<(unsigned)(B * nt), 512, 0, st>>>(uin, uout, pt_next, gout, w1, w2, b1, b2, L, d, nt);
  else
    resblock_bf16us_kernel<true><<<(unsigned)(B * nt), 512, 0, st>>>(uin, nullptr, nullptr, gout, w1, w2, b1, b2, L, d, nt);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap

#ifdef AP_TOOLS
extern "C" int ap_debug_no_bf16us(int on) {
  ap::g_no_bf16us = on;
  return 0;
}
#endif
