// AP_PREC_BF16, small batches: the deferred-skip form of the fused Residual_block.forward (WaveNet.py:75-97; ap_resblock_fwd_gate:
// h' and the bf16 gate image, no skip_conv) with ONE 64-sample tile per workgroup, for launches that have at most one 128-sample tile
// per CU.  There the persistent kernel (ap_resblock_bf16p.hip) gives every CU one tile and the launch lasts one tile's latency
// (50 us at B = 1, 57 at B = 2: tools/trace_small_batch.py); half-size tiles on twice as many workgroups halve the work per CU, and a
// short straight-line kernel has less latency to expose than a persistent one built to overlap consecutive tiles.
// Results are BIT-IDENTICAL to the persistent kernel's (a clip's result must not depend on the batch it travels in):
//   * GEMM1: accumulators start from b1, the same bf16 operands (u = h + part_t in fp32, RNE; the same packed weight image) enter the
//     same v_mfma_f32_32x32x16_bf16 sequence per output element -- chunk by chunk, six k-steps each (the matrix pipe's fp32
//     accumulation depends on the order of the k-steps only, not on the shape of the tile);
//   * the gate: the same gate_fast2 on the same channel pairs, bf16 RNE;
//   * GEMM2 (res_conv): accumulators start from b2 + part_t (fp32), sixteen k-steps in order, h' = (h + acc) sqrt(1/2).
// tests: the deferred-skip pair against the fused block bit for bit (tools/fuzz_blocks.py, tests/test_gpu_fuzz.py), small batches against
// the 512-clip batch (tests/test_gpu_parity.py).  Built for res = skip = 256 channels.
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

constexpr int SC_ = 256;                  // res = skip channels
constexpr int SXS_ = 96 + 8;              // bf16 per column row of the X chunk image (3 taps x 32 channels; 208-byte rows)
constexpr int SGS_ = 256 + 8;             // bf16 per column row of the g image (528-byte rows)
constexpr unsigned SFR_ = 64 * 16;        // bytes of one row tile's fragment of a k-step

// tanh(a) sigmoid(b): the arithmetic of ap_resblock_bf16p.hip's gate_fast2, operation for operation (the results must be its results)
__device__ __forceinline__ f32x2 gate_pair(f32x2 a, f32x2 b) {
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

// (Measured and not kept: the same straight-line kernel on 128-sample tiles, one workgroup per CU -- 34.7 us at B = 1, 43.8 at B = 2 against
// 25.2 / 41.7 here; from three clips on the persistent kernel is as fast or faster: 61 against 61 us at B = 3, 70 against 83 at B = 4.)
// (Also measured and not kept: 64-channel chunks -- four load round trips per tile instead of eight -- at one workgroup per CU: 26.0 against
// 25.3 us at B = 1.  The tile is bound by the 1.25 MB it moves through its CU's vector-memory path, 0.9 MB of it weight fragments.)
// (And: two adjacent tiles per 16-wave workgroup, so that the two column halves' waves request the same weight fragments a barrier
// interval apart and could share them in L1: 43.7 against 42.0 us at B = 2 -- what two tiles on one CU share is its delivery rate.)
// One workgroup = one (clip, 64-sample tile); 8 waves, wave w = gate channels [32 w, 32 w + 32) in GEMM1 and res rows [32 w, 32 w + 32)
// in GEMM2.  w1 / w2: this layer's images of ap_resblock_bf16p.hip (pack_w1_bf16_kernel: [wave][chunk 8][k-step 6][row tile 2][lane][8];
// pack_w2_bf16_kernel: [wave][row tile 2][k-step 16][lane][8], row tile 0 = res rows).
template <bool NOH>
__global__ __launch_bounds__(512, 4) void resblock_bf16s_kernel(const float *__restrict__ hin, const float *__restrict__ pt,
                                                                float *__restrict__ hout, void *__restrict__ gout,
                                                                const __bf16 *__restrict__ w1, const __bf16 *__restrict__ w2,
                                                                const float *__restrict__ b1, const float *__restrict__ b2, int L, int d,
                                                                int ntiles) {
  constexpr int C = SC_;
  constexpr int NT = 64;
  constexpr int XB = NT * SXS_;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * XB + NT * SGS_];   // 26.6 KB X ring + 33.8 KB g image
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
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  const __amdgpu_buffer_rsrc_t hrs = uni_rsrc(hin + (size_t)b * C * L, clip_bytes);
  const __amdgpu_buffer_rsrc_t w1rs = uni_rsrc(reinterpret_cast<const char *>(w1) + (size_t)wave * (8 * 6 * 2 * SFR_), 8 * 6 * 2 * SFR_);
  const __amdgpu_buffer_rsrc_t w2rs = uni_rsrc(reinterpret_cast<const char *>(w2) + (size_t)wave * (2 * 16 * SFR_), 16 * SFR_);   // row tile 0 only
  const __amdgpu_buffer_rsrc_t ptrs = uni_rsrc(pt, C * 4u);
  const unsigned lane16 = (unsigned)lane * 16u;

  // ---- GEMM1 staging: thread = (column sj, channel quad sq of the chunk's 32 channels): 4 channels x 3 taps
  const int sj = tid & 63, sq = tid >> 6;
  unsigned xv[3];
#pragma unroll
  for (int tap = 0; tap < 3; tap++) {
    const int tp = t0 + sj + (tap - 1) * d;
    xv[tap] = (tp >= 0 && tp < L) ? ((unsigned)tp + (unsigned)(4 * sq) * (unsigned)L) * 4u : 0x80000000u;
  }
  float xr[3][4];
  f32x4 pq;
  auto issue_x = [&](int ch) {
    pq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ptrs, (unsigned)(16 * sq), ch * 128, 0));
#pragma unroll
    for (int tap = 0; tap < 3; tap++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        xr[tap][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, xv[tap], (32 * ch + i) * L * 4, 0));
  };
  auto store_x = [&](__bf16 *dst) {                              // u = h + part_t inside the clip, 0 outside (WaveNet.py:84, :26-27)
#pragma unroll
    for (int tap = 0; tap < 3; tap++) {
      const bool ok = xv[tap] != 0x80000000u;
      const f32x4 u = {ok ? xr[tap][0] + pq[0] : 0.f, ok ? xr[tap][1] + pq[1] : 0.f, ok ? xr[tap][2] + pq[2] : 0.f, ok ? xr[tap][3] + pq[3] : 0.f};
      *reinterpret_cast<bf16x4 *>(dst + sj * SXS_ + 32 * tap + 4 * sq) = __builtin_convertvector(u, bf16x4);
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
      a[rt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1rs, lane16 + rt * SFR_, step * 2 * SFR_, 0));
  };
  bf16x8 a1[3][2];                                               // ring of three k-steps, requested two ahead (six per chunk: slot = ks % 3)
  load_a1(a1[0], 0);
  load_a1(a1[1], 1);
  issue_x(0);
  store_x(lds);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < 8; ch++) {
    const __bf16 *xb = lds + (ch & 1) * XB + j * SXS_ + 8 * hh;
    if (ch + 1 < 8) issue_x(ch + 1);
#pragma unroll
    for (int ks = 0; ks < 6; ks++) {
      const int nx = ch * 6 + ks + 2;
      load_a1(a1[(ks + 2) % 3], nx < 48 ? nx : 47);
      bf16x8 bq[2];
#pragma unroll
      for (int ct = 0; ct < 2; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(xb + 32 * ct * SXS_ + 16 * ks);
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks % 3][rt], bq[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch + 1 < 8) store_x(lds + ((ch + 1) & 1) * XB);
    __syncthreads();
  }

  // ---- GEMM2's first weight fragments go out under the gate
  const __amdgpu_buffer_rsrc_t ors = uni_rsrc(NOH ? (const float *)hin : (const float *)hout + (size_t)b * C * L, clip_bytes);
  bf16x8 a2[4];
  float hres[2][16];
  unsigned eo[2];
  if constexpr (!NOH) {
#pragma unroll
    for (int i = 0; i < 3; i++) a2[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16, i * SFR_, 0));
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
        const f32x2 g2 = gate_pair(a2v, b2v);
        pk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(g2, bf16x2));
      }
      *reinterpret_cast<uint2 *>(gim + (32 * ct + j) * SGS_ + 32 * wave + 8 * qq + 4 * hh) = make_uint2(pk[0], pk[1]);
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
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4s *>(gim + col * SGS_ + 32 * part + 8 * i), grs, off + 16u * i, 0, 0);
  }
  if constexpr (NOH) return;                                     // the net's last layer: its h' is never read (WaveNet.py:131-135)
  // the residual's h values in accumulator layout (the staging pass has just pulled these rows into L2), in flight under GEMM2
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    const int t = t0 + 32 * ct + j;
    eo[ct] = t < L ? ((unsigned)(32 * wave + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
#pragma unroll
    for (int r = 0; r < 16; r++)
      hres[ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, eo[ct], ((r & 3) + 8 * (r >> 2)) * L * 4, 0));
  }

  // ---- GEMM2, res_conv rows: accumulators start from b2 + part_t (u = h + part_t re-enters the residual: alias semantics, WaveNet.py:77-84)
  f32x16 acr[2];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int c = 32 * wave + 8 * q + 4 * hh;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(b2 + c), pv = *reinterpret_cast<const f32x4 *>(pt + c);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float v = bv[e] + pv[e];
      acr[0][4 * q + e] = v;
      acr[1][4 * q + e] = v;
    }
  }
  const __bf16 *gb = gim + j * SGS_ + 8 * hh;
#pragma unroll
  for (int ks = 0; ks < 16; ks++) {
    if (ks + 3 < 16) a2[(ks + 3) & 3] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16, (ks + 3) * SFR_, 0));
    bf16x8 bq[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + 32 * ct * SGS_ + 16 * ks);
#pragma unroll
    for (int ct = 0; ct < 2; ct++) acr[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks & 3], bq[ct], acr[ct], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- h' = (h + res) sqrt(1/2)  (WaveNet.py:97), rows out in accumulator layout (128-byte runs per half wave)
  const float RS = 0.707106781186547524f;
#pragma unroll
  for (int ct = 0; ct < 2; ct++)
#pragma unroll
    for (int r = 0; r < 16; r++)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (hres[ct][r] + acr[ct][r]) * RS), ors, eo[ct],
                                            ((r & 3) + 8 * (r >> 2)) * L * 4, 0);
}

// at most one 128-sample tile per CU: the launches whose duration is one tile's latency on the persistent kernel
bool resblock_bf16s_serves(const ap_ctx *ctx, int B, int L) {
  if (ctx->cfg.precision != AP_PREC_BF16 || ctx->C != SC_ || ctx->S != SC_) return false;
  if ((size_t)L * 1024 >= ((size_t)1 << 31)) return false;
  return (long long)B * ((L + 127) / 128) <= 256;
}

int launch_resblock_bf16s(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, void *gout, int B, int L, hipStream_t st) {
  const int C = SC_;
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + C) * C;
  const __bf16 *w1 = (const __bf16 *)ctx->w1p_bf + (size_t)layer * n1, *w2 = (const __bf16 *)ctx->w2p_bf + (size_t)layer * n2;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C, *b2 = ctx->b2 + (size_t)layer * 2 * C;
  const int nt = (L + 63) / 64;
  if (hout)
    resblock_bf16s_kernel<false><<<(unsigned)(B * nt), 512, 0, st>>>(hin, pt, hout, gout, w1, w2, b1, b2, L, d, nt);
  else
    resblock_bf16s_kernel<true><<<(unsigned)(B * nt), 512, 0, st>>>(hin, pt, nullptr, gout, w1, w2, b1, b2, L, d, nt);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
