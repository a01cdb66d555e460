// Input gradient of one Residual_block.forward (WaveNet.py:75-97) in AP_PREC_BF16: bf16 MFMA operands (RNE), fp32 accumulate -- the
// white-box attack's backward (robustness_eval/white_box_attack.py:392,437-439) at the bf16 forward's arithmetic, two launches per
// layer.  Nothing is kept by the forward pass except the layer inputs h (which it writes anyway): on the bf16 matrix pipe the dilated
// conv is cheap to recompute, a [B][2C][L] fp32 store per layer is not.
//
//   K1  resblock_bwd_gate_bf16_kernel:  y = DilConv_d(bf16(h + part_t)) + b1   (recomputed: the forward's GEMM1)
//                                       dg = [sqrt(1/2) W_res; W_skip]^T [dh'; dskip]
//                                       dy = gate'(y) . dg   -> bf16 image [clip][sample][2C] (1 KB per sample, channels contiguous)
//                                       (128-sample tiles, 8 waves; the y accumulators turn into packed fp16 gate factors before dg is accumulated)
//   K2  resblock_bwd_conv_bf16_kernel:  dh = sqrt(1/2) dh' + DilConv_d^T(dy)   (direct three-tap form: flops are not what bounds bf16)
// v_mfma_f32_32x32x16_bf16; B operands staged through LDS as [column][k] bf16 images with conflict-free 16-byte fragment reads, weights as
// bf16 A fragments (8 k per lane) streamed L2 -> registers through a ring three k-steps deep.  K2's staging is pure data movement: a
// (sample, tap) operand is a contiguous 256-byte run of the dy image.  Built for res = skip = 256 channels.
// Where the time goes at the white-box shape (B = 10: K1 0.30 ms, K2 0.19 ms per layer; MFMA pipes 26 % / 32 % busy, tools/time_bwd_bf16.py,
// docs/HISTORY.md H): every tile re-reads its weight fragments (1 MB / 0.8 MB) through the CU's vector-memory path beside its activations.
// Round 6: K1 has a second form that reads gate factors the forward kept instead of recomputing them (resblock_bwd_gate_fac_bf16_kernel,
// 0.13-0.14 ms per layer; the default of the differentiable purifier, docs/HISTORY.md I.5).
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));

constexpr int QC_ = 256;                  // res = skip channels
constexpr int XSB_ = 96 + 8;              // bf16 per column row of K1's X chunk image (3 taps x 32 channels; 208-byte rows)
constexpr int ZSB2_ = 128 + 8;            // bf16 per column row of K1's Z chunk image (128 rows of [dh'; dskip]; 272-byte rows)
constexpr int DSB_ = 2 * QC_ + 8;         // bf16 per column row of K1's dy tile (520: 1040-byte rows)
constexpr int YSB_ = 128 + 8;             // bf16 per column row of K2's chunk image (one tap x 128 channels; 272-byte rows)
constexpr unsigned FRB_ = 64 * 16;        // bytes of one row tile's fragment of a k-step (64 lanes x 8 bf16)

// Workgroup k runs on XCD k mod 8 (round-robin dispatch), and each XCD has its own L2: give every XCD one CONTIGUOUS run of tiles, in
// dispatch order, so that the tiles holding a tile's +-d taps (up to 32 tiles away) run on the same XCD at about the same time and a
// row of h / dy is fetched from HBM once, not once per tap.  Bijective for any grid size.
#ifdef AP_TOOLS
__device__ int g_bwdb_linear = 0;                                // tools/time_bwd_bf16.py: 1 = tile k on workgroup k (the A/B of this mapping)
#endif
__device__ __forceinline__ unsigned xcd_tile(unsigned k, unsigned G) {
#ifdef AP_TOOLS
  if (g_bwdb_linear) return k;
#endif
  const unsigned x = k & 7u, q = G >> 3, r = G & 7u;
  return x * q + (x < r ? x : r) + (k >> 3);
}

__device__ __forceinline__ bf16x8 cvt8(const float (&v)[8]) {
  const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return __builtin_convertvector(f, bf16x8);
}

}  // namespace

// ---- weight images (bf16, RNE) -----------------------------------------------------------------------------------------------
// K1, y:  the forward's GEMM1 image itself (ap_resblock_bf16p.hip, pack_w1_bf16_kernel: [wave 8][chunk 8][k-step 6][row tile 2][lane 64][8];
//         k-step ks of chunk ch: tap ks / 2, channel 32 ch + 16 (ks & 1) + 8 hh + jj; row tile rt: rt C + 32 wave + i) -- the recomputation
//         reads the very fragments the forward multiplied with.
// K1, dg: [wave 8][k-step 32][lane 64][8]; k = 16 ks + 8 hh + jj over [res output (256); skip output (256)];
//         row c = 32 wave + i; value W2[k][c], the res half times sqrt(1/2)
__global__ void pack_bw_w2t_kernel(const float *__restrict__ w2f, __bf16 *__restrict__ out) {
  constexpr int C = QC_;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 8u * 32 * 64 * 8) return;
  const int jj = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) & 31, w = idx >> 14;
  const int i = lane & 31, hh = lane >> 5;
  const int k = 16 * ks + 8 * hh + jj;
  const int c = 32 * w + i;
  const float v = w2f[(size_t)k * C + c];
  out[idx] = (__bf16)(k < C ? (float)((double)v * 0.70710678118654752440) : v);
}
// K2:     [wave 4][chunk 12][k-step 8][row tile 2][lane 64][8]; chunk = 4 tap' + block: pre-gate channel o = 128 block + 16 ks + 8 hh + jj;
//         row c = 64 wave + 32 rt + i; value W1[o][c][2 - tap']  (the transposed conv's taps are the forward's flipped)
__global__ void pack_bw_w1b_kernel(const float *__restrict__ w1f, __bf16 *__restrict__ out) {
  constexpr int C = QC_;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4u * 12 * 8 * 2 * 64 * 8) return;
  const int jj = idx & 7, lane = (idx >> 3) & 63, rt = (idx >> 9) & 1, ks = (idx >> 10) & 7;
  unsigned rest = idx >> 13;
  const int ch = rest % 12, w = rest / 12;
  const int i = lane & 31, hh = lane >> 5;
  const int tp = ch >> 2, o = 128 * (ch & 3) + 16 * ks + 8 * hh + jj;
  const int c = 64 * w + 32 * rt + i;
  out[idx] = (__bf16)w1f[((size_t)o * C + c) * 3 + (2 - tp)];
}

constexpr size_t BW_W2T_ = (size_t)8 * 32 * 64 * 8, BW_W1B_ = (size_t)4 * 12 * 8 * 2 * 64 * 8;

static int launch_pack_bwd_bf16(ap_ctx *ctx, hipStream_t st) {
  const size_t n1f = (size_t)2 * QC_ * QC_ * 3, n2 = (size_t)2 * QC_ * QC_;
  __bf16 *base = (__bf16 *)ctx->slab_bb;
  for (int n = 0; n < ctx->NL; n++) {
    __bf16 *p = base + (size_t)n * (BW_W2T_ + BW_W1B_);
    pack_bw_w2t_kernel<<<(unsigned)((BW_W2T_ + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, p);
    pack_bw_w1b_kernel<<<(unsigned)((BW_W1B_ + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1f, p + BW_W2T_);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// K1: one workgroup = one (clip, 128-sample tile); 8 waves (two per SIMD, <= 256 registers each), wave w = gate channels [32 w, 32 w + 32):
// 64 y rows, then 32 dg rows, x 128 columns.  The y accumulators do not live to the end: after GEMM 1 they become the gate's two
// derivative factors, packed as an fp16 pair per (channel, sample) (|factor| <= 1, 11 significant bits: four times finer than the bf16
// rounding dy gets anyway), which frees the registers GEMM 2 accumulates dg in.  Why 128 columns: a tile re-reads its weight fragments
// (1 MB) from L2, and the per-CU L2 delivery rate is what bounds the 64-column form.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void resblock_bwd_gate_bf16_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, const float *__restrict__ dh, const float *__restrict__ dskip,
    __bf16 *__restrict__ dy, const __bf16 *__restrict__ w1y, const float *__restrict__ b1, const __bf16 *__restrict__ w2t, int L, int d,
    int ntiles) {
  constexpr int C = QC_;
  constexpr int NT = 128;
  constexpr int XB = NT * XSB_, ZB = NT * ZSB2_;                   // bf16 elements per ring buffer
  __shared__ __attribute__((aligned(16))) __bf16 lds[NT * DSB_];   // 133 KB: the dy tile of the epilogue; the X ring, then the Z ring, alias its start
  static_assert(2 * XB <= NT * DSB_ && 2 * ZB <= NT * DSB_, "rings inside the dy tile");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int b = __builtin_amdgcn_readfirstlane((int)(tile / ntiles));
  const int t0 = __builtin_amdgcn_readfirstlane((int)(tile % ntiles) * NT);
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  const __amdgpu_buffer_rsrc_t hrs = uni_rsrc(hin + (size_t)b * C * L, clip_bytes);
  const float *dh_b = dh + (size_t)b * C * L, *ds_b = dskip + (size_t)b * C * L;
  const __amdgpu_buffer_rsrc_t w1rs = uni_rsrc(reinterpret_cast<const char *>(w1y) + (size_t)wave * (8 * 6 * 2 * FRB_), 8 * 6 * 2 * FRB_);
  const __amdgpu_buffer_rsrc_t w2rs = uni_rsrc(reinterpret_cast<const char *>(w2t) + (size_t)wave * (4 * 8 * FRB_), 4 * 8 * FRB_);
  const __amdgpu_buffer_rsrc_t ptrs = uni_rsrc(pt, C * 4u);
  const unsigned lane16 = (unsigned)lane * 16u;

  // ---- phase 1: y = DilConv(bf16(h + part_t)); chunk = 32 channels x 3 taps = 6 k-steps.  Staging thread = (column sj, channel octet sq)
  const int sj = tid & 127, sq = tid >> 7;
  unsigned xv[3];
#pragma unroll
  for (int tap = 0; tap < 3; tap++) {
    const int tp = t0 + sj + (tap - 1) * d;
    xv[tap] = (tp >= 0 && tp < L) ? ((unsigned)tp + (unsigned)(8 * sq) * (unsigned)L) * 4u : 0x80000000u;
  }
  float xr[3][8];
  f32x4 pq[2];
  auto issue_x = [&](int ch) {
    pq[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ptrs, (unsigned)(32 * sq), ch * 128, 0));
    pq[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ptrs, (unsigned)(32 * sq + 16), ch * 128, 0));
#pragma unroll
    for (int tap = 0; tap < 3; tap++)
#pragma unroll
      for (int i = 0; i < 8; i++)
        xr[tap][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, xv[tap], (32 * ch + i) * L * 4, 0));
  };
  auto store_x = [&](__bf16 *dst) {                              // u = h + part_t inside the clip, 0 outside (WaveNet.py:84, :26-27)
#pragma unroll
    for (int tap = 0; tap < 3; tap++) {
      const bool ok = xv[tap] != 0x80000000u;
      float u[8];
#pragma unroll
      for (int i = 0; i < 8; i++) u[i] = ok ? xr[tap][i] + pq[i >> 2][i & 3] : 0.f;
      *reinterpret_cast<bf16x8 *>(dst + sj * XSB_ + 32 * tap + 8 * sq) = cvt8(u);
    }
  };
  // phase 2's operand: chunk = 128 rows of [dh'; dskip] = 8 k-steps; staging thread = (column sj, 32 rows 32 sq ..)
  const int ts = t0 + sj;
  const unsigned zv = ts < L ? ((unsigned)ts + (unsigned)(32 * sq) * (unsigned)L) * 4u : 0x80000000u;
  float zr[32];
  auto issue_z = [&](int kc) {                                   // chunks 0, 1: dh' rows, 2, 3: dskip rows
    const __amdgpu_buffer_rsrc_t rs = uni_rsrc(kc < 2 ? dh_b : ds_b, clip_bytes);
#pragma unroll
    for (int i = 0; i < 32; i++) zr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, zv, ((kc & 1) * 128 + i) * L * 4, 0));
  };
  auto store_z = [&](__bf16 *dst) {
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = zr[8 * o + i];
      *reinterpret_cast<bf16x8 *>(dst + sj * ZSB2_ + 32 * sq + 8 * o) = cvt8(v);
    }
  };

  f32x16 accy[2][4];                                             // start from the conv's bias: rows rowoff(4 q .. 4 q + 3, hh) are channels 32 wave + 8 q + 4 hh ..
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(b1 + rt * C + 32 * wave + 8 * q + 4 * hh);
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int e = 0; e < 4; e++) accy[rt][ct][4 * q + e] = bv[e];
    }
  auto load_a1 = [&](bf16x8(&a)[2], int step) {                  // weight fragments of k-step `step` (0 .. 47), one step ahead of their MFMAs
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
      a[rt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1rs, lane16 + rt * FRB_, step * 2 * FRB_, 0));
  };
  // The weight fragments ride a ring of four k-steps, requested three steps (~1 500 cycles at two waves per SIMD) before their MFMAs: vmcnt
  // retires in order, so the first wait on a fragment requested AFTER a chunk's activation loads also waits for those -- the ring's depth is
  // the time the activation loads get before anything stalls on them (at depth one every k-step waited out an L2 round trip).
  bf16x8 a1[4][2];
  load_a1(a1[0], 0);
  load_a1(a1[1], 1);
  load_a1(a1[2], 2);
  issue_x(0);
  store_x(lds);
  __syncthreads();
  auto chunk1 = [&](auto P, int ch) {                            // P = ch & 1: the ring slot of k-step ks is (2 P + ks) & 3
    constexpr int pb = decltype(P)::value;
    const __bf16 *xb = lds + pb * XB + j * XSB_ + 8 * hh;
    if (ch + 1 < 8) issue_x(ch + 1);
#pragma unroll
    for (int ks = 0; ks < 6; ks++) {
      const int slot = (2 * pb + ks) & 3;
      const int nx = ch * 6 + ks + 3;
      load_a1(a1[(slot + 3) & 3], nx < 48 ? nx : 47);
      bf16x8 bq[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(xb + 32 * ct * XSB_ + 16 * ks);
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) accy[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[slot][rt], bq[ct], accy[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch + 1 < 8) store_x(lds + (pb ^ 1) * XB);
    __syncthreads();
  };
#pragma unroll 1
  for (int ch = 0; ch < 8; ch += 2) {
    chunk1(std::integral_constant<int, 0>{}, ch);
    chunk1(std::integral_constant<int, 1>{}, ch + 1);
  }
  issue_z(0);                                                    // in flight behind the gate arithmetic

  // ---- the gate's derivative factors: dy_tanh = dg . sg (1 - th^2), dy_sig = dg . th sg (1 - sg); packed (fp16, fp16)
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  f16x2 fac[4][16];
#pragma unroll
  for (int ct = 0; ct < 4; ct++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float at = fminf(fmaxf(accy[0][ct][r], -15.0f), 15.0f), as = fmaxf(accy[1][ct][r], -80.0f);
      const float E = exp_acc(2.0f * at), F = exp_acc(-as);
      const float R = __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
      const float th = (E - 1.0f) * (1.0f + F) * R, sg = (E + 1.0f) * R;
      const f32x2 f = {sg * (1.0f - th * th), th * sg * (1.0f - sg)};
      fac[ct][r] = __builtin_convertvector(f, f16x2);
    }

  // ---- phase 2: dg = W2^T [dh'; dskip]
  f32x16 accg[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++)
#pragma unroll
    for (int r = 0; r < 16; r++) accg[ct][r] = 0.f;
  auto load_a2 = [&](int step) {                                 // k-step `step` of 32
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16, step * FRB_, 0));
  };
  bf16x8 a2[4];
  a2[0] = load_a2(0);
  a2[1] = load_a2(1);
  a2[2] = load_a2(2);
  store_z(lds);
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc < 4; kc++) {
    const __bf16 *zb = lds + (kc & 1) * ZB + j * ZSB2_ + 8 * hh;
    if (kc + 1 < 4) issue_z(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {                               // (B fragments read in place: the SIMD's other wave covers the LDS latency,
      const int nx = kc * 8 + ks + 3;                              //  and a second set would not fit beside the packed factors)
      a2[(ks + 3) & 3] = load_a2(nx < 32 ? nx : 31);
      bf16x8 bq[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(zb + 32 * ct * ZSB2_ + 16 * ks);
#pragma unroll
      for (int ct = 0; ct < 4; ct++) accg[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks & 3], bq[ct], accg[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kc + 1 < 4) store_z(lds + ((kc + 1) & 1) * ZB);
    __syncthreads();
  }

  // ---- epilogue: dy = factor . dg into the tile image [column][2C] (bf16), then out as whole 1 KB sample rows
#pragma unroll
  for (int ct = 0; ct < 4; ct++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c0 = 32 * wave + 8 * q + 4 * hh;
      float vt[4], vs[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const f32x2 f = __builtin_convertvector(fac[ct][4 * q + e], f32x2);
        vt[e] = accg[ct][4 * q + e] * f[0];
        vs[e] = accg[ct][4 * q + e] * f[1];
      }
      const f32x4 ft = {vt[0], vt[1], vt[2], vt[3]}, fs = {vs[0], vs[1], vs[2], vs[3]};
      __bf16 *row = lds + (32 * ct + j) * DSB_;
      *reinterpret_cast<bf16x4 *>(row + c0) = __builtin_convertvector(ft, bf16x4);
      *reinterpret_cast<bf16x4 *>(row + C + c0) = __builtin_convertvector(fs, bf16x4);
    }
  __syncthreads();
  {
    const __amdgpu_buffer_rsrc_t ors = uni_rsrc(dy + (size_t)b * L * 2 * C, (unsigned)L * 2u * C * 2u);
    const int col = tid >> 2, part = tid & 3;                      // a column's 1 KB row leaves as 4 x 256 B
    const int t = t0 + col;
    const unsigned off = t < L ? (unsigned)t * 1024u + (unsigned)part * 256u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 16; i++)
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4b *>(lds + col * DSB_ + 128 * part + 8 * i), ors, off + 16u * i, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K1 from KEPT gate factors (round 6): the forward pass that runs for a gradient (ap_resblock_fwd_gate_save) writes the gate's two
// derivative factors as an fp16 pair per (channel, sample) in its accumulators' order (ap_resblock_bf16p.hip, SAVEF: 128 KB per
// 128-sample tile), so this launch neither recomputes the dilated conv (3/4 of K1's matrix work, its X staging) nor evaluates a
// transcendental: dg = [sqrt(1/2) W_res; W_skip]^T [dh'; dskip], dy = factor . dg.  Same tile / wave / lane / register geometry as the
// forward's gate phase: lane l of wave w reads back, per (column tile, q), the 16 bytes it wrote.
// ---------------------------------------------------------------------------------------------------------------------------
// DSI: dskip -- the same tensor for all 36 layers of a link -- arrives as a bf16 image [clip][sample][S] (ap_bwd_bf16_rows_image, once
// per link) instead of fp32 rows: its staging is four 16-byte loads and stores per thread and chunk, no convert, half the bytes.
template <bool DSI>
__global__ __launch_bounds__(512, 4) void resblock_bwd_gate_fac_bf16_kernel(const float *__restrict__ dh, const void *__restrict__ dskip,
                                                                            const void *__restrict__ fac, __bf16 *__restrict__ dy,
                                                                            const __bf16 *__restrict__ w2t, int L, int ntiles, int ntiles128) {
  // 64-sample tiles, two workgroups per CU (66 KB of LDS, <= 128 registers): the kernel streams -- dh', dskip, factors in, dy out -- and a
  // lone 128-sample workgroup per CU serialised its four load round trips and its epilogue with nothing to overlap them (0.17 ms per
  // layer at B = 10; this form: see DESIGN.md 3.6)
  constexpr int C = QC_;
  constexpr int NT = 64, CT = NT / 32, RPT = 128 * NT / 512;       // columns, column tiles, staged rows per thread and chunk (16)
  constexpr int ZB = NT * ZSB2_;
  __shared__ __attribute__((aligned(16))) __bf16 lds[NT * DSB_];   // 66.5 KB: the dy tile of the epilogue; the Z ring aliases its start
  static_assert(2 * ZB <= NT * DSB_, "ring inside the dy tile");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int b = __builtin_amdgcn_readfirstlane((int)(tile / ntiles));
  const int ti = __builtin_amdgcn_readfirstlane((int)(tile % ntiles));
  const int t0 = ti * NT;
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  const float *dh_b = dh + (size_t)b * C * L, *ds_b = DSI ? nullptr : static_cast<const float *>(dskip) + (size_t)b * C * L;
  const __amdgpu_buffer_rsrc_t dsirs = uni_rsrc(static_cast<const char *>(dskip) + (size_t)b * ((size_t)L * C * 2), (unsigned)L * C * 2u);   // DSI: [sample][S] bf16
  const __amdgpu_buffer_rsrc_t w2rs = uni_rsrc(reinterpret_cast<const char *>(w2t) + (size_t)wave * (4 * 8 * FRB_), 4 * 8 * FRB_);
  const __amdgpu_buffer_rsrc_t frs = uni_rsrc(reinterpret_cast<const char *>(fac) + (size_t)b * ((size_t)ntiles128 * 131072u), (unsigned)ntiles128 * 131072u);
  const unsigned lane16 = (unsigned)lane * 16u;

  auto load_a2 = [&](int step) {                                 // k-step `step` of 32
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16, step * FRB_, 0));
  };
  bf16x8 a2[4];
  a2[0] = load_a2(0);
  a2[1] = load_a2(1);
  a2[2] = load_a2(2);
  // operand: chunk = 128 rows of [dh'; dskip] = 8 k-steps; staging thread = (column sj, RPT rows RPT sq ..)
  const int sj = tid & (NT - 1), sq = tid / NT;
  const int ts = t0 + sj;
  const unsigned zv = ts < L ? ((unsigned)ts + (unsigned)(RPT * sq) * (unsigned)L) * 4u : 0x80000000u;
  float zr[RPT];
  u32x4b zq[RPT / 8];
  const unsigned ziv = ts < L ? (unsigned)ts * (unsigned)(C * 2) + (unsigned)(2 * RPT * sq) : 0x80000000u;   // DSI: this thread's RPT channels of the sample's row
  auto issue_z = [&](int kc) {                                   // chunks 0, 1: dh' rows, 2, 3: dskip rows
    if (DSI && kc >= 2) {
#pragma unroll
      for (int o = 0; o < RPT / 8; o++) zq[o] = __builtin_bit_cast(u32x4b, __builtin_amdgcn_raw_buffer_load_b128(dsirs, ziv + 16u * o, (kc & 1) * 256, 0));
      return;
    }
    const __amdgpu_buffer_rsrc_t rs = uni_rsrc(kc < 2 ? dh_b : ds_b, clip_bytes);
#pragma unroll
    for (int i = 0; i < RPT; i++) zr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, zv, ((kc & 1) * 128 + i) * L * 4, 0));
  };
  auto store_z = [&](__bf16 *dst, int kc) {
    if (DSI && kc >= 2) {
#pragma unroll
      for (int o = 0; o < RPT / 8; o++) *reinterpret_cast<u32x4b *>(dst + sj * ZSB2_ + RPT * sq + 8 * o) = zq[o];
      return;
    }
#pragma unroll
    for (int o = 0; o < RPT / 8; o++) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = zr[8 * o + i];
      *reinterpret_cast<bf16x8 *>(dst + sj * ZSB2_ + RPT * sq + 8 * o) = cvt8(v);
    }
  };
  issue_z(0);
  // the factors this lane wrote in the forward pass: [128-sample tile][wave][column tile][q][lane] x 16 bytes; in flight under the whole GEMM
  u32x4b fq[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ct++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int gct = (t0 >> 5) + ct;                            // column tile of the clip: 128-sample tile gct >> 2, its column tile gct & 3
      fq[ct][q] = __builtin_bit_cast(u32x4b, __builtin_amdgcn_raw_buffer_load_b128(frs, lane16, (((gct >> 2) * 8 + wave) * 16 + (gct & 3) * 4 + q) * 1024, 2));
    }

  f32x16 accg[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ct++)
#pragma unroll
    for (int r = 0; r < 16; r++) accg[ct][r] = 0.f;
  store_z(lds, 0);
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc < 4; kc++) {
    const __bf16 *zb = lds + (kc & 1) * ZB + j * ZSB2_ + 8 * hh;
    if (kc + 1 < 4) issue_z(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const int nx = kc * 8 + ks + 3;
      a2[(ks + 3) & 3] = load_a2(nx < 32 ? nx : 31);
      bf16x8 bq[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(zb + 32 * ct * ZSB2_ + 16 * ks);
#pragma unroll
      for (int ct = 0; ct < CT; ct++) accg[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks & 3], bq[ct], accg[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kc + 1 < 4) store_z(lds + ((kc + 1) & 1) * ZB, kc + 1);
    __syncthreads();
  }

  // ---- epilogue: dy = factor . dg into the tile image [column][2C] (bf16), then out as whole 1 KB sample rows
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int ct = 0; ct < CT; ct++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c0 = 32 * wave + 8 * q + 4 * hh;
      // (the WHOLE 16-byte vector is re-typed, then indexed: an element-wise bit_cast of the buffer-load builtin's vector is mis-folded to a splat)
      const f32x8 f = __builtin_convertvector(__builtin_bit_cast(f16x8, fq[ct][q]), f32x8);
      float vt[4], vs[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        vt[e] = accg[ct][4 * q + e] * f[2 * e];
        vs[e] = accg[ct][4 * q + e] * f[2 * e + 1];
      }
      const f32x4 ft = {vt[0], vt[1], vt[2], vt[3]}, fs = {vs[0], vs[1], vs[2], vs[3]};
      __bf16 *row = lds + (32 * ct + j) * DSB_;
      *reinterpret_cast<bf16x4 *>(row + c0) = __builtin_convertvector(ft, bf16x4);
      *reinterpret_cast<bf16x4 *>(row + C + c0) = __builtin_convertvector(fs, bf16x4);
    }
  __syncthreads();
  {
    const __amdgpu_buffer_rsrc_t ors = uni_rsrc(dy + (size_t)b * L * 2 * C, (unsigned)L * 2u * C * 2u);
    constexpr int PARTS = 512 / NT, PB = 1024 / PARTS;           // a column's 1 KB row leaves as PARTS x PB bytes
    const int col = tid / PARTS, part = tid % PARTS;
    const int t = t0 + col;
    const unsigned off = t < L ? (unsigned)t * 1024u + (unsigned)(part * PB) : 0x80000000u;
#pragma unroll
    for (int i = 0; i < PB / 16; i++)
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4b *>(lds + col * DSB_ + (PB / 2) * part + 8 * i), ors, off + 16u * i, 0, 0);
  }
}

#ifdef AP_TOOLS
__device__ unsigned long long *g_bwdb_stamp = nullptr;           // tools/clock_bwd_bf16.py: [workgroup][2] = (s_memtime, s_memrealtime) ticks around K2's chunk loop
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// K2: one workgroup = one (clip, 64-sample tile); 4 waves x (64 rows x 64 columns), 132 registers: three workgroups per CU.
// (Measured and not kept, docs/HISTORY.md H: 128-column tiles at two workgroups per CU and 64-column tiles at four per CU -- both 3-7 % slower.)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void resblock_bwd_conv_bf16_kernel(const __bf16 *__restrict__ dy, const float *__restrict__ dhp,
                                                                        float *__restrict__ dhin, const __bf16 *__restrict__ w1b,
                                                                        int L, int d, int ntiles) {
  constexpr int C = QC_;
  constexpr int CT = 2, NT = 32 * CT;                             // column tiles of 32, columns of a tile
  constexpr int YB = NT * YSB_;
  constexpr int PPT = 2 * CT, TPC = 16 / PPT;                     // 16-byte pieces per staging thread, threads per column
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * YB];     // 34.8 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int b = __builtin_amdgcn_readfirstlane((int)(tile / ntiles));
  const int t0 = __builtin_amdgcn_readfirstlane((int)(tile % ntiles) * NT);
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t yrs = uni_rsrc(dy + (size_t)b * L * 2 * C, (unsigned)L * 2u * C * 2u);
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc(reinterpret_cast<const char *>(w1b) + (size_t)wave * (12 * 8 * 2 * FRB_), 12 * 8 * 2 * FRB_);
  const unsigned lane16 = (unsigned)lane * 16u;
  // staging: thread = (column tid / TPC, part tid % TPC) of a chunk's 256-byte (column, tap) run: PPT x 16 B
  const int scol = tid / TPC, spart = tid % TPC;
  unsigned yv[3];
#pragma unroll
  for (int tp = 0; tp < 3; tp++) {
    const int t = t0 + scol + (tp - 1) * d;
    yv[tp] = (t >= 0 && t < L) ? (unsigned)t * 1024u + (unsigned)spart * (16u * PPT) : 0x80000000u;    // outside the clip: zeros (WaveNet.py:26-27)
  }
  u32x4b yq[PPT];
  auto issue_y = [&](int ch) {
    const unsigned v = (ch >> 2) == 0 ? yv[0] : (ch >> 2) == 1 ? yv[1] : yv[2];
#pragma unroll
    for (int i = 0; i < PPT; i++) yq[i] = __builtin_bit_cast(u32x4b, __builtin_amdgcn_raw_buffer_load_b128(yrs, v, (ch & 3) * 256 + 16 * i, 0));
  };
  auto store_y = [&](__bf16 *dst) {
#pragma unroll
    for (int i = 0; i < PPT; i++) *reinterpret_cast<u32x4b *>(dst + scol * YSB_ + 8 * PPT * spart + 8 * i) = yq[i];
  };
  f32x16 acc[2][CT];
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[rt][ct][r] = 0.f;
  auto load_a = [&](bf16x8(&a)[2], int step) {                   // k-step `step` of 96
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
      a[rt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + rt * FRB_, step * 2 * FRB_, 0));
  };
  bf16x8 aw[4][2];                                               // ring of four k-steps, requested three ahead (see K1)
  load_a(aw[0], 0);
  load_a(aw[1], 1);
  load_a(aw[2], 2);
  issue_y(0);
  store_y(lds);
  __syncthreads();
#ifdef AP_TOOLS
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
  for (int ch = 0; ch < 12; ch++) {
    const __bf16 *yb = lds + (ch & 1) * YB + j * YSB_ + 8 * hh;
    issue_y(ch + 1 < 12 ? ch + 1 : ch);
    constexpr int DB = 1;
    bf16x8 bq[DB + 1][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) bq[0][ct] = *reinterpret_cast<const bf16x8 *>(yb + 32 * ct * YSB_);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const int nx = ch * 8 + ks + 3;
      load_a(aw[(ks + 3) & 3], nx < 96 ? nx : 95);
      if (DB ? ks + 1 < 8 : ks > 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
          bq[(ks + 1) & DB][ct] = *reinterpret_cast<const bf16x8 *>(yb + 32 * ct * YSB_ + 16 * (ks + DB));
      }
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[ks & 3][rt], bq[ks & DB][ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_y(lds + ((ch + 1) & 1) * YB);
    __syncthreads();
  }
#ifdef AP_TOOLS
  if (g_bwdb_stamp && tid == 0) {                                // the in-kernel clock of the loop: d(s_memtime) / d(s_memrealtime) x 100 MHz
    g_bwdb_stamp[2 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memtime() - st_t0;
    g_bwdb_stamp[2 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
  // epilogue: the residual path, fp32 rows out
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  const __amdgpu_buffer_rsrc_t prs = uni_rsrc(dhp + (size_t)b * C * L, clip_bytes);
  const __amdgpu_buffer_rsrc_t ors = uni_rsrc(dhin + (size_t)b * C * L, clip_bytes);
  const float RS = 0.707106781186547524f;
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
      const int t = t0 + 32 * ct + j;
      const unsigned eo = t < L ? ((unsigned)(64 * wave + 32 * rt + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; r++) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 0));
#pragma unroll
      for (int r = 0; r < 16; r++)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_fmaf(RS, rv[r], acc[rt][ct][r])), ors, eo,
                                              ((r & 3) + 8 * (r >> 2)) * L * 4, 0);
    }
}

bool resblock_bwd_bf16_serves(const ap_ctx *ctx, int B, int L) {
  if ((ctx->cfg.precision != AP_PREC_BF16 && ctx->cfg.precision != AP_PREC_BF16_STORE) || ctx->C != QC_ || ctx->S != QC_ || !ctx->loaded) return false;
  if ((size_t)2 * QC_ * (size_t)L * 4 >= ((size_t)1 << 31)) return false;
  return (long long)B * ((L + 63) / 64) < (1ll << 31);
}

int launch_resblock_bwd_bf16(ap_ctx *ctx, int layer, const float *hin, const float *pt, const float *dhp, const float *dskip, void *dy,
                             float *dhin, int B, int L, hipStream_t st) {
  if (!resblock_bwd_bf16_serves(ctx, B, L)) {
    set_error("ap_resblock_bwd_bf16: built for AP_PREC_BF16 with res = skip = 256 channels and clips below 2^20 samples");
    return -22;
  }
  if (!ctx->bwd_ready) {                                         // (launch functions allocate nothing: include/audiopure.h)
    set_error("ap_resblock_bwd_bf16: the backward weight images are not built (ap_ctx_prepare_backward after every ap_ctx_load_wavenet)");
    return -22;
  }
  const __bf16 *p = (const __bf16 *)ctx->slab_bb + (size_t)layer * (BW_W2T_ + BW_W1B_);
  const __bf16 *w1y = (const __bf16 *)ctx->w1p_bf + (size_t)layer * ((size_t)2 * QC_ * QC_ * 3);   // the forward's GEMM1 image
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int nt = (L + 63) / 64, nt4 = (L + 127) / 128;
  resblock_bwd_gate_bf16_kernel<<<(unsigned)(B * nt4), 512, 0, st>>>(hin, pt, dhp, dskip, (__bf16 *)dy, w1y, ctx->b1 + (size_t)layer * 2 * QC_,
                                                                     p, L, d, nt4);
  resblock_bwd_conv_bf16_kernel<<<(unsigned)(B * nt), 256, 0, st>>>((const __bf16 *)dy, dhp, dhin, p + BW_W2T_, L, d, nt);
  AP_HIP(hipGetLastError());
  return 0;
}

// rows [B][C][L] fp32 -> image [B][L][C] bf16 (RNE), C = 256: a 64-sample x 256-channel tile per workgroup through LDS
__global__ __launch_bounds__(256) void rows_image_bf16_kernel(const float *__restrict__ x, __bf16 *__restrict__ img, int L, int ntiles) {
  constexpr int C = QC_;
  __shared__ float tile[64][C + 1];
  const int b = blockIdx.x / ntiles, t0 = (blockIdx.x % ntiles) * 64;
  const int tid = threadIdx.x;
  const float *xb = x + (size_t)b * C * L;
#pragma unroll 4
  for (int i = 0; i < 64; i++) {                                 // thread = (sample tid & 63, channel i * 4 + (tid >> 6)): coalesced along the samples
    const int c = 4 * i + (tid >> 6), t = t0 + (tid & 63);
    tile[tid & 63][c] = t < L ? xb[(size_t)c * L + t] : 0.f;
  }
  __syncthreads();
  __bf16 *ib = img + (size_t)b * L * C;
#pragma unroll 4
  for (int i = 0; i < 64; i++) {                                 // thread = channel tid of sample i: a 512-byte row per step
    const int t = t0 + i;
    if (t < L) ib[(size_t)t * C + tid] = (__bf16)tile[i][tid];
  }
}

int launch_rows_image_bf16(const float *x, void *img, int B, int L, hipStream_t st) {
  const int nt = (L + 63) / 64;
  rows_image_bf16_kernel<<<(unsigned)(B * nt), 256, 0, st>>>(x, (__bf16 *)img, L, nt);
  AP_HIP(hipGetLastError());
  return 0;
}

// the same gradient from the gate factors the forward pass kept (ap_resblock_fwd_gate_save): no recomputation of the dilated conv.
// dskip_is_image: dskip is the bf16 image [B][L][S] of ap_bwd_bf16_rows_image instead of fp32 rows [B][S][L]
int launch_resblock_bwd_bf16_saved(ap_ctx *ctx, int layer, const void *fac, const float *dhp, const void *dskip, int dskip_is_image, void *dy,
                                   float *dhin, int B, int L, hipStream_t st) {
  if (!resblock_bwd_bf16_serves(ctx, B, L)) {
    set_error("ap_resblock_bwd_bf16_saved: built for AP_PREC_BF16 with res = skip = 256 channels and clips below 2^20 samples");
    return -22;
  }
  if (!ctx->bwd_ready) {
    set_error("ap_resblock_bwd_bf16_saved: the backward weight images are not built (ap_ctx_prepare_backward after every ap_ctx_load_wavenet)");
    return -22;
  }
  const __bf16 *p = (const __bf16 *)ctx->slab_bb + (size_t)layer * (BW_W2T_ + BW_W1B_);
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int nt = (L + 63) / 64, nt4 = (L + 127) / 128;
  if (dskip_is_image) {
    if ((size_t)L * QC_ * 2 >= ((size_t)1 << 31)) { set_error("ap_resblock_bwd_bf16_saved: clip too long for the bf16 dskip image"); return -22; }
    resblock_bwd_gate_fac_bf16_kernel<true><<<(unsigned)(B * nt), 512, 0, st>>>(dhp, dskip, fac, (__bf16 *)dy, p, L, nt, nt4);
  } else {
    resblock_bwd_gate_fac_bf16_kernel<false><<<(unsigned)(B * nt), 512, 0, st>>>(dhp, dskip, fac, (__bf16 *)dy, p, L, nt, nt4);
  }
  resblock_bwd_conv_bf16_kernel<<<(unsigned)(B * nt), 256, 0, st>>>((const __bf16 *)dy, dhp, dhin, p + BW_W2T_, L, d, nt);
  AP_HIP(hipGetLastError());
  return 0;
}

// This context's two bf16 backward weight images (47 MB): allocation + pack + a host synchronisation, published when complete.
int prepare_bwd_bf16(ap_ctx *ctx, hipStream_t st) {
  if (ctx->bwd_ready) return 0;
  if (!ctx->slab_bb) AP_HIP(hipMalloc(&ctx->slab_bb, (size_t)ctx->NL * (BW_W2T_ + BW_W1B_) * 2));
  int rc = launch_pack_bwd_bf16(ctx, st);
  if (rc == 0) {
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize(prepare_backward)");
  }
  if (rc) {
    (void)hipFree(ctx->slab_bb);
    ctx->slab_bb = nullptr;
    return rc;
  }
  ctx->bwd_ready = true;
  return 0;
}

}  // namespace ap

extern "C" int ap_resblock_bwd_bf16(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, const float *dh_out,
                                    const float *dskip, void *dy_scratch, float *dh_in, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !h_in || !part_t_layer || !dh_out || !dskip || !dy_scratch || !dh_in) { ap::set_error("ap_resblock_bwd_bf16: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { ap::set_error("ap_resblock_bwd_bf16: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (dh_in == dh_out) { ap::set_error("ap_resblock_bwd_bf16: dh_in must not alias dh_out"); return -22; }
  return ap::launch_resblock_bwd_bf16(ctx, layer, h_in, part_t_layer, dh_out, dskip, dy_scratch, dh_in, B, L, (hipStream_t)stream);
}

extern "C" int ap_bwd_bf16_rows_image(const float *rows, void *image, int B, int C, int L, void *stream) {
  if (!rows || !image || B < 1 || L < 1) { ap::set_error("ap_bwd_bf16_rows_image: bad argument"); return -22; }
  if (C != 256) { ap::set_error("ap_bwd_bf16_rows_image: built for 256 channels (got %d)", C); return -22; }
  return ap::launch_rows_image_bf16(rows, image, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_bwd_bf16_saved(ap_ctx *ctx, int layer, const void *gate_factors, const float *dh_out, const void *dskip,
                                          int dskip_is_image, void *dy_scratch, float *dh_in, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !gate_factors || !dh_out || !dskip || !dy_scratch || !dh_in) { ap::set_error("ap_resblock_bwd_bf16_saved: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { ap::set_error("ap_resblock_bwd_bf16_saved: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (dh_in == dh_out) { ap::set_error("ap_resblock_bwd_bf16_saved: dh_in must not alias dh_out"); return -22; }
  return ap::launch_resblock_bwd_bf16_saved(ctx, layer, gate_factors, dh_out, dskip, dskip_is_image, dy_scratch, dh_in, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_bwd_bf16_available(ap_ctx *ctx, int B, int L) { return ctx && ap::resblock_bwd_bf16_serves(ctx, B, L) ? 1 : 0; }

#ifdef AP_TOOLS
extern "C" int ap_debug_bwdb_linear(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(ap::g_bwdb_linear), &v, sizeof(v)); }
extern "C" int ap_debug_bwdb_stamp(void *buf) {                  // buf: device memory, 16 bytes per K2 workgroup (or null: off)
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ap::g_bwdb_stamp), &buf, sizeof(buf));
}
#endif
