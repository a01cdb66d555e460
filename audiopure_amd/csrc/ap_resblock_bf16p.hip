// AP_PREC_BF16, persistent form: fused Residual_block.forward (WaveNet.py:75-97) with bf16 MFMA operands, fp32 accumulate,
// fp32 activations in HBM -- the same arithmetic and the same packed weight images as tools/csrc/ap_resblock_bf16.hip (outputs are
// bit-identical: tools/cmp_bf16_kernels.py), every dilation (d <= 32 stages one 192-column window per chunk for the three taps), restructured
// around what the round-2 ablations measured (tools/dbg_resblock_bf16.py):
//
//   * with GEMM1 emptied the old kernel still took 52 % of its time, and 60 % of THAT was the residual / skip
//     read-modify-write traffic (262 KB of loads + 262 KB of stores per 128-sample tile) issued in two bursts at the very
//     end of a workgroup's life: every CU alternated between a phase that used no memory bandwidth and a phase that used
//     nothing else, then paid a store drain (s_endpgm waits for vmcnt = 0), a workgroup launch and a cold first load.
//
// Here one workgroup per CU walks its tiles in a loop (XCD-local order), and the memory traffic of a tile is spread over
// the tile and into the next one:
//   - the h patch the residual needs is requested before and during the gate, the running skip rows during the gate and the
//     first epilogue (the GEMM1 accumulators free up as they are gated);
//   - the first X chunk of the NEXT tile is requested before GEMM2's second pass and packed -- with that tile's first
//     weight fragments and second chunk requested -- before the last epilogue's stores, so the stores of a tile drain
//     behind the next tile's GEMM1 (vmcnt retires in order: a load issued after a store cannot be waited for without
//     waiting for the store);
//   - the next chunk's FiLM add / bf16 pack / ds_write sit between the MFMAs of a chunk's fourth k-step (two X buffers
//     instead of three), so the chunk has no VALU-only tail;
//   - no register is spilled: a scratch reload inside the tile loop is a vector-memory load whose wait (vmcnt(0)) drains
//     every request in flight.  Lane geometry that is needed once per chunk / tile is re-derived from the lane id there
//     instead of being kept (x_geom, pack_ptv), the GEMM1 fragments are a ring of three k-steps (24 registers).
// Layout notes: X image [column][k] bf16, 208-B rows (conflict-free ds_read_b128 B fragments); g image [column][channel],
// 528-B rows; wave-private 32 x 32 fp32 output patch with 128-B rows (with the 16-lane groups of ds_read_b128 a padded
// 144-B row was 2-way conflicted, the unpadded one is conflict-free for both the column writes and the row reads).
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PT_ = 128;                 // time tile
constexpr int KC_ = 32;                  // channels per staged chunk -> 96 K rows = 6 k-steps of 16
constexpr int XS_ = 3 * KC_ + 8;         // bf16 per column row of the X image (208 B)
constexpr int GS_ = 256 + 8;             // bf16 per column row of the g image (528 B)
constexpr int PS_ = 32;                  // fp32 per row of the wave-private output patch (128 B)


// tanh(a) sigmoid(b) = (1 - E) / ((1 + E)(1 + F)), E = e^(-2a), F = e^(-b).  a is clamped to [-16, 16] first (one v_med3;
// tanh(+-16) rounds to +-1 in fp32, so the clamp changes no result): E stays finite, the sign comes out of 1 - E, and no
// abs / copysign pair is needed.  F may overflow to +inf: the denominator is +inf then and the gate 0, which is the limit.
// The same arithmetic, element for element, as gate_fast of tools/csrc/ap_resblock_bf16.hip (the two kernels are bit-identical).
// On a pair of values: plain arithmetic as two-wide fp32 operations (v_pk_mul_f32 / v_pk_add_f32: one issue slot for two
// gates; the file is built with -fno-slp-vectorize, so the pairing is written out), the three transcendentals per gate
// stay scalar; the caller converts the pair to bf16 with one v_cvt_pk_bf16_f32.
__device__ __forceinline__ f32x2 gate_fast2(f32x2 a, f32x2 b) {
  const f32x2 ac = {__builtin_amdgcn_fmed3f(a[0], -16.0f, 16.0f), __builtin_amdgcn_fmed3f(a[1], -16.0f, 16.0f)};
  const f32x2 ea = ac * -2.885390081777926815f;
  const f32x2 eb = b * -1.442695040888963407f;
  const f32x2 E = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
  const f32x2 F = {__builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
  const f32x2 den = (E + 1.0f) * (F + 1.0f);
  const f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  return (1.0f - E) * r;
}

// gate_fast2 that also hands out the gate's two derivative factors (the white-box backward's operands: ap_resblock_bwd_bf16.hip):
// f1 = d(tanh . sigmoid)/d(tanh arg) = sg (1 - th^2), f2 = d/d(sigmoid arg) = th sg (1 - sg), from the quantities the gate forms anyway
// (sg = (1 + E) r, th = g (1 + F), th sg = g).  The returned gate is gate_fast2's, operation for operation: a forward pass that keeps
// the factors writes the same h' and g image as one that does not.  F = +inf (sigmoid argument below -88): g = sg = 0 and 0 . inf is
// taken as 0.
__device__ __forceinline__ f32x2 gate_fast2_save(f32x2 a, f32x2 b, f32x2 &f1, f32x2 &f2) {
#pragma clang fp contract(off)                                  // one operation order in every instantiation (1 - th th as an fma in some, not in others: seen)
  const f32x2 ac = {__builtin_amdgcn_fmed3f(a[0], -16.0f, 16.0f), __builtin_amdgcn_fmed3f(a[1], -16.0f, 16.0f)};
  const f32x2 ea = ac * -2.885390081777926815f;
  const f32x2 eb = b * -1.442695040888963407f;
  const f32x2 E = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
  const f32x2 F = {__builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
  const f32x2 opE = E + 1.0f, opF = F + 1.0f;
  const f32x2 den = opE * opF;
  const f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 g = (1.0f - E) * r;
  const f32x2 sg = opE * r;
  f32x2 th = g * opF;
  th[0] = F[0] > 3.0e38f ? 0.f : th[0];
  th[1] = F[1] > 3.0e38f ? 0.f : th[1];
  f1 = sg * (1.0f - th * th);
  f2 = g * (1.0f - sg);
  return g;
}

using I0 = std::integral_constant<int, 0>;
using I1 = std::integral_constant<int, 1>;
using I2 = std::integral_constant<int, 2>;
using I3 = std::integral_constant<int, 3>;

}  // namespace

// ---- weight images -------------------------------------------------------------------------------------------
// GEMM1: [wave C/32][chunk C/32][kstep 6][rowtile 2][lane 64][8]; wave w owns gate channels [32w, 32w+32):
// row tile 0 = tanh rows, 1 = sigmoid rows; k-step ks of a chunk = tap ks/2, channels ch*32 + (ks&1)*16 + 8h + jj.
// perm (AP_PREC_BF16_STORE): the K order inside a chunk follows the channel order of the u image's rows -- position p holds the
// chunk's channel (p with bits 2 and 3 swapped), the order of a 32 x 32 accumulator tile's registers (ap_resblock_bf16u.hip).
__global__ void pack_w1_bf16_kernel(const float *__restrict__ w1f, __bf16 *__restrict__ out, int C, int perm) {
  const int NW = C / 32, NCH = C / KC_;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NW * NCH * 6 * 2 * 64 * 8;
  if (idx >= total) return;
  int jj = idx & 7;
  int lane = (idx >> 3) & 63;
  int rt = (idx >> 9) & 1;
  size_t rest = idx >> 10;
  int ks = rest % 6; rest /= 6;
  int ch = rest % NCH;
  int w = rest / NCH;
  int i = lane & 31, hh = lane >> 5;
  int tap = ks >> 1;
  int pos = (ks & 1) * 16 + 8 * hh + jj;
  if (perm) pos = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);
  int c = ch * KC_ + pos;
  int o = rt * C + 32 * w + i;
  out[idx] = (__bf16)w1f[((size_t)o * C + c) * 3 + tap];
}

// GEMM2: [wave][rowtile 2][kstep C/16][lane][8]; row tile 0 = res_conv rows of the wave's channels, 1 = skip rows.
__global__ void pack_w2_bf16_kernel(const float *__restrict__ w2f, __bf16 *__restrict__ out, int C) {
  const int NW = C / 32, NKS = C / 16;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NW * NKS * 2 * 64 * 8;
  if (idx >= total) return;
  int jj = idx & 7;
  int lane = (idx >> 3) & 63;
  size_t rest = idx >> 9;
  int ks = rest % NKS; rest /= NKS;
  int rt = rest & 1;
  int w = rest >> 1;
  int i = lane & 31, hh = lane >> 5;
  int k = ks * 16 + 8 * hh + jj;
  int o = rt * C + 32 * w + i;          // w2f = [res rows (C); skip rows (C)]
  out[idx] = (__bf16)w2f[(size_t)o * C + k];
}

// final conv's first 1x1: [wave S/64][rowtile 2][kstep S/16][lane 64][8]; wave w owns output rows [64w, 64w+64)
__global__ void pack_wf1_bf16_kernel(const float *__restrict__ wf1f, __bf16 *__restrict__ out, int S) {
  const int NKS = S / 16;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)S * S) return;
  int jj = idx & 7;
  int lane = (idx >> 3) & 63;
  size_t rest = idx >> 9;
  int ks = rest % NKS; rest /= NKS;
  int rt = rest & 1;
  int w = rest >> 1;
  int k = ks * 16 + 8 * (lane >> 5) + jj;
  int o = 64 * w + 32 * rt + (lane & 31);
  out[idx] = (__bf16)wf1f[(size_t)o * S + k];
}

// GEMM1 image for v_mfma_f32_16x16x32_bf16: [wave C/32][chunk C/32][k-step of 32 = tap 3][rowtile16 4][lane 64][8]; row tiles
// 0, 1 = tanh rows 0-15, 16-31 of the wave's 32 gate channels, 2, 3 = sigmoid rows; lane l: row l & 15, k = 8 (l >> 4) + jj,
// i.e. channel ch*32 + 8 (l >> 4) + jj of tap s -- the same k order inside a chunk as the 32x32x16 image (tap-major).
__global__ void pack_w1q_bf16_kernel(const float *__restrict__ w1f, __bf16 *__restrict__ out, int C) {
  const int NW = C / 32, NCH = C / KC_;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NW * NCH * 3 * 4 * 64 * 8;
  if (idx >= total) return;
  int jj = idx & 7;
  int lane = (idx >> 3) & 63;
  int rt = (idx >> 9) & 3;
  size_t rest = idx >> 11;
  int tap = rest % 3; rest /= 3;
  int ch = rest % NCH;
  int w = rest / NCH;
  int c = ch * KC_ + 8 * (lane >> 4) + jj;
  int o = (rt >> 1) * C + 32 * w + 16 * (rt & 1) + (lane & 15);
  out[idx] = (__bf16)w1f[((size_t)o * C + c) * 3 + tap];
}

int launch_pack_bf16(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C, S = ctx->S, NL = ctx->NL;
  for (int n = 0; n < NL; n++) {
    size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
    pack_w1_bf16_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1, (__bf16 *)ctx->w1p_bf + n * n1, C,
                                                                             ctx->cfg.precision == AP_PREC_BF16_STORE ? 1 : 0);
    pack_w2_bf16_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, (__bf16 *)ctx->w2p_bf + n * n2, C);
    if (ctx->w1q_bf && C == 256)
      pack_w1q_bf16_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1, (__bf16 *)ctx->w1q_bf + n * n1, C);
  }
  if (ctx->wf1p_bf && S % 64 == 0)
    pack_wf1_bf16_kernel<<<(unsigned)(((size_t)S * S + 255) / 256), 256, 0, st>>>(ctx->wf1f, (__bf16 *)ctx->wf1p_bf, S);
  AP_HIP(hipGetLastError());
  return 0;
}

#ifdef AP_TOOLS
// u = bf16(h + part_t) as the chain's operand image [clip][C / 32][L][32] (layer 0 of a sweep; every later layer's image is
// written by the previous layer's epilogue).  Lane = (sample, channel octet): eight reads L apart (each 64 B per 16 lanes),
// one 16-byte store; a wave writes 1 KB contiguous.
__global__ __launch_bounds__(256) void make_ub_kernel(const float *__restrict__ h, const float *__restrict__ pt,
                                                      __bf16 *__restrict__ ub, int C, int L, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // ((clip * C/32 + chunk) * L + t) * 4 + octet
  if (idx >= total) return;
  const int oct = (int)(idx & 3);
  const size_t row = idx >> 2;                                    // (clip * C/32 + chunk) * L + t
  const int t = (int)(row % (size_t)L);
  const size_t bc = row / (size_t)L;                              // clip * C/32 + chunk
  const int c0 = (int)(bc % (size_t)(C / 32)) * 32 + oct * 8;
  const float *hp = h + ((bc / (size_t)(C / 32)) * C + c0) * (size_t)L + t;
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const f32x2 v2 = {hp[(size_t)(2 * e) * L] + pt[c0 + 2 * e], hp[(size_t)(2 * e + 1) * L] + pt[c0 + 2 * e + 1]};
    o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
  }
  reinterpret_cast<u32x4 *>(ub)[idx] = o;
}

int launch_make_ub(const float *h, const float *pt, void *ub, int B, int C, int L, hipStream_t st) {
  const size_t total = (size_t)B * (C / 32) * L * 4;
  make_ub_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(h, pt, (__bf16 *)ub, C, L, total);
  AP_HIP(hipGetLastError());
  return 0;
}

__device__ unsigned long long *g_ptrace = nullptr;               // DBG 2048: [workgroup][wave][64] s_memtime stamps of one tile
#endif

// DBG (tools builds only; outputs wrong by construction): 1 no weight loads in GEMM1's loop, 2 no X loads, 4 no pack,
// 8 no GEMM1 MFMA, 16 no B-fragment LDS reads, 32 no gate math, 64 no GEMM2 MFMA, 128 no read-modify-write loads,
// 256 no stores, 512 no per-chunk barrier, 1024 every tile stages the same 128 columns of clip 0, 2048 phase stamps, 0x2000 no priority swap,
// 0x4000 default cache policy instead of nt (exact), 0x8000 contiguous walk inside a clip for every dilation (exact).
// RAG: clip lengths that are not a multiple of four.  Channel rows are then only 4-byte aligned (16-byte accesses at 4-byte
// aligned addresses are exact on this chip: tools/micro/unaligned_b128.hip), the clip's last column quad is partly outside
// it: staged samples are zeroed one by one (the quad's count of valid samples instead of one mask), and the epilogue stores
// of that quad are one to three dwords.
// M16 (instantiated in the tools library only): GEMM1 on v_mfma_f32_16x16x32_bf16 -- 16-row x 16-column tiles, 32 k per instruction,
// weight image w1q (pack_w1q), the same X image read with the 16 x 4 lane pattern (224-byte rows: conflict-free), accumulators 4 x 8
// tiles of 4 registers; gate, GEMM2, epilogues unchanged.  Outputs are BIT-IDENTICAL to the 32x32x16 form (the matrix pipe's fp32
// accumulation does not depend on how k is grouped into instructions) and the launch is 5-6 % SLOWER (5.44 against 5.12 ms at
// B = 256: twice the MFMA issue slots, a fragment ring of 32 registers in a kernel that had none to spare); the -6 % of the
// timing-only substitution (DBG 0x100000: two 16x16x32 on the SAME operand registers per 32x32x16) came from operand reuse, which
// a real tiling does not have.  DESIGN.md 3.4.
// UB (round 3 experiment, instantiated in the tools library only; ap_capi.hip run_net under tools bit 0x400000): the GEMM1 operand
// comes ready-made.  Beside h' (fp32, what the residual
// and the next layer's residual need) the first epilogue writes ub' = bf16(h' + part_t of the NEXT layer) -- the value the next
// layer's staging would compute, so the two forms are bit-identical -- as [clip][32-channel chunk][sample][32 channels]: a chunk's
// (column, tap) operand is one 64-byte run, staged with a 16-byte load + ds_write_b128 per lane (three per thread and chunk for
// every dilation and every clip length: no FiLM add / convert / mask in the loop, no alignment cases, 24.6 KB per chunk on the
// CU's memory path instead of 64), out-of-clip taps read through an out-of-range offset (zeros).  A wave's first-pass rows are
// exactly one chunk, so its ub' stores are contiguous 2 KB runs.  Layer 0's image comes from make_ub_kernel.  Result: eps of the
// 36-layer sweep BIT-IDENTICAL to the layer-wise form (L = 16000, 4001, 1002, 643, 130), each launch +1 %, the sweep +3 %
// (tools/ab_bf16_ub.py, profiles/r3_bf16_operand_images_experiment.txt): the block runs at the board's 1400 W power cap
// (tools/power_check.py), and 64 KB of extra stores per tile cost more energy than 0.3 MB less on the L2 -> CU path and the
// pack's VALU save.  (The "no X requests / no pack" ablations that promised 13-25 % leave the X image CONSTANT: what they measure
// is the matrix pipe's data-dependent power, not the staging -- tools/power_ablate_bf16.py.)
// DS (round 4, "deferred skip"): the block writes h' (fp32) and the gate output g as the bf16 image GEMM2 consumes anyway --
// [clip][sample][256 channels], 512 B per sample, lossless for the skip sum because skip_conv only ever sees bf16(g) -- and does
// NOT run the skip half of GEMM2 or its read-modify-write of `skip`.  skipgemm_bf16_kernel (ap_skipgemm_bf16.hip) adds
// sum_n W_skip,n g_n for a group of layers into `skip` in one K-concatenated GEMM.  h' is bit-identical to the fused form
// (same pass-0 code); per tile and layer the block moves 131 KB in + 131 KB + 64 KB out instead of 262 + 262 KB.
// NOH (with DS): the net's LAST layer -- its h' is never read (WaveNet.py:131-135 returns only the skip sum), so res_conv, the
// residual's re-read of h and the h' store are left out: GEMM1, the gate and the g image only.
// SAVEF (with DS; the differentiable purifier's forward pass, ap_resblock_fwd_gate_save): the gate's two derivative factors are also
// written, as an fp16 pair per (channel, sample), in the ORDER THE ACCUMULATORS HOLD THEM -- [clip][tile][wave 8][column tile 4][q 4][lane 64]
// x 16 bytes (128 KB per tile): one 16-byte store per lane and (column tile, q), 1 KB contiguous per wave; the backward's gate kernel has
// the same wave / lane / register geometry and reads them back the same way (ap_resblock_bwd_bf16.hip).  h' and the g image are
// bit-identical to the launch without it.
template <int DBG, int WS = -1, bool RAG = false, bool M16 = false, bool UB = false, bool DS = false, bool NOH = false, bool SAVEF = false>   // WS >= 0: window staging (d <= 32); WS = d mod 4 as far as the code needs it: 0, 1 (d = 1), 2 (d = 2)
__global__ __launch_bounds__(512, 2) void resblock_bf16p_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const void *__restrict__ wbase, unsigned wbytes, unsigned w1_off, unsigned w2_off,        // bf16 weight images (one slab)
    const void *__restrict__ bbase, unsigned bbytes, unsigned b1_off, unsigned b2_off,        // fp32 bias vectors (one slab)
    int L, int d, int accumulate, int ntiles, int nblk,
    const void *__restrict__ ubin, void *__restrict__ ubout, const float *__restrict__ ptn,     // UB: bf16 operand images in / out (out may be null), the next layer's part_t
    void *__restrict__ gout,                                                                    // DS: this layer's g image [clip][L][256] bf16
    void *__restrict__ fout = nullptr) {                                                        // SAVEF: the gate's derivative factors
  constexpr int C = 256, NW = 8, NCH = C / KC_, NKS = C / 16;
  static_assert(!SAVEF || (DS && !M16), "SAVEF: a form of the deferred-skip block");
  static_assert(!UB || (WS < 0 && !M16), "UB: one staging form");
  static_assert(!DS || (!UB && !M16), "DS: the product staging forms only");
  static_assert(!NOH || DS, "NOH: a form of the deferred-skip block");
  // cache policy: nt (aux bit 1) on the once-touched streams (the running skip rows in, both outputs out) and on the residual's
  // re-read of h (it hits what is still there and allocates nothing on a miss).  Only the tap loads and the weights allocate in
  // the XCD's L2, so h rows stay until the neighbouring tiles' taps and the residual have read them again: L2-miss reads 27.2 ->
  // 19.4 GB per 512-clip launch (traffic 1.31 -> 1.08 x algorithmic), -3 % time.  DBG 0x4000: default policy everywhere.
  constexpr int NT = (DBG & 0x4000) ? 0 : 2;
  // store policy experiments (tools): DBG 0x1000000 default, 0x2000000 sc0 + nt, 0x3000000 sc1 + nt, 0x5000000 sc1 (agent scope write-through)
  constexpr int NTS = (DBG & 0x7000000) == 0x1000000 ? 0 : (DBG & 0x7000000) == 0x2000000 ? 3 : (DBG & 0x7000000) == 0x3000000 ? 18 : (DBG & 0x7000000) == 0x5000000 ? 16 : NT;
  constexpr bool WIN = WS >= 0;
  // M16 without the window: 224-byte rows and no swizzle -- with the 16 x 4 lane pattern of the 16x16x32 B fragment the 16-lane groups
  // of ds_read_b128 are conflict-free iff the row stride is 14 (or 2) slots of 16 B mod 16 (208 B = 13 slots is 2-way); the
  // pack's ds_write_b128 is 2-way there (16 LDS cycles against the 13 its register transfer takes: measured free in round 2).
  // The window variants keep 208-byte rows (their scratch column quad leaves no LDS for longer ones).
  constexpr int XS = (M16 && !WIN) ? 112 : XS_;
  constexpr int SWZ = ((M16 && !WIN) || UB) ? 0 : 1;            // (UB: column pairs four apart per ds_write_b128 group -- conflict-free unswizzled)
  constexpr int XBYTES = (PT_ + (WIN ? 4 : 0)) * XS * 2;        // 26,624 B per X buffer (WIN: + a scratch column quad), two buffers
  constexpr int GOFF = 2 * XBYTES;
  constexpr int POFF = GOFF + PT_ * GS_ * 2;                   // output patches: 8 waves x 32 x 32 fp32
  constexpr int PTOFF = POFF + NW * 32 * PS_ * 4;              // part_t (C floats)
  constexpr int BOFF = PTOFF + C * 4;                          // b1 (2C floats: filter | gate rows), b2 (2C: res | skip rows)
  constexpr int PNOFF = BOFF + 4 * C * 4;                      // UB: the next layer's part_t (C floats)
  constexpr int LDS_BYTES = PNOFF + (UB ? C * 4 : 0);
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;

  // ---- tile walk: workgroups g, g+8, g+16, ... share an XCD (round-robin dispatch); each XCD takes a contiguous run of
  // (clip, tile) work and its workgroups interleave inside that run, so at any moment the CUs of an XCD hold neighbouring
  // tiles of one clip and the +-d taps / the residual patch are re-read from that XCD's L2.  Placement only affects speed.
  int t_first, t_step, t_end;
  {
    const int g = blockIdx.x, G = gridDim.x;
    if (G >= 8 && (G & 7) == 0) {
      const int xcd = g & 7, idx = g >> 3, q = nblk >> 3, r = nblk & 7;
      const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      t_first = base + idx;
      t_step = G >> 3;
      t_end = base + q + (xcd < r ? 1 : 0);
    } else {
      t_first = g;
      t_step = G;
      t_end = nblk;
    }
  }
  if (t_first >= t_end) return;

  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  auto clip_rsrc = [&](const float *base, int b) {
    const uint64_t hb = (uint64_t)(base + (size_t)b * C * L);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)clip_bytes, 0x00020000);
  };

  // ---- X staging geometry (fixed per thread): (tap = wave/2, column quad cg, channel octet oct); lane bits (low to high)
  // cg&3, oct, cg>>2 so a load covers whole 64-B runs per channel row.  Waves 6, 7 repeat tap 2.
  const int xtap = min(wave >> 1, 2);
  const int cg = ((wave & 1) * 4 + (lane >> 4)) * 4 + (lane & 3), oct = (lane >> 2) & 3;
  int xcol = 4 * cg, xk = (xtap * KC_ + oct * 8) ^ (SWZ * ((__builtin_popcount(cg & 7) & 1) << 4));   // ^: the X image's 32-byte swizzle (below)
  // (M16: both, and xwb below, are re-derived from the lane id in pack_ptv -- kept from here they are spilled, and a scratch
  //  reload in the chunk loop is a vector-memory load whose wait drains every request in flight)

  if (tid < C) reinterpret_cast<float *>(lds + PTOFF)[tid] = pt[tid];
  if constexpr (UB) {
    if (tid < C) reinterpret_cast<float *>(lds + PNOFF)[tid] = ubout ? ptn[tid] : 0.f;
  }
  // the bias vectors live in LDS for the whole kernel: fetched per tile from memory they sat behind the previous tile's
  // stores in the in-order vmcnt queue, kept in registers they spilled
  {
    const unsigned char *bb = static_cast<const unsigned char *>(bbase);
    reinterpret_cast<float *>(lds + BOFF)[tid] = reinterpret_cast<const float *>(bb + b1_off)[tid];
    reinterpret_cast<float *>(lds + BOFF)[2 * C + tid] = reinterpret_cast<const float *>(bb + b2_off)[tid];
  }

  float xr[32];
  // walk order inside a clip: for d >= 2 tiles the +-d taps of a tile are the centre columns of tiles s = d / PT_ away, and the
  // 32 tiles an XCD holds at a time are 32 consecutive walk positions.  Position p -> tile r + k s (residue classes r = 0..s-1
  // in turn, s capped at 16), so that a window of 32 positions holds whole runs ..., j - s, j, j + s, ...: L2-miss reads of
  // the d = 1024 / 2048 layers 28.5 / 31.9 -> see profiles/r2_bf16_fetch_by_layer.txt.  Placement only: results unchanged.
  const int wstep = __builtin_amdgcn_readfirstlane(min(max(d / PT_, 1), 16));
  const int wq = ntiles / wstep, wrem = ntiles % wstep;
  auto tile_bt = [&](int tile, int &b, int &t0) {
    b = __builtin_amdgcn_readfirstlane(tile / ntiles);
    int p = tile % ntiles;
    if ((DBG & 0x8000) == 0 && wstep > 1) {
      const int cut = wrem * (wq + 1);
      const int r = p < cut ? p / (wq + 1) : wrem + (p - cut) / wq;
      const int k = p < cut ? p % (wq + 1) : (p - cut) % wq;
      p = r + k * wstep;
    }
    t0 = __builtin_amdgcn_readfirstlane(p * PT_);
  };
  // zero padding (WaveNet.py:26-27) as an AND mask on the packed values: a column quad is inside the clip or outside it as a
  // whole (L % 4 == 0), the address is clamped.
  // WIN (d <= 32): the three taps of a tile are the same 128 + 2d columns, so a chunk stages ONE window of 192 columns
  // [t0 - 32, t0 + 160) -- 24.6 KB of loads instead of 64 KB -- converts every value once and writes it to the up to three
  // (column, tap) places of the X image it belongs to (ds_write_b64: four channels of one column).  A staging unit is
  // (column quad sq of 48, channel quad wc4 of 8): waves 0-5, lane bits (low to high) wc4, sq & 7, so a load covers eight
  // channel rows x 128 B and the sixteen lanes of a ds_write_b64 group cover all 32 banks.  The four samples of a unit land,
  // per tap, in one image quad (d % 4 == 0) or in two neighbouring ones (d = 1, 2): one address register per (tap, quad),
  // the sample's place inside the quad is an immediate.  A quad outside columns 0..127 is the scratch quad 128..131.
  struct Keep { unsigned m[1]; };
  Keep keep;
  const int sq = min(wave, 5) * 8 + (lane >> 3), wc4 = lane & 7;
  unsigned xwb[WIN ? 3 : 1][WIN ? 2 : 1];                      // [tap][first / second image quad of the unit's samples]
  auto calc_xwb = [&](int sq_, int wc4_) {
    if constexpr (WIN) {
#pragma unroll
      for (int T = 0; T < 3; T++) {
        const int qa = sq_ - 8 + ((T == 2 && WS != 0) ? -1 : 0) - (WS == 0 ? (T - 1) * (d >> 2) : 0);   // image quad of sample 0
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
          const int q = qa + h2;
          const int qc = (q >= 0 && q < PT_ / 4) ? q : PT_ / 4;
          xwb[T][h2] = (unsigned)(4 * qc * (XS * 2) + ((T * 2 * KC_ + wc4_ * 8) ^ ((__builtin_popcount(qc & 7) & 1) << 5)));
        }
      }
    }
  };
  calc_xwb(sq, wc4);
  // UB staging unit: thread = (column 16 wave + cw, channel octet ln & 3) for the three taps; cw pairs columns four apart inside
  // an 8-lane ds_write_b128 group (832 B apart = 16 banks: conflict-free), a wave's load is 16 columns x 64 B = 1 KB contiguous
  unsigned xv[UB ? 3 : 1];                                       // byte offsets of the three taps' rows inside the clip's image
  u32x4 xq[UB ? 3 : 1];
  auto ub_col = [&](int ln) { const int q = ln >> 2; return 16 * wave + (((q & 1) << 2) | ((q >> 1) & 3) | (q & 8)); };
  auto x_geom = [&](int t0_in, unsigned &voff, Keep &k) {
    const int t0 = (DBG & 1024) ? 8192 : t0_in;                  // timing-only: every tile stages the same 128 columns of clip 0
    if constexpr (UB) {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int col = ub_col(ln);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int t = t0 + col + (i - 1) * d;
        xv[i] = (t >= 0 && t < L) ? (unsigned)(t * 64 + (ln & 3) * 16) : 0x80000000u;     // outside the clip: zeros (WaveNet.py:26-27)
      }
      (void)voff; (void)k;
      return;
    }
    // the lane's geometry is re-derived from a lane id read here (volatile asm: not hoisted out of the tile loop): kept from
    // the prologue it was spilled, and a scratch reload waits with vmcnt(0) -- here, behind the first epilogue's stores
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    if constexpr (WIN) {
      const int tq = t0 - 32 + 4 * (min(wave, 5) * 8 + (ln >> 3));
      if constexpr (RAG) {                                       // (tq < 0: the whole quad is outside -- tile starts and the window's
        k.m[0] = tq >= 0 ? (unsigned)min(max(L - tq, 0), 4) : 0u;     //  left edge are multiples of four)
        voff = ((unsigned)min(max(tq, 0), L - 1) + (unsigned)((ln & 7) * 4) * (unsigned)L) * 4u;
      } else {
        k.m[0] = (tq >= 0 && tq < L) ? 0xffffffffu : 0u;
        voff = ((unsigned)min(max(tq, 0), L - 4) + (unsigned)((ln & 7) * 4) * (unsigned)L) * 4u;
      }
    } else {
      const int cg = ((wave & 1) * 4 + (ln >> 4)) * 4 + (ln & 3), oct = (ln >> 2) & 3;
      const int tp = t0 + 4 * cg + (xtap - 1) * d;
      if constexpr (RAG) {
        k.m[0] = tp >= 0 ? (unsigned)min(max(L - tp, 0), 4) : 0u;
        voff = ((unsigned)min(max(tp, 0), L - 1) + (unsigned)(oct * 8) * (unsigned)L) * 4u;
      } else {
        k.m[0] = (tp >= 0 && tp < L) ? 0xffffffffu : 0u;
        voff = ((unsigned)min(max(tp, 0), L - 4) + (unsigned)(oct * 8) * (unsigned)L) * 4u;
      }
    }
  };
  const __amdgpu_buffer_rsrc_t hrs_clip0 = clip_rsrc(hin, 0);
  auto issue_x1 = [&](const __amdgpu_buffer_rsrc_t &rs_in, unsigned voff, int ch, int e) {       // channel e of chunk ch
    if constexpr (DBG & 2) return;
    const __amdgpu_buffer_rsrc_t &rs = (DBG & 1024) ? hrs_clip0 : rs_in;      // timing-only: every tile stages clip 0 (cache-resident X)
    // (bit_cast the whole vector: element-wise bit_cast of the builtin's int vector is mis-folded to a splat)
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (ch * KC_ + e) * L * 4, 0));
#pragma unroll
    for (int i = 0; i < 4; i++) xr[e * 4 + i] = v[i];
  };
  auto issue_x = [&](const __amdgpu_buffer_rsrc_t &rs_in, unsigned voff, int ch) {
    if constexpr (UB) {                                          // rs_in: the clip's bf16 image; chunk ch = L x 64 bytes
      if constexpr (DBG & 2) return;
#pragma unroll
      for (int i = 0; i < 3; i++) xq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, xv[i], ch * L * 64, 0));
      (void)voff;
    } else if constexpr (WIN) {
      if (wave < 6) {                                            // wave-uniform
#pragma unroll
        for (int e = 0; e < 4; e++) issue_x1(rs_in, voff, ch, e);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) issue_x1(rs_in, voff, ch, e);
    }
  };
  float ptv8[8];
  u32x4 pkq;
  unsigned xwa = 0;                                             // UB: this thread's place in an X buffer (tap 0), re-derived per chunk
  auto pack_ptv = [&](int ch) {
    int ln;                                                     // (lane id read here, not kept: see x_geom)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    if constexpr (UB) {
      xwa = (unsigned)((ub_col(ln) * XS + (ln & 3) * 8) * 2);
      (void)ch;
      return;
    }
    if constexpr (M16) {                                        // the pack's LDS addresses live from here to the chunk's last piece only
      if constexpr (WIN) calc_xwb(min(wave, 5) * 8 + (ln >> 3), ln & 7);
      else {
        const int cg_ = ((wave & 1) * 4 + (ln >> 4)) * 4 + (ln & 3), oct_ = (ln >> 2) & 3;
        xcol = 4 * cg_;
        xk = (xtap * KC_ + oct_ * 8) ^ (SWZ * ((__builtin_popcount(cg_ & 7) & 1) << 4));
      }
    }
    if constexpr (WIN) {
      const float4 p0 = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(lds + PTOFF) + (ln & 7) * 4 + ch * KC_);
      ptv8[0] = p0.x; ptv8[1] = p0.y; ptv8[2] = p0.z; ptv8[3] = p0.w;
    } else {
      const float *ptc = reinterpret_cast<const float *>(lds + PTOFF) + ((ln >> 2) & 3) * 8 + ch * KC_;
      const float4 p0 = *reinterpret_cast<const float4 *>(ptc);
      const float4 p1 = *reinterpret_cast<const float4 *>(ptc + 4);
      ptv8[0] = p0.x; ptv8[1] = p0.y; ptv8[2] = p0.z; ptv8[3] = p0.w;
      ptv8[4] = p1.x; ptv8[5] = p1.y; ptv8[6] = p1.z; ptv8[7] = p1.w;
    }
  };
  // one eighth of a chunk's staging.  FiLM add (WaveNet.py:84), zero padding (:26-27) as an AND with the in-range mask.
  //   !WIN: sample i, channel pairs 2hf, 2hf+1 -> 4 adds, 2 cvt_pk, 2 and; the ds_write_b128 follows a sample's second piece.
  //   WIN:  sample i: piece 0 = 4 adds, 2 cvt_pk, 2 and (four channels); piece 1 = its three ds_write_b64 (taps -d, 0, +d).
  auto pack_piece = [&](unsigned char *dst, const Keep &keep, auto i_tag, auto hf_tag) {
    constexpr int i = decltype(i_tag)::value, hf = decltype(hf_tag)::value;
    if constexpr (DBG & 4) return;
    if constexpr (UB) {                                          // pieces (0,0), (1,0), (2,0): tap i's 16 bytes straight into the image
      if constexpr (hf == 0 && i < 3) *reinterpret_cast<u32x4 *>(dst + xwa + i * (KC_ * 2)) = xq[i];
      (void)keep;
      return;
    }
    // RAG: keep.m[0] counts the quad's valid samples -> all ones for sample i iff i < count
    const unsigned km = RAG ? (unsigned)(((int)i - (int)keep.m[0]) >> 31) : keep.m[0];
    if constexpr (WIN) {
      if (wave < 6) {                                            // wave-uniform
        if constexpr (hf == 0) {
#pragma unroll
          for (int e2 = 0; e2 < 2; e2++)
            pkq[e2] = __builtin_bit_cast(unsigned, __builtin_convertvector(
                                                       f32x2{xr[(2 * e2) * 4 + i] + ptv8[2 * e2],
                                                             xr[(2 * e2 + 1) * 4 + i] + ptv8[2 * e2 + 1]}, bf16x2)) & km;
        } else {
#pragma unroll
          for (int T = 0; T < 3; T++) {
            // place of sample 0 inside its image quad: tap -d: d mod 4; centre: 0; tap +d: (-d) mod 4
            const int a0 = WS == 0 || T == 1 ? 0 : (T == 0 ? WS : 4 - WS);
            const int pos = a0 + i;
            *reinterpret_cast<uint2 *>(dst + xwb[T][pos >> 2] + (pos & 3) * (XS * 2)) = make_uint2(pkq[0], pkq[1]);
          }
        }
      }
    } else {
#pragma unroll
      for (int e2 = 2 * hf; e2 < 2 * hf + 2; e2++)
        pkq[e2] = __builtin_bit_cast(unsigned, __builtin_convertvector(
                                                   f32x2{xr[(2 * e2) * 4 + i] + ptv8[2 * e2],
                                                         xr[(2 * e2 + 1) * 4 + i] + ptv8[2 * e2 + 1]}, bf16x2)) & km;
      if constexpr (hf == 1) *reinterpret_cast<u32x4 *>(dst + ((xcol + i) * XS + xk) * 2) = pkq;
    }
  };
  auto pack_all = [&](unsigned char *dst, const Keep &keep, int ch) {
    pack_ptv(ch);
    pack_piece(dst, keep, I0{}, I0{}); pack_piece(dst, keep, I0{}, I1{});
    pack_piece(dst, keep, I1{}, I0{}); pack_piece(dst, keep, I1{}, I1{});
    pack_piece(dst, keep, I2{}, I0{}); pack_piece(dst, keep, I2{}, I1{});
    pack_piece(dst, keep, I3{}, I0{}); pack_piece(dst, keep, I3{}, I1{});
  };

  // ---- weight fragment streams (this wave's 64 GEMM1 rows / 2 x 32 GEMM2 rows), L2 -> registers.  Buffer loads with the
  // fragment index in the scalar offset: one VGPR (lane * 16) addresses every fragment (64-bit per-lane pointers, one pair
  // per fragment slot, were hoisted out of the tile loop by the compiler and spilled).
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc(wbase, wbytes);
  (void)bbytes;
  const unsigned lane16 = (unsigned)lane * 16u;
  // GEMM1 image [wave][chunk][kstep 6][rowtile 2][lane][8 bf16]: fragment f of this wave = f KB from the wave's base
  auto ld_w1 = [&](int frag) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, w1_off + (wave * NCH * 12 + frag) * 1024, 0));
  };
  // GEMM2 image [wave][rowtile 2][kstep 16][lane][8 bf16]; gks runs over pass 0's then pass 1's k-steps
  auto ld_w2 = [&](int gks) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, w2_off + (wave * 2 * NKS + gks) * 1024, 0));
  };
  // X image swizzle: columns whose bits 2..4 have odd parity keep the two 32-byte halves of every 64 bytes of k swapped
  // (k-step s of such a column sits where k-step s ^ 1 would).  A ds_write_b128 is served in groups of 8 consecutive lanes
  // = column quads cg..cg+3 x 2 octets; the 832-byte quad stride puts quads cg and cg+2 on the same banks (2-way); with
  // the swap the four quads of a group have parities 0,1,1,0 and land on four different bank octets.  The 16-lane groups
  // of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}: MI355X_MICROARCH.md, LDS) each hold rows of ONE parity, so the
  // reads stay the conflict-free permutation they were (a swap keyed on a single column bit made them 2-way: measured).
  // Cost: one VGPR (a base for even and one for odd k-steps).
  // M16: image [wave][chunk][k32 step 3][rowtile16 4][lane][8] -> fragment fidx of this wave = fidx KB from the wave's base
  auto ld_w1q = [&](int fidx) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, w1_off + (wave * NCH * 12 + fidx) * 1024, 0));
  };
  // M16 B-fragment read: lane (c16 = lane & 15, q4 = lane >> 4) takes k-octet 4 s + q4 of column 16 ct16 + c16; the image's 32-byte
  // swizzle swaps octets o <-> o ^ 2 in columns whose bits 2..4 have odd parity: bits 2, 3 come from c16, bit 4 from ct16 & 1
  const int c16 = lane & 15, q4 = lane >> 4;
  const int par16 = __builtin_popcount((c16 >> 2) & 3) & 1;
  const int rd16e = (c16 * XS) * 2 + ((q4 ^ (SWZ * 2 * par16)) * 16);           // even ct16
  const int rd16o = (c16 * XS) * 2 + ((q4 ^ (SWZ * 2 * (par16 ^ 1))) * 16);     // odd ct16
  const int rdoff = (j * XS + 8 * hh) * 2;                      // this lane's B-fragment byte offset inside an X buffer
  const int rdsw = SWZ * (__builtin_popcount((j >> 2) & 7) & 1) * 32;
  const unsigned char *gb = lds + GOFF + (j * GS_ + 8 * hh) * 2;
  float *patch = reinterpret_cast<float *>(lds + POFF) + wave * 32 * PS_;
  const float RS = 0.707106781186547524f;

  // ---- first tile: parameters and chunk-0 request
  int b_cur, t0_cur;
  tile_bt(t_first, b_cur, t0_cur);
  __amdgpu_buffer_rsrc_t hrs = clip_rsrc(hin, b_cur);
  auto ub_rsrc = [&](const void *base, int b) {                  // a clip's bf16 image: [C / 32][L][32]
    const uint64_t hb = (uint64_t)base + (uint64_t)b * ((uint64_t)C * (uint64_t)L * 2u);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)(clip_bytes / 2u), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t urs = UB ? ub_rsrc(ubin, b_cur) : hrs;
#define AP_XRS (UB ? urs : hrs)
  unsigned xvoff;
  x_geom(t0_cur, xvoff, keep);
  // a tile's first X chunk and first weight fragments are requested at the END of the previous tile, ahead of that tile's
  // last stores (here for the first tile): a request issued after a store cannot be waited for without waiting for the
  // store (vmcnt retires in order)
  constexpr int RING = 3, PK = 3;                                // fragment ring depth (k-steps), k-step that carries the pack (a ring of
                                                                 // six with the pack in k-step 5 -- a chunk of lead for X -- measured +2 %)
  bf16x8 w[RING][2];                                                // GEMM1 fragment ring: k-step ks of a chunk uses w[ks % RING][row tile]
  bf16x8 wfq[2][4];                                                  // M16: ring of two k-steps of 32 (four 16-row fragments each)
  auto tile_head = [&]() {
    if constexpr (M16) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int rt = 0; rt < 4; rt++) wfq[s2][rt] = ld_w1q(s2 * 4 + rt);
    } else {
#pragma unroll
    for (int ks = 0; ks < RING; ks++)
#pragma unroll
      for (int rt = 0; rt < 2; rt++) w[ks][rt] = ld_w1(ks * 2 + rt);
    }
    pack_all(lds, keep, 0);
    issue_x(AP_XRS, xvoff, 1);                                   // chunk 1: packed in chunk 0's fourth k-step
  };
  issue_x(AP_XRS, xvoff, 0);
  __syncthreads();                                               // part_t, biases visible
  tile_head();

  int tile_iter = 0;
  auto mark = [&](int i) {
#ifdef AP_TOOLS
    if constexpr (DBG & 2048) {
      if (tile_iter == 6) {                                      // the 7th tile of every workgroup: steady state
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (lane == 0) g_ptrace[((size_t)blockIdx.x * NW + wave) * 64 + i] = t;
        if (i == 0 || i == 32) {                                 // wall clock (100 MHz) beside the shader clock: their ratio is the
          unsigned long long rt;                                 // clock the chip holds under this kernel
          asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
          if (lane == 0) g_ptrace[((size_t)blockIdx.x * NW + wave) * 64 + 40 + (i == 32)] = rt;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#endif
  };
  auto mark_half = [&](int ch) { mark(ch < NCH - 1 ? 3 + ch * 3 : 24); };
#pragma unroll 1
  for (int tile = t_first; tile < t_end; tile += t_step, tile_iter++) {
    mark(0);
    const int t0 = t0_cur;
    const int ntile = tile + t_step;
    // ================================================ GEMM1 =========================================================
    f32x16 acc[2][4];
    f32x4 acq[4][8];                                              // M16: [rowtile16][coltile16]; lane: column c16, rows 4 q4 + reg
    if constexpr (M16) {
#pragma unroll
      for (int rt = 0; rt < 4; rt++) {
        const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + ((rt >> 1) * C + 32 * wave + 16 * (rt & 1) + 4 * q4) * 4);
#pragma unroll
        for (int ct = 0; ct < 8; ct++) acq[rt][ct] = bv4;
      }
    } else
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + (rt * C + 32 * wave + 8 * q + 4 * hh) * 4);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {                        // (128 v_mov per wave and tile: entering the bias as the C operand of
          acc[rt][ct][4 * q + 0] = bv4[0];                      //  each tile's first MFMA instead measured the same -- hidden work)
          acc[rt][ct][4 * q + 1] = bv4[1];
          acc[rt][ct][4 * q + 2] = bv4[2];
          acc[rt][ct][4 * q + 3] = bv4[3];
        }
      }
    mark(1);
    __syncthreads();
    mark(2);

    auto mf = [&](const bf16x8 &a, const bf16x8 &bq, int rt, int ct) {
      if constexpr (DBG & 8) {
        asm volatile("" ::"v"(a), "v"(bq));
      } else if constexpr (DBG & 0x100000) {                   // timing only: the same FLOPs as two v_mfma_f32_16x16x32_bf16 (results wrong)
        f32x4 lo = __builtin_shufflevector(acc[rt][ct], acc[rt][ct], 0, 1, 2, 3);
        f32x4 hi = __builtin_shufflevector(acc[rt][ct], acc[rt][ct], 4, 5, 6, 7);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, hi, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) { acc[rt][ct][i] = lo[i]; acc[rt][ct][4 + i] = hi[i]; }
      } else {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[rt][ct], 0, 0, 0);
      }
    };
    auto rdb = [&](bf16x8 &dst, const unsigned char *xbe, const unsigned char *xbo, int ct, int ks) {   // ks: k-step 0..5 of the chunk
      if constexpr (DBG & 16) asm volatile("" : "=v"(dst));
      else dst = *reinterpret_cast<const bf16x8 *>(((ks & 1) ? xbo : xbe) + (32 * ct) * (XS * 2) + ks * 32);
    };
    float pre[4][16];
    unsigned evoff[4];                                          // E4 mapping: lane = (row lane>>3 (+8 per step), column quad lane&7)
    auto calc_evoff = [&]() {                                   // called after GEMM1 (t0 made opaque there: computed before the
      int t0o = t0;                                             // chunk loop the four offsets sat in registers through GEMM1)
      asm volatile("" : "+s"(t0o));
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        const int t = t0o + 32 * ct + 4 * (lane & 7);
        evoff[ct] = t < L ? ((unsigned)(32 * wave + (lane >> 3)) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      }                                                         // 0x80000000: outside the clip -> loads 0, store dropped
    };
    auto load_pre = [&](const __amdgpu_buffer_rsrc_t &rs, auto ct_tag) {
      constexpr int ct = decltype(ct_tag)::value;
      if constexpr (DBG & 128) {
#pragma unroll
        for (int r = 0; r < 16; r++) pre[ct][r] = 0.f;
      } else {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, evoff[ct], 8 * p * L * 4, NT));
#pragma unroll
          for (int i = 0; i < 4; i++) pre[ct][4 * p + i] = v[i];
        }
      }
    };
    // One chunk = six k-steps of eight MFMAs in explicit order (pinned with sched_barrier): column-tile-major pairs share one
    // B fragment, which is re-read for the next k-step right after its pair.  Global requests of a chunk and wave:
    //   * a k-step's two weight fragments are replaced right behind their last MFMA by those of the k-step three on (a ring
    //     of three k-steps, 24 registers: two k-steps = 16 MFMAs of lead, as the two half-chunk sets had at their tightest);
    //   * k-step 3: FiLM add / bf16 pack / ds_write of the next chunk (eight pieces, one per MFMA gap);
    //   * behind the last MFMA: the X rows of the chunk after next (the staging registers are free from the pack on, but
    //     every weight request issued before the next pack has to be OLDER than the X request -- vmcnt retires in order, so
    //     a wait for a younger fragment would be a wait for the X rows).
    // No spilled value may be reloaded inside the tile loop: a scratch reload is a vector-memory load, its wait is
    // s_waitcnt vmcnt(0), and that drains every request in flight (the round's earlier builds had one per chunk).
    // The two waves of a SIMD: the older one (waves 0-3) wins every issue conflict and reached the chunk barrier ~1.2 k cycles
    // ahead of the younger one, which then finished alone with its stalls exposed (tools/trace_resblock_bf16p.py): the
    // younger wave gets the priority in k-steps 0-2, the older one (by age) in 3-5.
    auto chunk = [&](const unsigned char *xbe, const unsigned char *xbo, int ch, unsigned char *pdst, auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;            // 0: chunks 0..NCH-3, 1: NCH-2 (no X request), 2: NCH-1
      constexpr bool LAST = KIND == 2, WITH_X = KIND == 0;
      bf16x8 bv[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) rdb(bv[ct], xbe, xbo, ct, 0);
      if constexpr (!(DBG & 0x2000)) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 6; ks++) {
        const bool reload = !(DBG & 1) && (ks + RING < 6 || !LAST);     // (ring 3) k-steps 3-5 fetch the next chunk's first three
        const int nfrag = ks + RING < 6 ? ch * 12 + (ks + RING) * 2 : (ch + 1) * 12 + (ks + RING - 6) * 2;
        if (ks == 3) {
          if constexpr (!(DBG & 0x2000)) __builtin_amdgcn_s_setprio(0);
          mark_half(ch);
        }
        if constexpr ((DBG & 2048) != 0 && (DBG & 0x8000000) != 0) {   // tools: stamps 42..47 = start of chunk 3's k-steps (DBG 2048 + 0x8000000)
          if (ch == 3) mark(42 + ks);
        }
        if (ks == PK) {
          if constexpr (!LAST) pack_ptv(ch + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          mf(w[ks % RING][0], bv[ct], 0, ct);
          if constexpr (!LAST) {
            if (ks == PK && ct == 0) pack_piece(pdst, keep, I0{}, I0{});
            if (ks == PK && ct == 1) pack_piece(pdst, keep, I1{}, I0{});
            if (ks == PK && ct == 2) pack_piece(pdst, keep, I2{}, I0{});
            if (ks == PK && ct == 3) pack_piece(pdst, keep, I3{}, I0{});
          }
          if (ct == 3 && reload) w[ks % RING][0] = ld_w1(nfrag);
          __builtin_amdgcn_sched_barrier(0);
          mf(w[ks % RING][1], bv[ct], 1, ct);
          if (ks < 5) rdb(bv[ct], xbe, xbo, ct, ks + 1);
          if constexpr (!LAST) {
            if (ks == PK && ct == 0) pack_piece(pdst, keep, I0{}, I1{});
            if (ks == PK && ct == 1) pack_piece(pdst, keep, I1{}, I1{});
            if (ks == PK && ct == 2) pack_piece(pdst, keep, I2{}, I1{});
            if (ks == PK && ct == 3) pack_piece(pdst, keep, I3{}, I1{});
          }
          if (ct == 3 && reload) w[ks % RING][1] = ld_w1(nfrag + 1);
          if constexpr (WITH_X) {
            if (ks == 5 && ct == 3) issue_x(AP_XRS, xvoff, ch + 2);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    // M16 chunk: three k-steps of 32 (one per tap), each 8 column tiles x 4 row tiles = 32 MFMAs of 16 cycles.  Four B fragments are
    // held at a time: a column tile's fragment is re-read for column tile + 4 (then for the next k-step) right behind its four
    // MFMAs.  A k-step's four weight fragments die with its last column tile and are replaced there by those of the k-step two
    // on (ring of two k-steps = 32 registers; slot = (3 ch + s) & 1, so the chunk loop runs in pairs with the phase a constant).
    // The middle k-step carries the next chunk's pack (one piece per column tile), the X request goes out behind the last MFMA.
    auto chunk16 = [&](const unsigned char *xb, int ch, unsigned char *pdst, auto kind_tag, auto ph_tag) {
      constexpr int KIND = decltype(kind_tag)::value, PH = decltype(ph_tag)::value;
      constexpr bool LAST = KIND == 2, WITH_X = KIND == 0;
      auto rdb16 = [&](bf16x8 &dst, int ct, int s2) {
        dst = *reinterpret_cast<const bf16x8 *>(xb + ((ct & 1) ? rd16o : rd16e) + (16 * ct) * (XS * 2) + s2 * 64);
      };
      bf16x8 bv[4];
#pragma unroll
      for (int c = 0; c < 4; c++) rdb16(bv[c], c, 0);
      if constexpr (!(DBG & 0x2000)) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 3; s2++) {
        const int slot = (PH + s2) & 1;
        const bool reload = s2 + 2 < 3 || !LAST;                // k-steps 1, 2 fetch the next chunk's first two
        const int nf = s2 + 2 < 3 ? (ch * 3 + s2 + 2) * 4 : ((ch + 1) * 3 + s2 + 2 - 3) * 4;
        if (s2 == 1) {
          if constexpr (!(DBG & 0x2000)) __builtin_amdgcn_s_setprio(0);
          if constexpr (!LAST) pack_ptv(ch + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < 8; ct++) {
#pragma unroll
          for (int rt = 0; rt < 4; rt++) {
            acq[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfq[slot][rt], bv[ct & 3], acq[rt][ct], 0, 0, 0);
            if (ct == 7 && reload) wfq[slot][rt] = ld_w1q(nf + rt);
            if constexpr (!LAST) {
              if (s2 == 1 && rt == 1) {
                if (ct == 0) pack_piece(pdst, keep, I0{}, I0{});
                if (ct == 1) pack_piece(pdst, keep, I0{}, I1{});
                if (ct == 2) pack_piece(pdst, keep, I1{}, I0{});
                if (ct == 3) pack_piece(pdst, keep, I1{}, I1{});
                if (ct == 4) pack_piece(pdst, keep, I2{}, I0{});
                if (ct == 5) pack_piece(pdst, keep, I2{}, I1{});
                if (ct == 6) pack_piece(pdst, keep, I3{}, I0{});
                if (ct == 7) pack_piece(pdst, keep, I3{}, I1{});
              }
            }
            if (rt == 3) {
              if (ct < 4) rdb16(bv[ct & 3], ct + 4, s2);
              else if (s2 < 2) rdb16(bv[ct & 3], ct - 4, s2 + 1);
            }
            if constexpr (WITH_X) {
              if (s2 == 2 && ct == 7 && rt == 3) issue_x(AP_XRS, xvoff, ch + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    if constexpr (M16) {
      static_assert(!M16 || NCH == 8, "paired chunk schedule");
#pragma unroll 1
      for (int ch = 0; ch < NCH - 2; ch += 2) {
        chunk16(lds, ch, lds + XBYTES, I0{}, I0{});
        __syncthreads();
        chunk16(lds + XBYTES, ch + 1, lds, I0{}, I1{});
        __syncthreads();
      }
      chunk16(lds, NCH - 2, lds + XBYTES, I1{}, I0{});
      __syncthreads();
      chunk16(lds + XBYTES, NCH - 1, nullptr, I2{}, I1{});
    } else {
#pragma unroll 1
    for (int ch = 0; ch < NCH - 2; ch++) {
      const unsigned char *xbe = lds + (ch & 1) * XBYTES + rdoff + rdsw, *xbo = xbe - 2 * rdsw;
      chunk(xbe, xbo, ch, lds + ((ch + 1) & 1) * XBYTES, I0{});
      mark(4 + ch * 3);
      if constexpr (!(DBG & 512)) __syncthreads();              // DBG 512 (timing only): no per-chunk barrier
      mark(5 + ch * 3);
    }
    {
      const unsigned char *xbe = lds + ((NCH - 2) & 1) * XBYTES + rdoff + rdsw, *xbo = xbe - 2 * rdsw;
      chunk(xbe, xbo, NCH - 2, lds + ((NCH - 1) & 1) * XBYTES, I1{});
      mark(4 + (NCH - 2) * 3);
      if constexpr (!(DBG & 512)) __syncthreads();
      mark(5 + (NCH - 2) * 3);
    }
    {
      const unsigned char *xbe = lds + ((NCH - 1) & 1) * XBYTES + rdoff + rdsw, *xbo = xbe - 2 * rdsw;
      chunk(xbe, xbo, NCH - 1, nullptr, I2{});
      mark(25);
    }
    }

    // ================================================ gate ==========================================================
    // GEMM2's first requests go out ahead of the gate: the CU's memory pipe serves in order and vmcnt retires in order.
    bf16x8 p0[4], p1[4];
    auto load_a4 = [&](bf16x8(&a)[4], int gks) {                 // gks = k-step over both passes, clamped to the last set
      const int g0 = gks <= 2 * NKS - 4 ? gks : 2 * NKS - 4;
#pragma unroll
      for (int s = 0; s < 4; s++) a[s] = ld_w2(g0 + s);
    };
    float4 bias[4];
    auto fetch_bias = [&](auto pass_tag) {
      constexpr int pass = decltype(pass_tag)::value;
      int ln;                                                    // (lane id read here, not kept: see x_geom)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int hh = ln >> 5;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = 32 * wave + 8 * q + 4 * hh;
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + (2 * C + pass * C + 32 * wave + 8 * q + 4 * hh) * 4);
        float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
        if (pass == 0) {                                         // u = h + part_t re-enters the residual (alias semantics)
          const float4 pv = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(lds + PTOFF) + c);
          v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
        }
        bias[q] = v;
      }
    };
    if constexpr (!NOH) {
      calc_evoff();
      load_a4(p0, 0);
      load_pre(hrs, I0{});                                       // the residual's h patch, first half (what fits beside the accumulators)
      load_pre(hrs, I1{});
    }
    __builtin_amdgcn_sched_barrier(0);

    const __amdgpu_buffer_rsrc_t srs = clip_rsrc(skip, b_cur);
    const __amdgpu_buffer_rsrc_t ors = clip_rsrc(hout, b_cur);
    float pre1[4][16];                                           // the running skip rows pass 1 adds into
    auto load_pre1 = [&](auto ct_tag) {
      constexpr int ct = decltype(ct_tag)::value;
      if ((DBG & 128) || !accumulate) {
#pragma unroll
        for (int r = 0; r < 16; r++) pre1[ct][r] = 0.f;
      } else {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, evoff[ct], 8 * p * L * 4, NT));
#pragma unroll
          for (int i = 0; i < 4; i++) pre1[ct][4 * p + i] = v[i];
        }
      }
    };
    auto gate_ct = [&](auto ct_tag) {
      constexpr int ct = decltype(ct_tag)::value;
      int ln;                                                    // (lane id read here, not kept: see x_geom)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      if constexpr (M16) {                                       // the two 16-column tiles of this 32-column tile x both channel halves
        const int c16 = ln & 15, q4 = ln >> 4;
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
          for (int rtp = 0; rtp < 2; rtp++) {
            const f32x4 a4 = acq[rtp][2 * ct + h2], b4 = acq[rtp + 2][2 * ct + h2];
            unsigned pk[2];
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const f32x2 a2 = {a4[e], a4[e + 1]}, b2 = {b4[e], b4[e + 1]};
              const f32x2 g2 = (DBG & 32) ? a2 + b2 : gate_fast2(a2, b2);
              pk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(g2, bf16x2));
            }
            *reinterpret_cast<uint2 *>(lds + GOFF + ((32 * ct + 16 * h2 + c16) * GS_ + 32 * wave + 16 * rtp + 4 * q4) * 2) = make_uint2(pk[0], pk[1]);
          }
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
      const int j = ln & 31, hh = ln >> 5;
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        unsigned pk[2];
        u32x4 fq;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 a2 = {acc[0][ct][4 * qq + e], acc[0][ct][4 * qq + e + 1]};
          const f32x2 b2 = {acc[1][ct][4 * qq + e], acc[1][ct][4 * qq + e + 1]};
          f32x2 g2;
          if constexpr (SAVEF) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            f32x2 f1, f2;
            g2 = gate_fast2_save(a2, b2, f1, f2);
            fq[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{f1[0], f2[0]}, f16x2));       // (tanh factor, sigmoid factor) of element e
            fq[e + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{f1[1], f2[1]}, f16x2));
          } else {
            g2 = (DBG & 32) ? a2 + b2 : gate_fast2(a2, b2);
          }
          pk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(g2, bf16x2));
        }
        *reinterpret_cast<uint2 *>(lds + GOFF + ((32 * ct + j) * GS_ + 32 * wave + 8 * qq + 4 * hh) * 2) = make_uint2(pk[0], pk[1]);
        if constexpr (SAVEF) {
          const uint64_t fb = (uint64_t)fout + (uint64_t)b_cur * ((uint64_t)ntiles * 131072u);
          const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)fb), fhi = __builtin_amdgcn_readfirstlane((uint32_t)(fb >> 32));
          const __amdgpu_buffer_rsrc_t frs =
              __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)fhi << 32) | flo), 0, (int)((unsigned)ntiles * 131072u), 0x00020000);
          // (the whole offset in the VGPR, soffset = 0: a >8-byte buffer store with an SGPR soffset reads its data late and the compiler
          //  does not guard the next write of those VGPRs -- the torn lanes of round 1, seen again here in the window-staging instantiations)
          __builtin_amdgcn_raw_buffer_store_b128(fq, frs, (unsigned)ln * 16u + (unsigned)((((t0 / PT_) * 8 + wave) * 16 + ct * 4 + qq) * 1024), 0, NTS);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // every gated column tile frees 32 accumulator registers: the rest of the h patch, then the first half of the skip rows
    gate_ct(I0{});
    if constexpr (!NOH) load_pre(hrs, I2{});
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I1{});
    if constexpr (!NOH) load_pre(hrs, I3{});
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I2{});
    if constexpr (!DS) load_pre1(I0{});
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I3{});
    if constexpr (!DS) load_pre1(I1{});
    if constexpr (!NOH) {
      fetch_bias(I0{});
      load_a4(p1, 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(26);
    __syncthreads();
    mark(27);

    // ================================================ GEMM2 =========================================================
    // two passes of 32 rows x 128 columns: pass 0 = res_conv rows -> h', pass 1 = skip_conv rows -> skip (WaveNet.py:93-97,:133)
    auto gemm2_loop = [&](f32x16(&ac)[4], int pass) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 v = bias[q];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          ac[ct][4 * q + 0] = v.x;
          ac[ct][4 * q + 1] = v.y;
          ac[ct][4 * q + 2] = v.z;
          ac[ct][4 * q + 3] = v.w;
        }
      }
      // B fragments (g image) are read one k-step ahead into the other of two register sets: the LDS latency of a k-step's
      // four reads hides behind the previous k-step's four MFMAs (read just before use it was exposed 32 times per tile)
      bf16x8 ba[4], bb[4];
      auto rdg = [&](bf16x8(&bq)[4], int ks) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * (GS_ * 2) + (ks & (NKS - 1)) * 32);
      };
      auto step = [&](const bf16x8 &a, const bf16x8(&use)[4], bf16x8(&nxt)[4], int ks) {
        rdg(nxt, ks + 1);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          if constexpr (DBG & 64) asm volatile("" ::"v"(a), "v"(use[ct]));
          else if constexpr (DBG & 0x100000) {
            f32x4 lo = __builtin_shufflevector(ac[ct], ac[ct], 0, 1, 2, 3), hi = __builtin_shufflevector(ac[ct], ac[ct], 4, 5, 6, 7);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, use[ct], lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, use[ct], hi, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) { ac[ct][i] = lo[i]; ac[ct][4 + i] = hi[i]; }
          } else ac[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, use[ct], ac[ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      rdg(ba, 0);
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += 8) {
        step(p0[0], ba, bb, ks + 0); step(p0[1], bb, ba, ks + 1); step(p0[2], ba, bb, ks + 2); step(p0[3], bb, ba, ks + 3);
        if (!DS || ks + 8 < NKS) load_a4(p0, pass * NKS + ks + 8);      // (DS: one pass -- nothing to prefetch past its last k-steps)
        __builtin_amdgcn_sched_barrier(0);
        step(p1[0], ba, bb, ks + 4); step(p1[1], bb, ba, ks + 5); step(p1[2], ba, bb, ks + 6); step(p1[3], bb, ba, ks + 7);
        if (!DS || ks + 12 < NKS) load_a4(p1, pass * NKS + ks + 12);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // epilogue of a pass: MFMA layout (4 rows x 1 column per lane) -> wave-private LDS patch -> 1 row x 4 columns per lane,
    // so the stores (like the operand loads) are 16 B per lane
    const __amdgpu_buffer_rsrc_t uors = UB ? ub_rsrc(ubout, b_cur) : hrs;
    auto epilogue = [&](const f32x16(&ac)[4], const float(&add)[4][16], const __amdgpu_buffer_rsrc_t &dst, float scale,
                        auto first_tag) {
      constexpr bool UBW = UB && decltype(first_tag)::value;     // pass 0 of the chain's form also writes the next layer's operand image
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        if constexpr (decltype(first_tag)::value && !DS) {      // pass 0: the h-patch registers of tiles 0, 1 are free again
          if (ct == 2) load_pre1(I2{});
          if (ct == 3) load_pre1(I3{});
        }
#pragma unroll
        for (int r = 0; r < 16; r++) patch[rowoff(r, hh) * PS_ + j] = ac[ct][r];
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const float4 v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * p) * PS_ + 4 * (lane & 7));
          f32x4 o;
          o[0] = (add[ct][4 * p + 0] + v.x) * scale;
          o[1] = (add[ct][4 * p + 1] + v.y) * scale;
          o[2] = (add[ct][4 * p + 2] + v.z) * scale;
          o[3] = (add[ct][4 * p + 3] + v.w) * scale;
          // row step in the VGPR offset, soffset = 0: a >8-byte buffer store with an SGPR soffset reads its data late and
          // the compiler does not guard the next write of those VGPRs (observed in round 1: torn lanes)
          if constexpr (DBG & 256) asm volatile("" ::"v"(o));
          else if constexpr (RAG) {
            const int nv = L - (t0 + 32 * ct + 4 * (lane & 7));  // valid samples of this lane's column quad (>= 4: all)
            const unsigned so = evoff[ct] + (unsigned)(8 * p * L * 4);
            const u32x4 ou = __builtin_bit_cast(u32x4, o);       // (whole-vector bit_cast: element-wise it is mis-folded to a splat)
            if (nv >= 4) __builtin_amdgcn_raw_buffer_store_b128(ou, dst, so, 0, NTS);
            else {
              if (nv >= 1) __builtin_amdgcn_raw_buffer_store_b32(ou[0], dst, so, 0, NT);
              if (nv >= 2) __builtin_amdgcn_raw_buffer_store_b32(ou[1], dst, so + 4u, 0, NT);
              if (nv >= 3) __builtin_amdgcn_raw_buffer_store_b32(ou[2], dst, so + 8u, 0, NT);
            }
          } else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dst, evoff[ct] + (unsigned)(8 * p * L * 4), 0, NTS);
          if constexpr (UBW) {                                  // h' + part_t(next layer) back into the patch, in place
#pragma clang fp contract(off)                                  // the sum of the ROUNDED h' and part_t, as the next layer's staging forms it: (add + v) * scale + pn as one fma
                                                                // is a different number (seen: the non-ragged instantiation contracted it, the ragged one did not)
            const float pn = reinterpret_cast<const float *>(lds + PNOFF)[32 * wave + (lane >> 3) + 8 * p];
            *reinterpret_cast<float4 *>(patch + ((lane >> 3) + 8 * p) * PS_ + 4 * (lane & 7)) = make_float4(o[0] + pn, o[1] + pn, o[2] + pn, o[3] + pn);
          }
        }
        if constexpr (UBW) {
          // column j of the patch, channels 16 hh .. 16 hh + 15 -> 32 bytes of ub'[clip][chunk = wave][t0 + 32 ct + j][.]
          asm volatile("" ::: "memory");                          // (the patch is written and read through different types: keep the order)
          if (ubout) {                                           // (uniform; null on the net's last layer)
            u32x4 lo4, hi4;
#pragma unroll
            for (int e = 0; e < 8; e++) {
              const f32x2 v2 = {patch[(16 * hh + 2 * e) * PS_ + j], patch[(16 * hh + 2 * e + 1) * PS_ + j]};
              const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
              if (e < 4) lo4[e] = pk; else hi4[e - 4] = pk;
            }
            const int t = t0 + 32 * ct + j;
            // (chunk step in the VGPR offset, soffset = 0: see the note on buffer stores above)
            const unsigned uo = t < L ? (unsigned)((wave * L + t) * 64 + hh * 32) : 0x80000000u;
            if constexpr (!(DBG & 256)) {
              __builtin_amdgcn_raw_buffer_store_b128(lo4, uors, uo, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(hi4, uors, uo + 16u, 0, 0);
            }
          }
          asm volatile("" ::: "memory");
        }
      }
    };

    if constexpr (!DS) {
      f32x16 ac[4];
      gemm2_loop(ac, 0);
      mark(28);
      fetch_bias(I1{});                                          // ahead of this pass's stores
      __builtin_amdgcn_sched_barrier(0);
      epilogue(ac, pre, ors, RS, std::true_type{});
      mark(29);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- next tile: parameters, and its first X chunk requested BEFORE the last stores of this tile
    int b_nxt = b_cur, t0_nxt = t0_cur;
    if (ntile < t_end) tile_bt(ntile, b_nxt, t0_nxt);
    // (unconditional from here: after its last tile a workgroup re-requests that tile's first chunk and drops it -- a
    // conditional request would keep the staging and fragment registers live across the whole tile)
    hrs = clip_rsrc(hin, b_nxt);
    if constexpr (UB) urs = ub_rsrc(ubin, b_nxt);
    x_geom(t0_nxt, xvoff, keep);
    issue_x(AP_XRS, xvoff, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DS) {
      // the ONE pass of this form (res_conv rows -> h'), in the place and with the request order the fused form gives its
      // last pass: the next tile's first chunk is in flight under the MFMAs, its pack and first fragments go out before the
      // stores, the stores drain behind the next tile's GEMM1
      if constexpr (NOH) {
        tile_head();
        __builtin_amdgcn_sched_barrier(0);
      } else {
        f32x16 ac[4];
        gemm2_loop(ac, 0);
        mark(30);
        tile_head();
        mark(31);
        __builtin_amdgcn_sched_barrier(0);
        epilogue(ac, pre, ors, RS, std::false_type{});
      }
      // g image -> HBM: 128 columns x 512 B, one 16-byte piece per lane and step (a wave moves two whole columns per step:
      // 1 KB contiguous on both sides; the padded 528-B LDS rows keep the 16-lane groups of ds_read_b128 on distinct banks)
      {
        int ln;                                                  // (lane id read here, not kept: see x_geom)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const uint64_t gb_ = (uint64_t)gout + (uint64_t)b_cur * ((uint64_t)L * 512u);
        const uint32_t glo = __builtin_amdgcn_readfirstlane((uint32_t)gb_);
        const uint32_t ghi = __builtin_amdgcn_readfirstlane((uint32_t)(gb_ >> 32));
        const __amdgpu_buffer_rsrc_t grs =
            __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)ghi << 32) | glo), 0, (int)((unsigned)L * 512u), 0x00020000);
        const int colw = 2 * wave + (ln >> 5), q = ln & 31;
        const unsigned char *src = lds + GOFF + colw * (GS_ * 2) + q * 16;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(src + 16 * i * (GS_ * 2));
          const int t = t0 + colw + 16 * i;
          const unsigned off = t < L ? (unsigned)t * 512u + (unsigned)q * 16u : 0x80000000u;   // outside the clip: dropped
          if constexpr (DBG & 256) asm volatile("" ::"v"(v));
          else __builtin_amdgcn_raw_buffer_store_b128(v, grs, off, 0, NTS);
        }
      }
      mark(32);
    } else {
      f32x16 ac[4];
      gemm2_loop(ac, 1);
      mark(30);
      // no barrier here: the X buffers have been free since the gate's barrier, the g image is next written by the next
      // tile's gate (eight barriers on), and the epilogue only touches this wave's own patch
      tile_head();
      mark(31);
      __builtin_amdgcn_sched_barrier(0);
      epilogue(ac, pre1, srs, 1.0f, std::false_type{});
      mark(32);
    }
    b_cur = b_nxt;
    t0_cur = t0_nxt;
  }
}

#ifdef AP_TOOLS
extern int g_dbg_bf16;
}  // namespace ap
extern "C" int ap_debug_ptrace(void *buf) {                       // device buffer of grid x 8 x 64 u64 (tools/trace_resblock_bf16p.py)
  unsigned long long *p = (unsigned long long *)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ap::g_ptrace), &p, sizeof(p));
}
namespace ap {
#endif

// -> 0 launched, 1 shape not served by this kernel (caller falls back to the per-tile kernel)
int launch_resblock_bf16p(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip, int accumulate,
                          int B, int L, hipStream_t st, const UbArgs *ub, void *gout, void *fout) {
  const int C = ctx->C, S = ctx->S;
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  if (C != 256 || S != 256 || L < 1) return 1;
  const bool rag = (L % 4) != 0;                                 // ragged clip length: the RAG instantiations
  // staging form: one window for the three taps where they overlap (d <= 32), else three tap loads
  const int ws = d == 1 ? 1 : d == 2 ? 2 : (d <= 32 && d % 4 == 0) ? 0 : d < 4 ? -2 : -1;
  if (ws == -2) return 1;                                        // (d = 3: no such dilation in a power-of-two cycle)
  const int n_cu = device_cu_count();                            // (the grid is one workgroup per CU)
  const int ntiles = (L + PT_ - 1) / PT_;
  const int nblk = B * ntiles;
  int grid = nblk < n_cu ? nblk : n_cu;
  if (grid >= 8) grid &= ~7;
  // one descriptor per parameter slab, the per-layer tensors addressed by byte offsets inside it
  const size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
  const char *wlo = (const char *)ctx->w1p_bf < (const char *)ctx->w2p_bf ? (const char *)ctx->w1p_bf : (const char *)ctx->w2p_bf;
  const unsigned w1_off = (unsigned)((const char *)ctx->w1p_bf - wlo + layer * n1 * 2);
  const unsigned w2_off = (unsigned)((const char *)ctx->w2p_bf - wlo + layer * n2 * 2);
  const unsigned wbytes = (unsigned)(((size_t)ctx->NL * (n1 + n2) + (size_t)S * S + (ctx->w1q_bf ? (size_t)ctx->NL * n1 : 0)) * 2);   // the whole bf16 slab
  const float *blo = ctx->b1 < ctx->b2 ? ctx->b1 : ctx->b2;
  const float *bhi = ctx->b1 < ctx->b2 ? ctx->b2 : ctx->b1;
  const unsigned b1_off = (unsigned)((ctx->b1 - blo + (size_t)layer * 2 * C) * 4);
  const unsigned b2_off = (unsigned)((ctx->b2 - blo + (size_t)layer * (C + S)) * 4);
  const unsigned bbytes = (unsigned)((bhi - blo + (size_t)ctx->NL * 2 * C) * 4);
#define AP_P_LAUNCH(D)                                                                                                        \
  resblock_bf16p_kernel<D><<<(unsigned)grid, 512, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, \
                                                           b2_off, L, d, accumulate, ntiles, nblk, nullptr, nullptr, nullptr, nullptr)
#define AP_P_LAUNCH_WIN(D, W)                                                                                                 \
  resblock_bf16p_kernel<D, W><<<(unsigned)grid, 512, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1_off, w2_off, blo, bbytes,  \
                                                              b1_off, b2_off, L, d, accumulate, ntiles, nblk, nullptr, nullptr, nullptr, nullptr)
#define AP_P_LAUNCH_RAG(W)                                                                                                    \
  resblock_bf16p_kernel<0, W, true><<<(unsigned)grid, 512, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1_off, w2_off, blo, bbytes, \
                                                                    b1_off, b2_off, L, d, accumulate, ntiles, nblk, nullptr, nullptr, nullptr, nullptr)
  if (gout) {                                                    // deferred-skip form: h' + the bf16 g image, no skip GEMM in the block
    if ((size_t)L * 512 >= ((size_t)1 << 31)) { set_error("AP_PREC_BF16: clip too long for the bf16 g image"); return -22; }
    if (fout && (size_t)ntiles * 131072 >= ((size_t)1 << 31)) { set_error("AP_PREC_BF16: clip too long for the gate-factor image"); return -22; }
#define AP_P_LAUNCH_DS(W, R)                                                                                                   \
  do {                                                                                                                         \
    if (fout && hout) /* the differentiable purifier's forward: + the gate's derivative factors */                             \
      resblock_bf16p_kernel<0, W, R, false, false, true, false, true><<<(unsigned)grid, 512, 0, st>>>(                         \
          hin, pt, hout, nullptr, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off, L, d, 0, ntiles, nblk, nullptr, nullptr, nullptr, gout, fout); \
    else if (fout)                                                                                                             \
      resblock_bf16p_kernel<0, W, R, false, false, true, true, true><<<(unsigned)grid, 512, 0, st>>>(                          \
          hin, pt, nullptr, nullptr, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off, L, d, 0, ntiles, nblk, nullptr, nullptr, nullptr, gout, fout); \
    else if (hout)                                                                                                             \
      resblock_bf16p_kernel<0, W, R, false, false, true><<<(unsigned)grid, 512, 0, st>>>(                                      \
          hin, pt, hout, nullptr, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off, L, d, 0, ntiles, nblk, nullptr, nullptr, nullptr, gout); \
    else /* h' not wanted (the net's last layer) */                                                                            \
      resblock_bf16p_kernel<0, W, R, false, false, true, true><<<(unsigned)grid, 512, 0, st>>>(                                \
          hin, pt, nullptr, nullptr, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off, L, d, 0, ntiles, nblk, nullptr, nullptr, nullptr, gout); \
  } while (0)
    if (rag) {
      if (ws == 1) AP_P_LAUNCH_DS(1, true);
      else if (ws == 2) AP_P_LAUNCH_DS(2, true);
      else AP_P_LAUNCH_DS(-1, true);
    } else if (ws == 0) AP_P_LAUNCH_DS(0, false);
    else if (ws == 1) AP_P_LAUNCH_DS(1, false);
    else if (ws == 2) AP_P_LAUNCH_DS(2, false);
    else AP_P_LAUNCH_DS(-1, false);
#undef AP_P_LAUNCH_DS
    AP_HIP(hipGetLastError());
    return 0;
  }
#ifdef AP_TOOLS
  if (ub) {                                                      // the operand-image experiment: images in / out, one staging form for every d
    if ((size_t)C * (size_t)L * 2 >= ((size_t)1 << 31)) { set_error("AP_PREC_BF16: clip too long for the bf16 operand image"); return -22; }
    if (rag)
      resblock_bf16p_kernel<0, -1, true, false, true><<<(unsigned)grid, 512, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1_off, w2_off, blo, bbytes,
                                                                                      b1_off, b2_off, L, d, accumulate, ntiles, nblk, ub->in, ub->out, ub->pt_next, nullptr);
    else
      resblock_bf16p_kernel<0, -1, false, false, true><<<(unsigned)grid, 512, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1_off, w2_off, blo, bbytes,
                                                                                       b1_off, b2_off, L, d, accumulate, ntiles, nblk, ub->in, ub->out, ub->pt_next, nullptr);
    AP_HIP(hipGetLastError());
    return 0;
  }
  if (rag) {
    if (ws == 1) AP_P_LAUNCH_RAG(1);
    else if (ws == 2) AP_P_LAUNCH_RAG(2);
    else AP_P_LAUNCH_RAG(-1);
  } else
  if (ws >= 0 && !(g_dbg_bf16 & 0x10000)) {                      // tools bit 0x10000: three-tap staging for every d (A/B)
    if (ws == 0) AP_P_LAUNCH_WIN(0, 0);
    else if (ws == 1) AP_P_LAUNCH_WIN(0, 1);
    else AP_P_LAUNCH_WIN(0, 2);
  } else if (ws > 0) {
    return 1;                                                    // d = 1, 2 without the window: the per-tile kernel
  } else
  if ((g_dbg_bf16 & 0x200000) && ctx->w1q_bf) {                  // tools bit 0x200000: GEMM1 on v_mfma_f32_16x16x32_bf16 (exact; A/B)
    const unsigned w1q_off = (unsigned)((const char *)ctx->w1q_bf - wlo + layer * n1 * 2);
    resblock_bf16p_kernel<0, -1, false, true><<<(unsigned)grid, 512, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1q_off, w2_off, blo,
                                                                            bbytes, b1_off, b2_off, L, d, accumulate, ntiles, nblk, nullptr, nullptr, nullptr, nullptr);
  } else
  if (g_dbg_bf16 & 0x100000) AP_P_LAUNCH(0x100000);             // timing only: v_mfma_f32_16x16x32_bf16 pairs in place of 32x32x16
  else if ((g_dbg_bf16 & 0x7000000) == 0x1000000) AP_P_LAUNCH(0x1000000);   // store cache policies (exact)
  else if ((g_dbg_bf16 & 0x7000000) == 0x2000000) AP_P_LAUNCH(0x2000000);
  else if ((g_dbg_bf16 & 0x7000000) == 0x3000000) AP_P_LAUNCH(0x3000000);
  else if ((g_dbg_bf16 & 0x7000000) == 0x5000000) AP_P_LAUNCH(0x5000000);
  else
  switch (g_dbg_bf16 & 0x800efff) {
    case 0: AP_P_LAUNCH(0); break;
    case 1: AP_P_LAUNCH(1); break;
    case 2: AP_P_LAUNCH(2); break;
    case 3: AP_P_LAUNCH(3); break;
    case 4: AP_P_LAUNCH(4); break;
    case 7: AP_P_LAUNCH(7); break;
    case 8: AP_P_LAUNCH(8); break;
    case 16: AP_P_LAUNCH(16); break;
    case 23: AP_P_LAUNCH(23); break;
    case 31: AP_P_LAUNCH(31); break;
    case 31 + 128: AP_P_LAUNCH(31 + 128); break;
    case 31 + 256: AP_P_LAUNCH(31 + 256); break;
    case 31 + 384: AP_P_LAUNCH(31 + 384); break;
    case 128: AP_P_LAUNCH(128); break;
    case 256: AP_P_LAUNCH(256); break;
    case 384: AP_P_LAUNCH(384); break;
    case 384 + 1: AP_P_LAUNCH(384 + 1); break;
    case 384 + 2: AP_P_LAUNCH(384 + 2); break;
    case 384 + 3: AP_P_LAUNCH(384 + 3); break;
    case 384 + 8: AP_P_LAUNCH(384 + 8); break;
    case 384 + 19: AP_P_LAUNCH(384 + 19); break;
    case 384 + 23: AP_P_LAUNCH(384 + 23); break;

    case 384 + 31 + 64: AP_P_LAUNCH(384 + 31 + 64); break;
    case 384 + 31 + 96: AP_P_LAUNCH(384 + 31 + 96); break;
    case 512: AP_P_LAUNCH(512); break;
    case 512 + 384: AP_P_LAUNCH(512 + 384); break;
    case 1024: AP_P_LAUNCH(1024); break;
    case 2048: AP_P_LAUNCH(2048); break;
    case 2048 + 384: AP_P_LAUNCH(2048 + 384); break;
    case 2048 + 0x8000000: AP_P_LAUNCH(2048 + 0x8000000); break;
    case 1024 + 384: AP_P_LAUNCH(1024 + 384); break;
    case 0x2000: AP_P_LAUNCH(0x2000); break;
    case 0x4000: AP_P_LAUNCH(0x4000); break;
    case 0x8000: AP_P_LAUNCH(0x8000); break;
    case 32: AP_P_LAUNCH(32); break;
    case 64: AP_P_LAUNCH(64); break;
    case 96: AP_P_LAUNCH(96); break;
    case 32 + 384: AP_P_LAUNCH(32 + 384); break;
    case 96 + 384: AP_P_LAUNCH(96 + 384); break;
    default: set_error("no such DBG instantiation"); return -22;
  }
#else
  if (rag) {                                                     // (d = 4 .. 32 ragged: the three-tap form, one instantiation fewer)
    if (ws == 1) AP_P_LAUNCH_RAG(1);
    else if (ws == 2) AP_P_LAUNCH_RAG(2);
    else AP_P_LAUNCH_RAG(-1);
  } else if (ws == 0) AP_P_LAUNCH_WIN(0, 0);
  else if (ws == 1) AP_P_LAUNCH_WIN(0, 1);
  else if (ws == 2) AP_P_LAUNCH_WIN(0, 2);
  else AP_P_LAUNCH(0);
#endif
#undef AP_P_LAUNCH
#undef AP_P_LAUNCH_WIN
#undef AP_P_LAUNCH_RAG
  AP_HIP(hipGetLastError());
  return 0;
}

#ifndef AP_TOOLS
// AP_PREC_BF16 dispatch of the product library: the persistent kernel serves every shape of the supported configuration
// (C = S = 256; any dilation of a power-of-two cycle, any clip length).  The one-tile-per-workgroup kernel of round 1
// (tools/csrc/ap_resblock_bf16.hip) is compiled into the tools library only, as the A/B baseline of tools/cmp_bf16_kernels.py.
int launch_resblock_bf16(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip, int accumulate,
                         int B, int L, hipStream_t st, const UbArgs *ub, void *gout, void *fout) {
  if (ctx->C != 256 || ctx->S != 256) {
    set_error("AP_PREC_BF16 is built for res_channels = skip_channels = 256 only (got %d / %d)", ctx->C, ctx->S);
    return -22;
  }
  const int rc = launch_resblock_bf16p(ctx, layer, hin, pt, hout, skip, accumulate, B, L, st, ub, gout, fout);
  if (rc == 1) {
    set_error("AP_PREC_BF16: shape not served (layer %d, L = %d)", layer, L);
    return -22;
  }
  return rc;
}
#endif

}  // namespace ap
