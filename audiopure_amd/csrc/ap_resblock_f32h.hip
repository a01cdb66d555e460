// AP_PREC_F32_SPLIT_F16 ("f32h"): fused Residual_block.forward (WaveNet.py:75-97) with every fp32 operand carried as
// TWO fp16 parts on v_mfma_f32_32x32x16_f16, fp32 accumulate.
//
// x = x0 + x1 with x0 = rne16(x), x1 = rne16(x - x0): 2 x 11 significant bits, |x - x0 - x1| <= 2^-22 |x| (fp32 itself
// keeps 2^-24), and x*w is summed from the three partial products x0w0, x0w1, x1w0 (the dropped x1w1 is < 2^-22 of the
// product).  Three fp16 MFMAs = 3/16 of the fp32 instruction's time, half of the 3-way split of
// ap_resblock_f32s.hip, and half as many accumulations: measured dot-product noise (K = 768) 2.4e-7 rms vs 1.3e-7 for
// plain fp32 and 3.2e-7 for the 3-way split.  fp16's narrow exponent is handled with exact power-of-two scales: weights
// x 2^4, activations x 2^4, gate outputs x 2^8 (biases and the epilogue carry the inverse; every scale is exact, so the
// only difference from fp32 arithmetic is the 22-bit operand representation).  A scaled activation is clamped to
// +-60000 (|h + part_t| < 3750) and residual parts below fp16's normal range (operands under 2^-7 of the typical
// magnitude) lose bits gracefully: absolute error <= 2^-29 of an O(1) activation.
//
// Structure = ap_resblock_f32s.hip with two images per operand.
#include <type_traits>

#include "ap_common.h"

namespace ap {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace f32h {
constexpr int BT = 128;                  // time tile
constexpr int BKC = 16;                  // channels per staged chunk -> 48 K rows = 3 k-steps (one per tap)
constexpr int XS = 3 * BKC + 8;          // fp16 per column row of an X image (112 B: conflict-free b128 reads)
constexpr int HT = 64;                   // columns per gate/GEMM2 half
constexpr int PSTR = 32;                 // fp32 row stride of the wave-private output patch (128-B rows: conflict-free for the column writes and the 16-lane groups of the b128 row reads; 144-B rows were 2-way)
constexpr float WSC = 16.0f;             // weight scale
constexpr float XSC = 16.0f;             // activation scale (GEMM1 B operand)
constexpr float GSC = 256.0f;            // gate-output scale (GEMM2 B operand)

__device__ __forceinline__ int rowoff_s(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// x ~ p[0] + p[1], both fp16 (RNE); the residual x - p[0] is exact in fp32
__device__ __forceinline__ void split2(float x, _Float16 (&p)[2]) {
  p[0] = (_Float16)x;
  p[1] = (_Float16)(x - (float)p[0]);
}

// four values at once -> two packed words per split
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2x4(const float (&x)[4], u32x2s (&out)[2]) {
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {
    const f32x2 v = {x[2 * pr], x[2 * pr + 1]};
    const h16x2 hi = __builtin_convertvector(v, h16x2);
    const f32x2 up = __builtin_convertvector(hi, f32x2);
    const h16x2 lo = __builtin_convertvector(v - up, h16x2);
    out[0][pr] = __builtin_bit_cast(unsigned, hi);
    out[1][pr] = __builtin_bit_cast(unsigned, lo);
  }
}

// same compensated exp and gate as the fp32 kernel (ap_kernels.hip) -- the gate is not where the two modes differ
__device__ __forceinline__ float exp_acc_s(float x) {
  const float L2E_HI = 1.44269502162933349609375f;
  const float L2E_LO = 1.92596299e-8f;
  float t = x * L2E_HI;
  float r = __builtin_fmaf(x, L2E_HI, -t);
  r = __builtin_fmaf(x, L2E_LO, r);
  float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, r * 0.693147182464599609375f, e);
}
__device__ __forceinline__ float gate_s(float a, float b) {      // a, b carry XSC * WSC; the result carries GSC
  a *= 1.0f / (XSC * WSC);
  b *= 1.0f / (XSC * WSC);
  a = fminf(fmaxf(a, -15.0f), 15.0f);
  b = fmaxf(b, -80.0f);
  float E = exp_acc_s(2.0f * a);
  float F = exp_acc_s(-b);
  return (E - 1.0f) * GSC * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
}
}  // namespace f32h
using namespace f32h;

// ---- weight images -------------------------------------------------------------------------------------------
// GEMM1: [wave C/32][chunk C/16][kstep 3 = tap][rowtile 2][split 2][lane 64][8]; wave w owns gate channels
// [32w, 32w+32): row tile 0 = tanh rows, 1 = sigmoid rows; k-step = tap, channels ch*16 + 8h + jj.
__global__ void pack_w1_splith_kernel(const float *__restrict__ w1f, _Float16 *__restrict__ out, int C) {
  const int NW = C / 32, NCH = C / BKC;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (.., lane, jj), both splits
  size_t total = (size_t)NW * NCH * 3 * 2 * 64 * 8;
  if (idx >= total) return;
  int jj = idx & 7;
  int lane = (idx >> 3) & 63;
  int rt = (idx >> 9) & 1;
  size_t rest = idx >> 10;
  int ks = rest % 3; rest /= 3;
  int ch = rest % NCH;
  int w = rest / NCH;
  int i = lane & 31, hh = lane >> 5;
  int c = ch * BKC + 8 * hh + jj;
  int o = rt * C + 32 * w + i;
  _Float16 p[2];
  split2(fminf(fmaxf(w1f[((size_t)o * C + c) * 3 + ks] * WSC, -60000.0f), 60000.0f), p);   // |w| < 3750, like the activations
  size_t frag = ((((size_t)w * NCH + ch) * 3 + ks) * 2 + rt) * 2;
#pragma unroll
  for (int s = 0; s < 2; s++) out[((frag + s) * 64 + lane) * 8 + jj] = p[s];
}

// GEMM2: [wave][pass 2][kstep C/16][split 2][lane][8]; pass 0 = res_conv rows of the wave's channels, 1 = skip rows.
__global__ void pack_w2_splith_kernel(const float *__restrict__ w2f, _Float16 *__restrict__ out, int C) {
  const int NW = C / 32, NKS = C / 16;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NW * 2 * NKS * 64 * 8;
  if (idx >= total) return;
  int jj = idx & 7;
  int lane = (idx >> 3) & 63;
  size_t rest = idx >> 9;
  int ks = rest % NKS; rest /= NKS;
  int pass = rest & 1;
  int w = rest >> 1;
  int i = lane & 31, hh = lane >> 5;
  int k = ks * 16 + 8 * hh + jj;
  int o = pass * C + 32 * w + i;
  _Float16 p[2];
  split2(fminf(fmaxf(w2f[(size_t)o * C + k] * WSC, -60000.0f), 60000.0f), p);
  size_t frag = (((size_t)w * 2 + pass) * NKS + ks) * 2;
#pragma unroll
  for (int s = 0; s < 2; s++) out[((frag + s) * 64 + lane) * 8 + jj] = p[s];
}

int launch_pack_splith(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C, S = ctx->S, NL = ctx->NL;
  for (int n = 0; n < NL; n++) {
    size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
    pack_w1_splith_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1, (_Float16 *)ctx->w1p_h + n * n1 * 2, C);
    pack_w2_splith_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, (_Float16 *)ctx->w2p_h + n * n2 * 2, C);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

// ---- the kernel ---------------------------------------------------------------------------------------------
// the three partial products kept, as (weight split, activation split)
#define AP_SPLIT_TERMS(F) F(0, 0) F(0, 1) F(1, 0)

template <int C, bool E4, bool TRACE>
__global__ __launch_bounds__(C / 32 * 64, 2) void resblock_f32h_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const _Float16 *__restrict__ w1p, const float *__restrict__ b1, const _Float16 *__restrict__ w2p,
    const float *__restrict__ b2, int L, int d, int accumulate, int ntiles, int nblk,
    unsigned long long *__restrict__ trace) {
  constexpr int NW = C / 32, NT = NW * 64, NCH = C / BKC;
  static_assert(NT == 512 && NCH % 2 == 0, "built for C = 256 (8 waves)");
  constexpr int GS = C + 8;                                    // fp16 per column row of a g image (528 B)
  constexpr int XIMG = BT * XS * 2;                            // 14,336 B per X image
  constexpr int XBUF = 2 * XIMG;                               // two splits per buffer
  constexpr int GIMG = BT * GS * 2;                            // 67,584 B per g image (all 128 columns)
  constexpr int UNION = (2 * XBUF > 2 * GIMG) ? 2 * XBUF : 2 * GIMG;
  constexpr int PTOFF = UNION;                                 // part_t (C floats)
  constexpr int LDS_BYTES = PTOFF + C * 4;
  constexpr int PATCH_FLOATS = E4 ? NW * 16 * PSTR : 4;        // 16 rows per wave: the epilogue goes half a row tile at a time
  static_assert(LDS_BYTES + PATCH_FLOATS * 4 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  __shared__ __attribute__((aligned(16))) float patch_mem[PATCH_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  auto mark = [&](int i) {                                     // phase stamps (tools/trace_resblock_bf16.py); trace builds only
    if constexpr (TRACE) {
      if (lane == 0) trace[((size_t)blockIdx.x * NW + wave) * 16 + i] = __builtin_readcyclecounter();
    }
  };
  mark(0);
  int b_, tile_;                                               // XCD-local walk (ap_common.h; speed only)
  ap_tile_of_block(blockIdx.x, nblk, ntiles, d, BT, b_, tile_);
  const int b = __builtin_amdgcn_readfirstlane(b_);
  const int t0 = __builtin_amdgcn_readfirstlane(tile_ * BT);
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  auto clip_rsrc = [&](const float *base) {
    const uint64_t hb = (uint64_t)(base + (size_t)b * C * L);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)clip_bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t hrs = clip_rsrc(hin);

  f32x16 acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 bv = *reinterpret_cast<const float4 *>(b1 + rt * C + 32 * wave + 8 * q + 4 * hh);
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        acc[rt][ct][4 * q + 0] = bv.x * (XSC * WSC);
        acc[rt][ct][4 * q + 1] = bv.y * (XSC * WSC);
        acc[rt][ct][4 * q + 2] = bv.z * (XSC * WSC);
        acc[rt][ct][4 * q + 3] = bv.w * (XSC * WSC);
      }
    }

  // ---- X staging: thread = (column tid&127, channel quad tid>>7) for each of the 3 taps: 12 buffer loads at the head
  // of a chunk; FiLM add (WaveNet.py:84), zero padding (:26-27), 2-way split and two ds_write_b64 per tap at its tail.
  if (tid < C) reinterpret_cast<float *>(lds + PTOFF)[tid] = pt[tid] * XSC;       // pre-scaled: u XSC = fma(h, XSC, pt XSC)
  const int col = tid & (BT - 1), q4 = (tid >> 7) * 4;
  unsigned voff[3];
  bool tok[3];
#pragma unroll
  for (int tap = 0; tap < 3; tap++) {
    const int tp = t0 + col + (tap - 1) * d;
    tok[tap] = (tp >= 0) && (tp < L);
    voff[tap] = ((unsigned)min(max(tp, 0), L - 1) + (unsigned)q4 * (unsigned)L) * 4u;
  }
  float xr[3][4];
  auto issue_loads = [&](int ch) {
#pragma unroll
    for (int tap = 0; tap < 3; tap++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        xr[tap][e] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(hrs, voff[tap], (ch * BKC + e) * L * 4, 0));
  };
  auto store_chunk = [&](unsigned char *dst, int ch) {
    const float4 pv = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(lds + PTOFF) + ch * BKC + q4);
    const float pte[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
    for (int tap = 0; tap < 3; tap++) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; e++)
        u[e] = tok[tap] ? __builtin_amdgcn_fmed3f(__builtin_fmaf(xr[tap][e], XSC, pte[e]), -60000.0f, 60000.0f) : 0.f;
      u32x2s pk[2];
      split2x4(u, pk);
#pragma unroll
      for (int s = 0; s < 2; s++)
        *reinterpret_cast<u32x2s *>(dst + s * XIMG + (col * XS + tap * BKC + q4) * 2) = pk[s];
    }
  };

  issue_loads(0);
  __syncthreads();                                              // part_t visible
  store_chunk(lds, 0);
  __syncthreads();
  mark(1);

  // ---- GEMM1: 48 k-steps (16 chunks x 3 taps), 24 MFMAs each: 2 row tiles x 4 column tiles x 3 partial products.
  // Weight fragments (2 row tiles x 2 splits = 16 VGPRs per k-step) stream from L2 one k-step ahead, ping-pong.
  auto load_a = [&](h16x8(&a)[2][2], const u32x4 *base) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int s = 0; s < 2; s++) a[rt][s] = __builtin_bit_cast(h16x8, base[(rt * 2 + s) * 64]);
  };
  const int rdoff = (j * XS + 8 * hh) * 2;                      // this lane's B-fragment byte offset inside an X image
  // bpre holds the first column tile's two B fragments of the k-step about to run; inside a chunk the next k-step's
  // are fetched under this one's last MFMAs, so a k-step boundary does not wait on LDS (across the chunk barrier the
  // other buffer is not valid yet: NEXT = false there and the caller refills bpre after the barrier)
  h16x8 bpre[2];
  auto read_b = [&](h16x8(&bv)[2], const unsigned char *xb, int ct) {
#pragma unroll
    for (int s = 0; s < 2; s++) bv[s] = *reinterpret_cast<const h16x8 *>(xb + s * XIMG + (32 * ct) * (XS * 2));
  };
  auto mma_k = [&](const h16x8(&a)[2][2], const unsigned char *xb, auto NEXT) {   // xb: buffer + rdoff + tap * 32
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      h16x8 bv[2];
      if (ct == 0) {
#pragma unroll
        for (int s = 0; s < 2; s++) bv[s] = bpre[s];
      } else {
        read_b(bv, xb, ct);
      }
      if (ct == 3 && decltype(NEXT)::value) read_b(bpre, xb + 32, 0);
#pragma unroll
      for (int rt = 0; rt < 2; rt++) {
#define AP_T(i, jx) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt][i], bv[jx], acc[rt][ct], 0, 0, 0);
        AP_SPLIT_TERMS(AP_T)
#undef AP_T
      }
    }
  };
  using YES = std::true_type;
  using NO = std::false_type;
  const u32x4 *ap = reinterpret_cast<const u32x4 *>(w1p) + (size_t)wave * NCH * 3 * 4 * 64 + lane;
  auto aset = [&](int kk) { return ap + (size_t)(kk < NCH * 3 ? kk : NCH * 3 - 1) * 4 * 64; };
  // the next chunk's FiLM add / split / pack (VALU) goes into the gaps of the third k-step's 24 MFMAs
  auto pack_between_mfmas = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int i = 0; i < 24; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      if (i % 6 == 1 && i < 18) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
  };
  // Three fragment sets = the three k-steps of a chunk; each is refilled for the next chunk right after its k-step.
  // vmcnt retires in issue order, so the (HBM-latency) X loads go first in a chunk: the next fragment set that was
  // requested after them is not needed until the next chunk starts, by which time the pack has consumed X anyway.
  h16x8 a0[2][2], a1[2][2], a2[2][2];
  load_a(a0, aset(0));
  load_a(a1, aset(1));
  load_a(a2, aset(2));
#pragma unroll 1
  for (int it = 0; it < NCH / 2; it++) {
    const int c0 = 2 * it, kk = 6 * it;
    const unsigned char *x0 = lds + rdoff, *x1 = lds + XBUF + rdoff;
    if (it == 4) mark(15);
    read_b(bpre, x0, 0);
    issue_loads(c0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a0, x0, YES{});
    __builtin_amdgcn_sched_barrier(0);
    if (it == 4) mark(7);
    load_a(a0, aset(kk + 3));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a1, x0 + 32, YES{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(a1, aset(kk + 4));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a2, x0 + 64, NO{});
    store_chunk(lds + XBUF, c0 + 1);
    pack_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    load_a(a2, aset(kk + 5));
    if (it == 4) mark(8);
    __syncthreads();
    if (it == 4) mark(9);
    if (it == 3) mark(3);
    read_b(bpre, x1, 0);
    issue_loads(c0 + 2 < NCH ? c0 + 2 : NCH - 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a0, x1, YES{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(a0, aset(kk + 6));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a1, x1 + 32, YES{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(a1, aset(kk + 7));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a2, x1 + 64, NO{});
    store_chunk(lds, c0 + 2 < NCH ? c0 + 2 : NCH - 1);          // after the last chunk: a harmless re-store
    pack_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    load_a(a2, aset(kk + 8));
    __syncthreads();
    if (it == 0) mark(2);
  }
  mark(4);

  // ---- gate (WaveNet.py:90) of all 128 columns -> two g images [col][channel] (they take over the X buffers' LDS);
  // GEMM2 in two passes of 32 rows x 128 columns (pass 0 = res_conv rows -> h', pass 1 = skip_conv rows -> skip;
  // WaveNet.py:93-97, :133).  Full-width passes stream W2 once per tile: with 64-column halves the 1 MB/tile of A
  // fragments through the CU's 64 B/clk L2 port cost as many cycles as the MFMAs.
  constexpr int NKS = C / 16;
  const float RS = 0.707106781186547524f;
  const __amdgpu_buffer_rsrc_t srs = clip_rsrc(skip);
  const __amdgpu_buffer_rsrc_t ors = clip_rsrc(hout);
  float *patch = patch_mem + (E4 ? wave * 16 * PSTR : 0);
  const unsigned char *gb = lds + (j * GS + 8 * hh) * 2;
  const u32x4 *ap2 = reinterpret_cast<const u32x4 *>(w2p) + (size_t)(wave * 2) * NKS * 2 * 64 + lane;
  const float *b2l = b2, *ptl = pt;
  asm volatile("" : "+s"(b2l), "+s"(ptl));

  // Order of the global traffic (the CU's memory pipe serves requests in order, and vmcnt retires in order):
  //   bias vectors and the first four k-steps of weight fragments are requested BEFORE anything with HBM latency or
  //   bulk (the values a pass adds into, the previous pass's stores), so a pass starts its MFMAs at once;
  //   pass 0's k-loop runs its fragment prefetch straight into pass 1's first k-steps (the two blocks are adjacent in
  //   the packed image), and pass 1's bias is fetched before pass 0's epilogue queues its 16 stores per wave.
  h16x8 p0[2][2], p1[2][2];
  auto load_a2 = [&](h16x8(&a)[2][2], int gks) {                 // gks = k-step over both passes, 0 .. 2 NKS - 1
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int s = 0; s < 2; s++)
        a[u][s] = __builtin_bit_cast(h16x8, ap2[(size_t)(((gks + u < 2 * NKS ? gks + u : 2 * NKS - 1) * 2 + s) * 64)]);
  };
  float4 bias[4];
  auto fetch_bias = [&](auto PTAG) {
    constexpr int pass = decltype(PTAG)::value;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = 32 * wave + 8 * q + 4 * hh;
      float4 v = *reinterpret_cast<const float4 *>(b2l + pass * C + c);
      if (pass == 0) {                                           // u = h + part_t re-enters the residual
        const float4 pv = *reinterpret_cast<const float4 *>(ptl + c);
        v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
      }
      bias[q] = v;
    }
  };
  fetch_bias(std::integral_constant<int, 0>{});                  // in flight under the gate
  load_a2(p0, 0);
  load_a2(p1, 2);
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      float gv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) gv[e] = gate_s(acc[0][ct][4 * qq + e], acc[1][ct][4 * qq + e]);
      u32x2s pk[2];
      split2x4(gv, pk);
#pragma unroll
      for (int s = 0; s < 2; s++)
        *reinterpret_cast<u32x2s *>(lds + s * GIMG + ((32 * ct + j) * GS + 32 * wave + 8 * qq + 4 * hh) * 2) = pk[s];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  mark(5);

  unsigned evoff[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    if constexpr (E4) {
      const int t = t0 + 32 * ct + 4 * (lane & 7);
      evoff[ct] = t < L ? ((unsigned)(32 * wave + (lane >> 3)) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
    } else {
      const int t = t0 + 32 * ct + j;
      evoff[ct] = t < L ? ((unsigned)(32 * wave + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
    }
  }                                                             // 0x80000000: outside the clip -> load 0 / store dropped
  auto gemm2_pass = [&](auto PTAG) {
    constexpr int pass = decltype(PTAG)::value;
    // what this pass adds into (h for the residual, the running skip) is fetched before its GEMM, used after it
    float pre[4][16];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      if constexpr (E4) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        pass == 0 ? hrs : srs, evoff[ct] + (unsigned)(8 * p * L * 4), 0, 2));   // nt: once-touched skip rows; the residual re-read of h hits or passes without allocating
#pragma unroll
          for (int i = 0; i < 4; i++) pre[ct][4 * p + i] = v[i];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++)
          pre[ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     pass == 0 ? hrs : srs, evoff[ct], ((r & 3) + 8 * (r >> 2)) * L * 4, 2));   // nt: once-touched skip rows; the residual re-read of h hits or passes without allocating
      }
    }
    f32x16 ac[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 v = bias[q];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {                           // the accumulator carries GSC WSC
        ac[ct][4 * q + 0] = v.x * (GSC * WSC);
        ac[ct][4 * q + 1] = v.y * (GSC * WSC);
        ac[ct][4 * q + 2] = v.z * (GSC * WSC);
        ac[ct][4 * q + 3] = v.w * (GSC * WSC);
      }
    }
    if (pass == 0) mark(10);
    // B fragments (two column tiles x both splits = one 64-column half of a k-step) are fetched one step ahead, under
    // the previous step's six MFMAs: a step boundary does not wait on LDS.  step = 2 ks + column half.
    auto read_b2 = [&](h16x8(&bq)[2][2], int step) {
      const int ks = (step >> 1) < NKS ? (step >> 1) : NKS - 1, hc = step & 1;
#pragma unroll
      for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
        for (int s = 0; s < 2; s++)
          bq[c2][s] = *reinterpret_cast<const h16x8 *>(gb + s * GIMG + (64 * hc + 32 * c2) * (GS * 2) + ks * 32);
    };
    auto mma2 = [&](const h16x8(&a)[2], const h16x8(&bq)[2][2], auto HC) {
      constexpr int hc = decltype(HC)::value;
#pragma unroll
      for (int c2 = 0; c2 < 2; c2++) {
#define AP_T(i, jx) ac[2 * hc + c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bq[c2][jx], ac[2 * hc + c2], 0, 0, 0);
        AP_SPLIT_TERMS(AP_T)
#undef AP_T
      }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    h16x8 q0[2][2], q1[2][2];
    read_b2(q0, 0);
    auto kstep = [&](const h16x8(&a)[2], int ks) {               // q0 holds (ks, half 0) on entry and (ks + 1, half 0) on exit
      read_b2(q1, 2 * ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma2(a, q0, H0{});
      __builtin_amdgcn_sched_barrier(0);
      read_b2(q0, 2 * ks + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma2(a, q1, H1{});
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 4) {
      kstep(p0[0], ks);
      kstep(p0[1], ks + 1);
      if (pass == 0 && ks == 0) mark(11);
      load_a2(p0, pass * NKS + ks + 4);
      kstep(p1[0], ks + 2);
      kstep(p1[1], ks + 3);
      load_a2(p1, pass * NKS + ks + 6);
    }
    if (pass == 0) {
      mark(12);
      fetch_bias(std::integral_constant<int, 1>{});              // ahead of this pass's stores
      __builtin_amdgcn_sched_barrier(0);
    }
    // exact power-of-two rescaling folded into the two epilogue constants: (pre GW + acc) (scale / GW)
    const float addm = (pass == 0 || accumulate) ? GSC * WSC : 0.0f;
    const float scale = (pass == 0 ? RS : 1.0f) * (1.0f / (GSC * WSC));
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      if constexpr (E4) {
        // the wave-private patch holds 16 rows: the tile's rows 0-15 (registers 0-7), then rows 16-31 (registers 8-15)
#pragma unroll
        for (int g = 0; g < 2; g++) {
#pragma unroll
          for (int r = 0; r < 8; r++) patch[rowoff_s(r, hh) * PSTR + j] = ac[ct][8 * g + r];
#pragma unroll
          for (int pp = 0; pp < 2; pp++) {
            const int p = 2 * g + pp;
            const float4 v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * pp) * PSTR + 4 * (lane & 7));
            f32x4 o;
            o[0] = __builtin_fmaf(pre[ct][4 * p + 0], addm, v.x) * scale;
            o[1] = __builtin_fmaf(pre[ct][4 * p + 1], addm, v.y) * scale;
            o[2] = __builtin_fmaf(pre[ct][4 * p + 2], addm, v.z) * scale;
            o[3] = __builtin_fmaf(pre[ct][4 * p + 3], addm, v.w) * scale;
            // offset in the VGPR, soffset = 0 (a >8-byte buffer store with an SGPR soffset reads its data late)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), pass == 0 ? ors : srs,
                                                   evoff[ct] + (unsigned)(8 * p * L * 4), 0, 2);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++)
          __builtin_amdgcn_raw_buffer_store_b32(
              __builtin_bit_cast(unsigned, __builtin_fmaf(pre[ct][r], addm, ac[ct][r]) * scale), pass == 0 ? ors : srs,
              evoff[ct], ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
      }
    }
  };
  gemm2_pass(std::integral_constant<int, 0>{});
  mark(13);
  mark(6);
  __builtin_amdgcn_sched_barrier(0);
  gemm2_pass(std::integral_constant<int, 1>{});
  mark(14);
}

#ifdef AP_TOOLS
extern unsigned long long *g_trace_bf16;
#endif

int launch_resblock_splith(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                          int accumulate, int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  if (C != 256 || S != 256) {
    set_error("AP_PREC_F32_SPLIT_F16 is built for res_channels = skip_channels = 256 only (got %d, %d)", C, S);
    return -22;
  }
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int ntiles = (L + BT - 1) / BT;
  const int nblk = B * ntiles;
  const _Float16 *w1p = (const _Float16 *)ctx->w1p_h + (size_t)layer * 2 * C * C * 3 * 2;
  const _Float16 *w2p = (const _Float16 *)ctx->w2p_h + (size_t)layer * (C + S) * C * 2;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C;
  const float *b2 = ctx->b2 + (size_t)layer * (C + S);
#ifdef AP_TOOLS
  if (L % 4 == 0 && L >= 4 && g_trace_bf16)
    resblock_f32h_kernel<256, true, true><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d,
                                                                         accumulate, ntiles, nblk, g_trace_bf16);
  else
#endif
  if (L % 4 == 0 && L >= 4)
    resblock_f32h_kernel<256, true, false><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d,
                                                                          accumulate, ntiles, nblk, nullptr);
  else
    resblock_f32h_kernel<256, false, false><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d,
                                                                           accumulate, ntiles, nblk, nullptr);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
