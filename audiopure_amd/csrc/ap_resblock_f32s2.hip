// AP_PREC_F32_SPLIT with the dilated conv in F(2,3) minimal-filtering form: Residual_block.forward
// (WaveNet.py:75-97) at fp32 accuracy on the bf16 matrix pipe -- operands split exactly into three bf16 parts, the six partial
// products >= 2^-16 of each product on v_mfma_f32_32x32x16_bf16 (ap_resblock_f32s.hip) -- with GEMM1 computed over the dilation
// pair as in ap_resblock_f32w.hip: outputs t and t + d share the taps u[t-d], u[t], u[t+d], u[t+2d], so the pair is four
// [2C x C] products instead of six:
//     m1 = W0 (u0 - u2)    m2 = (W0+W1+W2)/2 (u1 + u2)    m3 = (W0-W1+W2)/2 (u2 - u1)    m4 = W2 (u3 - u1)
//     y[t] = (m1 + m2) + m3 + b        y[t+d] = (m2 - m3) + m4 + b
// Transformed weights are computed in double, rounded to fp32 once, THEN split; the input differences are formed in fp32 (one
// more rounding than the direct form, as in the fp32 F(2,3) block), then split.  The block's matrix work is 3/4 of the direct form's.
//
// Why two kernels.  Four accumulator sets per output pair are twice the direct form's registers per output; on one CU that halves
// the tile and doubles the weight stream per output (3 splits x 2 B x 4 x 2C x C = 3.1 MB per tile already).  So the channels are
// split instead:
//   K1  f32s_gate_kernel: one workgroup = (clip, tile of 64 pairs = 128 outputs, HALF of the gate channels): 8 waves x 16 gate
//       channels (a 32-row MFMA tile = 16 tanh + 16 sigmoid rows) x 4 products x 64 pair columns = 128 accumulator registers, K = all
//       256 input channels; output transform + gate in registers; g = tanh . sigmoid leaves as fp32 [B][C][L] -- INTO h_out, which
//       is exactly that size and is overwritten by K2.
//   K2  f32s_out_kernel:  one workgroup = (clip, 128-sample tile): [res_conv; skip_conv] g (WaveNet.py:93-95) with g staged from
//       memory (3-way split in the staging pass), then h' = (h + part_t + res) sqrt(1/2) over h_out and skip (+)= in place.
//       A workgroup reads the g columns of its own tile only and has consumed all of them before its first store, so running in
//       place over h_out is safe.
// Per 128 outputs the weight stream is 3.1 MB (K1, both halves) + 0.8 MB (K2) against the direct form's 2.4 + 0.8 MB; the extra
// traffic is one fp32 write + read of g (2 x 16.4 MB per clip and layer) on a kernel that sits at 9 % of the HBM roofline.
#include <type_traits>

#include "ap_common.h"

#ifdef AP_S2_NOSB
#define AP_SB()
#else
#define AP_SB() __builtin_amdgcn_sched_barrier(0)
#endif

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int C2_ = 256;                 // res = skip channels
constexpr int NPT_ = 64;                 // pairs per K1 tile (128 outputs)
constexpr int KC1_ = 32;                 // input channels per K1 chunk: 4 products x 32 = 128 K rows = 8 k-steps (96 MFMAs per wave: a chunk
                                         // must outlast the memory latency of the next chunk's loads; 16-channel chunks did not: 26 ms per launch)
constexpr int XS1_ = 4 * KC1_ + 8;       // bf16 per pair-column row of a K1 X image (272 B: conflict-free ds_read_b128 / ds_write_b128)
constexpr int XIMG1_ = NPT_ * XS1_ * 2;  // 17,408 B per split image
constexpr int XBUF1_ = 3 * XIMG1_;       // three splits per buffer
constexpr int BT2_ = 128;                // K2 time tile
constexpr int KC2_ = 64;                 // g channels per K2 chunk = 4 k-steps (192 MFMAs per wave)
constexpr int XS2_ = KC2_ + 8;           // bf16 per column row of a K2 g image (144 B: conflict-free ds_read_b128)
constexpr int XIMG2_ = BT2_ * XS2_ * 2;  // 18,432 B per split image
constexpr int XBUF2_ = 3 * XIMG2_;

// four values -> three bf16 parts each (x = p0 + p1 + p2 exactly), packed two per dword (ap_resblock_f32s.hip: split3x4)
__device__ __forceinline__ void split3x4_(const float (&x)[4], u32x2 (&out)[3]) {
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {
    float v0 = x[2 * pr], v1 = x[2 * pr + 1];
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
      out[s][pr] = pk;
      if (s < 2) {
        v0 -= __builtin_bit_cast(float, pk << 16);
        v1 -= __builtin_bit_cast(float, pk & 0xffff0000u);
      }
    }
  }
}

__device__ __forceinline__ void split3_(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  const float r1 = x - (float)p[0];
  p[1] = (__bf16)r1;
  p[2] = (__bf16)(r1 - (float)p[1]);
}

// the fp32 kernels' compensated exp and gate (ap_common.h): the gate is not where the arithmetic modes differ
__device__ __forceinline__ float gate_(float a, float b) { return gate(a, b); }

// the six partial products kept, as (weight split, activation split)
#define AP_SPLIT_TERMS2(F) F(0, 0) F(0, 1) F(1, 0) F(0, 2) F(2, 0) F(1, 1)

__device__ __forceinline__ void block_to_tile(int bid, int nblk, int &logical) {   // XCD-contiguous runs (placement only)
  const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
  logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace

// K1 weight image: [half 2][wave 8][chunk 8][product 4][k half 2][split 3][lane 64][8]; lane (i, hh): row i < 16 = tanh row of gate
// channel 128 half + 16 wave + i, row i >= 16 = sigmoid row of channel ... + i - 16; k = input channel 32 chunk + 16 kh + 8 hh + jj.
__global__ void pack_w1_split23_kernel(const float *__restrict__ w1f, __bf16 *__restrict__ out) {
  constexpr int C = C2_;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (.., product, lane, jj), all 3 splits
  const size_t total = (size_t)2 * 8 * (C / KC1_) * 4 * 2 * 64 * 8;
  if (idx >= total) return;
  const int jj = idx & 7, lane = (idx >> 3) & 63, kh = (idx >> 9) & 1, prod = (idx >> 10) & 3;
  size_t rest = idx >> 12;
  const int ch = rest % (C / KC1_); rest /= (C / KC1_);
  const int w = rest & 7, half = (int)(rest >> 3);
  const int i = lane & 31, hh = lane >> 5;
  const int cg = 128 * half + 16 * w + (i & 15);
  const int o = (i >> 4) * C + cg;
  const int c = ch * KC1_ + 16 * kh + 8 * hh + jj;
  const float *p = w1f + ((size_t)o * C + c) * 3;
  const double w0 = p[0], w1 = p[1], w2 = p[2];
  const double v = prod == 0 ? w0 : prod == 1 ? (w0 + w1 + w2) * 0.5 : prod == 2 ? (w0 - w1 + w2) * 0.5 : w2;
  __bf16 sp[3];
  split3_((float)v, sp);
  const size_t frag = ((((((size_t)half * 8 + w) * (C / KC1_) + ch) * 4 + prod) * 2 + kh) * 3);
#pragma unroll
  for (int s = 0; s < 3; s++) out[((frag + s) * 64 + lane) * 8 + jj] = sp[s];
}

int launch_pack_split23(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C;
  const size_t n1f = (size_t)2 * C * C * 3, n1w = (size_t)4 * 2 * C * C;
  for (int n = 0; n < ctx->NL; n++)
    pack_w1_split23_kernel<<<(unsigned)((n1w + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1f, (__bf16 *)ctx->w1w_s + n * n1w * 3);
  AP_HIP(hipGetLastError());
  return 0;
}

// ---- K1: GEMM1 in F(2,3) form over half of the gate channels + output transform + gate -> g (fp32) --------------------------------
__global__ __launch_bounds__(512, 2) void f32s_gate_kernel(const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ gout,
                                                           const __bf16 *__restrict__ w1w, const float *__restrict__ b1, int L, int logd,
                                                           int np, int ntiles, int nblk) {
  constexpr int C = C2_, NCH = C / KC1_;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF1_ + C * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int d = 1 << logd;
  int logical;
  block_to_tile(blockIdx.x, nblk, logical);
  const int half = __builtin_amdgcn_readfirstlane(logical & 1);
  const int b = __builtin_amdgcn_readfirstlane((logical >> 1) / ntiles);
  const int p0 = __builtin_amdgcn_readfirstlane(((logical >> 1) % ntiles) * NPT_);
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  auto clip_rsrc = [&](const float *base) {
    const uint64_t hb = (uint64_t)(base + (size_t)b * C * L);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)clip_bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t hrs = clip_rsrc(hin);
  float *ptl = reinterpret_cast<float *>(lds + 2 * XBUF1_);
  if (tid < C) ptl[tid] = pt[tid];

  // accumulators: [product][column tile]; m2 starts from the conv's bias (both outputs of a pair contain m2 exactly once)
  f32x16 acc[4][2];
#pragma unroll
  for (int pr = 0; pr < 4; pr++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[pr][ct][r] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; q++) {                                 // registers 4q .. 4q+3: rows 8q + 4hh + (0..3); q < 2 tanh, q >= 2 sigmoid rows
    const int cg = 128 * half + 16 * wave + 8 * (q & 1) + 4 * hh;
    const float4 bv = *reinterpret_cast<const float4 *>(b1 + (q >> 1) * C + cg);
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      acc[1][ct][4 * q + 0] = bv.x;
      acc[1][ct][4 * q + 1] = bv.y;
      acc[1][ct][4 * q + 2] = bv.z;
      acc[1][ct][4 * q + 3] = bv.w;
    }
  }

  // ---- staging: thread = (pair column tid & 63, channel octet (tid >> 6) & 3, product pair tid >> 8).  Threads 0-255 form the
  // differences of products 0, 1 from taps u0, u1, u2; threads 256-511 those of products 2, 3 from u1, u2, u3: 24 four-byte loads,
  // FiLM add (WaveNet.py:84), zero padding (:26-27), two differences x 8 channels, 3-way split, six ds_write_b128 per chunk.
  const int pc = tid & (NPT_ - 1), q4 = ((tid >> 6) & 3) * 8, pp = tid >> 8;
  unsigned voff[3];
  bool tok[3];
  {
    const int p = p0 + pc;
    const int tf = ((p >> logd) << (logd + 1)) + (p & (d - 1));
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int tp = tf + (i + pp - 1) * d;                       // pp = 0: taps -d, 0, +d;  pp = 1: taps 0, +d, +2d
      tok[i] = p < np && tp >= 0 && tp < L;
      voff[i] = ((unsigned)min(max(tp, 0), L - 1) + (unsigned)q4 * (unsigned)L) * 4u;
    }
  }
  float xr[3][8];
  auto issue_loads = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int e = 0; e < 8; e++)
        xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, voff[i], (ch * KC1_ + e) * L * 4, 0));
  };
  auto store_chunk = [&](unsigned char *dst, int ch) {
    const float4 pv0 = *reinterpret_cast<const float4 *>(ptl + ch * KC1_ + q4), pv1 = *reinterpret_cast<const float4 *>(ptl + ch * KC1_ + q4 + 4);
    const float pte[8] = {pv0.x, pv0.y, pv0.z, pv0.w, pv1.x, pv1.y, pv1.z, pv1.w};
    float u[3][8];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) u[i][e] = tok[i] ? xr[i][e] + pte[e] : 0.f;
    float da[8], db[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (pp == 0) {                                              // (uniform per wave) u = {u0, u1, u2}
        da[e] = u[0][e] - u[2][e];                                // product 0: u0 - u2
        db[e] = u[1][e] + u[2][e];                                // product 1: u1 + u2
      } else {                                                    // u = {u1, u2, u3}
        da[e] = u[1][e] - u[0][e];                                // product 2: u2 - u1
        db[e] = u[2][e] - u[0][e];                                // product 3: u3 - u1
      }
    }
    u32x2 pa0[3], pa1[3], pb0[3], pb1[3];
    {
      const float a0_[4] = {da[0], da[1], da[2], da[3]}, a1_[4] = {da[4], da[5], da[6], da[7]};
      const float b0_[4] = {db[0], db[1], db[2], db[3]}, b1_[4] = {db[4], db[5], db[6], db[7]};
      split3x4_(a0_, pa0); split3x4_(a1_, pa1); split3x4_(b0_, pb0); split3x4_(b1_, pb1);
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
      *reinterpret_cast<u32x4 *>(dst + s * XIMG1_ + (pc * XS1_ + (2 * pp) * KC1_ + q4) * 2) = u32x4{pa0[s][0], pa0[s][1], pa1[s][0], pa1[s][1]};
      *reinterpret_cast<u32x4 *>(dst + s * XIMG1_ + (pc * XS1_ + (2 * pp + 1) * KC1_ + q4) * 2) = u32x4{pb0[s][0], pb0[s][1], pb1[s][0], pb1[s][1]};
    }
  };

  // ---- GEMM1: 8 chunks x 8 k-steps (product x k half, K = 16 channels each), 12 MFMAs each: 2 column tiles x 6 partial products.
  // A k-step is short (384 cycles of matrix pipe per wave) against an L2 round trip, so the weight fragments (3 splits = 12 VGPRs
  // per k-step) stream through a ring FOUR k-steps deep (fragments requested one k-step ahead: 25.5 ms per launch; this: see
  // DESIGN.md), and the B fragments of a k-step's second column tile / the next k-step's first are read from LDS under the
  // MFMAs of the tile before them.
  const u32x4 *ap = reinterpret_cast<const u32x4 *>(w1w) + ((size_t)(half * 8 + wave) * NCH * 8 * 3) * 64 + lane;
  auto load_a = [&](bf16x8(&a)[3], int kk) {                     // kk = (chunk * 4 + product) * 2 + k half, clamped to the last one
    const u32x4 *base = ap + (size_t)(kk < NCH * 8 ? kk : NCH * 8 - 1) * 3 * 64;
#pragma unroll
    for (int s = 0; s < 3; s++) a[s] = __builtin_bit_cast(bf16x8, base[s * 64]);
  };
  const int rdoff = (j * XS1_ + 8 * hh) * 2;
  auto read_b = [&](bf16x8(&bv)[3], const unsigned char *xb, int ks, int ct) {   // k-step ks of the chunk = product ks >> 1, k half ks & 1
#pragma unroll
    for (int s = 0; s < 3; s++)
      bv[s] = *reinterpret_cast<const bf16x8 *>(xb + s * XIMG1_ + (32 * ct) * (XS1_ * 2) + (ks >> 1) * (KC1_ * 2) + (ks & 1) * 32);
  };
  bf16x8 a[4][3], bx0[3], bx1[3];
  issue_loads(0);
  __syncthreads();                                              // part_t visible
  store_chunk(lds, 0);
  load_a(a[0], 0);
  load_a(a[1], 1);
  load_a(a[2], 2);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < NCH; ch++) {
    const unsigned char *xb = lds + (ch & 1) * XBUF1_ + rdoff;
    unsigned char *nb = lds + ((ch + 1) & 1) * XBUF1_;
    read_b(bx0, xb, 0, 0);
    auto kstep = [&](auto KS_TAG) {
      constexpr int ks = decltype(KS_TAG)::value, pr = ks >> 1;
      // vmcnt retires in issue order: a k-step's weight request goes out before the chunk's activation loads
      load_a(a[(ks + 3) & 3], 8 * ch + ks + 3);
      if constexpr (ks == 0) issue_loads(ch + 1 < NCH ? ch + 1 : NCH - 1);
      read_b(bx1, xb, ks, 1);
      AP_SB();
#define AP_T(i, jx) acc[pr][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 3][i], bx0[jx], acc[pr][0], 0, 0, 0);
      AP_SPLIT_TERMS2(AP_T)
#undef AP_T
      if constexpr (ks < 7) read_b(bx0, xb, ks + 1, 0);
      AP_SB();
#define AP_T(i, jx) acc[pr][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 3][i], bx1[jx], acc[pr][1], 0, 0, 0);
      AP_SPLIT_TERMS2(AP_T)
#undef AP_T
      if constexpr (ks == 7) store_chunk(nb, ch + 1 < NCH ? ch + 1 : NCH - 1);   // (after the last chunk: a harmless re-store into the idle buffer)
      AP_SB();
    };
    kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
    kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
    kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
    __syncthreads();
  }

  // ---- output transform + gate (WaveNet.py:90), straight from the accumulators: registers r < 8 are tanh rows of channels
  // (r & 3) + 8 (r >> 2) + 4 hh of the wave's 16, registers r + 8 the sigmoid rows of the same channels
  const __amdgpu_buffer_rsrc_t grs = clip_rsrc(gout);
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    const int p = p0 + 32 * ct + j;
    const int tf = ((p >> logd) << (logd + 1)) + (p & (d - 1));
    const bool ok0 = p < np && tf < L, ok1 = p < np && tf + d < L;
    const unsigned o0 = ok0 ? (unsigned)tf * 4u : 0x80000000u, o1 = ok1 ? (unsigned)(tf + d) * 4u : 0x80000000u;   // outside: store dropped
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int cg = 128 * half + 16 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float yt0 = (acc[0][ct][r] + acc[1][ct][r]) + acc[2][ct][r];
      const float ys0 = (acc[0][ct][r + 8] + acc[1][ct][r + 8]) + acc[2][ct][r + 8];
      const float yt1 = (acc[1][ct][r] - acc[2][ct][r]) + acc[3][ct][r];
      const float ys1 = (acc[1][ct][r + 8] - acc[2][ct][r + 8]) + acc[3][ct][r + 8];
      const unsigned row = (unsigned)cg * (unsigned)L * 4u;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gate_(yt0, ys0)), grs, o0 + (ok0 ? row : 0u), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gate_(yt1, ys1)), grs, o1 + (ok1 ? row : 0u), 0, 0);
    }
  }
}

// ---- K2: [res_conv; skip_conv] g + the block's two epilogues, in place over gio (g in, h' out) --------------------------------
__global__ __launch_bounds__(512, 2) void f32s_out_kernel(const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ gio,
                                                          float *__restrict__ skip, const __bf16 *__restrict__ w2p, const float *__restrict__ b2,
                                                          int L, int accumulate, int ntiles, int nblk) {
  constexpr int C = C2_, NCH = C / KC2_, NKS = C / 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF2_];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  int logical;
  block_to_tile(blockIdx.x, nblk, logical);
  const int b = __builtin_amdgcn_readfirstlane(logical / ntiles);
  const int t0 = __builtin_amdgcn_readfirstlane((logical % ntiles) * BT2_);
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  auto clip_rsrc = [&](const float *base) {
    const uint64_t hb = (uint64_t)(base + (size_t)b * C * L);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)clip_bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t hrs = clip_rsrc(hin), grs = clip_rsrc(gio), srs = clip_rsrc(skip);

  // accumulators [pass 0 = res rows | 1 = skip rows][column tile]: rows 32 wave + .. of each; start from the bias (+ part_t on the
  // res rows: u = h + part_t re-enters the residual, WaveNet.py:84,97)
  f32x16 acc[2][4];
#pragma unroll
  for (int pass = 0; pass < 2; pass++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = 32 * wave + 8 * q + 4 * hh;
      float4 v = *reinterpret_cast<const float4 *>(b2 + pass * C + c);
      if (pass == 0) {
        const float4 pv = *reinterpret_cast<const float4 *>(pt + c);
        v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
      }
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        acc[pass][ct][4 * q + 0] = v.x;
        acc[pass][ct][4 * q + 1] = v.y;
        acc[pass][ct][4 * q + 2] = v.z;
        acc[pass][ct][4 * q + 3] = v.w;
      }
    }

  // ---- g staging: thread = (column tid & 127, 16 channels (tid >> 7) * 16): sixteen four-byte loads (coalesced along the sample axis),
  // 3-way split, two ds_write_b128 per split image
  const int col = tid & (BT2_ - 1), o8 = (tid >> 7) * 16;
  const int tcol = t0 + col;
  const unsigned gvoff = tcol < L ? ((unsigned)tcol + (unsigned)o8 * (unsigned)L) * 4u : 0x80000000u;   // outside the clip: zeros
  float xr[16];
  auto issue_loads = [&](int ch) {
#pragma unroll
    for (int e = 0; e < 16; e++)
      xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, gvoff, (ch * KC2_ + e) * L * 4, 0));
  };
  auto store_chunk = [&](unsigned char *dst) {
#pragma unroll
    for (int o = 0; o < 2; o++) {
      u32x2 lo[3], hi[3];
      const float a[4] = {xr[8 * o + 0], xr[8 * o + 1], xr[8 * o + 2], xr[8 * o + 3]}, c[4] = {xr[8 * o + 4], xr[8 * o + 5], xr[8 * o + 6], xr[8 * o + 7]};
      split3x4_(a, lo);
      split3x4_(c, hi);
#pragma unroll
      for (int s = 0; s < 3; s++)
        *reinterpret_cast<u32x4 *>(dst + s * XIMG2_ + (col * XS2_ + o8 + 8 * o) * 2) = u32x4{lo[s][0], lo[s][1], hi[s][0], hi[s][1]};
    }
  };

  // ---- GEMM2: 4 chunks x 4 k-steps, 48 MFMAs each (2 passes x 4 column tiles x 6 partial products); weight image of
  // ap_resblock_f32s.hip: [wave][pass 2][k-step 16][split 3][lane][8]
  const u32x4 *ap = reinterpret_cast<const u32x4 *>(w2p) + (size_t)(wave * 2) * NKS * 3 * 64 + lane;
  auto load_a = [&](bf16x8(&a)[2][3], int ks) {
    const int k = ks < NKS ? ks : NKS - 1;
#pragma unroll
    for (int pass = 0; pass < 2; pass++)
#pragma unroll
      for (int s = 0; s < 3; s++) a[pass][s] = __builtin_bit_cast(bf16x8, ap[((size_t)pass * NKS + k) * 3 * 64 + s * 64]);
  };
  const int rdoff = (j * XS2_ + 8 * hh) * 2;
  auto mma_k = [&](const bf16x8(&a)[2][3], const unsigned char *xb) {      // xb: buffer + rdoff + (k-step in chunk) * 32
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      bf16x8 bv[3];
#pragma unroll
      for (int s = 0; s < 3; s++) bv[s] = *reinterpret_cast<const bf16x8 *>(xb + s * XIMG2_ + (32 * ct) * (XS2_ * 2));
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
#define AP_T(i, jx) acc[pass][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pass][i], bv[jx], acc[pass][ct], 0, 0, 0);
        AP_SPLIT_TERMS2(AP_T)
#undef AP_T
      }
    }
  };
  issue_loads(0);
  bf16x8 a0[2][3], a1[2][3];
  load_a(a0, 0);
  store_chunk(lds);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < NCH; ch++) {
    const unsigned char *xb = lds + (ch & 1) * XBUF2_ + rdoff;
    load_a(a1, 4 * ch + 1);
    issue_loads(ch + 1 < NCH ? ch + 1 : NCH - 1);
    AP_SB();
    mma_k(a0, xb);
    AP_SB();
    load_a(a0, 4 * ch + 2);
    AP_SB();
    mma_k(a1, xb + 32);
    AP_SB();
    load_a(a1, 4 * ch + 3);
    AP_SB();
    mma_k(a0, xb + 64);
    AP_SB();
    load_a(a0, 4 * ch + 4);
    AP_SB();
    mma_k(a1, xb + 96);
    store_chunk(lds + ((ch + 1) & 1) * XBUF2_);                  // (after the last chunk: a harmless re-store into the idle buffer)
    AP_SB();
    __syncthreads();
  }
  // every g load of this workgroup has been consumed (each thread waited for its own before writing LDS, and the barrier above
  // follows the last write): the stores below may overwrite the tile's g columns with h'

  // ---- epilogues: h' = (h + acc) sqrt(1/2) (WaveNet.py:97; part_t and the bias are in acc), skip (+)= acc (:131-133)
  const float RS = 0.707106781186547524f;
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const int t = t0 + 32 * ct + j;
    const unsigned eo = t < L ? ((unsigned)(32 * wave + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
    float hv[16], sv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      hv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 2));
      sv[r] = accumulate ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 2)) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (hv[r] + acc[0][ct][r]) * RS), grs, eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv[r] + acc[1][ct][r]), srs, eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
    }
  }
}

bool resblock_split23_serves(const ap_ctx *ctx, int B, int L) {
  if (ctx->cfg.precision != AP_PREC_F32_SPLIT || ctx->C != C2_ || ctx->S != C2_ || ctx->f32_form != 1 || !ctx->w1w_s) return false;
  if ((size_t)C2_ * (size_t)L * 4 >= ((size_t)1 << 31)) return false;       // a clip's rows must stay below the buffer descriptor's range
  return (long long)B * ((L + 63) / 64 + 2) * 2 < (1ll << 31);
}

int launch_resblock_split23(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip, int accumulate, int B, int L,
                            hipStream_t st) {
  const int C = C2_;
  const int logd = layer % ctx->cfg.dilation_cycle;
  const long long d = 1ll << logd;
  const long long np = (L / (2 * d)) * d + ((L % (2 * d)) < d ? (L % (2 * d)) : d);      // pairs whose first output is inside the clip
  const int nt1 = (int)((np + NPT_ - 1) / NPT_);
  const int nblk1 = B * nt1 * 2;
  const __bf16 *w1w = (const __bf16 *)ctx->w1w_s + (size_t)layer * 4 * 2 * C * C * 3;
  const __bf16 *w2p = (const __bf16 *)ctx->w2p_s + (size_t)layer * 2 * C * C * 3;
  f32s_gate_kernel<<<(unsigned)nblk1, 512, 0, st>>>(hin, pt, hout, w1w, ctx->b1 + (size_t)layer * 2 * C, L, logd, (int)np, nt1, nblk1);
  const int nt2 = (L + BT2_ - 1) / BT2_;
  const int nblk2 = B * nt2;
  f32s_out_kernel<<<(unsigned)nblk2, 512, 0, st>>>(hin, pt, hout, skip, w2p, ctx->b2 + (size_t)layer * 2 * C, L, accumulate, nt2, nblk2);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
