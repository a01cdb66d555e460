// AP_PREC_F32_SPLIT: fused Residual_block.forward (WaveNet.py:75-97) at fp32 accuracy on the bf16 matrix pipe.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of v_mfma_f32_32x32x2_f32, so an fp32 GEMM is cheaper as
// several bf16 ones: every fp32 operand is split EXACTLY into three bf16 parts x = x0 + x1 + x2 (x0 = rne(x),
// x1 = rne(x - x0), x2 = rne(x - x0 - x1): 3 x 8 mantissa bits), and x*w is summed from the six partial products
// whose weight is >= 2^-16 of the full product (x0w0, x0w1, x1w0, x0w2, x2w0, x1w1).  Each partial product of two
// bf16 values is exact in fp32, the accumulation is the MFMA's fp32 one, and the three dropped terms are below
// 2^-23 of the product -- fp32-class arithmetic (measured: 2x the rounding noise of the plain fp32 kernel, from the
// 6x longer accumulation chain; tests/test_gpu_parity.py holds both kernels to the same tolerances).
// 6 bf16 MFMAs = 6/16 of the fp32 instruction's time.
//
// Structure follows tools/csrc/ap_resblock_bf16.hip (128-sample tiles, 8 waves x (64 rows x 128 columns), bf16 [column][k] LDS
// images, 16-B read-modify-write through a wave-private patch) with three images per operand; chunks are 16 channels
// (48 K rows) so two X buffers x three splits fit, and gate + GEMM2 run per 64-column half so the three g images do.
#include <type_traits>

#include "ap_common.h"

namespace ap {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace f32s {
constexpr int BT = 128;                  // time tile
constexpr int BKC = 16;                  // channels per staged chunk -> 48 K rows = 3 k-steps (one per tap)
constexpr int XS = 3 * BKC + 8;          // bf16 per column row of an X image (112 B: conflict-free b128 reads)
constexpr int HT = 64;                   // columns per gate/GEMM2 half
constexpr int PSTR = 32;                 // fp32 row stride of the wave-private output patch (128-B rows: conflict-free for the column writes and the 16-lane groups of the b128 row reads; 144-B rows were 2-way)

__device__ __forceinline__ int rowoff_s(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ float bf_up(__bf16 v) { return (float)v; }

// x = p[0] + p[1] + p[2] exactly (fp32 has 24 mantissa bits, each part carries 8)
__device__ __forceinline__ void split3(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  const float r1 = x - bf_up(p[0]);
  p[1] = (__bf16)r1;
  const float r2 = r1 - bf_up(p[1]);
  p[2] = (__bf16)r2;
}

// four values at once, written so the conversions lower to v_cvt_pk_bf16_f32 (two per instruction) and the
// widenings to one shift / mask: 11 VALU per pair instead of ~20
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x4(const float (&x)[4], u32x2s (&out)[3]) {
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
__device__ __forceinline__ float gate_s(float a, float b) {
  a = fminf(fmaxf(a, -15.0f), 15.0f);
  b = fmaxf(b, -80.0f);
  float E = exp_acc_s(2.0f * a);
  float F = exp_acc_s(-b);
  return (E - 1.0f) * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
}
}  // namespace f32s
using namespace f32s;

// ---- weight images -------------------------------------------------------------------------------------------
// GEMM1: [wave C/32][chunk C/16][kstep 3 = tap][rowtile 2][split 3][lane 64][8]; wave w owns gate channels
// [32w, 32w+32): row tile 0 = tanh rows, 1 = sigmoid rows; k-step = tap, channels ch*16 + 8h + jj.
__global__ void pack_w1_split_kernel(const float *__restrict__ w1f, __bf16 *__restrict__ out, int C) {
  const int NW = C / 32, NCH = C / BKC;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (.., lane, jj), all 3 splits
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
  __bf16 p[3];
  split3(w1f[((size_t)o * C + c) * 3 + ks], p);
  size_t frag = ((((size_t)w * NCH + ch) * 3 + ks) * 2 + rt) * 3;
#pragma unroll
  for (int s = 0; s < 3; s++) out[((frag + s) * 64 + lane) * 8 + jj] = p[s];
}

// GEMM2: [wave][pass 2][kstep C/16][split 3][lane][8]; pass 0 = res_conv rows of the wave's channels, 1 = skip rows.
__global__ void pack_w2_split_kernel(const float *__restrict__ w2f, __bf16 *__restrict__ out, int C) {
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
  __bf16 p[3];
  split3(w2f[(size_t)o * C + k], p);
  size_t frag = (((size_t)w * 2 + pass) * NKS + ks) * 3;
#pragma unroll
  for (int s = 0; s < 3; s++) out[((frag + s) * 64 + lane) * 8 + jj] = p[s];
}

int launch_pack_split(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C, S = ctx->S, NL = ctx->NL;
  for (int n = 0; n < NL; n++) {
    size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
    pack_w1_split_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1, (__bf16 *)ctx->w1p_s + n * n1 * 3, C);
    pack_w2_split_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, (__bf16 *)ctx->w2p_s + n * n2 * 3, C);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

// ---- the kernel ---------------------------------------------------------------------------------------------
// the six partial products kept, as (weight split, activation split)
#define AP_SPLIT_TERMS(F) F(0, 0) F(0, 1) F(1, 0) F(0, 2) F(2, 0) F(1, 1)

template <int C, bool E4, bool TRACE>
__global__ __launch_bounds__(C / 32 * 64, 2) void resblock_f32s_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const __bf16 *__restrict__ w1p, const float *__restrict__ b1, const __bf16 *__restrict__ w2p,
    const float *__restrict__ b2, int L, int d, int accumulate, int ntiles, int nblk,
    unsigned long long *__restrict__ trace) {
  constexpr int NW = C / 32, NT = NW * 64, NCH = C / BKC;
  static_assert(NT == 512 && NCH % 2 == 0, "built for C = 256 (8 waves)");
  constexpr int GS = C + 8;                                    // bf16 per column row of a g image (528 B)
  constexpr int XIMG = BT * XS * 2;                            // 14,336 B per X image
  constexpr int XBUF = 3 * XIMG;                               // three splits per buffer
  constexpr int GIMG = HT * GS * 2;                            // 33,792 B per g image (64 columns)
  constexpr int UNION = (2 * XBUF > 3 * GIMG) ? 2 * XBUF : 3 * GIMG;
  constexpr int PTOFF = UNION;                                 // part_t (C floats)
  constexpr int LDS_BYTES = PTOFF + C * 4;
  constexpr int PATCH_FLOATS = E4 ? NW * 32 * PSTR : 4;
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
        acc[rt][ct][4 * q + 0] = bv.x;
        acc[rt][ct][4 * q + 1] = bv.y;
        acc[rt][ct][4 * q + 2] = bv.z;
        acc[rt][ct][4 * q + 3] = bv.w;
      }
    }

  // ---- X staging: thread = (column tid&127, channel quad tid>>7) for each of the 3 taps: 12 buffer loads at the head
  // of a chunk; FiLM add (WaveNet.py:84), zero padding (:26-27), 3-way split and three ds_write_b64 per tap at its tail.
  if (tid < C) reinterpret_cast<float *>(lds + PTOFF)[tid] = pt[tid];
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
      for (int e = 0; e < 4; e++) u[e] = tok[tap] ? xr[tap][e] + pte[e] : 0.f;
      u32x2s pk[3];
      split3x4(u, pk);
#pragma unroll
      for (int s = 0; s < 3; s++)
        *reinterpret_cast<u32x2s *>(dst + s * XIMG + (col * XS + tap * BKC + q4) * 2) = pk[s];
    }
  };

  issue_loads(0);
  __syncthreads();                                              // part_t visible
  store_chunk(lds, 0);
  __syncthreads();
  mark(1);

  // ---- GEMM1: 48 k-steps (16 chunks x 3 taps), 48 MFMAs each: 2 row tiles x 4 column tiles x 6 partial products.
  // Weight fragments (2 row tiles x 3 splits = 24 VGPRs per k-step) stream from L2 one k-step ahead, ping-pong.
  auto load_a = [&](bf16x8(&a)[2][3], const u32x4 *base) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int s = 0; s < 3; s++) a[rt][s] = __builtin_bit_cast(bf16x8, base[(rt * 3 + s) * 64]);
  };
  const int rdoff = (j * XS + 8 * hh) * 2;                      // this lane's B-fragment byte offset inside an X image
  // bpre holds the first column tile's three B fragments of the k-step about to run; inside a chunk the next k-step's
  // are fetched under this one's last MFMAs, so a k-step boundary does not wait on LDS (across the chunk barrier the
  // other buffer is not valid yet: NEXT = false there and the caller refills bpre after the barrier)
  bf16x8 bpre[3];
  auto read_b = [&](bf16x8(&bv)[3], const unsigned char *xb, int ct) {
#pragma unroll
    for (int s = 0; s < 3; s++) bv[s] = *reinterpret_cast<const bf16x8 *>(xb + s * XIMG + (32 * ct) * (XS * 2));
  };
  auto mma_k = [&](const bf16x8(&a)[2][3], const unsigned char *xb, auto NEXT) {   // xb: buffer + rdoff + tap * 32
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      bf16x8 bv[3];
      if (ct == 0) {
#pragma unroll
        for (int s = 0; s < 3; s++) bv[s] = bpre[s];
      } else {
        read_b(bv, xb, ct);
      }
      if (ct == 3 && decltype(NEXT)::value) read_b(bpre, xb + 32, 0);
#pragma unroll
      for (int rt = 0; rt < 2; rt++) {
#define AP_T(i, jx) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt][i], bv[jx], acc[rt][ct], 0, 0, 0);
        AP_SPLIT_TERMS(AP_T)
#undef AP_T
      }
    }
  };
  using YES = std::true_type;
  using NO = std::false_type;
  const u32x4 *ap = reinterpret_cast<const u32x4 *>(w1p) + (size_t)wave * NCH * 3 * 6 * 64 + lane;
  auto aset = [&](int kk) { return ap + (size_t)(kk < NCH * 3 ? kk : NCH * 3 - 1) * 6 * 64; };
  // the next chunk's FiLM add / split / pack (VALU) goes into the gaps of the third k-step's 48 MFMAs
  auto pack_between_mfmas = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 48; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      if (i % 12 == 1 && i < 36) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
  };
  bf16x8 a0[2][3], a1[2][3];
  load_a(a0, aset(0));
#pragma unroll 1
  for (int it = 0; it < NCH / 2; it++) {
    const int c0 = 2 * it, kk = 6 * it;
    const unsigned char *x0 = lds + rdoff, *x1 = lds + XBUF + rdoff;
    read_b(bpre, x0, 0);
    // vmcnt retires in issue order: the next weight set is requested before the (HBM-latency) X loads
    load_a(a1, aset(kk + 1));
    issue_loads(c0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a0, x0, YES{});
    __builtin_amdgcn_sched_barrier(0);
    if (it == 4) mark(10);
    load_a(a0, aset(kk + 2));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a1, x0 + 32, YES{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(a1, aset(kk + 3));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a0, x0 + 64, NO{});
    store_chunk(lds + XBUF, c0 + 1);
    pack_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    if (it == 4) mark(12);
    __syncthreads();
    if (it == 3) mark(3);
    if (it == 4) mark(13);
    read_b(bpre, x1, 0);
    load_a(a0, aset(kk + 4));
    issue_loads(c0 + 2 < NCH ? c0 + 2 : NCH - 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a1, x1, YES{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(a1, aset(kk + 5));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a0, x1 + 32, YES{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(a0, aset(kk + 6));
    __builtin_amdgcn_sched_barrier(0);
    mma_k(a1, x1 + 64, NO{});
    store_chunk(lds, c0 + 2 < NCH ? c0 + 2 : NCH - 1);          // after the last chunk: a harmless re-store
    pack_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (it == 0) mark(2);
  }
  mark(4);

  // ---- per 64-column half: gate (WaveNet.py:90) -> three g images [col][channel]; GEMM2 in two passes of 32 rows x 64
  // columns (pass 0 = res_conv rows -> h', pass 1 = skip_conv rows -> skip; WaveNet.py:93-97, :133).
  constexpr int NKS = C / 16;
  const float RS = 0.707106781186547524f;
  const __amdgpu_buffer_rsrc_t srs = clip_rsrc(skip);
  const __amdgpu_buffer_rsrc_t ors = clip_rsrc(hout);
  float *patch = patch_mem + (E4 ? wave * 32 * PSTR : 0);
  const unsigned char *gb = lds + (j * GS + 8 * hh) * 2;
  const u32x4 *ap2 = reinterpret_cast<const u32x4 *>(w2p) + (size_t)(wave * 2) * NKS * 3 * 64 + lane;
  const float *b2l = b2, *ptl = pt;
  asm volatile("" : "+s"(b2l), "+s"(ptl));

  auto half = [&](auto HTAG) {
    constexpr int h = decltype(HTAG)::value;
    // gate of this half's two column tiles
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
      const int ct = 2 * h + c2;
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        float gv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) gv[e] = gate_s(acc[0][ct][4 * qq + e], acc[1][ct][4 * qq + e]);
        u32x2s pk[3];
        split3x4(gv, pk);
#pragma unroll
        for (int s = 0; s < 3; s++)
          *reinterpret_cast<u32x2s *>(lds + s * GIMG + ((32 * c2 + j) * GS + 32 * wave + 8 * qq + 4 * hh) * 2) = pk[s];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    mark(h == 0 ? 5 : 8);

    unsigned evoff[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
      if constexpr (E4) {
        const int t = t0 + 64 * h + 32 * c2 + 4 * (lane & 7);
        evoff[c2] = t < L ? ((unsigned)(32 * wave + (lane >> 3)) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      } else {
        const int t = t0 + 64 * h + 32 * c2 + j;
        evoff[c2] = t < L ? ((unsigned)(32 * wave + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      }
    }                                                           // 0x80000000: outside the clip -> load 0 / store dropped
    auto gemm2_pass = [&](auto PTAG) {
      constexpr int pass = decltype(PTAG)::value;
      // what this pass adds into (h for the residual, the running skip) is fetched before its GEMM, used after it
      float pre[2][16];
#pragma unroll
      for (int c2 = 0; c2 < 2; c2++) {
        if constexpr (E4) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          pass == 0 ? hrs : srs, evoff[c2] + (unsigned)(8 * p * L * 4), 0, 2));   // nt: once-touched skip rows; the residual re-read of h hits or passes without allocating
#pragma unroll
            for (int i = 0; i < 4; i++) pre[c2][4 * p + i] = v[i];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++)
            pre[c2][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                       pass == 0 ? hrs : srs, evoff[c2], ((r & 3) + 8 * (r >> 2)) * L * 4, 2));   // nt: once-touched skip rows; the residual re-read of h hits or passes without allocating
        }
      }
      f32x16 ac[2];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = 32 * wave + 8 * q + 4 * hh;
        float4 v = *reinterpret_cast<const float4 *>(b2l + pass * C + c);
        if (pass == 0) {                                         // u = h + part_t re-enters the residual
          const float4 pv = *reinterpret_cast<const float4 *>(ptl + c);
          v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
          ac[c2][4 * q + 0] = v.x;
          ac[c2][4 * q + 1] = v.y;
          ac[c2][4 * q + 2] = v.z;
          ac[c2][4 * q + 3] = v.w;
        }
      }
      const u32x4 *apass = ap2 + (size_t)pass * NKS * 3 * 64;
      auto load_a2 = [&](bf16x8(&a)[2][3], int ks) {              // two k-steps x three splits
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int s = 0; s < 3; s++)
            a[u][s] = __builtin_bit_cast(bf16x8, apass[(size_t)(((ks + u < NKS ? ks + u : NKS - 1) * 3 + s) * 64)]);
      };
      auto mma2 = [&](const bf16x8(&a)[2][3], int ks) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int c2 = 0; c2 < 2; c2++) {
            bf16x8 bv[3];
#pragma unroll
            for (int s = 0; s < 3; s++)
              bv[s] = *reinterpret_cast<const bf16x8 *>(gb + s * GIMG + (32 * c2) * (GS * 2) + (ks + u) * 32);
#define AP_T(i, jx) ac[c2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][i], bv[jx], ac[c2], 0, 0, 0);
            AP_SPLIT_TERMS(AP_T)
#undef AP_T
          }
      };
      bf16x8 p0[2][3], p1[2][3];
      load_a2(p0, 0);
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += 4) {
        load_a2(p1, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma2(p0, ks);
        __builtin_amdgcn_sched_barrier(0);
        load_a2(p0, ks + 4);
        __builtin_amdgcn_sched_barrier(0);
        mma2(p1, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      const float addm = (pass == 0 || accumulate) ? 1.0f : 0.0f;
      const float scale = pass == 0 ? RS : 1.0f;
#pragma unroll
      for (int c2 = 0; c2 < 2; c2++) {
        if constexpr (E4) {
#pragma unroll
          for (int r = 0; r < 16; r++) patch[rowoff_s(r, hh) * PSTR + j] = ac[c2][r];
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * p) * PSTR + 4 * (lane & 7));
            f32x4 o;
            o[0] = __builtin_fmaf(pre[c2][4 * p + 0], addm, v.x) * scale;
            o[1] = __builtin_fmaf(pre[c2][4 * p + 1], addm, v.y) * scale;
            o[2] = __builtin_fmaf(pre[c2][4 * p + 2], addm, v.z) * scale;
            o[3] = __builtin_fmaf(pre[c2][4 * p + 3], addm, v.w) * scale;
            // offset in the VGPR, soffset = 0 (a >8-byte buffer store with an SGPR soffset reads its data late)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), pass == 0 ? ors : srs,
                                                   evoff[c2] + (unsigned)(8 * p * L * 4), 0, 2);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++)
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(unsigned, __builtin_fmaf(pre[c2][r], addm, ac[c2][r]) * scale), pass == 0 ? ors : srs,
                evoff[c2], ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
        }
      }
    };
    gemm2_pass(std::integral_constant<int, 0>{});
    mark(h == 0 ? 6 : 9);
    __builtin_amdgcn_sched_barrier(0);
    gemm2_pass(std::integral_constant<int, 1>{});
  };
  half(std::integral_constant<int, 0>{});
  mark(7);
  __syncthreads();                                              // every wave is done reading half 0's g images
  half(std::integral_constant<int, 1>{});
  mark(14);
}

#ifdef AP_TOOLS
extern unsigned long long *g_trace_bf16;
#endif

int launch_resblock_split(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                          int accumulate, int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  if (C != 256 || S != 256) {
    set_error("AP_PREC_F32_SPLIT is built for res_channels = skip_channels = 256 only (got %d, %d)", C, S);
    return -22;
  }
  // the dilated conv in F(2,3) form where built and selected (ap_resblock_f32s2.hip: 3/4 of this kernel's matrix work)
  if (hout && resblock_split23_serves(ctx, B, L)) return launch_resblock_split23(ctx, layer, hin, pt, hout, skip, accumulate, B, L, st);
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int ntiles = (L + BT - 1) / BT;
  const int nblk = B * ntiles;
  const __bf16 *w1p = (const __bf16 *)ctx->w1p_s + (size_t)layer * 2 * C * C * 3 * 3;
  const __bf16 *w2p = (const __bf16 *)ctx->w2p_s + (size_t)layer * (C + S) * C * 3;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C;
  const float *b2 = ctx->b2 + (size_t)layer * (C + S);
#ifdef AP_TOOLS
  if (L % 4 == 0 && L >= 4 && g_trace_bf16)
    resblock_f32s_kernel<256, true, true><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d,
                                                                         accumulate, ntiles, nblk, g_trace_bf16);
  else
#endif
  if (L % 4 == 0 && L >= 4)
    resblock_f32s_kernel<256, true, false><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d,
                                                                          accumulate, ntiles, nblk, nullptr);
  else
    resblock_f32s_kernel<256, false, false><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d,
                                                                           accumulate, ntiles, nblk, nullptr);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
