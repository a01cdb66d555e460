// AP_PREC_BF16: fused Residual_block.forward (WaveNet.py:75-97) with bf16 MFMA operands
// (v_mfma_f32_32x32x16_bf16: 16x the fp32 matrix rate), fp32 accumulation, fp32 activations in HBM.
// In this mode the block is HBM/L2-bound rather than MFMA-bound, so compared with the fp32
// kernel: 128-sample tiles (halve the weight stream per sample), 8 waves x (64 rows x 128 cols), X and the gate
// output g staged in LDS as bf16 [col][k] images read with conflict-free ds_read_b128, a cheap gate, and an
// XCD-local tile order so that the +-d conv taps of one clip are re-read from that XCD's L2.
#include <type_traits>

#include "ap_common.h"

namespace ap {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BT = 128;                 // time tile
constexpr int BKC = 32;                 // channels per staged chunk -> 96 K rows = 6 k-steps of 16
constexpr int XSTRIDE = 3 * BKC + 8;    // bf16 elements per column row of the X image (208 B: conflict-free b128 reads)

__device__ __forceinline__ int rowoff_b(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// tanh(a) sigmoid(b) = (E-1)/((E+1)(1+F)); plain hardware exp2/rcp are ample next to bf16 operand rounding (2^-9)
__device__ __forceinline__ float gate_fast(float a, float b) {
  a = fminf(fmaxf(a, -15.0f), 15.0f);
  b = fmaxf(b, -80.0f);
  const float E = __builtin_amdgcn_exp2f(a * 2.885390081777926815f);
  const float F = __builtin_amdgcn_exp2f(b * -1.442695040888963407f);
  return (E - 1.0f) * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
}

// ---- weight images -------------------------------------------------------------------------------------------
// GEMM1: [wave C/32][chunk C/32][kstep 6][rowtile 2][lane 64][8]; wave w owns gate channels [32w, 32w+32):
// row tile 0 = tanh rows, 1 = sigmoid rows; k-step ks of a chunk = tap ks/2, channels ch*32 + (ks&1)*16 + 8h + jj.
__global__ void pack_w1_bf16_kernel(const float *__restrict__ w1f, __bf16 *__restrict__ out, int C) {
  const int NW = C / 32, NCH = C / BKC;
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
  int c = ch * BKC + (ks & 1) * 16 + 8 * hh + jj;
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

int launch_pack_bf16(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C, S = ctx->S, NL = ctx->NL;
  for (int n = 0; n < NL; n++) {
    size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
    pack_w1_bf16_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1, (__bf16 *)ctx->w1p_bf + n * n1, C);
    pack_w2_bf16_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, (__bf16 *)ctx->w2p_bf + n * n2, C);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

// ---- the kernel ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(C / 32 * 64, 2) void resblock_bf16_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const __bf16 *__restrict__ w1p, const float *__restrict__ b1, const __bf16 *__restrict__ w2p,
    const float *__restrict__ b2, int L, int d, int accumulate, int ntiles, int nblk, int ablate) {
  constexpr int NW = C / 32, NT = NW * 64, NCH = C / BKC;
  static_assert(NT == 512, "bf16 kernel is built for C = 256 (8 waves)");
  constexpr int GSTRIDE = C + 8;                               // bf16 per column row of the g image (528 B)
  constexpr int XBYTES = BT * XSTRIDE * 2;                     // 26,624 B per X buffer
  constexpr int LDS_BYTES = (2 * XBYTES > BT * GSTRIDE * 2) ? 2 * XBYTES : BT * GSTRIDE * 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  // XCD-local order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch); give each XCD a contiguous run of
  // (clip, tile) work so the +-d taps and the residual patch of a clip are re-read from that XCD's L2.
  int logical;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int b = __builtin_amdgcn_readfirstlane(logical / ntiles);
  const int t0 = __builtin_amdgcn_readfirstlane((logical % ntiles) * BT);
  const float *hin_b;
  {
    const uint64_t hb = (uint64_t)(hin + (size_t)b * C * L);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    hin_b = (const float *)(((uint64_t)hi << 32) | lo);
  }
  const __amdgpu_buffer_rsrc_t hrs =
      __builtin_amdgcn_make_buffer_rsrc((void *)hin_b, 0, (int)((unsigned)C * (unsigned)L * 4u), 0x00020000);

  f32x16 acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float bv = b1[rt * C + 32 * wave + rowoff_b(r, hh)];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) acc[rt][ct][r] = bv;
    }

  // ---- X staging: thread = (column tid&127, channel octet q = tid>>7) for each of the 3 taps; 24 buffer loads issued
  // at the head of a chunk; FiLM add, zero-pad select, bf16 pack and one ds_write_b128 per tap at its tail.
  const int col = tid & (BT - 1), q8 = (tid >> 7) * 8;
  unsigned voff[3];
  bool tok[3];
#pragma unroll
  for (int tap = 0; tap < 3; tap++) {
    const int tp = t0 + col + (tap - 1) * d;
    tok[tap] = (tp >= 0) && (tp < L);
    voff[tap] = ((unsigned)min(max(tp, 0), L - 1) + (unsigned)q8 * (unsigned)L) * 4u;
  }
  float xr[3][8];
  float ptv[8];
  auto issue_loads = [&](int ch) {
#pragma unroll
    for (int e = 0; e < 8; e++) ptv[e] = pt[ch * BKC + q8 + e];
#pragma unroll
    for (int tap = 0; tap < 3; tap++)
#pragma unroll
      for (int e = 0; e < 8; e++)
        xr[tap][e] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(hrs, voff[tap], (ch * BKC + e) * L * 4, 0));
  };
  auto store_chunk = [&](unsigned char *dst) {
#pragma unroll
    for (int tap = 0; tap < 3; tap++) {
      bf16x8 pk;
#pragma unroll
      for (int e = 0; e < 8; e++) pk[e] = (__bf16)(tok[tap] ? xr[tap][e] + ptv[e] : 0.f);   // WaveNet.py:84, :26-27
      *reinterpret_cast<bf16x8 *>(dst + (col * XSTRIDE + tap * BKC + q8) * 2) = pk;
    }
  };

  issue_loads(0);
  store_chunk(lds);
  __syncthreads();
  const u32x4 *ap0 = reinterpret_cast<const u32x4 *>(w1p) + lane;

  // ---- GEMM1: per chunk 6 k-steps; A fragments (weights, this wave's 64 rows only) stream from L2 into registers in
  // sets of 3 k-steps, one set ahead of use.
  auto load_a3 = [&](bf16x8(&a)[3][2], const u32x4 *base) {
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
        a[s][rt] = __builtin_bit_cast(bf16x8, ((ablate & 4) ? ap0 : base)[(s * 2 + rt) * 64]);
  };
  auto mma3 = [&](const bf16x8(&a)[3][2], const unsigned char *xb, int rowbytes) {
#pragma unroll
    for (int s = 0; s < 3; s++) {
      bf16x8 bv[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
        bv[ct] = *reinterpret_cast<const bf16x8 *>(xb + (32 * ct) * rowbytes + s * 32);
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][rt], bv[ct], acc[rt][ct], 0, 0, 0);
    }
  };

  const u32x4 *ap = reinterpret_cast<const u32x4 *>(w1p) + (size_t)wave * NCH * 6 * 2 * 64 + lane;
  bf16x8 a0[3][2], a1[3][2];
  load_a3(a0, ap);
  const int rdoff = (j * XSTRIDE + 8 * hh) * 2;                // this lane's B-fragment byte offset inside an X buffer
#pragma unroll 1
  for (int ch = 0; ch < NCH; ch++) {
    if (ch + 1 < NCH && !(ablate & 8)) issue_loads(ch + 1);
    load_a3(a1, ap + (size_t)(ch * 6 + 3) * 128);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char *xb = lds + (ch & 1) * XBYTES + rdoff;
    mma3(a0, xb, XSTRIDE * 2);
    __builtin_amdgcn_sched_barrier(0);
    load_a3(a0, ap + (size_t)((ch + 1 < NCH ? ch + 1 : ch) * 6) * 128);
    __builtin_amdgcn_sched_barrier(0);
    mma3(a1, xb + 3 * 32, XSTRIDE * 2);
    __builtin_amdgcn_sched_barrier(0);
    if (ch + 1 < NCH) store_chunk(lds + ((ch + 1) & 1) * XBYTES);
    __syncthreads();
  }

  // ---- gate (WaveNet.py:90) -> g image [col][channel] bf16 (aliases the X buffers)
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      bf16x4 pk;
#pragma unroll
      for (int e = 0; e < 4; e++)
        pk[e] = (__bf16)((ablate & 2) ? acc[0][ct][4 * qq + e] : gate_fast(acc[0][ct][4 * qq + e], acc[1][ct][4 * qq + e]));
      *reinterpret_cast<bf16x4 *>(lds + ((32 * ct + j) * GSTRIDE + 32 * wave + 8 * qq + 4 * hh) * 2) = pk;
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  __syncthreads();

  // per-lane element offsets of this wave's 32-channel x 128-column output patch (same for h, h' and skip)
  unsigned evoff[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const int t = min(t0 + 32 * ct + j, L - 1);
    evoff[ct] = ((unsigned)(32 * wave + 4 * hh) * (unsigned)L + (unsigned)t) * 4u;
  }
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(skip + (size_t)b * C * L), 0, (int)((unsigned)C * (unsigned)L * 4u), 0x00020000);

  // ---- GEMM2 in two passes of 32 rows x 128 columns (64 accumulator VGPRs each, so the residual patch fits beside
  // them): pass 0 = res_conv rows -> h', pass 1 = skip_conv rows -> skip.  (WaveNet.py:93-97, :133)
  constexpr int NKS = C / 16;
  static_assert(NKS % 8 == 0, "GEMM2 k-steps processed in pairs of 4-step sets");
  const unsigned char *gb = lds + (j * GSTRIDE + 8 * hh) * 2;
  const float RS = 0.707106781186547524f;
  float *ho = hout + (size_t)b * C * L;
  float *sk = skip + (size_t)b * C * L;
  const float *b2l = b2, *ptl = pt;
  asm volatile("" : "+s"(b2l), "+s"(ptl));
  auto gemm2_pass = [&](auto pass_tag) {
    constexpr int pass = decltype(pass_tag)::value;
    // the values this pass adds into (h for the residual, running skip) are fetched before its GEMM and consumed
    // after it: 64 VGPRs, no exposed latency, and no float atomics (their ~1.3 TB/s chip-wide rate would cap the launch)
    float pre[4][16];
    if ((pass == 0 || accumulate) && !(ablate & 1)) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          pre[ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     pass == 0 ? hrs : srs, evoff[ct], ((r & 3) + 8 * (r >> 2)) * L * 4, 0));
    }
    f32x16 ac[4];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int c = 32 * wave + rowoff_b(r, hh);
      const float v = (pass == 0) ? b2l[c] + ptl[c] : b2l[C + c];   // u = h + part_t re-enters the residual
#pragma unroll
      for (int ct = 0; ct < 4; ct++) ac[ct][r] = v;
    }
    const u32x4 *ap2 = reinterpret_cast<const u32x4 *>(w2p) + (size_t)(wave * 2 + pass) * NKS * 64 + lane;
    bf16x8 p0[4], p1[4];
    auto load_a4 = [&](bf16x8(&a)[4], const u32x4 *base) {
#pragma unroll
      for (int s = 0; s < 4; s++) a[s] = __builtin_bit_cast(bf16x8, base[s * 64]);
    };
    auto mma4b = [&](const bf16x8(&a)[4], const unsigned char *xb) {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        bf16x8 bv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
          bv[ct] = *reinterpret_cast<const bf16x8 *>(xb + (32 * ct) * (GSTRIDE * 2) + s * 32);
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
          ac[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bv[ct], ac[ct], 0, 0, 0);
      }
    };
    load_a4(p0, ap2);
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 8) {
      load_a4(p1, ap2 + (size_t)(ks + 4) * 64);
      __builtin_amdgcn_sched_barrier(0);
      mma4b(p0, gb + ks * 32);
      __builtin_amdgcn_sched_barrier(0);
      load_a4(p0, ap2 + (size_t)(ks + 8 < NKS ? ks + 8 : ks) * 64);
      __builtin_amdgcn_sched_barrier(0);
      mma4b(p1, gb + (ks + 4) * 32);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      const int t = t0 + 32 * ct + j;
      const unsigned rbase = (unsigned)(32 * wave + 4 * hh) * (unsigned)L + (unsigned)t;
      if (t < L && !(ablate & 1)) {
        if (pass == 0) {
#pragma unroll
          for (int r = 0; r < 16; r++)
            ho[rbase + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)L] = (pre[ct][r] + ac[ct][r]) * RS;
        } else if (accumulate) {
#pragma unroll
          for (int r = 0; r < 16; r++)
            sk[rbase + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)L] = pre[ct][r] + ac[ct][r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) sk[rbase + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)L] = ac[ct][r];
        }
      }
    }
  };
  gemm2_pass(std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  gemm2_pass(std::integral_constant<int, 1>{});
}

int g_ablate_bf16 = 0;

int launch_resblock_bf16(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                         int accumulate, int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  if (C != 256) {
    set_error("AP_PREC_BF16 is built for res_channels = 256 only (got %d)", C);
    return -22;
  }
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int ntiles = (L + BT - 1) / BT;
  const int nblk = B * ntiles;
  const __bf16 *w1p = (const __bf16 *)ctx->w1p_bf + (size_t)layer * 2 * C * C * 3;
  const __bf16 *w2p = (const __bf16 *)ctx->w2p_bf + (size_t)layer * (C + S) * C;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C;
  const float *b2 = ctx->b2 + (size_t)layer * (C + S);
  resblock_bf16_kernel<256><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d, accumulate,
                                                            ntiles, nblk, g_ablate_bf16);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
