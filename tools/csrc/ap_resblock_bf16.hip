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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BT = 128;                 // time tile
constexpr int BKC = 32;                 // channels per staged chunk -> 96 K rows = 6 k-steps of 16
constexpr int XSTRIDE = 3 * BKC + 8;    // bf16 elements per column row of the X image (208 B: conflict-free b128 reads)

__device__ __forceinline__ int rowoff_b(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// tanh(a) sigmoid(b) = (1 - E) / ((1 + E)(1 + F)), E = e^(-2a), F = e^(-b); plain hardware exp2 / rcp are ample next to bf16
// operand rounding (2^-9).  a is clamped to [-16, 16] first (tanh(+-16) rounds to +-1 in fp32: no result changes), so E stays
// finite and the sign comes out of 1 - E; F may overflow to +inf, then the denominator is +inf and the gate 0, the limit.
__device__ __forceinline__ float gate_fast(float a, float b) {
  const float E = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(a, -16.0f, 16.0f) * -2.885390081777926815f);
  const float F = __builtin_amdgcn_exp2f(b * -1.442695040888963407f);
  return (1.0f - E) * __builtin_amdgcn_rcpf((1.0f + E) * (1.0f + F));
}

// (weight images: pack_w*_bf16_kernel / launch_pack_bf16 live in ap_resblock_bf16p.hip, the product file)

// ---- the kernel ---------------------------------------------------------------------------------------------
// What bounds it (tools/trace_resblock_bf16.py, s_memtime stamps per phase): the CU's one vector-memory address
// pipeline accepts a wave-wide load/store in ~16 cycles WHATEVER its width, so a tile's time is set by the NUMBER
// of VMEM instructions its 8 waves issue, not by bytes or latency.  Hence every global access here is 16 B per lane:
//   X4: the X image is staged with dwordx4 loads (4 samples x 8 channels per thread) when d % 4 == 0 and L % 4 == 0
//       (dword path otherwise: d = 1, 2);
//   E4: the GEMM2 results go through a wave-private LDS patch that turns the MFMA layout (4 rows x 1 column per
//       lane) into 1 row x 4 columns, so the residual / skip read-modify-write is dwordx4 both ways (L % 4 == 0).
constexpr bool STAGGER = false;       // waves 4-7 half a chunk behind 0-3 (measured: no gain, 6.41 vs 6.20 ms)
constexpr int PSTR = 36;                // fp32 row stride of the wave-private output patch (144 B: 16-B aligned rows)

template <int C, bool X4, bool E4, bool TRACE, int DBG = 0>
__global__ __launch_bounds__(C / 32 * 64, 2) void resblock_bf16_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const __bf16 *__restrict__ w1p, const float *__restrict__ b1, const __bf16 *__restrict__ w2p,
    const float *__restrict__ b2, int L, int d, int accumulate, int ntiles, int nblk AP_ABLATE_PARAM,
    unsigned long long *__restrict__ trace) {
  AP_ABLATE_DECL
  constexpr int NW = C / 32, NT = NW * 64, NCH = C / BKC;
  static_assert(NT == 512, "bf16 kernel is built for C = 256 (8 waves)");
  constexpr int GSTRIDE = C + 8;                               // bf16 per column row of the g image (528 B)
  constexpr int XBYTES = BT * XSTRIDE * 2;                     // 26,624 B per X buffer
  // LDS: ring of three X buffers; the g image starts at buffer 2 (buffer 1 is still read by the lagging waves when
  // the leading ones write g); then the wave-private output patches.  157,696 B of the CU's 160 KB.
  constexpr int GOFF = 2 * XBYTES;
  constexpr int GBYTES = GOFF + BT * GSTRIDE * 2;
  constexpr int PTOFF = GBYTES + (E4 ? NW * 32 * PSTR * 4 : 0);   // part_t (C floats), read back at pack time
  constexpr int LDS_BYTES = PTOFF + C * 4;
  static_assert(LDS_BYTES <= 160 * 1024 && GBYTES >= 3 * XBYTES, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  // phase timestamps of waves 0 and 7 (tools/trace_resblock_bf16.py); compiled out of the production instantiation
  auto mark = [&](int i) {
    if constexpr (TRACE) {
      if (lane == 0) trace[((size_t)blockIdx.x * NW + wave) * 16 + i] = __builtin_readcyclecounter();
    }
  };
  mark(0);
  // XCD-local order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch); give each XCD a contiguous run of
  // (clip, tile) work so the +-d taps and the residual patch of a clip are re-read from that XCD's L2.
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

  // ---- X staging.  FiLM add (WaveNet.py:84), zero padding (:26-27), bf16 pack, ds_write_b128 into the [col][k] image.
  // X4: thread = (tap = wave/2, column quad cg, channel octet oct); lane bits (low to high) cg&3, oct, cg>>2 so that a
  //     load covers whole 64-B runs per channel row and a store spreads over all 64 banks.  Waves 6, 7 repeat tap 2
  //     (a branch around them would make every later vmcnt wait the conservative one).
  // dword path: thread = (column tid&127, channel octet tid>>7) for each of the 3 taps.
  constexpr int NXR = X4 ? 32 : 24;
  float xr[NXR];
  if (tid < C) reinterpret_cast<float *>(lds + PTOFF)[tid] = pt[tid];
  unsigned voff[3];
  bool tok[3];
  int xcol, xk;                                                 // first column / k offset of this thread's LDS stores
  if constexpr (X4) {
    const int xtap = min(wave >> 1, 2);
    const int cg = ((wave & 1) * 4 + (lane >> 4)) * 4 + (lane & 3), oct = (lane >> 2) & 3;
    const int tp = t0 + 4 * cg + (xtap - 1) * d;
    tok[0] = (tp >= 0) && (tp < L);
    voff[0] = ((unsigned)min(max(tp, 0), L - 4) + (unsigned)(oct * 8) * (unsigned)L) * 4u;
    xcol = 4 * cg;
    xk = xtap * BKC + oct * 8;
  } else {
    const int col = tid & (BT - 1);
#pragma unroll
    for (int tap = 0; tap < 3; tap++) {
      const int tp = t0 + col + (tap - 1) * d;
      tok[tap] = (tp >= 0) && (tp < L);
      voff[tap] = ((unsigned)min(max(tp, 0), L - 1) + (unsigned)((tid >> 7) * 8) * (unsigned)L) * 4u;
    }
    xcol = col;
    xk = (tid >> 7) * 8;
  }
  const float *ptx = reinterpret_cast<const float *>(lds + PTOFF) + (X4 ? xk & (BKC - 1) : xk);
  auto issue_loads = [&](int ch) {
    if constexpr (X4) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        // (bit_cast the whole vector: element-wise bit_cast of the builtin's int vector is mis-folded to a splat)
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, voff[0], (ch * BKC + e) * L * 4, 0));
#pragma unroll
        for (int i = 0; i < 4; i++) xr[e * 4 + i] = v[i];
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 3; tap++)
#pragma unroll
        for (int e = 0; e < 8; e++)
          xr[tap * 8 + e] = __builtin_bit_cast(
              float, __builtin_amdgcn_raw_buffer_load_b32(hrs, voff[tap], (ch * BKC + e) * L * 4, 0));
    }
  };
  auto store_chunk = [&](unsigned char *dst, int ch) {
    float ptv[8];
    {
      const float4 p0 = *reinterpret_cast<const float4 *>(ptx + ch * BKC);
      const float4 p1 = *reinterpret_cast<const float4 *>(ptx + ch * BKC + 4);
      ptv[0] = p0.x; ptv[1] = p0.y; ptv[2] = p0.z; ptv[3] = p0.w;
      ptv[4] = p1.x; ptv[5] = p1.y; ptv[6] = p1.z; ptv[7] = p1.w;
    }
    if constexpr (X4) {
      // written so that it lowers to 32 v_add_f32, 16 v_cvt_pk_bf16_f32 (channel pairs of one sample) and 16 v_and
      // (zero padding) -- not the per-element convert + select + permute form (112 VALU).  (A v_pk_add_f32 form of the
      // adds produced wrong even samples in lanes 48-63 on this toolchain; left scalar.)
      const unsigned keep = tok[0] ? 0xffffffffu : 0u;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u32x4 pk;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++)
          pk[e2] = __builtin_bit_cast(unsigned, __builtin_convertvector(
                                                    f32x2{xr[(2 * e2) * 4 + i] + ptv[2 * e2],
                                                          xr[(2 * e2 + 1) * 4 + i] + ptv[2 * e2 + 1]}, bf16x2)) & keep;
        *reinterpret_cast<u32x4 *>(dst + ((xcol + i) * XSTRIDE + xk) * 2) = pk;
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 3; tap++) {
        bf16x8 pk;
#pragma unroll
        for (int e = 0; e < 8; e++) pk[e] = (__bf16)(tok[tap] ? xr[tap * 8 + e] + ptv[e] : 0.f);
        *reinterpret_cast<bf16x8 *>(dst + (xcol * XSTRIDE + tap * BKC + xk) * 2) = pk;
      }
    }
  };

  // X4 staging split into 8 pieces (sample i, channel-pair half hf) so that it can be placed between the MFMAs of the
  // chunk's second half instead of after them: 4 adds + 2 cvt_pk + 2 and per piece, the ds_write_b128 after a
  // sample's second piece.
  float ptv8[8];
  u32x4 pkq;
  auto pack_ptv = [&](int ch) {
    const float4 p0 = *reinterpret_cast<const float4 *>(ptx + ch * BKC);
    const float4 p1 = *reinterpret_cast<const float4 *>(ptx + ch * BKC + 4);
    ptv8[0] = p0.x; ptv8[1] = p0.y; ptv8[2] = p0.z; ptv8[3] = p0.w;
    ptv8[4] = p1.x; ptv8[5] = p1.y; ptv8[6] = p1.z; ptv8[7] = p1.w;
  };
  auto pack_piece = [&](unsigned char *dst, auto i_tag, auto hf_tag) {
    constexpr int i = decltype(i_tag)::value, hf = decltype(hf_tag)::value;
    if constexpr (DBG & 4) return;                             // timing-only: no pack
    const unsigned keep = tok[0] ? 0xffffffffu : 0u;
#pragma unroll
    for (int e2 = 2 * hf; e2 < 2 * hf + 2; e2++)
      pkq[e2] = __builtin_bit_cast(unsigned, __builtin_convertvector(
                                                 f32x2{xr[(2 * e2) * 4 + i] + ptv8[2 * e2],
                                                       xr[(2 * e2 + 1) * 4 + i] + ptv8[2 * e2 + 1]}, bf16x2)) & keep;
    if constexpr (hf == 1) *reinterpret_cast<u32x4 *>(dst + ((xcol + i) * XSTRIDE + xk) * 2) = pkq;
  };

  issue_loads(0);
  __syncthreads();                                              // part_t visible
  store_chunk(lds, 0);
  __syncthreads();
  mark(1);

  // ---- GEMM1: per chunk 6 k-steps; A fragments (weights, this wave's 64 rows only) stream from L2 into registers in
  // sets of 3 k-steps, one set ahead of use.
  auto load_a3 = [&](bf16x8(&a)[3][2], const u32x4 *base) {
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
      for (int rt = 0; rt < 2; rt++) a[s][rt] = __builtin_bit_cast(bf16x8, base[(s * 2 + rt) * 64]);
  };
  auto load_a3_loop = [&](bf16x8(&a)[3][2], const u32x4 *base) {   // DBG&1 (trace builds): no weight loads in the loop
    if constexpr (!(DBG & 1)) load_a3(a, base);
  };
  auto mma3 = [&](const bf16x8(&a)[3][2], const unsigned char *xb, int rowbytes) {
#pragma unroll
    for (int s = 0; s < 3; s++) {
      bf16x8 bv[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        if constexpr (DBG & 16) asm volatile("" : "=v"(bv[ct]));
        else bv[ct] = *reinterpret_cast<const bf16x8 *>(xb + (32 * ct) * rowbytes + s * 32);
      }
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          if constexpr (DBG & 8) {
            asm volatile("" :: "v"(a[s][rt]), "v"(bv[ct]));
          } else {
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][rt], bv[ct], acc[rt][ct], 0, 0, 0);
          }
        }
    }
  };

  const u32x4 *ap = reinterpret_cast<const u32x4 *>(w1p) + (size_t)wave * NCH * 6 * 2 * 64 + lane;
  bf16x8 a0[3][2], a1[3][2];
  const int rdoff = (j * XSTRIDE + 8 * hh) * 2;                // this lane's B-fragment byte offset inside an X buffer
  auto xbuf = [&](int ch) { return lds + (ch % 3) * XBYTES; };
  // One half-chunk = 24 MFMAs on 3 k-steps.  The CU's vector-memory path takes ~16 cycles per 16-B wave-wide load and
  // a wave issues in order, so loads sit between the MFMAs (never in a burst ahead of them), and the two waves of a
  // SIMD are half a chunk apart (waves 4-7 lag): while one runs its load-heavy half (6 weight + 10 X loads) the
  // other runs its light one (6 weight loads) and keeps the matrix pipe fed.  That needs a ring of three X buffers:
  // in iteration i the leading waves read chunk i, the lagging ones chunk i-1 then i, and everybody writes i+1.
  // vmcnt retires in issue order and a wait after a control-flow merge is the conservative one, so the loops are
  // branch-free (the last chunks re-fetch / re-store chunk 7 into an idle buffer).
  auto half_heavy = [&](const bf16x8(&use)[3][2], const unsigned char *xb, bf16x8(&nxt)[3][2], const u32x4 *nsrc,
                        int xchunk) {
    load_a3_loop(nxt, nsrc);
    if constexpr (!(DBG & 2)) issue_loads(xchunk);
    mma3(use, xb, XSTRIDE * 2);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 24; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (X4) {
        if (i < 16) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      } else {
        if (i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if ((i & 7) >= 3 && (i & 7) <= 6 && i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto half_light = [&](const bf16x8(&use)[3][2], const unsigned char *xb, bf16x8(&nxt)[3][2], const u32x4 *nsrc) {
    load_a3_loop(nxt, nsrc);
    mma3(use, xb, XSTRIDE * 2);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 24; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if ((i & 7) >= 3 && (i & 7) <= 6 && i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // X4: the light half in explicit order -- one filler group per MFMA, pinned with sched_barrier(0): next set's weight
  // fragments first (6 loads), the B fragments one k-step ahead, then the eight pack pieces of the next chunk.
  auto half_light_packed = [&](const bf16x8(&use)[3][2], const unsigned char *xb, bf16x8(&nxt)[3][2], const u32x4 *nsrc,
                               unsigned char *pdst, int pch) {
    bf16x8 bva[4], bvb[4];
    auto rdb = [&](bf16x8(&bv)[4], int s2) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        if constexpr (DBG & 16) asm volatile("" : "=v"(bv[ct]));
        else bv[ct] = *reinterpret_cast<const bf16x8 *>(xb + (32 * ct) * (XSTRIDE * 2) + s2 * 32);
      }
    };
    rdb(bva, 0);
    pack_ptv(pch);
    __builtin_amdgcn_sched_barrier(0);
    auto mf = [&](const bf16x8 &a, const bf16x8 &b, int rt, int ct) {
      if constexpr (DBG & 8) {                                 // timing-only: no MFMA (operands kept alive)
        asm volatile("" :: "v"(a), "v"(b));
      } else {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[rt][ct], 0, 0, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // k-step 0 (gaps 0-7): weight loads of the next set, B fragments of k-step 1
#pragma unroll
    for (int m = 0; m < 8; m++) {
      mf(use[0][m >> 2], bva[m & 3], m >> 2, m & 3);
      if (m < 6) {
        if constexpr (!(DBG & 1)) nxt[m >> 1][m & 1] = __builtin_bit_cast(bf16x8, nsrc[m * 64]);
      }
      if (m >= 2 && m < 6) {
        if constexpr (DBG & 16) asm volatile("" : "=v"(bvb[m - 2]));
        else bvb[m - 2] = *reinterpret_cast<const bf16x8 *>(xb + (32 * (m - 2)) * (XSTRIDE * 2) + 1 * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // k-step 1 (gaps 8-15): B fragments of k-step 2, pack pieces 0-3
#pragma unroll
    for (int m = 0; m < 8; m++) {
      mf(use[1][m >> 2], bvb[m & 3], m >> 2, m & 3);
      if (m == 1) pack_piece(pdst, I0{}, I0{});
      if (m == 3) pack_piece(pdst, I0{}, I1{});
      if (m == 5) pack_piece(pdst, I1{}, I0{});
      if (m == 7) pack_piece(pdst, I1{}, I1{});
      if (m >= 4) {
        if constexpr (DBG & 16) asm volatile("" : "=v"(bva[m - 4]));
        else bva[m - 4] = *reinterpret_cast<const bf16x8 *>(xb + (32 * (m - 4)) * (XSTRIDE * 2) + 2 * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // k-step 2 (gaps 16-23): pack pieces 4-7
#pragma unroll
    for (int m = 0; m < 8; m++) {
      mf(use[2][m >> 2], bva[m & 3], m >> 2, m & 3);
      if (m == 0) pack_piece(pdst, I2{}, I0{});
      if (m == 2) pack_piece(pdst, I2{}, I1{});
      if (m == 4) pack_piece(pdst, I3{}, I0{});
      if (m == 6) pack_piece(pdst, I3{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto set0 = [&](int ch) { return ap + (size_t)(ch * 6) * 128; };
  auto set1 = [&](int ch) { return ap + (size_t)(ch * 6 + 3) * 128; };
  if (!STAGGER || wave < NW / 2) {
    load_a3(a0, set0(0));
    if constexpr (DBG & 1) load_a3(a1, set1(0));
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch++) {
      const int nx = ch + 1 < NCH ? ch + 1 : ch;
      const unsigned char *xb = xbuf(ch) + rdoff;
      half_heavy(a0, xb, a1, set1(ch), nx);
      if (ch == 4) mark(10);
      if constexpr (X4 && !STAGGER) {
        half_light_packed(a1, xb + 3 * 32, a0, set0(nx), xbuf(ch + 1), nx);
        if (ch == 4) mark(11);
      } else {
        half_light(a1, xb + 3 * 32, a0, set0(nx));
        if (ch == 4) mark(11);
        store_chunk(xbuf(ch + 1), nx);
      }
      if (ch == 4) mark(12);
      __syncthreads();
      if (ch == 0) mark(2);
      if (ch == 3) mark(3);
      if (ch == 4) mark(13);
    }
  } else {
    issue_loads(1);
    load_a3(a0, set0(0));
    __builtin_amdgcn_sched_barrier(0);
    store_chunk(xbuf(1), 1);
    half_heavy(a0, xbuf(0) + rdoff, a1, set1(0), 2);
    __syncthreads();
    mark(2);
#pragma unroll 1
    for (int ch = 1; ch < NCH; ch++) {
      half_light(a1, xbuf(ch - 1) + rdoff + 3 * 32, a0, set0(ch));
      if (ch == 4) mark(10);
      store_chunk(xbuf(ch + 1), ch + 1 < NCH ? ch + 1 : NCH - 1);
      if (ch == 4) mark(11);
      half_heavy(a0, xbuf(ch) + rdoff, a1, set1(ch), ch + 2 < NCH ? ch + 2 : NCH - 1);
      if (ch == 4) mark(12);
      __syncthreads();
      if (ch == 3) mark(3);
      if (ch == 4) mark(13);
    }
    mma3(a1, xbuf(NCH - 1) + rdoff + 3 * 32, XSTRIDE * 2);
  }
  mark(4);

  // GEMM2's first requests go out BEFORE the gate and before anything with HBM latency or bulk: the CU's memory pipe
  // serves requests in order and vmcnt retires in order, so a bias vector or weight fragment requested after the
  // read-modify-write operands (or after the previous pass's 16 stores per wave) would hold the first MFMA until all
  // of those have gone through.  pass 0's k-loop prefetches straight into pass 1's fragments (adjacent in the packed
  // image) and pass 1's bias is fetched before pass 0's epilogue.
  constexpr int NKS = C / 16;
  static_assert(NKS % 8 == 0, "GEMM2 k-steps processed in pairs of 4-step sets");
  const float *b2l = b2, *ptl = pt;
  asm volatile("" : "+s"(b2l), "+s"(ptl));
  const u32x4 *ap2 = reinterpret_cast<const u32x4 *>(w2p) + (size_t)(wave * 2) * NKS * 64 + lane;   // pass 0, then pass 1
  bf16x8 p0[4], p1[4];
  auto load_a4 = [&](bf16x8(&a)[4], int gks) {                   // gks = k-step over both passes, clamped to the last set
    const u32x4 *base = ap2 + (size_t)(gks <= 2 * NKS - 4 ? gks : 2 * NKS - 4) * 64;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      if constexpr (!(DBG & 1024)) a[s] = __builtin_bit_cast(bf16x8, base[s * 64]);                  // timing-only: no W2 loads
    }
  };
  float4 bias[4];
  auto fetch_bias = [&](auto pass_tag) {
    constexpr int pass = decltype(pass_tag)::value;
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
  fetch_bias(std::integral_constant<int, 0>{});
  load_a4(p0, 0);
  load_a4(p1, 4);
  __builtin_amdgcn_sched_barrier(0);

  // ---- gate (WaveNet.py:90) -> g image [col][channel] bf16
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      bf16x4 pk;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if constexpr (DBG & 32) pk[e] = (__bf16)(acc[0][ct][4 * qq + e] + acc[1][ct][4 * qq + e]);   // timing-only: no gate math
        else pk[e] = (__bf16)gate_fast(acc[0][ct][4 * qq + e], acc[1][ct][4 * qq + e]);
      }
      *reinterpret_cast<bf16x4 *>(lds + GOFF + ((32 * ct + j) * GSTRIDE + 32 * wave + 8 * qq + 4 * hh) * 2) = pk;
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  __syncthreads();
  mark(5);

  // ---- GEMM2 in two passes of 32 rows x 128 columns (64 accumulator VGPRs each, so the values the pass adds into fit
  // beside them): pass 0 = res_conv rows -> h', pass 1 = skip_conv rows -> skip.  (WaveNet.py:93-97, :133)
  // Those values (h for the residual, the running skip) are fetched before the pass's GEMM and consumed after it: no
  // exposed latency, and no float atomics (their ~1.3 TB/s chip-wide rate would cap the launch).
  const unsigned char *gb = lds + GOFF + (j * GSTRIDE + 8 * hh) * 2;
  const float RS = 0.707106781186547524f;
  const __amdgpu_buffer_rsrc_t srs = clip_rsrc(skip);
  const __amdgpu_buffer_rsrc_t ors = clip_rsrc(hout);
  // output patch mapping.  E4: lane = (row lane>>3 (+8 per step), column quad lane&7); else the MFMA layout itself.
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
  }                                                             // 0x80000000: out of the clip's range -> load 0 / store dropped
  float *patch = reinterpret_cast<float *>(lds + GBYTES) + wave * 32 * PSTR;
  auto gemm2_pass = [&](auto pass_tag) {
    constexpr int pass = decltype(pass_tag)::value;
    float pre[4][16];
    if ((pass == 0 || accumulate) && !(ablate & 1) && !(DBG & 128)) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        if constexpr (E4) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const f32x4 v = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(pass == 0 ? hrs : srs, evoff[ct], 8 * p * L * 4, 2));   // nt: once-touched skip rows; the residual re-read of h hits or passes without allocating
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
    }
    f32x16 ac[4];
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
    auto mma4b = [&](const bf16x8(&a)[4], const unsigned char *xb) {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        bf16x8 bv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          if constexpr (DBG & 2048) asm volatile("" : "=v"(bv[ct]));                                 // timing-only: no g reads
          else bv[ct] = *reinterpret_cast<const bf16x8 *>(xb + (32 * ct) * (GSTRIDE * 2) + s * 32);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          if constexpr (DBG & 64) asm volatile("" :: "v"(a[s]), "v"(bv[ct]));                      // timing-only: no GEMM2 MFMA
          else ac[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bv[ct], ac[ct], 0, 0, 0);
        }
      }
    };
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 8) {
      mma4b(p0, gb + ks * 32);
      __builtin_amdgcn_sched_barrier(0);
      load_a4(p0, pass * NKS + ks + 8);
      __builtin_amdgcn_sched_barrier(0);
      mma4b(p1, gb + (ks + 4) * 32);
      __builtin_amdgcn_sched_barrier(0);
      load_a4(p1, pass * NKS + ks + 12);
      __builtin_amdgcn_sched_barrier(0);
    }
    mark(6 + 2 * pass);
    if (pass == 0) {
      fetch_bias(std::integral_constant<int, 1>{});              // ahead of this pass's stores
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool add = (pass == 0) || accumulate;
    const float scale = pass == 0 ? RS : 1.0f;
    if (!(ablate & 1)) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        if constexpr (E4) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if constexpr (!(DBG & 512)) patch[rowoff_b(r, hh) * PSTR + j] = ac[ct][r];
          }
#pragma unroll
          for (int p = 0; p < 4; p++) {
            float4 v;
            if constexpr (DBG & 512) { v.x = ac[ct][4 * p]; v.y = ac[ct][4 * p + 1]; v.z = ac[ct][4 * p + 2]; v.w = ac[ct][4 * p + 3]; }   // timing-only: no LDS transpose
            else v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * p) * PSTR + 4 * (lane & 7));
            f32x4 o;
            o[0] = ((add ? pre[ct][4 * p + 0] : 0.f) + v.x) * scale;
            o[1] = ((add ? pre[ct][4 * p + 1] : 0.f) + v.y) * scale;
            o[2] = ((add ? pre[ct][4 * p + 2] : 0.f) + v.z) * scale;
            o[3] = ((add ? pre[ct][4 * p + 3] : 0.f) + v.w) * scale;
            // row step in the VGPR offset, soffset = 0: a >8-byte buffer store with an SGPR soffset reads its data
            // late, and the compiler here does not guard the next write of those VGPRs (observed: torn y lanes)
            if constexpr (DBG & 256) asm volatile("" :: "v"(o));                                     // timing-only: no stores
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), pass == 0 ? ors : srs,
                                                        evoff[ct] + (unsigned)(8 * p * L * 4), 0, 2);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++)
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(unsigned, ((add ? pre[ct][r] : 0.f) + ac[ct][r]) * scale), pass == 0 ? ors : srs,
                evoff[ct], ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
        }
      }
    }
  };
  gemm2_pass(std::integral_constant<int, 0>{});
  mark(7);
  __builtin_amdgcn_sched_barrier(0);
  gemm2_pass(std::integral_constant<int, 1>{});
  mark(9);
}

#ifdef AP_TOOLS
int g_dbg_bf16 = 0;                           // ap_debug_bf16_dbg: timing-only template variants (bit 1 no weight loads,
                                              // 2 no X loads, 4 no pack, 8 no MFMA, 16 no B-fragment LDS reads in GEMM1)
int g_ablate_bf16 = 0;
unsigned long long *g_trace_bf16 = nullptr;   // ap_debug_trace: device buffer of nblk x 2 x 16 timestamps, or null
#endif

int launch_resblock_bf16(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                         int accumulate, int B, int L, hipStream_t st, const UbArgs *ub, void *gout, void *fout) {
  const int C = ctx->C, S = ctx->S;
  if (ub || gout) return launch_resblock_bf16p(ctx, layer, hin, pt, hout, skip, accumulate, B, L, st, ub, gout, fout);
  if (C != 256) {
    set_error("AP_PREC_BF16 is built for res_channels = 256 only (got %d)", C);
    return -22;
  }
  // dispatch: the persistent kernel (ap_resblock_bf16p.hip) where it serves the shape, else this file's per-tile kernel
  // (any d, any L).  Tools builds can force the per-tile kernel for A/B timing (ap_debug_bf16_dbg bit 0x1000).
#ifdef AP_TOOLS
  const bool force_tile = (g_dbg_bf16 & 0x1000) != 0;
  if (!force_tile && !g_trace_bf16)
#endif
  {
#ifdef AP_TOOLS
    // tools builds only: the one-wave-per-SIMD experiment (ap_resblock_bf16w.hip; bit-identical, 20 % slower: DESIGN.md 3.4)
    // serves the launch when bit 0x20000 asks for it -- tools/ab_bf16w.py, ablate_bf16w.py, trace_resblock_bf16w.py
    if (g_dbg_bf16 & 0x20000) {
      const int rcw = launch_resblock_bf16w(ctx, layer, hin, pt, hout, skip, accumulate, B, L, st);
      if (rcw != 1) return rcw;                                  // 1: shape not served there -> the eight-wave persistent kernel
    }
#endif
    const int rc = launch_resblock_bf16p(ctx, layer, hin, pt, hout, skip, accumulate, B, L, st);
    if (rc != 1) return rc;                                      // 1: shape not served there (L % 4, C, S) -> per-tile kernel
  }
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int ntiles = (L + BT - 1) / BT;
  const int nblk = B * ntiles;
  const __bf16 *w1p = (const __bf16 *)ctx->w1p_bf + (size_t)layer * 2 * C * C * 3;
  const __bf16 *w2p = (const __bf16 *)ctx->w2p_bf + (size_t)layer * (C + S) * C;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C;
  const float *b2 = ctx->b2 + (size_t)layer * (C + S);
  const bool e4 = (L % 4 == 0) && L >= 4, x4 = e4 && (d % 4 == 0);
#define AP_BF16_LAUNCH(X4, E4, TR)                                                                                  \
  resblock_bf16_kernel<256, X4, E4, TR><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d, \
                                                                        accumulate, ntiles, nblk                     \
                                                                        AP_ABLATE_ARG(g_ablate_bf16), AP_TRACE_BUF)
#ifdef AP_TOOLS
#define AP_TRACE_BUF g_trace_bf16
  if (!g_trace_bf16 && x4 && (g_dbg_bf16 & 0xfff)) {
#define AP_DBG_CASE(D) case D: resblock_bf16_kernel<256, true, true, false, D><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d, accumulate, ntiles, nblk, g_ablate_bf16, nullptr); break;
    switch (g_dbg_bf16 & 0xfff) {
      AP_DBG_CASE(1) AP_DBG_CASE(2) AP_DBG_CASE(3) AP_DBG_CASE(4) AP_DBG_CASE(7) AP_DBG_CASE(8) AP_DBG_CASE(16) AP_DBG_CASE(23) AP_DBG_CASE(31)
      AP_DBG_CASE(31 + 32) AP_DBG_CASE(31 + 64) AP_DBG_CASE(31 + 128) AP_DBG_CASE(31 + 256) AP_DBG_CASE(31 + 384) AP_DBG_CASE(31 + 512)
      AP_DBG_CASE(31 + 1024) AP_DBG_CASE(31 + 2048) AP_DBG_CASE(31 + 64 + 1024 + 2048) AP_DBG_CASE(31 + 384 + 512)
      AP_DBG_CASE(31 + 32 + 384 + 512) AP_DBG_CASE(4095)
      default: set_error("no such DBG instantiation"); return -22;
    }
#undef AP_DBG_CASE
  }
  else if (g_trace_bf16 && x4) {
    switch ((g_ablate_bf16 >> 6) & 3) {
      case 1: resblock_bf16_kernel<256, true, true, true, 1><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d, accumulate, ntiles, nblk, g_ablate_bf16, g_trace_bf16); break;
      case 2: resblock_bf16_kernel<256, true, true, true, 2><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d, accumulate, ntiles, nblk, g_ablate_bf16, g_trace_bf16); break;
      case 3: resblock_bf16_kernel<256, true, true, true, 3><<<(unsigned)nblk, 512, 0, st>>>(hin, pt, hout, skip, w1p, b1, w2p, b2, L, d, accumulate, ntiles, nblk, g_ablate_bf16, g_trace_bf16); break;
      default: AP_BF16_LAUNCH(true, true, true);
    }
  }
  else
#else
#define AP_TRACE_BUF nullptr
#endif
  if (x4) AP_BF16_LAUNCH(true, true, false);
  else if (e4) AP_BF16_LAUNCH(false, true, false);
  else AP_BF16_LAUNCH(false, false, false);
#undef AP_BF16_LAUNCH
#undef AP_TRACE_BUF
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
