// AP_PREC_BF16, one wave per SIMD: fused Residual_block.forward (WaveNet.py:75-97) with bf16 MFMA operands, fp32 accumulate,
// fp32 activations in HBM.  Same arithmetic, same k order and the same packed weight images as ap_resblock_bf16p.hip (the
// outputs are bit-identical: tests/test_gpu_parity.py::test_bf16_block_kernels_are_bit_identical), a different machine shape:
//
//   * 256 threads = FOUR waves, one per SIMD, each with the whole 512-entry register file of its SIMD: a wave owns 128 GEMM1
//     rows (the tanh and sigmoid rows of 64 gate channels) x 128 columns = 256 accumulator registers (the AGPR half), and has
//     the 256 arch VGPRs for operand rings, staging and the read-modify-write operands of the epilogue.
//   * every B fragment read from LDS feeds FOUR MFMAs instead of two (half the LDS read traffic of the eight-wave kernels:
//     768 KB instead of 1.5 MB per tile in GEMM1, 512 KB instead of 1 MB in GEMM2), every barrier has four participants,
//     and no two waves of a SIMD compete for its issue port or its matrix pipe: the instruction stream of a wave IS the
//     schedule of its SIMD, written out below (explicit order, pinned with sched_barrier).
//   * the spare registers carry what the eight-wave kernel had to burst: a fragment ring of eight k-steps (128 registers),
//     the residual's h patch requested during the gate, the running skip rows into the registers the dead GEMM1 accumulators free.
//
// RESULT (profiles/r3_bf16w_one_wave_per_simd_experiment.txt): bit-identical to the product kernel and 20 % SLOWER
// (6.1-6.4 ms against 5.2 per 256-clip launch).  A chunk of 96 MFMAs per wave takes 3.7 k cycles bare, 4.4 k with its 96 KB of weight
// fragments, 5.2 k when it also re-requests 48 KB of cache-resident X rows and 6-7 k when those rows are new -- with a fragment ring of
// 3, 4, 6 or 8 k-steps, with 5 or 11 k-steps between an X request and its pack, with the requests in bursts or one per MFMA gap:
// the time the CU's one address pipeline (64 B/clk, all waves) needs for a wave's memory instructions ADDS to its MFMA time,
// because the wave that waits at that pipeline's door is the only one that could issue its SIMD's MFMAs.  Two waves per SIMD hide
// part of it (the partner issues meanwhile); separate loader waves would hide all of it, and the register file has no room for
// them (the accumulators of the 512 x 128 tile are half of it, the fragment rings want registers, not LDS).  Tools library only.
//
// Layouts (unchanged): X image [column][k] bf16 with 208-B rows and the 32-byte parity swizzle, g image [column][channel]
// with 528-B rows, wave-private 32 x 32 fp32 output patch.  Weight images: wave v of this kernel reads the fragments that
// waves 2v and 2v+1 of the eight-wave kernels read (ap_kernels.hip: pack_bf16).
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PT_ = 128;                 // time tile
constexpr int KC_ = 32;                  // channels per staged chunk -> 96 K rows = 6 k-steps of 16
constexpr int XS_ = 3 * KC_ + 8;         // bf16 per column row of the X image (208 B)
constexpr int GS_ = 256 + 8;             // bf16 per column row of the g image (528 B)
constexpr int PS_ = 32;                  // fp32 per row of the wave-private output patch (128 B)


// the gate of ap_resblock_bf16p.hip, element for element (bit-identical kernels)
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

using I0 = std::integral_constant<int, 0>;
using I1 = std::integral_constant<int, 1>;
using I2 = std::integral_constant<int, 2>;
using I3 = std::integral_constant<int, 3>;

}  // namespace

#ifdef AP_TOOLS
__device__ unsigned long long *g_wtrace = nullptr;               // DBG 2048: [workgroup][wave][64] s_memtime stamps of one tile
#endif

// DBG (tools builds only; outputs wrong by construction): 1 no weight loads in GEMM1's loop, 2 no X loads, 4 no pack,
// 8 no GEMM1 MFMA, 32 no gate math, 64 no GEMM2 MFMA, 128 no read-modify-write loads, 256 no stores, 2048 phase stamps.
template <int DBG>
__global__ __launch_bounds__(256, 1) void resblock_bf16w_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const void *__restrict__ wbase, unsigned wbytes, unsigned w1_off, unsigned w2_off,        // bf16 weight images (one slab)
    const void *__restrict__ bbase, unsigned b1_off, unsigned b2_off,                         // fp32 bias vectors (one slab)
    int L, int d, int accumulate, int ntiles, int nblk) {
  constexpr int C = 256, NW = 4, NCH = C / KC_, NKS = C / 16;
  constexpr int NT = 2;                                         // nt cache policy on the once-touched streams (see bf16p)
  constexpr int XBYTES = PT_ * XS_ * 2;                         // 26,624 B per X buffer, two buffers
  constexpr int GOFF = 2 * XBYTES;
  constexpr int POFF = GOFF + PT_ * GS_ * 2;                    // output patches: 4 waves x 32 x 32 fp32
  constexpr int PTOFF = POFF + NW * 32 * PS_ * 4;               // part_t (C floats)
  constexpr int BOFF = PTOFF + C * 4;                           // b1 (2C floats), b2 (2C floats)
  constexpr int LDS_BYTES = BOFF + 4 * C * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3; owns the rows of the eight-wave kernels' waves 2w, 2w+1
  const int j = lane & 31, hh = lane >> 5;

  // ---- tile walk (as ap_resblock_bf16p.hip): XCD-local runs of (clip, tile) work, strided order inside a clip
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
  const int wstep = __builtin_amdgcn_readfirstlane(min(max(d / PT_, 1), 16));
  const int wq = ntiles / wstep, wrem = ntiles % wstep;
  auto tile_bt = [&](int tile, int &b, int &t0) {
    b = __builtin_amdgcn_readfirstlane(tile / ntiles);
    int p = tile % ntiles;
    if (wstep > 1) {
      const int cut = wrem * (wq + 1);
      const int r = p < cut ? p / (wq + 1) : wrem + (p - cut) / wq;
      const int k = p < cut ? p % (wq + 1) : (p - cut) % wq;
      p = r + k * wstep;
    }
    t0 = __builtin_amdgcn_readfirstlane(p * PT_);
  };

  if (tid < C) reinterpret_cast<float *>(lds + PTOFF)[tid] = pt[tid];
  {
    const unsigned char *bb = static_cast<const unsigned char *>(bbase);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      reinterpret_cast<float *>(lds + BOFF)[r * 256 + tid] = reinterpret_cast<const float *>(bb + b1_off)[r * 256 + tid];
      reinterpret_cast<float *>(lds + BOFF)[2 * C + r * 256 + tid] = reinterpret_cast<const float *>(bb + b2_off)[r * 256 + tid];
    }
  }

  // ---- X staging.  A unit = (column quad cg of 32, channel quad q8 of 8) for all three taps: per chunk and thread 3 x 4
  // loads of 16 B (lane bits low to high: q8, cg & 7 -- one load instruction covers eight channel rows x 128 B), FiLM add
  // (WaveNet.py:84), bf16 (RNE), zero padding (:26-27) as an AND with the tap's in-range mask, 3 x 4 ds_write_b64 (four
  // channels of one column; the sixteen lanes of a ds_write_b64 group cover all 32 banks).  d % 4 == 0 and L % 4 == 0: a
  // column quad is inside the clip or outside it as a whole, the address is clamped.
  const int q8 = lane & 7, cg = wave * 8 + (lane >> 3);
  const unsigned xwb = (unsigned)(4 * cg * (XS_ * 2) + ((q8 * 8) ^ ((__builtin_popcount(cg & 7) & 1) << 5)));   // + tap * 64 + sample * 208
  // Two register sets: chunk c lives in set c & 1 from its request (during chunk c - 3) to its pack (during chunk c - 1) -- eleven
  // k-steps of sixteen MFMAs between a request and its first use, where one set allowed five: with ONE wave per SIMD a k-step
  // takes half the time it took the eight-wave kernels, and an HBM miss does not.
  constexpr bool TWO = false;                                    // two staging register sets (chunk c + 3 requested during chunk c) or one (c + 2)
  float xrA[3][4][4], xrB[3][4][4];                              // [tap][channel of the quad][sample]; xrB unused unless TWO
  unsigned xvoff[3], xkeep[3];
  auto x_geom = [&](int t0) {
#pragma unroll
    for (int T = 0; T < 3; T++) {
      const int tp = t0 + 4 * cg + (T - 1) * d;
      xkeep[T] = (tp >= 0 && tp < L) ? 0xffffffffu : 0u;
      xvoff[T] = ((unsigned)min(max(tp, 0), L - 4) + (unsigned)(q8 * 4) * (unsigned)L) * 4u;
    }
  };
  auto issue_x_tap = [&](const __amdgpu_buffer_rsrc_t &rs, int ch, auto set_tag, auto t_tag) {
    constexpr int T = decltype(t_tag)::value;
    auto &xr = (TWO && decltype(set_tag)::value) ? xrB : xrA;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xvoff[T], (ch * KC_ + e) * L * 4, 0));
#pragma unroll
      for (int i = 0; i < 4; i++) xr[T][e][i] = v[i];
    }
  };
  auto issue_x1 = [&](const __amdgpu_buffer_rsrc_t &rs, int ch, auto set_tag, auto t_tag, auto e_tag) {
    constexpr int T = decltype(t_tag)::value, e = decltype(e_tag)::value;
    auto &xr = (TWO && decltype(set_tag)::value) ? xrB : xrA;
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xvoff[T], (ch * KC_ + e) * L * 4, 0));
#pragma unroll
    for (int i = 0; i < 4; i++) xr[T][e][i] = v[i];
  };
  auto issue_x = [&](const __amdgpu_buffer_rsrc_t &rs, int ch, auto set_tag) {
    issue_x_tap(rs, ch, set_tag, I0{});
    issue_x_tap(rs, ch, set_tag, I1{});
    issue_x_tap(rs, ch, set_tag, I2{});
  };
  float ptv[4];
  auto pack_ptv = [&](int ch) {
    const float4 p0 = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(lds + PTOFF) + q8 * 4 + ch * KC_);
    ptv[0] = p0.x; ptv[1] = p0.y; ptv[2] = p0.z; ptv[3] = p0.w;
  };
  // one twelfth of a chunk's staging: sample i of tap T -- 4 adds, 2 cvt_pk, 2 and, 1 ds_write_b64
  auto pack_piece = [&](unsigned char *dst, auto set_tag, auto t_tag, auto i_tag) {
    constexpr int T = decltype(t_tag)::value, i = decltype(i_tag)::value;
    auto &xr = (TWO && decltype(set_tag)::value) ? xrB : xrA;
    unsigned pk[2];
#pragma unroll
    for (int e2 = 0; e2 < 2; e2++)
      pk[e2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{xr[T][2 * e2][i] + ptv[2 * e2],
                                                                          xr[T][2 * e2 + 1][i] + ptv[2 * e2 + 1]}, bf16x2)) & xkeep[T];
    *reinterpret_cast<uint2 *>(dst + xwb + T * (2 * KC_) + i * (XS_ * 2)) = make_uint2(pk[0], pk[1]);
  };
  auto pack_tap = [&](unsigned char *dst, auto set_tag, auto t_tag) {
    pack_piece(dst, set_tag, t_tag, I0{}); pack_piece(dst, set_tag, t_tag, I1{});
    pack_piece(dst, set_tag, t_tag, I2{}); pack_piece(dst, set_tag, t_tag, I3{});
  };

  // ---- weight fragment streams, L2 -> registers (buffer loads, fragment index in the scalar offset)
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc(wbase, wbytes);
  const unsigned lane16 = (unsigned)lane * 16u;
  // GEMM1 image [wave8][chunk][kstep 6][rowtile 2][lane][8 bf16]; f = 2 u + rt selects (eight-wave id 2 wave + u, row tile rt)
  auto ld_w1 = [&](int frag, int f) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                          wrs, lane16 + (unsigned)((f & 1) * 1024), w1_off + ((2 * wave + (f >> 1)) * NCH * 12 + frag * 2) * 1024, 0));
  };
  // GEMM2 image [wave8][rowtile 2 = pass][kstep 16][lane][8 bf16]
  auto ld_w2 = [&](int pass, int ks, int u) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                          wrs, lane16, w2_off + (((2 * wave + u) * 2 + pass) * NKS + ks) * 1024, 0));
  };
  const int rdoff = (j * XS_ + 8 * hh) * 2;                     // this lane's B-fragment byte offset inside an X buffer
  const int rdsw = (__builtin_popcount((j >> 2) & 7) & 1) * 32;
  const unsigned char *gb = lds + GOFF + (j * GS_ + 8 * hh) * 2;
  float *patch = reinterpret_cast<float *>(lds + POFF) + wave * 32 * PS_;
  const float RS = 0.707106781186547524f;

  // ---- first tile
  int b_cur, t0_cur;
  tile_bt(t_first, b_cur, t0_cur);
  __amdgpu_buffer_rsrc_t hrs = clip_rsrc(hin, b_cur);
  x_geom(t0_cur);
  constexpr int RING = 8;                                        // fragment ring depth in k-steps (4 fragments each)
  bf16x8 w[RING][4];
  auto tile_head = [&]() {
#pragma unroll
    for (int ks = 0; ks < RING; ks++)
#pragma unroll
      for (int f = 0; f < 4; f++) w[ks][f] = ld_w1(ks, f);
    if constexpr (TWO) issue_x(hrs, 1, I1{});                    // chunk 1: packed in chunk 0 (a short lead, once per tile)
    pack_ptv(0);
    pack_tap(lds, I0{}, I0{}); pack_tap(lds, I0{}, I1{}); pack_tap(lds, I0{}, I2{});
    issue_x(hrs, TWO ? 2 : 1, I0{});                             // chunk 2 (1): packed in chunk 1 (0)
  };
  issue_x(hrs, 0, I0{});
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
        if (lane == 0) g_wtrace[((size_t)blockIdx.x * NW + wave) * 64 + i] = t;
        if (i == 0 || i == 30) {
          unsigned long long rt;
          asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
          if (lane == 0) g_wtrace[((size_t)blockIdx.x * NW + wave) * 64 + 40 + (i == 30)] = rt;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#endif
  };
#pragma unroll 1
  for (int tile = t_first; tile < t_end; tile += t_step, tile_iter++) {
    mark(0);
    const int t0 = t0_cur;
    const int ntile = tile + t_step;
    // ================================================ GEMM1 =========================================================
    f32x16 acc[4][4];                                            // [f = 2 u + rt][column tile]
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + ((f & 1) * C + 32 * (2 * wave + (f >> 1)) + 8 * q + 4 * hh) * 4);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          acc[f][ct][4 * q + 0] = bv4[0];
          acc[f][ct][4 * q + 1] = bv4[1];
          acc[f][ct][4 * q + 2] = bv4[2];
          acc[f][ct][4 * q + 3] = bv4[3];
        }
      }
    mark(1);
    __syncthreads();
    mark(2);

    auto rdb = [&](bf16x8 &dst, const unsigned char *xbe, const unsigned char *xbo, int ct, int ks) {
      dst = *reinterpret_cast<const bf16x8 *>(((ks & 1) ? xbo : xbe) + (32 * ct) * (XS_ * 2) + ks * 32);
    };
    // One chunk = six k-steps of sixteen MFMAs, column-tile-major: the four MFMAs of a column tile share its B fragment, which
    // is re-read for the next k-step right behind them.  A k-step's four weight fragments are replaced behind their last
    // MFMAs (column tile 3) by those of the k-step RING on.  Chunk c packs chunk c + 1 (tap T in k-step T, one sample per
    // column tile) out of register set (c + 1) & 1 and requests chunk c + 3 into the same set, tap T one k-step behind its
    // pack and behind that k-step's fragment requests.
    auto chunk = [&](const unsigned char *xbe, const unsigned char *xbo, int ch, unsigned char *pdst, auto set_tag, auto kind_tag, auto ph_tag) {
      constexpr int KIND = decltype(kind_tag)::value;            // 0: pack + request, 1: pack only, 2: neither (last chunk)
      constexpr bool LAST = KIND == 2, WITH_X = KIND == 0;
      constexpr int PH = decltype(ph_tag)::value;                // ring slot of this chunk's k-step 0 = (6 ch) mod RING
      bf16x8 bv[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) rdb(bv[ct], xbe, xbo, ct, 0);
      if constexpr (!LAST) pack_ptv(ch + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 6; ks++) {
        const bool reload = !(DBG & 1) && (ks + RING < 6 || !LAST);
        const int nfrag = ks + RING < 6 ? ch * 6 + ks + RING : (ch + 1) * 6 + ks + RING - 6;
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
#pragma unroll
          for (int f = 0; f < 4; f++) {
            if constexpr (DBG & 8) asm volatile("" ::"v"(w[(ks + PH) % RING][f]), "v"(bv[ct]));
            else acc[f][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(ks + PH) % RING][f], bv[ct], acc[f][ct], 0, 0, 0);
            if (ct == 3 && reload) w[(ks + PH) % RING][f] = ld_w1(nfrag, f);
            if constexpr (!LAST && !(DBG & 4)) {
              if (f == 1 && ks <= 2) {
                if (ks == 0) { if (ct == 0) pack_piece(pdst, set_tag, I0{}, I0{}); if (ct == 1) pack_piece(pdst, set_tag, I0{}, I1{});
                               if (ct == 2) pack_piece(pdst, set_tag, I0{}, I2{}); if (ct == 3) pack_piece(pdst, set_tag, I0{}, I3{}); }
                if (ks == 1) { if (ct == 0) pack_piece(pdst, set_tag, I1{}, I0{}); if (ct == 1) pack_piece(pdst, set_tag, I1{}, I1{});
                               if (ct == 2) pack_piece(pdst, set_tag, I1{}, I2{}); if (ct == 3) pack_piece(pdst, set_tag, I1{}, I3{}); }
                if (ks == 2) { if (ct == 0) pack_piece(pdst, set_tag, I2{}, I0{}); if (ct == 1) pack_piece(pdst, set_tag, I2{}, I1{});
                               if (ct == 2) pack_piece(pdst, set_tag, I2{}, I2{}); if (ct == 3) pack_piece(pdst, set_tag, I2{}, I3{}); }
              }
            }
            if constexpr (WITH_X && !(DBG & 2)) {
              if (f == 2 && ks >= 1 && ks <= 3) {               // one request per MFMA gap, never a burst: a wave that waits at
                const int chx = min(ch + (TWO ? 3 : 2), NCH - 1); // the memory pipe's door issues no MFMA either
                if (ks == 1) { if (ct == 0) issue_x1(hrs, chx, set_tag, I0{}, I0{}); if (ct == 1) issue_x1(hrs, chx, set_tag, I0{}, I1{});
                               if (ct == 2) issue_x1(hrs, chx, set_tag, I0{}, I2{}); if (ct == 3) issue_x1(hrs, chx, set_tag, I0{}, I3{}); }
                if (ks == 2) { if (ct == 0) issue_x1(hrs, chx, set_tag, I1{}, I0{}); if (ct == 1) issue_x1(hrs, chx, set_tag, I1{}, I1{});
                               if (ct == 2) issue_x1(hrs, chx, set_tag, I1{}, I2{}); if (ct == 3) issue_x1(hrs, chx, set_tag, I1{}, I3{}); }
                if (ks == 3) { if (ct == 0) issue_x1(hrs, chx, set_tag, I2{}, I0{}); if (ct == 1) issue_x1(hrs, chx, set_tag, I2{}, I1{});
                               if (ct == 2) issue_x1(hrs, chx, set_tag, I2{}, I2{}); if (ct == 3) issue_x1(hrs, chx, set_tag, I2{}, I3{}); }
              }
            }
            if (f == 3 && ks < 5) rdb(bv[ct], xbe, xbo, ct, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    auto xb = [&](int ch) { return lds + (ch & 1) * XBYTES + rdoff + rdsw; };
    static_assert(NCH == 8, "chunk schedule below is written for eight chunks");
    using P0 = std::integral_constant<int, 0>;
    using P6 = std::integral_constant<int, 6 % RING>;
    using P4 = std::integral_constant<int, 12 % RING>;
    using P2 = std::integral_constant<int, 18 % RING>;
#define AP_W_CHUNK(CH, PDST, SET, KIND, PH, M)                   \
    chunk(xb(CH), xb(CH) - 2 * rdsw, CH, PDST, SET{}, KIND{}, PH{}); \
    mark(M);                                                     \
    __syncthreads();                                             \
    mark(M + 1);
    AP_W_CHUNK(0, lds + XBYTES, I1, I0, P0, 3)
    AP_W_CHUNK(1, lds, I0, I0, P6, 5)
    AP_W_CHUNK(2, lds + XBYTES, I1, I0, P4, 7)
    AP_W_CHUNK(3, lds, I0, I0, P2, 9)
    AP_W_CHUNK(4, lds + XBYTES, I1, I0, P0, 11)
    AP_W_CHUNK(5, lds, I0, I0, P6, 13)
    AP_W_CHUNK(6, lds + XBYTES, I1, I1, P4, 15)
#undef AP_W_CHUNK
    chunk(xb(7), xb(7) - 2 * rdsw, 7, nullptr, I0{}, I2{}, P2{});
    mark(17);

    // ================================================ gate ==========================================================
    // Operand requests first (the CU's memory pipe serves in order): GEMM2's first weight fragments, then the residual's h
    // patch and the running skip rows -- all 256 operand registers of the two epilogues, requested while the gate runs.
    unsigned evoff[4];                                          // E4 mapping: lane = (row lane>>3 (+8 per step), column quad lane&7)
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      const int t = t0 + 32 * ct + 4 * (lane & 7);
      evoff[ct] = t < L ? ((unsigned)(64 * wave + (lane >> 3)) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
    }                                                           // 0x80000000: outside the clip -> loads 0, store dropped
    const unsigned ustep = 32u * (unsigned)L * 4u;              // rows of the second eight-wave id (u = 1)
    constexpr int R2 = 4;                                       // GEMM2 fragment ring depth in k-steps (2 fragments each)
    bf16x8 w2r[R2][2];
#pragma unroll
    for (int ks = 0; ks < R2; ks++)
#pragma unroll
      for (int u = 0; u < 2; u++) w2r[ks][u] = ld_w2(0, ks, u);
    const __amdgpu_buffer_rsrc_t srs = clip_rsrc(skip, b_cur);
    const __amdgpu_buffer_rsrc_t ors = clip_rsrc(hout, b_cur);
    float pre[2][4][16];                                        // pass 0: the h patch the residual adds; pass 1: the running skip rows
    auto load_pre = [&](const __amdgpu_buffer_rsrc_t &rs, auto u_tag, auto ct_tag, bool zero) {
      constexpr int u = decltype(u_tag)::value, ct = decltype(ct_tag)::value;
      if ((DBG & 128) || zero) {
#pragma unroll
        for (int r = 0; r < 16; r++) pre[u][ct][r] = 0.f;
      } else {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, evoff[ct] + (u ? ustep : 0u), 8 * p * L * 4, NT));
#pragma unroll
          for (int i = 0; i < 4; i++) pre[u][ct][4 * p + i] = v[i];
        }
      }
    };
    auto gate_ct = [&](auto u_tag, auto ct_tag) {
      constexpr int u = decltype(u_tag)::value, ct = decltype(ct_tag)::value;
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        unsigned pk[2];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 a2 = {acc[2 * u][ct][4 * qq + e], acc[2 * u][ct][4 * qq + e + 1]};
          const f32x2 b2 = {acc[2 * u + 1][ct][4 * qq + e], acc[2 * u + 1][ct][4 * qq + e + 1]};
          const f32x2 g2 = (DBG & 32) ? a2 + b2 : gate_fast2(a2, b2);
          pk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(g2, bf16x2));
        }
        *reinterpret_cast<uint2 *>(lds + GOFF + ((32 * ct + j) * GS_ + 32 * (2 * wave + u) + 8 * qq + 4 * hh) * 2) = make_uint2(pk[0], pk[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    const bool noacc = !accumulate;
    // the first half of the h patch is requested while the gate runs (what fits beside the accumulators' VGPR copies)
    load_pre(hrs, I0{}, I0{}, false);
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I0{}, I0{});
    load_pre(hrs, I0{}, I1{}, false);
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I0{}, I1{});
    load_pre(hrs, I0{}, I2{}, false);
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I0{}, I2{});
    load_pre(hrs, I0{}, I3{}, false);
    __builtin_amdgcn_sched_barrier(0);
    gate_ct(I0{}, I3{});
    gate_ct(I1{}, I0{});
    gate_ct(I1{}, I1{});
    gate_ct(I1{}, I2{});
    gate_ct(I1{}, I3{});
    mark(18);
    __syncthreads();
    mark(19);
    // the GEMM1 accumulators are dead from here
    load_pre(hrs, I1{}, I0{}, false); load_pre(hrs, I1{}, I1{}, false);
    load_pre(hrs, I1{}, I2{}, false); load_pre(hrs, I1{}, I3{}, false);
    __builtin_amdgcn_sched_barrier(0);

    // ================================================ GEMM2 =========================================================
    // two passes of 64 rows x 128 columns: pass 0 = res_conv rows -> h', pass 1 = skip_conv rows -> skip (WaveNet.py:93-97,:133)
    auto gemm2_loop = [&](f32x16(&ac)[2][4], auto pass_tag) {
      constexpr int pass = decltype(pass_tag)::value;
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int c = 32 * (2 * wave + u) + 8 * q + 4 * hh;
          const f32x4 v4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + (2 * C + pass * C + c) * 4);
          float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
          if (pass == 0) {                                       // u = h + part_t re-enters the residual (alias semantics)
            const float4 pv = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(lds + PTOFF) + c);
            v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
          }
#pragma unroll
          for (int ct = 0; ct < 4; ct++) {
            ac[u][ct][4 * q + 0] = v.x;
            ac[u][ct][4 * q + 1] = v.y;
            ac[u][ct][4 * q + 2] = v.z;
            ac[u][ct][4 * q + 3] = v.w;
          }
        }
      bf16x8 bq[4];                                              // g-image B fragments: re-read for the next k-step behind their last MFMA
#pragma unroll
      for (int ct = 0; ct < 4; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * (GS_ * 2));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
#pragma unroll
          for (int u = 0; u < 2; u++) {
            if constexpr (DBG & 64) asm volatile("" ::"v"(w2r[ks % R2][u]), "v"(bq[ct]));
            else ac[u][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2r[ks % R2][u], bq[ct], ac[u][ct], 0, 0, 0);
            if (u == 1 && ks + 1 < NKS) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * (GS_ * 2) + (ks + 1) * 32);
            if (ct == 3) {
              if (ks + R2 < NKS) w2r[ks % R2][u] = ld_w2(pass, ks + R2, u);
              else if (pass == 0) w2r[ks % R2][u] = ld_w2(1, ks + R2 - NKS, u);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    // epilogue of a pass: MFMA layout (4 rows x 1 column per lane) -> wave-private LDS patch -> 1 row x 4 columns per lane
    auto epi_piece = [&](const f32x16(&ac)[2][4], const __amdgpu_buffer_rsrc_t &dst, float scale, auto u_tag, auto ct_tag) {
      constexpr int u = decltype(u_tag)::value, ct = decltype(ct_tag)::value;
#pragma unroll
      for (int r = 0; r < 16; r++) patch[rowoff(r, hh) * PS_ + j] = ac[u][ct][r];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const float4 v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * p) * PS_ + 4 * (lane & 7));
        f32x4 o;
        o[0] = (pre[u][ct][4 * p + 0] + v.x) * scale;
        o[1] = (pre[u][ct][4 * p + 1] + v.y) * scale;
        o[2] = (pre[u][ct][4 * p + 2] + v.z) * scale;
        o[3] = (pre[u][ct][4 * p + 3] + v.w) * scale;
        if constexpr (DBG & 256) asm volatile("" ::"v"(o));
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dst,
                                                    evoff[ct] + (u ? ustep : 0u) + (unsigned)(8 * p * L * 4), 0, NT);
      }
    };

    // ---- next tile: parameters; its first X chunk is requested before this tile's stores
    int b_nxt = b_cur, t0_nxt = t0_cur;
    if (ntile < t_end) tile_bt(ntile, b_nxt, t0_nxt);
    {
      f32x16 ac[2][4];
      gemm2_loop(ac, I0{});
      mark(20);
      __builtin_amdgcn_sched_barrier(0);
      // pass 0's epilogue; every piece frees sixteen operand registers, which take the running skip rows of the same piece
#define AP_W_EPI0(U, CT)                                  \
      epi_piece(ac, ors, RS, U{}, CT{});                  \
      __builtin_amdgcn_sched_barrier(0);                  \
      load_pre(srs, U{}, CT{}, noacc);                    \
      __builtin_amdgcn_sched_barrier(0);
      AP_W_EPI0(I0, I0) AP_W_EPI0(I0, I1) AP_W_EPI0(I0, I2) AP_W_EPI0(I0, I3)
      AP_W_EPI0(I1, I0) AP_W_EPI0(I1, I1) AP_W_EPI0(I1, I2) AP_W_EPI0(I1, I3)
#undef AP_W_EPI0
      mark(21);
    }
    __builtin_amdgcn_sched_barrier(0);
    hrs = clip_rsrc(hin, b_nxt);
    x_geom(t0_nxt);
    issue_x(hrs, 0, I0{});                                       // (unconditional: after its last tile a workgroup re-requests and drops)
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x16 ac[2][4];
      gemm2_loop(ac, I1{});
      mark(22);
      // no barrier here: the X buffers have been free since the gate's barrier, the g image is next written by the next
      // tile's gate, and the epilogue only touches this wave's own patch
      epi_piece(ac, srs, 1.0f, I0{}, I0{}); epi_piece(ac, srs, 1.0f, I0{}, I1{});
      epi_piece(ac, srs, 1.0f, I0{}, I2{}); epi_piece(ac, srs, 1.0f, I0{}, I3{});
      mark(23);
      epi_piece(ac, srs, 1.0f, I1{}, I0{}); epi_piece(ac, srs, 1.0f, I1{}, I1{});
      epi_piece(ac, srs, 1.0f, I1{}, I2{}); epi_piece(ac, srs, 1.0f, I1{}, I3{});
      mark(24);
      __builtin_amdgcn_sched_barrier(0);
      tile_head();
      mark(30);
    }
    b_cur = b_nxt;
    t0_cur = t0_nxt;
  }
}

#ifdef AP_TOOLS
extern int g_dbg_bf16;
}  // namespace ap
extern "C" int ap_debug_wtrace(void *buf) {                       // device buffer of grid x 4 x 64 u64 (tools/trace_resblock_bf16w.py)
  unsigned long long *p = (unsigned long long *)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ap::g_wtrace), &p, sizeof(p));
}
namespace ap {
#endif

// -> 0 launched, 1 shape not served by this kernel
int launch_resblock_bf16w(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip, int accumulate,
                          int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  if (C != 256 || S != 256 || (L % 4) != 0 || L < 4 || (d % 4) != 0) return 1;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
      n = 256;
    n_cu = n;
  }
  const int ntiles = (L + PT_ - 1) / PT_;
  const int nblk = B * ntiles;
  int grid = nblk < n_cu ? nblk : n_cu;
  if (grid >= 8) grid &= ~7;
  const size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
  const char *wlo = (const char *)ctx->w1p_bf < (const char *)ctx->w2p_bf ? (const char *)ctx->w1p_bf : (const char *)ctx->w2p_bf;
  const unsigned w1_off = (unsigned)((const char *)ctx->w1p_bf - wlo + layer * n1 * 2);
  const unsigned w2_off = (unsigned)((const char *)ctx->w2p_bf - wlo + layer * n2 * 2);
  const unsigned wbytes = (unsigned)((size_t)ctx->NL * (n1 + n2) * 2);
  const float *blo = ctx->b1 < ctx->b2 ? ctx->b1 : ctx->b2;
  const unsigned b1_off = (unsigned)((ctx->b1 - blo + (size_t)layer * 2 * C) * 4);
  const unsigned b2_off = (unsigned)((ctx->b2 - blo + (size_t)layer * (C + S)) * 4);
#define AP_W_LAUNCH(D)                                                                                                         \
  resblock_bf16w_kernel<D><<<(unsigned)grid, 256, 0, st>>>(hin, pt, hout, skip, wlo, wbytes, w1_off, w2_off, blo, b1_off, b2_off, L, d, \
                                                           accumulate, ntiles, nblk)
#ifdef AP_TOOLS
  switch (g_dbg_bf16 & 0xfff) {
    case 0: AP_W_LAUNCH(0); break;
    case 1: AP_W_LAUNCH(1); break;
    case 2: AP_W_LAUNCH(2); break;
    case 3: AP_W_LAUNCH(3); break;
    case 4: AP_W_LAUNCH(4); break;
    case 6: AP_W_LAUNCH(6); break;
    case 7: AP_W_LAUNCH(7); break;
    case 8: AP_W_LAUNCH(8); break;
    case 15: AP_W_LAUNCH(15); break;
    case 32: AP_W_LAUNCH(32); break;
    case 64: AP_W_LAUNCH(64); break;
    case 128: AP_W_LAUNCH(128); break;
    case 256: AP_W_LAUNCH(256); break;
    case 384: AP_W_LAUNCH(384); break;
    case 384 + 15: AP_W_LAUNCH(384 + 15); break;
    case 384 + 96: AP_W_LAUNCH(384 + 96); break;
    case 2048: AP_W_LAUNCH(2048); break;
    default: set_error("bf16w: no such DBG instantiation"); return -22;
  }
#else
  AP_W_LAUNCH(0);
#endif
#undef AP_W_LAUNCH
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
