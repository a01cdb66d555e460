// AP_PREC_BF16_STORE: fused Residual_block.forward (WaveNet.py:75-97) with bf16 MFMA operands, fp32 accumulate and the residual
// stream stored in HBM as bf16 -- SURVEY.md 8(d)'s third precision row ("bf16 MFMA, bf16 storage").
//
// What is stored is not h but u = h + part_t of the layer that READS it, rounded once to bf16 (RNE): the reference adds part_t in
// place on the block input (WaveNet.py:77,84: `h = x` aliases), so u is both the dilated conv's operand (WaveNet.py:87) and the
// value the residual carries (WaveNet.py:97: (x + res) sqrt(.5) with x already u).  One image per layer serves both:
//
//     U[clip][C / 32][L][32] bf16      a sample's 32 channels of one chunk = one 64-byte row
//
//   * GEMM1 staging is pure data movement for every dilation and clip length: a chunk's (column, tap) operand is one row, moved with
//     one 16-byte load and one ds_write_b128 per lane (three per thread and chunk); taps outside the clip read through an
//     out-of-range offset (zeros: WaveNet.py:26-27).  No FiLM add, no convert, no mask, no alignment cases in the loop.
//   * Inside a row the 32 channels sit in MFMA-accumulator order: position p holds channel (p with bits 2 and 3 swapped) of the
//     chunk.  A lane of a 32 x 32 accumulator tile holds, for its column, rows (r & 3) + 8 (r >> 2) + 4 hh (r = 0..15): with this
//     order they are bytes [32 s + 16 hh, + 16) (s = r >> 3) of the row -- the epilogue packs its 16 values and stores them with two
//     16-byte stores straight from the accumulators (no LDS patch, no transpose), and the residual's u comes back the same way (two
//     16-byte loads per column tile).  GEMM1 does not care: its weight image is packed with the same K order (pack_w1_bf16_kernel, perm).
//   * The epilogue: h' = (u + (W_res g + b_res)) sqrt(.5) in fp32, then u' = bf16(h' + part_t of the NEXT layer) -- one rounding per
//     layer.  The net's last layer writes no image (WaveNet.py:131-135 returns the skip sum only).
//   * skip stays fp32: the block writes its gate output as the bf16 image [clip][L][256] GEMM2 consumes anyway, and
//     skipgemm_bf16_kernel adds a group of layers' skip_conv outputs in one K-concatenated GEMM (the deferred-skip form of
//     AP_PREC_BF16, unchanged).
// HBM bytes per 128-sample tile and layer: 64 KB in (u), 64 KB out (u'), 64 KB out (g) against 131 + 131 + 64 KB in AP_PREC_BF16.
// The machine shape is ap_resblock_bf16p.hip's: one persistent 8-wave workgroup per CU walking 128-sample tiles in XCD-local
// order, wave w = gate channels [32 w, 32 w + 32) in GEMM1 and res rows [32 w, 32 w + 32) in GEMM2, a ring of three k-steps of
// weight fragments L2 -> registers, the next chunk's staging inside the MFMA gaps of a chunk's fourth k-step, the next tile's first
// chunk requested ahead of this tile's stores.
// Oracle: oracle/diffwave_oracle.py eps_net(bf16_store=True).  Tests: tests/test_gpu_bf16_store.py.
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PT_ = 128;                 // time tile
constexpr int KC_ = 32;                  // channels per chunk -> 96 K rows = 6 k-steps of 16
constexpr int XS_ = 3 * KC_ + 8;         // bf16 per column row of the X image (208 B: conflict-free ds_read_b128 B fragments)
constexpr int GS_ = 256 + 8;             // bf16 per column row of the g image (528 B)

// tanh(a) sigmoid(b) = (1 - E) / ((1 + E)(1 + F)), E = e^(-2a), F = e^(-b): the arithmetic of ap_resblock_bf16p.hip's gate_fast2,
// operation for operation.
__device__ __forceinline__ f32x2 gate_pair_u(f32x2 a, f32x2 b) {
  const f32x2 ac = {__builtin_amdgcn_fmed3f(a[0], -16.0f, 16.0f), __builtin_amdgcn_fmed3f(a[1], -16.0f, 16.0f)};
  const f32x2 ea = ac * -2.885390081777926815f;
  const f32x2 eb = b * -1.442695040888963407f;
  const f32x2 E = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
  const f32x2 F = {__builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
  const f32x2 den = (E + 1.0f) * (F + 1.0f);
  const f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  return (1.0f - E) * r;
}

// the same gate, also handing out its two derivative factors sg (1 - th^2), th sg (1 - sg) (ap_resblock_bf16p.hip: gate_fast2_save; the
// returned gate is gate_pair_u's, operation for operation)
__device__ __forceinline__ f32x2 gate_pair_u_save(f32x2 a, f32x2 b, f32x2 &f1, f32x2 &f2) {
#pragma clang fp contract(off)
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

__host__ __device__ __forceinline__ int swap23(int p) { return (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1); }

}  // namespace

// u0 = bf16(ReLU(w0 x + b0) + part_t of layer 0) as the image the first block reads (WaveNet.py:147,168 then :82-84).
// Thread = (clip, chunk, sample, channel octet): eight channels of one sample -> one 16-byte store; a wave writes 1 KB contiguous.
__global__ __launch_bounds__(256) void init_conv_u_kernel(const float *__restrict__ x, const float *__restrict__ w0, const float *__restrict__ b0,
                                                          const float *__restrict__ pt, __bf16 *__restrict__ u, int C, int L, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // ((clip * C/32 + chunk) * L + t) * 4 + octet
  if (idx >= total) return;
  const int oct = (int)(idx & 3);
  const size_t row = idx >> 2;
  const int t = (int)(row % (size_t)L);
  const size_t bc = row / (size_t)L;
  const int chunk = (int)(bc % (size_t)(C / 32));
  const float xv = x[(bc / (size_t)(C / 32)) * (size_t)L + t];
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) {
#pragma clang fp contract(off)                                  // relu(fma(w0, x, b0)) as init_conv_kernel forms it, then a separate add
    const int c0 = chunk * 32 + swap23(oct * 8 + 2 * e), c1 = chunk * 32 + swap23(oct * 8 + 2 * e + 1);
    const float h0 = fmaxf(__builtin_fmaf(w0[c0], xv, b0[c0]), 0.f), h1 = fmaxf(__builtin_fmaf(w0[c1], xv, b0[c1]), 0.f);
    const f32x2 v2 = {h0 + pt[c0], h1 + pt[c1]};
    o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
  }
  reinterpret_cast<u32x4 *>(u)[idx] = o;
}

int launch_init_conv_u(ap_ctx *ctx, const float *x, const float *pt0, void *u, int B, int L, hipStream_t st) {
  const size_t total = (size_t)B * (ctx->C / 32) * L * 4;
  init_conv_u_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, ctx->w0, ctx->b0, pt0, (__bf16 *)u, ctx->C, L, total);
  AP_HIP(hipGetLastError());
  return 0;
}

// NOH: the net's last layer -- GEMM1, the gate and the g image only.
// SAVEF (the differentiable purifier's forward pass, ap_resblock_fwd_u_save): + the gate's derivative factors, an fp16 pair per (channel,
// sample) in the accumulators' order [clip][tile][wave][column tile][q][lane] x 16 bytes -- the image ap_resblock_bwd_bf16_saved reads
// (ap_resblock_bf16p.hip, SAVEF: same geometry); u' and the g image are bit-identical to the launch without it.
template <bool NOH, bool SAVEF = false>
__global__ __launch_bounds__(512, 2) void resblock_bf16u_kernel(
    const void *__restrict__ uin, void *__restrict__ uout, const float *__restrict__ ptn,       // images in / out, the NEXT layer's part_t
    const void *__restrict__ wbase, unsigned wbytes, unsigned w1_off, unsigned w2_off,        // bf16 weight images (one slab)
    const void *__restrict__ bbase, unsigned bbytes, unsigned b1_off, unsigned b2_off,        // fp32 bias vectors (one slab)
    int L, int d, int ntiles, int nblk, void *__restrict__ gout,                              // this layer's g image [clip][L][256] bf16
    void *__restrict__ fout = nullptr) {                                                      // SAVEF: the gate's derivative factors
  constexpr int C = 256, NW = 8, NCH = C / KC_, NKS = C / 16;
  constexpr int XS = XS_;
  constexpr int XBYTES = PT_ * XS * 2;                         // 26,624 B per X buffer, two buffers
  constexpr int GOFF = 2 * XBYTES;
  constexpr int BOFF = GOFF + PT_ * GS_ * 2;                   // b1 (2C floats: filter | gate rows), b2 res rows (C), next part_t (C)
  constexpr int LDS_BYTES = BOFF + 4 * C * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  (void)NW; (void)bbytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;

  // ---- tile walk (placement only): workgroups g, g+8, ... share an XCD; each XCD takes a contiguous run of (clip, tile) work
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

  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 2u;
  auto img_rsrc = [&](const void *base, int b) {               // a clip's image: [C / 32][L][32] bf16
    const uint64_t hb = (uint64_t)base + (uint64_t)b * (uint64_t)clip_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)clip_bytes, 0x00020000);
  };

  {
    const unsigned char *bb = static_cast<const unsigned char *>(bbase);
    float *lb = reinterpret_cast<float *>(lds + BOFF);
    lb[tid] = reinterpret_cast<const float *>(bb + b1_off)[tid];
    if (tid < C) {
      lb[2 * C + tid] = reinterpret_cast<const float *>(bb + b2_off)[tid];
      lb[3 * C + tid] = (!NOH && ptn) ? ptn[tid] : 0.f;
    }
  }

  // walk order inside a clip: position p -> tile r + k s (s = d / tile width capped at 16), so that the tiles an XCD holds at a
  // time include the ones whose centre columns are this tile's +-d taps (ap_resblock_bf16p.hip)
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

  // ---- staging unit: thread = (column 16 wave + cw, channel octet ln & 3) for the three taps; cw pairs columns four apart inside
  // an 8-lane ds_write_b128 group (832 B apart = 16 banks: conflict-free), a wave's load is 16 rows x 64 B = 1 KB contiguous
  unsigned xv[3];                                              // byte offsets of the three taps' rows inside the clip's image
  u32x4 xq[3];
  auto st_col = [&](int ln) { const int q = ln >> 2; return 16 * wave + (((q & 1) << 2) | ((q >> 1) & 3) | (q & 8)); };
  auto x_geom = [&](int t0) {
    // (lane id read here, not kept: a value kept from the prologue is spilled, and a scratch reload waits with vmcnt(0))
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int col = st_col(ln);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int t = t0 + col + (i - 1) * d;
      xv[i] = (t >= 0 && t < L) ? (unsigned)(t * 64 + (ln & 3) * 16) : 0x80000000u;     // outside the clip: zeros (WaveNet.py:26-27)
    }
  };
  auto issue_x = [&](const __amdgpu_buffer_rsrc_t &rs, int ch) {  // chunk ch of the clip's image = L x 64 bytes
#pragma unroll
    for (int i = 0; i < 3; i++) xq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xv[i], ch * L * 64, 0));
  };
  unsigned xwa = 0;                                             // this thread's place in an X buffer (tap 0), re-derived per chunk
  auto pack_geom = [&]() {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    xwa = (unsigned)((st_col(ln) * XS + (ln & 3) * 8) * 2);
  };
  auto pack_piece = [&](unsigned char *dst, auto i_tag) {       // tap i's 16 bytes straight into the image
    constexpr int i = decltype(i_tag)::value;
    if constexpr (i < 3) *reinterpret_cast<u32x4 *>(dst + xwa + i * (KC_ * 2)) = xq[i];
  };

  // ---- weight fragment streams (this wave's 64 GEMM1 rows / 32 GEMM2 rows), L2 -> registers: buffer loads with the fragment
  // index in the scalar offset, one VGPR (lane * 16) addresses every fragment
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc(wbase, wbytes);
  const unsigned lane16 = (unsigned)lane * 16u;
  // GEMM1 image [wave][chunk][kstep 6][rowtile 2][lane][8 bf16]: fragment f of this wave = f KB from the wave's base
  auto ld_w1 = [&](int frag) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, w1_off + (wave * NCH * 12 + frag) * 1024, 0));
  };
  // GEMM2 image [wave][rowtile 2][kstep 16][lane][8 bf16]; row tile 0 = res_conv rows
  auto ld_w2 = [&](int ks) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, w2_off + (wave * 2 * NKS + ks) * 1024, 0));
  };
  const int rdoff = (j * XS + 8 * hh) * 2;                      // this lane's B-fragment byte offset inside an X buffer
  const unsigned char *gb = lds + GOFF + (j * GS_ + 8 * hh) * 2;
  const float RS = 0.707106781186547524f;

  // ---- first tile: parameters and chunk-0 request
  int b_cur, t0_cur;
  tile_bt(t_first, b_cur, t0_cur);
  __amdgpu_buffer_rsrc_t urs = img_rsrc(uin, b_cur);
  x_geom(t0_cur);
  constexpr int RING = 3, PK = 3;                                // fragment ring depth (k-steps), k-step that carries the pack
  bf16x8 w[RING][2];                                             // GEMM1 fragment ring: k-step ks of a chunk uses w[ks % RING][row tile]
  auto tile_head = [&]() {
#pragma unroll
    for (int ks = 0; ks < RING; ks++)
#pragma unroll
      for (int rt = 0; rt < 2; rt++) w[ks][rt] = ld_w1(ks * 2 + rt);
    pack_geom();
    pack_piece(lds, I0{}); pack_piece(lds, I1{}); pack_piece(lds, I2{});
    issue_x(urs, 1);                                             // chunk 1: packed in chunk 0's fourth k-step
  };
  issue_x(urs, 0);
  __syncthreads();                                               // biases, next part_t visible
  tile_head();

#pragma unroll 1
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int t0 = t0_cur;
    const int ntile = tile + t_step;
    // ================================================ GEMM1 =========================================================
    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + (rt * C + 32 * wave + 8 * q + 4 * hh) * 4);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          acc[rt][ct][4 * q + 0] = bv4[0];
          acc[rt][ct][4 * q + 1] = bv4[1];
          acc[rt][ct][4 * q + 2] = bv4[2];
          acc[rt][ct][4 * q + 3] = bv4[3];
        }
      }
    __syncthreads();

    auto mf = [&](const bf16x8 &a, const bf16x8 &bq, int rt, int ct) {
      acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[rt][ct], 0, 0, 0);
    };
    auto rdb = [&](bf16x8 &dst, const unsigned char *xb, int ct, int ks) {   // ks: k-step 0..5 of the chunk
      dst = *reinterpret_cast<const bf16x8 *>(xb + (32 * ct) * (XS * 2) + ks * 32);
    };
    // One chunk = six k-steps of eight MFMAs in explicit order (pinned with sched_barrier): column-tile-major pairs share one B
    // fragment, re-read for the next k-step right after its pair; a k-step's two weight fragments are replaced right behind their
    // last MFMA by those of the k-step three on; k-step 3 carries the next chunk's staging; the X rows of the chunk after next are
    // requested behind the last MFMA (every weight request issued before the next pack has to be OLDER than the X request: vmcnt
    // retires in order).  The younger wave of a SIMD (waves 4-7) gets the priority in k-steps 0-2 (ap_resblock_bf16p.hip).
    auto chunk = [&](const unsigned char *xb, int ch, unsigned char *pdst, auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;            // 0: chunks 0..NCH-3, 1: NCH-2 (no X request), 2: NCH-1
      constexpr bool LAST = KIND == 2, WITH_X = KIND == 0;
      bf16x8 bv[4];
#pragma unroll
      for (int ct = 0; ct < 4; ct++) rdb(bv[ct], xb, ct, 0);
      if (wave >= 4) __builtin_amdgcn_s_setprio(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 6; ks++) {
        const bool reload = ks + RING < 6 || !LAST;              // k-steps 3-5 fetch the next chunk's first three
        const int nfrag = ks + RING < 6 ? ch * 12 + (ks + RING) * 2 : (ch + 1) * 12 + (ks + RING - 6) * 2;
        if (ks == 3) __builtin_amdgcn_s_setprio(0);
        if (ks == PK) {
          if constexpr (!LAST) pack_geom();
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          mf(w[ks % RING][0], bv[ct], 0, ct);
          if constexpr (!LAST) {
            if (ks == PK && ct == 0) pack_piece(pdst, I0{});
            if (ks == PK && ct == 1) pack_piece(pdst, I1{});
            if (ks == PK && ct == 2) pack_piece(pdst, I2{});
          }
          if (ct == 3 && reload) w[ks % RING][0] = ld_w1(nfrag);
          __builtin_amdgcn_sched_barrier(0);
          mf(w[ks % RING][1], bv[ct], 1, ct);
          if (ks < 5) rdb(bv[ct], xb, ct, ks + 1);
          if (ct == 3 && reload) w[ks % RING][1] = ld_w1(nfrag + 1);
          if constexpr (WITH_X) {
            if (ks == 5 && ct == 3) issue_x(urs, ch + 2);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
#pragma unroll 1
    for (int ch = 0; ch < NCH - 2; ch++) {
      chunk(lds + (ch & 1) * XBYTES + rdoff, ch, lds + ((ch + 1) & 1) * XBYTES, I0{});
      __syncthreads();
    }
    chunk(lds + ((NCH - 2) & 1) * XBYTES + rdoff, NCH - 2, lds + ((NCH - 1) & 1) * XBYTES, I1{});
    __syncthreads();
    chunk(lds + ((NCH - 1) & 1) * XBYTES + rdoff, NCH - 1, nullptr, I2{});

    // ================================================ gate ==========================================================
    // GEMM2's first requests and the residual's rows of u go out ahead of the gate.  The residual: this wave's res rows are chunk
    // `wave` of the image; lane (j, hh) of column tile ct takes bytes [32 s + 16 hh, + 16) of row t0 + 32 ct + j -- the values its
    // accumulator registers 8 s .. 8 s + 7 belong to.
    bf16x8 p0[4], p1[4];
    auto load_a4 = [&](bf16x8(&a)[4], int ks) {
      const int g0 = ks <= NKS - 4 ? ks : NKS - 4;
#pragma unroll
      for (int s = 0; s < 4; s++) a[s] = ld_w2(g0 + s);
    };
    u32x4 pre[4][2];
    unsigned ro[4];
    auto calc_ro = [&]() {                                       // (t0 made opaque: computed before the chunk loop the offsets sat in registers through GEMM1)
      int t0o = t0;
      asm volatile("" : "+s"(t0o));
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        const int t = t0o + 32 * ct + j;
        ro[ct] = t < L ? (unsigned)((wave * L + t) * 64 + hh * 16) : 0x80000000u;   // outside the clip: loads 0, store dropped
      }
    };
    if constexpr (!NOH) {
      calc_ro();
      load_a4(p0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        pre[ct][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ro[ct], 0, 2));          // (nt: hits what the staging left in L2, allocates nothing)
        pre[ct][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ro[ct] + 32u, 0, 2));
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    auto gate_ct = [&](auto ct_tag) {
      constexpr int ct = decltype(ct_tag)::value;
      int ln;                                                    // (lane id read here, not kept)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
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
            g2 = gate_pair_u_save(a2, b2, f1, f2);
            fq[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{f1[0], f2[0]}, f16x2));       // (tanh factor, sigmoid factor) of element e
            fq[e + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{f1[1], f2[1]}, f16x2));
          } else {
            g2 = gate_pair_u(a2, b2);
          }
          pk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(g2, bf16x2));
        }
        *reinterpret_cast<uint2 *>(lds + GOFF + ((32 * ct + j) * GS_ + 32 * wave + 8 * qq + 4 * hh) * 2) = make_uint2(pk[0], pk[1]);
        if constexpr (SAVEF) {
          const uint64_t fb = (uint64_t)fout + (uint64_t)b_cur * ((uint64_t)ntiles * 131072u);
          const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)fb), fhi = __builtin_amdgcn_readfirstlane((uint32_t)(fb >> 32));
          const __amdgpu_buffer_rsrc_t frs =
              __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)fhi << 32) | flo), 0, (int)((unsigned)ntiles * 131072u), 0x00020000);
          // (the whole offset in the VGPR, soffset = 0: see the note on 16-byte buffer stores in ap_resblock_bf16p.hip)
          __builtin_amdgcn_raw_buffer_store_b128(fq, frs, (unsigned)ln * 16u + (unsigned)((((t0 / PT_) * 8 + wave) * 16 + ct * 4 + qq) * 1024), 0, 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    gate_ct(I0{});
    gate_ct(I1{});
    gate_ct(I2{});
    gate_ct(I3{});
    float4 bias[4];
    if constexpr (!NOH) {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + (2 * C + 32 * wave + 8 * q + 4 * (ln >> 5)) * 4);
        bias[q] = make_float4(v4[0], v4[1], v4[2], v4[3]);
      }
      load_a4(p1, 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    // ================================================ GEMM2 (res_conv rows: WaveNet.py:93) ===========================
    auto gemm2_loop = [&](f32x16(&ac)[4]) {
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
      // B fragments (g image) are read one k-step ahead into the other of two register sets
      bf16x8 ba[4], bb[4];
      auto rdg = [&](bf16x8(&bq)[4], int ks) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * (GS_ * 2) + (ks & (NKS - 1)) * 32);
      };
      auto step = [&](const bf16x8 &a, const bf16x8(&use)[4], bf16x8(&nxt)[4], int ks) {
        rdg(nxt, ks + 1);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) ac[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, use[ct], ac[ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      rdg(ba, 0);
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += 8) {
        step(p0[0], ba, bb, ks + 0); step(p0[1], bb, ba, ks + 1); step(p0[2], ba, bb, ks + 2); step(p0[3], bb, ba, ks + 3);
        if (ks + 8 < NKS) load_a4(p0, ks + 8);
        __builtin_amdgcn_sched_barrier(0);
        step(p1[0], ba, bb, ks + 4); step(p1[1], bb, ba, ks + 5); step(p1[2], ba, bb, ks + 6); step(p1[3], bb, ba, ks + 7);
        if (ks + 12 < NKS) load_a4(p1, ks + 12);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // ---- next tile: parameters, and its first X chunk requested BEFORE the last stores of this tile (unconditional: after its
    // last tile a workgroup re-requests that tile's first chunk and drops it)
    const __amdgpu_buffer_rsrc_t uors = img_rsrc(uout, b_cur);
    const int b_this = b_cur;
    int b_nxt = b_cur, t0_nxt = t0_cur;
    if (ntile < t_end) tile_bt(ntile, b_nxt, t0_nxt);
    urs = img_rsrc(uin, b_nxt);
    x_geom(t0_nxt);
    issue_x(urs, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NOH) {
      tile_head();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      f32x16 ac[4];
      gemm2_loop(ac);
      tile_head();
      __builtin_amdgcn_sched_barrier(0);
      // epilogue, straight from the accumulators: u' = bf16((u + acc) sqrt(.5) + part_t of the next layer)  (WaveNet.py:97, :84)
      float4 pn[4];
      {
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const f32x4 v4 = *reinterpret_cast<const f32x4 *>(lds + BOFF + (3 * C + 32 * wave + 8 * q + 4 * (ln >> 5)) * 4);
          pn[q] = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
      }
      calc_ro();
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
#pragma clang fp contract(off)                                  // the oracle's operation order: (u + acc) * rs, then + part_t, each rounded to fp32
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const unsigned uw = pre[ct][s][e];
            const float u0 = __builtin_bit_cast(float, uw << 16), u1 = __builtin_bit_cast(float, uw & 0xffff0000u);
            const int r = 8 * s + 2 * e;                         // accumulator registers r, r + 1 = positions 16 s + 8 hh + 2 e, + 1 of the row
            const float4 pq = pn[r >> 2];
            const float pa = (r & 2) ? pq.z : pq.x, pb = (r & 2) ? pq.w : pq.y;
            const f32x2 v2 = {(u0 + ac[ct][r]) * RS + pa, (u1 + ac[ct][r + 1]) * RS + pb};
            o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
          }
          // (offset step in the VGPR, soffset = 0: a >8-byte buffer store with an SGPR soffset reads its data late: ap_resblock_bf16p.hip)
          __builtin_amdgcn_raw_buffer_store_b128(o, uors, ro[ct] + (unsigned)(32 * s), 0, 2);   // nt: written once, read by the next launch
        }
      }
    }
    // g image -> HBM: 128 columns x 512 B, one 16-byte piece per lane and step (a wave moves two whole columns per step)
    {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const uint64_t gb_ = (uint64_t)gout + (uint64_t)b_this * ((uint64_t)L * 512u);
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
        __builtin_amdgcn_raw_buffer_store_b128(v, grs, off, 0, 2);
      }
    }
    b_cur = b_nxt;
    t0_cur = t0_nxt;
  }
}

bool resblock_bf16u_serves(const ap_ctx *ctx, int L) {
  return ctx->cfg.precision == AP_PREC_BF16_STORE && ctx->C == 256 && ctx->S == 256 && L >= 1 &&
         (size_t)L * 512 < ((size_t)1 << 31);
}

// uout null: the net's last layer (no res_conv, no image out)
int launch_resblock_bf16u(ap_ctx *ctx, int layer, const void *uin, const float *pt_next, void *uout, void *gout, int B, int L, hipStream_t st,
                          void *fout) {
  if (!resblock_bf16u_serves(ctx, L)) {
    set_error("AP_PREC_BF16_STORE: built for res = skip = 256 channels, clips shorter than 2^22 samples (got %d / %d, L = %d)", ctx->C, ctx->S, L);
    return -22;
  }
  // launches of at most one tile per CU (one- and two-clip calls): 64-sample tiles on twice as many workgroups, bit-identical results
  if (!fout && resblock_bf16us_serves(ctx, B, L)) return launch_resblock_bf16us(ctx, layer, uin, pt_next, uout, gout, B, L, st);
  const int C = ctx->C, S = ctx->S;
  const int d = 1 << (layer % ctx->cfg.dilation_cycle);
  const int n_cu = device_cu_count();
  const int ntiles = (L + PT_ - 1) / PT_;
  const int nblk = B * ntiles;
  int grid = nblk < n_cu ? nblk : n_cu;
  if (grid >= 8) grid &= ~7;
  const size_t n1 = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C;
  const char *wlo = (const char *)ctx->w1p_bf < (const char *)ctx->w2p_bf ? (const char *)ctx->w1p_bf : (const char *)ctx->w2p_bf;
  const unsigned w1_off = (unsigned)((const char *)ctx->w1p_bf - wlo + layer * n1 * 2);
  const unsigned w2_off = (unsigned)((const char *)ctx->w2p_bf - wlo + layer * n2 * 2);
  const unsigned wbytes = (unsigned)(((size_t)ctx->NL * (n1 + n2) + (size_t)S * S) * 2);
  const float *blo = ctx->b1 < ctx->b2 ? ctx->b1 : ctx->b2;
  const float *bhi = ctx->b1 < ctx->b2 ? ctx->b2 : ctx->b1;
  const unsigned b1_off = (unsigned)((ctx->b1 - blo + (size_t)layer * 2 * C) * 4);
  const unsigned b2_off = (unsigned)((ctx->b2 - blo + (size_t)layer * (C + S)) * 4);
  const unsigned bbytes = (unsigned)((bhi - blo + (size_t)ctx->NL * 2 * C) * 4);
  if (fout) {                                                    // the differentiable purifier's forward: + the gate's derivative factors
    if ((size_t)ntiles * 131072 >= ((size_t)1 << 31)) { set_error("AP_PREC_BF16_STORE: clip too long for the gate-factor image"); return -22; }
    if (uout)
      resblock_bf16u_kernel<false, true><<<(unsigned)grid, 512, 0, st>>>(uin, uout, pt_next, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off,
                                                                            L, d, ntiles, nblk, gout, fout);
    else
      resblock_bf16u_kernel<true, true><<<(unsigned)grid, 512, 0, st>>>(uin, nullptr, nullptr, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off,
                                                                           L, d, ntiles, nblk, gout, fout);
    AP_HIP(hipGetLastError());
    return 0;
  }
  if (uout)
    resblock_bf16u_kernel<false><<<(unsigned)grid, 512, 0, st>>>(uin, uout, pt_next, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off, L, d,
                                                                 ntiles, nblk, gout);
  else
    resblock_bf16u_kernel<true><<<(unsigned)grid, 512, 0, st>>>(uin, nullptr, nullptr, wlo, wbytes, w1_off, w2_off, blo, bbytes, b1_off, b2_off, L, d,
                                                                ntiles, nblk, gout);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
