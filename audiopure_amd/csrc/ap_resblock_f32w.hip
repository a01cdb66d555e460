// AP_PREC_F32, shipped shape (res = skip = 256 channels): fused Residual_block.forward (WaveNet.py:75-97) with the dilated k = 3
// conv (WaveNet.py:87) in F(2,3) minimal-filtering form over the dilation pair.
//
// Outputs t and t + d of the conv share the taps u[t-d], u[t], u[t+d], u[t+2d] (u = h + part_t, zero outside the clip), so the
// pair is four [2C x C] products instead of six:
//     m1 = W0 (u0 - u2)    m2 = (W0+W1+W2)/2 (u1 + u2)    m3 = (W0-W1+W2)/2 (u2 - u1)    m4 = W2 (u3 - u1)
//     y[t] = (m1 + m2) + m3 + b        y[t+d] = (m2 - m3) + m4 + b
// GEMM1 is 8.39 GFLOP per clip instead of 12.58, the block 12.58 instead of 16.78 -- on the exact-fp32 matrix instruction
// (v_mfma_f32_32x32x2_f32), whose rate bounds this kernel.  Transformed weights are computed in double and rounded once at load;
// the oracle restates the same operation order (oracle/diffwave_oracle.py::winograd_dilated_conv) and holds the reference's
// golden vectors at the fp32 tolerances (tests/test_oracle_golden.py).
//
// Pairing: sample t is a pair's first output when floor(t / d) is even, its second otherwise; pair p has first output
// tf(p) = ((p >> log2 d) << (log2 d + 1)) + (p & (d - 1)).  A tile is 32 consecutive pairs = 64 outputs: for d < 32 one 64-sample
// window, for d >= 32 two 32-sample runs d apart.
//
// Machine shape: one persistent workgroup per CU, FOUR waves -- one per SIMD, each with the SIMD's whole register file: wave w
// owns the tanh and sigmoid rows of gate channels [64w, 64w + 64) = 128 GEMM1 rows x 32 pair columns x 4 products = 256
// accumulator registers; every LDS B fragment (one ds_read_b128 = four k-steps) feeds 16 MFMAs, every 16-byte weight fragment
// four.  Per chunk of 32 channels: 16 raw 4-byte loads per thread (4 taps x 4 channels of one pair), FiLM add, zero padding,
// the four input differences, four ds_write_b128 into a [product][column][k] image (144-byte rows: conflict-free 16-byte reads
// and writes); weights stream L2 -> registers through a ring one k-group (64 MFMAs) deep, requested after the unit that frees
// their registers so that no wait on them includes the HBM-latency activation loads issued at the chunk's top.
// Then: output transform + gate in registers -> g image [column][channel] (1040-byte rows) -> GEMM2 [(C + S) x C].[C x 64]
// (res rows and skip rows of the wave's 64 channels) -> h' = (h + part_t + res) sqrt(1/2), skip += by memory-side float atomics
// (one adder per element per launch: deterministic).
#include "ap_common.h"

namespace ap {

namespace {

constexpr int WC_ = 256;                 // res = skip channels
constexpr int NP_ = 32;                  // pairs per tile
constexpr int KCW_ = 32;                 // channels per staged chunk
constexpr int NCH_ = WC_ / KCW_;         // 8 chunks
constexpr int XS_ = KCW_ + 4;            // floats per (product, column) row of the X image
constexpr int XCOMP_ = NP_ * XS_;        // one product's image
constexpr int XBUF_ = 4 * XCOMP_;        // one chunk's image (18 KB)
constexpr int GS_ = WC_ + 4;             // floats per column row of the g image
constexpr int GOFF_ = 2 * XBUF_;
constexpr int POFF_ = GOFF_ + 64 * GS_;         // Q16: output patches, [wave 4][2][32 rows][36] floats (36.9 KB)
constexpr int PS_ = 36;
constexpr int LDS_FLOATS_ = GOFF_ + 64 * GS_;   // 103,424 B
constexpr int LDS_FLOATS_Q16_ = POFF_ + 4 * 2 * 32 * PS_;   // 140,288 B
constexpr unsigned UNIT_BYTES_ = 4 * 64 * 16;   // one (k-group, product) unit of a wave's GEMM1 image: 4 row tiles x 64 lanes x 16 B
constexpr unsigned W1W_WAVE_BYTES_ = NCH_ * 4 * 4 * UNIT_BYTES_;   // 512 KB per wave and layer
constexpr unsigned W2W_WAVE_BYTES_ = (WC_ / 8) * UNIT_BYTES_;      // 128 KB per wave and layer

}  // namespace

// GEMM1 image: [wave 4][chunk 8][k-group 4][product 4][row tile 4][lane 64][4]; lane (i, hh), element e = k-step e of the group:
// channel 32 chunk + 8 kg + 4 hh + e (the k a lane's ds_read_b128 B fragment holds for that step); row tiles interleave the tanh
// (even) and sigmoid (odd) halves as in pack_w1_kernel.
__global__ void pack_w1w_kernel(const float *__restrict__ w1f, float *__restrict__ out) {
  constexpr int C = WC_;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4u * NCH_ * 4 * 4 * 4 * 64 * 4) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, rt = (idx >> 8) & 3, comp = (idx >> 10) & 3, kg = (idx >> 12) & 3,
            ch = (idx >> 14) & 7, w = idx >> 17;
  const int i = lane & 31, hh = lane >> 5;
  const int c = 32 * ch + 8 * kg + 4 * hh + e;
  const int o = (rt & 1) * C + 64 * w + 32 * (rt >> 1) + i;
  const float *p = w1f + ((size_t)o * C + c) * 3;
  const double w0 = p[0], w1 = p[1], w2 = p[2];
  const double v = comp == 0 ? w0 : comp == 1 ? (w0 + w1 + w2) * 0.5 : comp == 2 ? (w0 - w1 + w2) * 0.5 : w2;
  out[idx] = (float)v;
}

// GEMM2 image: [wave 4][k-group 32][row tile 4][lane 64][4]; channel 8 kg + 4 hh + e; row tiles 0,1 = res_conv rows of the wave's
// 64 channels, 2,3 = skip_conv rows of the same channels (w2f = [res rows; skip rows]).
__global__ void pack_w2w_kernel(const float *__restrict__ w2f, float *__restrict__ out) {
  constexpr int C = WC_;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4u * (C / 8) * 4 * 64 * 4) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, rt = (idx >> 8) & 3, kg = (idx >> 10) & 31, w = idx >> 15;
  const int i = lane & 31, hh = lane >> 5;
  const int c = 8 * kg + 4 * hh + e;
  const int o = (rt >> 1) * C + 64 * w + 32 * (rt & 1) + i;
  out[idx] = w2f[(size_t)o * C + c];
}

int launch_pack_f32w(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  const size_t n1f = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C, n1w = (size_t)4 * 2 * C * C;
  for (int n = 0; n < ctx->NL; n++) {
    pack_w1w_kernel<<<(unsigned)((n1w + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1f, ctx->w1w + n * n1w);
    pack_w2w_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, ctx->w2w + n * n2);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

// SAVE (the differentiable path's forward pass, ap_resblock_fwd_save): the pre-gate activations y = DilConv(u) + b are also written
// to aout [B][2C][L] (rows 0..C-1 the tanh half, C..2C-1 the sigmoid half: what ap_resblock_bwd reads).
// Q16 (clip lengths that are multiples of four -- every shipped shape): the epilogue goes through wave-private LDS patches so that
// each 32 x 32 accumulator tile leaves as 1 row x 4 samples per lane -- 16-byte stores, the residual's h patch and the running skip
// rows as 16-byte loads (skip += as a read-modify-write: one workgroup owns a tile within a launch, launches are stream-ordered,
// so the sum is as deterministic as the float atomics of the 4-byte form).  A CU's store path moves a 4-byte-per-lane store
// stream at a fraction of its 16-byte rate: the 4-byte epilogue cost 6.5 % of the launch (tools/ab_f32w.py).
template <bool NOH, bool SAVE = false, bool Q16 = false>
__global__ __launch_bounds__(256, 1) void resblock_f32w_kernel(
    const float *__restrict__ hin, const float *__restrict__ pt, float *__restrict__ hout, float *__restrict__ skip,
    const float *__restrict__ w1w, const float *__restrict__ b1, const float *__restrict__ w2w,
    const float *__restrict__ b2, int L, int logd, int accumulate, int ntiles, int nblk, float *__restrict__ aout) {
  constexpr int C = WC_;
  __shared__ __attribute__((aligned(16))) float lds[Q16 ? LDS_FLOATS_Q16_ : LDS_FLOATS_];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int sq = tid >> 5;                                       // staging: channel quad of the chunk (0..7); j = the pair
  const int d = 1 << logd;

  // ---- tile walk: each XCD (workgroups g, g + 8, ... share one) takes a contiguous run of (clip, tile) work, its CUs walking
  // it side by side, so neighbouring tiles -- which share half their taps -- meet in one L2.  Placement only.
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

  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  const __amdgpu_buffer_rsrc_t w1rs = uni_rsrc(reinterpret_cast<const char *>(w1w) + (size_t)wave * W1W_WAVE_BYTES_, W1W_WAVE_BYTES_);
  const __amdgpu_buffer_rsrc_t w2rs = uni_rsrc(reinterpret_cast<const char *>(w2w) + (size_t)wave * W2W_WAVE_BYTES_, W2W_WAVE_BYTES_);
  const unsigned lane16 = (unsigned)lane * 16u;
  // biases and part_t through buffer loads as well: as plain pointer loads these loop-invariant per-lane values are hoisted out of the
  // persistent loop and spilled; laundered pointers lose their address space and become flat loads, which force vmcnt(0) waits
  const __amdgpu_buffer_rsrc_t b1rs = uni_rsrc(b1, 2u * C * 4u), b2rs = uni_rsrc(b2, 2u * C * 4u), ptrs = uni_rsrc(pt, C * 4u);
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t &rs, int idx) {    // four consecutive floats at element idx (16-byte aligned)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)idx * 4u, 0, 0));
  };

  auto load_a1 = [&](f32x4(&a)[4], unsigned unit) {             // one (k-group, product) unit of GEMM1 weights: 4 row tiles
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
      a[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w1rs, lane16 + rt * 1024u, unit * UNIT_BYTES_, 0));
  };
  auto load_a2 = [&](f32x4(&a)[4], unsigned kg) {
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
      a[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane16 + rt * 1024u, kg * UNIT_BYTES_, 0));
  };

  // g image column of a pair's first / second output: sample order inside the tile (d < 32: one 64-sample window; else two runs)
  const int col0 = d >= 32 ? j : (((j >> logd) << (logd + 1)) + (j & (d - 1)));
  const int col1 = d >= 32 ? 32 + j : col0 + d;

  // ---- per-tile state, set one tile ahead (the next tile's first loads are requested before this tile's epilogue)
  int b, p0;
  __amdgpu_buffer_rsrc_t hrs;
  unsigned voff[4];
  bool tok[4];
  auto set_tile = [&](int tile) {
    b = __builtin_amdgcn_readfirstlane(tile / ntiles);
    p0 = __builtin_amdgcn_readfirstlane((tile % ntiles) * NP_);
    hrs = uni_rsrc(hin + (size_t)b * C * L, clip_bytes);
    const int p = p0 + j;                                        // staging geometry of this thread's pair
    const int tf = ((p >> logd) << (logd + 1)) + (p & (d - 1));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int tp = tf + (k - 1) * d;
      tok[k] = (tp >= 0) && (tp < L);
      voff[k] = ((unsigned)min(max(tp, 0), L - 1) + (unsigned)(4 * sq) * (unsigned)L) * 4u;
    }
  };
    float xr[4][4];                                              // [channel of the quad][tap]
    f32x4 ptq;
    auto issue_x = [&](int ch) {
      ptq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ptrs, (unsigned)(16 * sq), ch * KCW_ * 4, 0));
#pragma unroll
      for (int cc = 0; cc < 4; cc++)
#pragma unroll
        for (int k = 0; k < 4; k++)
          xr[cc][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, voff[k], (ch * KCW_ + cc) * L * 4, 0));
    };
    auto store_x = [&](float *dst) {                             // FiLM add (WaveNet.py:84), zero padding (:26-27), input differences
      // (the loaded values pass through an empty asm: it pins this arithmetic -- and the wait for the loads -- HERE; instruction
      // selection would otherwise place it right behind the loads, half a chunk early)
      asm volatile("" : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[0][2]), "+v"(xr[0][3]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[1][2]),
                   "+v"(xr[1][3]), "+v"(xr[2][0]), "+v"(xr[2][1]), "+v"(xr[2][2]), "+v"(xr[2][3]), "+v"(xr[3][0]), "+v"(xr[3][1]),
                   "+v"(xr[3][2]), "+v"(xr[3][3]), "+v"(ptq));
      f32x4 c0, c1, c2, c3;
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {
        const float u0 = tok[0] ? xr[cc][0] + ptq[cc] : 0.f;
        const float u1 = tok[1] ? xr[cc][1] + ptq[cc] : 0.f;
        const float u2 = tok[2] ? xr[cc][2] + ptq[cc] : 0.f;
        const float u3 = tok[3] ? xr[cc][3] + ptq[cc] : 0.f;
        c0[cc] = u0 - u2;
        c1[cc] = u1 + u2;
        c2[cc] = u2 - u1;
        c3[cc] = u3 - u1;
      }
      float *q = dst + j * XS_ + 4 * sq;
      *reinterpret_cast<f32x4 *>(q) = c0;
      *reinterpret_cast<f32x4 *>(q + XCOMP_) = c1;
      *reinterpret_cast<f32x4 *>(q + 2 * XCOMP_) = c2;
      *reinterpret_cast<f32x4 *>(q + 3 * XCOMP_) = c3;
    };

    // ---- the first tile's first weights and first chunk of X (later tiles: requested at the end of the tile before)
    f32x4 a[4][4];                                               // [product][row tile]: a ring one k-group (four units, 64 MFMAs) deep
    set_tile(t_first);
#pragma unroll
    for (int u = 0; u < 4; u++) load_a1(a[u], (unsigned)u);
    issue_x(0);

#pragma unroll 1
  for (int tile = t_first; tile < t_end; tile += t_step) {
    f32x16 acc[4][4];                                            // [product][row tile]; the dilated conv's bias rides on m2 (in both outputs)
#pragma unroll
    for (int rt = 0; rt < 4; rt++) {
      const int obase = (rt & 1) * C + 64 * wave + 32 * (rt >> 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 bv = ld4(b1rs, obase + 8 * q + 4 * hh);
        acc[1][rt][4 * q + 0] = bv[0];
        acc[1][rt][4 * q + 1] = bv[1];
        acc[1][rt][4 * q + 2] = bv[2];
        acc[1][rt][4 * q + 3] = bv[3];
      }
#pragma unroll
      for (int r = 0; r < 16; r++) acc[0][rt][r] = acc[2][rt][r] = acc[3][rt][r] = 0.f;
    }
    store_x(lds);
    __syncthreads();

    // ---- GEMM1: 8 chunks x 4 k-groups x 4 products; a unit = 16 MFMAs on one B fragment
    const float *xfrag = lds + j * XS_ + 4 * hh;
#pragma unroll 1
    for (int ch = 0; ch < NCH_; ch++) {
      const float *xb = xfrag + (ch & 1) * XBUF_;
      issue_x(ch + 1 < NCH_ ? ch + 1 : ch);   // (no branches in this loop: the last chunk re-requests itself, unused)
      f32x4 bq[2];
      bq[0] = *reinterpret_cast<const f32x4 *>(xb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 4; kg++) {
#pragma unroll
        for (int comp = 0; comp < 4; comp++) {
          const int u = 4 * kg + comp;
          if (u + 1 < 16) bq[(u + 1) & 1] = *reinterpret_cast<const f32x4 *>(xb + ((u + 1) & 3) * XCOMP_ + ((u + 1) >> 2) * 8);
          // the next chunk's FiLM add / padding / differences / LDS writes ride in the MFMA gaps of unit 12 (the X loads were
          // requested twelve units = 12 k cycles ago)
          if (u == 12) store_x(lds + ((ch + 1) & 1) * XBUF_);
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
              acc[comp][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[comp][rt][e], bq[u & 1][e], acc[comp][rt], 0, 0, 0);
          if (u == 12) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          // the same product's unit of the next k-group takes over this unit's registers
          load_a1(a[comp], (unsigned)((16 * ch + u + 4) & (16 * NCH_ - 1)));   // (the last k-group wraps to the image's first units, unused)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }

    // sample of GEMM2 column (ct, j)
    const int tfirst = ((p0 >> logd) << (logd + 1)) + (p0 & (d - 1));
    int tcol[2];
    tcol[0] = tfirst + j;
    tcol[1] = tfirst + (d >= 32 ? d : 32) + j;
    // SAVE: this lane's pair's two samples as byte offsets into the clip's pre-gate rows (past the clip: dropped by the range check)
    __amdgpu_buffer_rsrc_t ars = hrs;
    unsigned sv0 = 0, sv1 = 0;
    if constexpr (SAVE) {
      ars = uni_rsrc(aout + (size_t)b * 2 * C * L, 2u * clip_bytes);
      const int pj = p0 + j;
      const int s0 = ((pj >> logd) << (logd + 1)) + (pj & (d - 1));
      sv0 = s0 < L ? (unsigned)s0 * 4u : 0x80000000u;
      sv1 = s0 + d < L ? (unsigned)(s0 + d) * 4u : 0x80000000u;
    }
    // ---- output transform, gate (WaveNet.py:90) -> g image [column][channel]
    {
      float *g0 = lds + GOFF_ + col0 * GS_ + 64 * wave + 4 * hh;
      float *g1 = lds + GOFF_ + col1 * GS_ + 64 * wave + 4 * hh;
#pragma unroll
      for (int p = 0; p < 2; p++) {
        // (pins the eight product tiles of this channel half in the accumulator registers up to here: the register allocator
        // would otherwise move all 256 accumulators to VGPRs at the loop's exit -- there is no room -- and spill them)
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) asm volatile("" : "+a"(acc[c4][2 * p]), "+a"(acc[c4][2 * p + 1]));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f32x4 v0, v1;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int r = 4 * q + e;
            const float ta = (acc[0][2 * p][r] + acc[1][2 * p][r]) + acc[2][2 * p][r];
            const float sa = (acc[0][2 * p + 1][r] + acc[1][2 * p + 1][r]) + acc[2][2 * p + 1][r];
            const float tb = (acc[1][2 * p][r] - acc[2][2 * p][r]) + acc[3][2 * p][r];
            const float sb = (acc[1][2 * p + 1][r] - acc[2][2 * p + 1][r]) + acc[3][2 * p + 1][r];
            v0[e] = gate(ta, sa);
            v1[e] = gate(tb, sb);
            if constexpr (SAVE) {                                // channel 64 wave + 32 p + rowoff(r, hh), samples of this lane's pair
              const unsigned so = ((unsigned)(64 * wave + 32 * p + 4 * hh) * (unsigned)L) * 4u;
              const int ro = ((r & 3) + 8 * (r >> 2)) * L * 4;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ta), ars, so + sv0, ro, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sa), ars, so + sv0 + (unsigned)C * (unsigned)L * 4u, ro, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tb), ars, so + sv1, ro, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sb), ars, so + sv1 + (unsigned)C * (unsigned)L * 4u, ro, 0);
            }
          }
          *reinterpret_cast<f32x4 *>(g0 + 32 * p + 8 * q) = v0;   // channels 64 wave + 32 p + 8 q + 4 hh + (0..3)
          *reinterpret_cast<f32x4 *>(g1 + 32 * p + 8 * q) = v1;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // ---- GEMM2: row tiles 0,1 = res_conv rows, 2,3 = skip_conv rows of this wave's 64 channels; 64 sample columns
    f32x16 acc2[4][2];
    {
#pragma unroll
      for (int rt = 0; rt < 4; rt++) {
        if (NOH && rt < 2) continue;
        const int cb = 64 * wave + 32 * (rt & 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int c = cb + 8 * q + 4 * hh;
          f32x4 bv = ld4(b2rs, (rt < 2 ? 0 : C) + c);
          if (rt < 2) bv += ld4(ptrs, c);                        // u = h + part_t re-enters the residual (WaveNet.py:84,97)
#pragma unroll
          for (int ct = 0; ct < 2; ct++) {
            acc2[rt][ct][4 * q + 0] = bv[0];
            acc2[rt][ct][4 * q + 1] = bv[1];
            acc2[rt][ct][4 * q + 2] = bv[2];
            acc2[rt][ct][4 * q + 3] = bv[3];
          }
        }
      }
    }
    f32x4 a2[2][4];                                              // ring of two k-groups (64 MFMAs)
#pragma unroll
    for (int k = 0; k < 2; k++) load_a2(a2[k], (unsigned)k);

    // The residual's h patch (this wave's 64 res rows x 64 columns) is requested now -- behind the first weight groups, so that
    // their waits do not include it -- and consumed after GEMM2: the epilogue never waits on memory.
    __builtin_amdgcn_sched_barrier(0);
    float hres[Q16 ? 1 : 2][Q16 ? 1 : 2][16];
    f32x4 hq[Q16 ? 2 : 1][Q16 ? 2 : 1][4];                      // Q16: [res row tile][column tile][row octet p]: row l / 8 + 8 p, samples 4 (l % 8) ..
    unsigned eo4[2] = {0x80000000u, 0x80000000u};
    if constexpr (Q16) {
      const int rowq = lane >> 3, cq = lane & 7;
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        const int t = tfirst + (ct ? (d >= 32 ? d : 32) : 0) + 4 * cq;
        eo4[ct] = t < L ? ((unsigned)(64 * wave + rowq) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      }
      if (!NOH) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int p = 0; p < 4; p++)
              hq[rt][ct][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, eo4[ct], (32 * rt + 8 * p) * L * 4, 2));
      }
    }
    if (!Q16 && !NOH) {
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const unsigned ev = ((unsigned)(64 * wave + 32 * rt + 4 * hh) * (unsigned)L + (unsigned)min(tcol[ct], L - 1)) * 4u;
#pragma unroll
          for (int r = 0; r < 16; r++)
            hres[rt][ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, ev, ((r & 3) + 8 * (r >> 2)) * L * 4, 2));
        }
    }
    __syncthreads();                                             // g image complete

    {
      const float *gfrag = lds + GOFF_ + j * GS_ + 4 * hh;
#pragma unroll 1
      for (int kg4 = 0; kg4 < C / 8; kg4 += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int kg = kg4 + k;
          f32x4 bq2[2];
          bq2[0] = *reinterpret_cast<const f32x4 *>(gfrag + 8 * kg);
          bq2[1] = *reinterpret_cast<const f32x4 *>(gfrag + 32 * GS_ + 8 * kg);
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int rt = NOH ? 2 : 0; rt < 4; rt++)
#pragma unroll
              for (int ct = 0; ct < 2; ct++)
                acc2[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[k & 1][rt][e], bq2[ct][e], acc2[rt][ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          load_a2(a2[k & 1], (unsigned)((kg + 2) & (C / 8 - 1)));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    const __amdgpu_buffer_rsrc_t ors = uni_rsrc((NOH ? skip : hout) + (size_t)b * C * L, clip_bytes);
    const __amdgpu_buffer_rsrc_t srs = uni_rsrc(skip + (size_t)b * C * L, clip_bytes);     // S == C
    if constexpr (Q16) {
      // ---- 16-byte epilogue (WaveNet.py:97, :133).  Order: the running skip rows are requested, the res tiles leave (h'), the NEXT
      // tile's first weights and X chunk are requested, the skip tiles leave -- so the next tile's first wait has only the 16 skip
      // stores behind its loads (vmcnt retires in order).
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const float RS = 0.707106781186547524f;   // float(math.sqrt(0.5))
      f32x4 sk4[2][2][4];
      if (accumulate) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int p = 0; p < 4; p++)
              sk4[rt][ct][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, eo4[ct], (32 * rt + 8 * p) * L * 4, 2));
      } else {
#pragma unroll
        for (int i = 0; i < 16; i++) (&sk4[0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float *patch0 = lds + POFF_ + wave * (2 * 32 * PS_);
      const int rowq = lane >> 3, cq = lane & 7;
      auto leave = [&](int k, const f32x16 &tile, auto &&out4) {  // accumulator tile -> patch k & 1 -> 4 x (row, 4 samples) per lane
        float *patch = patch0 + (k & 1) * (32 * PS_);
#pragma unroll
        for (int r = 0; r < 16; r++) patch[rowoff(r, hh) * PS_ + j] = tile[r];
#pragma unroll
        for (int p = 0; p < 4; p++) out4(p, *reinterpret_cast<const f32x4 *>(patch + (rowq + 8 * p) * PS_ + 4 * cq));
      };
      {
        if (!NOH) {
#pragma unroll
          for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
              leave(2 * rt + ct, acc2[rt][ct], [&](int p, f32x4 v) {
                const f32x4 o = (hq[rt][ct][p] + v) * RS;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ors, eo4[ct], (32 * rt + 8 * p) * L * 4, 2);
              });
        }
      }
      {
        set_tile(tile + t_step < t_end ? tile + t_step : tile);  // (the last tile re-requests itself, unused)
#pragma unroll
        for (int u = 0; u < 4; u++) load_a1(a[u], (unsigned)u);
        issue_x(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
            leave(2 * rt + ct, acc2[2 + rt][ct], [&](int p, f32x4 v) {
              const f32x4 o = sk4[rt][ct][p] + v;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), srs, eo4[ct], (32 * rt + 8 * p) * L * 4, 2);
            });
      }
      continue;
    }
    // ---- the NEXT tile's first weights and first chunk of X are requested here, ahead of this tile's stores: vmcnt retires in
    // order, so requested after them their wait would include the 64 float atomics (thousands of cycles each to retire with every
    // CU issuing them); requested before, it includes at most the plain h' stores (the counter holds 63: the wait for these loads
    // lets the youngest 63 operations -- the atomics -- stay outstanding)
    {
      set_tile(tile + t_step < t_end ? tile + t_step : tile);    // (the last tile re-requests itself, unused)
#pragma unroll
      for (int u = 0; u < 4; u++) load_a1(a[u], (unsigned)u);
      issue_x(0);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (WaveNet.py:97, :133): buffer stores / memory-side float atomics -- one VGPR offset per 32 x 32 tile, the row
    // stride in SGPR offsets, columns past the clip's end dropped by the range check (no address arithmetic, no branches)
    {
      typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
      const float RS = 0.707106781186547524f;   // float(math.sqrt(0.5))
#pragma unroll
      for (int rt = NOH ? 2 : 0; rt < 4; rt++) {
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const int t = tcol[ct];
          const unsigned eo = t < L ? ((unsigned)(64 * wave + 32 * (rt & 1) + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
          if (rt < 2) {
#pragma unroll
            for (int r = 0; r < 16; r++)
              // nt (also the skip store below): once-written streams must not displace the h rows in the XCD's L2, which
              // neighbouring tiles' taps and the residual read again
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (hres[rt & 1][ct][r] + acc2[rt][ct][r]) * RS), ors, eo,
                                                    ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
          } else if (accumulate) {
#pragma unroll
            for (int r = 0; r < 16; r++)
              (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc2[rt][ct][r], srs, (int)eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 0);
          } else {
            const u32x16 av = __builtin_bit_cast(u32x16, acc2[rt][ct]);   // (the whole vector, then index: element-wise bit_cast of a vector element mis-folds to a splat)
#pragma unroll
            for (int r = 0; r < 16; r++)
              __builtin_amdgcn_raw_buffer_store_b32(av[r], srs, eo, ((r & 3) + 8 * (r >> 2)) * L * 4, 2);
          }
        }
      }
    }
    // the next tile's staging writes X buffer 0 (last read before chunk 6's barrier) and its gate writes the g image only after
    // eight more barriers: no extra barrier needed here
  }
}

#ifdef AP_TOOLS
static int g_no_q16 = 0;                                        // ap_debug_f32w_q16(0): the 4-byte epilogue for every clip length (A/B)
#else
static constexpr int g_no_q16 = 0;
#endif

bool resblock_f32w_serves(const ap_ctx *ctx, int B, int L) {
  if (ctx->cfg.precision != AP_PREC_F32 || ctx->f32_form != 1 || ctx->C != WC_ || ctx->S != WC_ || !ctx->w1w) return false;
  if ((size_t)2 * WC_ * (size_t)L * 4 >= ((size_t)1 << 31)) return false;   // (the pre-gate rows of the SAVE form: 2C rows per clip)
  return (long long)B * ((L + 2 * NP_ - 1) / (2 * NP_) + 1) < (1ll << 31);
}

// returns 1 if the shape is not served (caller: the direct-form kernel)
int launch_resblock_f32w(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip, int accumulate,
                         int B, int L, hipStream_t st, float *aout) {
  if (!resblock_f32w_serves(ctx, B, L)) return 1;
  const int g_ncu = device_cu_count();
  const int logd = layer % ctx->cfg.dilation_cycle;
  const long long d = 1ll << logd;
  const long long np = (L / (2 * d)) * d + ((L % (2 * d)) < d ? (L % (2 * d)) : d);      // pairs whose first output is inside the clip
  const int ntiles = (int)((np + NP_ - 1) / NP_);
  const long long nblk = (long long)B * ntiles;
  if (nblk >= (1ll << 31)) return 1;
  const int C = ctx->C, S = ctx->S;
  const float *w1w = ctx->w1w + (size_t)layer * 4 * 2 * C * C;
  const float *w2w = ctx->w2w + (size_t)layer * (C + S) * C;
  const float *b1 = ctx->b1 + (size_t)layer * 2 * C;
  const float *b2 = ctx->b2 + (size_t)layer * (C + S);
  const unsigned grid = (unsigned)(nblk < g_ncu ? nblk : g_ncu);
  const bool q16 = (L & 3) == 0 && !g_no_q16;
  if (aout) {
    if (!hout) { set_error("resblock (save): needs an h' buffer"); return -22; }
    if (q16) resblock_f32w_kernel<false, true, true><<<grid, 256, 0, st>>>(hin, pt, hout, skip, w1w, b1, w2w, b2, L, logd, accumulate, ntiles, (int)nblk, aout);
    else resblock_f32w_kernel<false, true><<<grid, 256, 0, st>>>(hin, pt, hout, skip, w1w, b1, w2w, b2, L, logd, accumulate, ntiles, (int)nblk, aout);
    AP_HIP(hipGetLastError());
    return 0;
  }
  if (hout && q16) resblock_f32w_kernel<false, false, true><<<grid, 256, 0, st>>>(hin, pt, hout, skip, w1w, b1, w2w, b2, L, logd, accumulate, ntiles, (int)nblk, nullptr);
  else if (hout) resblock_f32w_kernel<false><<<grid, 256, 0, st>>>(hin, pt, hout, skip, w1w, b1, w2w, b2, L, logd, accumulate, ntiles, (int)nblk, nullptr);
  else if (q16) resblock_f32w_kernel<true, false, true><<<grid, 256, 0, st>>>(hin, pt, hout, skip, w1w, b1, w2w, b2, L, logd, accumulate, ntiles, (int)nblk, nullptr);
  else resblock_f32w_kernel<true><<<grid, 256, 0, st>>>(hin, pt, hout, skip, w1w, b1, w2w, b2, L, logd, accumulate, ntiles, (int)nblk, nullptr);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap

#ifdef AP_TOOLS
extern "C" int ap_debug_f32w_q16(int on) {
  ap::g_no_q16 = on ? 0 : 1;
  return 0;
}
#endif
