// Input gradient of one Residual_block.forward (WaveNet.py:75-97), AP_PREC_F32, shipped shape (res = skip = 256 channels): the
// reverse sweep of the differentiable purifier (the reference's white-box attack back-propagates through the defender:
// robustness_eval/white_box_attack.py:392,437-439; diffusion_models/diffwave_sde.py:200-204) as TWO fused launches per layer.
//
//   forward:   u = h + part_t;  y = DilConv_d(u) + b1;  g = tanh(y_t) sigmoid(y_s);  h' = (u + W_res g + b_res) sqrt(1/2);
//              skip_n = W_skip g + b_skip
//   backward:  dg = W_res^T (sqrt(1/2) dh') + W_skip^T dskip
//              dy_t = dg sigmoid(y_s) (1 - tanh(y_t)^2);   dy_s = dg tanh(y_t) sigmoid(y_s) (1 - sigmoid(y_s))
//              dh  = sqrt(1/2) dh' + DilConv_d^T(dy)                      (a k = 3 dilated conv 2C -> C with flipped taps)
//
// K1  resblock_bwd_gate_kernel:  dy = gate'(y) . ([W_res sqrt(1/2); W_skip]^T [dh'; dskip])  -- a [C x (C+S)].[(C+S) x 64] GEMM per
//     tile on v_mfma_f32_32x32x2_f32 with the gate's derivative as its epilogue (y: the pre-gate activations the forward pass kept,
//     ap_resblock_fwd_save); 4 waves x (64 rows x 64 columns), two workgroups per CU.
// K2  resblock_bwd_conv_kernel:  dh = sqrt(1/2) dh' + DilConv^T(dy) in the F(2,3) minimal-filtering form of ap_resblock_f32w.hip
//     (4 instead of 6 [C x 2C] products per dilation pair): one persistent workgroup per CU, four 512-register waves, each
//     64 rows x 64 pair columns x 4 products = 256 accumulator registers; a tile is 64 pairs = 128 outputs.
// Together 12.58 GFLOP per clip and layer: the forward block's work.  dy lives in a caller-owned scratch tensor between the two.
#include "ap_common.h"

namespace ap {

namespace {

constexpr int BC_ = 256;                 // res = skip channels
constexpr int ZS_ = 36;                  // floats per column row of a 32-k chunk image (32 + 4 pad: conflict-free 16-byte accesses)
constexpr unsigned FRAG_ = 64 * 16;      // one row tile's fragment of a k-group: 64 lanes x 16 B
// K2 geometry
constexpr int NPB_ = 64;                 // pairs per tile
constexpr int KCB_ = 32;                 // input channels per staged chunk (of 2C = 512)
constexpr int NCHB_ = 2 * BC_ / KCB_;    // 16 chunks
constexpr int XCOMPB_ = NPB_ * ZS_;      // one product's chunk image
constexpr int XBUFB_ = 4 * XCOMPB_;      // one chunk (36 KB)

}  // namespace

// K1 image: [wave 4][k-group 64][row tile 2][lane 64][4]; row c = 64 wave + 32 rt + i; k = 8 kg + 4 hh + e over the concatenation
// [res_conv output o (256); skip_conv output s (256)]; value W2[k][c], the res half times sqrt(1/2) (h' = (u + res) sqrt(1/2)).
__global__ void pack_w2t_kernel(const float *__restrict__ w2f, float *__restrict__ out) {
  constexpr int C = BC_;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4u * 64 * 2 * 64 * 4) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, rt = (idx >> 8) & 1, kg = (idx >> 9) & 63, w = idx >> 15;
  const int i = lane & 31, hh = lane >> 5;
  const int k = 8 * kg + 4 * hh + e;
  const int c = 64 * w + 32 * rt + i;
  const float v = w2f[(size_t)k * C + c];
  out[idx] = k < C ? (float)((double)v * 0.70710678118654752440) : v;
}

// K2 image: [wave 4][chunk 16][k-group 4][product 4][row tile 2][lane 64][4]; row c = 64 wave + 32 rt + i; k = pre-gate channel
// o = 32 chunk + 8 kg + 4 hh + e.  The transposed conv's taps are the forward's flipped: Wf[c][o][tap'] = W1[o][c][2 - tap'].
__global__ void pack_w1b_kernel(const float *__restrict__ w1f, float *__restrict__ out) {
  constexpr int C = BC_;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4u * NCHB_ * 4 * 4 * 2 * 64 * 4) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, rt = (idx >> 8) & 1, comp = (idx >> 9) & 3, kg = (idx >> 11) & 3,
            ch = (idx >> 13) & 15, w = idx >> 17;
  const int i = lane & 31, hh = lane >> 5;
  const int o = 32 * ch + 8 * kg + 4 * hh + e;
  const int c = 64 * w + 32 * rt + i;
  const float *p = w1f + ((size_t)o * C + c) * 3;
  const double f0 = p[2], f1 = p[1], f2 = p[0];                  // flipped taps
  const double v = comp == 0 ? f0 : comp == 1 ? (f0 + f1 + f2) * 0.5 : comp == 2 ? (f0 - f1 + f2) * 0.5 : f2;
  out[idx] = (float)v;
}

int launch_pack_bwd(ap_ctx *ctx, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  const size_t n1f = (size_t)2 * C * C * 3, n2 = (size_t)(C + S) * C, n1b = (size_t)4 * 2 * C * C;
  for (int n = 0; n < ctx->NL; n++) {
    pack_w2t_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(ctx->w2f + n * n2, ctx->w2t + n * n2);
    pack_w1b_kernel<<<(unsigned)((n1b + 255) / 256), 256, 0, st>>>(ctx->w1f + n * n1f, ctx->w1b + n * n1b);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// K1: dy = gate'(y) . (W2^T [dh'; dskip])
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void resblock_bwd_gate_kernel(const float *__restrict__ dh, const float *__restrict__ dskip,
                                                                   const float *__restrict__ pre, float *__restrict__ dy,
                                                                   const float *__restrict__ w2t, int L, int ntiles) {
  constexpr int C = BC_;
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * ZS_];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / ntiles));
  const int t0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % ntiles) * 64);
  auto uni_rsrc = [&](const void *base, unsigned bytes) {
    const uint64_t hb = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  const float *dh_b = dh + (size_t)b * C * L, *ds_b = dskip + (size_t)b * C * L;
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc(reinterpret_cast<const char *>(w2t) + (size_t)wave * 64 * 2 * FRAG_, 64u * 2 * FRAG_);
  const unsigned lane16 = (unsigned)lane * 16u;

  // staging: thread = (column sj of 64, row octet sq of 4): 8 rows of a 32-row chunk, two ds_write_b128
  const int sj = tid & 63, sq = tid >> 6;
  const int ts = t0 + sj;
  const unsigned zv = ts < L ? ((unsigned)ts + (unsigned)(8 * sq) * (unsigned)L) * 4u : 0x80000000u;   // past the clip: the range check returns 0
  float zr[8];
  auto issue_z = [&](int kc) {                                   // chunks 0..7: dh' rows, 8..15: dskip rows
    const __amdgpu_buffer_rsrc_t rs = uni_rsrc(kc < 8 ? dh_b : ds_b, clip_bytes);
    const int so = ((kc & 7) * 32) * L * 4;
#pragma unroll
    for (int i = 0; i < 8; i++) zr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, zv, so + i * L * 4, 0));
  };
  auto store_z = [&](float *dst) {
    asm volatile("" : "+v"(zr[0]), "+v"(zr[1]), "+v"(zr[2]), "+v"(zr[3]), "+v"(zr[4]), "+v"(zr[5]), "+v"(zr[6]), "+v"(zr[7]));
    float *q = dst + sj * ZS_ + 8 * sq;
    *reinterpret_cast<f32x4 *>(q) = f32x4{zr[0], zr[1], zr[2], zr[3]};
    *reinterpret_cast<f32x4 *>(q + 4) = f32x4{zr[4], zr[5], zr[6], zr[7]};
  };
  auto load_a = [&](f32x4(&a)[2], unsigned kg) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
      a[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + rt * FRAG_, kg * 2 * FRAG_, 0));
  };

  f32x4 a[4][2];                                                 // ring of one chunk (four k-groups)
#pragma unroll
  for (int kg = 0; kg < 4; kg++) load_a(a[kg], (unsigned)kg);
  issue_z(0);
  f32x16 acc[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[rt][ct][r] = 0.f;
  store_z(lds);
  __syncthreads();

  const float *zfrag = lds + j * ZS_ + 4 * hh;
#pragma unroll 1
  for (int kc = 0; kc < 16; kc++) {
    const float *zb = zfrag + (kc & 1) * 64 * ZS_;
    issue_z(kc + 1 < 16 ? kc + 1 : kc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kg = 0; kg < 4; kg++) {
      f32x4 bq[2];
      bq[0] = *reinterpret_cast<const f32x4 *>(zb + 8 * kg);
      bq[1] = *reinterpret_cast<const f32x4 *>(zb + 32 * ZS_ + 8 * kg);
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg][rt][e], bq[ct][e], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_a(a[kg], (unsigned)((4 * kc + kg + 4) & 63));          // (the last chunk wraps, unused)
      if (kg == 2) store_z(lds + ((kc + 1) & 1) * 64 * ZS_);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // epilogue: the gate's derivative (WaveNet.py:90) on the kept pre-gate activations; rows of the tanh half, then the sigmoid half
  const __amdgpu_buffer_rsrc_t prs = uni_rsrc(pre + (size_t)b * 2 * C * L, 2u * clip_bytes);
  const __amdgpu_buffer_rsrc_t ors = uni_rsrc(dy + (size_t)b * 2 * C * L, 2u * clip_bytes);
  const unsigned half = (unsigned)C * (unsigned)L * 4u;
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const int t = t0 + 32 * ct + j;
      const unsigned eo = t < L ? ((unsigned)(64 * wave + 32 * rt + 4 * hh) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      float yt[16], ys[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ro = ((r & 3) + 8 * (r >> 2)) * L * 4;
        yt[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, eo, ro, 0));
        ys[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, eo + half, ro, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ro = ((r & 3) + 8 * (r >> 2)) * L * 4;
        const float at = fminf(fmaxf(yt[r], -15.0f), 15.0f);
        const float E = exp_acc(2.0f * at), F = exp_acc(-fmaxf(ys[r], -80.0f));
        const float R = __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
        const float th = (E - 1.0f) * (1.0f + F) * R, sg = (E + 1.0f) * R;
        const float g = acc[rt][ct][r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g * sg * (1.0f - th * th)), ors, eo, ro, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g * th * sg * (1.0f - sg)), ors, eo + half, ro, 0);
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K2: dh = sqrt(1/2) dh' + DilConv^T(dy), F(2,3) over the dilation pair (see ap_resblock_f32w.hip for the pairing)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void resblock_bwd_conv_kernel(const float *__restrict__ dy, const float *__restrict__ dhp,
                                                                   float *__restrict__ dhin, const float *__restrict__ w1b,
                                                                   int L, int logd, int ntiles, int nblk) {
  constexpr int C = BC_;
  __shared__ __attribute__((aligned(16))) float lds[2 * XBUFB_];   // 72 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int d = 1 << logd;

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
  constexpr unsigned WAVE_BYTES = NCHB_ * 4 * 4 * 2 * FRAG_;     // 512 KB per wave and layer
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc(reinterpret_cast<const char *>(w1b) + (size_t)wave * WAVE_BYTES, WAVE_BYTES);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto load_a = [&](f32x4(&a)[2], unsigned unit) {               // one (k-group, product) unit: 2 row tiles
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
      a[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + rt * FRAG_, unit * 2 * FRAG_, 0));
  };

  // staging: thread = (pair sj of 64, channel quad sq of 4); a chunk's 32 channels = quads sq and sq + 4
  const int sj = tid & 63, sq = tid >> 6;

#pragma unroll 1
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int b = __builtin_amdgcn_readfirstlane(tile / ntiles);
    const int p0 = __builtin_amdgcn_readfirstlane((tile % ntiles) * NPB_);
    const __amdgpu_buffer_rsrc_t yrs = uni_rsrc(dy + (size_t)b * 2 * C * L, 2u * clip_bytes);
    unsigned voff[4];
    {
      const int p = p0 + sj;
      const int tf = ((p >> logd) << (logd + 1)) + (p & (d - 1));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int tp = tf + (k - 1) * d;                          // outside the clip: zero padding by the range check
        voff[k] = (tp >= 0 && tp < L) ? ((unsigned)tp + (unsigned)(4 * sq) * (unsigned)L) * 4u : 0x80000000u;
      }
    }
    float xr[2][4][4];                                           // [quad pass][channel][tap]
    auto issue_x = [&](int ch) {
#pragma unroll
      for (int ps = 0; ps < 2; ps++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
          for (int k = 0; k < 4; k++)
            xr[ps][cc][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, voff[k], (ch * KCB_ + 16 * ps + cc) * L * 4, 0));
    };
    auto store_x = [&](float *dst) {
#pragma unroll
      for (int ps = 0; ps < 2; ps++) {
        asm volatile("" : "+v"(xr[ps][0][0]), "+v"(xr[ps][0][1]), "+v"(xr[ps][0][2]), "+v"(xr[ps][0][3]), "+v"(xr[ps][1][0]), "+v"(xr[ps][1][1]),
                     "+v"(xr[ps][1][2]), "+v"(xr[ps][1][3]), "+v"(xr[ps][2][0]), "+v"(xr[ps][2][1]), "+v"(xr[ps][2][2]), "+v"(xr[ps][2][3]),
                     "+v"(xr[ps][3][0]), "+v"(xr[ps][3][1]), "+v"(xr[ps][3][2]), "+v"(xr[ps][3][3]));
        f32x4 c0, c1, c2, c3;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
          c0[cc] = xr[ps][cc][0] - xr[ps][cc][2];
          c1[cc] = xr[ps][cc][1] + xr[ps][cc][2];
          c2[cc] = xr[ps][cc][2] - xr[ps][cc][1];
          c3[cc] = xr[ps][cc][3] - xr[ps][cc][1];
        }
        float *q = dst + sj * ZS_ + 16 * ps + 4 * sq;
        *reinterpret_cast<f32x4 *>(q) = c0;
        *reinterpret_cast<f32x4 *>(q + XCOMPB_) = c1;
        *reinterpret_cast<f32x4 *>(q + 2 * XCOMPB_) = c2;
        *reinterpret_cast<f32x4 *>(q + 3 * XCOMPB_) = c3;
      }
    };

    f32x4 a[4][2];                                               // [product][row tile]: a ring one k-group deep
#pragma unroll
    for (int u = 0; u < 4; u++) load_a(a[u], (unsigned)u);
    issue_x(0);
    f32x16 acc[4][2][2];                                         // [product][row tile][column tile]
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[c4][rt][ct][r] = 0.f;
    store_x(lds);
    __syncthreads();

    const float *xfrag = lds + j * ZS_ + 4 * hh;
#pragma unroll 1
    for (int ch = 0; ch < NCHB_; ch++) {
      const float *xb = xfrag + (ch & 1) * XBUFB_;
      issue_x(ch + 1 < NCHB_ ? ch + 1 : ch);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 4; kg++) {
#pragma unroll
        for (int comp = 0; comp < 4; comp++) {
          const int u = 4 * kg + comp;
          f32x4 bq[2];
          bq[0] = *reinterpret_cast<const f32x4 *>(xb + comp * XCOMPB_ + kg * 8);
          bq[1] = *reinterpret_cast<const f32x4 *>(xb + comp * XCOMPB_ + 32 * ZS_ + kg * 8);
          if (u == 12) store_x(lds + ((ch + 1) & 1) * XBUFB_);
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
              for (int ct = 0; ct < 2; ct++)
                acc[comp][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[comp][rt][e], bq[ct][e], acc[comp][rt][ct], 0, 0, 0);
          if (u == 12) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 8, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          load_a(a[comp], (unsigned)((16 * ch + u + 4) & (16 * NCHB_ - 1)));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }

    // output transform, residual path, store: the wave's 64 rows x (64 first outputs, 64 second outputs)
    const __amdgpu_buffer_rsrc_t prs = uni_rsrc(dhp + (size_t)b * C * L, clip_bytes);
    const __amdgpu_buffer_rsrc_t ors = uni_rsrc(dhin + (size_t)b * C * L, clip_bytes);
    const float RS = 0.707106781186547524f;
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        asm volatile("" : "+a"(acc[0][rt][ct]), "+a"(acc[1][rt][ct]), "+a"(acc[2][rt][ct]), "+a"(acc[3][rt][ct]));
        const int p = p0 + 32 * ct + j;
        const int s0 = ((p >> logd) << (logd + 1)) + (p & (d - 1));
        const unsigned rb = (unsigned)(64 * wave + 32 * rt + 4 * hh) * (unsigned)L;
        const unsigned e0 = s0 < L ? (rb + (unsigned)s0) * 4u : 0x80000000u;
        const unsigned e1 = s0 + d < L ? (rb + (unsigned)(s0 + d)) * 4u : 0x80000000u;
        float r0[16], r1[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ro = ((r & 3) + 8 * (r >> 2)) * L * 4;
          r0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, e0, ro, 0));
          r1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, e1, ro, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ro = ((r & 3) + 8 * (r >> 2)) * L * 4;
          const float y0 = (acc[0][rt][ct][r] + acc[1][rt][ct][r]) + acc[2][rt][ct][r];
          const float y1 = (acc[1][rt][ct][r] - acc[2][rt][ct][r]) + acc[3][rt][ct][r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_fmaf(RS, r0[r], y0)), ors, e0, ro, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_fmaf(RS, r1[r], y1)), ors, e1, ro, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
}

bool resblock_bwd_serves(const ap_ctx *ctx, int B, int L) {
  if (ctx->cfg.precision != AP_PREC_F32 || ctx->C != BC_ || ctx->S != BC_ || !ctx->loaded) return false;
  if ((size_t)2 * BC_ * (size_t)L * 4 >= ((size_t)1 << 31)) return false;
  return (long long)B * ((L + 63) / 64 + 1) < (1ll << 31);
}

// dh_in = d loss / d (block input h), given dh' = d loss / d h', dskip = d loss / d skip_n and the kept pre-gate activations.
int launch_resblock_bwd(ap_ctx *ctx, int layer, const float *dhp, const float *dskip, const float *pre, float *dy, float *dhin,
                        int B, int L, hipStream_t st) {
  if (!resblock_bwd_serves(ctx, B, L)) {
    set_error("ap_resblock_bwd: built for AP_PREC_F32 with res = skip = 256 channels and clips below 2^20 samples");
    return -22;
  }
  if (!ctx->bwd_ready) {                                         // (launch functions allocate nothing: include/audiopure.h)
    set_error("ap_resblock_bwd: the backward weight images are not built (ap_ctx_prepare_backward after every ap_ctx_load_wavenet)");
    return -22;
  }
  const int g_ncu_b = device_cu_count();
  const int C = BC_;
  const int nt1 = (L + 63) / 64;
  resblock_bwd_gate_kernel<<<(unsigned)(B * nt1), 256, 0, st>>>(dhp, dskip, pre, dy, ctx->w2t + (size_t)layer * 2 * C * C, L, nt1);
  const int logd = layer % ctx->cfg.dilation_cycle;
  const long long d = 1ll << logd;
  const long long np = (L / (2 * d)) * d + ((L % (2 * d)) < d ? (L % (2 * d)) : d);
  const int ntiles = (int)((np + NPB_ - 1) / NPB_);
  const long long nblk = (long long)B * ntiles;
  const unsigned grid = (unsigned)(nblk < g_ncu_b ? nblk : g_ncu_b);
  resblock_bwd_conv_kernel<<<grid, 256, 0, st>>>(dy, dhp, dhin, ctx->w1b + (size_t)layer * 4 * 2 * C * C, L, logd, ntiles, (int)nblk);
  AP_HIP(hipGetLastError());
  return 0;
}

// The two backward weight images of this context (94 MB), built once per load: allocation + pack + a host synchronisation.  The
// pointers the launch functions look at are published only after the pack has completed.
int prepare_bwd_f32(ap_ctx *ctx, hipStream_t st) {
  if (ctx->bwd_ready) return 0;
  const size_t n1b = (size_t)ctx->NL * 4 * 2 * BC_ * BC_, n2 = (size_t)ctx->NL * 2 * BC_ * BC_;
  if (!ctx->slab_b) AP_HIP(hipMalloc(&ctx->slab_b, (n1b + n2) * sizeof(float)));
  ctx->w1b = (float *)ctx->slab_b;
  ctx->w2t = ctx->w1b + n1b;
  int rc = launch_pack_bwd(ctx, st);
  if (rc == 0) {
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize(prepare_backward)");
  }
  if (rc) {
    (void)hipFree(ctx->slab_b);
    ctx->slab_b = nullptr;
    ctx->w1b = ctx->w2t = nullptr;
    return rc;
  }
  ctx->bwd_ready = true;
  return 0;
}

}  // namespace ap

extern "C" int ap_resblock_bwd(ap_ctx *ctx, int layer, const float *dh_out, const float *dskip, const float *pre_gate, float *dy_scratch,
                               float *dh_in, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !dh_out || !dskip || !pre_gate || !dy_scratch || !dh_in) { ap::set_error("ap_resblock_bwd: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { ap::set_error("ap_resblock_bwd: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (dh_in == dh_out) { ap::set_error("ap_resblock_bwd: dh_in must not alias dh_out"); return -22; }
  return ap::launch_resblock_bwd(ctx, layer, dh_out, dskip, pre_gate, dy_scratch, dh_in, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_bwd_available(ap_ctx *ctx, int B, int L) { return ctx && ap::resblock_bwd_serves(ctx, B, L) ? 1 : 0; }
