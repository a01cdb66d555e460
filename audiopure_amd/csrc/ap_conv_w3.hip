// 3 x 3, stride 1, pad 1, ungrouped nn.Conv2d (the UNet's ResBlock / Up / Downsample convolutions, improved_diffusion/unet.py:60-104,
// 150-197; BASELINE configs[4]) in F(2,3) minimal-filtering form along W, on the exact-fp32 matrix instruction.
//
// Outputs (y, x0) and (y, x0 + 1), x0 even, share the four columns x0-1 .. x0+2 of each of the three input rows y-1, y, y+1:
//     m1 = sum_ky W[ky][0] (d0 - d2)   m2 = sum_ky (W[ky][0]+W[ky][1]+W[ky][2])/2 (d1 + d2)
//     m3 = sum_ky (W[ky][0]-W[ky][1]+W[ky][2])/2 (d2 - d1)   m4 = sum_ky W[ky][2] (d3 - d1)
//     out[x0] = (m1 + m2) + m3 + bias      out[x0+1] = (m2 - m3) + m4 + bias
// 12 instead of 18 multiplications per output pair and input channel: 2/3 of the direct form's matrix work; the implicit GEMM is
// M = Cout, N = B H W/2 pair columns, K = 3 Cin per product (k = (ky, ci)).  Same machine shape as ap_resblock_f32w.hip /
// ap_resblock_bwd.hip: one persistent workgroup per CU, four 512-register waves, 256 accumulator registers per wave (4 products x 4
// tiles of 32 x 32), K chunks of 32 staged through LDS as a [product][column][k] image with 16-byte fragment reads, weights as
// pre-transformed A fragments streamed L2 -> registers, epilogue = output transform + bias + residual + ReLU with one 8-byte
// store per (row, pair).  Transformed weights are computed in double at pack time.
#include "ap_common.h"

namespace ap {

namespace {
constexpr int ZSW_ = 36;                  // floats per column row of a 32-k chunk image
constexpr unsigned FRAGW_ = 64 * 16;      // bytes of one row tile's fragment of a k-group
}  // namespace

bool conv_w3_serves(int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int groups) {
  return kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && Cin % 32 == 0 && Cout % 128 == 0 && (W & 1) == 0 && W >= 2 &&
         H >= 1;
}
size_t conv_w3_elems(int Cout, int Cin) { return (size_t)4 * Cout * Cin * 3; }
// persistent one-workgroup-per-CU kernel: worth it from two tiles per CU on (measured at B = 256: the UNet's 32 x 32 and 16 x 16 maps
// 1.15-1.36 x faster than the direct kernels, its 8 x 8 / 4 x 4 maps -- 128 / 32 tiles -- 1.15-3 x slower: those keep the direct kernels)
long long conv_w3_tiles(int B, int H, int W, int Cout) {
  const int rt = Cout % 256 == 0 ? 2 : 1, ncol = rt == 2 ? 64 : 128;
  const long long npairs = (long long)B * H * (W / 2);
  return (long long)(Cout / (128 * rt)) * ((npairs + ncol - 1) / ncol);
}
bool conv_w3_worth(int B, int H, int W, int Cout) { return conv_w3_tiles(B, H, W, Cout) >= 512; }
// K slices for a layer with too few tiles (the UNet's 8 x 8 and 4 x 4 maps: 128 / 32 tiles): each slice sums its share of the
// (ky, channel-block) chunks and writes its TRANSFORMED partial outputs (the output transform is linear) to the caller's
// workspace; the conv launcher's reduce kernel adds the slices in order (deterministic), then bias / residual / ReLU.
// 0: not worth it (under 32 tiles, or fewer than 3 chunks per slice)
int conv_w3_splits(int B, int Cin, int H, int W, int Cout, size_t ws_bytes) {
  const long long t = conv_w3_tiles(B, H, W, Cout);
  if (t >= 512 || t < 32) return 0;
  const int nch = 3 * (Cin / 32);
  int S = 2;
  while (S < 8 && t * S < 512) S *= 2;
  while (S > 1 && (nch / S < 3 || (size_t)S * B * Cout * H * W * sizeof(float) > ws_bytes)) S /= 2;
  return S > 1 ? S : 0;
}
// rows per workgroup: 256 (RT = 2 row tiles per wave) where Cout allows it, else 128 (RT = 1, four column tiles per wave)
static int conv_w3_rt(int Cout) { return Cout % 256 == 0 ? 2 : 1; }

// image: [row block][wave 4][chunk 3 Cin/32][k-group 4][product 4][row tile RT][lane 64][4]; row = 128 RT rb + 32 RT wave + 32 rt + i;
// chunk = ky (Cin / 32) + cb; k = input channel 32 cb + 8 kg + 4 hh + e
__global__ void conv_pack_w3_kernel(const float *__restrict__ w, const float *__restrict__ scale, float *__restrict__ out, int Cout,
                                    int Cin, int RT) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)4 * Cout * Cin * 3;
  if (idx >= total) return;
  const int e = idx & 3, lane = (idx >> 2) & 63;
  size_t rest = idx >> 8;
  const int rt = rest % RT; rest /= RT;
  const int comp = rest & 3; rest >>= 2;
  const int kg = rest & 3; rest >>= 2;
  const int nch = 3 * (Cin / 32);
  const int ch = rest % nch; rest /= nch;
  const int wv = rest & 3; rest >>= 2;
  const int rb = (int)rest;
  const int i = lane & 31, hh = lane >> 5;
  const int ky = ch / (Cin / 32), cb = ch % (Cin / 32);
  const int ci = 32 * cb + 8 * kg + 4 * hh + e;
  const int co = 128 * RT * rb + 32 * RT * wv + 32 * rt + i;
  const float *p = w + (((size_t)co * Cin + ci) * 3 + ky) * 3;
  const double s = scale ? (double)scale[co] : 1.0;
  const double w0 = p[0] * s, w1 = p[1] * s, w2 = p[2] * s;
  const double v = comp == 0 ? w0 : comp == 1 ? (w0 + w1 + w2) * 0.5 : comp == 2 ? (w0 - w1 + w2) * 0.5 : w2;
  out[idx] = (float)v;
}

int launch_conv_pack_w3(const float *w, const float *scale, float *out, int Cout, int Cin, hipStream_t st) {
  const size_t n = conv_w3_elems(Cout, Cin);
  conv_pack_w3_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w, scale, out, Cout, Cin, conv_w3_rt(Cout));
  AP_HIP(hipGetLastError());
  return 0;
}

struct ConvW3Args {
  const float *x, *wimg, *bias, *res;
  float *out;
  int B, Cin, H, W, Cout, relu, x_cstride, x_coff, o_cstride, o_coff;
  int npairs, ntile_n, nblk;     // pair columns B H W/2; column tiles; total tiles = row blocks x column tiles x K slices
  int splits;                    // > 1: K slices; raw transformed partial sums go to part [splits][B][Cout][H][W]
  float *part;
};

// RT row tiles x CT column tiles per wave (RT CT = 4): workgroup tile = (128 RT rows) x (32 CT pair columns)
// NB (W / 2 a power of two <= 64: every map of the UNet): a wave's 64 staging lanes are whole image rows of consecutive pairs, so a
// lane loads only its own two columns (x0, x0 + 1: one aligned 8-byte load) and takes x0 - 1 / x0 + 2 from its neighbours' registers
// (v_mov_b32_dpp wave_shr / wave_shl; zero at the row's ends) -- a quarter of the gather's load instructions.
template <int RT, bool NB>
__global__ __launch_bounds__(256, 1) void conv2d_w3_kernel(ConvW3Args a) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  constexpr int CT = 4 / RT, NCOL = 32 * CT;
  constexpr int XCOMP = NCOL * ZSW_, XBUF = 4 * XCOMP;
  __shared__ __attribute__((aligned(16))) float lds[2 * XBUF];   // RT = 2: 72 KB, RT = 1: 144 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int HW = a.H * a.W, W2 = a.W >> 1, HW2 = a.H * W2;
  const int CB = a.Cin / 32, NCH = 3 * CB;

  int t_first, t_step, t_end;
  {
    const int g = blockIdx.x, G = gridDim.x, nblk = a.nblk;
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
  const unsigned xbytes = (unsigned)((size_t)a.B * a.x_cstride * HW * 4);
  const unsigned obytes = (unsigned)((size_t)a.B * a.o_cstride * HW * 4);
  const unsigned rbytes = (unsigned)((size_t)a.B * a.Cout * HW * 4);
  const __amdgpu_buffer_rsrc_t xrs = uni_rsrc(a.x, xbytes);
  const unsigned wave_bytes = (unsigned)NCH * 4 * 4 * RT * FRAGW_;
  const unsigned lane16 = (unsigned)lane * 16u;

  // staging: thread = (column sj of NCOL, channel quad sq): NCOL / 64 column passes x (2 / (NCOL / 64 ... )) -- laid out below
  constexpr int TPC = 256 / NCOL;          // threads per column: 4 (NCOL = 64) or 2 (NCOL = 128)
  constexpr int QP = 8 / TPC;              // channel quads per thread and chunk: 2 or 4
  const int sj = tid % NCOL, sq = tid / NCOL;

#pragma unroll 1
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int zs = __builtin_amdgcn_readfirstlane(tile % a.splits);            // K slice (fastest: the slices of an output tile run side by side)
    const int tl = tile / a.splits;
    const int rb = __builtin_amdgcn_readfirstlane(tl / a.ntile_n);
    const int n0 = __builtin_amdgcn_readfirstlane((tl % a.ntile_n) * NCOL);
    const int c0 = __builtin_amdgcn_readfirstlane((int)((long long)NCH * zs / a.splits));
    const int c1 = __builtin_amdgcn_readfirstlane((int)((long long)NCH * (zs + 1) / a.splits));
    const __amdgpu_buffer_rsrc_t wrs =
        uni_rsrc(reinterpret_cast<const char *>(a.wimg) + ((size_t)rb * 4 + wave) * wave_bytes, wave_bytes);
    auto load_a = [&](f32x4(&aa)[RT], unsigned unit) {           // one (k-group, product) unit: RT row tiles
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
        aa[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + rt * FRAGW_, unit * RT * FRAGW_, 0));
    };
    // this thread's pair column: image, row, first column of the pair; element offsets of the three input rows' x0 (or out of range)
    unsigned vrow[3];
    bool okl, okr;                                              // columns x0 - 1 / x0 + 2 inside the image
    {
      const int n = n0 + sj;
      const bool nv = n < a.npairs;
      const int bb = nv ? n / HW2 : 0, rem = nv ? n - bb * HW2 : 0;
      const int y = rem / W2, x0 = 2 * (rem - y * W2);
      okl = x0 > 0;
      okr = x0 + 2 < a.W;
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int yy = y + ky - 1;
        vrow[ky] = (nv && yy >= 0 && yy < a.H)
                       ? (unsigned)((((size_t)bb * a.x_cstride + a.x_coff + 4 * sq) * HW + (size_t)yy * a.W + x0) * 4)
                       : 0x80000000u;
      }
    }
    float xr[NB ? 1 : QP][4][4];                                // [quad pass][channel][tap]  (gather form)
    f32x2 xn[NB ? QP : 1][4];                                   // [quad pass][channel] columns (x0, x0 + 1)  (neighbour form)
    auto issue_x = [&](int ch) {
      const int ky = ch / CB, cb = ch - ky * CB;
      const unsigned v = ky == 0 ? vrow[0] : ky == 1 ? vrow[1] : vrow[2];
      const bool rok = v != 0x80000000u;                         // (the input row is inside the image)
      if constexpr (NB) {
#pragma unroll
        for (int ps = 0; ps < QP; ps++)
#pragma unroll
          for (int cc = 0; cc < 4; cc++)
            xn[ps][cc] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, v, (32 * cb + 4 * TPC * ps + cc) * HW * 4, 0));
        return;
      }
      const unsigned vl = (okl && rok) ? v - 4u : 0x80000000u, vr = (okr && rok) ? v + 8u : 0x80000000u, vm = rok ? v + 4u : 0x80000000u;
#pragma unroll
      for (int ps = 0; ps < (NB ? 1 : QP); ps++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
          const int so = (32 * cb + 4 * TPC * ps + cc) * HW * 4;
          xr[ps][cc][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vl, so, 0));
          xr[ps][cc][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, v, so, 0));
          xr[ps][cc][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vm, so, 0));
          xr[ps][cc][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vr, so, 0));
        }
    };
    auto store_x = [&](float *dst) {
#pragma unroll
      for (int ps = 0; ps < QP; ps++) {
        f32x4 c0, c1, c2, c3;
        if constexpr (NB) {
          asm volatile("" : "+v"(xn[ps][0]), "+v"(xn[ps][1]), "+v"(xn[ps][2]), "+v"(xn[ps][3]));
#pragma unroll
          for (int cc = 0; cc < 4; cc++) {
            const float d1 = xn[ps][cc][0], d2 = xn[ps][cc][1];
            // lane i's x0 - 1 is lane i - 1's x0 + 1, its x0 + 2 is lane i + 1's x0 (same image row; zero at the row's ends)
            const float fl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d2), 0x138, 0xf, 0xf, true));
            const float fr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d1), 0x130, 0xf, 0xf, true));
            const float d0 = okl ? fl : 0.f, d3 = okr ? fr : 0.f;
            c0[cc] = d0 - d2;
            c1[cc] = d1 + d2;
            c2[cc] = d2 - d1;
            c3[cc] = d3 - d1;
          }
        } else {
          asm volatile("" : "+v"(xr[ps][0][0]), "+v"(xr[ps][0][1]), "+v"(xr[ps][0][2]), "+v"(xr[ps][0][3]), "+v"(xr[ps][1][0]), "+v"(xr[ps][1][1]),
                       "+v"(xr[ps][1][2]), "+v"(xr[ps][1][3]), "+v"(xr[ps][2][0]), "+v"(xr[ps][2][1]), "+v"(xr[ps][2][2]), "+v"(xr[ps][2][3]),
                       "+v"(xr[ps][3][0]), "+v"(xr[ps][3][1]), "+v"(xr[ps][3][2]), "+v"(xr[ps][3][3]));
#pragma unroll
          for (int cc = 0; cc < 4; cc++) {
            c0[cc] = xr[ps][cc][0] - xr[ps][cc][2];
            c1[cc] = xr[ps][cc][1] + xr[ps][cc][2];
            c2[cc] = xr[ps][cc][2] - xr[ps][cc][1];
            c3[cc] = xr[ps][cc][3] - xr[ps][cc][1];
          }
        }
        float *q = dst + sj * ZSW_ + 4 * TPC * ps + 4 * sq;
        *reinterpret_cast<f32x4 *>(q) = c0;
        *reinterpret_cast<f32x4 *>(q + XCOMP) = c1;
        *reinterpret_cast<f32x4 *>(q + 2 * XCOMP) = c2;
        *reinterpret_cast<f32x4 *>(q + 3 * XCOMP) = c3;
      }
    };
    // (channel of (ps, cc) for thread sq: 4 TPC ps + 4 sq' + cc with sq' = sq: the vrow offsets carry 4 sq channels, the scalar
    //  offset 4 TPC ps + cc)

    f32x4 aw[4][RT];
#pragma unroll
    for (int u = 0; u < 4; u++) load_a(aw[u], (unsigned)(16 * c0 + u));
    issue_x(c0);
    f32x16 acc[4][RT][CT];
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[c4][rt][ct][r] = 0.f;
    store_x(lds);
    __syncthreads();

    const float *xfrag = lds + j * ZSW_ + 4 * hh;
    const unsigned nunit = (unsigned)NCH * 16u;
#pragma unroll 1
    for (int ch = c0; ch < c1; ch++) {
      const float *xb = xfrag + ((ch - c0) & 1) * XBUF;
      issue_x(ch + 1 < c1 ? ch + 1 : ch);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 4; kg++) {
#pragma unroll
        for (int comp = 0; comp < 4; comp++) {
          const int u = 4 * kg + comp;
          f32x4 bq[CT];
#pragma unroll
          for (int ct = 0; ct < CT; ct++) bq[ct] = *reinterpret_cast<const f32x4 *>(xb + comp * XCOMP + 32 * ct * ZSW_ + kg * 8);
          if (u == 12) store_x(lds + ((ch - c0 + 1) & 1) * XBUF);
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
              for (int ct = 0; ct < CT; ct++)
                acc[comp][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[comp][rt][e], bq[ct][e], acc[comp][rt][ct], 0, 0, 0);
          if (u == 12) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, (QP == 2 ? 3 : 5) + (NB ? QP : 0), 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 4 * QP, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          {
            unsigned nu = (unsigned)(16 * ch + u + 4);
            nu = nu < nunit ? nu : nu - nunit;                   // (the last k-group wraps to the image's first units, unused)
            load_a(aw[comp], nu);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }

    // ---- output transform, bias, residual, ReLU; lane (j, hh) of column tile ct holds pair column n0 + 32 ct + j, rows crowoff
    const __amdgpu_buffer_rsrc_t ors = uni_rsrc(a.out, obytes);
    const __amdgpu_buffer_rsrc_t rrs = uni_rsrc(a.res ? a.res : a.out, a.res ? rbytes : 0u);
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < CT; ct++) {
        asm volatile("" : "+a"(acc[0][rt][ct]), "+a"(acc[1][rt][ct]), "+a"(acc[2][rt][ct]), "+a"(acc[3][rt][ct]));
        const int n = n0 + 32 * ct + j;
        const bool nv = n < a.npairs;
        const int bb = nv ? n / HW2 : 0, rem = nv ? n - bb * HW2 : 0;
        const int pix = 2 * rem;                                // y W + x0  (rem = y W/2 + x0/2)
        const int co0 = 128 * RT * rb + 32 * RT * wave + 32 * rt + 4 * hh;
        const unsigned eo = nv ? (unsigned)((((size_t)bb * a.o_cstride + a.o_coff + co0) * HW + pix) * 4) : 0x80000000u;
        const unsigned er = nv ? (unsigned)((((size_t)bb * a.Cout + co0) * HW + pix) * 4) : 0x80000000u;
        if (a.splits > 1) {                                     // raw partial of this K slice: [slice][B][Cout][H][W]
          const __amdgpu_buffer_rsrc_t prs = uni_rsrc(a.part + (size_t)zs * a.B * a.Cout * HW, rbytes);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const f32x2 o = {(acc[0][rt][ct][r] + acc[1][rt][ct][r]) + acc[2][rt][ct][r], (acc[1][rt][ct][r] - acc[2][rt][ct][r]) + acc[3][rt][ct][r]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), prs, er, ro * HW * 4, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        f32x2 rv[16];
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ro = (r & 3) + 8 * (r >> 2);
          bv[r] = a.bias ? a.bias[co0 + ro] : 0.f;
          rv[r] = a.res ? __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rrs, er, ro * HW * 4, 0)) : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ro = (r & 3) + 8 * (r >> 2);
          float y0 = (acc[0][rt][ct][r] + acc[1][rt][ct][r]) + acc[2][rt][ct][r];
          float y1 = (acc[1][rt][ct][r] - acc[2][rt][ct][r]) + acc[3][rt][ct][r];
          y0 = (y0 + bv[r]) + rv[r][0];                          // (bias, then residual: the order of the direct kernels)
          y1 = (y1 + bv[r]) + rv[r][1];
          if (a.relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
          const f32x2 o = {y0, y1};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ors, eo, ro * HW * 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
}

static int g_ncu_w3 = 0;
#ifdef AP_TOOLS
static int g_w3_nb = 1;
#else
static constexpr int g_w3_nb = 1;
#endif

int launch_conv_w3(const float *x, const float *wimg, const float *bias, const float *res, float *out, int B, int Cin, int H, int W,
                   int Cout, int relu, int x_cstride, int x_coff, int o_cstride, int o_coff, hipStream_t st, int splits, float *part) {
  if (g_ncu_w3 == 0) {
    int dev = 0, n = 0;
    AP_HIP(hipGetDevice(&dev));
    AP_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    g_ncu_w3 = n > 0 ? n : 256;
  }
  const int RT = conv_w3_rt(Cout);
  ConvW3Args a;
  a.x = x; a.wimg = wimg; a.bias = bias; a.res = res; a.out = out;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.relu = relu & 1;
  a.x_cstride = x_cstride; a.x_coff = x_coff; a.o_cstride = o_cstride; a.o_coff = o_coff;
  a.npairs = B * H * (W / 2);
  const int ncol = RT == 2 ? 64 : 128;
  a.ntile_n = (a.npairs + ncol - 1) / ncol;
  a.splits = splits > 1 ? splits : 1;
  a.part = part;
  const long long nblk = (long long)(Cout / (128 * RT)) * a.ntile_n * a.splits;
  a.nblk = (int)nblk;
  const unsigned grid = (unsigned)(nblk < g_ncu_w3 ? nblk : g_ncu_w3);
  const int w2 = W / 2;
  const bool nb = g_w3_nb && w2 <= 64 && (w2 & (w2 - 1)) == 0;    // whole image rows per 64 staging lanes: the neighbour form
  if (RT == 2) {
    if (nb) conv2d_w3_kernel<2, true><<<grid, 256, 0, st>>>(a);
    else conv2d_w3_kernel<2, false><<<grid, 256, 0, st>>>(a);
  } else {
    if (nb) conv2d_w3_kernel<1, true><<<grid, 256, 0, st>>>(a);
    else conv2d_w3_kernel<1, false><<<grid, 256, 0, st>>>(a);
  }
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap

#ifdef AP_TOOLS
extern "C" int ap_debug_conv_w3_nb(int on) {                     // A/B: the neighbour staging form on / off
  ap::g_w3_nb = on;
  return 0;
}
#endif
