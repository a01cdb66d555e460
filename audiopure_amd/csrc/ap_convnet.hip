// 2-D ConvNet classifier kernels (audio_models/ConvNets_SpeechCommands/models/*, fed by the
// mel front-end): conv-as-GEMM on the exact-fp32 MFMA, plus the small NCHW kernels a lowered network needs
// (per-channel affine = eval-mode BatchNorm, ReLU, add, pooling, channel copies for cat / slices).
//
// conv2d_f32_kernel: implicit GEMM per group,  M = Cout/g, N = B*Ho*Wo, K = (Cin/g)*kh*kw.
// Workgroup tile 64 (M) x 64 (N), 4 waves of 32x32 (v_mfma_f32_32x32x2_f32), K chunks of 16 staged in LDS
// ([k][m] weights pre-transposed at plan time, [k][n] im2col gather), register-prefetched one chunk ahead.
// Epilogue fuses the folded-BN bias, an optional residual tensor and ReLU.
#include "ap_common.h"

namespace ap {

constexpr int CBM = 64, CBN = 64, CBK = 16;

struct ConvArgs {
  const float *x, *wT, *bias, *res;
  float *out;
  int B, Cin, H, W, Cout, Ho, Wo, kh, kw, stride, pad, groups, relu;
  int x_cstride;     // channels of the tensor x lives in (>= Cin when x is a channel slice of a wider tensor)
  int x_coff;        // first channel of the slice
};

__device__ __forceinline__ int crowoff(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__global__ __launch_bounds__(256) void conv2d_f32_kernel(ConvArgs a) {
  __shared__ float As[2][CBK][CBM];
  __shared__ float Bs[2][CBK][CBN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int j = lane & 31, hh = lane >> 5;
  const int Mg = a.Cout / a.groups, Cg = a.Cin / a.groups, KK = a.kh * a.kw, Kg = Cg * KK;
  const int HoWo = a.Ho * a.Wo, N = a.B * HoWo;
  const int g = blockIdx.z, m0 = blockIdx.y * CBM, n0 = blockIdx.x * CBN;

  // B (im2col) loader: thread -> column n = tid & 63, k rows (tid >> 6) + 4 i
  const int nl = tid & 63, kq = tid >> 6;
  const int n = n0 + nl;
  const bool nvalid = n < N;
  const int bb = nvalid ? n / HoWo : 0, pp = nvalid ? n % HoWo : 0;
  const int iy0 = (pp / a.Wo) * a.stride - a.pad, ix0 = (pp % a.Wo) * a.stride - a.pad;
  const float *xb = a.x + ((size_t)bb * a.x_cstride + a.x_coff + (size_t)g * Cg) * a.H * a.W;
  // A loader: thread -> row m = tid & 63, k rows (tid >> 6) + 4 i ; wT is [groups][Kg][Mg]
  const float *wg = a.wT + (size_t)g * Kg * Mg;
  const bool mvalid = (m0 + nl) < Mg;

  float ar[4], br[4];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = k0 + kq + 4 * i;
      float av = 0.f, bv = 0.f;
      if (k < Kg) {
        if (mvalid) av = wg[(size_t)k * Mg + m0 + nl];
        const int ci = k / KK, r = k - ci * KK, ky = r / a.kw, kx = r - ky * a.kw;
        const int iy = iy0 + ky, ix = ix0 + kx;
        if (nvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) bv = xb[((size_t)ci * a.H + iy) * a.W + ix];
      }
      ar[i] = av;
      br[i] = bv;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      As[buf][kq + 4 * i][nl] = ar[i];
      Bs[buf][kq + 4 * i][nl] = br[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;

  const int nchunk = (Kg + CBK - 1) / CBK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunk; c++) {
    if (c + 1 < nchunk) load_chunk((c + 1) * CBK);
    const int buf = c & 1;
#pragma unroll
    for (int s = 0; s < CBK / 2; s++)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + hh][32 * wm + j], Bs[buf][2 * s + hh][32 * wn + j], acc, 0,
                                                 0, 0);
    if (c + 1 < nchunk) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane (j, hh) holds column n0 + 32 wn + j, rows m0 + 32 wm + crowoff(r, hh)
  const int nn = n0 + 32 * wn + j;
  if (nn < N) {
    const int ob = nn / HoWo, op = nn % HoWo;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + 32 * wm + crowoff(r, hh);
      if (m < Mg) {
        const int co = g * Mg + m;
        const size_t off = ((size_t)ob * a.Cout + co) * HoWo + op;
        float v = acc[r];
        if (a.bias) v += a.bias[co];
        if (a.res) v += a.res[off];
        if (a.relu) v = fmaxf(v, 0.f);
        a.out[off] = v;
      }
    }
  }
}

// w [Cout][Cin/g][kh][kw] (* per-output-channel scale) -> wT [groups][Kg][Mg]
__global__ void conv_pack_kernel(const float *__restrict__ w, const float *__restrict__ scale, float *__restrict__ wT,
                                 int Cout, int Kg, int groups) {
  const int Mg = Cout / groups;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Cout * Kg) return;
  const int m = idx % Mg;
  size_t rest = idx / Mg;
  const int k = rest % Kg, g = rest / Kg;
  const int co = g * Mg + m;
  float v = w[(size_t)co * Kg + k];
  if (scale) v *= scale[co];
  wT[idx] = v;
}

// y = x * scale[c] + shift[c] (eval-mode BatchNorm), optional ReLU; x may be a channel slice of a wider tensor
__global__ void affine_kernel(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                              float *__restrict__ y, int C, int HW, int x_cstride, int x_coff, int relu, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int p = idx % HW;
  size_t rest = idx / HW;
  const int c = rest % C;
  const size_t b = rest / C;
  float v = x[((size_t)b * x_cstride + x_coff + c) * HW + p];
  if (scale) v = v * scale[c] + shift[c];
  if (relu) v = fmaxf(v, 0.f);
  y[idx] = v;
}

// y = a + b (optionally ReLU); operands may be channel slices
__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, int C, int HW,
                           int a_cs, int a_co, int b_cs, int b_co, int relu, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int p = idx % HW;
  size_t rest = idx / HW;
  const int c = rest % C;
  const size_t n = rest / C;
  float v = a[((size_t)n * a_cs + a_co + c) * HW + p] + b[((size_t)n * b_cs + b_co + c) * HW + p];
  if (relu) v = fmaxf(v, 0.f);
  y[idx] = v;
}

// copy C channels of src (slice) into dst at channel offset (torch.cat / slices)
__global__ void copy_channels_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int HW, int s_cs,
                                     int s_co, int d_cs, int d_co, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int p = idx % HW;
  size_t rest = idx / HW;
  const int c = rest % C;
  const size_t n = rest / C;
  dst[((size_t)n * d_cs + d_co + c) * HW + p] = src[((size_t)n * s_cs + s_co + c) * HW + p];
}

// max / average pooling (count_include_pad semantics of F.avg_pool2d default; no padding used by the reference nets)
__global__ void pool2d_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W, int Ho, int Wo, int k,
                              int stride, int pad, int is_max, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ox = idx % Wo;
  size_t rest = idx / Wo;
  const int oy = rest % Ho;
  const size_t bc = rest / Ho;
  const float *xp = x + bc * (size_t)H * W;
  float v = is_max ? -INFINITY : 0.f;
  for (int ky = 0; ky < k; ky++) {
    const int iy = oy * stride - pad + ky;
    for (int kx = 0; kx < k; kx++) {
      const int ix = ox * stride - pad + kx;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      if (is_max) { if (in) v = fmaxf(v, xp[(size_t)iy * W + ix]); }
      else if (in) v += xp[(size_t)iy * W + ix];
    }
  }
  y[idx] = is_max ? v : v / (float)(k * k);
}

}  // namespace ap

using namespace ap;

extern "C" int ap_conv2d_pack(const float *w, const float *scale, float *wT, int Cout, int Cin_g, int kh, int kw,
                              int groups, void *stream) {
  if (!w || !wT || Cout < 1 || Cin_g < 1 || groups < 1 || Cout % groups) { set_error("ap_conv2d_pack: bad argument"); return -22; }
  const int Kg = Cin_g * kh * kw;
  size_t n = (size_t)Cout * Kg;
  conv_pack_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, scale, wT, Cout, Kg, groups);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_conv2d_fwd(const float *x, const float *wT, const float *bias, const float *res, float *out, int B,
                             int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int groups, int relu,
                             int x_cstride, int x_coff, void *stream) {
  if (!x || !wT || !out || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || kh < 1 || kw < 1 || stride < 1 || pad < 0 ||
      groups < 1 || Cin % groups || Cout % groups || x_cstride < x_coff + Cin) {
    set_error("ap_conv2d_fwd: bad argument");
    return -22;
  }
  ConvArgs a;
  a.x = x; a.wT = wT; a.bias = bias; a.res = res; a.out = out;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
  a.groups = groups; a.relu = relu; a.x_cstride = x_cstride; a.x_coff = x_coff;
  a.Ho = (H + 2 * pad - kh) / stride + 1;
  a.Wo = (W + 2 * pad - kw) / stride + 1;
  if (a.Ho < 1 || a.Wo < 1) { set_error("ap_conv2d_fwd: empty output"); return -22; }
  const long long N = (long long)B * a.Ho * a.Wo;
  const int Mg = Cout / groups;
  dim3 grid((unsigned)((N + CBN - 1) / CBN), (unsigned)((Mg + CBM - 1) / CBM), (unsigned)groups);
  conv2d_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_affine_nchw(const float *x, const float *scale, const float *shift, float *y, int B, int C, int HW,
                              int x_cstride, int x_coff, int relu, void *stream) {
  if (!x || !y || B < 1 || C < 1 || HW < 1 || (scale && !shift)) { set_error("ap_affine_nchw: bad argument"); return -22; }
  size_t total = (size_t)B * C * HW;
  affine_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, scale, shift, y, C, HW, x_cstride, x_coff,
                                                                             relu, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_add_nchw(const float *a, const float *b, float *y, int B, int C, int HW, int a_cstride, int a_coff,
                           int b_cstride, int b_coff, int relu, void *stream) {
  if (!a || !b || !y || B < 1 || C < 1 || HW < 1) { set_error("ap_add_nchw: bad argument"); return -22; }
  size_t total = (size_t)B * C * HW;
  add_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, b, y, C, HW, a_cstride, a_coff, b_cstride,
                                                                          b_coff, relu, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_copy_channels(const float *src, float *dst, int B, int C, int HW, int s_cstride, int s_coff, int d_cstride,
                                int d_coff, void *stream) {
  if (!src || !dst || B < 1 || C < 1 || HW < 1) { set_error("ap_copy_channels: bad argument"); return -22; }
  size_t total = (size_t)B * C * HW;
  copy_channels_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, C, HW, s_cstride, s_coff,
                                                                                    d_cstride, d_coff, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_pool2d(const float *x, float *y, int BC, int H, int W, int k, int stride, int pad, int is_max, void *stream) {
  if (!x || !y || BC < 1 || H < 1 || W < 1 || k < 1 || stride < 1 || pad < 0) { set_error("ap_pool2d: bad argument"); return -22; }
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho < 1 || Wo < 1) { set_error("ap_pool2d: empty output"); return -22; }
  size_t total = (size_t)BC * Ho * Wo;
  pool2d_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, H, W, Ho, Wo, k, stride, pad, is_max,
                                                                             total);
  AP_HIP(hipGetLastError());
  return 0;
}
