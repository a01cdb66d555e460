// Element-wise pieces of the input-gradient of the DiffWave eps-network (the white-box adaptive
// attack back-propagates through the purifier, robustness_eval/white_box_attack.py:392,437-439).  The three GEMM-shaped
// terms of a block's backward -- the recomputed dilated conv, W2^T [dh'; dskip] and the transposed dilated conv --
// run on ap_conv2d_fwd (AP_CONV_1D, dilation); these kernels are what sits between them.
#include "ap_common.h"

namespace ap {

// forward (WaveNet.py:90): g = tanh(a_t) sigmoid(a_s), a = [a_t (C rows); a_s (C rows)]
// backward: da_t = dg sigmoid(a_s) (1 - tanh(a_t)^2);  da_s = dg tanh(a_t) sigmoid(a_s) (1 - sigmoid(a_s))
__global__ void gate_bwd_kernel(const float *__restrict__ a, const float *__restrict__ dg, float *__restrict__ da, int C,
                                int L, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [B][C][L]
  if (idx >= total) return;
  const int t = idx % L;
  size_t rest = idx / L;
  const int c = rest % C;
  const size_t b = rest / C;
  const size_t it = ((size_t)b * 2 * C + c) * L + t, is = it + (size_t)C * L;
  const float at = fminf(fmaxf(a[it], -15.0f), 15.0f), as = a[is];
  const float E = expf(2.0f * at);
  const float th = (E - 1.0f) / (E + 1.0f);
  const float sg = 1.0f / (1.0f + expf(-as));
  const float g = dg[idx];
  da[it] = g * sg * (1.0f - th * th);
  da[is] = g * th * sg * (1.0f - sg);
}

// final_conv backward through the ReLU (WaveNet.py:160-162): dr[b][c][t] = r > 0 ? w2[c] deps[b][t] : 0
__global__ void relu_outer_bwd_kernel(const float *__restrict__ r, const float *__restrict__ w2,
                                      const float *__restrict__ deps, float *__restrict__ dr, int S, int L, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [B][S][L]
  if (idx >= total) return;
  const int t = idx % L;
  size_t rest = idx / L;
  const int c = rest % S;
  const size_t b = rest / S;
  dr[idx] = r[idx] > 0.f ? w2[c] * deps[b * L + t] : 0.f;
}

// init_conv backward (WaveNet.py:147,168): h0 = relu(w0[c] x + b0[c])  ->  dx[b][t] = sum_c [h0 > 0] w0[c] dh0[b][c][t]
// Workgroup = 64 samples x 4 channel quarters; eight channels' loads in flight per thread, the quarters summed through LDS in a fixed
// order (one thread walking all C channels with a dependent load pair per step ran at 0.7 TB/s: 0.23 ms at B = 10).
__global__ __launch_bounds__(256) void init_conv_bwd_kernel(const float *__restrict__ h0, const float *__restrict__ w0,
                                                            const float *__restrict__ dh0, float *__restrict__ dx, int C, int L) {
  __shared__ float part[4][64];
  const int b = blockIdx.y;
  const int tl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const int cq = (C + 3) / 4, c0 = q * cq, c1 = min(C, c0 + cq);
  float s = 0.f;
  if (t < L) {
    const size_t base = (size_t)b * C * L + t;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
      float hv[8], dv[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { hv[i] = h0[base + (size_t)(c + i) * L]; dv[i] = dh0[base + (size_t)(c + i) * L]; }
#pragma unroll
      for (int i = 0; i < 8; i++) s = hv[i] > 0.f ? __builtin_fmaf(w0[c + i], dv[i], s) : s;
    }
    for (; c < c1; c++) {
      const float hv = h0[base + (size_t)c * L], dv = dh0[base + (size_t)c * L];
      s = hv > 0.f ? __builtin_fmaf(w0[c], dv, s) : s;
    }
  }
  part[q][tl] = s;
  __syncthreads();
  if (q == 0 && t < L) dx[(size_t)b * L + t] = (part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]);
}

// ---- pieces of the input gradient of the lowered 2-D classifiers (audiopure_amd/convnet.py backward) ----
// dst[:, d_coff : d_coff + C] += src[:, s_coff : s_coff + C]   (a value feeding several consumers / torch.cat slices)
__global__ void acc_channels_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int HW, int s_cs,
                                    int s_co, int d_cs, int d_co, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int p = idx % HW;
  size_t rest = idx / HW;
  const int c = rest % C;
  const size_t n = rest / C;
  dst[((size_t)n * d_cs + d_co + c) * HW + p] += src[((size_t)n * s_cs + s_co + c) * HW + p];
}

// out = y > 0 ? dy : 0   (backward through a fused ReLU, y = the forward output)
__global__ void relu_mask_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// transposed-conv helper for stride > 1: out [BC][Hz][Wz] = dy [BC][Ho][Wo] spread on a stride-s grid, zeros elsewhere
__global__ void zero_insert_kernel(const float *__restrict__ dy, float *__restrict__ out, int Ho, int Wo, int Hz, int Wz,
                                   int s, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over out
  if (idx >= total) return;
  const int x = idx % Wz;
  size_t rest = idx / Wz;
  const int y = rest % Hz;
  const size_t bc = rest / Hz;
  float v = 0.f;
  if (y % s == 0 && x % s == 0 && y / s < Ho && x / s < Wo) v = dy[(bc * Ho + y / s) * Wo + x / s];
  out[idx] = v;
}

// nn.MaxPool2d / F.avg_pool2d backward, gathered per input element (first maximum of a window wins, like torch)
__global__ void pool2d_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int H,
                                  int W, int Ho, int Wo, int k, int stride, int pad, int is_max, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over dx [BC][H][W]
  if (idx >= total) return;
  const int ix = idx % W;
  size_t rest = idx / W;
  const int iy = rest % H;
  const size_t bc = rest / H;
  const float *xp = x + bc * (size_t)H * W;
  float s = 0.f;
  // windows (oy, ox) with oy*stride - pad <= iy < oy*stride - pad + k
  int oy0 = (iy + pad - k + stride) / stride;
  if (iy + pad - k + 1 <= 0) oy0 = 0;
  int ox0 = (ix + pad - k + stride) / stride;
  if (ix + pad - k + 1 <= 0) ox0 = 0;
  const int oy1 = min((iy + pad) / stride, Ho - 1), ox1 = min((ix + pad) / stride, Wo - 1);
  for (int oy = oy0; oy <= oy1; oy++)
    for (int ox = ox0; ox <= ox1; ox++) {
      const float g = dy[(bc * Ho + oy) * Wo + ox];
      if (!is_max) { s += g / (float)(k * k); continue; }
      // is (iy, ix) the first maximum of this window?
      float best = -INFINITY;
      int by = -1, bx = -1;
      for (int ky = 0; ky < k; ky++) {
        const int yy = oy * stride - pad + ky;
        if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < k; kx++) {
          const int xx = ox * stride - pad + kx;
          if (xx < 0 || xx >= W) continue;
          const float v = xp[(size_t)yy * W + xx];
          if (v > best) { best = v; by = yy; bx = xx; }
        }
      }
      if (by == iy && bx == ix) s += g;
    }
  dx[idx] = s;
}

// votes of a batch of score rows into a running histogram (certified_robust.py:58-65: argmax per sample, then a
// per-class count); first maximum wins like Tensor.max.  A row with a NaN or an infinity is not a vote: it is counted in
// slot K so the caller can refuse the batch instead of certifying on a numerically broken classifier / denoiser.
__global__ void argmax_hist_kernel(const float *__restrict__ scores, long long *__restrict__ counts, int B, int K) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *r = scores + (size_t)b * K;
  int am = 0;
  bool finite = true;
  for (int k = 0; k < K; k++) {
    finite = finite && (fabsf(r[k]) <= 3.402823466e38f);       // false for NaN and +-inf
    if (r[k] > r[am]) am = k;
  }
  atomicAdd(reinterpret_cast<unsigned long long *>(counts + (finite ? am : K)), 1ull);
}

}  // namespace ap

using namespace ap;

extern "C" int ap_acc_channels(const float *src, float *dst, int B, int C, int HW, int s_cstride, int s_coff, int d_cstride,
                               int d_coff, void *stream) {
  if (!src || !dst || B < 1 || C < 1 || HW < 1 || s_cstride < s_coff + C || d_cstride < d_coff + C) { set_error("ap_acc_channels: bad argument"); return -22; }
  const size_t total = (size_t)B * C * HW;
  acc_channels_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, C, HW, s_cstride, s_coff,
                                                                                       d_cstride, d_coff, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_relu_mask(const float *dy, const float *y, float *out, size_t n, void *stream) {
  if (!dy || !y || !out || n < 1) { set_error("ap_relu_mask: bad argument"); return -22; }
  relu_mask_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, y, out, n);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_zero_insert2d(const float *dy, float *out, int BC, int Ho, int Wo, int Hz, int Wz, int stride, void *stream) {
  if (!dy || !out || BC < 1 || Ho < 1 || Wo < 1 || stride < 1 || Hz < (Ho - 1) * stride + 1 || Wz < (Wo - 1) * stride + 1) { set_error("ap_zero_insert2d: bad argument"); return -22; }
  const size_t total = (size_t)BC * Hz * Wz;
  zero_insert_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, out, Ho, Wo, Hz, Wz, stride, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_pool2d_bwd(const float *x, const float *dy, float *dx, int BC, int H, int W, int k, int stride, int pad,
                             int is_max, void *stream) {
  if (!x || !dy || !dx || BC < 1 || H < 1 || W < 1 || k < 1 || stride < 1 || pad < 0) { set_error("ap_pool2d_bwd: bad argument"); return -22; }
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho < 1 || Wo < 1) { set_error("ap_pool2d_bwd: empty output"); return -22; }
  const size_t total = (size_t)BC * H * W;
  pool2d_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, dy, dx, H, W, Ho, Wo, k, stride, pad,
                                                                                     is_max, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_argmax_hist(const float *scores, long long *counts, int B, int K, void *stream) {
  if (!scores || !counts || B < 1 || K < 1) { set_error("ap_argmax_hist: bad argument"); return -22; }
  argmax_hist_kernel<<<(B + 255) / 256, 256, 0, (hipStream_t)stream>>>(scores, counts, B, K);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_gate_bwd(const float *a, const float *dg, float *da, int B, int C, int L, void *stream) {
  if (!a || !dg || !da || B < 1 || C < 1 || L < 1) { set_error("ap_gate_bwd: bad argument"); return -22; }
  const size_t total = (size_t)B * C * L;
  gate_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, dg, da, C, L, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_relu_outer_bwd(const float *r, const float *w2, const float *deps, float *dr, int B, int S, int L,
                                 void *stream) {
  if (!r || !w2 || !deps || !dr || B < 1 || S < 1 || L < 1) { set_error("ap_relu_outer_bwd: bad argument"); return -22; }
  const size_t total = (size_t)B * S * L;
  relu_outer_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(r, w2, deps, dr, S, L, total);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_init_conv_bwd(const float *h0, const float *w0, const float *dh0, float *dx, int B, int C, int L,
                                void *stream) {
  if (!h0 || !w0 || !dh0 || !dx || B < 1 || C < 1 || L < 1) { set_error("ap_init_conv_bwd: bad argument"); return -22; }
  dim3 grid((L + 63) / 64, B);
  init_conv_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(h0, w0, dh0, dx, C, L);
  AP_HIP(hipGetLastError());
  return 0;
}
