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
__global__ void init_conv_bwd_kernel(const float *__restrict__ h0, const float *__restrict__ w0,
                                     const float *__restrict__ dh0, float *__restrict__ dx, int C, int L) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  float s = 0.f;
  for (int c = 0; c < C; c++) {
    const size_t i = ((size_t)b * C + c) * L + t;
    if (h0[i] > 0.f) s = __builtin_fmaf(w0[c], dh0[i], s);
  }
  dx[(size_t)b * L + t] = s;
}

// votes of a batch of score rows into a running histogram (certified_robust.py:58-65: argmax per sample, then a
// per-class count); first maximum wins like Tensor.max
__global__ void argmax_hist_kernel(const float *__restrict__ scores, long long *__restrict__ counts, int B, int K) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *r = scores + (size_t)b * K;
  int am = 0;
  for (int k = 1; k < K; k++)
    if (r[k] > r[am]) am = k;
  atomicAdd(reinterpret_cast<unsigned long long *>(counts + am), 1ull);
}

}  // namespace ap

using namespace ap;

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
  dim3 grid((L + 255) / 256, B);
  init_conv_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(h0, w0, dh0, dx, C, L);
  AP_HIP(hipGetLastError());
  return 0;
}
