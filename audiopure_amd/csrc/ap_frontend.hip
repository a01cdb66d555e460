// Classifier front-end kernels: M5 raw-waveform classifier (audio_models/M5/M5Net.py:4-38).
// One workgroup per utterance; every intermediate stays in LDS (64 KB clip + <45 KB activations).
#include "ap_common.h"

namespace ap {

// BatchNorm(eval) folded into the conv: w' = w * s, b' = (b - mean) * s + beta, s = gamma / sqrt(var + eps).
// Output weight layout is transposed to [ci][k][co] so that threads (co fastest) read coalesced.
__global__ void m5_fold_kernel(const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ gamma,
                               const float *__restrict__ beta, const float *__restrict__ mean,
                               const float *__restrict__ var, float eps, float *__restrict__ wT,
                               float *__restrict__ bo, int co, int ci, int k) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int n = co * ci * k;
  if (idx < n) {
    int o = idx / (ci * k), r = idx % (ci * k);
    float s = gamma[o] / sqrtf(var[o] + eps);
    wT[(size_t)r * co + o] = w[idx] * s;
  }
  if (idx < co) {
    float s = gamma[idx] / sqrtf(var[idx] + eps);
    bo[idx] = (b[idx] - mean[idx]) * s + beta[idx];
  }
}

int launch_m5_fold(ap_m5 *m, const float *blob, float bn_eps, hipStream_t st) {
  const int nc = m->n_channel;
  int ci[4] = {1, nc, nc, 2 * nc}, co[4] = {nc, nc, 2 * nc, 2 * nc}, k[4] = {m->k1, 3, 3, 3};
  size_t o = 0;
  for (int i = 0; i < 4; i++) {
    size_t nw = (size_t)co[i] * ci[i] * k[i];
    const float *w = blob + o, *b = w + nw, *g = b + co[i], *be = g + co[i], *mu = be + co[i], *var = mu + co[i];
    o += nw + 5 * (size_t)co[i];
    m5_fold_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(w, b, g, be, mu, var, bn_eps, m->w[i], m->b[i], co[i],
                                                               ci[i], k[i]);
  }
  size_t nf = (size_t)m->n_output * 2 * nc;
  AP_HIP(hipMemcpyAsync(m->fcw, blob + o, nf * sizeof(float), hipMemcpyDeviceToDevice, st));
  AP_HIP(hipMemcpyAsync(m->fcb, blob + o + nf, m->n_output * sizeof(float), hipMemcpyDeviceToDevice, st));
  AP_HIP(hipGetLastError());
  return 0;
}

// conv(k taps, stride) + folded BN + ReLU + MaxPool(4) from `in` [ci][Lin] (LDS or global) to `out` [co][Q] (LDS)
template <bool FIRST>
__device__ __forceinline__ void m5_stage(const float *in, int Lin, int ci, const float *__restrict__ wT,
                                         const float *__restrict__ bias, int co, int k, int stride, float *out, int Q) {
  const int nth = blockDim.x;
  for (int idx = threadIdx.x; idx < co * Q; idx += nth) {
    const int o = idx % co, q = idx / co;       // co fastest: a wave shares q -> LDS broadcast reads of `in`
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int p0 = 4 * q * stride;
    for (int c = 0; c < ci; c++) {
      const float *row = in + (size_t)c * Lin + p0;
      const float *wr = wT + (size_t)c * k * co + o;
      for (int t = 0; t < k; t++) {
        const float w = wr[(size_t)t * co];
        a0 = __builtin_fmaf(w, row[t], a0);
        a1 = __builtin_fmaf(w, row[t + stride], a1);
        a2 = __builtin_fmaf(w, row[t + 2 * stride], a2);
        a3 = __builtin_fmaf(w, row[t + 3 * stride], a3);
      }
    }
    float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) + bias[o];   // max commutes with the per-channel bias add and ReLU
    out[(size_t)o * Q + q] = fmaxf(mx, 0.f);
  }
}

__global__ __launch_bounds__(256) void m5_kernel(const float *__restrict__ x, float *__restrict__ logprobs,
                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                 const float *__restrict__ w3, const float *__restrict__ b3,
                                                 const float *__restrict__ w4, const float *__restrict__ b4,
                                                 const float *__restrict__ fcw, const float *__restrict__ fcb, int L,
                                                 int nc, int k1, int stride, int n_out, int Q1, int Q2, int Q3, int Q4,
                                                 int stage_x) {
  extern __shared__ float sm[];
  const int b = blockIdx.x;
  const float *xb = x + (size_t)b * L;
  float *o1 = sm, *o2 = o1 + nc * Q1, *o3 = o2 + nc * Q2, *o4 = o3 + 2 * nc * Q3, *feat = o4 + 2 * nc * Q4,
        *logit = feat + 2 * nc, *xs = logit + 64;
  const float *xin = xb;
  if (stage_x) {
    for (int i = threadIdx.x; i < L; i += blockDim.x) xs[i] = xb[i];
    xin = xs;
    __syncthreads();
  }
  m5_stage<true>(xin, L, 1, w1, b1, nc, k1, stride, o1, Q1);
  __syncthreads();
  m5_stage<false>(o1, Q1, nc, w2, b2, nc, 3, 1, o2, Q2);
  __syncthreads();
  m5_stage<false>(o2, Q2, nc, w3, b3, 2 * nc, 3, 1, o3, Q3);
  __syncthreads();
  m5_stage<false>(o3, Q3, 2 * nc, w4, b4, 2 * nc, 3, 1, o4, Q4);
  __syncthreads();
  // F.avg_pool1d(x, x.shape[-1]) (M5Net.py:34)
  for (int c = threadIdx.x; c < 2 * nc; c += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < Q4; q++) s += o4[c * Q4 + q];
    feat[c] = s / (float)Q4;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < 2 * nc; c++) s = __builtin_fmaf(fcw[o * 2 * nc + c], feat[c], s);
    logit[o] = s + fcb[o];
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // F.log_softmax (M5Net.py:38)
    float mx = logit[0];
    for (int o = 1; o < n_out; o++) mx = fmaxf(mx, logit[o]);
    float se = 0.f;
    for (int o = 0; o < n_out; o++) se += expf(logit[o] - mx);
    float lse = mx + logf(se);
    for (int o = 0; o < n_out; o++) logprobs[(size_t)b * n_out + o] = logit[o] - lse;
  }
}

int launch_m5(ap_m5 *m, const float *x, float *logprobs, int B, int L, hipStream_t st) {
  const int nc = m->n_channel;
  if (L < m->k1) { set_error("m5: clip length %d shorter than the first kernel %d", L, m->k1); return -22; }
  const int P1 = (L - m->k1) / m->stride + 1, Q1 = P1 / 4;
  const int Q2 = (Q1 - 2) / 4, Q3 = (Q2 - 2) / 4, Q4 = (Q3 - 2) / 4;
  if (Q1 < 3 || Q2 < 3 || Q3 < 3 || Q4 < 1) { set_error("m5: clip length %d too short for four conv/pool stages", L); return -22; }
  size_t act = (size_t)nc * Q1 + (size_t)nc * Q2 + (size_t)2 * nc * Q3 + (size_t)2 * nc * Q4 + 2 * nc + 64;
  int stage_x = ((act + L) * sizeof(float) <= 150 * 1024);
  size_t smem = (act + (stage_x ? L : 0)) * sizeof(float);
  if (smem > 160 * 1024) { set_error("m5: clip length %d needs %zu bytes of LDS", L, smem); return -22; }
  static bool attr_set = false;
  if (!attr_set) {
    AP_HIP(hipFuncSetAttribute((const void *)m5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  m5_kernel<<<B, 256, smem, st>>>(x, logprobs, m->w[0], m->b[0], m->w[1], m->b[1], m->w[2], m->b[2], m->w[3], m->b[3],
                                 m->fcw, m->fcb, L, nc, m->k1, m->stride, m->n_output, Q1, Q2, Q3, Q4, stage_x);
  AP_HIP(hipGetLastError());
  return 0;
}

// ---- dL/dx through M5 (white-box attack, white_box_attack.py:392,437-439: loss = CE(classifier(purifier(x)))) ----
// One workgroup per clip: recompute the four stages keeping, per pooled element, its value and which of the 4 window
// positions won the max; then fc -> avg-pool -> stage 4..1 backward, each input gradient GATHERED over the outputs
// that read it (no atomics).  Parameters are frozen (eval): only the input gradient is formed.
template <bool FIRST>
__device__ __forceinline__ void m5_stage_save(const float *in, int Lin, int ci, const float *__restrict__ wT,
                                              const float *__restrict__ bias, int co, int k, int stride, float *out,
                                              unsigned char *arg, int Q) {
  for (int idx = threadIdx.x; idx < co * Q; idx += blockDim.x) {
    const int o = idx % co, q = idx / co;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const int p0 = 4 * q * stride;
    for (int c = 0; c < ci; c++) {
      const float *row = in + (size_t)c * Lin + p0;
      const float *wr = wT + (size_t)c * k * co + o;
      for (int t = 0; t < k; t++) {
        const float w = wr[(size_t)t * co];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = __builtin_fmaf(w, row[t + i * stride], a[i]);
      }
    }
    int am = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
      if (a[i] > a[am]) am = i;                                  // first maximum wins, like nn.MaxPool1d
    out[(size_t)o * Q + q] = fmaxf(a[am] + bias[o], 0.f);
    arg[(size_t)o * Q + q] = (unsigned char)am;
  }
}

// din[c][j] = sum over outputs (o, pre-pool position p) that read input j and won their pooling window
__device__ __forceinline__ void m5_stage_bwd(const float *dout, const float *pout, const unsigned char *arg, int Q, int co,
                                             const float *__restrict__ wT, int ci, int k, int stride, float *din, int Lin) {
  for (int idx = threadIdx.x; idx < ci * Lin; idx += blockDim.x) {
    const int j = idx % Lin, c = idx / Lin;
    int plo = (j - k + 1 + stride - 1) / stride;
    if (j - k + 1 <= 0) plo = 0;
    int phi = j / stride;
    if (phi > 4 * Q - 1) phi = 4 * Q - 1;                        // positions past 4Q are dropped by the floor pooling
    float s = 0.f;
    for (int p = plo; p <= phi; p++) {
      const int q = p >> 2, sub = p & 3, t = j - p * stride;
      const float *wr = wT + ((size_t)c * k + t) * co;
      for (int o = 0; o < co; o++) {
        const size_t e = (size_t)o * Q + q;
        if (arg[e] == sub && pout[e] > 0.f) s = __builtin_fmaf(wr[o], dout[e], s);
      }
    }
    din[idx] = s;
  }
}

// (1 024 threads: every loop below strides by blockDim.x and each output is one thread's own fma chain, so the thread count changes
// the time -- the stages are latency-bound gathers -- and not a bit of the result)
__global__ __launch_bounds__(1024) void m5_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dlogp,
                                                     float *__restrict__ dx, const float *__restrict__ w1,
                                                     const float *__restrict__ b1, const float *__restrict__ w2,
                                                     const float *__restrict__ b2, const float *__restrict__ w3,
                                                     const float *__restrict__ b3, const float *__restrict__ w4,
                                                     const float *__restrict__ b4, const float *__restrict__ fcw,
                                                     const float *__restrict__ fcb, int L, int nc, int k1, int stride,
                                                     int n_out, int Q1, int Q2, int Q3, int Q4) {
  extern __shared__ float sm[];
  const int b = blockIdx.x;
  const float *xb = x + (size_t)b * L;
  const int n1 = nc * Q1, n2 = nc * Q2, n3 = 2 * nc * Q3, n4 = 2 * nc * Q4;
  float *o1 = sm, *o2 = o1 + n1, *o3 = o2 + n2, *o4 = o3 + n3;          // pooled activations
  float *g1 = o4 + n4, *g2 = g1 + n1, *g3 = g2 + n2, *g4 = g3 + n3;     // their gradients
  float *feat = g4 + n4, *logit = feat + 2 * nc, *dlogit = logit + 64;
  unsigned char *a1 = reinterpret_cast<unsigned char *>(dlogit + 64), *a2 = a1 + n1, *a3 = a2 + n2, *a4 = a3 + n3;
  m5_stage_save<true>(xb, L, 1, w1, b1, nc, k1, stride, o1, a1, Q1);
  __syncthreads();
  m5_stage_save<false>(o1, Q1, nc, w2, b2, nc, 3, 1, o2, a2, Q2);
  __syncthreads();
  m5_stage_save<false>(o2, Q2, nc, w3, b3, 2 * nc, 3, 1, o3, a3, Q3);
  __syncthreads();
  m5_stage_save<false>(o3, Q3, 2 * nc, w4, b4, 2 * nc, 3, 1, o4, a4, Q4);
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * nc; c += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < Q4; q++) s += o4[c * Q4 + q];
    feat[c] = s / (float)Q4;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < 2 * nc; c++) s = __builtin_fmaf(fcw[o * 2 * nc + c], feat[c], s);
    logit[o] = s + fcb[o];
  }
  __syncthreads();
  if (threadIdx.x == 0) {          // log_softmax backward: dlogit = dlogp - softmax * sum(dlogp)
    float mx = logit[0];
    for (int o = 1; o < n_out; o++) mx = fmaxf(mx, logit[o]);
    float se = 0.f, sd = 0.f;
    for (int o = 0; o < n_out; o++) { se += expf(logit[o] - mx); sd += dlogp[(size_t)b * n_out + o]; }
    for (int o = 0; o < n_out; o++) dlogit[o] = dlogp[(size_t)b * n_out + o] - expf(logit[o] - mx) / se * sd;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * nc; c += blockDim.x) {   // fc and the average pool
    float s = 0.f;
    for (int o = 0; o < n_out; o++) s = __builtin_fmaf(fcw[o * 2 * nc + c], dlogit[o], s);
    for (int q = 0; q < Q4; q++) g4[c * Q4 + q] = s / (float)Q4;
  }
  __syncthreads();
  m5_stage_bwd(g4, o4, a4, Q4, 2 * nc, w4, 2 * nc, 3, 1, g3, Q3);
  __syncthreads();
  m5_stage_bwd(g3, o3, a3, Q3, 2 * nc, w3, nc, 3, 1, g2, Q2);
  __syncthreads();
  m5_stage_bwd(g2, o2, a2, Q2, nc, w2, nc, 3, 1, g1, Q1);
  __syncthreads();
  m5_stage_bwd(g1, o1, a1, Q1, nc, w1, 1, k1, stride, dx + (size_t)b * L, L);
}

int launch_m5_bwd(ap_m5 *m, const float *x, const float *dlogp, float *dx, int B, int L, hipStream_t st) {
  const int nc = m->n_channel;
  if (L < m->k1) { set_error("m5: clip length %d shorter than the first kernel %d", L, m->k1); return -22; }
  const int P1 = (L - m->k1) / m->stride + 1, Q1 = P1 / 4;
  const int Q2 = (Q1 - 2) / 4, Q3 = (Q2 - 2) / 4, Q4 = (Q3 - 2) / 4;
  if (Q1 < 3 || Q2 < 3 || Q3 < 3 || Q4 < 1) { set_error("m5: clip length %d too short for four conv/pool stages", L); return -22; }
  const size_t act = (size_t)nc * Q1 + (size_t)nc * Q2 + (size_t)2 * nc * Q3 + (size_t)2 * nc * Q4;
  const size_t smem = (2 * act + 2 * nc + 128) * sizeof(float) + ((act + 15) & ~(size_t)15);
  if (smem > 160 * 1024) { set_error("m5 backward: clip length %d needs %zu bytes of LDS", L, smem); return -22; }
  static bool attr_set = false;
  if (!attr_set) {
    AP_HIP(hipFuncSetAttribute((const void *)m5_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  m5_bwd_kernel<<<B, 1024, smem, st>>>(x, dlogp, dx, m->w[0], m->b[0], m->w[1], m->b[1], m->w[2], m->b[2], m->w[3], m->b[3],
                                     m->fcw, m->fcb, L, nc, m->k1, m->stride, m->n_output, Q1, Q2, Q3, Q4);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap

// mel front-end is declared in the header and built in a later file of this round.
