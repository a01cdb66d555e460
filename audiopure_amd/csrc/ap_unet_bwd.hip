// Input-gradient pieces of the Improved-Diffusion UNet (SURVEY section 8 f-1 for the DiffSpec defense: the white-box
// attack back-propagates through RevImprovedDiffusion, adaptive_attack_eval.py:102-104 + white_box_attack.py:437-439).
// The convolutions' input gradients run on ap_conv2d_fwd with flipped / transposed weights (ap_zero_insert2d for the
// stride-2 Downsample); these kernels are GroupNorm32 (+ scale-shift + SiLU) and QKVAttention backward.  Parameters are
// frozen and the timestep embedding does not depend on the input, so only d/dx is formed.
#include "ap_common.h"

namespace ap {

__device__ __forceinline__ float block_sum256(float v, float *red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// forward (groupnorm_kernel): xh = (x - mean) rstd; y0 = gamma xh + beta; y1 = y0 (1 + sc) + sh; y = act(y1).
// backward: dxh = dy act'(y1) (1 + sc) gamma;  dx = rstd (dxh - mean(dxh) - xh mean(dxh xh))  over the group.
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, const float *__restrict__ ss,
                                                            const float *__restrict__ dy, float *__restrict__ dx, int C,
                                                            int HW, int groups, float eps, int act) {
  __shared__ float red[256];
  const int b = blockIdx.x / groups, g = blockIdx.x % groups, cpg = C / groups, n = cpg * HW;
  const size_t off = ((size_t)b * C + (size_t)g * cpg) * HW;
  const float *xp = x + off, *dyp = dy + off;
  float *dxp = dx + off;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += xp[i];
  const float mean = block_sum256(s, red) / (float)n;
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float dlt = xp[i] - mean;
    v = __builtin_fmaf(dlt, dlt, v);
  }
  const float rstd = 1.0f / sqrtf(block_sum256(v, red) / (float)n + eps);
  auto dxh_of = [&](int i, float &xh) {
    const int c = g * cpg + i / HW;
    xh = (xp[i] - mean) * rstd;
    float y1 = xh * gamma[c] + beta[c], k = gamma[c];
    if (ss) {
      const float sc = 1.0f + ss[(size_t)b * 2 * C + c];
      y1 = y1 * sc + ss[(size_t)b * 2 * C + C + c];
      k *= sc;
    }
    float d = dyp[i];
    if (act == 2) {
      const float sg = 1.0f / (1.0f + expf(-y1));
      d *= sg * (1.0f + y1 * (1.0f - sg));
    } else if (act == 1) {
      d = y1 > 0.f ? d : 0.f;
    }
    return d * k;
  };
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float xh;
    const float d = dxh_of(i, xh);
    s1 += d;
    s2 = __builtin_fmaf(d, xh, s2);
  }
  const float m1 = block_sum256(s1, red) / (float)n;
  const float m2 = block_sum256(s2, red) / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    float xh;
    const float d = dxh_of(i, xh);
    dxp[i] = rstd * (d - m1 - xh * m2);
  }
}

// QKVAttention backward (unet.py:239-252), layout qkv [B][heads][3 ch][T], out/dout [B][heads][ch][T].
// With W = softmax_s(scale2 q_t . k_s) and D_t = sum_c dout[c][t] out[c][t] (= sum_s W dW):
//   dS[t][s] = W[t][s] (dout_t . v_s - D_t);  dq_t = scale2 sum_s dS k_s;  dk_s = scale2 sum_t dS q_t;  dv_s = sum_t W dout_t.
// Pass Q (one query per thread, K and V of the head in LDS): the row's softmax statistics (max, 1/sum), D_t and dq.
// Pass KV (one key per thread, Q and dout of the head in LDS): dk and dv, using the saved row statistics.
// Every output element is written by exactly one thread: no atomics.
template <int CH>
__global__ __launch_bounds__(256) void attention_bwd_q_kernel(const float *__restrict__ qkv, const float *__restrict__ out,
                                                              const float *__restrict__ dout, float *__restrict__ dqkv,
                                                              float *__restrict__ stats, int T, float scale2) {
  extern __shared__ float sm[];
  float *ks = sm, *vs = sm + (size_t)CH * T;
  const int bh = blockIdx.x;
  const float *base = qkv + (size_t)bh * 3 * CH * T;
  for (int i = threadIdx.x; i < CH * T; i += 256) {
    ks[i] = base[(size_t)CH * T + i];
    vs[i] = base[(size_t)2 * CH * T + i];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    float q[CH], g[CH], dq[CH];
    float D = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      q[c] = base[(size_t)c * T + t] * scale2;
      g[c] = dout[((size_t)bh * CH + c) * T + t];
      D = __builtin_fmaf(g[c], out[((size_t)bh * CH + c) * T + t], D);
      dq[c] = 0.f;
    }
    float mx = -INFINITY;
    for (int s = 0; s < T; s++) {
      float w = 0.f;
#pragma unroll
      for (int c = 0; c < CH; c++) w = __builtin_fmaf(q[c], ks[c * T + s], w);
      mx = fmaxf(mx, w);
    }
    float l = 0.f;
    for (int s = 0; s < T; s++) {
      float w = 0.f;
#pragma unroll
      for (int c = 0; c < CH; c++) w = __builtin_fmaf(q[c], ks[c * T + s], w);
      l += expf(w - mx);
    }
    const float inv = 1.0f / l;
    for (int s = 0; s < T; s++) {
      float w = 0.f, dw = 0.f;
#pragma unroll
      for (int c = 0; c < CH; c++) {
        w = __builtin_fmaf(q[c], ks[c * T + s], w);
        dw = __builtin_fmaf(g[c], vs[c * T + s], dw);
      }
      const float ds = expf(w - mx) * inv * (dw - D);
#pragma unroll
      for (int c = 0; c < CH; c++) dq[c] = __builtin_fmaf(ds, ks[c * T + s], dq[c]);
    }
    float *dqp = dqkv + (size_t)bh * 3 * CH * T;
#pragma unroll
    for (int c = 0; c < CH; c++) dqp[(size_t)c * T + t] = dq[c] * scale2;
    float *st = stats + ((size_t)bh * T + t) * 3;
    st[0] = mx;
    st[1] = inv;
    st[2] = D;
  }
}

template <int CH>
__global__ __launch_bounds__(256) void attention_bwd_kv_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                               float *__restrict__ dqkv, const float *__restrict__ stats,
                                                               int T, float scale2) {
  extern __shared__ float sm[];
  float *qs = sm, *gs = sm + (size_t)CH * T;                      // Q (pre-scaled by scale2) and dout of the head
  const int bh = blockIdx.x;
  const float *base = qkv + (size_t)bh * 3 * CH * T;
  for (int i = threadIdx.x; i < CH * T; i += 256) {
    qs[i] = base[i] * scale2;
    gs[i] = dout[(size_t)bh * CH * T + i];
  }
  __syncthreads();
  const float *stb = stats + (size_t)bh * T * 3;
  for (int s = threadIdx.x; s < T; s += 256) {
    float k[CH], v[CH], dk[CH], dv[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
      k[c] = base[(size_t)(CH + c) * T + s];
      v[c] = base[(size_t)(2 * CH + c) * T + s];
      dk[c] = 0.f;
      dv[c] = 0.f;
    }
    for (int t = 0; t < T; t++) {
      float w = 0.f, dw = 0.f;
#pragma unroll
      for (int c = 0; c < CH; c++) {
        w = __builtin_fmaf(qs[c * T + t], k[c], w);
        dw = __builtin_fmaf(gs[c * T + t], v[c], dw);
      }
      const float p = expf(w - stb[t * 3]) * stb[t * 3 + 1];
      const float ds = p * (dw - stb[t * 3 + 2]);
#pragma unroll
      for (int c = 0; c < CH; c++) {
        dv[c] = __builtin_fmaf(p, gs[c * T + t], dv[c]);
        dk[c] = __builtin_fmaf(ds, qs[c * T + t], dk[c]);          // qs carries scale2 already
      }
    }
    float *dp = dqkv + (size_t)bh * 3 * CH * T;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      dp[(size_t)(CH + c) * T + s] = dk[c];
      dp[(size_t)(2 * CH + c) * T + s] = dv[c];
    }
  }
}

// backward of upsample2x_kernel: dx[y][x] = sum of the 2x2 block of dy it was copied to
__global__ void upsample2x_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int H, int W, size_t total) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int xx = idx % W;
  size_t rest = idx / W;
  const int yy = rest % H;
  const size_t bc = rest / H;
  const float *p = dy + (bc * 2 * H + 2 * yy) * 2 * W + 2 * xx;
  dx[idx] = (p[0] + p[1]) + (p[2 * W] + p[2 * W + 1]);
}

}  // namespace ap

using namespace ap;

extern "C" int ap_groupnorm_bwd(const float *x, const float *gamma, const float *beta, const float *scale_shift,
                                const float *dy, float *dx, int B, int C, int HW, int groups, float eps, int act,
                                void *stream) {
  if (!x || !gamma || !beta || !dy || !dx || B < 1 || C < 1 || HW < 1 || groups < 1 || C % groups) {
    set_error("ap_groupnorm_bwd: bad argument");
    return -22;
  }
  groupnorm_bwd_kernel<<<(unsigned)(B * groups), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, scale_shift, dy, dx, C, HW, groups,
                                                                               eps, act);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_attention_qkv_bwd(const float *qkv, const float *out, const float *dout, float *dqkv, float *stats, int B,
                                    int C, int T, int heads, void *stream) {
  if (!qkv || !out || !dout || !dqkv || !stats || B < 1 || C < 1 || T < 1 || heads < 1 || C % heads) {
    set_error("ap_attention_qkv_bwd: bad argument");
    return -22;
  }
  const int ch = C / heads;
  const size_t smem = (size_t)2 * ch * T * sizeof(float);
  if (smem > 160 * 1024) { set_error("ap_attention_qkv_bwd: two [ch][T] images of one head (%zu bytes) exceed the LDS", smem); return -22; }
  const float scale2 = 1.0f / sqrtf((float)ch);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(B * heads);
#define AP_ATTB(CHV)                                                                                                       \
  do {                                                                                                                     \
    static bool attr = false;                                                                                              \
    if (!attr) {                                                                                                           \
      AP_HIP(hipFuncSetAttribute((const void *)attention_bwd_q_kernel<CHV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
      AP_HIP(hipFuncSetAttribute((const void *)attention_bwd_kv_kernel<CHV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr = true;                                                                                                         \
    }                                                                                                                      \
    attention_bwd_q_kernel<CHV><<<grid, 256, smem, st>>>(qkv, out, dout, dqkv, stats, T, scale2);                           \
    attention_bwd_kv_kernel<CHV><<<grid, 256, smem, st>>>(qkv, dout, dqkv, stats, T, scale2);                               \
  } while (0)
  switch (ch) {
    case 8: AP_ATTB(8); break;
    case 16: AP_ATTB(16); break;
    case 32: AP_ATTB(32); break;
    case 64: AP_ATTB(64); break;
    default: set_error("ap_attention_qkv_bwd: channels per head %d not built (8, 16, 32, 64)", ch); return -22;
  }
#undef AP_ATTB
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_upsample_nearest2x_bwd(const float *dy, float *dx, int BC, int H, int W, void *stream) {
  if (!dy || !dx || BC < 1 || H < 1 || W < 1) { set_error("ap_upsample_nearest2x_bwd: bad argument"); return -22; }
  const size_t total = (size_t)BC * H * W;
  upsample2x_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, dx, H, W, total);
  AP_HIP(hipGetLastError());
  return 0;
}
