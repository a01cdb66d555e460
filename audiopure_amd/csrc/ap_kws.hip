// Keyword-spotting route: KWSModel.forward (audio_models/RCNN_KWS/model.py:66-114) and the
// default-parameter torchaudio mel front-end its script builds (kws_adaptive_attack_eval.py:65-67), for clips of any
// length.  Both are tiny next to the purifier (a 5-step GRU over 64 units for a 1 s clip): one workgroup per clip /
// per frame, everything in LDS, plain fp32 VALU.
#include "ap_common.h"

#include <math.h>

struct ap_kws {
  int n_mels, hidden, num_classes, groups;
  float *blob;          // device copy of the state dict, state-dict order (ap_kws_blob_elems)
  size_t n;
};

namespace ap {

struct KwsOff {          // offsets (floats) into the blob
  size_t dw_w, dw_b, pw_w, pw_b, gru[2][2][4], wx_w, wx_b, vt, u, total;
};

static KwsOff kws_layout(int n_mels, int H, int K) {
  KwsOff o;
  size_t p = 0;
  const int groups = n_mels / 20 > 0 ? n_mels / 20 : 1;
  o.dw_w = p; p += (size_t)n_mels * 5;
  o.dw_b = p; p += n_mels;
  o.pw_w = p; p += (size_t)H * (n_mels / groups);
  o.pw_b = p; p += H;
  for (int l = 0; l < 2; l++)
    for (int d = 0; d < 2; d++) {
      const int in = l == 0 ? H : 2 * H;
      o.gru[l][d][0] = p; p += (size_t)3 * H * in;    // weight_ih
      o.gru[l][d][1] = p; p += (size_t)3 * H * H;     // weight_hh
      o.gru[l][d][2] = p; p += 3 * H;                 // bias_ih
      o.gru[l][d][3] = p; p += 3 * H;                 // bias_hh
    }
  o.wx_w = p; p += (size_t)4 * H * H;
  o.wx_b = p; p += 2 * H;
  o.vt = p; p += 2 * H;
  o.u = p; p += (size_t)K * 2 * H;
  o.total = p;
  return o;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// x [B][n_mels][T] mel-dB -> log-probabilities [B][K]
__global__ __launch_bounds__(256) void kws_kernel(const float *__restrict__ x, float *__restrict__ logp,
                                                  const float *__restrict__ w, KwsOff o, int n_mels, int H, int K,
                                                  int groups, int T, int T1, int T2) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  float *dw = sm;                                   // [n_mels][T1]
  float *seq0 = dw + (size_t)n_mels * T1;           // [T2][2H]  (layer input; first layer uses the first H columns)
  float *seq1 = seq0 + (size_t)T2 * 2 * H;          // [T2][2H]  layer output
  float *gi = seq1 + (size_t)T2 * 2 * H;            // [2 dirs][T2][3H]
  float *gh = gi + (size_t)2 * T2 * 3 * H;          // [2 dirs][3H]
  float *hcur = gh + 2 * 3 * H;                     // [2 dirs][H]
  float *e = hcur + 2 * H;                          // [T2]
  float *cvec = e + T2;                             // [2H]
  float *logit = cvec + 2 * H;                      // [K]
  const float *xb = x + (size_t)b * n_mels * T;
  // sepconv.0: depthwise Conv1d(k = 5, stride 2)   (model.py:7-9)
  for (int i = tid; i < n_mels * T1; i += nth) {
    const int c = i / T1, t = i - c * T1;
    float s = w[o.dw_b + c];
    for (int k = 0; k < 5; k++) s = __builtin_fmaf(w[o.dw_w + c * 5 + k], xb[(size_t)c * T + 2 * t + k], s);
    dw[i] = s;
  }
  __syncthreads();
  // sepconv.1: grouped pointwise Conv1d(k = 1, stride 8)   (model.py:10-11)
  const int cpg = n_mels / groups, opg = H / groups;
  for (int i = tid; i < H * T2; i += nth) {
    const int oc = i % H, t = i / H;
    const int g = oc / opg;
    float s = w[o.pw_b + oc];
    for (int c = 0; c < cpg; c++) s = __builtin_fmaf(w[o.pw_w + (size_t)oc * cpg + c], dw[(size_t)(g * cpg + c) * T1 + 8 * t], s);
    seq0[(size_t)t * 2 * H + oc] = s;
  }
  __syncthreads();
  // 2-layer bidirectional GRU, gate order r, z, n (torch.nn.GRU), initial state 0   (model.py:22,31,99-100)
  for (int l = 0; l < 2; l++) {
    const int in = l == 0 ? H : 2 * H;
    for (int i = tid; i < 2 * T2 * 3 * H; i += nth) {            // input projections of every step, both directions
      const int j = i % (3 * H), t = (i / (3 * H)) % T2, d = i / (3 * H * T2);
      const float *wi = w + o.gru[l][d][0] + (size_t)j * in;
      float s = w[o.gru[l][d][2] + j];
      for (int k = 0; k < in; k++) s = __builtin_fmaf(wi[k], seq0[(size_t)t * 2 * H + k], s);
      gi[i] = s;
    }
    for (int i = tid; i < 2 * H; i += nth) hcur[i] = 0.f;
    __syncthreads();
    for (int s_ = 0; s_ < T2; s_++) {
      for (int i = tid; i < 2 * 3 * H; i += nth) {
        const int j = i % (3 * H), d = i / (3 * H);
        const float *wh = w + o.gru[l][d][1] + (size_t)j * H;
        float s = w[o.gru[l][d][3] + j];
        for (int k = 0; k < H; k++) s = __builtin_fmaf(wh[k], hcur[d * H + k], s);
        gh[i] = s;
      }
      __syncthreads();
      for (int i = tid; i < 2 * H; i += nth) {
        const int u = i % H, d = i / H;
        const int t = d == 0 ? s_ : T2 - 1 - s_;
        const float *g_i = gi + ((size_t)d * T2 + t) * 3 * H, *g_h = gh + d * 3 * H;
        const float r = sigmoidf_(g_i[u] + g_h[u]);
        const float z = sigmoidf_(g_i[H + u] + g_h[H + u]);
        const float n = tanhf(g_i[2 * H + u] + r * g_h[2 * H + u]);
        const float hn = (1.0f - z) * n + z * hcur[i];
        hcur[i] = hn;
        seq1[(size_t)t * 2 * H + d * H + u] = hn;
      }
      __syncthreads();
    }
    for (int i = tid; i < T2 * 2 * H; i += nth) seq0[i] = seq1[i];
    __syncthreads();
  }
  // additive attention e_t = Vt . tanh(Wx_b out_t + b)   (model.py:37-47,102-108)
  for (int t = 0; t < T2; t++) {
    for (int i = tid; i < 2 * H; i += nth) {
      const float *wr = w + o.wx_w + (size_t)i * 2 * H;
      float s = w[o.wx_b + i];
      for (int k = 0; k < 2 * H; k++) s = __builtin_fmaf(wr[k], seq0[(size_t)t * 2 * H + k], s);
      cvec[i] = tanhf(s) * w[o.vt + i];
    }
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int k = 0; k < 2 * H; k++) s += cvec[k];
      e[t] = s;
    }
    __syncthreads();
  }
  if (tid == 0) {                                                  // softmax over time (model.py:56)
    float mx = e[0];
    for (int t = 1; t < T2; t++) mx = fmaxf(mx, e[t]);
    float se = 0.f;
    for (int t = 0; t < T2; t++) { e[t] = expf(e[t] - mx); se += e[t]; }
    for (int t = 0; t < T2; t++) e[t] /= se;
  }
  __syncthreads();
  for (int i = tid; i < 2 * H; i += nth) {                         // c = a . out   (model.py:57)
    float s = 0.f;
    for (int t = 0; t < T2; t++) s = __builtin_fmaf(e[t], seq0[(size_t)t * 2 * H + i], s);
    cvec[i] = s;
  }
  __syncthreads();
  for (int k = tid; k < K; k += nth) {
    float s = 0.f;
    for (int i = 0; i < 2 * H; i++) s = __builtin_fmaf(w[o.u + (size_t)k * 2 * H + i], cvec[i], s);
    logit[k] = s;
  }
  __syncthreads();
  if (tid == 0) {                                                  // log_softmax (model.py:59)
    float mx = logit[0];
    for (int k = 1; k < K; k++) mx = fmaxf(mx, logit[k]);
    float se = 0.f;
    for (int k = 0; k < K; k++) se += expf(logit[k] - mx);
    const float lse = mx + logf(se);
    for (int k = 0; k < K; k++) logp[(size_t)b * K + k] = logit[k] - lse;
  }
}

// MelSpectrogram(sample_rate=16000, n_mels) with torchaudio's defaults + AmplitudeToDB('power'): n_fft = win = 400,
// hop 200, periodic Hann, centre + reflect padding, power spectrum, HTK filterbank without normalisation; 201 bins by a
// direct DFT with a 400-entry twiddle table (400 is not a power of two; 160 kMAC per frame).  One workgroup per frame.
struct HtkPts { float f[66]; };     // n_mels + 2 filter edge frequencies (Hz), n_mels <= 64

__global__ __launch_bounds__(256) void melspec_htk_kernel(const float *__restrict__ x, float *__restrict__ out, HtkPts pts,
                                                          int n_mels, int n_frames, int L) {
  constexpr int NF = 400, NB = 201;
  __shared__ float xs[NF], cs[NF], sn[NF], pw[NB];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float *xb = x + (size_t)b * L;
  for (int n = tid; n < NF; n += 256) {
    int idx = f * 200 + n - NF / 2;
    if (idx < 0) idx = -idx;                                       // reflect (no edge repeat)
    if (idx >= L) idx = 2 * (L - 1) - idx;
    idx = min(max(idx, 0), L - 1);
    const float ang = 6.283185307179586f * (float)n / (float)NF;
    float sv, cv;
    sincosf(ang, &sv, &cv);
    cs[n] = cv;
    sn[n] = sv;
    xs[n] = xb[idx] * (0.5f - 0.5f * cv);                          // periodic Hann
  }
  __syncthreads();
  for (int k = tid; k < NB; k += 256) {
    float re = 0.f, im = 0.f;
    int ph = 0;
    for (int n = 0; n < NF; n++) {
      re = __builtin_fmaf(xs[n], cs[ph], re);
      im = __builtin_fmaf(xs[n], sn[ph], im);
      ph += k;
      if (ph >= NF) ph -= NF;
    }
    pw[k] = re * re + im * im;
  }
  __syncthreads();
  for (int m = tid; m < n_mels; m += 256) {
    const float f0 = pts.f[m], f1 = pts.f[m + 1], f2 = pts.f[m + 2];
    float s = 0.f;
    for (int k = 0; k < NB; k++) {
      const float fr = 8000.0f * (float)k / (float)(NB - 1);
      const float wgt = fmaxf(0.f, fminf((fr - f0) / (f1 - f0), (f2 - fr) / (f2 - f1)));
      s = __builtin_fmaf(wgt, pw[k], s);
    }
    out[((size_t)b * n_mels + m) * n_frames + f] = 10.0f * log10f(fmaxf(s, 1e-10f));
  }
}

}  // namespace ap

using namespace ap;

extern "C" size_t ap_kws_blob_elems(int n_mels, int hidden, int num_classes) {
  if (n_mels < 1 || hidden < 1 || num_classes < 1) return 0;
  return kws_layout(n_mels, hidden, num_classes).total;
}

extern "C" int ap_kws_create(int n_mels, int hidden, int num_classes, const float *blob_dev, size_t n_elems, void *stream,
                             ap_kws **out) {
  if (!blob_dev || !out) { set_error("ap_kws_create: null argument"); return -22; }
  const int groups = n_mels / 20 > 0 ? n_mels / 20 : 1;
  if (n_mels < 1 || n_mels > 64 || hidden < 1 || hidden > 128 || num_classes < 1 || num_classes > 64 || n_mels % groups ||
      hidden % groups) {
    set_error("ap_kws_create: unsupported shape (n_mels <= 64, hidden <= 128, num_classes <= 64)");
    return -22;
  }
  if (n_elems != kws_layout(n_mels, hidden, num_classes).total) {
    set_error("ap_kws_create: blob has %zu elements, expected %zu", n_elems, kws_layout(n_mels, hidden, num_classes).total);
    return -22;
  }
  ap_kws *k = new (std::nothrow) ap_kws();
  if (!k) { set_error("out of host memory"); return -12; }
  k->n_mels = n_mels; k->hidden = hidden; k->num_classes = num_classes; k->groups = groups; k->n = n_elems; k->blob = nullptr;
  if (hipMalloc((void **)&k->blob, n_elems * sizeof(float)) != hipSuccess) { delete k; set_error("ap_kws_create: hipMalloc"); return -12; }
  AP_HIP(hipMemcpyAsync(k->blob, blob_dev, n_elems * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  AP_HIP(hipStreamSynchronize((hipStream_t)stream));
  *out = k;
  return 0;
}

extern "C" int ap_kws_destroy(ap_kws *k) {
  if (!k) return 0;
  if (k->blob) (void)hipFree(k->blob);
  delete k;
  return 0;
}

extern "C" int ap_kws_fwd(ap_kws *k, const float *mel, float *logprobs, int B, int T, void *stream) {
  if (!k || !mel || !logprobs || B < 1) { set_error("ap_kws_fwd: bad argument"); return -22; }
  const int T1 = (T - 5) / 2 + 1, T2 = T1 >= 1 ? (T1 - 1) / 8 + 1 : 0;
  if (T < 5 || T2 < 1) { set_error("ap_kws_fwd: %d mel frames are too few for the separable conv", T); return -22; }
  const int H = k->hidden;
  const size_t smem = ((size_t)k->n_mels * T1 + 2 * (size_t)T2 * 2 * H + (size_t)2 * T2 * 3 * H + 6 * H + 2 * H + T2 + 2 * H +
                       k->num_classes) * sizeof(float);
  if (smem > 160 * 1024) { set_error("ap_kws_fwd: %d mel frames need %zu bytes of LDS", T, smem); return -22; }
  static bool attr = false;
  if (!attr) {
    AP_HIP(hipFuncSetAttribute((const void *)kws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  kws_kernel<<<B, 256, smem, (hipStream_t)stream>>>(mel, logprobs, k->blob, kws_layout(k->n_mels, H, k->num_classes), k->n_mels, H,
                                                   k->num_classes, k->groups, T, T1, T2);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_melspec_db_htk(const float *x, float *out, int n_mels, int B, int L, void *stream) {
  if (!x || !out || B < 1 || L < 201) { set_error("ap_melspec_db_htk: bad argument (L >= 201 for reflect padding)"); return -22; }
  if (n_mels < 1 || n_mels > 64) { set_error("ap_melspec_db_htk: n_mels %d outside [1, 64]", n_mels); return -22; }
  HtkPts pts;
  const double m1 = 2595.0 * log10(1.0 + 8000.0 / 700.0);
  for (int i = 0; i < n_mels + 2; i++) pts.f[i] = (float)(700.0 * (pow(10.0, (m1 * i / (n_mels + 1)) / 2595.0) - 1.0));
  const int n_frames = 1 + L / 200;
  dim3 grid(n_frames, B);
  melspec_htk_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, out, pts, n_mels, n_frames, L);
  AP_HIP(hipGetLastError());
  return 0;
}
