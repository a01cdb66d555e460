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

// Backward of melspec_htk_kernel.  Pass 1 (one workgroup per frame): recompute the frame's spectrum, push d(out) back
// through dB / filterbank / |.|^2 / DFT / window into a per-frame segment gradient scr[b][f][400]; pass 2 gathers the
// overlapping segments and the reflected borders into dx (every dx sample is written once: no atomics).
__global__ __launch_bounds__(256) void melspec_htk_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                              float *__restrict__ scr, HtkPts pts, int n_mels, int n_frames,
                                                              int L) {
  constexpr int NF = 400, NB = 201;
  __shared__ float xs[NF], cs[NF], sn[NF], gre[NB], gim[NB], pw[NB], ds[64];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float *xb = x + (size_t)b * L;
  for (int n = tid; n < NF; n += 256) {
    int idx = f * 200 + n - NF / 2;
    if (idx < 0) idx = -idx;
    if (idx >= L) idx = 2 * (L - 1) - idx;
    idx = min(max(idx, 0), L - 1);
    const float ang = 6.283185307179586f * (float)n / (float)NF;
    float sv, cv;
    sincosf(ang, &sv, &cv);
    cs[n] = cv;
    sn[n] = sv;
    xs[n] = xb[idx] * (0.5f - 0.5f * cv);
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
    gre[k] = re;
    gim[k] = im;
    pw[k] = re * re + im * im;
  }
  __syncthreads();
  for (int m = tid; m < n_mels; m += 256) {
    const float f0 = pts.f[m], f1 = pts.f[m + 1], f2 = pts.f[m + 2];
    float s = 0.f;
    for (int k = 0; k < NB; k++) {
      const float fr = 8000.0f * (float)k / (float)(NB - 1);
      s = __builtin_fmaf(fmaxf(0.f, fminf((fr - f0) / (f1 - f0), (f2 - fr) / (f2 - f1))), pw[k], s);
    }
    const float g = dout[((size_t)b * n_mels + m) * n_frames + f];
    ds[m] = s > 1e-10f ? g * 4.342944819032518f / s : 0.f;         // d/ds 10 log10(max(s, 1e-10))
  }
  __syncthreads();
  for (int k = tid; k < NB; k += 256) {
    const float fr = 8000.0f * (float)k / (float)(NB - 1);
    float dp = 0.f;
    for (int m = 0; m < n_mels; m++) {
      const float f0 = pts.f[m], f1 = pts.f[m + 1], f2 = pts.f[m + 2];
      dp = __builtin_fmaf(fmaxf(0.f, fminf((fr - f0) / (f1 - f0), (f2 - fr) / (f2 - f1))), ds[m], dp);
    }
    gre[k] *= 2.0f * dp;
    gim[k] *= 2.0f * dp;
  }
  __syncthreads();
  float *so = scr + ((size_t)b * n_frames + f) * NF;
  for (int n = tid; n < NF; n += 256) {
    float a = 0.f;
    int ph = 0;
    for (int k = 0; k < NB; k++) {
      a = __builtin_fmaf(gre[k], cs[ph], a);
      a = __builtin_fmaf(gim[k], sn[ph], a);
      ph += n;
      if (ph >= NF) ph -= NF;
    }
    so[n] = a * (0.5f - 0.5f * cs[n]);
  }
}

__global__ void melspec_htk_bwd_gather_kernel(const float *__restrict__ scr, float *__restrict__ dx, int n_frames, int L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= L) return;
  const float *sb = scr + (size_t)b * n_frames * 400;
  auto seg = [&](int p) {                                          // sum over the frames covering padded position p
    float s = 0.f;
    const int f1 = p / 200;                                        // frames f with 0 <= p - 200 f < 400: f1 and f1 - 1
    if (f1 < n_frames) s += sb[(size_t)f1 * 400 + (p - 200 * f1)];
    if (f1 >= 1 && f1 - 1 < n_frames) s += sb[(size_t)(f1 - 1) * 400 + (p - 200 * (f1 - 1))];
    return s;
  };
  const int P = L + 400;                                           // padded length; only positions < 200 (n_frames - 1) + 400 are read
  float g = seg(t + 200);
  if (t >= 1 && t <= 200) g += seg(200 - t);                       // left reflection: padded p < 200 reads x[200 - p]
  const int pr = 2 * (L - 1) - t + 200;                            // right reflection: padded p >= L + 200 reads x[2(L-1) - (p - 200)]
  if (pr >= L + 200 && pr < P && t <= L - 2) g += seg(pr);
  dx[(size_t)b * L + t] = g;
}

// ---- input gradients of the KWS route (white-box attack of kws_adaptive_attack_eval.py:132-143) ------------------
// KWSModel: one workgroup per clip; the forward pass is recomputed keeping every GRU gate in a per-clip global scratch
// (sequences are a handful of steps), then back-propagation through log-softmax / U / attention / the two GRU layers
// (BPTT, both directions) / the separable conv.  Parameters are frozen: only d/d(mel) is formed.
struct KwsScr {                 // offsets (floats) into one clip's scratch
  size_t dw, seq[3], gate[2], e, a, cvec, dseq[2], din[2], ddw, total;
};
static __host__ __device__ KwsScr kws_scratch(int n_mels, int H, int T1, int T2) {
  KwsScr o;
  size_t p = 0;
  o.dw = p; p += (size_t)n_mels * T1;
  for (int i = 0; i < 3; i++) { o.seq[i] = p; p += (size_t)T2 * 2 * H; }     // layer inputs: seq[0] (H used), seq[1], output seq[2]
  for (int l = 0; l < 2; l++) { o.gate[l] = p; p += (size_t)2 * T2 * 5 * H; } // per dir, step: r, z, n, hn = W_hn h + b_hn, h_prev
  o.e = p; p += T2;
  o.a = p; p += T2;
  o.cvec = p; p += 2 * H;
  for (int i = 0; i < 2; i++) { o.dseq[i] = p; p += (size_t)T2 * 2 * H; }
  for (int i = 0; i < 2; i++) { o.din[i] = p; p += (size_t)T2 * 2 * H; }
  o.ddw = p; p += (size_t)n_mels * T1;
  o.total = p;
  return o;
}

__global__ __launch_bounds__(256) void kws_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dlogp,
                                                      float *__restrict__ dx, float *__restrict__ scratch,
                                                      const float *__restrict__ w, KwsOff o, KwsScr so, int n_mels, int H,
                                                      int K, int groups, int T, int T1, int T2) {
  __shared__ float gi[6 * 128], gh[6 * 128], hcur[2 * 128], dh[2 * 128], tmp[2 * 128], logit[64], dlogit[64];
  const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  float *S = scratch + (size_t)b * so.total;
  const float *xb = x + (size_t)b * n_mels * T;
  float *dw = S + so.dw;
  // ---------------- forward with saves ----------------
  for (int i = tid; i < n_mels * T1; i += nth) {
    const int c = i / T1, t = i - c * T1;
    float s = w[o.dw_b + c];
    for (int k = 0; k < 5; k++) s = __builtin_fmaf(w[o.dw_w + c * 5 + k], xb[(size_t)c * T + 2 * t + k], s);
    dw[i] = s;
  }
  __syncthreads();
  const int cpg = n_mels / groups, opg = H / groups;
  float *seq0 = S + so.seq[0];
  for (int i = tid; i < H * T2; i += nth) {
    const int oc = i % H, t = i / H, g = oc / opg;
    float s = w[o.pw_b + oc];
    for (int c = 0; c < cpg; c++) s = __builtin_fmaf(w[o.pw_w + (size_t)oc * cpg + c], dw[(size_t)(g * cpg + c) * T1 + 8 * t], s);
    seq0[(size_t)t * 2 * H + oc] = s;
  }
  __syncthreads();
  for (int l = 0; l < 2; l++) {
    const int in = l == 0 ? H : 2 * H;
    const float *sin = S + so.seq[l];
    float *sout = S + so.seq[l + 1], *G = S + so.gate[l];
    for (int i = tid; i < 2 * H; i += nth) hcur[i] = 0.f;
    __syncthreads();
    for (int s_ = 0; s_ < T2; s_++) {
      for (int i = tid; i < 2 * 3 * H; i += nth) {
        const int j = i % (3 * H), d = i / (3 * H);
        const int t = d == 0 ? s_ : T2 - 1 - s_;
        const float *wi = w + o.gru[l][d][0] + (size_t)j * in, *wh = w + o.gru[l][d][1] + (size_t)j * H;
        float a = w[o.gru[l][d][2] + j], c = w[o.gru[l][d][3] + j];
        for (int k = 0; k < in; k++) a = __builtin_fmaf(wi[k], sin[(size_t)t * 2 * H + k], a);
        for (int k = 0; k < H; k++) c = __builtin_fmaf(wh[k], hcur[d * H + k], c);
        gi[i] = a;
        gh[i] = c;
      }
      __syncthreads();
      for (int i = tid; i < 2 * H; i += nth) {
        const int u = i % H, d = i / H;
        const int t = d == 0 ? s_ : T2 - 1 - s_;
        const float *g_i = gi + d * 3 * H, *g_h = gh + d * 3 * H;
        const float r = sigmoidf_(g_i[u] + g_h[u]), z = sigmoidf_(g_i[H + u] + g_h[H + u]);
        const float n = tanhf(g_i[2 * H + u] + r * g_h[2 * H + u]);
        float *gs = G + ((size_t)d * T2 + t) * 5 * H;
        gs[u] = r; gs[H + u] = z; gs[2 * H + u] = n; gs[3 * H + u] = g_h[2 * H + u]; gs[4 * H + u] = hcur[i];
        const float hn = (1.0f - z) * n + z * hcur[i];
        tmp[i] = hn;
        sout[(size_t)t * 2 * H + d * H + u] = hn;
      }
      __syncthreads();
      for (int i = tid; i < 2 * H; i += nth) hcur[i] = tmp[i];
      __syncthreads();
    }
  }
  const float *out = S + so.seq[2];
  float *e = S + so.e, *av = S + so.a, *cvec = S + so.cvec;
  for (int t = tid; t < T2; t += nth) {                           // few steps: one thread per step
    float s = 0.f;
    for (int i = 0; i < 2 * H; i++) {
      const float *wr = w + o.wx_w + (size_t)i * 2 * H;
      float q = w[o.wx_b + i];
      for (int k = 0; k < 2 * H; k++) q = __builtin_fmaf(wr[k], out[(size_t)t * 2 * H + k], q);
      s = __builtin_fmaf(tanhf(q), w[o.vt + i], s);
    }
    e[t] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = e[0];
    for (int t = 1; t < T2; t++) mx = fmaxf(mx, e[t]);
    float se = 0.f;
    for (int t = 0; t < T2; t++) { av[t] = expf(e[t] - mx); se += av[t]; }
    for (int t = 0; t < T2; t++) av[t] /= se;
  }
  __syncthreads();
  for (int i = tid; i < 2 * H; i += nth) {
    float s = 0.f;
    for (int t = 0; t < T2; t++) s = __builtin_fmaf(av[t], out[(size_t)t * 2 * H + i], s);
    cvec[i] = s;
  }
  __syncthreads();
  for (int k = tid; k < K; k += nth) {
    float s = 0.f;
    for (int i = 0; i < 2 * H; i++) s = __builtin_fmaf(w[o.u + (size_t)k * 2 * H + i], cvec[i], s);
    logit[k] = s;
  }
  __syncthreads();
  // ---------------- backward ----------------
  if (tid == 0) {
    float mx = logit[0];
    for (int k = 1; k < K; k++) mx = fmaxf(mx, logit[k]);
    float se = 0.f, sd = 0.f;
    for (int k = 0; k < K; k++) { se += expf(logit[k] - mx); sd += dlogp[(size_t)b * K + k]; }
    for (int k = 0; k < K; k++) dlogit[k] = dlogp[(size_t)b * K + k] - expf(logit[k] - mx) / se * sd;
  }
  __syncthreads();
  for (int i = tid; i < 2 * H; i += nth) {                         // dc = U^T dlogit
    float s = 0.f;
    for (int k = 0; k < K; k++) s = __builtin_fmaf(w[o.u + (size_t)k * 2 * H + i], dlogit[k], s);
    tmp[i] = s;
  }
  __syncthreads();
  float *dout = S + so.dseq[1];                                    // gradient of the layer-1 output sequence
  if (tid == 0) {                                                  // da_t = dc . out_t ; softmax backward -> de
    float dot = 0.f;
    for (int t = 0; t < T2; t++) {
      float s = 0.f;
      for (int i = 0; i < 2 * H; i++) s = __builtin_fmaf(tmp[i], out[(size_t)t * 2 * H + i], s);
      e[t] = s;                                                    // reuse e as da
      dot = __builtin_fmaf(av[t], s, dot);
    }
    for (int t = 0; t < T2; t++) e[t] = av[t] * (e[t] - dot);      // de
  }
  __syncthreads();
  for (int i = tid; i < T2 * 2 * H; i += nth) dout[i] = av[i / (2 * H)] * tmp[i % (2 * H)];
  __syncthreads();
  for (int t = 0; t < T2; t++) {                                   // e_t = Vt . tanh(Wx out_t + b)
    for (int i = tid; i < 2 * H; i += nth) {
      const float *wr = w + o.wx_w + (size_t)i * 2 * H;
      float q = w[o.wx_b + i];
      for (int k = 0; k < 2 * H; k++) q = __builtin_fmaf(wr[k], out[(size_t)t * 2 * H + k], q);
      const float th = tanhf(q);
      dh[i] = e[t] * w[o.vt + i] * (1.0f - th * th);               // d pre-activation
    }
    __syncthreads();
    for (int k = tid; k < 2 * H; k += nth) {
      float s = 0.f;
      for (int i = 0; i < 2 * H; i++) s = __builtin_fmaf(w[o.wx_w + (size_t)i * 2 * H + k], dh[i], s);
      dout[(size_t)t * 2 * H + k] += s;
    }
    __syncthreads();
  }
  // GRU layers, last to first: BPTT in both directions.  Each (direction, t) input gradient is written exactly once
  // (din[d]) and the two directions are summed afterwards, so the result does not depend on an atomic's order.
  for (int l = 1; l >= 0; l--) {
    const int in = l == 0 ? H : 2 * H;
    const float *G = S + so.gate[l];
    const float *dso = S + so.dseq[l];                             // d(output sequence of this layer)
    float *din0 = S + so.din[0], *din1 = S + so.din[1];
    for (int i = tid; i < 2 * H; i += nth) dh[i] = 0.f;            // d h carried backwards through time
    __syncthreads();
    for (int s_ = T2 - 1; s_ >= 0; s_--) {
      for (int i = tid; i < 2 * H; i += nth) {
        const int u = i % H, d = i / H;
        const int t = d == 0 ? s_ : T2 - 1 - s_;
        const float *gs = G + ((size_t)d * T2 + t) * 5 * H;
        const float r = gs[u], z = gs[H + u], n = gs[2 * H + u], hnv = gs[3 * H + u], hp = gs[4 * H + u];
        const float dht = dh[i] + dso[(size_t)t * 2 * H + d * H + u];
        const float dpre_n = dht * (1.0f - z) * (1.0f - n * n);
        const float dpre_r = dpre_n * hnv * r * (1.0f - r);
        const float dpre_z = dht * (hp - n) * z * (1.0f - z);
        gi[d * 3 * H + u] = dpre_r;                                // d(input-side gate pre-activations)
        gi[d * 3 * H + H + u] = dpre_z;
        gi[d * 3 * H + 2 * H + u] = dpre_n;
        gh[d * 3 * H + u] = dpre_r;                                // hidden side: n's term is scaled by r
        gh[d * 3 * H + H + u] = dpre_z;
        gh[d * 3 * H + 2 * H + u] = dpre_n * r;
        tmp[i] = dht * z;                                          // direct path h_prev -> h
      }
      __syncthreads();
      for (int i = tid; i < 2 * H; i += nth) {                     // dh_prev = z dht + W_hh^T dgh
        const int u = i % H, d = i / H;
        float s = tmp[i];
        for (int jn = 0; jn < 3 * H; jn++) s = __builtin_fmaf(w[o.gru[l][d][1] + (size_t)jn * H + u], gh[d * 3 * H + jn], s);
        dh[i] = s;
      }
      for (int i = tid; i < 2 * in; i += nth) {                    // d input_t = W_ih^T dgi, per direction
        const int k = i % in, d = i / in;
        const int t = d == 0 ? s_ : T2 - 1 - s_;
        float s = 0.f;
        for (int jn = 0; jn < 3 * H; jn++) s = __builtin_fmaf(w[o.gru[l][d][0] + (size_t)jn * in + k], gi[d * 3 * H + jn], s);
        (d == 0 ? din0 : din1)[(size_t)t * in + k] = s;
      }
      __syncthreads();
    }
    float *dst = S + so.dseq[0];                                   // layer 1 -> d(layer-0 output) [T2][2H]; layer 0 -> d(pointwise out) [T2][H]
    for (int i = tid; i < T2 * in; i += nth) dst[i] = din0[i] + din1[i];
    __syncthreads();
  }
  // separable conv backward
  {
    const float *dpw = S + so.dseq[0];                             // [T2][H]
    float *ddw = S + so.ddw;                                       // [n_mels][T1]
    for (int i = tid; i < n_mels * T1; i += nth) {
      const int c = i / T1, t1 = i - c * T1;
      float s = 0.f;
      if (t1 % 8 == 0 && t1 / 8 < T2) {
        const int t2 = t1 / 8, g = c / cpg, cl = c - g * cpg;
        for (int oc = g * opg; oc < (g + 1) * opg; oc++) s = __builtin_fmaf(w[o.pw_w + (size_t)oc * cpg + cl], dpw[(size_t)t2 * H + oc], s);
      }
      ddw[i] = s;
    }
    __syncthreads();
    float *dxb = dx + (size_t)b * n_mels * T;
    for (int i = tid; i < n_mels * T; i += nth) {
      const int c = i / T, t = i - c * T;
      float s = 0.f;
      for (int k = 0; k < 5; k++) {
        const int tt = t - k;
        if (tt >= 0 && (tt & 1) == 0 && tt / 2 < T1) s = __builtin_fmaf(w[o.dw_w + c * 5 + k], ddw[(size_t)c * T1 + tt / 2], s);
      }
      dxb[i] = s;
    }
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

// d(input mel) of ap_kws_fwd for an upstream gradient on the log-probabilities.  scratch: ap_kws_bwd_scratch_elems floats.
extern "C" size_t ap_kws_bwd_scratch_elems(const ap_kws *k, int B, int T) {
  if (!k || B < 1 || T < 5) return 0;
  const int T1 = (T - 5) / 2 + 1, T2 = (T1 - 1) / 8 + 1;
  return kws_scratch(k->n_mels, k->hidden, T1, T2).total * (size_t)B;
}

extern "C" int ap_kws_bwd(ap_kws *k, const float *mel, const float *dlogprobs, float *dmel, float *scratch, int B, int T,
                          void *stream) {
  if (!k || !mel || !dlogprobs || !dmel || !scratch || B < 1) { set_error("ap_kws_bwd: bad argument"); return -22; }
  const int T1 = (T - 5) / 2 + 1, T2 = T1 >= 1 ? (T1 - 1) / 8 + 1 : 0;
  if (T < 5 || T2 < 1) { set_error("ap_kws_bwd: %d mel frames are too few for the separable conv", T); return -22; }
  if (k->hidden > 128 || k->num_classes > 64) { set_error("ap_kws_bwd: hidden <= 128 and num_classes <= 64"); return -22; }
  kws_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(mel, dlogprobs, dmel, scratch, k->blob,
                                                     kws_layout(k->n_mels, k->hidden, k->num_classes),
                                                     kws_scratch(k->n_mels, k->hidden, T1, T2), k->n_mels, k->hidden,
                                                     k->num_classes, k->groups, T, T1, T2);
  AP_HIP(hipGetLastError());
  return 0;
}

// d(waveform) of ap_melspec_db_htk.  scratch: B * (1 + L / 200) * 400 floats.
extern "C" int ap_melspec_db_htk_bwd(const float *x, const float *dout, float *dx, float *scratch, int n_mels, int B, int L,
                                     void *stream) {
  if (!x || !dout || !dx || !scratch || B < 1 || L < 201) { set_error("ap_melspec_db_htk_bwd: bad argument"); return -22; }
  if (n_mels < 1 || n_mels > 64) { set_error("ap_melspec_db_htk_bwd: n_mels %d outside [1, 64]", n_mels); return -22; }
  HtkPts pts;
  const double m1 = 2595.0 * log10(1.0 + 8000.0 / 700.0);
  for (int i = 0; i < n_mels + 2; i++) pts.f[i] = (float)(700.0 * (pow(10.0, (m1 * i / (n_mels + 1)) / 2595.0) - 1.0));
  const int n_frames = 1 + L / 200;
  melspec_htk_bwd_kernel<<<dim3(n_frames, B), 256, 0, (hipStream_t)stream>>>(x, dout, scratch, pts, n_mels, n_frames, L);
  AP_HIP(hipGetLastError());
  melspec_htk_bwd_gather_kernel<<<dim3((L + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(scratch, dx, n_frames, L);
  AP_HIP(hipGetLastError());
  return 0;
}
