// Mel-dB front-end exactly as the eval scripts configure torchaudio (adaptive_attack_eval.py:83-85;
// SURVEY.md Appendix A.4): center=True zero padding, periodic Hann 2048, hop 512, |rFFT|^2 (1025 bins),
// slaney mel filterbank with slaney area norm, 10*log10(clamp(., 1e-10)).  One workgroup per (clip, frame):
// 2048-point Stockham radix-2 FFT entirely in LDS (44 KB), then one thread per mel filter.
#include <math.h>

#include "ap_common.h"

namespace ap {

constexpr int NFFT = 2048, HOP = 512, NBIN = NFFT / 2 + 1, MAX_MELS = 128;

struct MelPts {
  float f[MAX_MELS + 2];   // filter corner frequencies in Hz (n_mels + 2 used)
};

// forward 2048-point Stockham radix-2 FFT in LDS (all 256 threads); returns the buffer holding the result
__device__ __forceinline__ float2 *fft2048(float2 *bufA, float2 *bufB, const float2 *tw, int tid) {
  float2 *src = bufA, *dst = bufB;
#pragma unroll 1
  for (int Ns = 1; Ns < NFFT; Ns <<= 1) {
    const int tstride = (NFFT / 2) / Ns;
    for (int jj = tid; jj < NFFT / 2; jj += 256) {
      const int k = jj & (Ns - 1);
      const float2 w = tw[k * tstride];
      const float2 a = src[jj];
      const float2 c = src[jj + NFFT / 2];
      const float2 bw = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
      const int o = ((jj - k) << 1) + k;
      dst[o] = make_float2(a.x + bw.x, a.y + bw.y);
      dst[o + Ns] = make_float2(a.x - bw.x, a.y - bw.y);
    }
    __syncthreads();
    float2 *t = src; src = dst; dst = t;
  }
  return src;
}

__global__ __launch_bounds__(256) void melspec_kernel(const float *__restrict__ x, float *__restrict__ out, MelPts pts,
                                                      int n_mels, int n_frames, int L) {
  __shared__ float2 bufA[NFFT];
  __shared__ float2 bufB[NFFT];
  __shared__ float2 tw[NFFT / 2];
  __shared__ float pw[NBIN];
  const int tid = threadIdx.x;
  const int frame = blockIdx.x, b = blockIdx.y;
  const float *xb = x + (size_t)b * L;
  for (int m = tid; m < NFFT / 2; m += 256) {
    float s, c;
    sincospif((float)m * (1.0f / 1024.0f), &s, &c);   // exp(-2 pi i m / 2048)
    tw[m] = make_float2(c, -s);
  }
  for (int n = tid; n < NFFT; n += 256) {
    const int t = frame * HOP - NFFT / 2 + n;            // center=True, pad_mode='constant'
    const float v = (t >= 0 && t < L) ? xb[t] : 0.f;
    const float w = 0.5f - 0.5f * cospif((float)n * (1.0f / 1024.0f));   // periodic hann: 0.5 - 0.5 cos(2 pi n / N)
    bufA[n] = make_float2(v * w, 0.f);
  }
  __syncthreads();
  float2 *src = fft2048(bufA, bufB, tw, tid);
  for (int k = tid; k < NBIN; k += 256) {
    const float2 v = src[k];
    pw[k] = v.x * v.x + v.y * v.y;          // power = 2
  }
  __syncthreads();
  if (tid < n_mels) {
    const float f0 = pts.f[tid], f1 = pts.f[tid + 1], f2 = pts.f[tid + 2];
    const float binhz = 8000.0f / (float)(NBIN - 1);   // all_freqs = linspace(0, sr/2, n_freqs)
    int lo = (int)floorf(f0 / binhz), hi = (int)ceilf(f2 / binhz);
    lo = max(lo, 0);
    hi = min(hi, NBIN - 1);
    const float enorm = 2.0f / (f2 - f0);              // slaney area normalisation
    const float id = 1.0f / (f1 - f0), iu = 1.0f / (f2 - f1);
    float s = 0.f;
    for (int k = lo; k <= hi; k++) {
      const float fr = (float)k * binhz;
      const float w = fmaxf(0.f, fminf((fr - f0) * id, (f2 - fr) * iu));
      s = __builtin_fmaf(w * enorm, pw[k], s);
    }
    out[((size_t)b * n_mels + tid) * n_frames + frame] = 10.0f * log10f(fmaxf(s, 1e-10f));   // AmplitudeToDB('power')
  }
}

// d(loss)/dx through the mode-0 front-end (white-box attack on a spectrogram classifier, white_box_attack.py:392,437):
// per (clip, frame) recompute the spectrum, then  dmel = dout 10 / (ln10 mel)  (0 where the 1e-10 clamp is active),
// dP[k] = sum_m fb[k][m] dmel[m],  G[k] = 2 dP[k] X[k]  (k <= 1024),  dseg[n] = Re sum_k G[k] e^{+2 pi i k n / N}
// = Re FFT(conj(G) zero-extended)[n]  -- the same forward FFT --, times the window, into scratch [B][F][2048];
// melspec_bwd_gather then sums the (up to 4) frames that cover each sample: no atomics, deterministic.
__global__ __launch_bounds__(256) void melspec_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                          float *__restrict__ scratch, MelPts pts, int n_mels,
                                                          int n_frames, int L) {
  __shared__ float2 bufA[NFFT];
  __shared__ float2 bufB[NFFT];
  __shared__ float2 tw[NFFT / 2];
  __shared__ float dm[MAX_MELS];
  const int tid = threadIdx.x;
  const int frame = blockIdx.x, b = blockIdx.y;
  const float *xb = x + (size_t)b * L;
  for (int m = tid; m < NFFT / 2; m += 256) {
    float s, c;
    sincospif((float)m * (1.0f / 1024.0f), &s, &c);
    tw[m] = make_float2(c, -s);
  }
  for (int n = tid; n < NFFT; n += 256) {
    const int t = frame * HOP - NFFT / 2 + n;
    const float v = (t >= 0 && t < L) ? xb[t] : 0.f;
    const float w = 0.5f - 0.5f * cospif((float)n * (1.0f / 1024.0f));
    bufA[n] = make_float2(v * w, 0.f);
  }
  __syncthreads();
  float2 *X = fft2048(bufA, bufB, tw, tid);
  float2 *other = (X == bufA) ? bufB : bufA;
  const float binhz = 8000.0f / (float)(NBIN - 1);
  if (tid < n_mels) {
    const float f0 = pts.f[tid], f1 = pts.f[tid + 1], f2 = pts.f[tid + 2];
    int lo = max((int)floorf(f0 / binhz), 0), hi = min((int)ceilf(f2 / binhz), NBIN - 1);
    const float enorm = 2.0f / (f2 - f0), id = 1.0f / (f1 - f0), iu = 1.0f / (f2 - f1);
    float s = 0.f;
    for (int k = lo; k <= hi; k++) {
      const float fr = (float)k * binhz;
      const float w = fmaxf(0.f, fminf((fr - f0) * id, (f2 - fr) * iu));
      s = __builtin_fmaf(w * enorm, X[k].x * X[k].x + X[k].y * X[k].y, s);
    }
    const float g = dout[((size_t)b * n_mels + tid) * n_frames + frame];
    dm[tid] = s > 1e-10f ? g * 4.342944819032518f / s : 0.f;      // 10 / ln 10
  }
  __syncthreads();
  for (int k = tid; k < NFFT; k += 256) {
    float2 v = make_float2(0.f, 0.f);
    if (k < NBIN) {
      const float fr = (float)k * binhz;
      float dp = 0.f;
      for (int m = 0; m < n_mels; m++) {
        const float f0 = pts.f[m], f1 = pts.f[m + 1], f2 = pts.f[m + 2];
        if (fr > f0 && fr < f2) {
          const float w = fmaxf(0.f, fminf((fr - f0) / (f1 - f0), (f2 - fr) / (f2 - f1)));
          dp = __builtin_fmaf(w * 2.0f / (f2 - f0), dm[m], dp);
        }
      }
      v = make_float2(2.0f * dp * X[k].x, -2.0f * dp * X[k].y);    // conj(G[k])
    }
    other[k] = v;
  }
  __syncthreads();
  float2 *R = fft2048(other, X, tw, tid);                          // X's storage is the scratch buffer now
  float *sc = scratch + ((size_t)b * n_frames + frame) * NFFT;
  for (int n = tid; n < NFFT; n += 256) {
    const float w = 0.5f - 0.5f * cospif((float)n * (1.0f / 1024.0f));
    sc[n] = R[n].x * w;
  }
}

__global__ void melspec_bwd_gather_kernel(const float *__restrict__ scratch, float *__restrict__ dx, int n_frames, int L) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  // frames f with 0 <= t - (f HOP - NFFT/2) < NFFT
  const int u = t + NFFT / 2;
  int fhi = u / HOP, flo = (u - NFFT + HOP) / HOP;
  if (u - NFFT + 1 <= 0) flo = 0;
  fhi = min(fhi, n_frames - 1);
  float s = 0.f;
  for (int f = flo; f <= fhi; f++) s += scratch[((size_t)b * n_frames + f) * NFFT + (u - f * HOP)];
  dx[(size_t)b * L + t] = s;
}

// librosa.power_to_db(S, ref=np.max) (top_db = 80) on the dB values of one clip: db - max(db), floored at -80
// (transforms/transforms_stft.py:111-113).
__global__ __launch_bounds__(256) void mel_refmax_kernel(float *__restrict__ out, int n) {
  __shared__ float red[256];
  float *p = out + (size_t)blockIdx.x * n;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, p[i]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  for (int i = threadIdx.x; i < n; i += 256) p[i] = fmaxf(p[i] - mx, -80.0f);
}

static double hz_to_mel_slaney(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz_slaney(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

}  // namespace ap

extern "C" int ap_melspec_db_bwd(const float *x, const float *dout, float *dx, float *scratch, int n_mels, int B, int L,
                                 void *stream) {
  using namespace ap;
  if (!x || !dout || !dx || !scratch || B < 1 || L < 1) { set_error("ap_melspec_db_bwd: bad argument"); return -22; }
  if (n_mels < 1 || n_mels > MAX_MELS) { set_error("ap_melspec_db_bwd: n_mels %d outside [1, %d]", n_mels, MAX_MELS); return -22; }
  MelPts pts;
  const double m0 = hz_to_mel_slaney(0.0), m1 = hz_to_mel_slaney(8000.0);
  for (int i = 0; i < n_mels + 2; i++) pts.f[i] = (float)mel_to_hz_slaney(m0 + (m1 - m0) * i / (n_mels + 1));
  const int n_frames = 1 + L / HOP;
  hipStream_t st = (hipStream_t)stream;
  melspec_bwd_kernel<<<dim3(n_frames, B), 256, 0, st>>>(x, dout, scratch, pts, n_mels, n_frames, L);
  melspec_bwd_gather_kernel<<<dim3((L + 255) / 256, B), 256, 0, st>>>(scratch, dx, n_frames, L);
  AP_HIP(hipGetLastError());
  return 0;
}

extern "C" int ap_melspec_db(const float *x, float *out, int n_mels, int mode, int B, int L, void *stream) {
  using namespace ap;
  if (!x || !out || B < 1 || L < 1) { set_error("ap_melspec_db: bad argument"); return -22; }
  if (n_mels < 1 || n_mels > MAX_MELS) { set_error("ap_melspec_db: n_mels %d outside [1, %d]", n_mels, MAX_MELS); return -22; }
  if (mode != 0 && mode != 1) { set_error("ap_melspec_db: mode %d", mode); return -22; }
  MelPts pts;
  const double m0 = hz_to_mel_slaney(0.0), m1 = hz_to_mel_slaney(8000.0);
  for (int i = 0; i < n_mels + 2; i++) pts.f[i] = (float)mel_to_hz_slaney(m0 + (m1 - m0) * i / (n_mels + 1));
  const int n_frames = 1 + L / HOP;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(n_frames, B);
  melspec_kernel<<<grid, 256, 0, st>>>(x, out, pts, n_mels, n_frames, L);
  if (mode == 1) mel_refmax_kernel<<<B, 256, 0, st>>>(out, n_mels * n_frames);
  AP_HIP(hipGetLastError());
  return 0;
}
