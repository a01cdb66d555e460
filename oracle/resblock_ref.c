/* ORACLE — test infrastructure only (never linked into or called by the product).
 *
 * Plain-C restatement, independent of PyTorch/oneDNN, of the pieces of AudioPure's hot path that carry the
 * arithmetic: weight-norm fold, one DiffWave Residual_block.forward, and the DDPM x_{t-1} update.
 * Citations: diffusion_models/DiffWave_Unconditional/WaveNet.py (WN), diffusion_models/diffwave_ddpm.py (DD).
 * Built by oracle/Makefile into oracle/_build/liboracle_ref.so; tests/test_oracle_c.py checks it against the
 * PyTorch-CPU oracle (which is itself pinned to the reference's golden vectors).  Accumulation is in double so
 * that it is a tie-breaker between the two fp32 implementations it is compared with.
 */
#include <math.h>
#include <stddef.h>

/* W[o][i] = g[o] * v[o][i] / ||v[o]||_2   (nn.utils.weight_norm dim=0; WN:23-34) */
void ap_oracle_fold(const float *g, const float *v, float *w, int O, int IK) {
  for (int o = 0; o < O; o++) {
    double s = 0.0;
    for (int i = 0; i < IK; i++) s += (double)v[(size_t)o * IK + i] * v[(size_t)o * IK + i];
    const float scale = g[o] / (float)sqrt(s);
    for (int i = 0; i < IK; i++) w[(size_t)o * IK + i] = v[(size_t)o * IK + i] * scale;
  }
}

/* One Residual_block.forward (WN:75-97) on x [B][C][L] with the FiLM vector part_t [C] (= fc_t(emb), WN:82-83):
 *   u = x + part_t                       (in-place alias, WN:77,84: the residual uses u, not x)
 *   y = DilConv_{k=3, dilation d, zero pad d}(u) + b_dil                (WN:87, :26-27)
 *   g = tanh(y[:C]) * sigmoid(y[C:])                                    (WN:90)
 *   h_out = (u + W_res g + b_res) * sqrt(0.5) ; skip_out = W_skip g + b_skip   (WN:93-97) */
void ap_oracle_resblock(const float *x, const float *part_t, const float *w_dil, const float *b_dil,
                        const float *w_res, const float *b_res, const float *w_skip, const float *b_skip, int B, int C,
                        int S, int L, int d, float *h_out, float *skip_out, float *gate_scratch /* [C][L] */) {
  const float rs = (float)sqrt(0.5);
  for (int b = 0; b < B; b++) {
    const float *xb = x + (size_t)b * C * L;
    for (int c = 0; c < C; c++) {
#pragma omp parallel for
      for (int t = 0; t < L; t++) {
        double ya = b_dil[c], yb = b_dil[C + c];
        for (int ci = 0; ci < C; ci++) {
          for (int k = 0; k < 3; k++) {
            const int tp = t + (k - 1) * d;
            if (tp < 0 || tp >= L) continue;                      /* zero padding of u */
            const double u = (double)(xb[(size_t)ci * L + tp] + part_t[ci]);
            ya += (double)w_dil[((size_t)c * C + ci) * 3 + k] * u;
            yb += (double)w_dil[((size_t)(C + c) * C + ci) * 3 + k] * u;
          }
        }
        gate_scratch[(size_t)c * L + t] = (float)(tanh(ya) * (1.0 / (1.0 + exp(-yb))));
      }
    }
#pragma omp parallel for
    for (int t = 0; t < L; t++) {
      for (int o = 0; o < C; o++) {
        double r = b_res[o];
        for (int c = 0; c < C; c++) r += (double)w_res[(size_t)o * C + c] * gate_scratch[(size_t)c * L + t];
        const float u = xb[(size_t)o * L + t] + part_t[o];
        h_out[((size_t)b * C + o) * L + t] = (float)((u + r) * rs);
      }
      for (int o = 0; o < S; o++) {
        double r = b_skip[o];
        for (int c = 0; c < C; c++) r += (double)w_skip[(size_t)o * C + c] * gate_scratch[(size_t)c * L + t];
        skip_out[((size_t)b * S + o) * L + t] = (float)r;
      }
    }
  }
}

/* One reverse DDPM step (DD:159-160, :99-102): mu = (x - (1-a)/sqrt(1-ab) eps)/sqrt(a); x' = mu + sigma z (t>0) */
void ap_oracle_ddpm_step(const float *x, const float *eps, const float *z, float alpha, float alpha_bar, float sigma,
                         int t, size_t n, float *out) {
  const float c2 = (1.0f - alpha) / sqrtf(1.0f - alpha_bar), sa = sqrtf(alpha);
  for (size_t i = 0; i < n; i++) {
    const float mu = (x[i] - c2 * eps[i]) / sa;
    out[i] = (t > 0) ? mu + sigma * z[i] : mu;
  }
}
