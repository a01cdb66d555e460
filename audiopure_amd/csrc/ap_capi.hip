// C-ABI of include/audiopure.h: context, weight ingestion, the epsilon-network driver and the
// sampling chains (DDPM / VP-SDE Euler / one-shot).  Host logic only; kernels live in ap_kernels.hip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "ap_common.h"

namespace ap {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap_;
  va_start(ap_, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap_);
  va_end(ap_);
}

int hip_fail(hipError_t e, const char *what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return -5;  // -EIO
}

BlobLayout blob_layout(const ap_config &c) {
  BlobLayout b;
  const size_t C = c.res_channels, S = c.skip_channels, Ein = c.embed_dim_in, Emid = c.embed_dim_mid,
               Eout = c.embed_dim_out;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += n; return r; };
  b.init_b = take(C); b.init_g = take(C); b.init_v = take(C);
  b.fc1_w = take(Emid * Ein); b.fc1_b = take(Emid);
  b.fc2_w = take(Eout * Emid); b.fc2_b = take(Eout);
  b.blk0 = o;
  size_t q = 0;
  auto tk = [&](size_t n) { size_t r = q; q += n; return r; };
  b.fct_w = tk(C * Eout); b.fct_b = tk(C);
  b.dil_b = tk(2 * C); b.dil_g = tk(2 * C); b.dil_v = tk(2 * C * C * 3);
  b.res_b = tk(C); b.res_g = tk(C); b.res_v = tk(C * C);
  b.skip_b = tk(S); b.skip_g = tk(S); b.skip_v = tk(S * C);
  b.blk_stride = q;
  o += q * (size_t)c.num_res_layers;
  b.f1_b = take(S); b.f1_g = take(S); b.f1_v = take(S * S);
  b.f2_w = take(S); b.f2_b = take(1);
  b.total = o;
  return b;
}

static int check_cfg(const ap_config &c) {
  if (c.res_channels != c.skip_channels) {
    set_error("config: skip_channels (%d) must equal res_channels (%d) in this build", c.skip_channels,
              c.res_channels);
    return -22;
  }
  if (c.res_channels != 64 && c.res_channels != 128 && c.res_channels != 256) {
    set_error("config: res_channels %d unsupported (64, 128, 256)", c.res_channels);
    return -22;
  }
  if (c.num_res_layers < 1 || c.dilation_cycle < 1 || c.dilation_cycle > 20 || c.T < 1 || c.embed_dim_in % 2 ||
      c.embed_dim_in < 2 || c.embed_dim_mid < 1 || c.embed_dim_out < 1) {
    set_error("config: invalid layer/embedding/schedule sizes");
    return -22;
  }
  if (c.precision != AP_PREC_F32 && c.precision != AP_PREC_BF16 && c.precision != AP_PREC_F32_SPLIT && c.precision != AP_PREC_BF16_STORE) {
    set_error("config: precision %d not built (AP_PREC_F32, AP_PREC_BF16, AP_PREC_F32_SPLIT, AP_PREC_BF16_STORE)", c.precision);
    return -22;
  }
  if (c.precision != AP_PREC_F32 && c.res_channels != 256) {
    set_error("config: AP_PREC_BF16 / AP_PREC_BF16_STORE / AP_PREC_F32_SPLIT are built for res_channels = 256 only (got %d)",
              c.res_channels);
    return -22;
  }
  return 0;
}

}  // namespace ap

using namespace ap;

extern "C" const char *ap_last_error(void) { return g_err; }
extern "C" int ap_version(void) { return 100; }

extern "C" size_t ap_wavenet_blob_elems(const ap_config *cfg) {
  if (!cfg) return 0;
  return blob_layout(*cfg).total;
}

extern "C" int ap_ctx_create(const ap_config *cfg, ap_ctx **out) {
  if (!cfg || !out) { set_error("ap_ctx_create: null argument"); return -22; }
  int rc = check_cfg(*cfg);
  if (rc) return rc;
  ap_ctx *c = new (std::nothrow) ap_ctx();
  if (!c) { set_error("out of host memory"); return -12; }
  c->cfg = *cfg;
  c->C = cfg->res_channels; c->S = cfg->skip_channels; c->NL = cfg->num_res_layers; c->NW = c->C / 64;
  c->loaded = false;
  c->slab = nullptr;
  c->slab_bf = nullptr;
  c->w1p_bf = c->w2p_bf = c->wf1p_bf = c->w1q_bf = nullptr;
  c->slab_s = nullptr;
  c->slab_w = nullptr;
  c->w1w = c->w2w = nullptr;
  c->slab_b = nullptr;
  c->slab_bb = nullptr;
  c->w2t = c->w1b = nullptr;
  c->bwd_ready = false;
  c->f32_form = cfg->precision == AP_PREC_F32_SPLIT ? 0 : 1;     // (the split mode's F(2,3) form is opt-in: measured +3 %, looser adversarial bound)
  c->w1p_s = c->w2p_s = c->w1w_s = nullptr;
  c->profile = false;
  c->ev_used = 0;
  c->skip_group = 0;
  const int T = cfg->T;
  // calc_diffusion_hyperparams (util.py:96-123), closed form in double rounded to fp32
  c->Beta.resize(T); c->Alpha.resize(T); c->Alpha_bar.resize(T); c->Sigma.resize(T);
  c->sde_beta.resize(T); c->sde_ac.resize(T);
  double ab = 1.0, ab_prev = 1.0;
  for (int t = 0; t < T; t++) {
    double beta = (T > 1) ? (double)cfg->beta_0 + ((double)cfg->beta_T - (double)cfg->beta_0) * t / (T - 1)
                          : (double)cfg->beta_0;
    float bf = (float)beta;
    float af = 1.0f - bf;
    ab_prev = ab;
    ab *= (double)af;
    c->Beta[t] = bf;
    c->Alpha[t] = af;
    c->Alpha_bar[t] = (float)ab;
    double bt = (t == 0) ? (double)bf : (double)bf * (1.0 - ab_prev) / (1.0 - ab);
    c->Sigma[t] = (float)sqrt(bt);
    c->sde_beta[t] = bf;
    c->sde_ac[t] = (float)ab;
  }
  *out = c;
  return 0;
}

extern "C" int ap_ctx_destroy(ap_ctx *ctx) {
  if (!ctx) return 0;
  if (ctx->slab) (void)hipFree(ctx->slab);
  if (ctx->slab_bf) (void)hipFree(ctx->slab_bf);
  if (ctx->slab_s) (void)hipFree(ctx->slab_s);
  if (ctx->slab_w) (void)hipFree(ctx->slab_w);
  if (ctx->slab_b) (void)hipFree(ctx->slab_b);
  if (ctx->slab_bb) (void)hipFree(ctx->slab_bb);
  for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
  delete ctx;
  return 0;
}

extern "C" int ap_profile_enable(ap_ctx *ctx, int enable) {
  if (!ctx) { set_error("ap_profile_enable: null ctx"); return -22; }
  ctx->profile = enable != 0;
  ctx->ev_used = 0;
  return 0;
}

extern "C" int ap_profile_is_enabled(ap_ctx *ctx) { return ctx && ctx->profile ? 1 : 0; }

// per kind: [0] residual-block launches, [1] skip-GEMM launches of the deferred-skip form
static int profile_sum(ap_ctx *ctx, double ms_by_kind[2], int64_t n_by_kind[2]) {
  ms_by_kind[0] = ms_by_kind[1] = 0.0;
  n_by_kind[0] = n_by_kind[1] = 0;
  for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
    AP_HIP(hipEventSynchronize(ctx->ev[i + 1]));
    float ms = 0.f;
    AP_HIP(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
    const int k = (i / 2 < ctx->ev_kind.size() && ctx->ev_kind[i / 2]) ? 1 : 0;
    ms_by_kind[k] += ms;
    n_by_kind[k] += 1;
  }
  ctx->ev_used = 0;
  return 0;
}

extern "C" int ap_profile_read(ap_ctx *ctx, double *total_ms, int64_t *launches) {
  if (!ctx || !total_ms || !launches) { set_error("ap_profile_read: null argument"); return -22; }
  double ms[2];
  int64_t n[2];
  int rc = profile_sum(ctx, ms, n);
  if (rc) return rc;
  *total_ms = ms[0] + ms[1];                                      // the skip GEMMs are part of the layers' work: their time is folded in,
  *launches = n[0];                                               // the count stays "one per residual layer"
  return 0;
}

extern "C" int ap_profile_read_split(ap_ctx *ctx, double *ms_by_kind, int64_t *launches_by_kind) {
  if (!ctx || !ms_by_kind || !launches_by_kind) { set_error("ap_profile_read_split: null argument"); return -22; }
  return profile_sum(ctx, ms_by_kind, launches_by_kind);
}

extern "C" int ap_ctx_set_skip_group(ap_ctx *ctx, int layers_per_group) {
  if (!ctx) { set_error("ap_ctx_set_skip_group: null ctx"); return -22; }
  if (layers_per_group < 0 || layers_per_group > ctx->NL) { set_error("ap_ctx_set_skip_group: %d outside [0, %d]", layers_per_group, ctx->NL); return -22; }
  if (layers_per_group > 0 && ((ctx->cfg.precision != AP_PREC_BF16 && ctx->cfg.precision != AP_PREC_BF16_STORE) || ctx->C != 256 || ctx->S != 256)) {
    set_error("ap_ctx_set_skip_group: the deferred-skip form is built for AP_PREC_BF16 / AP_PREC_BF16_STORE with res = skip = 256 channels");
    return -22;
  }
  ctx->skip_group = layers_per_group;
  return 0;
}

extern "C" int ap_ctx_set_f32_form(ap_ctx *ctx, int form) {
  if (!ctx) { set_error("ap_ctx_set_f32_form: null ctx"); return -22; }
  if (form != 0 && form != 1) { set_error("ap_ctx_set_f32_form: form %d (0 direct, 1 minimal-filtering)", form); return -22; }
  if (ctx->cfg.precision != AP_PREC_F32 && ctx->cfg.precision != AP_PREC_F32_SPLIT) {
    set_error("ap_ctx_set_f32_form: AP_PREC_F32 / AP_PREC_F32_SPLIT contexts only");
    return -22;
  }
  ctx->f32_form = form;
  return 0;
}

extern "C" int ap_ctx_get_f32_form(ap_ctx *ctx) {
  return ctx && (ctx->cfg.precision == AP_PREC_F32 || ctx->cfg.precision == AP_PREC_F32_SPLIT) && ctx->f32_form == 1 && ctx->C == 256 &&
                 ctx->S == 256
             ? 1
             : 0;
}

extern "C" int ap_ctx_set_schedule(ap_ctx *ctx, const float *beta, const float *alpha, const float *alpha_bar,
                                   const float *sigma, int T) {
  if (!ctx || !beta || !alpha || !alpha_bar || !sigma) { set_error("set_schedule: null argument"); return -22; }
  if (T < 1) { set_error("set_schedule: T must be >= 1"); return -22; }
  ctx->Beta.assign(beta, beta + T);
  ctx->Alpha.assign(alpha, alpha + T);
  ctx->Alpha_bar.assign(alpha_bar, alpha_bar + T);
  ctx->Sigma.assign(sigma, sigma + T);
  ctx->cfg.T = T;
  return 0;
}

extern "C" int ap_ctx_set_sde_schedule(ap_ctx *ctx, const float *betas, const float *ac, int T) {
  if (!ctx || !betas || !ac || T < 1) { set_error("set_sde_schedule: bad argument"); return -22; }
  ctx->sde_beta.assign(betas, betas + T);
  ctx->sde_ac.assign(ac, ac + T);
  return 0;
}

extern "C" int ap_ctx_get_schedule(ap_ctx *ctx, int which, float *out_host, int n) {
  if (!ctx || !out_host) { set_error("get_schedule: null argument"); return -22; }
  const std::vector<float> *v = nullptr;
  switch (which) {
    case 0: v = &ctx->Beta; break;
    case 1: v = &ctx->Alpha; break;
    case 2: v = &ctx->Alpha_bar; break;
    case 3: v = &ctx->Sigma; break;
    case 4: v = &ctx->sde_beta; break;
    case 5: v = &ctx->sde_ac; break;
    default: set_error("get_schedule: which=%d", which); return -22;
  }
  if (n != (int)v->size()) { set_error("get_schedule: n=%d but table has %zu entries", n, v->size()); return -22; }
  memcpy(out_host, v->data(), sizeof(float) * n);
  return 0;
}

extern "C" int ap_ctx_load_wavenet(ap_ctx *ctx, const float *blob_dev, size_t n_elems, const float *embed_freq_dev,
                                   void *stream) {
  if (!ctx || !blob_dev || !embed_freq_dev) { set_error("load_wavenet: null argument"); return -22; }
  const ap_config &c = ctx->cfg;
  BlobLayout bl = blob_layout(c);
  if (n_elems != bl.total) {
    set_error("load_wavenet: blob has %zu elements, config needs %zu", n_elems, bl.total);
    return -22;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t C = ctx->C, S = ctx->S, NL = ctx->NL;
  const size_t Ein = c.embed_dim_in, Emid = c.embed_dim_mid, Eout = c.embed_dim_out;
  if (!ctx->slab) {
    size_t n = 0;
    auto take = [&](size_t k) { size_t r = n; n += (k + 63) & ~(size_t)63; return r; };   // 256-byte aligned pieces
    size_t o_freq = take(Ein / 2), o_fc1w = take(Emid * Ein), o_fc1b = take(Emid), o_fc2w = take(Eout * Emid),
           o_fc2b = take(Eout), o_fctw = take(NL * C * Eout), o_fctb = take(NL * C), o_w0 = take(C), o_b0 = take(C),
           o_w1f = take(NL * 2 * C * C * 3), o_w2f = take(NL * (C + S) * C), o_wf1f = take(S * S),
           o_b1 = take(NL * 2 * C), o_b2 = take(NL * (C + S)), o_bf1 = take(S), o_wf2 = take(S), o_bf2 = take(64),
           o_w1p = take(NL * 2 * C * C * 3), o_w2p = take(NL * (C + S) * C), o_wf1p = take(S * S),
           o_norm = take(2 * C > S ? 2 * C : S);
    float *p = nullptr;
    AP_HIP(hipMalloc((void **)&p, n * sizeof(float)));
    ctx->slab = p; ctx->slab_elems = n;
    ctx->emb_freq = p + o_freq; ctx->fc1_w = p + o_fc1w; ctx->fc1_b = p + o_fc1b; ctx->fc2_w = p + o_fc2w;
    ctx->fc2_b = p + o_fc2b; ctx->fct_w = p + o_fctw; ctx->fct_b = p + o_fctb; ctx->w0 = p + o_w0; ctx->b0 = p + o_b0;
    ctx->w1f = p + o_w1f; ctx->w2f = p + o_w2f; ctx->wf1f = p + o_wf1f; ctx->b1 = p + o_b1; ctx->b2 = p + o_b2;
    ctx->bf1 = p + o_bf1; ctx->wf2 = p + o_wf2; ctx->bf2 = p + o_bf2; ctx->w1p = p + o_w1p; ctx->w2p = p + o_w2p;
    ctx->wf1p = p + o_wf1p; ctx->norms = p + o_norm;
  }
  AP_HIP(hipMemcpyAsync(ctx->emb_freq, embed_freq_dev, sizeof(float) * (Ein / 2), hipMemcpyDeviceToDevice, st));
  int rc = launch_fold_and_pack(ctx, blob_dev, st);
  if (rc) return rc;
  ctx->bwd_ready = false;                                       // (re-loaded weights: the backward images are rebuilt by the next ap_ctx_prepare_backward)
  if (c.precision == AP_PREC_BF16 || c.precision == AP_PREC_BF16_STORE) {
    const size_t n1 = NL * 2 * C * C * 3, n2 = NL * (C + S) * C;
    if (!ctx->slab_bf) {
#ifdef AP_TOOLS                                                   // + the 16x16x32 GEMM1 image of the tools library's M16 instantiation
      AP_HIP(hipMalloc(&ctx->slab_bf, (n1 + n2 + (size_t)S * S + n1) * 2));
#else
      AP_HIP(hipMalloc(&ctx->slab_bf, (n1 + n2 + (size_t)S * S) * 2));
#endif
      ctx->w1p_bf = ctx->slab_bf;
      ctx->w2p_bf = (char *)ctx->slab_bf + n1 * 2;
      ctx->wf1p_bf = (char *)ctx->slab_bf + (n1 + n2) * 2;     // final conv's first 1x1 (bf16 A-operand image)
#ifdef AP_TOOLS
      ctx->w1q_bf = (char *)ctx->slab_bf + (n1 + n2 + (size_t)S * S) * 2;   // GEMM1 image for the 16x16x32 MFMA shape
#endif
    }
    rc = launch_pack_bf16(ctx, st);
    if (rc) return rc;
  }
  if (c.precision == AP_PREC_F32 && C == 256 && S == 256) {
    const size_t n1w = NL * 4 * 2 * C * C, n2 = NL * (C + S) * C;
    if (!ctx->slab_w) {
      AP_HIP(hipMalloc(&ctx->slab_w, (n1w + n2) * sizeof(float)));
      ctx->w1w = (float *)ctx->slab_w;
      ctx->w2w = ctx->w1w + n1w;
    }
    rc = launch_pack_f32w(ctx, st);
    if (rc) return rc;
  }
  if (c.precision == AP_PREC_F32_SPLIT) {
    const size_t n1 = NL * 2 * C * C * 3, n2 = NL * (C + S) * C;
    const size_t n1w = (C == 256 && S == 256) ? NL * 4 * 2 * C * C : 0;    // F(2,3)-transformed GEMM1 image (ap_resblock_f32s2.hip)
    if (!ctx->slab_s) {
      AP_HIP(hipMalloc(&ctx->slab_s, (n1 + n2 + n1w) * 3 * 2));
      ctx->w1p_s = ctx->slab_s;
      ctx->w2p_s = (char *)ctx->slab_s + n1 * 3 * 2;
      ctx->w1w_s = n1w ? (char *)ctx->slab_s + (n1 + n2) * 3 * 2 : nullptr;
    }
    rc = launch_pack_split(ctx, st);
    if (rc) return rc;
    if (ctx->w1w_s) {
      rc = launch_pack_split23(ctx, st);
      if (rc) return rc;
    }
  }
  AP_HIP(hipStreamSynchronize(st));
  ctx->loaded = true;
  return 0;
}

extern "C" int ap_ctx_prepare_backward(ap_ctx *ctx, void *stream) {
  if (!ctx || !ctx->loaded) { set_error("ap_ctx_prepare_backward: weights not loaded (ap_ctx_load_wavenet)"); return -22; }
  if (ctx->bwd_ready) return 0;
  if (ctx->C != 256 || ctx->S != 256) { set_error("ap_ctx_prepare_backward: the fused backward kernels are built for res = skip = 256 channels"); return -22; }
  if (ctx->cfg.precision == AP_PREC_F32) return prepare_bwd_f32(ctx, (hipStream_t)stream);
  if (ctx->cfg.precision == AP_PREC_BF16 || ctx->cfg.precision == AP_PREC_BF16_STORE) return prepare_bwd_bf16(ctx, (hipStream_t)stream);
  set_error("ap_ctx_prepare_backward: no fused backward in precision %d (AP_PREC_F32, AP_PREC_BF16, AP_PREC_BF16_STORE)", ctx->cfg.precision);
  return -22;
}

extern "C" int ap_ctx_get_folded(ap_ctx *ctx, int which, int layer, float *out_dev, size_t n_elems, void *stream) {
  if (!ctx || !ctx->loaded || !out_dev) { set_error("get_folded: context not loaded / null"); return -22; }
  const size_t C = ctx->C, S = ctx->S;
  const float *src = nullptr;
  size_t n = 0;
  if (which != 3 && which != 4 && (layer < 0 || layer >= ctx->NL)) { set_error("get_folded: layer %d", layer); return -22; }
  switch (which) {
    case 0: src = ctx->w1f + (size_t)layer * 2 * C * C * 3; n = 2 * C * C * 3; break;
    case 1: src = ctx->w2f + (size_t)layer * (C + S) * C; n = C * C; break;
    case 2: src = ctx->w2f + (size_t)layer * (C + S) * C + C * C; n = S * C; break;
    case 3: src = ctx->wf1f; n = S * S; break;
    case 4: src = ctx->w0; n = C; break;
    default: set_error("get_folded: which=%d", which); return -22;
  }
  if (n != n_elems) { set_error("get_folded: expected %zu elements, got %zu", n, n_elems); return -22; }
  AP_HIP(hipMemcpyAsync(out_dev, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// workspace: h_a [B C L] | h_b [B C L] | skip [B S L] | x_a [B L] | x_b [B L] | part_t [NL C + Eout]
//            | AP_PREC_BF16 with a skip group G > 0: g images [G][B][L][C] bf16 (the deferred-skip form: ap_skipgemm_bf16.hip)
// AP_PREC_BF16_STORE: h_a, h_b are the bf16 u images [B][C/32][L][32] (half the bytes); always the deferred-skip form (G = 0 means
//            one group of all layers)
//            | tools builds, AP_PREC_BF16: ub_a, ub_b [B][C/32][L][32] bf16 (the operand-image experiment: ap_resblock_bf16p.hip, UB)
#ifdef AP_TOOLS
namespace ap { extern int g_dbg_bf16; }
#endif
namespace {
struct Ws {
  float *ha, *hb, *skip, *xa, *xb, *pt;
  void *uba, *ubb;
  char *gimg;          // [G][B][L][C] bf16, or null
  size_t gslot;        // bytes per slot
  size_t bytes;
};
inline size_t al(size_t n) { return (n + 63) & ~(size_t)63; }
inline int skip_group_of(const ap_ctx *ctx) {                    // layers per skip GEMM; 0: the fused block (skip per layer)
  if (ctx->cfg.precision == AP_PREC_BF16_STORE) return ctx->skip_group > 0 ? ctx->skip_group : ctx->NL;
  return ctx->cfg.precision == AP_PREC_BF16 ? ctx->skip_group : 0;
}
Ws carve(const ap_ctx *ctx, void *base, int B, int L) {
  Ws w;
  const bool bstore = ctx->cfg.precision == AP_PREC_BF16_STORE;
  size_t act = al(bstore ? ((size_t)B * ctx->C * L + 1) / 2 : (size_t)B * ctx->C * L), sk = al((size_t)B * ctx->S * L), xl = al((size_t)B * L),
         pt = al((size_t)ctx->NL * ctx->C + ctx->cfg.embed_dim_out);
  float *p = (float *)base;
  w.ha = p; p += act;
  w.hb = p; p += act;
  w.skip = p; p += sk;
  w.xa = p; p += xl;
  w.xb = p; p += xl;
  w.pt = p; p += pt;
  w.uba = w.ubb = nullptr;
  w.gimg = nullptr;
  w.gslot = 0;
  if (skip_group_of(ctx) > 0) {
    w.gslot = (size_t)B * L * ctx->C * 2;                       // (a multiple of 512 bytes)
    w.gimg = (char *)p;
    p += al(((size_t)skip_group_of(ctx) * w.gslot + 3) / 4);
  }
#ifdef AP_TOOLS
  if (ctx->cfg.precision == AP_PREC_BF16) {
    const size_t ub = al(((size_t)B * ctx->C * L + 1) / 2);
    w.uba = p; p += ub;
    w.ubb = p; p += ub;
  }
#endif
  w.bytes = (size_t)((char *)p - (char *)base);
  return w;
}
int check_run(ap_ctx *ctx, int B, int L, void *ws, size_t ws_bytes, const char *who) {
  if (!ctx || !ctx->loaded) { set_error("%s: weights not loaded (ap_ctx_load_wavenet)", who); return -22; }
  if (B < 1 || L < 1) { set_error("%s: B=%d L=%d", who, B, L); return -22; }
  if ((size_t)B * ctx->C * (size_t)L >= ((size_t)1 << 40)) { set_error("%s: batch too large", who); return -22; }
  size_t need = ap_workspace_bytes(ctx, B, L);
  if (!ws || ws_bytes < need) { set_error("%s: workspace %zu bytes < required %zu", who, ws_bytes, need); return -22; }
  if (((uintptr_t)ws & 15) != 0) { set_error("%s: workspace must be 16-byte aligned", who); return -22; }
  return 0;
}

// the 36-layer sweep: h ping-pongs between ha/hb, skip accumulates (WaveNet.py:120-135,164-170)
// one AP_PREC_BF16_STORE block launch (its own pair of profile events: kind 0)
int launch_resblock_u_timed(ap_ctx *ctx, int layer, const void *uin, const float *pt_next, void *uout, void *gout, int B, int L, hipStream_t st,
                            void *fout = nullptr) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profile) {
    if (ctx->ev_used + 2 > ctx->ev.size())
      for (int i = 0; i < 2; i++) {
        hipEvent_t e;
        AP_HIP(hipEventCreate(&e));
        ctx->ev.push_back(e);
      }
    e0 = ctx->ev[ctx->ev_used];
    e1 = ctx->ev[ctx->ev_used + 1];
    if (ctx->ev_kind.size() < ctx->ev.size() / 2) ctx->ev_kind.resize(ctx->ev.size() / 2, 0);
    ctx->ev_kind[ctx->ev_used / 2] = 0;
    ctx->ev_used += 2;
    AP_HIP(hipEventRecord(e0, st));
  }
  const int rc = launch_resblock_bf16u(ctx, layer, uin, pt_next, uout, gout, B, L, st, fout);
  if (e1) AP_HIP(hipEventRecord(e1, st));
  return rc;
}

// one skip GEMM (its own pair of profile events: kind 1)
int skip_gemm_timed(ap_ctx *ctx, int n0, int nl, const Ws &w, int B, int L, hipStream_t st) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profile) {
    if (ctx->ev_used + 2 > ctx->ev.size())
      for (int i = 0; i < 2; i++) {
        hipEvent_t e;
        AP_HIP(hipEventCreate(&e));
        ctx->ev.push_back(e);
      }
    e0 = ctx->ev[ctx->ev_used];
    e1 = ctx->ev[ctx->ev_used + 1];
    if (ctx->ev_kind.size() < ctx->ev.size() / 2) ctx->ev_kind.resize(ctx->ev.size() / 2, 0);
    ctx->ev_kind[ctx->ev_used / 2] = 1;
    ctx->ev_used += 2;
    AP_HIP(hipEventRecord(e0, st));
  }
  const int rc = launch_skipgemm_bf16(ctx, n0, nl, w.gimg, w.skip, n0 > 0, B, L, st);
  if (e1) AP_HIP(hipEventRecord(e1, st));
  return rc;
}

// AP_PREC_BF16_STORE: the sweep over bf16 u images (ap_resblock_bf16u.hip), skip through the deferred-skip GEMM
int run_net_bstore(ap_ctx *ctx, const float *x, const Ws &w, int B, int L, hipStream_t st) {
  int rc = launch_init_conv_u(ctx, x, w.pt, w.ha, B, L, st);
  if (rc) return rc;
  void *uin = w.ha, *uout = w.hb;
  const int G = skip_group_of(ctx);
  for (int n0 = 0; n0 < ctx->NL; n0 += G) {
    const int nl = ctx->NL - n0 < G ? ctx->NL - n0 : G;
    for (int n = n0; n < n0 + nl; n++) {
      const bool last = n + 1 == ctx->NL;
      rc = launch_resblock_u_timed(ctx, n, uin, last ? nullptr : w.pt + (size_t)(n + 1) * ctx->C, last ? nullptr : uout,
                                   w.gimg + (size_t)(n - n0) * w.gslot, B, L, st);
      if (rc) return rc;
      void *t = uin; uin = uout; uout = t;
    }
    rc = skip_gemm_timed(ctx, n0, nl, w, B, L, st);
    if (rc) return rc;
  }
  return 0;
}

int run_net(ap_ctx *ctx, const float *x, float step, const Ws &w, int B, int L, hipStream_t st) {
  int rc = launch_embed(ctx, step, w.pt, st);
  if (rc) return rc;
  if (ctx->cfg.precision == AP_PREC_BF16_STORE) return run_net_bstore(ctx, x, w, B, L, st);
  rc = launch_init_conv(ctx, x, w.ha, B, L, st);
  if (rc) return rc;
  float *hin = w.ha, *hout = w.hb;
#ifdef AP_TOOLS
  // tools bit 0x400000 (tools/ab_bf16_ub.py): each layer hands the next one its bf16 operand image -- bit-identical, 3 % slower
  // (the block runs at the board's power cap; the image's extra stores cost more than its staging saves: DESIGN.md 3.4)
  if (w.uba != nullptr && (ap::g_dbg_bf16 & 0x400000)) {
    rc = launch_make_ub(w.ha, w.pt, w.uba, B, ctx->C, L, st);
    if (rc) return rc;
    void *uin = w.uba, *uout = w.ubb;
    for (int n = 0; n < ctx->NL; n++) {
      const bool last = n + 1 == ctx->NL;
      const UbArgs ub = {uin, last ? nullptr : uout, w.pt + (size_t)(last ? n : n + 1) * ctx->C};
      rc = launch_resblock(ctx, n, hin, w.pt + (size_t)n * ctx->C, hout, w.skip, n > 0, B, L, st, nullptr, &ub);
      if (rc) return rc;
      float *t = hin; hin = hout; hout = t;
      void *u = uin; uin = uout; uout = u;
    }
    return 0;
  }
#endif
  if (w.gimg) {
    // deferred-skip form (AP_PREC_BF16): every block writes h' and its bf16 g image; after each group of G layers one GEMM adds
    // the group's skip_conv outputs into skip (a plain store for the first group)
    const int G = ctx->skip_group;
    for (int n0 = 0; n0 < ctx->NL; n0 += G) {
      const int nl = ctx->NL - n0 < G ? ctx->NL - n0 : G;
      for (int n = n0; n < n0 + nl; n++) {
        // (the last layer's h' is never read -- WaveNet.py:131-135 returns the skip sum only: null = leave res_conv and its store out)
        rc = launch_resblock(ctx, n, hin, w.pt + (size_t)n * ctx->C, n + 1 == ctx->NL ? nullptr : hout, nullptr, 0, B, L, st, nullptr, nullptr,
                             w.gimg + (size_t)(n - n0) * w.gslot);
        if (rc) return rc;
        float *t = hin; hin = hout; hout = t;
      }
      rc = skip_gemm_timed(ctx, n0, nl, w, B, L, st);
      if (rc) return rc;
    }
    return 0;
  }
  // (the last layer's h' is never read -- WaveNet.py:131-135 returns the skip sum only; the minimal-filtering block leaves
  // res_conv and its store out when handed a null h')
  const bool drop_last = resblock_f32w_serves(ctx, B, L);
  for (int n = 0; n < ctx->NL; n++) {
    rc = launch_resblock(ctx, n, hin, w.pt + (size_t)n * ctx->C, (drop_last && n + 1 == ctx->NL) ? nullptr : hout, w.skip, n > 0, B, L, st);
    if (rc) return rc;
    float *t = hin; hin = hout; hout = t;
  }
  return 0;
}
}  // namespace

extern "C" size_t ap_workspace_bytes(const ap_ctx *ctx, int B, int L) {
  if (!ctx || B < 1 || L < 1) return 0;
  return carve(ctx, nullptr, B, L).bytes;
}

extern "C" int ap_embed(ap_ctx *ctx, float step, float *part_t_dev, void *stream) {
  if (!ctx || !ctx->loaded || !part_t_dev) { set_error("ap_embed: not loaded / null"); return -22; }
  return launch_embed(ctx, step, part_t_dev, (hipStream_t)stream);
}

extern "C" int ap_init_conv(ap_ctx *ctx, const float *x, float *h, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !x || !h || B < 1 || L < 1) { set_error("ap_init_conv: bad argument"); return -22; }
  return launch_init_conv(ctx, x, h, B, L, (hipStream_t)stream);
}

extern "C" int ap_init_conv_u(ap_ctx *ctx, const float *x, const float *part_t_layer0, void *u_out, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !x || !part_t_layer0 || !u_out || B < 1 || L < 1) { set_error("ap_init_conv_u: bad argument"); return -22; }
  if (!resblock_bf16u_serves(ctx, L)) { set_error("ap_init_conv_u: AP_PREC_BF16_STORE contexts with res = skip = 256 channels only"); return -22; }
  return launch_init_conv_u(ctx, x, part_t_layer0, u_out, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_fwd_u(ap_ctx *ctx, int layer, const void *u_in, const float *part_t_next, void *u_out, void *g_image, int B, int L,
                                 void *stream) {
  if (!ctx || !ctx->loaded || !u_in || !g_image) { set_error("ap_resblock_fwd_u: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { set_error("ap_resblock_fwd_u: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (u_out && !part_t_next) { set_error("ap_resblock_fwd_u: u_out needs the next layer's part_t"); return -22; }
  if (u_in == u_out) { set_error("ap_resblock_fwd_u: u_out must not alias u_in"); return -22; }
  return launch_resblock_u_timed(ctx, layer, u_in, part_t_next, u_out, g_image, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_fwd(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out,
                               float *skip, int accumulate_skip, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !h_in || !part_t_layer || !h_out || !skip) { set_error("ap_resblock_fwd: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { set_error("ap_resblock_fwd: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (h_in == h_out) { set_error("ap_resblock_fwd: h_out must not alias h_in"); return -22; }
  return launch_resblock(ctx, layer, h_in, part_t_layer, h_out, skip, accumulate_skip, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_fwd_gate(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out,
                                    void *g_image, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !h_in || !part_t_layer || !g_image) { set_error("ap_resblock_fwd_gate: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { set_error("ap_resblock_fwd_gate: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (h_in == h_out) { set_error("ap_resblock_fwd_gate: h_out must not alias h_in"); return -22; }
  if (ctx->cfg.precision != AP_PREC_BF16) { set_error("ap_resblock_fwd_gate: AP_PREC_BF16 only"); return -22; }
  return launch_resblock(ctx, layer, h_in, part_t_layer, h_out, nullptr, 0, B, L, (hipStream_t)stream, nullptr, nullptr, g_image);
}

extern "C" int ap_resblock_fwd_u_save(ap_ctx *ctx, int layer, const void *u_in, const float *part_t_next, void *u_out, void *g_image,
                                      void *gate_factors, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !u_in || !g_image || !gate_factors) { set_error("ap_resblock_fwd_u_save: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { set_error("ap_resblock_fwd_u_save: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (u_out && !part_t_next) { set_error("ap_resblock_fwd_u_save: u_out needs the next layer's part_t"); return -22; }
  if (u_in == u_out) { set_error("ap_resblock_fwd_u_save: u_out must not alias u_in"); return -22; }
  return launch_resblock_u_timed(ctx, layer, u_in, part_t_next, u_out, g_image, B, L, (hipStream_t)stream, gate_factors);
}

extern "C" size_t ap_gate_factor_bytes(int B, int L) {
  if (B < 1 || L < 1) return 0;
  return (size_t)B * (size_t)((L + 127) / 128) * 131072u;
}

extern "C" int ap_resblock_fwd_gate_save(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out,
                                         void *g_image, void *gate_factors, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !h_in || !part_t_layer || !g_image || !gate_factors) { set_error("ap_resblock_fwd_gate_save: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { set_error("ap_resblock_fwd_gate_save: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (h_in == h_out) { set_error("ap_resblock_fwd_gate_save: h_out must not alias h_in"); return -22; }
  if (ctx->cfg.precision != AP_PREC_BF16) { set_error("ap_resblock_fwd_gate_save: AP_PREC_BF16 only"); return -22; }
  return launch_resblock(ctx, layer, h_in, part_t_layer, h_out, nullptr, 0, B, L, (hipStream_t)stream, nullptr, nullptr, g_image, gate_factors);
}

extern "C" int ap_skip_gemm(ap_ctx *ctx, int layer0, int n_layers, const void *g_images, float *skip, int accumulate_skip,
                            int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !g_images || !skip) { set_error("ap_skip_gemm: not loaded / null"); return -22; }
  if (ctx->cfg.precision != AP_PREC_BF16 && ctx->cfg.precision != AP_PREC_BF16_STORE) { set_error("ap_skip_gemm: AP_PREC_BF16 / AP_PREC_BF16_STORE only"); return -22; }
  return launch_skipgemm_bf16(ctx, layer0, n_layers, g_images, skip, accumulate_skip, B, L, (hipStream_t)stream);
}

extern "C" int ap_resblock_fwd_save(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out,
                                    float *skip, float *pre_gate, int accumulate_skip, int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !h_in || !part_t_layer || !h_out || !skip || !pre_gate) { set_error("ap_resblock_fwd_save: not loaded / null"); return -22; }
  if (layer < 0 || layer >= ctx->NL || B < 1 || L < 1) { set_error("ap_resblock_fwd_save: layer=%d B=%d L=%d", layer, B, L); return -22; }
  if (h_in == h_out) { set_error("ap_resblock_fwd_save: h_out must not alias h_in"); return -22; }
  return launch_resblock(ctx, layer, h_in, part_t_layer, h_out, skip, accumulate_skip, B, L, (hipStream_t)stream, pre_gate);
}

extern "C" int ap_final_affine(ap_ctx *ctx, const float *skip, const float *x, float *eps_out, float *out, float ca,
                               float cb, float cs, const float *z, uint64_t seed, uint32_t draw, uint64_t utt_offset,
                               int B, int L, void *stream) {
  if (!ctx || !ctx->loaded || !skip) { set_error("ap_final_affine: not loaded / null"); return -22; }
  if (out && !x) { set_error("ap_final_affine: out needs x"); return -22; }
  if (B < 1 || L < 1) { set_error("ap_final_affine: B=%d L=%d", B, L); return -22; }
  return launch_final_affine(ctx, skip, x, eps_out, out, ca, cb, cs, z, seed, draw, utt_offset, B, L,
                             (hipStream_t)stream);
}

extern "C" int ap_affine_noise(const float *x, float *out, float ca, float cs, const float *z, uint64_t seed,
                               uint32_t draw, uint64_t utt_offset, int B, int L, void *stream) {
  if (!out || B < 1 || L < 1) { set_error("ap_affine_noise: bad argument"); return -22; }
  if (!x && ca != 0.f) { set_error("ap_affine_noise: x is NULL but ca != 0"); return -22; }
  return launch_affine_noise(x, out, ca, cs, z, seed, draw, utt_offset, B, L, (hipStream_t)stream);
}

extern "C" int ap_eps_fwd(ap_ctx *ctx, const float *x, float step, float *eps_out, int B, int L, void *workspace,
                          size_t ws_bytes, void *stream) {
  int rc = check_run(ctx, B, L, workspace, ws_bytes, "ap_eps_fwd");
  if (rc) return rc;
  if (!x || !eps_out) { set_error("ap_eps_fwd: null tensor"); return -22; }
  hipStream_t st = (hipStream_t)stream;
  Ws w = carve(ctx, workspace, B, L);
  rc = run_net(ctx, x, step, w, B, L, st);
  if (rc) return rc;
  return launch_final_affine(ctx, w.skip, nullptr, eps_out, nullptr, 0.f, 0.f, 0.f, nullptr, 0, 0, 0, B, L, st);
}

extern "C" int ap_eps_affine(ap_ctx *ctx, const float *x, float step, float ca, float cb, float *eps_out, float *out,
                             int B, int L, void *workspace, size_t ws_bytes, void *stream) {
  int rc = check_run(ctx, B, L, workspace, ws_bytes, "ap_eps_affine");
  if (rc) return rc;
  if (!x || (!eps_out && !out)) { set_error("ap_eps_affine: null tensor"); return -22; }
  hipStream_t st = (hipStream_t)stream;
  Ws w = carve(ctx, workspace, B, L);
  rc = run_net(ctx, x, step, w, B, L, st);
  if (rc) return rc;
  return launch_final_affine(ctx, w.skip, x, eps_out, out, ca, cb, 0.f, nullptr, 0, 0, 0, B, L, st);
}

extern "C" int ap_purify_chain(ap_ctx *ctx, const float *x0, float qa, float qs, const ap_step *steps, int n_steps,
                               const float *z_all, uint64_t seed, uint64_t utt_offset, float *x_out, int B, int L,
                               void *workspace, size_t ws_bytes, void *stream) {
  int rc = check_run(ctx, B, L, workspace, ws_bytes, "ap_purify_chain");
  if (rc) return rc;
  if (!x0 || !x_out || (n_steps > 0 && !steps) || n_steps < 0) { set_error("ap_purify_chain: bad argument"); return -22; }
  hipStream_t st = (hipStream_t)stream;
  Ws w = carve(ctx, workspace, B, L);
  const size_t BL = (size_t)B * L;
  const float *cur = x0;
  if (!(qa == 1.0f && qs == 0.0f)) {
    float *dst = (n_steps == 0) ? x_out : w.xa;
    rc = launch_affine_noise(x0, dst, qa, qs, z_all, seed, 0, utt_offset, B, L, st);
    if (rc) return rc;
    cur = dst;
  } else if (n_steps == 0) {
    AP_HIP(hipMemcpyAsync(x_out, x0, BL * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  for (int i = 0; i < n_steps; i++) {
    const ap_step &s = steps[i];
    rc = run_net(ctx, cur, s.step, w, B, L, st);
    if (rc) return rc;
    float *dst = (i == n_steps - 1) ? x_out : ((cur == w.xa) ? w.xb : w.xa);
    const float *z = (z_all && s.cs != 0.f) ? z_all + (size_t)s.draw * BL : nullptr;
    rc = launch_final_affine(ctx, w.skip, cur, nullptr, dst, s.ca, s.cb, s.cs, z, seed, (uint32_t)s.draw, utt_offset,
                             B, L, st);
    if (rc) return rc;
    cur = dst;
  }
  return 0;
}

extern "C" int ap_purify_ddpm(ap_ctx *ctx, const float *x0, int t_star, int do_diffuse, const float *z_all,
                              uint64_t seed, uint64_t utt_offset, float *x_out, int B, int L, void *workspace,
                              size_t ws_bytes, void *stream) {
  if (!ctx) { set_error("ap_purify_ddpm: null ctx"); return -22; }
  const int T = (int)ctx->Alpha.size();
  if (t_star < 1 || t_star > T) { set_error("ap_purify_ddpm: reverse_timestep %d outside [1, %d]", t_star, T); return -22; }
  // diffwave_ddpm.py:66-67
  const double ab = ctx->Alpha_bar[t_star - 1];
  float qa = 1.f, qs = 0.f;
  if (do_diffuse) { qa = (float)sqrt(ab); qs = (float)sqrt(1.0 - ab); }
  std::vector<ap_step> steps(t_star);
  int draw = 1;
  for (int i = 0; i < t_star; i++) {
    const int t = t_star - 1 - i;                      // diffwave_ddpm.py:95
    const double a = ctx->Alpha[t], abt = ctx->Alpha_bar[t];
    ap_step s;
    s.step = (float)t;                                 // :157
    s.ca = (float)(1.0 / sqrt(a));                     // :159  mu = (x - (1-a)/sqrt(1-ab) eps)/sqrt(a)
    s.cb = (float)(-(1.0 - a) / sqrt(1.0 - abt) / sqrt(a));
    s.cs = (t > 0) ? ctx->Sigma[t] : 0.f;              // :99-102,160
    s.draw = (t > 0) ? draw++ : 0;
    steps[i] = s;
  }
  return ap_purify_chain(ctx, x0, qa, qs, steps.data(), t_star, z_all, seed, utt_offset, x_out, B, L, workspace,
                         ws_bytes, stream);
}

extern "C" int ap_purify_sde(ap_ctx *ctx, const float *x0, int t_star, const float *z_all, uint64_t seed,
                             uint64_t utt_offset, float *x_out, int B, int L, void *workspace, size_t ws_bytes,
                             void *stream) {
  if (!ctx) { set_error("ap_purify_sde: null ctx"); return -22; }
  const int N = (int)ctx->sde_beta.size();
  if (t_star < 1 || t_star > N) { set_error("ap_purify_sde: t %d outside [1, %d]", t_star, N); return -22; }
  // diffwave_sde.py:189-190: a = cumprod(1 - betas); x = x0 sqrt(a[t-1]) + e sqrt(1 - a[t-1])
  const double a_last = ctx->sde_ac[t_star - 1];
  const float qa = (float)sqrt(a_last), qs = (float)sqrt(1.0 - a_last);
  std::vector<ap_step> steps(t_star);
  for (int i = 0; i < t_star; i++) {
    const int k = t_star - 1 - i;
    const double beta = ctx->sde_beta[k];              // discrete_betas[k]; beta_t = beta*N, dt = 1/N (:77, :203)
    const double ack = ctx->sde_ac[k];
    ap_step s;
    s.step = (float)k;                                 // compute_eps_t(x, disc_steps[0]) (:95)
    s.ca = (float)(1.0 + 0.5 * beta);                  // y + (0.5 beta N y) / N
    s.cb = (float)(-beta / sqrt(1.0 - ack));           // - beta N (eps / sqrt(1-ac)) / N   (:99,:104,:125)
    s.cs = (k > 0) ? (float)(sqrt(beta) * sqrt(1.0 - (double)ctx->sde_ac[k - 1]) / sqrt(1.0 - ack)) : 0.f;   // :109-115
    s.draw = 1 + i;
    steps[i] = s;
  }
  return ap_purify_chain(ctx, x0, qa, qs, steps.data(), t_star, z_all, seed, utt_offset, x_out, B, L, workspace,
                         ws_bytes, stream);
}

extern "C" int ap_one_shot_denoise(ap_ctx *ctx, const float *x_t, int t_star, float *x0_hat, int B, int L,
                                   void *workspace, size_t ws_bytes, void *stream) {
  if (!ctx) { set_error("ap_one_shot_denoise: null ctx"); return -22; }
  const int T = (int)ctx->Alpha_bar.size();
  if (t_star < 1 || t_star > T) { set_error("ap_one_shot_denoise: reverse_timestep %d outside [1, %d]", t_star, T); return -22; }
  const int t = t_star - 1;                            // diffwave_ddpm.py:176
  const double ab = ctx->Alpha_bar[t];
  ap_step s;
  s.step = (float)t;
  s.ca = (float)sqrt(1.0 / ab);                        // :200  sqrt_recip_alphas_bar
  s.cb = (float)(-sqrt(1.0 / ab - 1.0));               // :201  sqrt_recipm1_alphas_bar
  s.cs = 0.f;
  s.draw = 0;
  return ap_purify_chain(ctx, x_t, 1.f, 0.f, &s, 1, nullptr, 0, 0, x0_hat, B, L, workspace, ws_bytes, stream);
}

// ---- M5 ----------------------------------------------------------------------------------------
extern "C" size_t ap_m5_blob_elems(int n_output, int n_channel, int first_kernel) {
  size_t n = 0;
  int ci[4] = {1, n_channel, n_channel, 2 * n_channel};
  int co[4] = {n_channel, n_channel, 2 * n_channel, 2 * n_channel};
  int k[4] = {first_kernel, 3, 3, 3};
  for (int i = 0; i < 4; i++) n += (size_t)co[i] * ci[i] * k[i] + 5 * (size_t)co[i];
  n += (size_t)n_output * 2 * n_channel + n_output;
  return n;
}

extern "C" int ap_m5_create(int n_output, int n_channel, int first_kernel, int stride, float bn_eps,
                            const float *blob_dev, size_t n_elems, void *stream, ap_m5 **out) {
  if (!blob_dev || !out) { set_error("ap_m5_create: null argument"); return -22; }
  if (n_output < 1 || n_output > 64 || n_channel < 1 || n_channel > 64 || first_kernel < 1 || first_kernel > 256 ||
      stride < 1) {
    set_error("ap_m5_create: unsupported shape (n_output<=64, n_channel<=64, first_kernel<=256)");
    return -22;
  }
  if (n_elems != ap_m5_blob_elems(n_output, n_channel, first_kernel)) {
    set_error("ap_m5_create: blob has %zu elements, expected %zu", n_elems,
              ap_m5_blob_elems(n_output, n_channel, first_kernel));
    return -22;
  }
  ap_m5 *m = new (std::nothrow) ap_m5();
  if (!m) { set_error("out of host memory"); return -12; }
  m->n_output = n_output; m->n_channel = n_channel; m->k1 = first_kernel; m->stride = stride;
  int ci[4] = {1, n_channel, n_channel, 2 * n_channel};
  int co[4] = {n_channel, n_channel, 2 * n_channel, 2 * n_channel};
  int k[4] = {first_kernel, 3, 3, 3};
  size_t n = 0;
  size_t ow[4], ob[4];
  for (int i = 0; i < 4; i++) { ow[i] = n; n += al((size_t)co[i] * ci[i] * k[i]); ob[i] = n; n += al(co[i]); }
  size_t ofw = n; n += al((size_t)n_output * 2 * n_channel);
  size_t ofb = n; n += al(n_output);
  hipError_t e = hipMalloc((void **)&m->slab, n * sizeof(float));
  if (e != hipSuccess) { delete m; return hip_fail(e, "hipMalloc(m5)"); }
  for (int i = 0; i < 4; i++) { m->w[i] = m->slab + ow[i]; m->b[i] = m->slab + ob[i]; }
  m->fcw = m->slab + ofw; m->fcb = m->slab + ofb;
  int rc = launch_m5_fold(m, blob_dev, bn_eps, (hipStream_t)stream);
  if (rc) { (void)hipFree(m->slab); delete m; return rc; }
  e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) { (void)hipFree(m->slab); delete m; return hip_fail(e, "sync(m5)"); }
  *out = m;
  return 0;
}

extern "C" int ap_m5_destroy(ap_m5 *m) {
  if (!m) return 0;
  if (m->slab) (void)hipFree(m->slab);
  delete m;
  return 0;
}

extern "C" int ap_m5_bwd(ap_m5 *m, const float *x, const float *dlogprobs, float *dx, int B, int L, void *stream) {
  if (!m || !x || !dlogprobs || !dx || B < 1) { set_error("ap_m5_bwd: bad argument"); return -22; }
  return launch_m5_bwd(m, x, dlogprobs, dx, B, L, (hipStream_t)stream);
}

extern "C" int ap_m5_fwd(ap_m5 *m, const float *x, float *logprobs, int B, int L, void *stream) {
  if (!m || !x || !logprobs || B < 1) { set_error("ap_m5_fwd: bad argument"); return -22; }
  return launch_m5(m, x, logprobs, B, L, (hipStream_t)stream);
}
