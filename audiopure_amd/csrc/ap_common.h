// Shared declarations for the gfx950 kernels and the C-ABI (include/audiopure.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../include/audiopure.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ap {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define AP_HIP(x)                                  \
  do {                                             \
    hipError_t _e = (x);                           \
    if (_e != hipSuccess) return ap::hip_fail(_e, #x); \
  } while (0)

// CU count of the calling thread's current HIP device (the persistent kernels launch one workgroup per CU).  One table entry per
// device: a process may drive several devices, and their CU counts / partition modes may differ.  A racing first call writes the
// same value twice.
inline int device_cu_count() {
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (n_cu_of[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    n_cu_of[dev] = n;
  }
  return n_cu_of[dev];
}

// Timing-only hooks (ablation masks, phase stamps, dispatch overrides: tools/*.py) exist only in a -DAP_TOOLS build
// (`python __graft_entry__.py --tools` -> tools/lib/libaudiopure_hip_tools.so).  The shipped library has no `ablate` kernel
// argument, no ap_debug_* symbol and no stamped instantiation.
#ifdef AP_TOOLS
#define AP_ABLATE_PARAM , int ablate
#define AP_ABLATE_ARG(x) , (x)
#define AP_ABLATE_DECL
#define AP_TOOLS_VAR static int
#else
#define AP_ABLATE_PARAM
#define AP_ABLATE_ARG(x)
#define AP_ABLATE_DECL constexpr int ablate = 0;
#define AP_TOOLS_VAR static constexpr int
#endif

constexpr int TT = 128;   // time-tile (samples) of the fused kernels
constexpr int KC = 32;    // channels per staged K-chunk of the dilated conv (x3 taps = 96 K rows)

// Offsets (in floats) of the reference state-dict tensors inside the weight blob.
struct BlobLayout {
  size_t init_b, init_g, init_v;
  size_t fc1_w, fc1_b, fc2_w, fc2_b;
  size_t blk0, blk_stride;
  // inside a block
  size_t fct_w, fct_b, dil_b, dil_g, dil_v, res_b, res_g, res_v, skip_b, skip_g, skip_v;
  size_t f1_b, f1_g, f1_v, f2_w, f2_b;
  size_t total;
};
BlobLayout blob_layout(const ap_config &c);

}  // namespace ap

struct ap_ctx {
  ap_config cfg;
  int C, S, NL, NW;
  bool loaded;
  // schedule (host), reference: util.py:96-123 and diffwave_sde.py:56-60
  std::vector<float> Beta, Alpha, Alpha_bar, Sigma;
  std::vector<float> sde_beta, sde_ac;
  // device weights
  float *slab;            // one allocation holding everything below
  size_t slab_elems;
  float *emb_freq;        // [Ein/2]
  float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  float *fct_w, *fct_b;   // [NL][C][Eout], [NL][C]
  float *w0, *b0;         // folded init conv [C], bias [C]
  float *w1f;             // [NL][2C][C][3] folded
  float *w2f;             // [NL][C+S][C] folded (res rows then skip rows)
  float *wf1f;            // [S][S] folded
  float *b1;              // [NL][2C]
  float *b2;              // [NL][C+S]
  float *bf1, *wf2, *bf2; // [S], [S], [1]
  float *w1p, *w2p, *wf1p;  // packed fp32 MFMA A-operand images
  void *w1p_bf, *w2p_bf;    // packed bf16 images (AP_PREC_BF16), own allocation
  void *wf1p_bf;            // final conv's first 1x1 as a bf16 image (same allocation)
  void *w1q_bf;             // GEMM1 image for v_mfma_f32_16x16x32_bf16 (same allocation, behind wf1p_bf)
  void *slab_bf;
  void *w1p_s, *w2p_s;      // 3-way bf16-split images (AP_PREC_F32_SPLIT), own allocation
  void *w1w_s;              // ... and the F(2,3)-transformed, 3-way-split GEMM1 image of ap_resblock_f32s2.hip (same allocation)
  void *slab_s;
  float *w1w, *w2w;         // AP_PREC_F32, C = S = 256: F(2,3)-transformed GEMM1 image and GEMM2 image of ap_resblock_f32w.hip, own allocation
  void *slab_w;
  float *w2t, *w1b;         // backward images of ap_resblock_bwd.hip (ap_ctx_prepare_backward), own allocation
  void *slab_b;
  void *slab_bb;            // bf16 backward images of ap_resblock_bwd_bf16.hip (ap_ctx_prepare_backward)
  bool bwd_ready;           // the backward images of this context's precision are built from the weights now loaded
  int f32_form;             // AP_PREC_F32 / AP_PREC_F32_SPLIT: 1 = minimal-filtering (F(2,3)) block where built (default), 0 = direct-form block
  float *norms;           // scratch for row norms
  // optional per-launch timing of the residual-block kernel (bench.py roofline leg)
  bool profile;
  std::vector<hipEvent_t> ev;   // pairs (start, stop), one pair per launch since the last reset
  std::vector<char> ev_kind;    // per pair: 0 residual-block launch, 1 skip-GEMM launch of the deferred-skip form
  size_t ev_used;
  int skip_group;               // AP_PREC_BF16: layers per skip GEMM of the deferred-skip form (0: fused block, skip per layer)
};

struct ap_m5 {
  int n_output, n_channel, k1, stride;
  float *slab;
  float *w[4], *b[4];
  float *fcw, *fcb;
};

// kernel launchers (defined in the .hip files)
#ifdef __HIPCC__
namespace ap {
// row of accumulator register r of a 32x32 MFMA tile held by lane half hh
__device__ __forceinline__ int rowoff(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// exp(x) on the hardware exp2 with a compensated argument: ~2 ulp over the range the gate uses.
__device__ __forceinline__ float exp_acc(float x) {
  const float L2E_HI = 1.44269502162933349609375f;   // float(log2 e)
  const float L2E_LO = 1.92596299e-8f;               // log2 e - L2E_HI
  float t = x * L2E_HI;
  float r = __builtin_fmaf(x, L2E_HI, -t);
  r = __builtin_fmaf(x, L2E_LO, r);
  float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, r * 0.693147182464599609375f, e);
}

// tanh(a) * sigmoid(b) = (E - 1) / ((E + 1) (1 + F)),  E = e^{2a}, F = e^{-b}   (WaveNet.py:90)
__device__ __forceinline__ float gate(float a, float b) {
  a = fminf(fmaxf(a, -15.0f), 15.0f);    // tanh(+-15) == +-1 in fp32
  b = fmaxf(b, -80.0f);                  // keep F finite: sigmoid(-80) ~ 1.8e-35
  float E = exp_acc(2.0f * a);
  float F = exp_acc(-b);
  return (E - 1.0f) * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
}

// Which (clip, tile) a workgroup of a one-tile-per-workgroup block kernel takes.  Placement only -- any bijection gives the same
// results; this one is for the per-XCD L2s: workgroups b, b + 8, ... share an XCD (round-robin dispatch), so each XCD takes a
// contiguous run of (clip, position) work, and inside a clip position p maps to tile r + k s (residue classes r = 0 .. s-1 in
// turn, s = dilation / tile width capped at 16): the tiles an XCD holds at one time then include the ones d columns away,
// whose centre columns are this tile's +-d taps.
__device__ __forceinline__ void ap_tile_of_block(int bid, int nblk, int ntiles, int d, int tile_cols, int &b, int &tile) {
  const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  b = logical / ntiles;
  int p = logical % ntiles;
  const int s = min(max(d / tile_cols, 1), 16);
  if (s > 1) {
    const int wq = ntiles / s, wrem = ntiles % s, cut = wrem * (wq + 1);
    const int cls = p < cut ? p / (wq + 1) : wrem + (p - cut) / wq;
    const int k = p < cut ? p % (wq + 1) : (p - cut) % wq;
    p = cls + k * s;
  }
  tile = p;
}
}  // namespace ap
#endif

namespace ap {
int launch_fold_and_pack(ap_ctx *ctx, const float *blob, hipStream_t st);
int launch_embed(ap_ctx *ctx, float step, float *part_t, hipStream_t st);
int launch_init_conv(ap_ctx *ctx, const float *x, float *h, int B, int L, hipStream_t st);
// AP_PREC_BF16 chain form: bf16 images of u = h + part_t, [clip][C / 32][L][32] (ap_resblock_bf16p.hip, UB); `out` is null on the
// net's last layer, `pt_next` is the next layer's part_t
struct UbArgs {
  const void *in;
  void *out;
  const float *pt_next;
};
int launch_make_ub(const float *h, const float *pt, void *ub, int B, int C, int L, hipStream_t st);
int launch_resblock(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                    int accumulate, int B, int L, hipStream_t st, float *aout = nullptr, const UbArgs *ub = nullptr, void *gout = nullptr,
                    void *fout = nullptr);
int launch_final_affine(ap_ctx *ctx, const float *skip, const float *x, float *eps_out, float *out, float ca,
                        float cb, float cs, const float *z, uint64_t seed, uint32_t draw, uint64_t utt_offset,
                        int B, int L, hipStream_t st);
int launch_affine_noise(const float *x, float *out, float ca, float cs, const float *z, uint64_t seed,
                        uint32_t draw, uint64_t utt_offset, int B, int L, hipStream_t st);
int launch_pack_bf16(ap_ctx *ctx, hipStream_t st);
int launch_final_affine_bf16(ap_ctx *ctx, const float *skip, const float *x, float *eps_out, float *out, float ca, float cb,
                             float cs, const float *z, uint64_t seed, uint32_t draw, uint64_t utt_offset, int B, int L,
                             hipStream_t st);                    // returns 1 if the shape is not served (caller: fp32 kernel)
// `gout` non-null: the deferred-skip form -- the block writes h' and its gate output as a bf16 image [clip][L][256] to gout and
// leaves `skip` alone; launch_skipgemm_bf16 then adds a group of layers' skip_conv outputs in one GEMM (ap_skipgemm_bf16.hip)
// ap_resblock_bf16s.hip: the deferred-skip block for launches of at most one 128-sample tile per CU (64-sample tiles, bit-identical results)
bool resblock_bf16s_serves(const ap_ctx *ctx, int B, int L);
int launch_resblock_bf16s(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, void *gout, int B, int L, hipStream_t st);
int launch_resblock_bf16(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                         int accumulate, int B, int L, hipStream_t st, const UbArgs *ub = nullptr, void *gout = nullptr, void *fout = nullptr);
int launch_resblock_bf16p(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                          int accumulate, int B, int L, hipStream_t st, const UbArgs *ub = nullptr, void *gout = nullptr, void *fout = nullptr);   // persistent form; returns 1 if the shape is not served
                                                                  // (fout, with gout: + the gate's derivative factors, 128 KB per 128-sample tile)
// AP_PREC_BF16_STORE (ap_resblock_bf16u.hip): the residual stream as bf16 images of u = h + part_t, [clip][C / 32][L][32]
bool resblock_bf16u_serves(const ap_ctx *ctx, int L);
int launch_init_conv_u(ap_ctx *ctx, const float *x, const float *pt0, void *u, int B, int L, hipStream_t st);
bool resblock_bf16us_serves(const ap_ctx *ctx, int B, int L);     // AP_PREC_BF16_STORE, at most one 128-sample tile per CU: 64-sample tiles (ap_resblock_bf16us.hip)
int launch_resblock_bf16us(ap_ctx *ctx, int layer, const void *uin, const float *pt_next, void *uout, void *gout, int B, int L, hipStream_t st);
int launch_resblock_bf16u(ap_ctx *ctx, int layer, const void *uin, const float *pt_next, void *uout, void *gout, int B, int L, hipStream_t st,
                          void *fout = nullptr);                  // uout null: the net's last layer; fout: + the gate's derivative factors
int launch_skipgemm_bf16(ap_ctx *ctx, int layer0, int nl, const void *gimg, float *skip, int accumulate, int B, int L, hipStream_t st);
int launch_resblock_bf16w(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                          int accumulate, int B, int L, hipStream_t st);   // one wave per SIMD; returns 1 if the shape is not served
int launch_pack_f32w(ap_ctx *ctx, hipStream_t st);
bool resblock_f32w_serves(const ap_ctx *ctx, int B, int L);       // AP_PREC_F32 in minimal-filtering form, and this shape is built
int launch_resblock_f32w(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                         int accumulate, int B, int L, hipStream_t st, float *aout = nullptr);   // hout null: the net's last layer (no res_conv, no h'); returns 1 if the shape is not served
int launch_pack_split(ap_ctx *ctx, hipStream_t st);
int launch_pack_split23(ap_ctx *ctx, hipStream_t st);            // ap_resblock_f32s2.hip: AP_PREC_F32_SPLIT with the dilated conv in F(2,3) form
bool resblock_split23_serves(const ap_ctx *ctx, int B, int L);
int launch_resblock_split23(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip, int accumulate, int B, int L,
                            hipStream_t st);
int launch_resblock_split(ap_ctx *ctx, int layer, const float *hin, const float *pt, float *hout, float *skip,
                          int accumulate, int B, int L, hipStream_t st);
int prepare_bwd_f32(ap_ctx *ctx, hipStream_t st);                // ap_ctx_prepare_backward, per precision (allocate + pack + synchronise)
int prepare_bwd_bf16(ap_ctx *ctx, hipStream_t st);
int launch_m5(ap_m5 *m, const float *x, float *logprobs, int B, int L, hipStream_t st);
int launch_m5_bwd(ap_m5 *m, const float *x, const float *dlogp, float *dx, int B, int L, hipStream_t st);
int launch_m5_fold(ap_m5 *m, const float *blob, float bn_eps, hipStream_t st);
}  // namespace ap
