// AP_PREC_BF16, deferred-skip form (round 4): skip (+)= sum over a group of layers of W_skip,n . g_n + b_skip,n
// (WaveNet.py:95-97 `skip = skip_conv(out)`, :131-133 `skip += skip_n` -- the sum over layers taken inside one GEMM).
//
// The DS instantiations of resblock_bf16p_kernel write each layer's gate output as the bf16 image GEMM2 consumes anyway,
// g[slot][clip][sample][256 channels] (512 B per sample), instead of running skip_conv and its read-modify-write of `skip`
// per layer.  This kernel is the K-concatenated GEMM [S x nl C] . [nl C x L] over those images: per 128-sample tile it streams
// nl x 64 KB of g from HBM (read once, 2 B per element instead of the 8 B a per-layer read-modify-write of fp32 skip costs),
// accumulates in the matrix pipe's fp32 accumulators over all nl x 256 k, and touches `skip` once per group -- a plain store for
// the first group of an evaluation, one read-modify-write otherwise.
//
// One persistent workgroup per CU (8 waves; wave w owns skip rows [32w, 32w + 32) x 128 columns = 64 accumulator registers),
// a chunk = one layer's 256 k:
//   * g tile: global -> registers (8 x 16 B per thread, requested one chunk ahead) -> ds_write_b128 into the other of two
//     [column][channel] images with 528-byte rows (the block kernel's g image: conflict-free ds_read_b128 B fragments);
//   * weights: the skip rows of the block's packed W2 image ([wave][row tile 1][k-step][lane][8 bf16]), L2 -> registers through a
//     ring of one chunk (16 fragments): a fragment is replaced right behind its last MFMA by the same k-step's of the next chunk,
//     so every weight request is at least a chunk older than its use and the g requests (HBM latency) never sit in front of a
//     fragment that is needed soon (vmcnt retires in order);
//   * one barrier per chunk; the epilogue goes through a wave-private LDS patch (aliasing the image just consumed) so the
//     read-modify-write is 16 B per lane both ways, as in the block kernel.
// The next tile's first chunk is requested under the last chunk's MFMAs.
#include <type_traits>

#include "ap_common.h"

namespace ap {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SPT = 128;                  // time tile
constexpr int SGS = 256 + 8;              // bf16 per column row of the g image in LDS (528 B)
constexpr int SPS = 32;                   // fp32 per row of the wave-private output patch (128 B)

__device__ __forceinline__ int srowoff(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

}  // namespace

template <bool RAG>
__global__ __launch_bounds__(512, 2) void skipgemm_bf16_kernel(
    const void *__restrict__ gimg, float *__restrict__ skip,
    const void *__restrict__ wbase, unsigned wbytes, unsigned w2_off, unsigned w2_lstride,   // packed bf16 W2 slab; first layer of the group; bytes per layer
    const float *__restrict__ b2, unsigned b2_lstride,                                        // skip-row bias of the first layer; floats per layer
    int B, int L, int nl, int accumulate, int ntiles, int nblk) {
  constexpr int C = 256, NW = 8, NKS = C / 16;
  constexpr int GB = SPT * SGS * 2;                              // 67,584 B per g image
  constexpr int BSOFF = 2 * GB;                                  // summed bias (C floats)
  constexpr int LDS_BYTES = BSOFF + C * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(NW * 32 * SPS * 4 <= GB, "the output patches alias one g image");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;

  int tile = blockIdx.x;
  const int tstep = gridDim.x;
  if (tile >= nblk) return;

  if (tid < C) {                                                 // sum of the group's skip_conv biases (WaveNet.py:95 per layer)
    float s = 0.f;
    for (int n = 0; n < nl; n++) s += b2[(size_t)n * b2_lstride + tid];
    reinterpret_cast<float *>(lds + BSOFF)[tid] = s;
  }

  auto uni_rsrc = [&](uint64_t hb, unsigned bytes) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)hb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t wrs = uni_rsrc((uint64_t)wbase, wbytes);
  const unsigned img_bytes = (unsigned)L * 512u;                 // one (slot, clip) image
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  auto g_rsrc = [&](int slot, int b) { return uni_rsrc((uint64_t)gimg + ((uint64_t)slot * (uint64_t)B + (uint64_t)b) * (uint64_t)img_bytes, img_bytes); };
  const unsigned lane16 = (unsigned)lane * 16u;
  // skip rows of layer (group's first + slot), k-step ks: [wave][row tile 2][k-step 16][lane][8 bf16], row tile 1
  auto ld_a = [&](int slot, int ks) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, w2_off + (unsigned)slot * w2_lstride + (unsigned)((wave * 2 * NKS + NKS + ks) * 1024), 0));
  };
  auto tile_bt = [&](int tl, int &b, int &t0) {
    b = __builtin_amdgcn_readfirstlane(tl / ntiles);
    t0 = __builtin_amdgcn_readfirstlane((tl % ntiles) * SPT);
  };

  // staging unit: piece p = tid + 512 i -> column 2 wave + (lane >> 5) + 16 i, 16-byte piece q = lane & 31 of its 512-byte row
  const int colw = 2 * wave + (lane >> 5), q = lane & 31;
  u32x4 st[8];
  auto issue_g = [&](int slot, int b, int t0) {
    const __amdgpu_buffer_rsrc_t rs = g_rsrc(slot, b);
#pragma unroll
    for (int i = 0; i < 8; i++)                                  // (a column at or past L lies past the image: the range check returns zeros)
      st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(t0 + colw + 16 * i) * 512u + (unsigned)q * 16u, 0, 2));
  };
  auto write_g = [&](unsigned char *buf) {
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4 *>(buf + (colw + 16 * i) * (SGS * 2) + q * 16) = st[i];
  };

  int b_cur, t0_cur;
  tile_bt(tile, b_cur, t0_cur);
  issue_g(0, b_cur, t0_cur);
  bf16x8 a[NKS];                                                 // weight fragments of the chunk in flight / next chunk (ring of one chunk)
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) a[ks] = ld_a(0, ks);
  write_g(lds);
  __syncthreads();                                               // first image and the bias sums visible
  int par = 0;                                                   // image the next chunk computes from

  const int rdoff = (j * SGS + 8 * hh) * 2;                      // this lane's B-fragment byte offset inside an image

#pragma unroll 1
  for (; tile < nblk; tile += tstep) {
    const int t0 = t0_cur, b = b_cur;
    int b_nxt = b_cur, t0_nxt = t0_cur;
    if (tile + tstep < nblk) tile_bt(tile + tstep, b_nxt, t0_nxt);   // (past the last tile: the last tile's first chunk again, dropped)

    f32x16 acc[4];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(lds + BSOFF + (32 * wave + 8 * qq + 4 * hh) * 4);
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        acc[ct][4 * qq + 0] = bv4[0];
        acc[ct][4 * qq + 1] = bv4[1];
        acc[ct][4 * qq + 2] = bv4[2];
        acc[ct][4 * qq + 3] = bv4[3];
      }
    }

    unsigned evoff[4];                                           // lane = (row lane >> 3 (+8 per step), column quad lane & 7)
    float pre[4][16];
    const __amdgpu_buffer_rsrc_t srs = uni_rsrc((uint64_t)(skip + (size_t)b * C * L), clip_bytes);

    // one chunk: 16 k-steps x 4 column tiles from image `par`; the next chunk's image is requested first and written last
    auto chunk = [&](int nslot, int nb, int nt0, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const unsigned char *gb = lds + par * GB + rdoff;
      issue_g(nslot, nb, nt0);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 ba[4], bb[4];
      auto rdg = [&](bf16x8(&bq)[4], int ks) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) bq[ct] = *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * (SGS * 2) + ks * 32);
      };
      rdg(ba, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
        if (ks + 1 < NKS) { if (ks & 1) rdg(ba, ks + 1); else rdg(bb, ks + 1); }
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], (ks & 1) ? bb[ct] : ba[ct], acc[ct], 0, 0, 0);
        a[ks] = ld_a(nslot, ks);                                 // the same k-step of the next chunk
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (LAST) {                                      // the running skip rows, requested behind the last MFMA (B fragments are dead)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          const int t = t0 + 32 * ct + 4 * (lane & 7);
          evoff[ct] = t < L ? ((unsigned)(32 * wave + (lane >> 3)) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
        }
        if (accumulate) {
#pragma unroll
          for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int p = 0; p < 4; p++) {
              const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, evoff[ct], 8 * p * L * 4, 2));
#pragma unroll
              for (int i = 0; i < 4; i++) pre[ct][4 * p + i] = v[i];
            }
        } else {
#pragma unroll
          for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) pre[ct][r] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      write_g(lds + (par ^ 1) * GB);
      __syncthreads();
      par ^= 1;
    };

#pragma unroll 1
    for (int slot = 0; slot + 1 < nl; slot++) chunk(slot + 1, b, t0, std::false_type{});
    chunk(0, b_nxt, t0_nxt, std::true_type{});

    // epilogue: MFMA layout (4 rows x 1 column per lane) -> wave-private patch (in the image just consumed: every wave is past the
    // chunk's barrier) -> 1 row x 4 columns per lane -> 16-byte read-modify-write
    float *patch = reinterpret_cast<float *>(lds + (par ^ 1) * GB) + wave * 32 * SPS;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
#pragma unroll
      for (int r = 0; r < 16; r++) patch[srowoff(r, hh) * SPS + j] = acc[ct][r];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const float4 v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * p) * SPS + 4 * (lane & 7));
        f32x4 o;
        o[0] = pre[ct][4 * p + 0] + v.x;
        o[1] = pre[ct][4 * p + 1] + v.y;
        o[2] = pre[ct][4 * p + 2] + v.z;
        o[3] = pre[ct][4 * p + 3] + v.w;
        // (row step in the VGPR offset, soffset = 0: a > 8-byte buffer store with an SGPR soffset reads its data late)
        const unsigned so = evoff[ct] + (unsigned)(8 * p * L * 4);
        const u32x4 ou = __builtin_bit_cast(u32x4, o);           // (whole-vector bit_cast: element-wise it is mis-folded to a splat)
        if constexpr (RAG) {
          const int nv = L - (t0 + 32 * ct + 4 * (lane & 7));    // valid samples of this lane's column quad (>= 4: all)
          if (nv >= 4) __builtin_amdgcn_raw_buffer_store_b128(ou, srs, so, 0, 2);
          else {
            if (nv >= 1) __builtin_amdgcn_raw_buffer_store_b32(ou[0], srs, so, 0, 2);
            if (nv >= 2) __builtin_amdgcn_raw_buffer_store_b32(ou[1], srs, so + 4u, 0, 2);
            if (nv >= 3) __builtin_amdgcn_raw_buffer_store_b32(ou[2], srs, so + 8u, 0, 2);
          }
        } else __builtin_amdgcn_raw_buffer_store_b128(ou, srs, so, 0, 2);
      }
    }
    __syncthreads();                                             // the patches are the image the next tile's second chunk is written to
    b_cur = b_nxt;
    t0_cur = t0_nxt;
  }
}

// skip (+)= sum_{n = layer0 .. layer0 + nl - 1} (W_skip,n g_n + b_skip,n); gimg = [nl][B][L][256] bf16 (slot n - layer0)
int launch_skipgemm_bf16(ap_ctx *ctx, int layer0, int nl, const void *gimg, float *skip, int accumulate, int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  if (C != 256 || S != 256 || !ctx->w2p_bf) { set_error("skip GEMM: AP_PREC_BF16 context with res = skip = 256 channels only"); return -22; }
  if (nl < 1 || layer0 < 0 || layer0 + nl > ctx->NL || B < 1 || L < 1) { set_error("skip GEMM: layers [%d, %d) B=%d L=%d", layer0, layer0 + nl, B, L); return -22; }
  if ((size_t)L * 512 >= ((size_t)1 << 31)) { set_error("skip GEMM: clip too long for the bf16 g image"); return -22; }
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (n_cu_of[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    n_cu_of[dev] = n;
  }
  const int ntiles = (L + SPT - 1) / SPT;
  const int nblk = B * ntiles;
  const int grid = nblk < n_cu_of[dev] ? nblk : n_cu_of[dev];
  const size_t n2 = (size_t)(C + S) * C;
  const unsigned wbytes = (unsigned)((size_t)ctx->NL * n2 * 2);
  const unsigned w2_off = (unsigned)((size_t)layer0 * n2 * 2);
  const float *b2 = ctx->b2 + (size_t)layer0 * (C + S) + C;
  if (L % 4)
    skipgemm_bf16_kernel<true><<<(unsigned)grid, 512, 0, st>>>(gimg, skip, ctx->w2p_bf, wbytes, w2_off, (unsigned)(n2 * 2), b2, (unsigned)(C + S), B, L, nl,
                                                             accumulate, ntiles, nblk);
  else
    skipgemm_bf16_kernel<false><<<(unsigned)grid, 512, 0, st>>>(gimg, skip, ctx->w2p_bf, wbytes, w2_off, (unsigned)(n2 * 2), b2, (unsigned)(C + S), B, L, nl,
                                                              accumulate, ntiles, nblk);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap
