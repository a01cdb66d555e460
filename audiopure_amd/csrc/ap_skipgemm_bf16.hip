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
//   * g tile: global -> registers (8 x 16 B per thread, requested two chunks ahead) -> ds_write_b128 into the other of two
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

  // staging unit: piece p = tid + 512 i -> column 2 wave + (lane >> 5) + 16 i, 16-byte piece q = lane & 31 of its 512-byte row.
  // Two register sets: inside a tile the image of chunk k + 2 is requested at the top of chunk k (two chunks = 128 KB per CU in
  // flight; one chunk in flight left the HBM pipe empty between a chunk's wait and the next request: 3.7 TB/s), across a tile
  // boundary one chunk ahead (the epilogue's skip rows need the registers).
  // (the lane's geometry is re-derived from a lane id read on the spot -- volatile asm, not hoisted out of the tile loop: kept
  //  across the loop the eight offsets of each side are spilled, and a scratch reload is a vector-memory load whose wait drains
  //  every request in flight)
  u32x4 st[2][8];
  auto issue_g = [&](u32x4(&dst)[8], int slot, int b, int t0) {
    const __amdgpu_buffer_rsrc_t rs = g_rsrc(slot, b);
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned v0 = (unsigned)(t0 + 2 * wave + (ln >> 5)) * 512u + (unsigned)(ln & 31) * 16u;
#pragma unroll
    for (int i = 0; i < 8; i++)                                  // (a column at or past L lies past the image: the range check returns zeros)
      dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, v0 + (unsigned)(16 * i) * 512u, 0, 2));
  };
  auto write_g = [&](const u32x4(&src)[8], unsigned char *buf) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    unsigned char *d0 = buf + (2 * wave + (ln >> 5)) * (SGS * 2) + (ln & 31) * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4 *>(d0 + 16 * i * (SGS * 2)) = src[i];
  };

  int b_cur, t0_cur;
  tile_bt(tile, b_cur, t0_cur);
  issue_g(st[0], 0, b_cur, t0_cur);
  bf16x8 a[NKS];                                                 // weight fragments of the chunk in flight / next chunk (ring of one chunk)
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) a[ks] = ld_a(0, ks);
  write_g(st[0], lds);
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

    // one chunk k: 16 k-steps x 4 column tiles from image `par`.  Image requests at its top, by kind: first chunk of a tile (0):
    // chunks 1 and 2 -> sets 1 and 0; a middle chunk (1; P = k & 1): chunk k + 2 -> set P; the last chunk (2): the NEXT tile's
    // chunk 0 -> set 0.  At its end the next chunk's image (set 1 / P ^ 1 / 0) goes to the other LDS image, then the barrier.
    auto chunk = [&](int k, auto kind_tag, auto p_tag) {
      constexpr int KIND = decltype(kind_tag)::value, P = decltype(p_tag)::value;
      constexpr bool LAST = KIND == 2;
      constexpr int WSET = KIND == 0 ? 1 : KIND == 1 ? (P ^ 1) : 0;
      const unsigned char *gb = lds + par * GB + rdoff;
      if constexpr (KIND == 0) {
        issue_g(st[1], 1, b, t0);
        if (nl >= 3) issue_g(st[0], 2, b, t0);
      } else if constexpr (KIND == 1) {
        if (k + 2 <= nl - 1) issue_g(st[P], k + 2, b, t0);
      } else {
        issue_g(st[0], 0, b_nxt, t0_nxt);
      }
      const int nslot = LAST ? 0 : k + 1;
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 bv[4];                                              // a column tile's B fragment is re-read for the next k-step right behind its MFMA
      auto rdb = [&](int ct, int ks) { return *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * (SGS * 2) + ks * 32); };
#pragma unroll
      for (int ct = 0; ct < 4; ct++) bv[ct] = rdb(ct, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bv[ct], acc[ct], 0, 0, 0);
          if (ks + 1 < NKS) bv[ct] = rdb(ct, ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
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
      write_g(st[WSET], lds + (par ^ 1) * GB);
      __syncthreads();
      par ^= 1;
    };

    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    if (nl >= 2) {
      chunk(0, K0{}, K0{});
      int k = 1;
#pragma unroll 1
      for (; k + 1 <= nl - 2; k += 2) {
        chunk(k, K1{}, K1{});
        chunk(k + 1, K1{}, K0{});
      }
      if (k <= nl - 2) chunk(k, K1{}, K1{});
    }
    chunk(nl - 1, K2{}, K0{});

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

#ifdef AP_TOOLS
// tools/csrc/ap_skipgemm_bf16w.hip (tools library only): the same GEMM on 256-sample tiles, measured equal and not shipped
int launch_skipgemm_bf16w(ap_ctx *ctx, int layer0, int nl, const void *gimg, float *skip, int accumulate, int B, int L, hipStream_t st);
extern int g_skipgemm_wide;                                      // tools/cmp_skipgemm.py: 1 = the 256-sample kernel
#endif

// skip (+)= sum_{n = layer0 .. layer0 + nl - 1} (W_skip,n g_n + b_skip,n); gimg = [nl][B][L][256] bf16 (slot n - layer0)
int launch_skipgemm_bf16(ap_ctx *ctx, int layer0, int nl, const void *gimg, float *skip, int accumulate, int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  if (C != 256 || S != 256 || !ctx->w2p_bf) { set_error("skip GEMM: AP_PREC_BF16 context with res = skip = 256 channels only"); return -22; }
  if (nl < 1 || layer0 < 0 || layer0 + nl > ctx->NL || B < 1 || L < 1) { set_error("skip GEMM: layers [%d, %d) B=%d L=%d", layer0, layer0 + nl, B, L); return -22; }
  // (1024 L: a clip's fp32 skip rows -- the out-of-clip sentinel offset 0x80000000 of the epilogue must lie beyond them)
  if ((size_t)L * 1024 >= ((size_t)1 << 31)) { set_error("skip GEMM: clip too long (L * 1024 bytes of skip rows per clip must stay below 2^31)"); return -22; }
  const int n_cu = device_cu_count();
  const int ntiles = (L + SPT - 1) / SPT;
  const int nblk = B * ntiles;
  const int grid = nblk < n_cu ? nblk : n_cu;
  const size_t n2 = (size_t)(C + S) * C;
  const unsigned wbytes = (unsigned)((size_t)ctx->NL * n2 * 2);
  const unsigned w2_off = (unsigned)((size_t)layer0 * n2 * 2);
  const float *b2 = ctx->b2 + (size_t)layer0 * (C + S) + C;
#ifdef AP_TOOLS
  if (g_skipgemm_wide == 1) return launch_skipgemm_bf16w(ctx, layer0, nl, gimg, skip, accumulate, B, L, st);
#endif
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
