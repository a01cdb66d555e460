// Tools library only (never shipped; built by `python __graft_entry__.py --tools`): the deferred-skip GEMM of
// audiopure_amd/csrc/ap_skipgemm_bf16.hip on 256-sample tiles -- VERDICT r4 item 3a, "two column tiles per weight pass".
// Measured equal to the product kernel (10.03 against 10.07 ms at B = 128: profiles/r5_skipgemm_two_column_tiles_ab.txt) and
// therefore not shipped: the GEMM streams its g images at 3.8 TB/s, the weight fragments are not what it waits for.
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

// ---------------------------------------------------------------------------------------------------------------------------
// Tools library only (the A/B of tools/cmp_skipgemm.py; measured equal to the product kernel -- 10.03 against 10.07 ms at B = 128 -- and
// therefore not shipped: the GEMM streams its g images at 3.8 TB/s, the weight fragments are not what it waits for).
// The same GEMM on 256-sample tiles (two column tiles per weight pass): a W_skip fragment feeds eight column tiles of
// 32 instead of four, so the 4.7 MB of fragments a group of 36 layers streams through the CU's vector-memory path serve twice the columns.
// To fit LDS and registers a chunk is HALF a layer (128 k = 8 k-steps): two [256 columns][128 k] images of 272-byte rows (139 KB), a
// fragment ring of one chunk (8), 128 accumulator registers; the running skip rows are read in the epilogue, not ahead of it.  Every
// output element sees the same k-steps in the same order as in the 128-sample kernel: results are bit-identical (tools/cmp_skipgemm.py).
// ---------------------------------------------------------------------------------------------------------------------------
template <bool RAG>
__global__ __launch_bounds__(512, 2) void skipgemm_bf16w_kernel(
    const void *__restrict__ gimg, float *__restrict__ skip,
    const void *__restrict__ wbase, unsigned wbytes, unsigned w2_off, unsigned w2_lstride,
    const float *__restrict__ b2, unsigned b2_lstride,
    int B, int L, int nl, int accumulate, int ntiles, int nblk) {
  constexpr int C = 256, NW = 8, NKS = 8, WT = 256;              // k-steps per chunk (half a layer), columns per tile
  constexpr int RB = 128 * 2 + 16;                               // bytes per column row of an image (272: conflict-free 16-byte reads)
  constexpr int GB = WT * RB;                                    // 69,632 B per image
  constexpr int BSOFF = 2 * GB;
  constexpr int LDS_BYTES = BSOFF + C * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(NW * 32 * SPS * 4 <= GB, "the output patches alias one image");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;

  int tile = blockIdx.x;
  const int tstep = gridDim.x;
  if (tile >= nblk) return;
  const int nc = 2 * nl;                                         // chunks per tile

  if (tid < C) {
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
  const unsigned img_bytes = (unsigned)L * 512u;
  const unsigned clip_bytes = (unsigned)C * (unsigned)L * 4u;
  auto g_rsrc = [&](int slot, int b) { return uni_rsrc((uint64_t)gimg + ((uint64_t)slot * (uint64_t)B + (uint64_t)b) * (uint64_t)img_bytes, img_bytes); };
  const unsigned lane16 = (unsigned)lane * 16u;
  // chunk c = (layer slot c >> 1, k half c & 1); skip rows: [wave][row tile 2][k-step 16][lane][8 bf16], row tile 1
  auto ld_a = [&](int c, int ks) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
        wrs, lane16, w2_off + (unsigned)(c >> 1) * w2_lstride + (unsigned)((wave * 2 * 16 + 16 + 8 * (c & 1) + ks) * 1024), 0));
  };
  auto tile_bt = [&](int tl, int &b, int &t0) {
    b = __builtin_amdgcn_readfirstlane(tl / ntiles);
    t0 = __builtin_amdgcn_readfirstlane((tl % ntiles) * WT);
  };
  // staging: piece p = tid + 512 i -> column (tid >> 4) + 32 i, 16-byte piece q = tid & 15 of the column's 256-byte half row
  u32x4 st[2][8];
  auto issue_g = [&](u32x4(&dst)[8], int c, int b, int t0) {
    const __amdgpu_buffer_rsrc_t rs = g_rsrc(c >> 1, b);
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned v0 = (unsigned)(t0 + 4 * wave + (ln >> 4)) * 512u + (unsigned)(c & 1) * 256u + (unsigned)(ln & 15) * 16u;
#pragma unroll
    for (int i = 0; i < 8; i++)                                  // (a column at or past L lies past the image: the range check returns zeros)
      dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, v0 + (unsigned)(32 * i) * 512u, 0, 2));
  };
  auto write_g = [&](const u32x4(&src)[8], unsigned char *buf) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    unsigned char *d0 = buf + (4 * wave + (ln >> 4)) * RB + (ln & 15) * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4 *>(d0 + 32 * i * RB) = src[i];
  };

  int b_cur, t0_cur;
  tile_bt(tile, b_cur, t0_cur);
  issue_g(st[0], 0, b_cur, t0_cur);
  bf16x8 a[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) a[ks] = ld_a(0, ks);
  write_g(st[0], lds);
  __syncthreads();
  int par = 0;
  const int rdoff = j * RB + 16 * hh;

#pragma unroll 1
  for (; tile < nblk; tile += tstep) {
    const int t0 = t0_cur, b = b_cur;
    int b_nxt = b_cur, t0_nxt = t0_cur;
    if (tile + tstep < nblk) tile_bt(tile + tstep, b_nxt, t0_nxt);

    f32x16 acc[8];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(lds + BSOFF + (32 * wave + 8 * qq + 4 * hh) * 4);
#pragma unroll
      for (int ct = 0; ct < 8; ct++) {
        acc[ct][4 * qq + 0] = bv4[0];
        acc[ct][4 * qq + 1] = bv4[1];
        acc[ct][4 * qq + 2] = bv4[2];
        acc[ct][4 * qq + 3] = bv4[3];
      }
    }
    const __amdgpu_buffer_rsrc_t srs = uni_rsrc((uint64_t)(skip + (size_t)b * C * L), clip_bytes);

    // one chunk: 8 k-steps x 8 column tiles from image `par`; image requests as in the 128-sample kernel (two chunks ahead inside a tile,
    // the next tile's first chunk from the last one)
    auto chunk = [&](int k, auto kind_tag, auto p_tag) {
      constexpr int KIND = decltype(kind_tag)::value, P = decltype(p_tag)::value;
      constexpr bool LAST = KIND == 2;
      constexpr int WSET = KIND == 0 ? 1 : KIND == 1 ? (P ^ 1) : 0;
      const unsigned char *gb = lds + par * GB + rdoff;
      if constexpr (KIND == 0) {
        issue_g(st[1], 1, b, t0);
        if (nc >= 3) issue_g(st[0], 2, b, t0);
      } else if constexpr (KIND == 1) {
        if (k + 2 <= nc - 1) issue_g(st[P], k + 2, b, t0);
      } else {
        issue_g(st[0], 0, b_nxt, t0_nxt);
      }
      const int nchunk = LAST ? 0 : k + 1;
      __builtin_amdgcn_sched_barrier(0);
      auto rdb = [&](int ct, int ks) { return *reinterpret_cast<const bf16x8 *>(gb + (32 * ct) * RB + ks * 32); };
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
#pragma unroll
        for (int g4 = 0; g4 < 2; g4++) {                         // two groups of four column tiles share the k-step's fragment
          bf16x8 bv[4];
#pragma unroll
          for (int c4 = 0; c4 < 4; c4++) bv[c4] = rdb(4 * g4 + c4, ks);
#pragma unroll
          for (int c4 = 0; c4 < 4; c4++) acc[4 * g4 + c4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bv[c4], acc[4 * g4 + c4], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        a[ks] = ld_a(nchunk, ks);                                // the same k-step of the next chunk
        __builtin_amdgcn_sched_barrier(0);
      }
      write_g(st[WSET], lds + (par ^ 1) * GB);
      __syncthreads();
      par ^= 1;
    };

    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    {                                                            // nc >= 2 always
      chunk(0, K0{}, K0{});
      int k = 1;
#pragma unroll 1
      for (; k + 1 <= nc - 2; k += 2) {
        chunk(k, K1{}, K1{});
        chunk(k + 1, K1{}, K0{});
      }
      if (k <= nc - 2) chunk(k, K1{}, K1{});
    }
    chunk(nc - 1, K2{}, K0{});

    // epilogue: accumulator layout -> wave-private patch (in the image just consumed) -> 1 row x 4 columns per lane -> 16-byte (read-modify-)write
    float *patch = reinterpret_cast<float *>(lds + (par ^ 1) * GB) + wave * 32 * SPS;
#pragma unroll
    for (int ct = 0; ct < 8; ct++) {
      const int t = t0 + 32 * ct + 4 * (lane & 7);
      const unsigned evoff = t < L ? ((unsigned)(32 * wave + (lane >> 3)) * (unsigned)L + (unsigned)t) * 4u : 0x80000000u;
      f32x4 pre[4];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pre[p] = accumulate ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, evoff, 8 * p * L * 4, 2)) : z;
      }
#pragma unroll
      for (int r = 0; r < 16; r++) patch[srowoff(r, hh) * SPS + j] = acc[ct][r];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const float4 v = *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * p) * SPS + 4 * (lane & 7));
        f32x4 o;
        o[0] = pre[p][0] + v.x;
        o[1] = pre[p][1] + v.y;
        o[2] = pre[p][2] + v.z;
        o[3] = pre[p][3] + v.w;
        const unsigned so = evoff + (unsigned)(8 * p * L * 4);
        const u32x4 ou = __builtin_bit_cast(u32x4, o);
        if constexpr (RAG) {
          const int nv = L - t;
          if (nv >= 4) __builtin_amdgcn_raw_buffer_store_b128(ou, srs, so, 0, 2);
          else {
            if (nv >= 1) __builtin_amdgcn_raw_buffer_store_b32(ou[0], srs, so, 0, 2);
            if (nv >= 2) __builtin_amdgcn_raw_buffer_store_b32(ou[1], srs, so + 4u, 0, 2);
            if (nv >= 3) __builtin_amdgcn_raw_buffer_store_b32(ou[2], srs, so + 8u, 0, 2);
          }
        } else __builtin_amdgcn_raw_buffer_store_b128(ou, srs, so, 0, 2);
      }
    }
    __syncthreads();
    b_cur = b_nxt;
    t0_cur = t0_nxt;
  }
}


int g_skipgemm_wide = -1;                                        // tools/cmp_skipgemm.py: 1 = this kernel

int launch_skipgemm_bf16w(ap_ctx *ctx, int layer0, int nl, const void *gimg, float *skip, int accumulate, int B, int L, hipStream_t st) {
  const int C = ctx->C, S = ctx->S;
  const int n_cu = device_cu_count();
  const size_t n2 = (size_t)(C + S) * C;
  const unsigned wbytes = (unsigned)((size_t)ctx->NL * n2 * 2);
  const unsigned w2_off = (unsigned)((size_t)layer0 * n2 * 2);
  const float *b2 = ctx->b2 + (size_t)layer0 * (C + S) + C;
  const int nt2 = (L + 255) / 256, nb2 = B * nt2, grid2 = nb2 < n_cu ? nb2 : n_cu;
  if (L % 4)
    skipgemm_bf16w_kernel<true><<<(unsigned)grid2, 512, 0, st>>>(gimg, skip, ctx->w2p_bf, wbytes, w2_off, (unsigned)(n2 * 2), b2, (unsigned)(C + S), B, L, nl,
                                                               accumulate, nt2, nb2);
  else
    skipgemm_bf16w_kernel<false><<<(unsigned)grid2, 512, 0, st>>>(gimg, skip, ctx->w2p_bf, wbytes, w2_off, (unsigned)(n2 * 2), b2, (unsigned)(C + S), B, L, nl,
                                                                accumulate, nt2, nb2);
  AP_HIP(hipGetLastError());
  return 0;
}

}  // namespace ap

extern "C" void ap_debug_skipgemm_wide(int v) { ap::g_skipgemm_wide = v; }
