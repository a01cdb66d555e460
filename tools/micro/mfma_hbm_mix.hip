// The bf16 block's mix without the block: per "tile" and CU N v_mfma_f32_32x32x16_bf16 on random operands beside fp32 streams read
// from HBM and written back (touched once, like h / skip in and h' / skip out), nothing else -- no weight stream, no staging, no
// gate, no transposes.  What tile rate does the 1400 W cap allow THAT?  (rocm-smi polled meanwhile.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_hbm_mix.hip -o /tmp/mfma_hbm_mix -lpthread
//   /tmp/mfma_hbm_mix [seconds] [workgroups per CU] [fused | ds_block | ds_skip] [0 = 32x32x16 | 1 = 16x16x32]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one workgroup per CU (512 threads = 8 waves x (64 x 128 accumulators)), persistent over `tiles` tiles.  A tile is `ng` groups of 16 MFMAs
// per wave (128 per workgroup) with `nld` 16-byte loads and `nst` 16-byte stores per lane spread evenly over the groups (8 KB per
// workgroup each).  Mixes (round 4: the "board ceiling" is no longer a one-design number):
//   fused      the round-3 fused block: 4096 MFMAs, 262 KB in, 262 KB out per tile                 (ng 32, nld 32, nst 32)
//   ds_block   the deferred-skip block: 3584 MFMAs, 131 KB in (h), 196 KB out (h' + bf16 g image)  (ng 28, nld 16, nst 24)
//   ds_skip    the skip GEMM, one tile = 36 layers: 18432 MFMAs, 36 x 64 KB in, 131 KB out         (ng 144, nld 288, nst 16)
// SHAPE 1: every v_mfma_f32_32x32x16_bf16 replaced by two v_mfma_f32_16x16x32_bf16 (the same flops).
template <bool MEM, bool MMA, int SHAPE>
__global__ __launch_bounds__(512, 2) void mix_k(const bf16x8 *__restrict__ ops, const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                float *__restrict__ out, int tiles, int ng, int nld, int nst) {
  __shared__ bf16x8 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = ops[(blockIdx.x & 7) * 4096 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  f32x4 acq[2][4][2] = {};
  f32x4 carry = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; t++) {
    const size_t tile = (size_t)t * gridDim.x + blockIdx.x;
    const f32x4 *sp = src + tile * (size_t)nld * 512 + threadIdx.x;
    f32x4 *dp = dst + tile * (size_t)nst * 512 + threadIdx.x;
    int il = 0, is = 0;
#pragma unroll 1
    for (int g = 0; g < ng; g++) {
      f32x4 v = carry;
      if (MEM)
        for (const int e = (g + 1) * nld / ng; il < e; il++) v += __builtin_nontemporal_load(sp + il * 512);
      if (MMA) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int base = (((t * 64 + g * 2 + s) * 8 + wave) * 37) & 4095;
          bf16x8 a[2], b[4];
          for (int r = 0; r < 2; r++) a[r] = lds[(base + r * 64 + lane) & 4095];
          for (int c = 0; c < 4; c++) b[c] = lds[(base + 128 + c * 64 + lane) & 4095];
          for (int r = 0; r < 2; r++)
            for (int c = 0; c < 4; c++) {
              if (SHAPE == 0) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[c], acc[r][c], 0, 0, 0);
              else {
                acq[r][c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[c], acq[r][c][0], 0, 0, 0);
                acq[r][c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[c], a[r], acq[r][c][1], 0, 0, 0);
              }
            }
        }
      }
      if (MEM)
        for (const int e = (g + 1) * nst / ng; is < e; is++) __builtin_nontemporal_store(v + carry, dp + is * 512);
      carry = v;
    }
  }
  float s = carry[0];
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) { for (int e = 0; e < 16; e++) s += acc[r][c][e]; for (int e = 0; e < 4; e++) s += acq[r][c][0][e] + acq[r][c][1][e]; }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

static bool smi(double &w, double &mhz) {
  FILE *f = popen("rocm-smi --showpower --showclocks 2>/dev/null", "r");
  if (!f) return false;
  char line[512]; bool pw = false, ck = false;
  while (fgets(line, sizeof line, f)) {
    const char *p = strstr(line, "Package Power (W): ");
    if (p && !strstr(line, "Max")) { w = atof(p + 19); pw = true; }
    p = strstr(line, "sclk clock level:");
    if (p) { const char *q = strchr(p, '('); if (q) { mhz = atof(q + 1); ck = true; } }
  }
  pclose(f);
  return pw && ck;
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int bpc = argc > 2 ? atoi(argv[2]) : 1;                   // workgroups per CU (1 or 2)
  const char *mix = argc > 3 ? argv[3] : "fused";
  const int shape = argc > 4 ? atoi(argv[4]) : 0;                 // 0: 32x32x16, 1: 16x16x32
  int ng = 32, nld = 32, nst = 32, tiles = 125 / bpc;             // ~125 tiles per CU and launch = the block kernel's 256-clip launch
  double alg_kb = 524.288;                                        // algorithmic bytes the tile stands for (the roofline's numerator)
  if (!strcmp(mix, "ds_block")) { ng = 28; nld = 16; nst = 24; }
  else if (!strcmp(mix, "ds_skip")) { ng = 144; nld = 288; nst = 16; alg_kb = 0.0; }   // (its time per tile / 36 adds to ds_block's per tile: one layer)
  const int nblk = 256 * bpc;
  std::vector<unsigned short> h(8 * 4096 * 8);
  srand(1);
  for (auto &v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8 *ops; float *out; f32x4 *src, *dst;
  const size_t nsrc = (size_t)nblk * tiles * nld * 512, ndst = (size_t)nblk * tiles * nst * 512;
  hipMalloc(&ops, h.size() * 2); hipMalloc(&out, nblk * 512 * 4); hipMalloc(&src, nsrc * 16); hipMalloc(&dst, ndst * 16);
  hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemset(src, 0x3c, nsrc * 16);                                // (finite, non-zero fp32 pattern)
  printf("mix %s, %s, %d workgroup(s) per CU: per tile %d MFMAs, %.0f KB in, %.0f KB out\n", mix, shape ? "16x16x32 (two per 32x32x16)" : "32x32x16", bpc,
         ng * 128, nld * 8.192, nst * 8.192);
  const char *names[] = {"MFMA + HBM streams (the mix)", "HBM streams alone", "MFMA alone"};
  for (int cs = 0; cs < 3; cs++) {
    std::atomic<bool> stop{false};
    std::vector<double> ws, cl;
    std::thread th([&] { while (!stop) { double w, c; if (smi(w, c)) { ws.push_back(w); cl.push_back(c); } std::this_thread::sleep_for(std::chrono::milliseconds(200)); } });
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int i = 0; i < 4; i++) {
#define LAUNCH(M, A)                                                                                     \
  do {                                                                                                   \
    if (shape) mix_k<M, A, 1><<<nblk, 512>>>(ops, src, dst, out, tiles, ng, nld, nst);                   \
    else mix_k<M, A, 0><<<nblk, 512>>>(ops, src, dst, out, tiles, ng, nld, nst);                         \
  } while (0)
        if (cs == 0) LAUNCH(true, true);
        else if (cs == 1) LAUNCH(true, false);
        else LAUNCH(false, true);
        launches++;
      }
      hipDeviceSynchronize();
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true; th.join();
    double w = 0, c = 0; size_t n = 0;
    for (size_t i = 2; i < ws.size(); i++) { w += ws[i]; c += cl[i]; n++; }
    const double ms = el / launches * 1e3, tl = (double)nblk * tiles, bytes = (nld + nst) * 8192.0;
    printf("%-30s %8.3f ms per launch of %d tiles = %7.3f us per tile and CU: %6.1f TFLOP/s, %5.2f TB/s (read + write)", names[cs], ms, (int)tl,
           ms * 1e3 / tiles * 1.0 / bpc, cs == 1 ? 0.0 : tl * ng * 128 * 32768.0 / ms / 1e9, cs == 2 ? 0.0 : tl * bytes / ms / 1e9);
    if (alg_kb > 0 && cs == 0) printf(" = %.3f of 8 TB/s on the tile's ALGORITHMIC %.0f KB", tl * alg_kb * 1024 / ms / 1e9 / 8.0, alg_kb);
    printf("; %5.0f W, %5.0f MHz\n", n ? w / n : 0.0, n ? c / n : 0.0);
  }
  return 0;
}
