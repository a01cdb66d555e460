// What the 1400 W package cap leaves the bf16 matrix pipe: bare v_mfma_f32_32x32x16_bf16 on all CUs (64 x 128 output tile per wave,
// operands re-read from LDS every step, 4 waves per SIMD), sustained for a few seconds per case while rocm-smi is polled from a
// host thread -- random operands, all-zero operands, and random operands beside an HBM copy stream (the block kernel's mix).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power -lpthread && /tmp/mfma_power
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

__global__ __launch_bounds__(512, 2) void mfma_k(const bf16x8 *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ bf16x8 lds[4096];                                   // 64 KB of operands
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[(blockIdx.x & 7) * 4096 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  for (int it = 0; it < iters; it++) {
    const int base = ((it * 8 + wave) * 37) & 4095;
    bf16x8 a[2], b[4];
    for (int r = 0; r < 2; r++) a[r] = lds[(base + r * 64 + lane) & 4095];
    for (int c = 0; c < 4; c++) b[c] = lds[(base + 128 + c * 64 + lane) & 4095];
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[c], acc[r][c], 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the fp32 matrix instruction the headline block and the conv family run on: v_mfma_f32_32x32x2_f32, same tile per wave
__global__ __launch_bounds__(512, 2) void mfma32_k(const float *__restrict__ src, float *__restrict__ out, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = src[(blockIdx.x & 7) * 16384 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][4] = {};
  for (int it = 0; it < iters; it++) {
    const int base = ((it * 8 + wave) * 389) & 16383;
    float a[2], b[4];
    for (int r = 0; r < 2; r++) a[r] = lds[(base + r * 64 + lane) & 16383];
    for (int c = 0; c < 4; c++) b[c] = lds[(base + 128 + c * 64 + lane) & 16383];
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 4; c++) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[c], acc[r][c], 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 2; r++) for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[r][c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ void copy_k(const f32x4 *__restrict__ a, f32x4 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
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
  const int nblk = 256 * 8, iters = 4096;
  std::vector<unsigned short> h(8 * 4096 * 8);
  srand(1);
  for (auto &v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8 *src, *zero; float *out; f32x4 *ca, *cb;
  const size_t cn = (size_t)1 << 28;                              // 4 GB in, 4 GB out per copy launch
  hipMalloc(&src, h.size() * 2); hipMalloc(&zero, h.size() * 2); hipMalloc(&out, nblk * 512 * 4);
  hipMalloc(&ca, cn * 16); hipMalloc(&cb, cn * 16);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemset(zero, 0, h.size() * 2); hipMemset(ca, 1, cn * 16);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  double w0 = 0, c0 = 0; smi(w0, c0);
  printf("idle: %.0f W, %.0f MHz\n", w0, c0);
  float *src32, *zero32;
  std::vector<float> h32(8 * 16384);
  for (auto &v : h32) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  hipMalloc(&src32, h32.size() * 4); hipMalloc(&zero32, h32.size() * 4);
  hipMemcpy(src32, h32.data(), h32.size() * 4, hipMemcpyHostToDevice); hipMemset(zero32, 0, h32.size() * 4);
  const char *names[] = {"bf16 32x32x16, random operands", "bf16 32x32x16, all-zero operands", "bf16 random operands + HBM copy stream", "HBM copy stream alone",
                         "fp32 32x32x2, random operands", "fp32 32x32x2, all-zero operands"};
  for (int cs = 0; cs < 6; cs++) {
    std::atomic<bool> stop{false};
    std::vector<double> ws, cs_;
    std::thread th([&] { while (!stop) { double w, c; if (smi(w, c)) { ws.push_back(w); cs_.push_back(c); } std::this_thread::sleep_for(std::chrono::milliseconds(200)); } });
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0, copies = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      if (cs < 3) for (int i = 0; i < 4; i++) { mfma_k<<<nblk, 512, 0, s1>>>(cs == 1 ? zero : src, out, iters); launches++; }
      if (cs >= 4) for (int i = 0; i < 4; i++) { mfma32_k<<<nblk, 512, 0, s1>>>(cs == 5 ? zero32 : src32, out, iters); launches++; }
      if (cs == 2 || cs == 3) { copy_k<<<2048, 256, 0, s2>>>(ca, cb, cn); copies++; }
      hipStreamSynchronize(s1); hipStreamSynchronize(s2);
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true; th.join();
    double w = 0, c = 0; size_t n = 0;
    for (size_t i = 2; i < ws.size(); i++) { w += ws[i]; c += cs_[i]; n++; }
    const double flop = (double)launches * nblk * 8.0 * iters * 8.0 * (cs >= 4 ? 4096.0 : 32768.0);
    printf("%-40s %7.1f TFLOP/s  %6.2f TB/s copied (read + write)  %5.0f W  %5.0f MHz  (%zu samples)\n", names[cs], flop / el / 1e12,
           copies * cn * 32.0 / el / 1e12, n ? w / n : 0.0, n ? c / n : 0.0, n);
  }
  return 0;
}
