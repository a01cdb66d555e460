// Does a 16-byte buffer / global load from a 4-byte-aligned (not 16-byte-aligned) address return the four dwords at that
// address on this chip?   hipcc --offload-arch=gfx950 -O2 tools/micro/unaligned_b128.hip -o /tmp/ua && /tmp/ua
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *src, unsigned *out_buf, unsigned *out_glb, int n) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n * 4, 0x00020000);
  const int i = threadIdx.x;                       // dword offset i: 0..63
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)i * 4u, 0, 0);
  u32x4 g;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(src + i) : "memory");
  for (int e = 0; e < 4; e++) { out_buf[i * 4 + e] = a[e]; out_glb[i * 4 + e] = g[e]; }
}
int main() {
  const int n = 1024;
  unsigned h[n], *d, *ob, *og, rb[256], rg[256];
  for (int i = 0; i < n; i++) h[i] = i;
  (void)hipMalloc(&d, n * 4); (void)hipMalloc(&ob, 1024); (void)hipMalloc(&og, 1024);
  (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, ob, og, n);
  (void)hipMemcpy(rb, ob, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(rg, og, 1024, hipMemcpyDeviceToHost);
  int badb = 0, badg = 0;
  for (int i = 0; i < 64; i++) for (int e = 0; e < 4; e++) { badb += rb[i * 4 + e] != (unsigned)(i + e); badg += rg[i * 4 + e] != (unsigned)(i + e); }
  printf("buffer_load_b128 unaligned: %s (%d wrong)   global_load_dwordx4 unaligned: %s (%d wrong)\n", badb ? "WRONG" : "ok", badb, badg ? "WRONG" : "ok", badg);
  printf("lane 1 buffer: %u %u %u %u   lane 5 buffer: %u %u %u %u\n", rb[4], rb[5], rb[6], rb[7], rb[20], rb[21], rb[22], rb[23]);
  return 0;
}
