#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
  int v = 1;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  out[256 + l] = v;
}
int main() {
  unsigned* d; hipMalloc(&d, 320 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[5] = {"swap32.r0", "swap32.r1", "swap16.r0", "swap16.r1", "rowscan"};
  for (int i = 0; i < 5; i++) { printf("%s:", names[i]); for (int l = 0; l < 64; l++) printf(" %u", h[i*64+l]); printf("\n"); }
  return 0;
}
