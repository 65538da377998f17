// Which XCD does workgroup (x, y) of a 2-D grid run on?  Reads HW_REG_XCC_ID in every workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = v & 0xf;
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  for (int gx : {60, 64, 61}) {
    int gy = 4;
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), 0, 0, d);
    int h[4096]; hipMemcpy(h, d, gx * gy * 4, hipMemcpyDeviceToHost);
    int ok_lin = 0, ok_x = 0;
    for (int i = 0; i < gx * gy; ++i) { ok_lin += h[i] == (i & 7); ok_x += h[i] == ((i % gx) & 7); }
    printf("grid (%d,%d): xcc == linear%%8 for %d/%d, xcc == x%%8 for %d/%d; first row:", gx, gy, ok_lin, gx * gy, ok_x, gx * gy);
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf(" | row 1:"); for (int i = 0; i < 8; ++i) printf(" %d", h[gx + i]);
    printf("\n");
  }
  return 0;
}
