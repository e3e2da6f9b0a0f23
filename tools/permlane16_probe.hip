// Probe of v_permlane16_swap_b32 on gfx950: prints, per lane, the two operands after the swap (a = lane, b = 100 + lane before).
// Build: hipcc --offload-arch=gfx950 -O2 tools/permlane16_probe.hip -o tools/permlane16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    int a = threadIdx.x, b = 100 + threadIdx.x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
}
int main() {
    int* d; int h[128];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int r = 0; r < 4; ++r) { printf("a row %d:", r); for (int i = 0; i < 16; ++i) printf(" %3d", h[r * 16 + i]); printf("\n"); }
    for (int r = 0; r < 4; ++r) { printf("b row %d:", r); for (int i = 0; i < 16; ++i) printf(" %3d", h[64 + r * 16 + i]); printf("\n"); }
    return 0;
}
