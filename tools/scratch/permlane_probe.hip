// what v_permlane32_swap / v_permlane16_swap return on gfx950 (run on the GPU box: hipcc --offload-arch=gfx950 permlane_probe.hip -o p && ./p)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out)
{
    const unsigned v = threadIdx.x;
    const v2u a = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    const v2u b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[threadIdx.x] = a.x; out[64 + threadIdx.x] = a.y; out[128 + threadIdx.x] = b.x; out[192 + threadIdx.x] = b.y;
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap .x", "permlane32_swap .y", "permlane16_swap .x", "permlane16_swap .y"};
    for (int q = 0; q < 4; q++) { printf("%s:", names[q]); for (int i = 0; i < 64; i++) printf(" %u", h[q * 64 + i]); printf("\n"); }
    return 0;
}
