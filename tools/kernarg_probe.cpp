// Does libelag_hip.so's load-time default (HIP_FORCE_DEV_KERNARG=1 unless the user exported a value) reach the HIP runtime of a C++ host
// that merely links the library?  Times a chain of dependent launches of a tiny kernel with a 256-byte argument block.
//   hipcc -O2 tools/kernarg_probe.cpp -Iinclude -Lparelag_amd -lelag_hip -Wl,-rpath,$PWD/parelag_amd -o /tmp/kernarg_probe
//   /tmp/kernarg_probe ; HIP_FORCE_DEV_KERNARG=0 /tmp/kernarg_probe ; HIP_FORCE_DEV_KERNARG=1 /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "elag_hip.h"

struct Args { double *p; int n; double pad[30]; };
__global__ void touch(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += a.pad[3] + a.n; }

int main()
{
    const char *e = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG as main() sees it: %s\n", e ? e : "(unset)");
    if (elag_hip_init(0)) { printf("init failed: %s\n", elag_hip_last_error()); return 1; }
    double *d = nullptr;
    if (hipMalloc(&d, 64) != hipSuccess) return 1;
    (void)hipMemset(d, 0, 64);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Args a{d, 1, {0}};
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, a);
        (void)hipStreamSynchronize(st);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, st, a);
        const auto t1 = std::chrono::steady_clock::now();
        (void)hipEventRecord(e1, st);
        (void)hipStreamSynchronize(st);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  chain of 2000 dependent launches: device %.2f us per launch, host enqueue %.2f us per launch\n", ms * 1e3 / 2000,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    }
    elag_hip_finalize();
    return 0;
}
