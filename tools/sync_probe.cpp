// Microbenchmark: what does a device -> host scalar round trip cost?
//   (a) kernel; hipMemcpyAsync(4 B, D2H, pinned destination); hipStreamSynchronize
//   (b) kernel writes the scalar and then a sequence number into host-mapped pinned memory; the host spins on the sequence number
//   (c) like (a) but the kernel's result is copied by a second one-thread kernel into pinned memory, then hipStreamSynchronize
// build: hipcc --offload-arch=gfx950 -O2 tools/sync_probe.cpp -o gpurun_out/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void work(int *d, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *d = v; }
__global__ void work_publish(int *d, volatile int *host, int v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { *d = v; host[0] = v * 3; __threadfence_system(); host[16] = v; }
}
__global__ void publish(const int *d, volatile int *host, int seq)
{
    host[0] = *d; __threadfence_system(); host[16] = seq;
}
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 64));
    int *h; CK(hipHostMalloc(&h, 256, hipHostMallocDefault));
    for (int i = 0; i < 64; i++) h[i] = 0;
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int i = 1; i <= N; i++) {
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st, d, i);
            CK(hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (h[0] != i) { printf("bad a\n"); return 1; }
        }
        double a = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        t0 = now();
        for (int i = 1; i <= N; i++) {
            hipLaunchKernelGGL(work_publish, dim3(64), dim3(256), 0, st, d, (volatile int *)h, i + 100000 * (rep + 1));
            while (((volatile int *)h)[16] != i + 100000 * (rep + 1)) {}
            if (h[0] != 3 * (i + 100000 * (rep + 1))) { printf("bad b\n"); return 1; }
        }
        double b = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int i = 1; i <= N; i++) {
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st, d, i);
            hipLaunchKernelGGL(publish, dim3(1), dim3(1), 0, st, (const int *)d, (volatile int *)h, i + 500000 * (rep + 1));
            while (((volatile int *)h)[16] != i + 500000 * (rep + 1)) {}
            if (h[0] != i) { printf("bad c\n"); return 1; }
        }
        double c = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int i = 1; i <= N; i++) {
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, st, d, i);
            CK(hipStreamSynchronize(st));
        }
        double e = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        printf("rep %d: (a) copy + sync %.2f us   (b) kernel publishes, host spins %.2f us   (c) publish kernel, host spins %.2f us   (d) kernel + sync, no copy %.2f us\n", rep, a, b, c, e);
    }
    return 0;
}
