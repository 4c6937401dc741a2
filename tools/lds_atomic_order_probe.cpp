// Are same-address LDS f64 atomic adds of ONE wavefront instruction applied in ascending lane order on gfx950?
// Every trial: 64 lanes hold values of wildly different magnitude, groups of G lanes add them to one LDS double per group with a single
// ds_add_f64; the result is compared bit for bit with the left-to-right sum over the group's lanes (and with the right-to-left one).
//   hipcc -O2 --offload-arch=gfx950 tools/lds_atomic_order_probe.cpp -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

template <int G>
__global__ void probe(const double *v, double *out, int *cas_out)
{
    __shared__ double acc[64];
    __shared__ int key[64];
    const int t = threadIdx.x, g = t / G;
    acc[t] = 0.0; key[t] = -1;
    __syncthreads();
    atomicAdd(&acc[g], v[blockIdx.x * 64 + t]);
    // who wins a same-slot compare-and-swap of one instruction?
    const int prev = atomicCAS(&key[g], -1, t);
    __syncthreads();
    if ((t % G) == 0) { out[blockIdx.x * 64 + g] = acc[g]; cas_out[blockIdx.x * 64 + g] = key[g]; }
    (void)prev;
}

template <int G>
int run(int trials)
{
    std::mt19937_64 rng(1234 + G);
    std::vector<double> v((size_t)trials * 64);
    for (auto &x : v) {
        const int e = (int)(rng() % 60) - 30;
        x = std::ldexp((double)(rng() % 2000001) / 1e6 - 1.0, e);
    }
    double *dv, *dout; int *dc;
    hipMalloc(&dv, v.size() * 8); hipMalloc(&dout, v.size() * 8); hipMalloc(&dc, v.size() * 4);
    hipMemcpy(dv, v.data(), v.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe<G>, dim3(trials), dim3(64), 0, 0, dv, dout, dc);
    std::vector<double> out(v.size()); std::vector<int> cas(v.size());
    hipMemcpy(out.data(), dout, v.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cas.data(), dc, v.size() * 4, hipMemcpyDeviceToHost);
    int asc = 0, desc = 0, other = 0, cas_first = 0, ngroups = 0;
    for (int b = 0; b < trials; b++)
        for (int g = 0; g < 64 / G; g++) {
            double up = 0.0, dn = 0.0;
            for (int l = 0; l < G; l++) up += v[(size_t)b * 64 + g * G + l];
            for (int l = G - 1; l >= 0; l--) dn += v[(size_t)b * 64 + g * G + l];
            const double r = out[(size_t)b * 64 + g];
            if (!memcmp(&r, &up, 8)) asc++; else if (!memcmp(&r, &dn, 8)) desc++; else other++;
            if (cas[(size_t)b * 64 + g] == g * G) cas_first++;
            ngroups++;
        }
    printf("G=%2d: %d groups: ascending-lane sum %d, descending only %d, neither %d; CAS won by the lowest lane %d\n", G, ngroups, asc, desc, other, cas_first);
    hipFree(dv); hipFree(dout); hipFree(dc);
    return other + desc;
}

int main()
{
    int bad = 0;
    bad += run<2>(4000); bad += run<4>(4000); bad += run<8>(4000); bad += run<16>(4000); bad += run<32>(4000); bad += run<64>(4000);
    printf(bad ? "NOT lane-ordered\n" : "lane-ordered in every trial\n");
    return 0;
}
