// Staging of caller vectors: ELAG_HIP_HOST pointers go through temporary device buffers.
#pragma once
#include "common.hpp"

namespace elag {

inline void check_loc(int loc) { ELAG_REQUIRE(loc == ELAG_HIP_HOST || loc == ELAG_HIP_DEVICE, "bad location flag %d", loc); }

struct VecIn {
    const double *d = nullptr;
    double *tmp = nullptr;
    VecIn(const double *p, size_t n, int loc)
    {
        if (loc == ELAG_HIP_DEVICE || p == nullptr) { d = p; return; }
        tmp = dev_alloc<double>(n);
        if (n) ELAG_HIP_CHECK(hipMemcpyAsync(tmp, p, n * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
        d = tmp;
    }
    ~VecIn() { if (tmp) { (void)hipStreamSynchronize(ctx().stream); dev_free(tmp); } }      // (dev_alloc's blocks go back through dev_free: they may be pooled)
};
struct VecInOut {
    double *d = nullptr, *host = nullptr, *tmp = nullptr;
    size_t n;
    VecInOut(double *p, size_t n_, int loc, bool copy_in) : n(n_)
    {
        if (loc == ELAG_HIP_DEVICE) { d = p; return; }
        host = p;
        tmp = dev_alloc<double>(n);
        if (copy_in && n) ELAG_HIP_CHECK(hipMemcpyAsync(tmp, p, n * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
        d = tmp;
    }
    void finish()
    {
        if (!tmp) return;
        if (n) ELAG_HIP_CHECK(hipMemcpyAsync(host, tmp, n * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
        ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    }
    ~VecInOut() { if (tmp) { (void)hipStreamSynchronize(ctx().stream); dev_free(tmp); } }
};


}  // namespace elag
