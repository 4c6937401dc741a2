// Driver for the sanitizer builds of the host table code (tests/test_host_tables_sanitizers.py): every threaded entry point of
// csrc/host_tables.hip and csrc/host_amge_tables.hip on random ragged tables, with one thread and with six (the split threshold lowered
// so that these small tables are split), outputs compared entry for entry.  Built with -fsanitize=thread and with
// -fsanitize=address,undefined; a report fails the run.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <cstdarg>
#include <stdexcept>

#include "../../include/elag_hip.h"

namespace elag {                                   // what context.hip provides to the table code (declarations: csrc/common.hpp)
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string &m) { fprintf(stderr, "elag_hip error: %s\n", m.c_str()); }
void check_device_error() {}                       // (no device in this build)
[[noreturn]] void fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}
}

using V = std::vector<int64_t>;
static std::mt19937_64 rng(7);

static void ragged(int64_t nrow, int64_t ncol, int maxlen, V &ptr, V &idx)
{
    ptr.assign(1, 0); idx.clear();
    for (int64_t r = 0; r < nrow; r++) {
        const int len = (int)(rng() % (uint64_t)(maxlen + 1));
        std::vector<char> used((size_t)ncol, 0);
        for (int k = 0; k < len; k++) {
            int64_t c = (int64_t)(rng() % (uint64_t)ncol);
            if (used[(size_t)c]) continue;
            used[(size_t)c] = 1; idx.push_back(c);
        }
        ptr.push_back((int64_t)idx.size());
    }
}

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "call failed at line %d\n", __LINE__); exit(2); } } while (0)

struct Out { std::vector<V> ints; std::vector<std::vector<double>> dbls; std::vector<std::vector<int32_t>> i32; };

static Out run_all()
{
    Out o;
    // ---- table product (signed) and minimal intersection sets
    V ap, ai, bp, bi;
    ragged(6000, 900, 6, ap, ai); ragged(900, 700, 5, bp, bi);
    std::vector<double> as(ai.size()), bs(bi.size());
    for (auto &v : as) v = (rng() & 1) ? 1.0 : -1.0;
    for (auto &v : bs) v = (rng() & 1) ? 1.0 : -1.0;
    int64_t cap = 0;
    for (int64_t m : ai) cap += bp[(size_t)m + 1] - bp[(size_t)m];
    V op(6001), oi((size_t)cap); std::vector<double> os((size_t)cap); int64_t nnz = 0;
    CHECK(elag_hip_host_table_product(6000, ap.data(), ai.data(), as.data(), 900, bp.data(), bi.data(), bs.data(), 700, op.data(), oi.data(), os.data(), &nnz));
    oi.resize((size_t)nnz); os.resize((size_t)nnz);
    o.ints.push_back(op); o.ints.push_back(oi); o.dbls.push_back(os);
    V mis(700); std::vector<double> ori(700); int64_t nset = 0;
    CHECK(elag_hip_host_minimal_intersection_sets(6000, op.data(), oi.data(), os.data(), 700, nullptr, mis.data(), ori.data(), &nset));
    o.ints.push_back(mis); o.dbls.push_back(ori); o.ints.push_back(V(1, nset));
    // ---- lookup, expand, sort, pack
    V lp, li; ragged(3000, 500, 40, lp, li);
    V qrow, qid;
    for (int64_t r = 0; r < 3000; r++) for (int k = 0; k < 5; k++) { qrow.push_back(r); qid.push_back((int64_t)(rng() % 500)); }
    V pos(qrow.size());
    CHECK(elag_hip_host_lookup_find(3000, lp.data(), li.data(), 500, (int64_t)qrow.size(), qrow.data(), qid.data(), pos.data()));
    o.ints.push_back(pos);
    V starts(3000), counts(3000); int64_t total = 0;
    for (int64_t r = 0; r < 3000; r++) { starts[(size_t)r] = lp[(size_t)r]; counts[(size_t)r] = lp[(size_t)r + 1] - lp[(size_t)r]; total += counts[(size_t)r]; }
    V ei((size_t)total), eo((size_t)total);
    CHECK(elag_hip_host_expand_ranges(3000, starts.data(), counts.data(), total, ei.data(), eo.data()));
    o.ints.push_back(ei); o.ints.push_back(eo);
    V key(500); for (auto &k : key) k = (int64_t)(rng() % 3);
    V sorted = li;
    CHECK(elag_hip_host_sort_rows_by_key(3000, lp.data(), sorted.data(), 500, key.data()));
    o.ints.push_back(sorted);
    std::vector<double> T(500 * 3), packed((size_t)li.size() * 3);
    for (auto &v : T) v = (double)(rng() % 1000) / 7.0;
    CHECK(elag_hip_host_pack_cols(3000, lp.data(), li.data(), 500, 3, T.data(), packed.data()));
    o.dbls.push_back(packed);
    // ---- block families -> CSR (overlapping blocks), and the element-dof table
    V rp, ri, cp, ci; ragged(2500, 400, 5, rp, ri); ragged(2500, 300, 4, cp, ci);
    V ld(2500); int64_t nv = 0;
    for (int64_t e = 0; e < 2500; e++) { ld[(size_t)e] = cp[(size_t)e + 1] - cp[(size_t)e]; nv += (rp[(size_t)e + 1] - rp[(size_t)e]) * ld[(size_t)e]; }
    std::vector<double> vals((size_t)nv);
    for (auto &v : vals) v = (double)((int)(rng() % 5) - 2);
    elag_hip_block_family fam{2500, rp.data(), ri.data(), cp.data(), ci.data(), vals.data(), nv, ld.data(), nullptr, 0};
    V indptr(401); int64_t knnz = 0;
    CHECK(elag_hip_host_block_families_to_csr(1, &fam, 400, 300, indptr.data(), nullptr, nullptr, nullptr));
    std::vector<int32_t> indices((size_t)indptr[400]); std::vector<double> data((size_t)indptr[400]);
    CHECK(elag_hip_host_block_families_to_csr(1, &fam, 400, 300, indptr.data(), indices.data(), data.data(), &knnz));
    indices.resize((size_t)knnz); data.resize((size_t)knnz);
    o.ints.push_back(indptr); o.i32.push_back(indices); o.dbls.push_back(data);
    // ---- the agglomerate tables on a made-up level: 4000 elements in 500 agglomerates, 6 dofs per element out of 3000
    const int64_t nE = 4000, nAE = 500, ndof = 3000;
    V aep(nAE + 1), els(nE), ep(nE + 1), eidx;
    for (int64_t a = 0; a <= nAE; a++) aep[(size_t)a] = a * 8;
    for (int64_t e = 0; e < nE; e++) els[(size_t)e] = e;
    ep[0] = 0;
    for (int64_t e = 0; e < nE; e++) {
        const int64_t base = (e / 8) * 6;                        // the elements of an agglomerate share a window of dofs
        for (int k = 0; k < 6; k++) eidx.push_back((base + (int64_t)(rng() % 12)) % ndof);
        ep[(size_t)e + 1] = (int64_t)eidx.size();
    }
    V sep(ndof); for (auto &s : sep) s = (int64_t)(rng() % 4 == 0);
    V adp(nAE + 1), nint(nAE), adofs(eidx.size()), loc(eidx.size());
    CHECK(elag_hip_host_amge_dof_agglomeration(nAE, aep.data(), els.data(), ep.data(), eidx.data(), ndof, sep.data(), adp.data(), nint.data(), adofs.data(), loc.data()));
    adofs.resize((size_t)adp[(size_t)nAE]);
    o.ints.push_back(adp); o.ints.push_back(nint); o.ints.push_back(adofs); o.ints.push_back(loc);
    V fptr(2001), iptr(nE + 1), efp, efi;
    for (int64_t f = 0; f <= 2000; f++) fptr[(size_t)f] = f;
    for (int64_t e = 0; e <= nE; e++) iptr[(size_t)e] = e / 2;
    ragged(nE, 2000, 6, efp, efi);
    V tptr(nE + 1);
    CHECK(elag_hip_host_amge_elem_dof_table(nE, efp.data(), efi.data(), fptr.data(), iptr.data(), 2000, tptr.data(), nullptr));
    V tidx((size_t)tptr[(size_t)nE]);
    CHECK(elag_hip_host_amge_elem_dof_table(nE, efp.data(), efi.data(), fptr.data(), iptr.data(), 2000, tptr.data(), tidx.data()));
    o.ints.push_back(tptr); o.ints.push_back(tidx);
    return o;
}

int main()
{
    elag_hip_host_set_min_work(64);
    elag_hip_host_set_threads(1);
    rng.seed(7);
    const Out one = run_all();
    elag_hip_host_set_threads(6);
    rng.seed(7);
    const Out many = run_all();
    if (one.ints != many.ints || one.dbls != many.dbls || one.i32 != many.i32) { fprintf(stderr, "results depend on the thread count\n"); return 3; }
    printf("HOST_TABLES_THREADS_OK %zu integer tables, %zu value tables\n", one.ints.size() + one.i32.size(), one.dbls.size());
    return 0;
}
