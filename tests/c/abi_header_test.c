/* The drop-in boundary is a C ABI: this file is plain C99 (no C++), includes the public header, and calls entry points that
 * need no device -- the host integer tables -- plus the error channel.  Built and run by tests/test_abi_load.py with gcc. */
#include "elag_hip.h"

#include <stdio.h>
#include <string.h>

int main(void)
{
    /* MultOrientation of the element -> edge and edge -> vertex tables of one triangle: the boundary of a boundary is empty */
    const int64_t a_ptr[2] = {0, 3}, a_idx[3] = {0, 1, 2};
    const double a_sgn[3] = {1.0, 1.0, -1.0};
    const int64_t b_ptr[4] = {0, 2, 4, 6}, b_idx[6] = {0, 1, 1, 2, 0, 2};
    const double b_sgn[6] = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0};
    int64_t out_ptr[2], out_idx[6], nnz = -1;
    double out_sgn[6];
    if (elag_hip_host_table_product(1, a_ptr, a_idx, a_sgn, 3, b_ptr, b_idx, b_sgn, 3, out_ptr, out_idx, out_sgn, &nnz) != 0 || nnz != 0) {
        printf("FAILED: signed product (%s)\n", elag_hip_last_error());
        return 1;
    }
    /* the pattern product lists the vertices in first-touch order */
    if (elag_hip_host_table_product(1, a_ptr, a_idx, NULL, 3, b_ptr, b_idx, NULL, 3, out_ptr, out_idx, NULL, &nnz) != 0 || nnz != 3 ||
        out_idx[0] != 0 || out_idx[1] != 1 || out_idx[2] != 2) {
        printf("FAILED: pattern product\n");
        return 1;
    }
    {
        const int64_t starts[2] = {4, 10}, counts[2] = {2, 1};
        int64_t idx[3], owner[3];
        if (elag_hip_host_expand_ranges(2, starts, counts, 3, idx, owner) != 0 || idx[0] != 4 || idx[1] != 5 || idx[2] != 10 || owner[2] != 1) {
            printf("FAILED: expand_ranges\n");
            return 1;
        }
        /* a wrong total is refused with a message, not a crash */
        if (elag_hip_host_expand_ranges(2, starts, counts, 2, idx, owner) == 0 || strstr(elag_hip_last_error(), "add up") == NULL) {
            printf("FAILED: error channel\n");
            return 1;
        }
    }
    printf("C_ABI_OK\n");
    return 0;
}
