// Segmented sort of unique 64-bit keys (library sort step; rocPRIM LSD radix).
//
// Used for (i) the stable sort by posA of each (chrA,chrB) bucket (tiddit_cluster.pyx:152) and
// (ii) the stable sort by posB inside x-clusters too large for the in-kernel rank sort
// (DBSCAN.py:79-81).  Stability is obtained by construction: keys are (coordinate << 32 | index),
// hence unique, so any correct sort yields the stable order.
#include "tdt_common.h"

#include <rocprim/rocprim.hpp>

int tdt_segsort_u64(tdt_ctx *ctx, int slot, const unsigned long long *d_in, unsigned long long *d_out, size_t n,
                    unsigned nseg, const unsigned *d_begin, const unsigned *d_end) {
    if (n == 0 || nseg == 0) return TDT_OK;
    size_t tmp = 0;
    TDT_HIP(rocprim::segmented_radix_sort_keys(nullptr, tmp, d_in, d_out, (unsigned)n, nseg, d_begin, d_end, 0, 64,
                                               ctx->stream));
    void *d_tmp = nullptr;
    int rc = tdt_scratch(ctx, slot, tmp ? tmp : 16, &d_tmp);
    if (rc) return rc;
    TDT_HIP(rocprim::segmented_radix_sort_keys(d_tmp, tmp, d_in, d_out, (unsigned)n, nseg, d_begin, d_end, 0, 64,
                                               ctx->stream));
    return TDT_OK;
}
