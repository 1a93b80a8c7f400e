// The 8-byte alignment records of the coverage path (csrc/tdt_coverage.hip reads them, csrc/tdt_ingest.hip and the packing kernels
// write them).  Kept in a header of their own: profiles/traffic.json records the hash of every source a kernel's counters were
// measured on, and these layouts are part of cov_accumulate.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

// ---- packed alignment record of the coverage path (8 B instead of 11), written by the ingest kernel next to the field arrays:
//   low word  = reference_start (int32)
//   high word = span:24 | min(mapq,63):6 | unmapped(0x4):1 | duplicate(0x400):1     span = reference_end - reference_start;
//               span 0xffffff = escape: the true end is read from the `end` array (reads spanning >= 16 Mb)
#define COV_PK_SPAN 0xffffffu
__host__ __device__ __forceinline__ unsigned long long cov_pack_record(int start, int end, unsigned mapq, unsigned flag) {
    const long long span = (long long)end - (long long)start;
    const unsigned sp = (span < 0 || span >= (long long)COV_PK_SPAN) ? COV_PK_SPAN : (unsigned)span;
    const unsigned info = sp | ((mapq > 63u ? 63u : mapq) << 24) | ((flag & 0x4u) ? 1u << 30 : 0u) | ((flag & 0x400u) ? 1u << 31 : 0u);
    return ((unsigned long long)info << 32) | (unsigned)start;
}

// ---- BINNED alignment record (8 B) for ONE histogram's bin size: everything about a read that does not depend on where the
// accumulating workgroup's window is, precomputed when the record is written (the ingest kernel, or tdt_cov_pack_binned_device):
//   low word  = first_bin << 2 | shape      first_bin = reference_start // bin_size (tiddit_coverage.pyx:50), clamped to the contig's bins
//       shape 0 (COV_BN_SINGLE)  the read lies in one bin (:53-57, always bases / bin_size — also in the contig's last bin)
//       shape 1 (COV_BN_MULTI)   first and last bin differ and the last bin is NOT the contig's last one (:59-66); histograms with
//                                bins >= 129 bp (cov_accumulate MODE 0) take only reads of exactly two bins here, the small-bin
//                                flavour (MODE 1) up to 255 bins after the first
//       shape 2 (COV_BN_SLOW)    any other valid read (more bins, or ending in the contig's last bin: other denominator, :67-69):
//                                replayed literally from the start / end arrays
//       shape 3 (COV_BN_INVALID) start < 0, end <= start or a bin beyond the contig: the reference raises IndexError when such a
//                                read passes the filter
//   high word = duplicate:1 | unmapped:1 | min(mapq,63):6 (top byte, as in cov_pack_record: the read filter is one range test)
//               | table indices: MODE 0  bases_last_bin:10 | bases_first_bin:10 | 0000   (the first-bin index already scaled to the
//                                        16-byte table entries: the accumulation launch masks it out with one AND)
//                                MODE 1  bins_after_first:8 | bases_last_bin:8 | bases_first_bin:8
//   bases_first_bin = end - start (single) or (first_bin+1)*bin_size - start; bases_last_bin = (end-1) - end_bin*bin_size, the
//   reference's one-short count (:63).
#define COV_BN_SINGLE 0u
#define COV_BN_MULTI 1u
#define COV_BN_SLOW 2u
#define COV_BN_INVALID 3u
struct CovBinSpec {
    unsigned z = 0;            // bin size; 0 = no binned records for this histogram (bin_size 1 or >= 1024)
    unsigned magic = 0;        // floor(x / z) = mulhi(x, magic) >> shift for 0 <= x < 2^31
    int shift = 0;
    int mode1 = 0;             // the small-bin kernel flavour's field layout
    const int *d_nbins = nullptr;   // bins of every contig (device array, by contig id)
    int n_contigs = 0;
};
__host__ __device__ __forceinline__ unsigned long long cov_bin_record(int start, int end, unsigned mapq, unsigned flag, int nbins, unsigned z,
                                                                      unsigned magic, int shift, bool mode1) {
    const unsigned top = ((mapq > 63u ? 63u : mapq) << 24) | ((flag & 0x4u) ? 1u << 30 : 0u) | ((flag & 0x400u) ? 1u << 31 : 0u);
    const int last_bin = nbins - 1;
    auto div = [&](int x) -> int { return (int)((unsigned)(((unsigned long long)(unsigned)x * magic) >> 32) >> shift); };
    unsigned shape = COV_BN_INVALID, fields = 0;
    int fb = 0;
    if (start >= 0 && end > start && nbins > 0) {
        fb = div(start);
        const int eb = div(end - 1);
        if (eb > last_bin) {
            fb = fb > last_bin ? last_bin : fb;
        } else if (eb == fb) {
            shape = COV_BN_SINGLE;
            fields = mode1 ? (unsigned)(end - start) : (unsigned)(end - start) << 4;
        } else if (eb >= last_bin || eb - fb > (mode1 ? 255 : 1)) {
            shape = COV_BN_SLOW;
        } else {
            shape = COV_BN_MULTI;
            const unsigned bf = (unsigned)(fb + 1) * z - (unsigned)start, bl = (unsigned)(end - 1) - (unsigned)eb * z;
            fields = mode1 ? (bf | (bl << 8) | ((unsigned)(eb - fb) << 16)) : ((bf << 4) | (bl << 14));
        }
    } else if (start >= 0 && nbins > 0) {
        fb = div(start);
        fb = fb > last_bin ? last_bin : fb;
    }
    return ((unsigned long long)(top | fields) << 32) | ((unsigned)fb << 2) | shape;
}

