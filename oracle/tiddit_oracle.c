/*
 * tiddit_oracle.c — CPU restatement of the TIDDIT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a plain-C restatement of the reference algorithm, function by
 * function, each citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (tiddit_amd/) never does and
 * fails loudly when the HIP library is missing.
 *
 * Pinning: every function here is checked (tests/test_oracle.py) against golden vectors that
 * tests/golden/make_golden.py produced by running the REAL reference (cythonized
 * tiddit_coverage.pyx / tiddit_gc.pyx / tiddit_cluster.pyx and DBSCAN.py from /root/reference)
 * in the build container — known-answer tables, random property cases and the sha256 of the
 * config-1 coverage stream and the 100k/1M-point DBSCAN labels.
 *
 * Build: gcc -O2 -shared -fPIC (NO -ffast-math: the float32 divide below must be IEEE).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * create_coverage  — tiddit_coverage.pyx:10-21
 *   bins = int(ceil(LN / float(bin_size)));  end_bin_size = LN - (bins-1)*bin_size
 * ---------------------------------------------------------------------------------------- */
void orc_create_coverage(int64_t LN, int64_t bin_size, int64_t *nbins, int64_t *end_bin_size) {
    int64_t bins = (int64_t)ceil((double)LN / (double)bin_size);
    *nbins = bins;
    *end_bin_size = LN - (bins - 1) * bin_size;
}

/* ------------------------------------------------------------------------------------------
 * update_coverage — tiddit_coverage.pyx:48-74.  Literal, including:
 *   - `cdef float bases_*`: the quotient is float32(bases)/float32(den), widened to double (:53-57)
 *   - the single-bin branch always divides by bin_size (:55-57)
 *   - bases_last_bin = (ref_end-1) - end_bin*bin_size, one less than the true overlap (:63)
 *   - the contig's last bin divides by end_bin_size (:66-69); middle bins get +1.0 (:71-72)
 * Returns 0, or -1 when a bin index is out of range (the reference raises IndexError there).
 * ---------------------------------------------------------------------------------------- */
int orc_update_coverage(int64_t ref_start, int64_t ref_end, int bin_size, double *cov, int64_t nbins,
                        int end_bin_size) {
    int first_bin = (int)(ref_start / bin_size);
    int end_bin = (int)((ref_end - 1) / bin_size);
    float bases_first_bin;
    if (first_bin < 0 || end_bin < 0 || first_bin >= nbins || end_bin >= nbins) return -1;
    if (end_bin == first_bin) {
        bases_first_bin = (float)(ref_end - ref_start);
        cov[first_bin] = (double)(bases_first_bin / (float)bin_size) + cov[first_bin];
        return 0;
    }
    bases_first_bin = (float)(((int64_t)(first_bin + 1) * bin_size) - ref_start);
    cov[first_bin] = (double)(bases_first_bin / (float)bin_size) + cov[first_bin];
    float bases_last_bin = (float)((ref_end - 1) - (int64_t)end_bin * bin_size);
    if (end_bin < nbins - 1)
        cov[end_bin] = (double)(bases_last_bin / (float)bin_size) + cov[end_bin];
    else
        cov[end_bin] = (double)(bases_last_bin / (float)end_bin_size) + cov[end_bin];
    for (int i = first_bin + 1; i < end_bin; i++) cov[i] = 1.0 + cov[i];
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The per-read loop that feeds update_coverage:
 *   --cov : __main__.py:229-242   keep iff !unmapped(0x4) && !duplicate(0x400) && mapq >= q
 *   --sv  : tiddit_signal.pyx:169-182   same predicate (q = -q, bin 50)
 * Secondary / supplementary alignments ARE counted.  Reads are visited in stream order.
 * Returns the number of reads kept, or -(i+1) if read i indexes out of range.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_coverage_stream(const int32_t *start, const int32_t *end, const uint8_t *mapq, const uint16_t *flag,
                            int64_t n, int bin_size, int min_q, double *cov, int64_t nbins, int end_bin_size) {
    int64_t kept = 0;
    for (int64_t i = 0; i < n; i++) {
        if ((flag[i] & 0x4) || (flag[i] & 0x400)) continue;
        if ((int)mapq[i] >= min_q) {
            if (orc_update_coverage(start[i], end[i], bin_size, cov, nbins, end_bin_size)) return -(i + 1);
            kept++;
        }
    }
    return kept;
}

/* ------------------------------------------------------------------------------------------
 * binned_gc — tiddit_gc.pyx:6-33.  Literal per-character loop:
 *   number_of_bins = ceil(len/bin) (:9); per bin slice [start, start+bin) clipped to the contig
 *   n  += char in {N,n} (:22-23);  gc += char in {C,c,G,g} (:24-25);  chars += 1 (:21)
 *   out = -1 if n/bin_size > n_cutoff (true division, divides by bin_size not chars, :27)
 *         else round(100*gc/chars)  (Python round = half-to-even on the double, :30)
 * ---------------------------------------------------------------------------------------- */
int64_t orc_binned_gc(const uint8_t *seq, int64_t len, int64_t bin_size, double n_cutoff, int8_t *out) {
    int64_t nbins = (int64_t)ceil((double)len / (double)bin_size);
    int64_t next_start = 0;
    for (int64_t b = 0; b < nbins; b++) {
        int64_t stop = next_start + bin_size;
        if (stop > len) stop = len;
        int64_t n = 0, gc = 0, chars = 0;
        for (int64_t i = next_start; i < stop; i++) {
            uint8_t c = seq[i];
            chars++;
            if (c == 'N' || c == 'n') n++;
            else if (c == 'C' || c == 'c' || c == 'G' || c == 'g') gc++;
        }
        if ((double)n / (double)bin_size > n_cutoff) out[b] = -1;
        else out[b] = (int8_t)nearbyint((double)(100 * gc) / (double)chars); /* FE_TONEAREST = half-even */
        next_start += bin_size;
    }
    return nbins;
}

/* ------------------------------------------------------------------------------------------
 * x_coordinate_clustering — DBSCAN.py:33-64.  Literal state machine.  data is row-major
 * [n, stride] int64, column 0 = x.  eps is compared as a double exactly like numpy does when
 * epsilon is a Python float (int64 distance < float64 eps); integer eps are exact in double.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_x_clustering(const int64_t *data, int64_t n, int64_t stride, double eps, int64_t m, double *clusters) {
    for (int64_t i = 0; i < n; i++) clusters[i] = -1;
    int64_t cluster_id = -1;
    int cluster = 0;
    for (int64_t i = 0; i < n - m + 1; i++) {
        int64_t cur = data[i * stride];
        int64_t hi = i + m + 1;
        if (hi > n) hi = n; /* slice data[i+1:i+m+1] truncates at the array end */
        int64_t maxd = -1;
        int has = 0;
        for (int64_t j = i + 1; j < hi; j++) {
            int64_t d = data[j * stride] - cur;
            if (d < 0) d = -d;
            if (!has || d > maxd) maxd = d;
            has = 1;
        }
        if (!has) return INT64_MIN; /* max([]) raises ValueError in the reference (m == 1) */
        if ((double)maxd < eps) {
            if (cluster) clusters[i + m - 1] = (double)cluster_id;
            else {
                cluster_id += 1;
                cluster = 1;
                for (int64_t j = i; j < i + m; j++) clusters[j] = (double)cluster_id;
            }
        } else cluster = 0;
    }
    return cluster_id;
}

typedef struct { int64_t y; int64_t idx; } ypair;
static int ypair_cmp(const void *a, const void *b) {
    const ypair *p = (const ypair *)a, *q = (const ypair *)b;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx); /* list.sort(key=y) is stable: ties keep index order */
}

/* One x-cluster's y pass — DBSCAN.py:76-122 (members already gathered in index order). */
static int64_t y_pass_cluster(ypair *yc, int64_t k, double eps, int64_t m, double cluster, int64_t cluster_id,
                              double *clusters, double *sub) {
    qsort(yc, (size_t)k, sizeof(ypair), ypair_cmp);
    for (int64_t i = 0; i < k; i++) sub[i] = -1;
    int active = 0;
    int64_t sub_id = 0;
    for (int64_t i = 0; i < k - m + 1; i++) {
        int64_t maxd = -1;
        int has = 0;
        for (int64_t j = i + 1; j < i + m; j++) { /* next = y[i+1:i+m] — window of m-1 */
            int64_t d = yc[j].y - yc[i].y;
            if (d < 0) d = -d;
            if (!has || d > maxd) maxd = d;
            has = 1;
        }
        if (!has) return INT64_MIN;
        if ((double)maxd < eps) {
            if (active) sub[i + m - 1] = (double)sub_id;
            else {
                sub_id += 1;
                active = 1;
                for (int64_t j = i; j < i + m; j++) sub[j] = (double)sub_id;
            }
        } else active = 0;
    }
    for (int64_t i = 0; i < k; i++) {
        if (sub[i] == 1) clusters[yc[i].idx] = cluster;
        else if (sub[i] > -1) clusters[yc[i].idx] = sub[i] + (double)cluster_id - 1;
        else clusters[yc[i].idx] = -1;
    }
    return sub_id;
}

/* ------------------------------------------------------------------------------------------
 * y_coordinate_clustering — DBSCAN.py:66-123, LITERAL: for every id in set(clusters) (ascending
 * small-int floats, -1 skipped) build the full-array mask `clusters == cluster` (:72) — O(K*N),
 * which is where the reference spends its time.  `ids` = the distinct non-negative labels present
 * BEFORE the pass, ascending (computed by the caller exactly like set() sees them).
 * ---------------------------------------------------------------------------------------- */
int64_t orc_y_clustering_literal(const int64_t *data, int64_t n, int64_t stride, double eps, int64_t m,
                                 int64_t cluster_id, double *clusters) {
    /* distinct labels present now (the set is built once, before any mutation) */
    int64_t maxid = -1;
    for (int64_t i = 0; i < n; i++) if (clusters[i] > (double)maxid) maxid = (int64_t)clusters[i];
    char *present = (char *)calloc((size_t)(maxid + 2), 1);
    for (int64_t i = 0; i < n; i++) if (clusters[i] >= 0) present[(int64_t)clusters[i]] = 1;
    ypair *yc = (ypair *)malloc(sizeof(ypair) * (size_t)(n ? n : 1));
    double *sub = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    for (int64_t c = 0; c <= maxid; c++) {
        if (!present[c]) continue;
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++) /* class_member_mask = (clusters == cluster) */
            if (clusters[i] == (double)c) { yc[k].y = data[i * stride + 1]; yc[k].idx = i; k++; }
        int64_t sub_id = y_pass_cluster(yc, k, eps, m, (double)c, cluster_id, clusters, sub);
        if (sub_id == INT64_MIN) { cluster_id = INT64_MIN; break; }
        if (sub_id > 1) cluster_id += sub_id - 1;
    }
    free(present); free(yc); free(sub);
    return cluster_id;
}

/* ------------------------------------------------------------------------------------------
 * Same pass with the membership mask replaced by one linear sweep: x-clusters produced by
 * orc_x_clustering are contiguous index ranges with ascending ids, so a
 * single left-to-right walk visits the clusters in set() order.  Identical output (asserted in
 * tests against the literal version and the golden labels); O(N log s) — used for large inputs
 * and as the cpu_baseline "port".  Returns INT64_MIN+1 if the labels are not contiguous/ascending.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_y_clustering_sweep(const int64_t *data, int64_t n, int64_t stride, double eps, int64_t m,
                               int64_t cluster_id, double *clusters) {
    ypair *yc = (ypair *)malloc(sizeof(ypair) * (size_t)(n ? n : 1));
    double *sub = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    double *orig = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    memcpy(orig, clusters, sizeof(double) * (size_t)n);
    double last = -1;
    int64_t i = 0;
    while (i < n) {
        if (orig[i] < 0) { i++; continue; }
        double c = orig[i];
        if (c <= last) { cluster_id = INT64_MIN + 1; break; }
        last = c;
        int64_t k = 0;
        while (i < n && orig[i] == c) { yc[k].y = data[i * stride + 1]; yc[k].idx = i; k++; i++; }
        int64_t sub_id = y_pass_cluster(yc, k, eps, m, c, cluster_id, clusters, sub);
        if (sub_id == INT64_MIN) { cluster_id = INT64_MIN; break; }
        if (sub_id > 1) cluster_id += sub_id - 1;
    }
    free(yc); free(sub); free(orig);
    return cluster_id;
}

/* DBSCAN.main — DBSCAN.py:125-129.  literal != 0 selects the O(K*N) y pass. */
int64_t orc_dbscan_main(const int64_t *data, int64_t n, int64_t stride, double eps, int64_t m, double *clusters,
                        int literal) {
    int64_t id = orc_x_clustering(data, n, stride, eps, m, clusters);
    if (id == INT64_MIN) return id;
    return literal ? orc_y_clustering_literal(data, n, stride, eps, m, id, clusters)
                   : orc_y_clustering_sweep(data, n, stride, eps, m, id, clusters);
}

/* ------------------------------------------------------------------------------------------
 * get_region — tiddit_variant.pyx:54-151, literal loop over ONE contig's coordinate-sorted records.
 * PARITY UNPINNED at the fetch boundary (pysam absent): samfile.fetch(chr, q_start, q_end) is taken to
 * return the records with pos < q_end and bam_endpos > q_start, in file order.
 * out[7] = bases, n_reads, low_q, n_discs, n_splits, crossing_f, crossing_r.
 * ---------------------------------------------------------------------------------------- */
void orc_get_region(const int32_t *start, const int32_t *end, const uint8_t *mapq, const uint16_t *flag,
                    const int32_t *mate_tid, const int32_t *mate_pos, const int32_t *tlen, const uint8_t *has_sa, int64_t n,
                    int tid, int64_t contig_length, int64_t rstart, int64_t rend, int64_t bp, int min_q, int64_t max_ins,
                    int64_t *out) {
    int64_t low_q = 0, n_reads = 0, bases = 0, n_discs = 0, n_splits = 0, crossing_r = 0, crossing_f = 0;
    int64_t q_start = rstart, q_end = rend + max_ins;
    if (q_end > contig_length) q_end = contig_length;
    if (q_start >= q_end) q_start = q_end - 10;
    for (int64_t i = 0; i < n; i++) {
        if (!((int64_t)start[i] < q_end && (int64_t)end[i] > q_start)) continue; /* region fetch */
        if (flag[i] & 0x4) continue;
        int64_t rs = start[i];
        if (!(flag[i] & 0x8)) {
            if (mate_pos[i] > rend && rs > rend) continue;
        } else {
            if (rs > rend) continue;
        }
        if (flag[i] & 0x400) continue;
        if (!(rs > rend)) {
            n_reads++;
            if ((int)mapq[i] < min_q) low_q++;
        }
        if ((int)mapq[i] < min_q) continue;
        int64_t re = end[i];
        int64_t r_start = rs, r_end = re;
        if (rs < bp - 20 && r_end > bp + 20) crossing_r++;
        int mate_bp_read = (mate_pos[i] < bp - 50 && r_end > bp + 50);
        int64_t isz = tlen[i] < 0 ? -(int64_t)tlen[i] : tlen[i];
        int discordant = (isz > max_ins || mate_tid[i] != tid);
        if (mate_bp_read && !discordant) crossing_f++;
        if (re < rstart) continue;
        else if (rs > rend) continue;
        if (rs < rstart) r_start = rstart;
        if (re > rend) r_end = rend;
        bases += r_end - r_start + 1;
        if (has_sa[i]) n_splits++;
        if (discordant) n_discs++;
    }
    out[0] = bases; out[1] = n_reads; out[2] = low_q; out[3] = n_discs; out[4] = n_splits; out[5] = crossing_f; out[6] = crossing_r;
}
