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

/* ------------------------------------------------------------------------------------------
 * BAM record walk (SAM/BAM spec v1 §4.2; what pysam's AlignedSegment attributes expose):
 *   reference_start = pos, reference_end = htslib bam_endpos (pos + sum of M/D/N/=/X lengths, pos + 1 when that is
 *   0 or the read is unmapped), mapq, flag, next_reference_id, next_reference_start, isize, l_seq,
 *   cigartuples[0] / [-1] (raw len<<4|op words, 0xffffffff without CIGAR), has_tag("SA") (aux walk).
 * PARITY UNPINNED at this boundary (pysam/htslib are not installable here): written from the
 * specification, independent of the product's decoders (csrc/tdt_bam.hip, csrc/tdt_ingest.hip).
 * `raw` = inflated BAM bytes from the first record on.  Returns the number of whole records decoded (<= max).
 * ---------------------------------------------------------------------------------------- */
static int32_t orc_i32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
static uint32_t orc_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t orc_u16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

int64_t orc_bam_walk(const uint8_t *raw, int64_t len, int64_t max, int32_t *tid, int32_t *pos, int32_t *end, uint8_t *mapq,
                     uint16_t *flag, int32_t *mate_tid, int32_t *mate_pos, int32_t *tlen, int32_t *l_seq, uint32_t *cig_first,
                     uint32_t *cig_last, int64_t *rec_off, uint8_t *has_sa) {
    int64_t o = 0, n = 0;
    while (o + 4 <= len && n < max) {
        int64_t bs = orc_i32(raw + o);
        if (bs < 32 || o + 4 + bs > len) break;
        const uint8_t *r = raw + o + 4;
        int l_name = r[8], n_cig = orc_u16(r + 12);
        int32_t ls = orc_i32(r + 16);
        tid[n] = orc_i32(r); pos[n] = orc_i32(r + 4); mapq[n] = r[9]; flag[n] = orc_u16(r + 14); l_seq[n] = ls;
        mate_tid[n] = orc_i32(r + 20); mate_pos[n] = orc_i32(r + 24); tlen[n] = orc_i32(r + 28);
        const uint8_t *c = r + 32 + l_name;
        int64_t rlen = 0;
        for (int k = 0; k < n_cig; k++) {
            uint32_t w = orc_u32(c + 4 * k);
            int op = w & 0xf;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += w >> 4;
        }
        cig_first[n] = n_cig ? orc_u32(c) : 0xffffffffu;
        cig_last[n] = n_cig ? orc_u32(c + 4 * (n_cig - 1)) : 0xffffffffu;
        end[n] = pos[n] + (int32_t)((rlen && !(flag[n] & 0x4)) ? rlen : 1);
        rec_off[n] = o;
        /* aux fields: tag[2] type[1] value */
        const uint8_t *a = c + 4 * (int64_t)n_cig + (ls + 1) / 2 + ls, *stop = r + bs;
        uint8_t sa = 0;
        while (a + 3 <= stop) {
            if (a[0] == 'S' && a[1] == 'A') sa = 1;
            char t = (char)a[2];
            a += 3;
            if (t == 'A' || t == 'c' || t == 'C') a += 1;
            else if (t == 's' || t == 'S') a += 2;
            else if (t == 'i' || t == 'I' || t == 'f') a += 4;
            else if (t == 'Z' || t == 'H') { while (a < stop && *a) a++; a++; }
            else if (t == 'B') {
                char st = (char)a[0];
                int64_t cnt = orc_u32(a + 1);
                int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                a += 5 + cnt * sz;
            } else break;
        }
        has_sa[n] = sa;
        o += 4 + bs;
        n++;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * The per-read chain of tiddit_signal.worker — tiddit_signal.pyx:169-221 — for the reads of ONE contig in file order
 * (`for read in samfile.fetch(chromosome, until_eof=True)`, :169), statement by statement; the string work (names,
 * sequences, SA_analysis, row building) stays with the caller, which gets one action byte per read:
 *   bit 0  update_coverage was called (:181-182; done here, on `cov`)     bit 1  clips.append (:190-197)
 *   bit 2  SA_analysis is called (:199-202)                               bit 3  data.append, a discordant row (:211-221)
 * mate_tid/tid stand for next_reference_name/reference_name (names are unique per id).  max_ins is the `int max_ins`
 * argument (:147).  Returns the number of update_coverage calls, or -(i+1) when read i indexes a bin out of range.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_signal_worker(int64_t n, const int32_t *tid, const int32_t *pos, const int32_t *end, const uint8_t *mapq,
                          const uint16_t *flag, const int32_t *mate_tid, const int32_t *tlen, const uint32_t *cig_first,
                          const uint32_t *cig_last, const uint8_t *has_sa, int min_q, int max_ins, int min_anchor_len,
                          int min_clip_len, int bin_size, double *cov, int64_t nbins, int end_bin_size, uint8_t *action) {
    int64_t updates = 0;
    for (int64_t i = 0; i < n; i++) {
        uint8_t act = 0;
        action[i] = 0;
        if ((flag[i] & 0x4) || (flag[i] & 0x400)) continue;                       /* :171 is_unmapped or is_duplicate */
        int read_mapq = mapq[i];
        if (read_mapq >= min_q) {                                                 /* :181-182 */
            if (orc_update_coverage(pos[i], end[i], bin_size, cov, nbins, end_bin_size)) return -(i + 1);
            updates++;
            act |= 1;
        }
        action[i] = act;
        if ((flag[i] & 0x800) || (flag[i] & 0x100)) continue;                     /* :184 supplementary or secondary */
        if (read_mapq < min_q) continue;                                          /* :188 */
        int64_t isz = tlen[i] < 0 ? -(int64_t)tlen[i] : (int64_t)tlen[i];
        int same = mate_tid[i] == tid[i];
        if (isz < max_ins && same) {                                              /* :191 */
            uint32_t f = cig_first[i], l = cig_last[i];
            int f_op = f & 0xf, l_op = l & 0xf;
            int64_t f_len = f >> 4, l_len = l >> 4;
            if ((f_op == 4 && f_len > min_clip_len) && (l_op == 0 && l_len > min_anchor_len)) act |= 2;          /* :193 */
            else if (l_op == 4 && l_len > min_clip_len && (f_op == 0 && f_len > min_anchor_len)) act |= 2;       /* :196 */
        }
        if (has_sa[i]) act |= 4;                                                  /* :199 */
        action[i] = act;
        if (flag[i] & 0x8) continue;                                              /* :204 mate_is_unmapped */
        if (!(flag[i] & 0x1)) continue;                                           /* :207 not is_paired */
        if (isz > max_ins || !same) act |= 8;                                     /* :211 */
        action[i] = act;
    }
    return updates;
}

/* ------------------------------------------------------------------------------------------
 * numpy.average of a contiguous float64 slice — tiddit_variant.pyx:265-283 (avg_a, avg_b: mean of the 50-bp coverage bins of
 * [start/50, end/50]) and :307-315 (covM: mean of the bins between the breakpoints whose gc != -1).  numpy.average(a) is
 * a.mean() = numpy.add.reduce(a) / len(a); the reduction of a contiguous float64 array is numpy's PAIRWISE summation
 * (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum_DOUBLE): blocks of at most 128 elements are summed with eight
 * interleaved accumulators, larger ranges are split in halves (the first a multiple of 8); the array is fed to that routine
 * in chunks of 8192 elements.  Restated here so that the
 * device kernel can be held to numpy's exact result; tests compare BOTH with numpy itself (third-party dependency of the
 * reference, importable everywhere).
 * ---------------------------------------------------------------------------------------- */
static double orc_pairwise(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise(a, n2) + orc_pairwise(a + n2, n - n2);
    }
}

/* mean of a[0..n) the way numpy.add.reduce + true_divide produce it; NaN for n == 0 (numpy.average of an empty slice).
 * The reduction runs over the ufunc machinery's buffer-size chunks of 8192 elements: the result starts at the identity 0.0 and
 * every chunk's pairwise sum is added in order (probed against numpy 2.2 on views of every alignment: this, and only this,
 * decomposition matches on all of them). */
double orc_np_mean(const double *a, int64_t n) {
    if (n <= 0) return NAN;
    double res = 0.0;
    for (int64_t o = 0; o < n; o += 8192) res += orc_pairwise(a + o, n - o < 8192 ? n - o : 8192);
    return res / (double)n;
}

/* covM's masked mean: the elements with gc > -1, compacted in order (boolean indexing), then numpy.average; *kept = how many */
double orc_np_masked_mean(const double *a, const int8_t *gc, int64_t n, int64_t *kept) {
    double *tmp = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++)
        if (gc[i] > -1) tmp[k++] = a[i];
    *kept = k;
    double m = orc_np_mean(tmp, k);
    free(tmp);
    return m;
}
