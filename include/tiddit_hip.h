/*
 * tiddit_hip.h — C ABI of libtiddit_hip.so, the MI355X (gfx950) implementation of TIDDIT's
 * signal-aggregation and clustering hot path.
 *
 * The reference (SciLifeLab/TIDDIT v3.9.5) has no FFI: its boundary for this path is the Python
 * module-function level.  Each entry point below names the reference interface
 * it replaces (file:line under tiddit/); the tiddit_amd python modules bind them with ctypes and re-exports the
 * reference's own function names/signatures (see INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions
 *   - plain C: opaque handles, pointers + sizes, no C++/torch types.
 *   - every function returns 0 (TDT_OK) or a negative TDT_E_* code; tdt_last_error() gives the
 *     message of the last failure on the calling thread.
 *   - `h_*` / unprefixed pointers are caller-owned HOST memory; `d_*` pointers are caller-owned
 *     DEVICE memory on the context's GPU.  Calls taking only d_* pointers are asynchronous on the
 *     context stream (tdt_ctx_stream); calls that return results to host memory synchronise.
 *   - coordinates are 0-based; `end` is exclusive (htslib bam_endpos / pysam reference_end).
 *   - there is NO CPU fallback: without a usable GPU tdt_ctx_create fails.
 */
#ifndef TIDDIT_HIP_H
#define TIDDIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDT_OK 0
#define TDT_E_ARG (-1)      /* bad argument (null pointer, bin_size <= 0, m < 2, ...)              */
#define TDT_E_HIP (-2)      /* a HIP runtime call failed (message in tdt_last_error)               */
#define TDT_E_RANGE (-3)    /* a read indexes a bin outside its contig (reference: IndexError)     */
#define TDT_E_INEXACT (-4)  /* a bin left the exact-arithmetic domain (|acc| >= 2^53)              */
#define TDT_E_NOMEM (-5)
#define TDT_E_UNSUPPORTED (-6) /* input outside the device path's domain (e.g. coordinate span >= 2^32) */
#define TDT_E_KEY (-7)      /* a name that is not in the table (reference: KeyError); tdt_last_error() = the name */

typedef struct tdt_ctx tdt_ctx;
typedef struct tdt_cov tdt_cov;

/* ---- context ------------------------------------------------------------------------------ */
int tdt_version(void);
/* The measurement macros the library was compiled with, space separated ("" = the product build).  Ablation builds (tools/build_variant.sh)
 * exist to price parts of a kernel and may compute wrong results; tiddit_amd._native refuses to load a library that reports any unless
 * TIDDIT_ALLOW_VARIANT=1. */
const char *tdt_build_flags(void);
const char *tdt_last_error(void);
int tdt_device_count(int *count);
int tdt_ctx_create(int device, tdt_ctx **out);
void tdt_ctx_destroy(tdt_ctx *ctx);
int tdt_ctx_sync(tdt_ctx *ctx);
/* The hipStream_t all asynchronous work of this context is enqueued on (as void*). */
void *tdt_ctx_stream(tdt_ctx *ctx);
/* Enqueue work on an externally owned hipStream_t (e.g. torch's current stream); 0 restores the
 * context's own stream. */
int tdt_ctx_set_stream(tdt_ctx *ctx, void *hip_stream);
/* Make the context's device the calling thread's current one: for a helper thread of the caller that pins host memory (tdt_host_alloc)
 * before it has called anything else of the library (bamio's file-reading thread). */
int tdt_ctx_bind_thread(tdt_ctx *ctx);

/* ---- binned read-depth histogram ---------------------------------------------------------- *
 * Replaces tiddit_coverage.create_coverage (tiddit_coverage.pyx:10-21), the per-read
 * update_coverage calls (tiddit_coverage.pyx:48-74) driven by __main__.py:229-242 (--cov) and
 * tiddit_signal.pyx:169-182 (--sv), including their read filter
 *     keep iff !(flag & 0x4) && !(flag & 0x400) && mapq >= min_q .
 * Bins are bit-identical to the reference's float64 arrays: every per-read contribution
 * float32(bases)/float32(den) is added as an exact 2^-S fixed-point int64 (S from tdt_cov_scale_bits). */
int tdt_cov_create(tdt_ctx *ctx, const int64_t *contig_len, int n_contigs, int bin_size, tdt_cov **out);
void tdt_cov_destroy(tdt_cov *cov);
/* bins = ceil(LN/bin_size); end_bin_size = LN - (bins-1)*bin_size      (tiddit_coverage.pyx:15-17) */
int tdt_cov_nbins(tdt_cov *cov, int tid, int64_t *nbins, int *end_bin_size);
int tdt_cov_scale_bits(tdt_cov *cov);
/* zero all accumulators (a fresh create_coverage) */
int tdt_cov_reset(tdt_cov *cov);
/* Add n alignment records of contig `tid`.  Host arrays are staged through pinned memory and copied
 * with hipMemcpyAsync (double-buffered), the kernel applies the read filter on device. */
int tdt_cov_push(tdt_cov *cov, int tid, const int32_t *start, const int32_t *end, const uint8_t *mapq,
                 const uint16_t *flag, size_t n, int min_q);
/* Same with device-resident arrays (asynchronous). */
int tdt_cov_push_device(tdt_cov *cov, int tid, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq,
                        const uint16_t *d_flag, size_t n, int min_q);
/* Several contigs' device-resident streams in ONE launch (no inter-launch gaps or tails): item i adds
 * n[i] records of contig tids[i]; the pointer arrays are HOST arrays of device pointers. */
int tdt_cov_push_device_multi(tdt_cov *cov, int n_items, const int *tids, const int32_t *const *d_start,
                              const int32_t *const *d_end, const uint8_t *const *d_mapq, const uint16_t *const *d_flag,
                              const size_t *n, int min_q);
/* The same with PACKED records — 8 bytes per read instead of 11: low word = reference_start (int32), high word =
 * span:24 | min(mapq,63):6 | unmapped(0x4):1 | duplicate(0x400):1, span = reference_end - reference_start (0xffffff: the true end
 * is read from d_end[i], which may otherwise be NULL).  This is the layout a producer that already sits on the device (the ingest
 * kernel) hands over; tdt_cov_pack_device converts the four arrays.  min_q > 63 is TDT_E_UNSUPPORTED here. */
int tdt_cov_pack_device(tdt_ctx *ctx, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq, const uint16_t *d_flag, size_t n,
                        uint64_t *d_packed);
int tdt_cov_push_packed_device_multi(tdt_cov *cov, int n_items, const int *tids, const uint64_t *const *d_packed, const int32_t *const *d_end,
                                     const size_t *n, int min_q);
/* BINNED records — 8 bytes per read, made for THIS histogram's bin size (2 <= bin_size < 1024, else TDT_E_UNSUPPORTED): low word =
 * first_bin << 2 | shape (0 one bin, 1 two bins [up to 256 for bins <= 128 bp], 2 replay literally, 3 invalid), high word = the filter
 * byte of the packed record | the two table indices bases_first_bin / bases_last_bin of tiddit_coverage.pyx:53-63 (csrc/tdt_common.h:
 * cov_bin_record).  The division, the bin split and the validity checks are done once when the record is written — by the ingest kernel
 * (tdt_ingest_bin_for) or by tdt_cov_pack_binned_device from the four arrays — and leave the accumulation launch.  d_start / d_end serve
 * the reads replayed literally (shape 2) and must be given. */
int tdt_cov_pack_binned_device(tdt_cov *cov, int tid, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq, const uint16_t *d_flag,
                               size_t n, uint64_t *d_binned);
int tdt_cov_push_binned_device_multi(tdt_cov *cov, int n_items, const int *tids, const uint64_t *const *d_binned, const int32_t *const *d_start,
                                     const int32_t *const *d_end, const size_t *n, int min_q);
/* All contigs at once: d_out holds tdt_cov_total_bins doubles, contig tid starts at tdt_cov_offset(tid)
 * (contigs are padded to 16-byte boundaries). */
int tdt_cov_total_bins(tdt_cov *cov, int64_t *total);
int tdt_cov_offset(tdt_cov *cov, int tid, int64_t *off);
int tdt_cov_finish_all_device(tdt_cov *cov, double *d_out);
/* the same into HOST memory: out = float64[tdt_cov_total_bins], contig t at out + tdt_cov_offset(t) — one launch, one copy, one wait for
 * a header of thousands of contigs (tiddit_coverage.pyx:10-21 creates one array per @SQ line) */
int tdt_cov_finish_all(tdt_cov *c, double *out);
/* Convert contig tid's accumulators to float64 bins (exact), copy to host / leave on device.
 * Returns TDT_E_RANGE / TDT_E_INEXACT if any pushed read / bin violated the domain. */
int tdt_cov_finish(tdt_cov *cov, int tid, double *out_bins);
int tdt_cov_finish_device(tdt_cov *cov, int tid, double *d_out_bins);
/* Number of reads that passed the filter so far (all contigs); synchronises. */
int tdt_cov_kept(tdt_cov *cov, int64_t *kept);

/* ---- binned GC / N-mask histogram ---------------------------------------------------------- *
 * Replaces tiddit_gc.binned_gc's per-character loop (tiddit_gc.pyx:14-31): out[bin] = -1 if
 * n/bin_size > n_cutoff else round_half_even(100*gc/chars); chars = bytes actually in the bin. */
int tdt_gc_bins(tdt_ctx *ctx, const uint8_t *seq, int64_t len, int bin_size, double n_cutoff, int8_t *out);
/* d_seq must be 16-byte aligned; asynchronous. */
int tdt_gc_bins_device(tdt_ctx *ctx, const uint8_t *d_seq, int64_t len, int bin_size, double n_cutoff, int8_t *d_out);
/* The same bins straight from the FASTA bytes of a contig, line ends still in place (what pysam.FastaFile.fetch hides,
 * tiddit_gc.pyx:14-19): raw = the nbytes the contig occupies in the file, len bases, linebases / linewidth as in the
 * .fai index.  TDT_E_UNSUPPORTED for layouts the kernel does not take (bins > 2048, contigs >= 2^31 bases, line ends
 * longer than two bytes): strip the line ends on the host and call tdt_gc_bins then. */
int tdt_gc_bins_fasta(tdt_ctx *ctx, const uint8_t *raw, int64_t nbytes, int64_t len, int linebases, int linewidth, int bin_size,
                      double n_cutoff, int8_t *out);
int tdt_gc_bins_fasta_device(tdt_ctx *ctx, const uint8_t *d_raw, int64_t nbytes, int64_t len, int linebases, int linewidth, int bin_size,
                             double n_cutoff, int8_t *d_out);
/* tiddit_gc.main's loop (tiddit_gc.pyx:35-42) over MANY contigs in one call: `raw` = the FASTA bytes of n contigs, contig i at raw_off[i]
 * (a multiple of 16) for raw_len[i] bytes, len[i] bases at linebases[i] / linewidth[i] per line; its int8 bins are written to
 * out + out_off[i].  One copy in, one launch per contig, one copy out, one wait (a GRCh38-shaped reference has 3 366 contigs). */
int tdt_gc_bins_fasta_many(tdt_ctx *ctx, const uint8_t *raw, int64_t nbytes, int n, const int64_t *raw_off, const int64_t *raw_len,
                           const int64_t *len, const int32_t *linebases, const int32_t *linewidth, int bin_size, double n_cutoff,
                           int8_t *out, const int64_t *out_off, int64_t out_bytes);

/* ---- signal clustering ("DBSCAN") ----------------------------------------------------------- *
 * Replaces DBSCAN.x_coordinate_clustering (DBSCAN.py:33-64), y_coordinate_clustering (:66-123) and
 * main (:125-129).  `data` is the reference's row-major int64 [n, stride] array (column 0 = posA,
 * column 1 = posB) in the order the caller would hand to DBSCAN.main; labels come back as float64
 * in that same order, exactly as the reference returns them.  eps is compared like numpy compares
 * an int64 distance with the Python number: d < eps.
 *   mode 0: main (x then y);  mode 1: x pass only.  *last_id receives the final cluster_id. */
int tdt_dbscan(tdt_ctx *ctx, const int64_t *data, size_t n, size_t stride, double eps, int m, int mode,
               double *labels, int64_t *last_id);
/* DBSCAN.y_coordinate_clustering(data, eps, m, cluster_id, clusters) (DBSCAN.py:66-123) on caller-supplied x labels: `labels` holds
 * them on entry (float64: -1 or the cluster numbers 0, 1, 2, ... — every number one contiguous range, ascending along the array, as
 * x_coordinate_clustering returns them for any eps / m) and the relabelled result on return; sub-run 1 of a cluster keeps its label,
 * extra sub-runs get cluster_id + 1, cluster_id + 2, ...; *last_id = the returned cluster_id.  TDT_E_UNSUPPORTED for label arrays of
 * another shape, for a cluster_id below the largest label (the ids produced would collide with clusters not visited yet), for
 * clusters above 128 members and for m > 64: tdt_dbscan_y_segments takes those. */
int tdt_dbscan_y(tdt_ctx *ctx, const int64_t *data, size_t n, size_t stride, double eps, int m, int64_t cluster_id, double *labels,
                 int64_t *last_id);
/* The same pass for ARBITRARY label arrays (the reference selects a cluster's members by value, DBSCAN.py:68-75: any size, any
 * values, not necessarily contiguous).  The caller lists the members of all clusters it wants visited: y[i] (posB), seg[i] = the
 * cluster's position in the visiting order (0 .. nseg-1; the reference's order is Python's set() iteration, which the Python
 * front end supplies), keep[i] = the member's current label.  out[i] = keep[i] for sub-run 1, cluster_id + (extra sub-runs of the
 * segments visited before) + s - 1 for sub-run s > 1, -1 otherwise; *last_id = the final cluster_id (:112-122).  The segments are
 * treated as independent, which is what the reference does as long as no produced id equals a value still to be visited; callers
 * replay one segment per call otherwise (tiddit_amd/DBSCAN.py). */
int tdt_dbscan_y_segments(tdt_ctx *ctx, const int64_t *y, const int32_t *seg, const double *keep, size_t n, int nseg, double eps, int m,
                          int64_t cluster_id, double *out, int64_t *last_id);
int tdt_dbscan_y_device(tdt_ctx *ctx, const int32_t *d_xlab, const uint32_t *d_y, size_t n, uint64_t eps, int m, int64_t cluster_id,
                        double *d_labels, int64_t *d_last_id, int *too_large);
/* Device-resident batch: nb independent buckets ((chrA,chrB) pairs), bucket b owning points
 * [bucket_off[b], bucket_off[b+1]) of d_x/d_y (uint32 coordinates, each bucket in DBSCAN.main input
 * order).  bucket_off is a HOST array of nb+1 offsets.  Labels (float64, ids restart per bucket)
 * are written to d_labels; per-bucket final cluster_id to d_last_id (int64[nb], may be NULL).
 * Asynchronous unless large x-clusters need the segmented sort (one 8-byte readback). */
int tdt_dbscan_device(tdt_ctx *ctx, const uint32_t *d_x, const uint32_t *d_y, size_t n, const int64_t *bucket_off,
                      int nb, uint64_t eps, int m, int mode, double *d_labels, int64_t *d_last_id);
/* tiddit_cluster.pyx:152-154 in one call: stable sort of each bucket by posA, then DBSCAN.main.
 * perm_out[i] = index (within the whole input) of the point at sorted position i. */
int tdt_sort_dbscan(tdt_ctx *ctx, const int64_t *posA, const int64_t *posB, size_t n, const int64_t *bucket_off, int nb,
                    double eps, int m, uint32_t *perm_out, double *labels_out);
/* The same call, also reporting per bucket the number of x-runs (x_coordinate_clustering's cluster_id + 1, DBSCAN.py:33-64) in
 * runs_out[nb] and DBSCAN.main's final cluster_id in last_out[nb] (either may be NULL): what a caller that cut ONE oversized
 * bucket into pieces at gaps >= eps needs to re-base the pieces' ids (DBSCAN.py:112-122 numbers extra sub-runs after ALL x-runs). */
int tdt_sort_dbscan_ex(tdt_ctx *ctx, const int64_t *posA, const int64_t *posB, size_t n, const int64_t *bucket_off, int nb,
                       double eps, int m, uint32_t *perm_out, double *labels_out, int64_t *runs_out, int64_t *last_out);
/* The same for 32-bit columns (alignment positions are below 2^31), without any host pass over the data: labels come back as int32 in
 * SIGNAL order (labels_by_signal[i] = cluster of the i-th input signal, -1 = noise), 4 bytes per signal over PCIe.  max_pos bounds posA
 * (e.g. the longest contig; <= 0: unknown) so that only the digits that can differ are sorted.  Columns and labels in pinned memory
 * (tdt_host_alloc) are moved by DMA directly; pageable memory is staged.  runs_out / last_out as in tdt_sort_dbscan_ex (may be NULL). */
int tdt_cluster_columns(tdt_ctx *ctx, const int32_t *posA, const int32_t *posB, size_t n, const int64_t *bucket_off, int nb, double eps, int m,
                        int64_t max_pos, int32_t *labels_by_signal, int64_t *runs_out, int64_t *last_out);
int tdt_host_alloc(size_t bytes, void **out);     /* pinned host memory */
int tdt_host_free(void *p);

/* ---- multi-GPU exchange (one process per GPU, RCCL over xGMI) ------------------------------------- *
 * The clustering path shards by (chrA,chrB) bucket (tiddit_cluster.pyx:140-154 keeps no cross-bucket state): every rank
 * clusters its buckets with tdt_dbscan_device / tdt_sort_dbscan and ONE variable-count all-gather of the label arrays gives
 * every rank the whole cluster set.  When one BAM is read as byte-range shards (tdt_ingest_push_bounded) every rank bins
 * its reads into a full-genome histogram and ONE sum all-reduce of the float64 bins finishes it — the bins are multiples of
 * 2^-S below 2^53, so the sum is exact and bit-identical to the single-GPU result.
 * tdt_comm_unique_id: rank 0 obtains the 128-byte RCCL id and hands it to the other ranks by whatever channel launched them
 * (MPI, a file, torch.distributed); tdt_comm_init is collective over the `world` ranks, each with its own context/GPU.
 * tdt_allgatherv: rank r contributes counts[r] elements of elem_bytes; they arrive at d_recv + displs[r]*elem_bytes on every
 * rank (counts/displs are HOST arrays, identical on all ranks; counts[rank] == send_count).  Both calls are asynchronous on the
 * context stream.  RCCL is bound at run time (an instance already in the process, e.g. PyTorch's, is reused). */
typedef struct tdt_comm tdt_comm;
int tdt_comm_unique_id(uint8_t *id128);
int tdt_comm_init(tdt_ctx *ctx, const uint8_t *id128, int rank, int world, tdt_comm **out);
int tdt_comm_destroy(tdt_comm *comm);
int tdt_allgatherv(tdt_comm *comm, const void *d_send, size_t send_count, void *d_recv, const size_t *counts, const size_t *displs,
                   int elem_bytes);
int tdt_allreduce_sum_f64(tdt_comm *comm, double *d_buf, size_t n);
/* The broadcasts tdt_allgatherv issues, as a pure host function (no RCCL, no GPU): ops[k] = root rank, destination byte range in
 * d_recv, and whether this rank reads its send buffer (it is the root).  Empty contributions issue nothing; overlapping ranges and
 * ranges beyond recv_capacity_bytes (0 = not checked) are refused.  ops must hold `world` entries. */
typedef struct tdt_gather_op {
    int root;
    int from_send;
    size_t offset_bytes;
    size_t nbytes;
} tdt_gather_op;
int tdt_allgatherv_plan(int rank, int world, const size_t *counts, const size_t *displs, int elem_bytes, size_t recv_capacity_bytes,
                        tdt_gather_op *ops, int *n_ops);

/* ---- discordant-pair candidate selection ------------------------------------------------------ *
 * Replaces the per-read predicate chain of tiddit_signal.worker (tiddit_signal.pyx:171-211): a read is a
 * discordant-pair signal iff its contig is processed (contig_ok[tid], = LN >= min_contig) and it is mapped,
 * not duplicate, primary, mapq >= min_q, paired with a mapped mate, and |tlen| > max_ins or the mate lies on
 * another contig.  The indices of the selected reads are returned in stream order (wavefront compaction). */
int tdt_signal_select(tdt_ctx *ctx, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mate_tid,
                      const int32_t *tlen, size_t n, const uint8_t *contig_ok, int n_contigs, int min_q, int64_t max_ins,
                      uint32_t *out_idx, size_t *out_count);
int tdt_signal_select_device(tdt_ctx *ctx, const uint16_t *d_flag, const uint8_t *d_mapq, const int32_t *d_tid,
                             const int32_t *d_mate_tid, const int32_t *d_tlen, size_t n, const uint8_t *d_contig_ok, int n_contigs,
                             int min_q, int64_t max_ins, uint32_t *d_out_idx, uint64_t *d_count);

/* The WHOLE per-read chain of tiddit_signal.worker (tiddit_signal.pyx:171-221) on a decoded batch resident in HBM.  d_arrays14 =
 * the pointer table of tdt_ingest_arrays (tdt_bam_decode's field order, then the raw record bytes); contig_ok is a HOST array.
 * Every read gets an action byte — 2: clipped read for local assembly (:190-197), 4: carries an SA tag, SA_analysis is called
 * (:199-202), 8: a discordant-pair row is appended (:204-221) — and the reads with any bit set are compacted in stream order
 * with their fields and raw record bytes, so that the host touches only those.  tdt_signal_scan returns how many (*n_sel) and
 * their total record bytes; tdt_signal_scan_result copies them out: meta = n_sel records of 28 bytes {uint32 index in the
 * batch; int32 tid, pos, end, mate_tid, sa_rel (offset of the SA:Z string inside the record, -1 without); uint16 flag; uint8
 * action; uint8 0}, raw_end[k] = end offset of record k inside raw (record k starts at raw_end[k-1], its block_size field
 * first).  The result of the calling thread's last scan is kept until its next one. */
int tdt_signal_scan(tdt_ctx *ctx, const void *const *d_arrays14, size_t n, const uint8_t *contig_ok, int n_contigs, int min_q, int64_t max_ins,
                    int min_anchor_len, int min_clip_len, size_t *n_sel, size_t *raw_bytes);
int tdt_signal_scan_result(tdt_ctx *ctx, void *meta, uint32_t *raw_end, uint8_t *raw);
/* The clips.append entries of worker (:192-197) as text: for the selected records `which[0..m)` (indices into the arrays of
 * tdt_signal_scan_result) `>query_name|contig|pos+1\nSEQUENCE\n`, concatenated.  Two-call protocol like tdt_format_coverage. */
int tdt_format_clips(const void *meta, const uint32_t *raw_end, const uint8_t *raw, const uint32_t *which, size_t m, const char *contig,
                     char *out, size_t out_cap, size_t *out_len);

/* ---- signal tables (host, no Python in the per-row path) ------------------------------------------ *
 * Replaces, for the discordant / split / clipped reads tdt_signal_scan selected:
 *   - the merge loop of tiddit_signal.main (tiddit_signal.pyx:246-284: data[chrA][chrB][fragment], splits[..] +=) and its three
 *     writers (:298-332: discordants_{sample}.tab, splits_{sample}.tab, the clip FASTA files),
 *   - the signal table of tiddit_cluster.main (tiddit_cluster.pyx:47-105: find_discordant_pos :7-37, the clip QUIRK :67-70),
 *   - the per-row half of its regrouping (:156-254): which signals make up which candidate, in the reference's insertion order.
 * One table per job, used by one thread at a time.  names = the header's contig names, NUL-terminated, back to back, in header
 * order; contigs >= min_contig are main()'s `chromosomes`.  Byte-identical output for coordinate-sorted AND unsorted input (see
 * csrc/tdt_sigtab.hip).  TDT_E_KEY where the reference raises KeyError (a split row whose chrB is not in the header). */
int tdt_sigtab_create(const char *names, const int64_t *lengths, int n_contigs, int64_t min_contig, void **out);
void tdt_sigtab_destroy(void *t);
/* the arrays of tdt_signal_scan_result, batch by batch in file order.  *stopped == n_sel: done; else the index of a split read whose
 * SA tag tdt_split_fields does not take — hand its row over with tdt_sigtab_add_split_row (or drop it) and call again with
 * resume = *stopped + 1. */
int tdt_sigtab_add(void *t, const void *meta, const uint32_t *raw_end, const uint8_t *raw, size_t n_sel, size_t raw_len, int min_q, size_t resume,
                   size_t *stopped);
int tdt_sigtab_add_split_row(void *t, int tid, const char *chrA, const char *chrB, const char *qname, const int64_t *six, int is_reverse, int sa_minus);
int tdt_sigtab_add_clips(void *t, int tid, const char *bytes, size_t len);
int tdt_sigtab_clips(void *t, int tid, const char **ptr, size_t *len);      /* clips/{contig}.fa of one contig (valid until the table changes) */
int tdt_sigtab_stats(void *t, int64_t *out8);
/* the row log as one blob (N-rank job: rows travel to the owner rank of their chrA; owner == NULL: all rows) — two-call protocol */
int tdt_sigtab_export(void *t, const int32_t *owner, int dest, void *out, size_t cap, size_t *need);
int tdt_sigtab_import(void *t, const void *blob, size_t len);
int tdt_sigtab_format(void *t, size_t *n_segments_disc, size_t *n_segments_split);
/* kind 0 = discordants, 1 = splits; segments: 5 int64 per contig pair with rows (chrA id, chrB id, offset, length, rows) */
int tdt_sigtab_text(void *t, int kind, const char **ptr, size_t *len, int64_t *segments);
/* bytes per contig of one output — what 0: the rows of discordants_{sample}.tab with that chrA, 1: splits_{sample}.tab, 2: the contig's
 * clip FASTA — and one such block written at `offset` of an open file: a file is its blocks in header order, so N ranks place theirs */
int tdt_sigtab_sizes(void *t, int what, int64_t *out_per_contig);
int tdt_sigtab_pwrite(void *t, int what, int contig, int fd, int64_t offset);
int tdt_sigtab_cluster_table(void *t, int is_mp, int64_t min_contig, size_t *n_signals, int *n_buckets);
int tdt_sigtab_cluster_columns(void *t, int32_t *posA, int32_t *posB, int64_t *bucket_off, int32_t *bucket_a, int32_t *bucket_b);
int tdt_sigtab_regroup(void *t, const int32_t *labels, size_t *n_candidates, size_t *n_members, size_t *name_bytes);
int tdt_sigtab_regroup_result(void *t, int32_t *cand4, int32_t *startA, int32_t *endA, int32_t *startB, int32_t *endB, int32_t *posA, int32_t *posB,
                              uint8_t *oriA, uint8_t *oriB, char *names);

/* ---- masked medians of the coverage bins -------------------------------------------------------- *
 * Replaces the per-bin Python loop + numpy.median of determine_ploidy (tiddit_coverage_analysis.pyx:14-27).
 * cov/gc are the concatenated float64 bins / int8 GC bins; segment s = [seg_off[2s], seg_off[2s+1]) (segments may
 * overlap: per-contig segments plus one covering everything).  For every segment the selected values are
 * { cov[i] : cov[i] > 0 and gc[i] != -1 }; count[s] = how many, lower[s]/upper[s] = the two middle order statistics
 * (equal for odd counts) — numpy.median is their mean.  Radix select on the device, no sort. */
int tdt_masked_medians(tdt_ctx *ctx, const double *cov, const int8_t *gc, const int64_t *seg_off, int nseg, double *lower,
                       double *upper, int64_t *count);
/* The same for data in n_parts separate host arrays (determine_ploidy's per-contig arrays): results 0 .. n_parts-1 are the parts' medians,
 * result n_parts the median over all of them; every part is copied to the device from where it lies. */
int tdt_masked_medians_parts(tdt_ctx *ctx, const double *const *cov_parts, const int8_t *const *gc_parts, const int64_t *part_len, int n_parts,
                             double *lower, double *upper, int64_t *count);

/* ---- library statistics on the device (tiddit_stats.statistics, tiddit_stats.py:5-78) ------------------------- *
 * tdt_stats_push_device takes the decoded field arrays of one ingest batch (device pointers) and applies the sampling loop (:17-47):
 * placed reads are numbered across batches, the cut-off after n_reads is honoured to the read, the insert sizes of the passing
 * pairs are appended in order to a device list; *done = the sample is complete.  tdt_stats_counts: sampled, sum and count of the
 * read lengths, innie, outtie, number and sum of the insert sizes (+ 2 per-batch figures).  tdt_stats_moments: numpy's
 * mean((x - mean)^2) of the list (numpy.std = its square root; numpy's summation order reproduced) and the two order statistics
 * k0 <= k1 that numpy.percentile interpolates between (radix select).  Everything is bit-identical to the numpy calls (:52-56). */
typedef struct tdt_stats tdt_stats;
int tdt_stats_create(tdt_ctx *ctx, int64_t n_reads, int min_mapq, int64_t max_ins_len, tdt_stats **out);
int tdt_stats_destroy(tdt_stats *s);
int tdt_stats_push_device(tdt_stats *s, const int32_t *d_tid, const int32_t *d_pos, const int32_t *d_mate_tid, const int32_t *d_mate_pos,
                          const int32_t *d_tlen, const int32_t *d_l_seq, const uint16_t *d_flag, const uint8_t *d_mapq, size_t n, int *done);
int tdt_stats_counts(tdt_stats *s, int64_t *counts9);
int tdt_stats_moments(tdt_stats *s, double mean, int64_t k0, int64_t k1, double *mean_sqdev, int32_t *order0, int32_t *order1);

/* ---- region means of the coverage bins per SV candidate --------------------------------------------- *
 * Replaces the per-candidate numpy.average calls of tiddit_variant.define_variant (tiddit_variant.pyx:265-283: avg_a / avg_b over
 * the 50-bp bins [start/50, end/50]; :307-315: covM over the bins between the breakpoints with gc != -1).  cov / gc are the
 * concatenated float64 coverage bins / int8 GC bins; segment q = [seg_lo[q], seg_hi[q]) indexes them (seg_lo/seg_hi/masked are
 * HOST arrays in both entry points); masked[q] != 0 keeps only the bins with gc > -1 (order preserved).  mean[q] equals
 * numpy.average of that slice BIT FOR BIT (numpy's chunked pairwise summation order is reproduced), NaN for an empty one;
 * count[q] = number of bins averaged (the reference falls back to the contig's coverage when covM has <= 4 of them). */
int tdt_segment_means(tdt_ctx *ctx, const double *cov, const int8_t *gc, int64_t total, const int64_t *seg_lo, const int64_t *seg_hi,
                      const uint8_t *masked, size_t nq, double *mean, int64_t *count);
int tdt_segment_means_device(tdt_ctx *ctx, const double *d_cov, const int8_t *d_gc, const int64_t *seg_lo, const int64_t *seg_hi,
                             const uint8_t *masked, size_t nq, double *d_mean, int64_t *d_count);

/* ---- regional evidence counts per SV candidate --------------------------------------------------- *
 * Replaces the per-candidate BAM re-scan of tiddit_variant.get_region (tiddit_variant.pyx:54-151).  The arrays
 * are ONE contig's coordinate-sorted alignment records (has_sa[i] != 0 iff the record carries an SA tag, tid = the
 * contig's id for the `next_reference_name != reference_name` test).  Query q = (start, end, bp) as in the
 * reference call get_region(samfile, chr, start, end, bp, min_q, max_ins, ...).  out[q*7 + ...] =
 * bases, n_reads, low_q, n_discs, n_splits, crossing_f, crossing_r  (the reference then returns
 * coverage = bases/(end-start+1) and frac_low_q = low_q/n_reads). */
int tdt_region_counts(tdt_ctx *ctx, const int32_t *start, const int32_t *end, const uint8_t *mapq, const uint16_t *flag,
                      const int32_t *mate_tid, const int32_t *mate_pos, const int32_t *tlen, const uint8_t *has_sa, size_t n, int tid,
                      int64_t contig_length, const int32_t *q_start, const int32_t *q_end, const int32_t *q_bp, size_t nq, int min_q,
                      int64_t max_ins, int64_t *out);
int tdt_region_counts_device(tdt_ctx *ctx, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq, const uint16_t *d_flag,
                             const int32_t *d_mate_tid, const int32_t *d_mate_pos, const int32_t *d_tlen, const uint8_t *d_has_sa,
                             size_t n, int tid, int max_span, int64_t contig_length, const int32_t *d_q_start, const int32_t *d_q_end,
                             const int32_t *d_q_bp, size_t nq, int min_q, int64_t max_ins, int64_t *d_out);

/* ---- coverage table text (host, threaded) ----------------------------------------------------------- *
 * The row loop of print_coverage (tiddit_coverage.pyx:30-44) for one contig: kind 0 = bed rows
 * `name \t 1+i*bin \t (i+1)*bin+1 \t value \n` (last row ends at contig_len), kind 1 = wig values, one per line.
 * Values are written exactly like Python's `"{}".format(numpy.float64)`.  Two-call protocol: out == NULL returns the
 * size in *out_len; then call again with a buffer of at least that many bytes. */
/* The numeric half of tiddit_signal.SA_analysis (tiddit_signal.pyx:11-145) for the selected records `which` (those carrying an SA tag):
 * out[k] = {int32 status (0: SA mapQ below min_q, 1: valid, 2: unusual tag — run the literal code), read_start, read_end, split_pos,
 * sa_split (before the swap), seg_start, seg_end, uint32 chr_off, chr_len (the SA contig name inside raw), uint8 is_reverse, sa_minus,
 * 2 pad} — 40 bytes.  The contig-name order and the swap (:118-140) are string decisions left to the caller.  Host function. */
int tdt_split_fields(const void *meta, const uint32_t *raw_end, const uint8_t *raw, size_t raw_len, const uint32_t *which, size_t m, int min_q, void *out);
int tdt_format_coverage(const double *values, size_t n, const char *name, int64_t bin_size, int64_t contig_len, int kind, char *out,
                        size_t out_cap, size_t *out_len);

/* FASTA index: writes `fai_path` in samtools faidx format (name, bases, offset of the first base, bases per line, bytes per
 * line) — what the reference obtains from pysam.faidx when the .fai is missing (__main__.py:95-97). */
int tdt_fasta_write_fai(const char *fasta_path, const char *fai_path);

/* ---- BGZF inflate (host, threaded) ---------------------------------------------------------------- *
 * Replaces pysam/htslib's block reader (`pysam.AlignmentFile(bam, "r", threads=n)`, tiddit_signal.pyx:159,
 * __main__.py:224).  tdt_bgzf_scan hops the block headers of `comp[0..len)`: it reports how many WHOLE blocks are
 * present, their compressed size (`consumed`) and what they inflate to (`produced`, never more than max_out).
 * tdt_bgzf_inflate inflates exactly such a span (len = consumed, out_len = produced) on `threads` host threads
 * (<= 0: tdt_host_threads' value), every block straight to its final offset, CRC32 and ISIZE verified.
 * tdt_host_threads(n) sets the worker count for the host stages (n <= 0: query only) and returns the previous
 * value; default min(hardware threads, 64) or $TIDDIT_HOST_THREADS. */
int tdt_host_threads(int n);
int tdt_bgzf_scan(const uint8_t *comp, size_t len, size_t max_out, size_t *n_blocks, size_t *consumed, size_t *produced);
int tdt_bgzf_inflate(const uint8_t *comp, size_t len, uint8_t *out, size_t out_len, int threads);

/* Same contract as tdt_bgzf_inflate, but the blocks are inflated ON THE DEVICE (one wavefront per block, CRC32 and ISIZE
 * verified there): `comp` is a host pointer to whole BGZF blocks; `out` is a host buffer, or a device pointer when
 * out_on_device != 0 (the record decode and the histogram kernels then read it in place). */
int tdt_bgzf_inflate_hbm(tdt_ctx *ctx, const uint8_t *comp, size_t len, uint8_t *out, size_t out_len, int out_on_device);

/* ---- BAM ingest on the device ------------------------------------------------------------------------ *
 * The per-read attribute access of the reference's loops (`for read in samfile.fetch(until_eof=True)`,
 * __main__.py:229-240, tiddit_signal.pyx:169-221) as one call per batch of BGZF blocks: inflate, find the records,
 * decode the packed arrays — all in HBM.  n_ref = number of @SQ contigs (record sanity check).
 * tdt_ingest_push: `comp` = `len` bytes of WHOLE BGZF blocks (host memory, see tdt_bgzf_scan); `skip` = inflated bytes
 * in front of the first record (the BAM header; first call only).  The incomplete record at the end of the batch is
 * kept and prepended to the next call.  Records are located by parallel per-segment guesses that the host confirms
 * against the block_size chain; a batch that cannot be confirmed is chased serially on the host instead (counted in
 * tdt_ingest_carry's host_chases) — the result is the sequential decode either way.  After a push has returned an error
 * the stream position is undefined: further pushes on that object are refused.  tdt_ingest_arrays: device pointers, valid until the next push, in tdt_bam_decode's
 * output order (tid, pos, end, mapq, flag, mate_tid, mate_pos, tlen, l_seq, cigar_first, cigar_last, rec_off, sa_off)
 * followed by the batch's raw record bytes (rec_off / sa_off index into them).  tdt_ingest_edges: record indices
 * where the contig id changes (*n = (size_t)-1 when there are more than 8191, i.e. the input is not coordinate
 * sorted).  tdt_ingest_carry: bytes of the pending partial record (0 after a well-formed file). */
typedef struct tdt_ingest tdt_ingest;
int tdt_ingest_create(tdt_ctx *ctx, int n_ref, tdt_ingest **out);
int tdt_ingest_destroy(tdt_ingest *g);
int tdt_ingest_push(tdt_ingest *g, const uint8_t *comp, size_t len, size_t skip, size_t *n_records);
/* Sharded reads (one process per GPU on one file).  skip = (size_t)-1: the stream starts at a BGZF block somewhere
 * inside the file and the first record is located by the guess (*first_off = its inflated offset).  own_bytes: the
 * first own_bytes inflated bytes of this call belong to this shard, the blocks after them only complete its last
 * record; records starting at or beyond own_bytes are left to the next shard and *next_off = offset of the first of
 * them from that boundary.  The caller checks next_off of shard r against first_off of shard r+1: shard 0 starts from
 * the header (exact), so agreement at every seam makes the whole decode exact.  A bounded push ends the stream. */
int tdt_ingest_push_bounded(tdt_ingest *g, const uint8_t *comp, size_t len, size_t skip, size_t own_bytes, size_t *n_records,
                            size_t *first_off, size_t *next_off);
/* Optional overlap: start the host-to-device copy of the NEXT span (pinned memory) on the copy stream; the next push of
 * exactly this (pointer, len) finds it there, so the transfer hides behind the kernels of the push issued in between. */
int tdt_ingest_prefetch(tdt_ingest *g, const uint8_t *comp, size_t len);
int tdt_ingest_arrays(tdt_ingest *g, const void **out14, size_t *raw_len);
int tdt_ingest_edges(tdt_ingest *g, uint32_t *edges, size_t cap, size_t *n);
/* the contig id of the run that starts at edges[k], for every k tdt_ingest_edges reported (up to 8191 runs per batch: a GRCh38-shaped
 * file has ~1 900 contigs with reads) */
int tdt_ingest_edge_tids(tdt_ingest *g, int32_t *tids, size_t cap);
/* device pointer of the batch's PACKED coverage records (see tdt_cov_push_packed_device_multi), valid until the next push */
int tdt_ingest_packed(tdt_ingest *g, const uint64_t **d_packed);
/* From the next push on the reader writes BINNED records for `cov` (NULL: the generic packed records again) into the column
 * tdt_ingest_packed returns; *binned = 1 when it does (0: that histogram's bin size has no binned form, the column stays generic). */
int tdt_ingest_bin_for(tdt_ingest *g, tdt_cov *cov, int *binned);
/* Enqueue the FIRST HALF of a coming span's push — its copy (or the prefetched one), its block table, the inflate + CRC kernels and the
 * copy of their status word — on the reader's own (low-priority) inflate streams, without waiting for anything.  A span inflates into an
 * output buffer of its own, a fixed gap into it (1 MB; TIDDIT_INGEST_GAP), and the partial record the batch before it ends with is copied
 * in front of the output by the push itself (a record longer than the gap moves the output once): so a span's inflate depends on NOTHING
 * the launch stream holds and may be started as soon as the span is in host memory — before the push of the span in front of it, while
 * the current batch's consumers run; the chip then never leaves the inflate kernel, and the record search, field decode and the caller's
 * kernels of batch k run beside span k+1's inflate.  At most two spans may be begun beyond the current batch; their pushes must follow in
 * the same order with exactly these (pointer, length) pairs.  The current batch (field arrays AND raw bytes) stays valid until the next
 * push; tdt_ingest_retain may be called with spans begun ahead.  (htslib's reader threads decompress ahead of the consumer in the same way.) */
int tdt_ingest_push_ahead(tdt_ingest *g, const uint8_t *comp, size_t len);
/* Keep the current batch beyond the next push: its device buffers (everything tdt_ingest_arrays / tdt_ingest_packed returned) move into
 * *handle and stay valid until tdt_ingest_release; the reader continues with fresh buffers.  Used by `tiddit --sv` to scan the batches its
 * library statistics were sampled from without reading and inflating them a second time. */
/* Where the last push spent its time (ms): [0] BGZF block table (host), [1] host-to-device copy of the span (the prefetch copy when
 * [6] = 1: it ran on the copy stream behind the previous batch's kernels), [2] inflate + CRC kernels of the span (elapsed on its inflate
 * stream: it overlaps the previous span's tail and the launch stream's kernels), [3] record-finding kernels, [4] chain check (host),
 * [5] field decode kernel, [6] prefetched, [7] wall time of the push call itself (the first half of a span begun ahead lies outside
 * it).  The figures are those of the CURRENT batch: spans begun ahead since its push do not disturb them. */
int tdt_ingest_timing(tdt_ingest *g, double *out8);
typedef struct tdt_retained tdt_retained;
int tdt_ingest_retain(tdt_ingest *g, tdt_retained **handle);
int tdt_ingest_release(tdt_retained *handle);
int tdt_ingest_carry(tdt_ingest *g, size_t *bytes, size_t *host_chases);
/* The readers' device buffers (hundreds of MB to GB each) are kept in a per-device cache between readers and retained batches instead
 * of going back to the driver (bounded: a quarter of the device's memory, at most 64 GB; TIDDIT_INGEST_CACHE_MB=<n> sets it, 0 switches
 * it off).  The library returns the cache to the driver by itself when any of its own device allocations fails and when the device's
 * last context is destroyed; an application that shares the device with other allocators (PyTorch, RCCL) calls this between jobs.
 * *released (may be NULL) = bytes handed back.  No counterpart in the reference (htslib keeps its buffers on the host). */
int tdt_device_cache_flush(tdt_ctx *ctx, uint64_t *released);
uint64_t tdt_device_cache_bytes(tdt_ctx *ctx);
/* Test hook: the next n device allocations of the library are refused once, as if the driver were out of memory, and take the
 * library's own way out (the cache goes back to the driver, the allocation is tried again — also when the cache held nothing, so the
 * hook never turns an allocation into a failure).  n = 0: off. */
void tdt_debug_fail_next_malloc(int n);
int tdt_copy_to_host(tdt_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* Measurement aid (bench.py, tools/calib_stream.py): what a plain streaming read of `bytes` of device memory reaches on this device — every
 * lane four 16-byte loads in flight, nothing written; `workgroups_per_cu` workgroups of 256 threads per CU walk the buffer grid-stride
 * (blocked = 0) or each its own contiguous share (blocked = 1: the way cov_accumulate's workgroups own 256 KB of reads each); best and mean
 * launch time over `reps` launches after one untimed pass (HIP events on the context's stream).  The roofline object of the bench quotes
 * the best of a small sweep beside the data sheet's 8 TB/s. */
int tdt_calib_stream_read(tdt_ctx *ctx, const void *d_buf, size_t bytes, int reps, int workgroups_per_cu, int blocked, double *best_ms,
                          double *mean_ms);

/* ---- alignment-record decode (host) ---------------------------------------------------------- *
 * Replaces the per-read pysam attribute access that feeds the path (read.reference_start,
 * reference_end, mapq, flag, next_reference_id, next_reference_start, isize, cigartuples[0]/[-1],
 * has_tag("SA") — __main__.py:229-240, tiddit_signal.pyx:169-221).  `buf` is uncompressed BAM record
 * data (after the header), starting at a record boundary.  Decodes up to max_records whole records into
 * the caller's arrays (any output pointer may be NULL); *consumed = bytes of whole records decoded.
 * end = htslib bam_endpos(); cigar_first/cigar_last = raw (len<<4|op) words or 0xffffffff without CIGAR;
 * rec_off = byte offset of each record (at its block_size field); sa_off = offset of the SA:Z string or -1. */
int tdt_bam_decode(const uint8_t *buf, size_t len, size_t max_records, size_t *consumed, size_t *n_records,
                   int32_t *tid, int32_t *pos, int32_t *end, uint8_t *mapq, uint16_t *flag, int32_t *mate_tid,
                   int32_t *mate_pos, int32_t *tlen, int32_t *l_seq, uint32_t *cigar_first, uint32_t *cigar_last,
                   uint64_t *rec_off, int64_t *sa_off);

/* ---- library statistics (host) ------------------------------------------------------------------------ *
 * The sampling loop of tiddit_stats.statistics (tiddit_stats.py:17-47) over decoded field arrays: state[0] n_sampled, [1] sum of read
 * lengths, [2] read lengths counted, [3] innie, [4] outtie, [5] done (n_sampled > n_reads) carry over from batch to batch (zero them
 * first); the template lengths of the pairs that pass every test are appended to out_tlen (room for n), *n_out = how many. */
int tdt_stats_scan(const int32_t *tid, const int32_t *pos, const int32_t *mate_tid, const int32_t *mate_pos, const int32_t *tlen,
                   const int32_t *l_seq, const uint16_t *flag, const uint8_t *mapq, size_t n, int64_t n_reads, int min_mapq,
                   int64_t max_ins_len, int64_t *state, int32_t *out_tlen, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* TIDDIT_HIP_H */
