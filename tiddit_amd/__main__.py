"""``tiddit`` command line on the MI355X hot path: ``python -m tiddit_amd --cov ...`` / ``--sv ...``.

Same two modes and the same flags as the reference driver (tiddit/__main__.py:22-71, :211-219).
``--cov`` is complete (byte-identical .bed/.wig for the same alignments).  ``--sv`` runs the stages this
repository implements — library statistics, signal extraction + coverage (device), GC (device), ploidy,
clustering (device) — and writes the signal ``.tab`` files, ``{o}.ploidies.tab`` and
``{o}.candidates.tab``.  Variant typing / filtering / VCF (tiddit_variant.pyx) and local assembly are out
of scope: the run stops after the candidates table and says so.
"""
import argparse
import os
import sys
import time


def _sv_parser():
    p = argparse.ArgumentParser("""tiddit --sv --bam inputfile [-o prefix] --ref ref.fasta""")
    p.add_argument("--sv", help="call structural variation", required=False, action="store_true")
    p.add_argument("--force_overwrite", help="force the analysis and overwrite any data in the output folder", required=False, action="store_true")
    p.add_argument("--bam", type=str, required=True, help="coordinate sorted bam file(required)")
    p.add_argument("--ref", type=str, help="reference fasta", required=True)
    p.add_argument("-o", type=str, default="output", help="output prefix(default=output)")
    p.add_argument("-i", type=int, help="paired reads maximum allowed insert size (default= 99.9th percentile of insert size)")
    p.add_argument("-d", type=str, help="expected reads orientations, possible values \"innie\" (-> <-) or \"outtie\" (<- ->)")
    p.add_argument("-p", type=int, default=3, help="Minimum number of supporting pairs in order to call a variant (default 3)")
    p.add_argument("--threads", type=int, default=1, help="Number of threads (default=1)")
    p.add_argument("-r", type=int, default=3, help="Minimum number of supporting split reads to call a variant (default 3)")
    p.add_argument("-q", type=int, default=5, help="Minimum mapping quality to consider an alignment (default 5)")
    p.add_argument("-n", type=int, default=2, help="the ploidy of the organism,(default = 2)")
    p.add_argument("-e", type=int, help="clustering distance parameter (default = half the average insert size)")
    p.add_argument("-c", type=float, help="average coverage, overwrites the estimated average coverage")
    p.add_argument("-l", type=int, default=3, help="min-pts parameter (default=3),must be set >= 2")
    p.add_argument("-s", type=int, default=25000000, help="Number of reads to sample when computing library statistics(default=25000000)")
    p.add_argument("--force_ploidy", action="store_true", help="force the ploidy to be set to -n across the entire genome")
    p.add_argument("--n_mask", type=float, default=0.5, help="exclude regions from coverage calculation if they contain more than this fraction of N (default = 0.5)")
    p.add_argument("--p_ratio", type=float, default=0.1, help="minimum discordant pair/normal pair ratio at the breakpoint junction(default=0.1)")
    p.add_argument("--r_ratio", type=float, default=0.1, help="minimum split read/coverage ratio at the breakpoint junction(default=0.1)")
    p.add_argument("--max_coverage", type=float, default=4, help="filter call if X times higher than chromosome average coverage (default=4)")
    p.add_argument("--min_contig", type=int, default=10000, help="Skip calling on small contigs (default < 10000 bp)")
    p.add_argument("-z", type=int, default=50, help="minimum variant size (default=50)")
    p.add_argument("--skip_assembly", action="store_true", help="Skip running local assembly")
    p.add_argument("--bwa", type=str, default="bwa", help="path to bwa executable file(default=bwa)")
    p.add_argument("--min_clip", type=int, default=4, help="Minimum clip reads to initiate local assembly of a region(default=4)")
    p.add_argument("--padding", type=int, default=100, help="Extend the local assembly by this number of bases (default=100bp)")
    p.add_argument("--min_pts_clips", type=int, default=3, help="min-pts parameter for the clustering of candidates for local assembly (default=3)")
    p.add_argument("--max_assembly_reads", type=int, default=100000, help="Skip assembly of regions containing too many reads")
    p.add_argument("--max_local_assembly_region", type=int, default=2000, help="maximum size of the clip read cluster for local assembly")
    p.add_argument("--min_anchor_len", type=int, default=60, help="minimum mapped bases to be considered a clip read  (default=60 bp)")
    p.add_argument("--min_clip_len", type=int, default=25, help="minimum clipped bases to be considered a clip read (default=25 bp)")
    p.add_argument("--min_contig_len", type=int, default=200, help="minimum contig length for SV analysis (default=200 bp)")
    p.add_argument("-k", type=int, default=91, help="kmer lenght used by the local assembler (default=91 bp)")
    return p


def _cov_parser():
    p = argparse.ArgumentParser("""tiddit --cov --bam inputfile [-o prefix]""")
    p.add_argument("--cov", help="generate a coverage bed/wig file", required=False, action="store_true")
    p.add_argument("--bam", type=str, required=True, help="coordinate sorted bam file(required)")
    p.add_argument("-o", type=str, default="output", help="output prefix(default=output)")
    p.add_argument("-z", type=int, default=500, help="use bins of specified size(default = 500bp) to measure the coverage of the entire bam file")
    p.add_argument("-w", help="generate wig instead of bed", required=False, action="store_true")
    p.add_argument("-q", type=int, help="minimum mapping quality(default=20)", required=False, default=20)
    p.add_argument("--ref", type=str, help="reference fasta, used for reading cram")
    return p


def run_cov(args):
    from . import tiddit_coverage
    from .bamio import DeviceBatch, open_bam
    if not os.path.isfile(args.bam):
        print("error,  could not find the bam file")
        quit()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                                      # one process per GPU on ONE file: sharded ingest + all-reduce
        import torch
        import torch.distributed as dist
        from . import dist as tdist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group("nccl")
        bam_header, coverage_data, _ = tdist.coverage_sharded(args.bam, args.z, args.q)
        if dist.get_rank() != 0:
            return
    else:
        reader = open_bam(args.bam)                                        # inflate + record decode on the device by default
        bam_header = reader.header
        coverage_data, end_bin_size = tiddit_coverage.create_coverage(bam_header, args.z)
        hist = tiddit_coverage.CoverageHistogram(bam_header, args.z)
        if hasattr(reader, "bin_for"):
            reader.bin_for(hist)                 # the ingest kernel writes the coverage records for this bin size
        import numpy
        for b in reader.batches():
            if isinstance(b, DeviceBatch):
                hist.push_device_batch(b, args.q)
                reader.ahead()                   # (the launch reads the batch's coverage records, not its raw bytes: the next span's inflate may follow it)
                continue
            tid = b.tid
            edges = numpy.flatnonzero(numpy.diff(tid)) + 1
            for lo, hi in zip(numpy.concatenate([[0], edges]), numpy.concatenate([edges, [len(tid)]])):
                if tid[lo] >= 0:
                    hist.push(int(tid[lo]), b.pos[lo:hi], b.end[lo:hi], b.mapq[lo:hi], b.flag[lo:hi], args.q)
        reader.close()
        if len(coverage_data) > 64:          # a header of thousands of contigs: every contig's bins in one launch + copy (tdt_cov_finish_all)
            allbins = hist.finish_all()
            for i, contig in enumerate(coverage_data):
                o = hist.offset(i)
                coverage_data[contig] = allbins[o:o + hist.nbins(i)[0]].copy()
        else:
            for contig in coverage_data:
                coverage_data[contig] = hist.finish(contig)
        hist.close()
    if args.w:
        tiddit_coverage.print_coverage(coverage_data, bam_header, args.z, "wig", args.o + ".wig")
    else:
        tiddit_coverage.print_coverage(coverage_data, bam_header, args.z, "bed", args.o + ".bed")


STAGE_SECONDS = {}          # wall seconds of the last run_sv, stage by stage (bench.py reads it)
STAGE_NOTES = {}            # ... and counts that are not seconds


def write_candidates(path, contigs, sv_clusters):
    with open(path, "w") as f:
        f.write("#chrA\tposA\tchrB\tposB\tcluster\tN_discordants\tN_splits\tN_contigs\tstartA\tendA\tstartB\tendB\n")
        for chrA in contigs:
            if chrA not in sv_clusters:
                continue
            for chrB in sv_clusters[chrA]:
                for cid, c in sv_clusters[chrA][chrB].items():
                    f.write("\t".join(map(str, [chrA, c["posA"], chrB, c["posB"], cid, c["N_discordants"], c["N_splits"], c["N_contigs"],
                                                c["startA"], c["endA"], c["startB"], c["endB"]])) + "\n")


def variant_stage(tiddit_variant, tiddit_vcf_header, prefix, contigs, bam_header, library, sample_id, version, args, sv_clusters, min_mapq, samples,
                  coverage_data, contig_number, max_ins_len, gc_dictionary):
    """the tail of the reference's driver (__main__.py:193-207) with the two modules handed in: header, variants per contig sorted by
    position, {prefix}.vcf.  -> False (nothing written) when the reference package is not there."""
    if tiddit_variant is None or tiddit_vcf_header is None:
        return False
    vcf_header = tiddit_vcf_header.main(bam_header, library, sample_id, version)
    variants = tiddit_variant.main(args.bam, sv_clusters, args, library, min_mapq, samples, coverage_data, contig_number, max_ins_len, gc_dictionary)
    with open(prefix + ".vcf", "w") as f:
        f.write(vcf_header + "\n")
        for chrom in contigs:
            if chrom not in variants:
                continue
            for variant in sorted(variants[chrom], key=lambda x: x[0]):
                f.write("\t".join(variant[1]) + "\n")
    return True


def run_sv(args, version):
    from . import tiddit_cluster, tiddit_coverage_analysis, tiddit_gc, tiddit_signal, tiddit_stats
    from .bamio import BamReader
    from .fasta import FastaFile
    if args.l < 2:
        print("error, too low --l value!")
        quit()
    if not args.skip_assembly:
        print("error, local assembly is outside this build's scope; rerun with --skip_assembly")
        quit()
    if not os.path.isfile(args.ref):
        print("error,  could not find the reference file")
        quit()
    FastaFile(args.ref)          # builds ref.fai when it is missing (pysam.faidx in the reference)
    if not (args.bam.endswith(".bam") or args.bam.endswith(".cram")):
        print("error, the input file is not a bam file, make sure that the file extension is .bam or .cram")
        quit()
    if args.bam.endswith(".cram"):
        print("error, CRAM input is not supported by this build (BAM only)")
        quit()
    if not os.path.isfile(args.bam):
        print("error,  could not find the bam file")
        quit()
    reader = BamReader(args.bam, batch_bytes=1 << 20)    # (the header only: small pieces — the default piece inflates 3-4 ms of blocks nobody reads)
    bam_header = reader.header
    reader.close()
    chromosomes = [c["SN"] for c in bam_header["SQ"]]
    try:
        sample_id = bam_header["RG"][0]["SM"]
    except Exception:
        sample_id = args.bam.split("/")[-1].split(".")[0]
    samples = [sample_id]
    contigs = list(chromosomes)
    contig_number = {c: i for i, c in enumerate(contigs)}
    contig_length = {c["SN"]: c["LN"] for c in bam_header["SQ"]}
    prefix = args.o
    # one process per GPU on ONE file (BASELINE configs[4]; `torchrun --nproc-per-node N -m tiddit_amd --sv ...`): rank 0 owns the
    # output files and the host-only stages, the signal scan and the clustering are shared (tiddit_signal.main_sharded,
    # tiddit_cluster.main_sharded)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    dist = None
    # TIDDIT_FORCE_DIST=1: take the N-rank code path with whatever WORLD_SIZE says, also 1 — a one-GPU box can then run every
    # collective of the job over real RCCL (backend nccl refuses two ranks on one device; tests/test_gpu_sv_e2e.py)
    multi = world > 1 or os.environ.get("TIDDIT_FORCE_DIST") == "1"
    if multi:
        import torch
        import torch.distributed as dist
        from . import dist as tdist
        backend = os.environ.get("TIDDIT_DIST_BACKEND", "nccl")          # gloo: ranks sharing one GPU (tests)
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        own_group = not dist.is_initialized()
        if own_group:
            dist.init_process_group(backend)
        rank = dist.get_rank()
    if rank == 0:
        try:
            os.mkdir("{}_tiddit".format(prefix))
            os.mkdir("{}_tiddit/clips".format(prefix))
        except Exception:
            if not args.force_overwrite:
                print("Eror output folder exists")
                if multi:
                    dist.destroy_process_group()
                    os._exit(1)                                          # (the other ranks sit in the broadcast below: take the job down)
                quit()
    min_mapq = args.q
    max_ins_len = 100000
    T = STAGE_SECONDS
    T.clear()
    STAGE_NOTES.clear()
    from .trace import stage
    # The GC / N-mask bins depend on the reference FASTA alone (the reference computes them after the signals, __main__.py:166): a
    # second host thread with its own library context (own streams and scratch) reads the FASTA and runs the GC kernel BESIDE the BAM
    # scan — the scan waits inside the library (inflate-kernel bound; its row path holds no Python since round 4), so the thread's
    # short Python moments cost it nothing, and 3 GB more over a link that carries 54 GB in 2.2 s are noise.  (While the rows were
    # Python objects the thread started behind the scan, tiddit_signal.AFTER_SCAN, because the two fought over the GIL: that left
    # 0.42 s of a 3-Gb job waiting for it.)  On N ranks every rank computes the bins of ITS contigs (dist.shard_contigs) and rank 0
    # receives them.  TIDDIT_GC_OVERLAP=0: in sequence on rank 0's main thread; =after: the thread starts when the scan is over.
    gc_job = None
    gc_mode = os.environ.get("TIDDIT_GC_OVERLAP", "1")
    if gc_mode != "0":
        import threading
        from . import _native
        gc_job = {}
        gc_mine = list(chromosomes)
        if multi:
            owned = tdist.shard_contigs([contig_length[c] for c in chromosomes], world)[rank]
            gc_mine = [chromosomes[i] for i in owned]

        def gc_thread():
            try:
                ctx = _native.Context(_native.default_context().device)
                fasta = FastaFile(args.ref)
                gc_job["result"] = tiddit_gc.gc_of_contigs(fasta, gc_mine, 50, 0.5, ctx=ctx)
            except BaseException as e:           # re-raised on the main thread
                gc_job["error"] = e
            gc_job["seconds"] = time.time() - gc_job["t0"]

        def start_gc():
            if "thread" not in gc_job:
                gc_job["t0"] = time.time()
                gc_job["thread"] = threading.Thread(target=gc_thread, name="tiddit-gc")
                gc_job["thread"].start()
        if gc_mode == "after":
            tiddit_signal.AFTER_SCAN.append(start_gc)
        elif not multi and gc_mode != "scan":
            start_gc()          # one process: now, beside the statistics pass too (started behind it, a 240-Mb job — 0.06 s of scan for 0.055 s of GC
                                # thread — waited 3-24 ms for the thread at its end; TIDDIT_GC_OVERLAP=scan starts it behind the statistics as before)
    t = time.time()
    with stage("tiddit: library statistics"):
        if not multi:
            try:
                library = tiddit_stats.statistics(args.bam, args.ref, min_mapq, max_ins_len, args.s, carry=True)
            except BaseException:
                if gc_job is not None and "thread" in gc_job:
                    gc_job["thread"].join()                  # (no helper thread outlives the error)
                raise
        else:
            # The sample is the head of the file = the head of rank 0's byte range: rank 0 samples it through its own share's reader and
            # keeps the batches for its scan.  Nothing in inflate / record decode / the coverage records depends on the statistics, so the
            # other ranks do not wait idle for the broadcast: they ingest the head of THEIR shares meanwhile (bamio.preingest) and their
            # scans start from those batches.  TIDDIT_DIST_PREINGEST=0: wait idle (what the first N-rank build did).
            import pickle
            import numpy
            from . import bamio
            wire = torch.zeros(8192, dtype=torch.uint8, device=tdist._wire_device())
            overlap = os.environ.get("TIDDIT_DIST_PREINGEST", "1") != "0" and os.environ.get("TIDDIT_HOST_INGEST") != "1"
            if rank == 0:
                library = tiddit_stats.statistics(args.bam, args.ref, min_mapq, max_ins_len, args.s, carry=overlap, shard=(0, world) if overlap else None)
                blob = pickle.dumps(library, protocol=4)
                host = numpy.zeros(8192, dtype=numpy.uint8)
                big = len(blob) > host.size - 4               # (never seen: a dozen numbers) -> marker here, the object by itself below
                host[:4] = numpy.array([0xffffffff if big else len(blob)], dtype="<u4").view(numpy.uint8)
                if not big:
                    host[4:4 + len(blob)] = numpy.frombuffer(blob, dtype=numpy.uint8)
                wire.copy_(torch.from_numpy(host))
                dist.broadcast(wire, 0)
                if big:
                    tdist.broadcast_object(library, 0)
            else:
                work = dist.broadcast(wire, 0, async_op=True)
                held = 0
                if overlap:
                    held = bamio.preingest(args.bam, (rank, world), 50, stop=work.is_completed,
                                           chunk=int(os.environ.get("TIDDIT_INGEST_CHUNK", str(448 << 20))))
                work.wait()
                host = wire.cpu().numpy()
                size = int(host[:4].view("<u4")[0])
                library = tdist.broadcast_object(None, 0) if size == 0xffffffff else pickle.loads(host[4:4 + size].tobytes())
                STAGE_NOTES["batches ingested beside rank 0's statistics"] = held
    max_ins_len = args.i if args.i else library["percentile_insert_size"]
    T["library statistics"] = time.time() - t
    if gc_job is not None and gc_mode != "after":
        start_gc()              # (the N-rank job and TIDDIT_GC_OVERLAP=scan: beside the scan only)

    t = time.time()
    # one process: the blocks of discordants / splits / clips are placed by a thread while the job goes on (tiddit_cluster takes the tables
    # over, not the files); finish_writes() below waits for it.  TIDDIT_BACKGROUND_WRITES=0: written before tiddit_signal.main returns.
    tiddit_signal.BACKGROUND_WRITES = (not multi) and os.environ.get("TIDDIT_BACKGROUND_WRITES", "1") != "0"
    with stage("tiddit: signal extraction + coverage"):
        signal_main = tiddit_signal.main_sharded if multi else tiddit_signal.main
        try:
            coverage_data = signal_main(args.bam, args.ref, prefix, min_mapq, max_ins_len, sample_id, args.threads, args.min_contig,
                                        False, args.min_anchor_len, args.min_clip_len)
        finally:
            if gc_job is not None and gc_mode == "after":
                tiddit_signal.AFTER_SCAN.remove(start_gc)
            tiddit_signal.BACKGROUND_WRITES = False
            if gc_job is not None and sys.exc_info()[0] is not None and "thread" in gc_job:
                gc_job["thread"].join()          # (the scan failed: no helper thread outlives the error)
    if rank == 0:
        print("extracted signals in:")
        print(t - time.time())
    T["signal extraction + coverage"] = time.time() - t
    T.update({"  " + k: v for k, v in tiddit_signal.STAGE_SECONDS.items()})
    try:
        _after_scan(args, prefix, rank, multi, T, gc_job, start_gc if gc_job is not None else None, chromosomes, contigs, contig_length, samples,
                    library, coverage_data, bam_header, max_ins_len, min_mapq, sample_id, version, contig_number, own_group if multi else False)
    except BaseException:
        # no helper thread outlives the error: the writer thread of BACKGROUND_WRITES is joined (its own error, if any, is not the one to report)
        try:
            tiddit_signal.finish_writes()
        except BaseException:
            pass
        raise


def _after_scan(args, prefix, rank, multi, T, gc_job, start_gc, chromosomes, contigs, contig_length, samples, library, coverage_data, bam_header,
                max_ins_len, min_mapq, sample_id, version, contig_number, own_group):
    """run_sv behind the BAM scan: GC bins, ploidy table, clustering, candidates table, the signal files complete"""
    from . import tiddit_cluster, tiddit_coverage_analysis, tiddit_gc, tiddit_signal
    from .trace import stage
    if multi:
        import torch.distributed as dist
        from . import dist as tdist
    t = time.time()
    gc_dictionary = None
    with stage("tiddit: GC bins"):
        if gc_job is None:
            if rank == 0:
                gc_dictionary = tiddit_gc.main(args.ref, chromosomes, args.threads, 50, 0.5)
        else:
            start_gc()                          # (already running unless the thread waits for the end of the scan, or the scan never got there)
            gc_job["thread"].join()
            if "error" in gc_job:
                raise gc_job["error"]
            gc_dictionary = gc_job["result"]
            if multi:                           # every rank's contigs to rank 0 (int8 bins: 60 MB for a human genome)
                import pickle
                parts = tdist.gather_bytes(pickle.dumps(gc_dictionary, protocol=4), 0)
                if rank == 0:
                    merged = {}
                    for p_ in parts:
                        merged.update(pickle.loads(p_))
                    gc_dictionary = {c: merged[c] for c in chromosomes}
    T["GC bins"] = time.time() - t
    if gc_job is not None:
        T["  GC bins, on their own thread beside the scan"] = gc_job["seconds"]
    if rank == 0:
        t = time.time()
        with stage("tiddit: ploidy"):
            library = tiddit_coverage_analysis.determine_ploidy(coverage_data, contigs, library, args.n, prefix, args.c, args.ref, 50,
                                                                bam_header, gc_dictionary)
        print("calculated coverage in:")
        print(time.time() - t)
        T["ploidy (masked medians)"] = time.time() - t
    if not args.e:
        args.e = int(library["avg_insert_size"] / 2.0)
    if not args.e:
        args.e = 50
    t = time.time()
    with stage("tiddit: clustering"):
        cluster_main = tiddit_cluster.main_sharded if multi else tiddit_cluster.main
        sv_clusters = cluster_main(prefix, contigs, contig_length, samples, library["mp"], args.e, args.l, max_ins_len, args.min_contig,
                                   args.skip_assembly, args.r)
    T["clustering"] = time.time() - t
    T.update({"  " + k: v for k, v in tiddit_cluster.STAGE_SECONDS.items()})
    if rank == 0:
        print("generated clusters in")
        print(T["clustering"])
        t = time.time()
        write_candidates(prefix + ".candidates.tab", contigs, sv_clusters)
        T["candidates table"] = time.time() - t
    tiddit_signal.finish_writes()                                        # the signal files are complete from here on
    if tiddit_signal.WRITE_SECONDS:
        T["signal files placed (writer thread, beside ploidy and clustering)"] = tiddit_signal.WRITE_SECONDS["writer thread"]
        T["  waited for the writer thread"] = tiddit_signal.WRITE_SECONDS["waited for it"]
    if rank == 0:
        # Variant typing / filtering / the VCF (tiddit_variant.pyx, tiddit_vcf_header.py) are outside this build's scope.  When the
        # reference package itself is importable (it needs pysam) the candidates are handed to it, as the reference's driver does
        # (__main__.py:193-207), so that a full installation still ends with {prefix}.vcf.
        try:
            import tiddit.tiddit_variant as tiddit_variant
            import tiddit.tiddit_vcf_header as tiddit_vcf_header
        except Exception:
            tiddit_variant = tiddit_vcf_header = None
        t = time.time()
        if variant_stage(tiddit_variant, tiddit_vcf_header, prefix, contigs, bam_header, library, sample_id, version, args, sv_clusters, min_mapq,
                         samples, coverage_data, contig_number, max_ins_len, gc_dictionary):
            T["variant typing (reference package)"] = time.time() - t
        else:
            print("variant typing/filtering (tiddit_variant) is outside this build's scope; candidates written to {}.candidates.tab".format(prefix))
    if multi:
        dist.barrier()                                                   # every output file exists when any rank returns
        if own_group:
            # (a process that exits with its RCCL group alive can die in the group's watchdog thread while the HIP runtime unloads)
            dist.destroy_process_group()


def main(argv=None):
    version = "3.9.5"
    if argv is not None:
        sys.argv = [sys.argv[0]] + list(argv)
    parser = argparse.ArgumentParser("""tiddit-{}""".format(version), add_help=False)
    parser.add_argument("--sv", help="call structural variation", required=False, action="store_true")
    parser.add_argument("--cov", help="generate a coverage bed file", required=False, action="store_true")
    args, unknown = parser.parse_known_args()
    if args.sv:
        from .hostutil import quiet_gc
        # The job's helper threads (row building, GC bins) spend their time inside the library without the GIL and need it only for
        # moments in between; with CPython's default 5 ms switch interval each of those moments waits up to 5 ms while the main thread
        # runs a Python loop (the GC thread beside merge + write: 1.35 s for 0.46 s of work on a 3-Gb genome).
        interval = sys.getswitchinterval()
        sys.setswitchinterval(min(interval, 0.0005))
        try:
            with quiet_gc(freeze=True):      # one job, one process: the collector stays off from the first stage to the last
                run_sv(_sv_parser().parse_args(), version)
        finally:
            sys.setswitchinterval(interval)
    elif args.cov:
        run_cov(_cov_parser().parse_args())
    else:
        parser.print_help()


if __name__ == "__main__":
    main()
