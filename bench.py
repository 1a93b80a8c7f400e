#!/usr/bin/env python3
"""bench.py — throughput of the TIDDIT hot path on MI355X (one process per GPU).

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in
HBM when the timed region starts:
  * coverage (the JSON line's `value`): BASELINE configs[1] — 3 Gb genome (24 contigs x 125 Mb),
    30x 150-bp coordinate-sorted alignment stream (600 M reads), 500-bp bins, --cov read filter
    (q 20): reset accumulators -> ONE cov_accumulate launch over all 24 contigs -> ONE int64->float64 pass;
  * clustering (reported under "dbscan"): BASELINE configs[2] — gen_points(5_000_000), one chr pair,
    e=500 l=3, x pass + y pass (two launches); and under "dbscan_shared" BASELINE configs[4]'s shape: ONE list of
    300 (chrA,chrB) buckets (60x-shaped, 10 M signals), bin-packed over the ranks by signal count, clustered
    where they live, labels all-gathered over RCCL so that every rank ends with the whole cluster set.
With N > 1 ranks the job is ONE shared problem ("scaling": "strong"): the genome's contigs (coverage, GC) and the
buckets are split over the ranks, one BAM is read as byte-range shards with one exact all-reduce of the bins
(tiddit_amd.dist.coverage_sharded); `value` = units of the WHOLE problem / slowest rank's time.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line: the COMPACT record of the run (<= LINE_BUDGET bytes: the contract's keys, the whole `roofline` object with every
section's fractions, `cpu_baseline`, the sections' headline figures); the detailed record goes to gpurun_out/bench_detail_n<N>.json
($TIDDIT_BENCH_DETAIL) and to stdout only with --full-line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TRAFFIC_NOTE = "HBM bytes per launch from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the committed build, gfx950 correction applied); not observed by this run"


_EXPORTED = None


def exported_kernels():
    """demangled names (template arguments kept, parameter list dropped) of the __global__ functions in the BUILT library"""
    global _EXPORTED
    if _EXPORTED is None:
        import re
        import subprocess
        _EXPORTED = set()
        try:
            blob = open(os.path.join(REPO, "tiddit_amd", "libtiddit_hip.so"), "rb").read()
            syms = sorted(set(m.decode() for m in re.findall(rb"_Z[0-9]+[a-z][A-Za-z0-9_]+", blob)))
            import shutil
            filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
            out = subprocess.run([filt], input="\n".join(syms), capture_output=True, text=True, timeout=60).stdout.splitlines()
            for line in out:
                line = line.strip()
                if line.startswith("void "):
                    line = line[5:]
                depth, cut = 0, len(line)
                for i, ch in enumerate(line):                    # drop the parameter list: the first '(' outside <...>
                    if ch == "<":
                        depth += 1
                    elif ch == ">":
                        depth -= 1
                    elif ch == "(" and depth == 0:
                        cut = i
                        break
                _EXPORTED.add(line[:cut])
        except Exception:
            pass
    return _EXPORTED


def profiled_traffic(kernel):
    """bytes per launch of `kernel` (demangled name, template arguments included) from the committed rocprofv3 PMC summary — or
    None when the BUILT library no longer exports a kernel of that name (a profile of another build says nothing about this one)"""
    tp = os.path.join(REPO, "profiles", "traffic.json")
    if kernel not in exported_kernels():
        return None
    try:
        import hashlib
        prof = json.load(open(tp))
        src = {"cov_": ("tdt_coverage.hip", "tdt_cov_record.h"), "dbt_": ("tdt_dbscan_tile.h",), "gc_": ("tdt_gc.hip",)}
        for pre, files in src.items():                       # ... or when the kernel's source changed since the profile was taken
            if kernel.startswith(pre):
                for f in files:
                    want = prof.get("sources_sha256", {}).get(f)
                    if want is None or hashlib.sha256(open(os.path.join(REPO, "tiddit_amd", "csrc", f), "rb").read()).hexdigest() != want:
                        return None
        v = prof["kernels"].get(kernel)
        return None if v is None else float(v["bytes_per_launch"])
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--contig-len", type=int, default=125_000_000)
    ap.add_argument("--depth", type=int, default=30)
    ap.add_argument("--bin", type=int, default=500)
    ap.add_argument("--min-q", type=int, default=20)
    ap.add_argument("--dbscan-n", type=int, default=5_000_000)
    ap.add_argument("--cpu-contigs", type=int, default=8, help="contigs of the stream the CPU baseline (oracle) is timed on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cov-sv", action="store_true", help="skip the 50-bp / q>=5 pass over the same stream")
    ap.add_argument("--no-dbscan", action="store_true")
    ap.add_argument("--no-gc", action="store_true")
    ap.add_argument("--gc-len", type=int, default=3_000_000_000, help="reference bases for the GC histogram pass")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--no-next", action="store_true", help="skip the small next-row kernels (masked medians, regional evidence counts)")
    ap.add_argument("--no-sv-e2e", action="store_true", help="skip BASELINE configs[3]: tiddit --sv --skip_assembly on a WGS-shaped synthetic BAM")
    ap.add_argument("--sv-mb", type=int, default=240, help="genome size (Mb, 24 chromosomes with GRCh38's relative lengths, 30x 150-bp pairs) of that BAM")
    ap.add_argument("--sv-cpu-full-mb", type=int, default=300, help="up to this genome size the CPU legs of sv_e2e run on the whole file, above it on a bounded sample of contigs")
    ap.add_argument("--full-line", action="store_true", help="print the detailed record (tens of KB) instead of the compact line")
    ap.add_argument("--ingest-mb", type=int, default=8, help="Mb per contig (2 contigs, 30x, 100-bp reads) of the BAM the ingest pass reads")
    return ap.parse_args()


def main():
    t_run0 = time.perf_counter()
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    import torch
    import torch.distributed as dist
    # TIDDIT_BENCH_SHARE_GPU=1 (tests only): the N ranks share GPU 0 and exchange over gloo — the N-rank code paths of every section on a
    # one-GPU box.  Production: one rank per GPU, backend "nccl" (= RCCL on ROCm).
    share = os.environ.get("TIDDIT_BENCH_SHARE_GPU") == "1"
    gpu = 0 if share else local_rank
    if share:
        os.environ["TIDDIT_HIP_DEVICE"] = "0"
        os.environ["TIDDIT_DIST_BACKEND"] = "gloo"
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("TIDDIT_BENCH_FORCE_DIST") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # backend "nccl" is RCCL on ROCm
    wire = torch.device("cpu") if share else dev

    def rank_max(x):
        """slowest rank's time"""
        if not use_dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=wire)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    from tiddit_amd import _native, dist as tdist, synth, tiddit_coverage  # noqa: F401
    ctx = _native.default_context(gpu)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)

    def barrier():
        if use_dist:
            dist.barrier()

    # ---------------------------------------------------------------- coverage: data resident in HBM
    C_all, L, z = args.contigs, args.contig_len, args.bin
    mine = tdist.shard_contigs([L] * C_all, world)[rank]        # the genome's contigs of this rank (all of them when N = 1)
    C = len(mine)
    reads = []
    with torch.cuda.stream(stream):
        for c in mine:
            reads.append(synth.gen_reads_device(L, args.depth, dev, seed=synth.SEED + c))
    torch.cuda.synchronize()
    n_reads = [int(r[0].numel()) for r in reads]
    hist = tiddit_coverage.CoverageHistogram([("s%02d" % (c + 1), L) for c in mine] or [("none", 1)], z, ctx=ctx)
    nbins = [hist.nbins(c)[0] for c in range(C)]
    out_all = torch.empty(hist.total_bins(), dtype=torch.float64, device=dev)
    outs = [out_all[hist.offset(c):hist.offset(c) + nbins[c]] for c in range(C)]
    total_reads, total_bins = sum(n_reads), sum(nbins)                            # this rank's share
    job_reads, job_bins = C_all * int(L * args.depth / 150), C_all * -(-L // z)   # the whole genome
    items4 = [(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n_reads[c])
              for c in range(C)]
    # the stream as the ingest kernel leaves it for the coverage path: 8-byte packed records (start | span:24 mapq:6 unmapped dup),
    # packed here once, outside every timed region, by the library's own packing kernel
    packed = [torch.empty(n_reads[c], dtype=torch.int64, device=dev) for c in range(C)]
    pk_a, pk_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):                                   # (second pass timed: what a caller holding four arrays would pay once)
        if rep:
            pk_a.record(stream)
        for c in range(C):
            _native.check(ctx.lib.tdt_cov_pack_device(ctx.handle, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(),
                                                      reads[c][3].data_ptr(), n_reads[c], packed[c].data_ptr()))
        if rep:
            pk_b.record(stream)
        ctx.sync()
    torch.cuda.synchronize()
    pack_ms = pk_a.elapsed_time(pk_b)
    items_packed = [(c, packed[c].data_ptr(), reads[c][1].data_ptr(), n_reads[c]) for c in range(C)]
    # ... and as the ingest kernel leaves it when the reader is bound to THIS histogram (DeviceBamReader.bin_for): 8-byte binned records
    # (first bin << 2 | shape, filter byte, table indices — csrc/tdt_common.h: cov_bin_record), made here by the library's packing kernel
    binned = [torch.empty(n_reads[c], dtype=torch.int64, device=dev) for c in range(C)]
    torch.cuda.synchronize()
    for rep in range(2):
        if rep:
            pk_a.record(stream)
        for c in range(C):
            hist.pack_binned_device(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n_reads[c],
                                    binned[c].data_ptr())
        if rep:
            pk_b.record(stream)
        ctx.sync()
    torch.cuda.synchronize()
    bin_pack_ms = pk_a.elapsed_time(pk_b)
    items = [(c, binned[c].data_ptr(), reads[c][0].data_ptr(), reads[c][1].data_ptr(), n_reads[c]) for c in range(C)]

    ev_pairs = []

    def cov_step(timed):
        hist.reset()
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
        hist.push_binned_device_multi(items, args.min_q)      # all contigs of the genome in ONE cov_accumulate launch
        if timed:
            b.record(stream)
            ev_pairs.append((a, b))
        hist.finish_all_device(out_all.data_ptr())      # one int64 -> float64 pass over all bins

    for _ in range(args.warmup):
        cov_step(False)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cov_step(True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t_cov = time.perf_counter() - t0
    t_cov = rank_max(t_cov)
    ms_per_step = 1e3 * t_cov / args.steps
    kern_all = sorted(a.elapsed_time(b) for a, b in ev_pairs)
    kern_ms = sum(kern_all) / len(kern_all)                                      # avg cov_accumulate launch (whole genome)
    # bytes the launch's OWN input layout holds: 8-byte packed records + 8 B/bin.  (SURVEY §8(d)'s 12 B/read describes the
    # four-array contract — 11 B/read of arrays + 1 pad — and prices only `four_array_layout` below; pricing the packed launch with
    # it would count bytes that are never moved.)
    alg_bytes_launch = 8.0 * total_reads + 8.0 * total_bins
    alg_bytes_4 = 11.0 * total_reads + 8.0 * total_bins
    survey_bytes = 12.0 * total_reads + 8.0 * total_bins
    achieved = alg_bytes_launch / (kern_ms * 1e-3) / 1e9
    # the same launch fed from the four separate arrays (start, end, mapq, flag: 11 B/read), for the record
    ev4 = []
    for k4 in range(args.warmup + min(args.steps, 10)):
        hist.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        hist.push_device_multi(items4, args.min_q)
        b.record(stream)
        if k4 >= args.warmup:
            ev4.append((a, b))
    torch.cuda.synchronize()
    ms4 = sorted(a.elapsed_time(b) for a, b in ev4)
    evp = []                                                 # ... and from the bin-size-agnostic packed records (cov_pack_record)
    for kp in range(args.warmup + min(args.steps, 10)):
        hist.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        hist.push_packed_device_multi(items_packed, args.min_q)
        b.record(stream)
        if kp >= args.warmup:
            evp.append((a, b))
    torch.cuda.synchronize()
    msp = sorted(a.elapsed_time(b) for a, b in evp)
    hist.reset()                                             # (the headline layout's bins are the ones verified below)
    hist.push_binned_device_multi(items, args.min_q)
    hist.finish_all_device(out_all.data_ptr())
    torch.cuda.synchronize()
    traffic = profiled_traffic("cov_accumulate<true, 0, false, 8, 2>")
    if traffic is not None and world == 1:
        traffic *= 1.0     # (the profile was taken on this very workload: 600 M reads per launch)
    elif traffic is not None:
        traffic *= total_reads / float(job_reads)

    # the yardstick beside the data sheet's peak: a plain streaming read of as many bytes as the headline launch's algorithmic traffic, from
    # one buffer, one launch (tdt_calib_stream_read: four 16-byte loads in flight per lane, nothing written)
    stream_read = None
    try:
        cal = torch.empty(int(alg_bytes_launch) // 8, dtype=torch.int64, device=dev)
        cal.random_(0, 1 << 40)
        torch.cuda.synchronize()
        sweep = []
        for wpc, blocked in ((2, 0), (3, 0), (8, 0), (8, 1), (64, 1)):
            cb, cm = ctypes.c_double(0), ctypes.c_double(0)
            _native.check(ctx.lib.tdt_calib_stream_read(ctx.handle, cal.data_ptr(), cal.numel() * 8, 10, wpc, blocked, ctypes.byref(cb), ctypes.byref(cm)))
            sweep.append({"workgroups_per_cu": wpc, "walk": "blocked" if blocked else "grid-stride", "best_ms": cb.value, "mean_ms": cm.value,
                          "GB_per_s": cal.numel() * 8 / (cm.value * 1e-3) / 1e9})
        top = max(sweep, key=lambda r_: r_["GB_per_s"])
        stream_read = {"bytes": cal.numel() * 8, "GB_per_s": top["GB_per_s"], "frac_of_peak": top["GB_per_s"] / HBM_PEAK_GBS, "best_config": top, "sweep": sweep,
                       "note": "tdt_calib_stream_read: a read-only kernel (four 16-byte loads in flight per lane, nothing written) over one buffer of the "
                               "headline launch's algorithmic bytes, mean of 10 launches per configuration, the best of five: what streaming this device's "
                               "HBM reaches in practice; 'frac_of_stream_read' = achieved / this"}
        del cal
    except Exception as e:                                    # (a measurement aid: its absence never fails the bench)
        stream_read = {"error": str(e)}

    result = {
        "metric": "cov bins/sec, 30x WGS synthetic (signals clustered/sec: see 'dbscan')",
        "value": job_bins / (t_cov / args.steps),
        "unit": "bins/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: coverage histogram, %d contigs x %d bp (%.2f Gb), %dx 150-bp sorted "
                               "stream, %d-bp bins, q>=%d filter; contigs split over the %d rank(s)" % (C_all, L, C_all * L / 1e9, args.depth, z, args.min_q, world),
                   "reads": job_reads, "bins": job_bins, "reads_rank0": total_reads, "bins_rank0": total_bins, "launches_per_step": 3,
                   "layout_short": "8-byte BINNED records in HBM as the ingest kernel writes them (DESIGN 3.1): `value` prices the ACCUMULATE stage of "
                                   "ingest -> accumulate; roofline.contract_* is SURVEY 8(d)'s four arrays in one launch",
                   "layout": "8-byte BINNED records in HBM (first_bin << 2 | shape, filter byte, the two table indices of tiddit_coverage.pyx:53-63: "
                             "csrc/tdt_cov_record.h cov_bin_record) — what the ingest kernel writes when the reader is bound to this histogram "
                             "(DeviceBamReader.bin_for): the division and the bin split are done once where the record is made.  "
                             "'packed_layout' times the same launch from the bin-size-agnostic packed records (start | span:24 mapq:6 flags), "
                             "'four_array_layout' from separate start/end/mapq/flag arrays (11 B/read), 'pack_*_ms' the one-off conversions from "
                             "four arrays (outside the timed region)",
                   "pack_binned_from_four_arrays_ms": bin_pack_ms, "pack_from_four_arrays_ms": pack_ms,
                   "arithmetic": "int64 accumulation of the reference's float32 quotients at 2^-S fixed point (exact), float64 bins out"},
        "reads_per_sec": job_reads / (t_cov / args.steps),
        "roofline": {"bound": "hbm", "kernel": "cov_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": TRAFFIC_NOTE,
                     "frac_traffic": None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": kern_ms,
                     "median_launch_ms": kern_all[len(kern_all) // 2], "min_launch_ms": kern_all[0],
                     "algorithmic_bytes_per_launch": alg_bytes_launch,
                     "stream_read": stream_read,
                     "frac_of_stream_read": (achieved / stream_read["GB_per_s"]) if stream_read and "GB_per_s" in stream_read else None,
                     "bytes_model": "8 B/read (binned record) + 8 B/bin: what this launch's input layout holds",
                     "pack_binned_from_four_arrays_ms": bin_pack_ms,
                     "from_four_arrays_incl_pack": {"ms": bin_pack_ms + kern_ms, "frac": alg_bytes_4 / ((bin_pack_ms + kern_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                    "note": "a caller holding SURVEY §8(d)'s four arrays pays the one-off packing pass first; the device ingest writes the records itself"}},
        "packed_layout": {"avg_launch_ms": sum(msp) / len(msp), "median_launch_ms": msp[len(msp) // 2], "min_launch_ms": msp[0],
                          "algorithmic_bytes_per_launch": alg_bytes_launch, "bytes_model": "8 B/read (packed record) + 8 B/bin",
                          "frac": alg_bytes_launch / (sum(msp) / len(msp) * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "four_array_layout": {"avg_launch_ms": sum(ms4) / len(ms4), "median_launch_ms": ms4[len(ms4) // 2], "min_launch_ms": ms4[0],
                              "algorithmic_bytes_per_launch": alg_bytes_4, "bytes_model": "11 B/read (start i32, end i32, mapq u8, flag u16) + 8 B/bin",
                              "frac": alg_bytes_4 / (sum(ms4) / len(ms4) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "frac_survey_8d_12B_per_read": survey_bytes / (sum(ms4) / len(ms4) * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }

    # ---------------------------------------------------------------- CPU baseline + in-bench parity (rank 0, N=1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        k = min(args.cpu_contigs, C)
        t_cpu = 0.0
        ok = True
        kept = 0
        for c in range(k):
            s, e, mq, fl = [t.cpu().numpy() for t in reads[c]]
            fl = fl.view(np.uint16)
            t1 = time.perf_counter()
            want, kk = oracle.coverage_stream(s, e, mq, fl, L, z, args.min_q)
            t_cpu += time.perf_counter() - t1
            kept += kk
            ok = ok and np.array_equal(outs[c].cpu().numpy(), want)
        if not ok:
            raise SystemExit("PARITY FAILURE: GPU bins differ from the CPU oracle")
        result["cpu_baseline"] = {"value": sum(nbins[:k]) / t_cpu, "unit": "bins/s", "cores": 1, "kind": "port",
                                  "reads_per_sec": sum(n_reads[:k]) / t_cpu,
                                  "sample": "%d of %d contigs (%d reads, %d bins), oracle/tiddit_oracle.c scalar C port of the "
                                            "update_coverage loop, arrays in memory; bins verified bit-identical to the GPU's"
                                            % (k, C, sum(n_reads[:k]), sum(nbins[:k]))}
        # the same port on ALL host cores (one contig per thread: contigs are independent; ctypes releases the GIL), every contig of
        # the genome — which also verifies the remaining contigs against the GPU's bins
        from concurrent.futures import ThreadPoolExecutor
        host = [[t.cpu().numpy() for t in reads[c]] for c in range(C)]
        ncores = os.cpu_count() or 1
        nthreads = max(1, min(ncores, C))

        def one(c):
            s, e, mq, fl = host[c]
            return oracle.coverage_stream(s, e, mq, fl.view(np.uint16), L, z, args.min_q)[0]

        t1 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as pool:
            wants = list(pool.map(one, range(C)))
        t_all = time.perf_counter() - t1
        for c in range(C):
            if not np.array_equal(outs[c].cpu().numpy(), wants[c]):
                raise SystemExit("PARITY FAILURE: GPU bins of contig %d differ from the CPU oracle" % c)
        del host, wants
        result["cpu_baseline_all_cores"] = {"value": total_bins / t_all, "unit": "bins/s", "reads_per_sec": total_reads / t_all, "cores": nthreads,
                                            "host_cores": ncores, "cpu_model": cpu_model(), "kind": "port",
                                            "sample": "all %d contigs (%d reads), one contig per thread on %d threads; every contig's bins verified "
                                                      "bit-identical to the GPU's" % (C, total_reads, nthreads)}
        result["parity_checked"] = True
    # host-buffer path (tdt_cov_push: pinned double-buffered hipMemcpyAsync + kernel), PCIe/host-memcpy bound;
    # reported for completeness, never the headline value
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        hs, he, hm, hf = [t.cpu().numpy() for t in reads[0]]
        hf = hf.view(np.uint16)
        hist.reset()
        ctx.sync()
        t1 = time.perf_counter()
        hist.push(0, hs, he, hm, hf, args.min_q)
        ctx.sync()
        t_host = time.perf_counter() - t1
        result["host_push"] = {"reads_per_sec": len(hs) / t_host, "GB_per_s": 11.0 * len(hs) / t_host / 1e9,
                               "note": "one contig (%d reads) from pageable numpy arrays through the pinned staging ring" % len(hs)}
    # ---------------------------------------------------------------- coverage, SV flavour: same stream, 50-bp bins, q >= 5
    # (tiddit_signal.pyx:181-182,235: what `tiddit --sv` accumulates; a 150-bp read covers 3-5 bins)
    if not args.no_cov_sv:
        zs, qs_ = 50, 5
        hist_sv = tiddit_coverage.CoverageHistogram([("s%02d" % (c + 1), L) for c in mine] or [("none", 1)], zs, ctx=ctx)
        nb_sv = [hist_sv.nbins(c)[0] for c in range(C)]
        out_sv = torch.empty(hist_sv.total_bins(), dtype=torch.float64, device=dev)
        sv_ev = []

        for c in range(C):                                   # the same buffers, re-made for the 50-bp histogram (the headline is done with them)
            hist_sv.pack_binned_device(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n_reads[c],
                                       binned[c].data_ptr())
        ctx.sync()

        def sv_step(timed):
            hist_sv.reset()
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            hist_sv.push_binned_device_multi(items, qs_)
            if timed:
                b.record(stream)
                sv_ev.append((a, b))
            hist_sv.finish_all_device(out_sv.data_ptr())

        for _ in range(max(args.warmup, 8)):           # the GPU sat idle while the CPU baseline ran: let the clocks come back up
            sv_step(False)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sv_step(True)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t_sv = time.perf_counter() - t0
        t_sv = rank_max(t_sv)
        sv_all = sorted(a.elapsed_time(b) for a, b in sv_ev)
        sv_ms = sum(sv_all) / len(sv_all)
        sv_bytes = 8.0 * total_reads + 8.0 * sum(nb_sv)                 # packed records + bins (see the headline's bytes_model)
        sv_ach = sv_bytes / (sv_ms * 1e-3) / 1e9
        sv_traffic = profiled_traffic("cov_accumulate<true, 1, false, 4, 2>")
        if sv_traffic is not None:
            sv_traffic *= total_reads / float(job_reads)
        svres = {"metric": "cov bins/sec, SV flavour (50-bp bins, q>=5)", "value": C_all * -(-L // zs) / (t_sv / args.steps), "unit": "bins/s",
                 "reads_per_sec": job_reads / (t_sv / args.steps), "ms_per_step": 1e3 * t_sv / args.steps,
                 "config": {"workload": "the same %d-read stream, %d-bp bins (%d bins), q>=%d filter: what `tiddit --sv` accumulates; contigs split over the rank(s)"
                                        % (job_reads, zs, C_all * -(-L // zs), qs_)},
                 "roofline": {"bound": "hbm", "kernel": "cov_accumulate (small-bin flavour)", "achieved": sv_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": sv_ach / HBM_PEAK_GBS, "traffic": sv_traffic, "traffic_source": TRAFFIC_NOTE,
                              "frac_traffic": None if sv_traffic is None else sv_traffic / (sv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "avg_launch_ms": sv_ms, "median_launch_ms": sv_all[len(sv_all) // 2],
                              "min_launch_ms": sv_all[0], "algorithmic_bytes_per_launch": sv_bytes,
                              "bytes_model": "8 B/read (binned record) + 8 B/bin"}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            import oracle
            t_cpu = 0.0
            for c in range(C):
                s, e, mq, fl = [t.cpu().numpy() for t in reads[c]]
                fl = fl.view(np.uint16)
                t1 = time.perf_counter()
                want, _ = oracle.coverage_stream(s, e, mq, fl, L, zs, qs_)
                t_cpu += time.perf_counter() - t1
                o = hist_sv.offset(c)
                if not np.array_equal(out_sv[o:o + nb_sv[c]].cpu().numpy(), want):
                    raise SystemExit("PARITY FAILURE: GPU 50-bp bins of contig %d differ from the CPU oracle" % c)
            svres["cpu_baseline"] = {"value": sum(nb_sv) / t_cpu, "unit": "bins/s", "cores": 1, "kind": "port", "reads_per_sec": total_reads / t_cpu,
                                     "sample": "all %d contigs (%d reads), oracle/tiddit_oracle.c; bins verified bit-identical to the GPU's" % (C, total_reads)}
            svres["parity_checked"] = True
        result["coverage_sv"] = svres
        del out_sv
        hist_sv.close()
    del reads, outs, out_all, packed, binned
    hist.close()
    torch.cuda.empty_cache()

    # ---------------------------------------------------------------- clustering, ONE bucket list shared by the ranks (configs[4]'s shape)
    if not args.no_dbscan:
        result["dbscan_shared"] = dbscan_shared(args, ctx, stream, dev, rank, world, use_dist, barrier, wire, rank_max)

    # ---------------------------------------------------------------- clustering (configs[2]): one chr pair, one GPU
    if not args.no_dbscan and world == 1:
        n = args.dbscan_n
        pts = synth.gen_points(n, seed=synth.SEED)
        x = torch.from_numpy(pts[:, 0].astype(np.uint32).view(np.int32)).to(dev)
        y = torch.from_numpy(pts[:, 1].astype(np.uint32).view(np.int32)).to(dev)
        lab = torch.empty(n, dtype=torch.float64, device=dev)
        lid = torch.empty(1, dtype=torch.int64, device=dev)
        off = np.array([0, n], dtype=np.int64)
        torch.cuda.synchronize()
        dev_ms = []

        def db_step(timed):
            with torch.cuda.stream(stream):
                if timed:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                _native.check(ctx.lib.tdt_dbscan_device(ctx.handle, x.data_ptr(), y.data_ptr(), n, _native.ptr(off), 1,
                                                        ctypes.c_uint64(500), 3, 0, lab.data_ptr(), lid.data_ptr()))
                if timed:
                    b.record(stream)
                    dev_ms.append((a, b))

        for _ in range(args.warmup):
            db_step(False)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            db_step(True)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t_db = time.perf_counter() - t0
        t_db = rank_max(t_db)
        k_ms = sum(a.elapsed_time(b) for a, b in dev_ms) / len(dev_ms)
        db_ach = 16.0 * n / (k_ms * 1e-3) / 1e9
        t1_, t2_ = profiled_traffic("dbt_tile<true, false, false>"), profiled_traffic("dbt_finish1")
        db_traffic = None if t1_ is None or t2_ is None else t1_ + t2_
        dbres = {"metric": "signals clustered/sec", "value": n / (t_db / args.steps), "unit": "signals/s",
                 "ms_per_step": 1e3 * t_db / args.steps,
                 "config": {"workload": "BASELINE configs[2]: gen_points(%d) one chr pair, e=500 l=3, one GPU" % n},
                 "roofline": {"bound": "hbm", "kernel": "tdt_dbscan_device: dbt_tile + dbt_finish1 (2 launches)", "achieved": db_ach, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": db_ach / HBM_PEAK_GBS, "traffic": db_traffic, "traffic_source": TRAFFIC_NOTE,
                              "frac_traffic": None if db_traffic is None else db_traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_pass_ms": k_ms,
                              "algorithmic_bytes_per_pass": 16.0 * n}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            import oracle
            t1 = time.perf_counter()
            want = oracle.dbscan_main(pts, 500, 3)
            t_sweep = time.perf_counter() - t1
            if not np.array_equal(lab.cpu().numpy(), want):
                raise SystemExit("PARITY FAILURE: GPU labels differ from the CPU oracle")
            ns = min(n, 200_000)
            t1 = time.perf_counter()
            oracle.dbscan_main(pts[:ns], 500, 3, literal=True)
            t_lit = time.perf_counter() - t1
            dbres["cpu_baseline"] = {"value": n / t_sweep, "unit": "signals/s", "cores": 1, "kind": "port",
                                     "sample": "all %d points, oracle C port with the O(N) membership sweep" % n,
                                     "literal_O(KN)": {"value": ns / t_lit, "unit": "signals/s",
                                                       "sample": "first %d points, literal `clusters == cluster` mask per x-cluster "
                                                                 "(DBSCAN.py:72) in C" % ns}}
            dbres["parity_checked"] = True
        # tiddit_cluster.main's call (tiddit_cluster.pyx:140-154): host int64 (posA, posB) of every (chrA,chrB) bucket in signal order ->
        # stable sort by posA + DBSCAN.main per bucket, labels back on the host.  Wall clock of the synchronous C call.
        if rank == 0:
            def time_sort_dbscan(posA, posB, off):
                nn = len(posA)
                perm = np.empty(nn, dtype=np.uint32)
                labs = np.empty(nn, dtype=np.float64)
                ts = []
                for _ in range(1 + max(3, min(args.steps, 10))):
                    t1 = time.perf_counter()
                    _native.check(ctx.lib.tdt_sort_dbscan(ctx.handle, _native.ptr(posA), _native.ptr(posB), nn, _native.ptr(off), len(off) - 1,
                                                          500.0, 3, _native.ptr(perm), _native.ptr(labs)))
                    ts.append(time.perf_counter() - t1)
                return sorted(ts[1:])[len(ts[1:]) // 2], perm, labs
            by_signal = pts[np.argsort(pts[:, 2], kind="stable")]                   # arrival order, as the .tab files list the signals
            pa, pb = np.ascontiguousarray(by_signal[:, 0]), np.ascontiguousarray(by_signal[:, 1])
            t_one, perm1, lab1 = time_sort_dbscan(pa, pb, np.array([0, n], dtype=np.int64))
            # ~300 buckets shaped like a human WGS run: 24 intra-chromosomal buckets hold most signals, 276 inter-chromosomal ones the rest
            rng = np.random.default_rng(99)
            w_in = np.array([248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57], dtype=np.float64)
            sizes = np.concatenate([(0.85 * n * w_in / w_in.sum()).astype(np.int64), (0.15 * n * rng.dirichlet(np.ones(276) * 0.7)).astype(np.int64)])
            order = rng.permutation(len(sizes))
            sizes = sizes[order]
            offm = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            nm = int(offm[-1])
            pam, pbm = pa[:nm].copy(), pb[:nm].copy()
            t_many, permm, labm = time_sort_dbscan(pam, pbm, offm)
            # what tiddit_cluster.main calls now: int32 columns built in pinned memory, no host pass, labels back in signal order (4 B each)
            from tiddit_amd.hostutil import PinnedPool
            pool = PinnedPool()

            def time_columns(posA, posB, off, tag):
                nn = len(posA)
                a32, b32, l32 = pool.take(tag + "a", nn, np.int32), pool.take(tag + "b", nn, np.int32), pool.take(tag + "l", nn, np.int32)
                a32[:], b32[:] = posA, posB
                bound = int(posA.max())                       # (the caller's bound, e.g. the longest contig: not part of the call)
                ts = []
                for _ in range(1 + max(3, min(args.steps, 10))):
                    t1 = time.perf_counter()
                    _native.check(ctx.lib.tdt_cluster_columns(ctx.handle, _native.ptr(a32), _native.ptr(b32), nn, _native.ptr(off), len(off) - 1,
                                                              500.0, 3, bound, _native.ptr(l32), None, None))
                    ts.append(time.perf_counter() - t1)
                return sorted(ts[1:])[len(ts[1:]) // 2], l32.copy()
            tc_one, lc1 = time_columns(pa, pb, np.array([0, n], dtype=np.int64), "one")
            tc_many, lcm = time_columns(pam, pbm, offm, "many")
            by1 = np.empty(n)
            by1[perm1] = lab1
            bym = np.empty(nm)
            bym[permm] = labm
            if not (np.array_equal(lc1.astype(np.float64), by1) and np.array_equal(lcm.astype(np.float64), bym)):
                raise SystemExit("PARITY FAILURE: tdt_cluster_columns differs from tdt_sort_dbscan")
            pool.close()
            sres = {"metric": "tdt_sort_dbscan from host int64 columns (the round-2 call of tiddit_cluster.main), signals/sec",
                    "cluster_columns": {"metric": "tdt_cluster_columns: pinned int32 columns in, int32 labels in signal order out (what tiddit_cluster.main calls)",
                                        "one_bucket": {"signals": n, "ms": 1e3 * tc_one, "value": n / tc_one},
                                        "many_buckets": {"signals": nm, "buckets": int(len(sizes)), "ms": 1e3 * tc_many, "value": nm / tc_many},
                                        "note": "H2D 8 B/signal by DMA from the caller's pinned columns (posB rides behind the sort of the posA digits), radix sort on the significant "
                                                "digits, both clustering passes, labels scattered to signal order on the device, D2H 4 B/signal; labels verified equal to tdt_sort_dbscan's"},
                    "one_bucket": {"signals": n, "ms": 1e3 * t_one, "value": n / t_one},
                    "many_buckets": {"signals": nm, "buckets": int(len(sizes)), "largest_bucket": int(sizes.max()), "ms": 1e3 * t_many, "value": nm / t_many},
                    "unit": "signals/s", "note": "includes the host passes over the two int64 columns (16 B/signal read), the H2D copy of their 32-bit offsets (8 B/signal), the device radix sort by (bucket, posA), both clustering passes and the D2H copy of order + int32 labels (8 B/signal)"}
            if world == 1 and not args.no_cpu_baseline:
                import oracle
                o1 = np.argsort(pa, kind="stable")
                w1 = oracle.dbscan_main(np.stack([pa[o1], pb[o1], o1], 1).astype(np.int64), 500, 3)
                if not (np.array_equal(perm1, o1.astype(np.uint32)) and np.array_equal(lab1, w1)):
                    raise SystemExit("PARITY FAILURE: tdt_sort_dbscan (one bucket) differs from stable argsort + the CPU oracle")
                t1 = time.perf_counter()
                for b in range(len(sizes)):
                    lo, hi = int(offm[b]), int(offm[b + 1])
                    if hi == lo:
                        continue
                    ob = np.argsort(pam[lo:hi], kind="stable")
                    wb = oracle.dbscan_main(np.stack([pam[lo:hi][ob], pbm[lo:hi][ob], ob], 1).astype(np.int64), 500, 3)
                    if not (np.array_equal(permm[lo:hi], (ob + lo).astype(np.uint32)) and np.array_equal(labm[lo:hi], wb)):
                        raise SystemExit("PARITY FAILURE: tdt_sort_dbscan bucket %d differs from stable argsort + the CPU oracle" % b)
                sres["cpu_baseline"] = {"value": nm / (time.perf_counter() - t1), "unit": "signals/s", "cores": 1, "kind": "port",
                                        "sample": "the same %d buckets: numpy stable argsort + oracle C DBSCAN per bucket" % len(sizes)}
                sres["parity_checked"] = True
            dbres["sort_dbscan"] = sres
        result["dbscan"] = dbres

    # ---------------------------------------------------------------- GC / N-mask histogram (50-bp bins, cutoff 0.5)
    if not args.no_gc:
        G_all = args.gc_len
        G = G_all // world                                   # the reference's bases are split over the ranks (whole bins each)
        G -= G % 50
        with torch.cuda.stream(stream):
            gen = torch.Generator(device=dev)
            gen.manual_seed(synth.SEED + 7 + rank)
            code = torch.randint(0, 256, (G,), generator=gen, device=dev, dtype=torch.uint8)
            # bytes: mostly ACGT, some lower case, ~3 % N (runs come from the low bits of a coarse index)
            lut = torch.tensor(list(b"ACGTACGTACGTacgtNnRYACGTGGCCAATT"), device=dev, dtype=torch.uint8)
            seq = lut[(code & 31).long()] if G <= 500_000_000 else None
            if seq is None:
                seq = torch.empty(G, dtype=torch.uint8, device=dev)
                step = 500_000_000
                for o in range(0, G, step):
                    seq[o:o + step] = lut[(code[o:o + step] & 31).long()]
            del code
            gout = torch.empty(-(-G // 50), dtype=torch.int8, device=dev)
        torch.cuda.synchronize()
        gev = []

        def gc_step(timed):
            with torch.cuda.stream(stream):
                if timed:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                _native.check(ctx.lib.tdt_gc_bins_device(ctx.handle, seq.data_ptr(), G, 50, 0.5, gout.data_ptr()))
                if timed:
                    b.record(stream)
                    gev.append((a, b))

        for _ in range(args.warmup):
            gc_step(False)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gc_step(True)
        torch.cuda.synchronize()
        barrier()
        t_gc = time.perf_counter() - t0
        t_gc = rank_max(t_gc)
        g_ms = sum(a.elapsed_time(b) for a, b in gev) / len(gev)
        g_ach = (G + G / 50.0) / (g_ms * 1e-3) / 1e9
        g_traffic = profiled_traffic("gc_small_bins")
        if g_traffic is not None:
            g_traffic *= G / float(G_all)
        gres = {"metric": "gc bins/sec", "value": (G / 50.0) * world / (t_gc / args.steps), "unit": "bins/s",
                "bases_per_sec": G * world / (t_gc / args.steps), "ms_per_step": 1e3 * t_gc / args.steps,
                "config": {"workload": "GC/N-mask histogram, %d bases split over %d rank(s), 50-bp bins, n_cutoff 0.5" % (G * world, world)},
                "roofline": {"bound": "hbm", "kernel": "gc_small_bins", "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": g_ach / HBM_PEAK_GBS, "traffic": g_traffic, "traffic_source": TRAFFIC_NOTE, "avg_launch_ms": g_ms,
                             "algorithmic_bytes_per_launch": G + G / 50.0}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            import oracle
            ns = min(G, 200_000_000)
            hs = seq[:ns].cpu().numpy()
            t1 = time.perf_counter()
            want = oracle.binned_gc(hs, 50, 0.5)
            t_cpu = time.perf_counter() - t1
            if not np.array_equal(gout[:len(want)].cpu().numpy()[:ns // 50], want[:ns // 50]):
                raise SystemExit("PARITY FAILURE: GPU GC bins differ from the CPU oracle")
            gres["cpu_baseline"] = {"value": len(want) / t_cpu, "unit": "bins/s", "cores": 1, "kind": "port",
                                    "sample": "first %d bases, oracle C port of the per-character loop" % ns}
            gres["parity_checked"] = True
        result["gc"] = gres
        del seq, gout

    # ---- BAM ingest: BGZF inflate + record decode on the device, from a file to the packed arrays (next-row path, §8(f)2)
    if not args.no_ingest:
        from tiddit_amd import bamio, synth_bam
        path = "/tmp/tiddit_bench_real6_%d.bam" % args.ingest_mb    # one file per node, written by local rank 0, read by every rank
        if local_rank == 0 and not os.path.exists(path):
            # reads cut from a common reference (overlapping reads share sequence), qualities in runs, zlib level 6: the shape of a
            # real coordinate-sorted BAM (3.7x compression) rather than independent random bytes
            synth_bam.write_bulk_bam(path + ".tmp", [("chr1", args.ingest_mb * 1_000_000), ("chr2", args.ingest_mb * 1_000_000)], depth=30,
                                     threads=min(16, os.cpu_count() or 1), level=6, realistic=True)
            os.replace(path + ".tmp", path)
        barrier()
        fsize = os.path.getsize(path)

        timing_on, batch_times = [False], []
        ingest_chunk = [448 << 20]                                # the reader's default span
        span_count = [0]
        ingest_split = [True]                                     # ... and its default for short files: four spans (here 4 x 65 MB)

        def ingest_pass():
            if use_dist:     # ONE file, byte-range shards, seam check, one exact all-reduce of the 500-bp bins (dist.coverage_sharded)
                _, _, k = tdist.coverage_sharded(path, 500, 20, ctx=ctx)
                tk = torch.tensor([k], dtype=torch.int64, device=wire)
                dist.all_reduce(tk)
                return int(tk.item())
            r = bamio.DeviceBamReader(path, ctx=ctx, chunk=ingest_chunk[0], split_small=ingest_split[0])
            r.collect_timing = timing_on[0]
            k = nb_ = 0
            for b in r.batches():
                k += len(b)
                nb_ += 1
            batch_times[:] = r.timings
            span_count[0] = nb_
            r.close()
            return k

        nrec = ingest_pass()                                       # warm-up: page cache, device buffers ...
        ingest_pass()                                              # ... and the reader's pinned ring: a process's SECOND pass still pays for it (38 ms against 22-23 from the third on, tools/time_ingest_passes.py)
        torch.cuda.synchronize()
        barrier()
        isteps = max(1, min(args.steps, 3))
        t0 = time.perf_counter()
        for _ in range(isteps):
            ingest_pass()
        ctx.sync()
        barrier()
        t_in = (time.perf_counter() - t0) / isteps
        t_in = rank_max(t_in)
        ires = {"metric": "BAM records decoded/sec (file -> packed arrays in HBM)", "value": nrec / t_in, "unit": "records/s",
                "ms_per_step": 1e3 * t_in, "bam_MB_per_sec": fsize / t_in / 1e6,
                "config": {"workload": "%d-record coordinate-sorted BAM (%.0f MB BGZF, zlib level 6, reads cut from a common reference), inflate + CRC32 + record decode on the device, read as %d spans%s"
                                       % (nrec, fsize / 1e6, span_count[0], "" if world == 1 else "; the ONE file read as %d byte-range shards, 500-bp coverage bins all-reduced (exact)" % world)}}
        if world == 1:
            # where a pass spends its time, batch by batch (one extra pass with the stage timers on: they wait for every batch's decode kernel)
            keys = ("read_ms", "wait_for_reader_ms", "block_table_ms", "h2d_ms", "inflate_crc_ms", "find_records_ms", "chain_check_ms", "decode_ms", "push_wall_ms")
            ires["per_batch"] = {"note": "read_ms: positional reads of the span by the reader thread (overlaps the device work of the previous batch); "
                                         "wait_for_reader_ms: what the consumer actually waited for it; h2d_ms: PCIe copy of the compressed span (prefetched = on the copy "
                                         "stream behind the previous batch's kernels); inflate_crc / find_records / decode: kernels (HIP events); block_table / chain_check: host"}
            for label, chunk in (("span_448MB", 448 << 20), ("span_64MB", 64 << 20)):
                ingest_chunk[0] = chunk
                ingest_split[0] = False                           # (the file as ONE 448-MB span / as 64-MB spans, whatever the reader's default)
                ingest_pass()                                     # buffers of this span size
                t1 = time.perf_counter()
                ingest_pass()
                t_plain = time.perf_counter() - t1
                timing_on[0] = True
                t1 = time.perf_counter()
                ingest_pass()
                t_timed = time.perf_counter() - t1
                timing_on[0] = False
                ires["per_batch"][label] = {"batches": len(batch_times), "pass_wall_ms": 1e3 * t_plain, "pass_wall_ms_with_timers": 1e3 * t_timed,
                                            "sum_ms": {k: round(sum(b[k] for b in batch_times), 3) for k in keys},
                                            "h2d_prefetched_batches": sum(1 for b in batch_times if b["h2d_prefetched"]),
                                            "rows": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in b.items()} for b in batch_times[:8]]}
            ingest_chunk[0] = 448 << 20
            ingest_split[0] = True
            # what the BINNED coverage records cost where the pipeline pays for them: the record-decode kernel (bam_decode_fields) with the
            # reader bound to a 500-bp histogram (it then also computes first_bin, the bin shape and the two table indices of
            # tiddit_coverage.pyx:50-63 and writes the 8-byte record) against the same kernel unbound — the headline launch reads these records
            def decode_ms(bind):
                best = None
                for _ in range(4):
                    r = bamio.DeviceBamReader(path, ctx=ctx, chunk=448 << 20, split_small=False)
                    r.collect_timing = True
                    hh = None
                    if bind:
                        hh = tiddit_coverage.CoverageHistogram([(n_, l_) for n_, l_ in zip(r.references, r.lengths)], 500, ctx=ctx)
                        r.bin_for(hh)
                    k = 0
                    for b in r.batches():
                        k += len(b)
                    ms = sum(t["decode_ms"] for t in r.timings)
                    r.close()
                    if hh is not None:
                        hh.close()
                    best = ms if best is None else min(best, ms)
                return best, k
            ms_bound, k_ = decode_ms(True)
            ms_plain, _ = decode_ms(False)
            ires["binning"] = {"decode_ms_bound_to_histogram": ms_bound, "decode_ms_unbound": ms_plain, "records": k_,
                               "binning_ms_per_600M_reads": (ms_bound - ms_plain) / k_ * 600e6,
                               "decode_ms_per_600M_reads_bound": ms_bound / k_ * 600e6,
                               "note": "bam_decode_fields (HIP events, best of 4 passes) with and without tdt_ingest_bin_for; the difference scaled to "
                                       "configs[1]'s 600 M reads is what the binned layout of the headline launch costs in the ingest kernel"}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            lib = ctx.lib
            threads = int(lib.tdt_host_threads(0))
            t1 = time.perf_counter()
            r = bamio.BamReader(path)
            hk = 0
            for b in r.batches():
                hk += len(b)
            r.close()
            t_host = time.perf_counter() - t1
            if hk != nrec:
                raise SystemExit("PARITY FAILURE: device ingest decoded %d records, host path %d" % (nrec, hk))
            ires["cpu_baseline"] = {"value": hk / t_host, "unit": "records/s", "cores": threads, "kind": "port",
                                    "sample": "same file, zlib inflate + C record decode on %d host threads (this library's host path)" % threads}
        result["ingest"] = ires

    # ---- the other next-row kernels: masked coverage medians, regional evidence counts — rank 0, small
    if not args.no_next and rank == 0:
        from tiddit_amd import tiddit_coverage_analysis as tca
        rng = np.random.default_rng(11)
        pairs = [(rng.gamma(30, 1.0, 250_000), np.where(rng.random(250_000) < 0.03, -1, 41).astype(np.int8)) for _ in range(24)]
        tca.masked_medians(pairs[:2])
        t0 = time.perf_counter()
        med, allm = tca.masked_medians(pairs)
        t_dev = time.perf_counter() - t0
        nres = {"median": {"metric": "masked medians of 24 x 250000 coverage bins + the genome-wide one (determine_ploidy), host arrays in",
                           "ms": 1e3 * t_dev, "bins_per_sec": 6_000_000 / t_dev}}
        if not args.no_cpu_baseline:
            t0 = time.perf_counter()
            ref = [float(np.median(c[(c > 0) & (g != -1)])) for c, g in pairs]
            refall = float(np.median(np.concatenate([c[(c > 0) & (g != -1)] for c, g in pairs])))
            t_np = time.perf_counter() - t0
            if med != ref or allm != refall:
                raise SystemExit("PARITY FAILURE: device medians differ from numpy.median")
            nres["median"]["cpu_baseline"] = {"value": 6_000_000 / t_np, "unit": "bins/s", "cores": 1, "kind": "port",
                                              "sample": "numpy boolean mask + numpy.median, the reference's own method"}
            nres["median"]["parity_checked"] = True
        # regional evidence counts: one contig of the coverage stream as the read table, 100k candidate regions
        L1 = args.contig_len
        r_start, r_end, r_mapq, r_flag = synth.gen_reads_device(L1, args.depth, dev, seed=synth.SEED + 77)
        n1 = int(r_start.numel())
        g2 = torch.Generator(device=dev)
        g2.manual_seed(5)
        ins = (torch.randn(n1, generator=g2, device=dev) * 50 + 350).to(torch.int32)
        mate_pos = torch.clamp(r_start + ins, 0, L1 - 1)
        tlen = (mate_pos - r_start + 150).to(torch.int32)
        mate_tid = torch.where(torch.rand(n1, generator=g2, device=dev) < 0.01, 3, 0).to(torch.int32)
        has_sa = (torch.rand(n1, generator=g2, device=dev) < 0.01).to(torch.uint8)
        NQ = 100_000
        qs = torch.randint(0, L1 - 5000, (NQ,), generator=g2, device=dev, dtype=torch.int32)
        qe = qs + torch.randint(0, 2000, (NQ,), generator=g2, device=dev, dtype=torch.int32)
        qb = torch.where(torch.rand(NQ, generator=g2, device=dev) < 0.5, qs, qe)
        rout = torch.zeros(NQ, 7, dtype=torch.int64, device=dev)
        max_span = int((r_end - r_start).max().item())
        torch.cuda.synchronize()

        def region_call():
            _native.check(ctx.lib.tdt_region_counts_device(ctx.handle, r_start.data_ptr(), r_end.data_ptr(), r_mapq.data_ptr(), r_flag.data_ptr(),
                                                           mate_tid.data_ptr(), mate_pos.data_ptr(), tlen.data_ptr(), has_sa.data_ptr(), n1, 0,
                                                           max_span, L1, qs.data_ptr(), qe.data_ptr(), qb.data_ptr(), NQ, 5, 600, rout.data_ptr()))

        with torch.cuda.stream(stream):
            region_call()
            stream.synchronize()
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(stream)
            for _ in range(10):
                region_call()
            eb.record(stream)
            stream.synchronize()
        r_ms = ea.elapsed_time(eb) / 10
        nres["region"] = {"metric": "get_region evidence counts, candidates/sec (one 30x contig of %d reads resident, %d candidate regions)" % (n1, NQ),
                          "value": NQ / (r_ms * 1e-3), "unit": "candidates/s", "ms": r_ms}
        if not args.no_cpu_baseline:
            import oracle
            tab = dict(start=r_start.cpu().numpy(), end=r_end.cpu().numpy(), mapq=r_mapq.cpu().numpy(), flag=r_flag.cpu().numpy().view(np.uint16),
                       mate_tid=mate_tid.cpu().numpy(), mate_pos=mate_pos.cpu().numpy(), tlen=tlen.cpu().numpy(), has_sa=has_sa.cpu().numpy())
            hq = (qs.cpu().numpy(), qe.cpu().numpy(), qb.cpu().numpy())
            got = rout.cpu().numpy()
            t0 = time.perf_counter()
            for q in range(0, NQ, NQ // 16):
                want = oracle.get_region_counts(tab, 0, L1, int(hq[0][q]), int(hq[1][q]), int(hq[2][q]), 5, 600)
                if not np.array_equal(got[q], want):
                    raise SystemExit("PARITY FAILURE: region counts differ from the literal loop at candidate %d" % q)
            t_lit = (time.perf_counter() - t0) / 16
            nres["region"]["cpu_baseline"] = {"value": 1.0 / t_lit, "unit": "candidates/s", "cores": 1, "kind": "port",
                                              "sample": "16 of the candidates through the literal per-read loop (full scan of the contig's reads per candidate)"}
            nres["region"]["parity_checked"] = True
        result["next_rows"] = nres

    # ---- BASELINE configs[3]: `tiddit --sv --skip_assembly` end to end, from the BAM file to the candidates table (rank 0)
    # (one process, one GPU: at N > 1 the section is left out rather than run beside idle ranks)
    # (N > 1: ONE job over the ranks — byte-range shards of the one BAM, exact all-reduce of the bins, rows sent to the owner rank of
    # their chrA, which writes, clusters and regroups them: tiddit_amd.__main__.run_sv under WORLD_SIZE > 1 = BASELINE configs[4]'s code path)
    if not args.no_sv_e2e:
        r_ = sv_e2e(args, ctx, not args.no_cpu_baseline and world == 1, rank, world, local_rank, barrier)
        if rank == 0:
            result["sv_e2e"] = r_

    if rank == 0:
        # every fraction a reader of the driver's one-line record needs, inside `roofline` (the sections' own dictionaries hold the detail)
        def compact(r, ms_key):
            rf = r.get("roofline") if r else None
            if not rf:
                return None
            tr = rf.get("traffic")
            alg = rf.get("algorithmic_bytes_per_launch", rf.get("algorithmic_bytes_per_pass"))
            sr = (result["roofline"].get("stream_read") or {}).get("GB_per_s")
            return {"ms": rf.get(ms_key), "frac": rf.get("frac"), "traffic_ratio": None if tr is None or not alg else tr / alg,
                    "frac_of_stream_read": None if not sr or rf.get("achieved") is None else rf["achieved"] / sr}
        sections = {"coverage_sv": compact(result.get("coverage_sv"), "avg_launch_ms"), "dbscan": compact(result.get("dbscan"), "avg_pass_ms"),
                    "gc": compact(result.get("gc"), "avg_launch_ms")}
        ing = result.get("ingest")
        if ing:
            sections["ingest"] = {"ms": ing.get("ms_per_step"), "records_per_sec": ing.get("value"), "bam_MB_per_sec": ing.get("bam_MB_per_sec"),
                                  "bound": "the inflate kernel's chain of dependent instructions per 64-bit window (vector ~60 %, scalar ~50 % busy, a wave issues one instruction per ~16 cycles) and the round trips of its match copies (DESIGN.md 3.6), not HBM: no fraction of the HBM roofline is claimed"}
            if "binning" in ing:
                result["roofline"]["binning_ms_per_600M_reads"] = ing["binning"]["binning_ms_per_600M_reads"]
        sv = result.get("sv_e2e")
        if sv:
            sections["sv_e2e"] = {"wall_s": sv.get("wall_s"), "serial_s": sv.get("serial_s")}
        result["roofline"]["sections"] = {k: v for k, v in sections.items() if v}
        if "four_array_layout" in result:
            f4 = result["four_array_layout"]
            result["roofline"]["contract"] = {"layout": "SURVEY 8(d): start i32, end i32, mapq u8, flag u16 - all of update_coverage in ONE launch, no packing pass",
                                              "ms": f4["avg_launch_ms"], "frac": f4["frac"], "frac_survey_8d_12B_per_read": f4["frac_survey_8d_12B_per_read"],
                                              "frac_of_stream_read": (f4["frac"] * HBM_PEAK_GBS / result["roofline"]["stream_read"]["GB_per_s"])
                                              if (result["roofline"].get("stream_read") or {}).get("GB_per_s") else None,
                                              "bins_per_sec": result["config"]["bins"] / (f4["avg_launch_ms"] * 1e-3)}
        # how long the whole run took on this rank and what of it was input synthesis / CPU legs: at N > 1 no CPU leg runs (they are
        # rank 0's at N = 1 only), and the driver's limit for a scaling run is 1 800 s
        result["run_s"] = {"total": time.perf_counter() - t_run0, "sv_bam_generation": (result.get("sv_e2e") or {}).get("bam_generation_s"),
                           "cpu_legs": "rank 0 at N = 1 only" if world > 1 or args.no_cpu_baseline else "included"}
        emit(result, args)
    if use_dist:
        dist.destroy_process_group()


# ---- the printed line
# The driver reads ONE JSON line from stdout and keeps a bounded tail of it: round 4's 20-KB line (every section's detail inline)
# no longer fitted and was recorded as unparsed.  The printed line is therefore the COMPACT record (the contract's keys, the whole
# `roofline` with every section's {ms, frac, traffic_ratio, frac_of_stream_read}, `cpu_baseline`, and per section the figures a
# reader quotes); the detailed record of the same run goes to gpurun_out/bench_detail_n<N>.json (or $TIDDIT_BENCH_DETAIL) and to
# stdout only with --full-line.
LINE_BUDGET = 7000        # (round 3's 16.8-KB line was parsed, round 4's 20-KB line was not; round 5 printed 4.7 KB)


def _short(v, n=96):
    if isinstance(v, str) and len(v) > n:
        return v[:n - 3] + "..."
    if isinstance(v, float):
        return float("%.6g" % v)
    return v


def _pick(d, keys, n=96):
    return {k: _short(d[k], n) for k in keys if d is not None and k in d and d[k] is not None}


def compact_line(result):
    line = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = result.get("vs_baseline")
    line.update(_pick(result, ("dtype", "data", "reads_per_sec")))
    cfg = result.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "reads", "bins", "launches_per_step"), 170)
    if cfg.get("layout_short"):                                  # (what the run itself says its input layout was — nothing is asserted here)
        line["config"]["layout"] = _short(cfg["layout_short"], 240)
    rf = result.get("roofline") or {}
    r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "avg_launch_ms", "median_launch_ms",
                   "min_launch_ms", "algorithmic_bytes_per_launch", "frac_of_stream_read", "bytes_model", "pack_binned_from_four_arrays_ms",
                   "binning_ms_per_600M_reads"), 80)
    if "traffic" not in r:
        r["traffic"] = None
    if r["traffic"] is not None and rf.get("traffic_source"):    # (only a run that found a matching profiles/traffic.json quotes it)
        r["traffic_source"] = _short(rf["traffic_source"], 200)
    sr = rf.get("stream_read")
    if sr:
        r["stream_read"] = _pick(sr, ("GB_per_s", "frac_of_peak", "bytes"))
    if rf.get("from_four_arrays_incl_pack"):
        r["from_four_arrays_incl_pack"] = _pick(rf["from_four_arrays_incl_pack"], ("ms", "frac"))
    if rf.get("contract"):
        r["contract"] = _pick(rf["contract"], ("ms", "frac", "frac_survey_8d_12B_per_read", "frac_of_stream_read", "bins_per_sec", "layout"), 130)
        # the same as scalar keys of `roofline`: the driver's parser keeps scalars only
        r.update({"contract_" + k: _short(v) for k, v in rf["contract"].items() if k in ("ms", "frac", "bins_per_sec", "frac_of_stream_read") and v is not None})
    if sr and sr.get("GB_per_s") is not None:
        r["stream_read_GBps"] = _short(sr["GB_per_s"])
    secs = {}
    for k, v in (rf.get("sections") or {}).items():
        secs[k] = {kk: _short(vv) for kk, vv in v.items() if kk != "bound" and vv is not None}
    for sec, key in (("coverage_sv", "cov_sv"), ("dbscan", "dbscan"), ("gc", "gc")):
        if secs.get(sec, {}).get("frac") is not None:
            r[key + "_frac"] = secs[sec]["frac"]
            r[key + "_ms"] = secs[sec].get("ms")
    if secs.get("ingest", {}).get("records_per_sec") is not None:
        r["ingest_records_per_sec"] = secs["ingest"]["records_per_sec"]
    if secs.get("sv_e2e", {}).get("wall_s") is not None:
        r["sv_e2e_wall_s"] = secs["sv_e2e"]["wall_s"]
    r["sections"] = secs
    line["roofline"] = r
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "reads_per_sec", "sample"), 120)
    ca = result.get("cpu_baseline_all_cores")
    if ca:
        line["cpu_baseline_all_cores"] = _pick(ca, ("value", "unit", "cores", "host_cores", "cpu_model"), 60)
    if "parity_checked" in result:
        line["parity_checked"] = _short(result["parity_checked"], 120)
    if result.get("run_s"):
        line["run_s"] = {k: _short(v) for k, v in result["run_s"].items() if v is not None}

    def section(name, keys, sub=()):
        s = result.get(name)
        if not s:
            return
        o = _pick(s, keys, 110)
        for sk, skeys in sub:
            if s.get(sk):
                o[sk] = _pick(s[sk], skeys, 80)
        line[name] = o

    cpu = ("cpu_baseline", ("value", "unit", "cores", "kind"))
    section("coverage_sv", ("value", "unit", "ms_per_step", "parity_checked"), (cpu,))
    section("dbscan", ("metric", "value", "unit", "ms_per_step", "parity_checked"),
            (("roofline", ("achieved", "frac", "traffic", "avg_pass_ms", "algorithmic_bytes_per_pass")), cpu))
    sd = (result.get("dbscan") or {}).get("sort_dbscan")
    if sd and "dbscan" in line:
        cc = sd.get("cluster_columns") or {}
        line["dbscan"]["cluster_columns"] = {"one_bucket": _pick(cc.get("one_bucket"), ("signals", "ms", "value")),
                                              "many_buckets": _pick(cc.get("many_buckets"), ("signals", "buckets", "ms", "value")), "unit": "signals/s"}
    section("dbscan_shared", ("value", "unit", "ms_per_step"))
    section("gc", ("value", "unit", "ms_per_step", "parity_checked"), (cpu,))
    section("ingest", ("value", "unit", "ms_per_step", "bam_MB_per_sec"))
    sv = result.get("sv_e2e")
    if sv:
        o = _pick(sv, ("metric", "value", "unit", "wall_s", "serial_s", "candidates", "speedup_vs_1_core_estimate"), 60)
        o["config"] = _pick(sv.get("config") or {}, ("workload",), 150)
        st = sv.get("stage_seconds")
        if isinstance(st, dict):
            o["stage_seconds"] = {k[:28]: _short(v) for k, v in st.items() if isinstance(v, (int, float)) and not k.startswith(" ") and v >= 0.002}
        if sv.get("cpu_baseline"):
            o["cpu_baseline"] = _pick(sv["cpu_baseline"], ("value", "unit", "cores", "kind", "sample_fraction_of_genome"))
        if sv.get("cpu_baseline_all_cores"):
            o["cpu_baseline_all_cores"] = _pick(sv["cpu_baseline_all_cores"], ("value", "unit", "cores"))
        o["parity_checked"] = _short(sv.get("parity_checked"), 110)
        line["sv_e2e"] = o
    return line


def emit(result, args):
    path = os.environ.get("TIDDIT_BENCH_DETAIL") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out",
                                                                 "bench_detail_n%d.json" % result.get("n_gpus", 1))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(json.dumps(result) + "\n")
        detail = os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
    except OSError as e:
        detail = "not written: %s" % e
    if getattr(args, "full_line", False):
        try:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)
        return
    # whatever native libraries still hold in C stdio buffers (RCCL's start-up banner is written through libc's stdout and would
    # otherwise be flushed at exit, BEHIND the JSON line) goes out first: the record is the last line of stdout
    try:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    line = compact_line(result)
    line["detail"] = detail
    text = json.dumps(line)
    # the budget is a hard promise: drop the per-section extras (never the contract's keys, roofline or cpu_baseline) until it holds
    for k in ("sv_e2e", "ingest", "gc", "dbscan_shared", "coverage_sv", "dbscan", "cpu_baseline_all_cores"):
        if len(text) <= LINE_BUDGET:
            break
        line.pop(k, None)
        text = json.dumps(line)
    print(text, flush=True)


def shared_bucket_sizes(total, seed=99):
    """~300 (chrA,chrB) buckets shaped like a human WGS run: the 24 intra-chromosomal buckets hold 85 % of the signals
    (GRCh38's relative chromosome lengths), 276 inter-chromosomal ones share the rest; shuffled."""
    rng = np.random.default_rng(seed)
    w_in = np.array([248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57], dtype=np.float64)
    sizes = np.concatenate([(0.85 * total * w_in / w_in.sum()).astype(np.int64), (0.15 * total * rng.dirichlet(np.ones(276) * 0.7)).astype(np.int64)])
    return sizes[rng.permutation(len(sizes))]


def shared_step(bucket_sizes, cluster_local, use_dist, group=None):
    """One pass over the shared bucket list: this rank clusters the buckets tiddit_amd.dist.shard_buckets gives it and the label
    arrays of ALL ranks are assembled on every rank (dist.cluster_buckets_distributed: one padded all-gather, int32 on the wire).
    With one rank there is nothing to exchange.  -> (owned, per-rank label arrays); dist.split_gathered gives the per-bucket view"""
    from tiddit_amd import dist as tdist
    if not use_dist:
        return [list(range(len(bucket_sizes)))], [cluster_local(list(range(len(bucket_sizes))))]
    return tdist.cluster_buckets_distributed(bucket_sizes, cluster_local, group, flat=True)


def dbscan_shared(args, ctx, stream, dev, rank, world, use_dist, barrier, wire=None, rank_max=lambda x: x):
    import torch
    import torch.distributed as dist
    from tiddit_amd import _native, dist as tdist, synth
    sizes = shared_bucket_sizes(2 * args.dbscan_n)
    total = int(sizes.sum())
    owned = tdist.shard_buckets(sizes, world)[rank]
    L = 250_000_000
    pts = {b: synth.gen_points(int(sizes[b]), L=L, seed=synth.SEED + 5000 + b) for b in owned if sizes[b]}
    cat = [pts[b] for b in owned if sizes[b]]
    xs = np.concatenate([p[:, 0] for p in cat]) if cat else np.zeros(0, np.int64)
    ys = np.concatenate([p[:, 1] for p in cat]) if cat else np.zeros(0, np.int64)
    nmine = len(xs)
    x = torch.from_numpy(xs.astype(np.uint32).view(np.int32)).to(dev)
    y = torch.from_numpy(ys.astype(np.uint32).view(np.int32)).to(dev)
    lab = torch.empty(max(nmine, 1), dtype=torch.float64, device=dev)
    off = np.concatenate([[0], np.cumsum([int(sizes[b]) for b in owned])]).astype(np.int64)
    ev = []

    def cluster_local(ids):
        assert list(ids) == list(owned)
        with torch.cuda.stream(stream):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            if nmine:
                _native.check(ctx.lib.tdt_dbscan_device(ctx.handle, x.data_ptr(), y.data_ptr(), nmine, _native.ptr(off), len(off) - 1,
                                                        ctypes.c_uint64(500), 3, 0, lab.data_ptr(), None))
            b.record(stream)
            ev.append((a, b))
            stream.synchronize()
        return lab[:nmine] if wire is None or wire == dev else lab[:nmine].to(wire)

    for _ in range(args.warmup):
        shared_step(sizes, cluster_local, use_dist)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        owned_all, parts = shared_step(sizes, cluster_local, use_dist)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t = rank_max(time.perf_counter() - t0)
    labels = tdist.split_gathered(sizes, owned_all, parts)
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))
    res = {"metric": "signals clustered/sec, one shared list of (chrA,chrB) buckets", "value": total / (t / args.steps), "unit": "signals/s",
           "ms_per_step": 1e3 * t / args.steps, "device_ms_rank0": k_ms,
           "config": {"workload": "BASELINE configs[4] shape: %d buckets, %d signals (60x-shaped: 2 x the configs[2] count; largest bucket %d), e=500 l=3; "
                                  "buckets bin-packed over %d rank(s) by signal count, labels all-gathered (int32 on the wire) so every rank holds the whole cluster set"
                                  % (len(sizes), total, int(sizes.max()), world), "signals_rank0": nmine}}
    if rank == 0 and not args.no_cpu_baseline:
        import oracle
        t1 = time.perf_counter()
        for b in range(len(sizes)):
            if not sizes[b]:
                continue
            p = pts[b] if b in pts else synth.gen_points(int(sizes[b]), L=L, seed=synth.SEED + 5000 + b)
            want = oracle.dbscan_main(p, 500, 3)
            if not np.array_equal(labels[b].cpu().numpy().astype(np.float64), want):
                raise SystemExit("PARITY FAILURE: bucket %d of the shared list differs from the CPU oracle" % b)
        res["cpu_baseline"] = {"value": total / (time.perf_counter() - t1), "unit": "signals/s", "cores": 1, "kind": "port",
                               "sample": "all %d buckets: point generation + oracle C DBSCAN per bucket (every label of every rank's share verified)" % len(sizes)}
        res["parity_checked"] = True
    return res


def sv_e2e(args, ctx, with_oracle, rank=0, world=1, local_rank=0, barrier=lambda: None):
    """One `tiddit --sv --skip_assembly` run (tiddit_amd.__main__, the CLI a user calls) on a synthetic WGS-shaped BAM
    (tiddit_amd.synth_bam.write_wgs_sv_bam: 24 chromosomes + chrM + two short scaffolds, 30x, planted DEL/DUP/INV/BND), file in the
    page cache, with the per-stage wall the CLI records; then the CPU restatement of the same path (oracle/signal_oracle.py +
    oracle/cluster_oracle.py, one core) on the same file, timed, and every output compared with it."""
    import contextlib
    import hashlib
    import io
    import shutil
    from tiddit_amd import __main__ as cli, synth_bam
    mb = args.sv_mb
    d = os.path.join(os.environ.get("TIDDIT_BENCH_TMP", "/tmp"), "tiddit_bench_sv_%d" % mb)
    bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
    contigs = synth_bam.wgs_contigs(mb)
    t_gen = None
    if local_rank == 0 and not (os.path.exists(bam) and os.path.exists(fa)):      # one file per node, every rank reads its byte range
        os.makedirs(d, exist_ok=True)
        t0 = time.perf_counter()
        seqs = synth_bam.write_fasta(fa, contigs)
        synth_bam.write_wgs_sv_bam(bam + ".tmp", contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
        del seqs
        os.replace(bam + ".tmp", bam)
        t_gen = time.perf_counter() - t0
    barrier()
    out = os.path.join(d, "run%d" % world)
    walls, stages = [], None
    for rep in range(2):                                            # first pass warms the page cache and the device buffers
        if rank == 0:
            shutil.rmtree(out + "_tiddit", ignore_errors=True)
        barrier()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite"])
        ctx.sync()
        barrier()                                                   # the job is done when its slowest rank is
        walls.append(time.perf_counter() - t0)
        stages = dict(cli.STAGE_SECONDS)
        if rep == 0:
            first_stages = stages
    all_stages = None
    if world > 1:
        import torch.distributed as dist
        all_stages = [None] * world
        dist.all_gather_object(all_stages, stages)
        all_notes = [None] * world
        dist.all_gather_object(all_notes, dict(cli.STAGE_NOTES))
    if rank != 0:
        return None
    res = {"metric": "tiddit --sv --skip_assembly end to end (BAM file -> candidates table), wall seconds", "wall_s": walls[-1],
           "first_pass_wall_s": walls[0], "stage_seconds": {k: round(v, 4) for k, v in stages.items()},
           "first_pass_stage_seconds": {k: round(v, 4) for k, v in first_stages.items()},      # (a fresh process: allocations, first touches)
           "config": {"workload": "BASELINE configs[3]: %d-Mb genome (24 chromosomes + chrM + 2 scaffolds), 30x 150-bp pairs, planted DEL/DUP/INV/BND at 3 per Mb; "
                                  "%.0f MB BAM (zlib level 1, reads cut from the reference), file in the page cache%s" % (mb, os.path.getsize(bam) / 1e6,
                                  "" if world == 1 else "; ONE job on %d ranks: byte-range shards of the file, rows sent once to the owner rank of their chrA, every owner writes its blocks and clusters / regroups its buckets" % world)},
           "bam_generation_s": t_gen, "candidates": sum(1 for l in open(out + ".candidates.tab") if not l.startswith("#"))}
    # what only rank 0 does (at N = 1: the stages an N-rank job could not share out) — the Amdahl term of DESIGN.md section 6
    serial_keys = ("library statistics", "GC bins", "ploidy (masked medians)", "candidates table", "  candidates to rank 0")
    res["serial_s"] = round(sum(v for k, v in stages.items() if k in serial_keys), 4)
    res["serial_stages"] = list(serial_keys)
    if all_stages is not None:
        keys = []
        for st in all_stages:
            keys += [k for k in st if k not in keys]
        res["stage_seconds_max_over_ranks"] = {k: round(max(st.get(k, 0.0) for st in all_stages), 4) for k in keys}
        res["stage_seconds_per_rank"] = [{k: round(v, 4) for k, v in st.items()} for st in all_stages]
        res["notes_per_rank"] = all_notes
    if with_oracle:
        res.update(sv_e2e_cpu_legs(mb, bam, fa, out, contigs, walls[-1], args.sv_cpu_full_mb))
    return res


def sv_e2e_cpu_legs(mb, bam, fa, out, contigs, gpu_wall, full_mb=300):
    """The CPU restatement of the same stages on the same file, timed on one core and on all cores, and the product's outputs compared
    with it.  Up to 300 Mb the whole file; beyond that a BOUNDED SAMPLE: the library statistics as they are (a prefix of the file) and,
    for everything else, a run of contigs at the end of the genome holding >= 4 % of it (oracle/signal_oracle.signal_main_sample finds
    them by a binary search over the BGZF blocks) — rows, clips, coverage, GC, ploidy medians and candidates of those contigs."""
    import contextlib
    import io
    import shutil
    import oracle
    from oracle import cluster_oracle, signal_oracle
    from tiddit_amd import bamio, tiddit_cluster, tiddit_signal, tiddit_stats
    names = [n for n, _ in contigs]
    length = dict(contigs)
    big = [t for t, (n, ln) in enumerate(contigs) if ln >= 10000]
    total = float(sum(ln for _, ln in contigs))
    if mb <= full_mb:
        tids = list(big)
    else:
        tids, acc = [], 0
        for t in reversed(big):
            tids.insert(0, t)
            acc += contigs[t][1]
            if acc >= 0.04 * total and len(tids) >= 6:            # (several contigs, so that the all-cores leg has something to fan out)
                break
    frac = sum(contigs[t][1] for t in tids) / total
    S = [names[t] for t in tids]
    inS = set(S)
    ncores = os.cpu_count() or 1
    T1 = {}
    # ---- library statistics (one core; the all-cores leg cannot split a sequential sample either: the reference does not)
    t0 = time.perf_counter()
    lib_cpu = signal_oracle.statistics_prefix(bam, 5, 100000, 25000000)
    T1["library statistics"] = time.perf_counter() - t0
    with contextlib.redirect_stdout(io.StringIO()):
        lib = tiddit_stats.statistics(bam, fa, 5, 100000, 25000000)
    bamio.set_carry(None)
    if any(lib[k] != v for k, v in lib_cpu.items()):
        raise SystemExit("PARITY FAILURE: library statistics differ from the CPU restatement")
    max_ins = lib["percentile_insert_size"]
    # ---- signal extraction + coverage: one core, then one process per contig on all cores
    t0 = time.perf_counter()
    cov, disc, split, clip_each, n_rec = signal_oracle.signal_main_sample(bam, tids, 5, max_ins, 60, 25, 1)
    T1["signal extraction + coverage"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    covP, discP, splitP, clipP, _ = signal_oracle.signal_main_sample(bam, tids, 5, max_ins, 60, 25, min(ncores, len(tids)))
    t_sig_all = time.perf_counter() - t0
    if discP != disc or splitP != split or clipP != clip_each or any(not np.array_equal(covP[c], cov[c]) for c in S):
        raise SystemExit("PARITY FAILURE: the CPU restatement on all cores differs from the one-core run")
    keep = lambda text: "".join(l + "\n" for l in text.splitlines() if l.split("\t")[1] in inS and l.split("\t")[2] in inS)
    if keep(open(out + "_tiddit/discordants_WGS.tab").read()) != keep(disc) or keep(open(out + "_tiddit/splits_WGS.tab").read()) != keep(split):
        raise SystemExit("PARITY FAILURE: signal tables differ from the CPU restatement")
    for c in S:
        if open(out + "_tiddit/clips/%s.fa" % c).read() != clip_each[c]:
            raise SystemExit("PARITY FAILURE: clip FASTA of %s differs from the CPU restatement" % c)
    with contextlib.redirect_stdout(io.StringIO()):
        _, chroms, gcov, _, _, _ = tiddit_signal.scan_signals(bam, 5, max_ins, 10000, 60, 25, 50)
    for c in S:
        if not np.array_equal(gcov[c], cov[c]):
            raise SystemExit("PARITY FAILURE: 50-bp coverage of %s differs from the CPU restatement" % c)
    # ---- GC bins (C port of the per-character loop) and the masked medians of determine_ploidy (numpy, as the reference)
    from tiddit_amd import tiddit_gc
    t0 = time.perf_counter()
    seqs = {}
    with open(fa, "rb") as f:                                   # (plain FASTA read; the reference goes through pysam.FastaFile)
        name = None
        for block in f.read().split(b">")[1:]:
            head, _, body = block.partition(b"\n")
            name = head.split()[0].decode()
            if name in inS:
                seqs[name] = np.frombuffer(body.replace(b"\n", b""), dtype=np.uint8)
    gc_cpu = {c: oracle.binned_gc(seqs[c], 50, 0.5) for c in S}
    T1["GC bins"] = time.perf_counter() - t0
    gc_gpu = tiddit_gc.main(fa, S, 1, 50, 0.5)
    if any(not np.array_equal(np.asarray(gc_gpu[c]), gc_cpu[c]) for c in S):
        raise SystemExit("PARITY FAILURE: GC bins differ from the CPU port")
    t0 = time.perf_counter()
    med_cpu = [float(np.median(cov[c][(cov[c] > 0) & (gc_cpu[c][:len(cov[c])] != -1)])) for c in S]
    T1["ploidy (masked medians)"] = time.perf_counter() - t0
    from tiddit_amd import tiddit_coverage_analysis as tca
    med_gpu, _ = tca.masked_medians([(gcov[c], gc_gpu[c]) for c in S])
    if [float(x) for x in med_gpu] != med_cpu:
        raise SystemExit("PARITY FAILURE: masked coverage medians differ from numpy.median")
    # ---- clustering of the contig pairs inside the run
    tmp = out + "_cpu"
    shutil.rmtree(tmp + "_tiddit", ignore_errors=True)
    os.makedirs(tmp + "_tiddit")
    open(tmp + "_tiddit/discordants_WGS.tab", "w").write(disc)
    open(tmp + "_tiddit/splits_WGS.tab", "w").write(split)
    eps = int(lib["avg_insert_size"] / 2.0) or 50
    cargs = (S, length, ["WGS"], lib["mp"], eps, 3, max_ins, 10000, True, 3)
    t0 = time.perf_counter()
    want = cluster_oracle.main(tmp, *cargs)
    T1["clustering"] = time.perf_counter() - t0
    got = tiddit_cluster.main(out, *cargs)
    if cluster_oracle.canonical(got) != cluster_oracle.canonical(want):
        raise SystemExit("PARITY FAILURE: candidates differ from the CPU restatement")
    one = sum(T1.values())
    allc = one - T1["signal extraction + coverage"] + t_sig_all
    scale = lambda T, sig: T["library statistics"] + (sig + T["GC bins"] + T["ploidy (masked medians)"] + T["clustering"]) / frac
    what = "the whole file" if frac > 0.999 else "library statistics on the file's sampled prefix; the other stages on the last %d contigs (%s .. %s: %.1f %% of the genome, %d records)" % (
        len(S), S[0], S[-1], 100 * frac, n_rec)
    return {"records_in_cpu_sample": int(n_rec),
            "cpu_baseline": {"value": one, "unit": "s", "cores": 1, "kind": "port", "stage_seconds": {k: round(v, 3) for k, v in T1.items()},
                             "sample_fraction_of_genome": frac, "whole_file_estimate_s": scale(T1, T1["signal extraction + coverage"]),
                             "sample": what + "; the same stages as the GPU run: statistics (zlib inflate of the prefix + the sampling loop + numpy), signal extraction + "
                                              "50-bp coverage (zlib inflate, C record walk + tiddit_signal.worker chain, Python rows), GC (C port), masked medians (numpy), "
                                              "clustering (C DBSCAN restatement + Python regroup)"},
            "cpu_baseline_all_cores": {"value": allc, "unit": "s", "cores": min(ncores, len(tids)), "host_cores": ncores, "cpu_model": cpu_model(), "kind": "port",
                                       "signal_extraction_s": t_sig_all, "whole_file_estimate_s": scale(T1, t_sig_all),
                                       "sample": "the same sample; signal extraction as one process per contig (the reference's own fan-out, tiddit_signal.pyx:259), the "
                                                 "other stages as in the one-core leg (sequential in the reference too)"},
            "speedup_vs_1_core_estimate": scale(T1, T1["signal extraction + coverage"]) / gpu_wall,
            "parity_checked": "library statistics; for %s: signal tables (rows of contig pairs inside the sample), clip FASTA, 50-bp coverage, GC bins, masked medians and the "
                              "candidates dictionary equal the CPU restatement (one core == all cores)" % ("every contig" if frac > 0.999 else "the sampled contigs")}


if __name__ == "__main__":
    main()
