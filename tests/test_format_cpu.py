"""The native coverage table writer (tdt_format_coverage, host code) against the reference's Python formatting
(`"{}".format(numpy.float64)` rows of print_coverage, tiddit_coverage.pyx:30-44) — CPU only."""
import ctypes
import os

import numpy as np

from tiddit_amd import _native, build, tiddit_coverage


def _native_rows(values, name, bin_size, ln, kind):
    lib = _native.load()
    values = np.ascontiguousarray(values, dtype=np.float64)
    size = ctypes.c_size_t(0)
    _native.check(lib.tdt_format_coverage(_native.ptr(values), len(values), name.encode(), bin_size, ln, kind, None, 0, ctypes.byref(size)))
    buf = np.empty(size.value + 1, dtype=np.uint8)
    _native.check(lib.tdt_format_coverage(_native.ptr(values), len(values), name.encode(), bin_size, ln, kind, _native.ptr(buf), len(buf), ctypes.byref(size)))
    return bytes(buf[:size.value]).decode()


def test_repr_layout_matches_python_on_hard_values():
    build.build()
    rng = np.random.default_rng(3)
    specials = [0.0, 1.0, 2.0, 10.0, 100.0, 0.5, 0.1, 0.30000001192092896, 0.0020000000949949026, 0.7226277589797974, 1e-4, 9.999e-5, 1e-5,
                5e-324, 2.2250738585072014e-308, 1e15, 1e16, 9999999999999998.0, 1.2345678901234567e16, 123456789012345680.0, 1e21, 1e22, 1e100,
                1.7976931348623157e308, 0.1 + 0.2, 1 / 3, 2 / 3, 1e-7, 123456.789, 4.35, 0.000123456, 1234567.0, 1e-310, float(2 ** 53),
                float(2 ** 53 + 2), 33.333333333333336, 149.99999999999997, -0.0, -1.5, -2e-7, float("inf"), float("-inf"), float("nan")]
    q32 = (rng.integers(0, 3000, 20000).astype(np.float32) / np.float32(500)).astype(np.float64)            # the quotients bins are made of
    sums = np.cumsum(q32)                                                                                     # ... and sums of them
    wide = np.exp(rng.uniform(-40, 40, 20000)) * rng.choice([1.0, 1.0, 1.0], 20000)
    ints = rng.integers(0, 10 ** 17, 2000).astype(np.float64)
    vals = np.concatenate([np.array(specials), q32, sums, wide, ints])
    got = _native_rows(vals, "x", 7, 10, 1).split("\n")[:-1]
    want = ["{}".format(v) for v in vals]
    bad = [(w, g) for w, g in zip(want, got) if w != g]
    assert not bad, bad[:5]
    assert len(got) == len(vals)


def test_bed_and_wig_rows_match_the_python_loop(tmp_path):
    build.build()
    rng = np.random.default_rng(4)
    header = {"SQ": [{"SN": "chr1", "LN": 1_000_137}, {"SN": "HLA-A*01:01:01:01", "LN": 3503}, {"SN": "tiny", "LN": 20}]}
    for z in (500, 50, 1, 977):
        cov = {c["SN"]: np.cumsum(rng.integers(0, 400, -(-c["LN"] // z)).astype(np.float32) / np.float32(z)).astype(np.float64) % 97
               for c in header["SQ"]}
        for kind in ("bed", "wig"):
            out = str(tmp_path / ("o.%s" % kind))
            tiddit_coverage.print_coverage(cov, header, z, kind, out)
            want = "#chromosome\tstart\tend\tcoverage\n" if kind == "bed" else "track type=wiggle_0 name=\"Coverage\" description=\"Per bin average coverage\"\n"
            for c in header["SQ"]:
                if kind == "wig":
                    want += "fixedStep chrom={} start=1 step={}\n".format(c["SN"], z)
                want += tiddit_coverage._rows_python(cov[c["SN"]], c["SN"], c["LN"], z, kind)
            assert open(out).read() == want, (z, kind)


def test_native_fai_matches_python_index(tmp_path):
    """tdt_fasta_write_fai (one memchr pass) writes the same .fai as the line-by-line Python indexer: LF and CRLF files, a last
    line without a line end, empty sequences, descriptions after the name, lines longer than the read buffer"""
    build.build()
    from tiddit_amd import fasta
    rng = np.random.default_rng(6)
    for eol, last_eol in ((b"\n", True), (b"\r\n", True), (b"\n", False)):
        p = str(tmp_path / ("x%d%d.fa" % (len(eol), last_eol)))
        with open(p, "wb") as f:
            for name, ln, width in (("chr1 first contig", 12345, 60), ("c2\tdesc", 61, 61), ("empty", 0, 60), ("long", 20_000_000, 20_000_000),
                                    ("last", 777, 70)):
                f.write(b">" + name.encode() + eol)
                s = np.array(list(b"ACGTN"), np.uint8)[rng.integers(0, 5, ln)].tobytes()
                for o in range(0, ln, width):
                    f.write(s[o:o + width] + eol)
            if not last_eol:
                f.seek(-len(eol), 2)
                f.truncate()
        want = fasta.build_fai(p)
        py = open(p + ".fai").read()
        _native.check(_native.load().tdt_fasta_write_fai(p.encode(), (p + ".fai2").encode()))
        assert open(p + ".fai2").read() == py and len(want) == 5


def test_variant_stage_hand_off_with_stub_modules(tmp_path):
    """`tiddit --sv` ends by handing the candidates to the reference's own variant stage when that package is importable
    (__main__.py:193-207); it never is in this image (it needs pysam), so the hand-off is exercised with stand-in modules: same
    call signatures, variants of every contig written sorted by position behind the header, contigs in header order"""
    import types
    from tiddit_amd import __main__ as cli
    seen = {}

    def variant_main(bam, sv_clusters, args, library, min_mapq, samples, coverage_data, contig_number, max_ins_len, gc_dictionary):
        seen["args"] = (bam, sorted(sv_clusters), library["mp"], min_mapq, samples, sorted(coverage_data), contig_number, max_ins_len, sorted(gc_dictionary))
        return {"chr2": [(5, ["chr2", "5", "SV_2"]), (1, ["chr2", "1", "SV_1"])], "chr1": [(9, ["chr1", "9", "SV_3"])], "chrUn": [(1, ["x"])]}
    variant = types.SimpleNamespace(main=variant_main)
    header = types.SimpleNamespace(main=lambda bam_header, library, sample_id, version: "##fileformat=VCFv4.1 %s %s" % (sample_id, version))
    args = types.SimpleNamespace(bam="x.bam")
    prefix = str(tmp_path / "o")
    ok = cli.variant_stage(variant, header, prefix, ["chr1", "chr2", "chr3"], {"SQ": []}, {"mp": False}, "S", "3.9.5", args, {"chr1": {}}, 5, ["S"],
                           {"chr1": None}, {"chr1": 0}, 600, {"chr1": None})
    assert ok and open(prefix + ".vcf").read() == "##fileformat=VCFv4.1 S 3.9.5\nchr1\t9\tSV_3\nchr2\t1\tSV_1\nchr2\t5\tSV_2\n"
    assert seen["args"] == ("x.bam", ["chr1"], False, 5, ["S"], ["chr1"], {"chr1": 0}, 600, ["chr1"])
    assert not cli.variant_stage(None, None, prefix + "2", [], {}, {}, "S", "v", args, {}, 5, [], {}, {}, 1, {}) and not os.path.exists(prefix + "2.vcf")
