"""bench.py prints ONE bounded JSON line (the driver keeps a bounded tail of stdout; round 4's 20-KB line was recorded as unparsed):
the compact line made from a detailed record must keep the contract's keys, the roofline with every section's fractions and
the CPU baseline, and stay inside its budget whatever the sections hold."""
import glob
import importlib.util
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_for_line_test", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _records():
    out = []
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r0*_bench_*.json"))):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            if "roofline" in d and "value" in d:
                out.append((os.path.basename(f), d))
    return out


def test_compact_line_of_every_committed_record(bench, capsys, tmp_path, monkeypatch):
    recs = _records()
    assert recs
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")
    for name, d in recs:
        monkeypatch.setenv("TIDDIT_BENCH_DETAIL", str(tmp_path / name))

        class A:
            full_line = False
        bench.emit(d, A)
        printed = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
        assert len(printed) == 1 and len(printed[0]) <= bench.LINE_BUDGET, (name, len(printed[0]))
        line = json.loads(printed[0])
        for k in contract:
            assert k in line, (name, k)
        assert line["value"] == pytest.approx(d["value"], rel=1e-5) and line["config"]["workload"]
        rf = line["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in rf, (name, k)
        assert rf["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-5)
        if d["n_gpus"] == 1 and "cpu_baseline" in d:
            assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
        for sec, v in (d["roofline"].get("sections") or {}).items():
            assert sec in rf["sections"], (name, sec)
        # what the driver's parser keeps are the SCALAR keys of `roofline`: the contract launch and the sections' fractions are there too
        if d["roofline"].get("contract"):
            assert rf["contract_frac"] == pytest.approx(d["roofline"]["contract"]["frac"], rel=1e-5) and rf["contract_ms"] > 0 and rf["contract_bins_per_sec"] > 0
        for sec, key in (("coverage_sv", "cov_sv_frac"), ("dbscan", "dbscan_frac"), ("gc", "gc_frac")):
            if (d["roofline"].get("sections") or {}).get(sec):
                assert rf[key] == pytest.approx(d["roofline"]["sections"][sec]["frac"], rel=1e-5), (name, key)
        if (d["roofline"].get("stream_read") or {}).get("GB_per_s"):
            assert rf["stream_read_GBps"] == pytest.approx(d["roofline"]["stream_read"]["GB_per_s"], rel=1e-5)
        # provenance is quoted only when the run has it
        assert ("traffic_source" in rf) == (rf["traffic"] is not None and bool(d["roofline"].get("traffic_source")))
        assert ("layout" in line["config"]) == bool(d["config"].get("layout_short"))
        assert json.loads(open(str(tmp_path / name)).read()) == d         # the detailed record travels beside the line


def test_compact_line_quotes_no_provenance_the_run_does_not_have(bench, capsys, tmp_path, monkeypatch):
    """a record whose run found no matching profiles/traffic.json (traffic None) and that names no input layout prints neither"""
    name, d = _records()[-1]
    d = json.loads(json.dumps(d))
    d["roofline"]["traffic"] = None
    d["config"].pop("layout_short", None)
    monkeypatch.setenv("TIDDIT_BENCH_DETAIL", str(tmp_path / "d.json"))

    class A:
        full_line = False
    bench.emit(d, A)
    line = json.loads(capsys.readouterr().out.strip())
    assert line["roofline"]["traffic"] is None and "traffic_source" not in line["roofline"] and "layout" not in line["config"]


def test_compact_line_sheds_sections_before_it_breaks_the_budget(bench, capsys, tmp_path, monkeypatch):
    name, d = _records()[-1]
    d = json.loads(json.dumps(d))
    d.setdefault("sv_e2e", {})["stage_seconds"] = {"stage %d of a very long list of stage names" % i: 1.0 + i for i in range(400)}
    d["sv_e2e"].setdefault("config", {})["workload"] = "x" * 5000
    monkeypatch.setenv("TIDDIT_BENCH_DETAIL", str(tmp_path / "d.json"))

    class A:
        full_line = False
    bench.emit(d, A)
    out = capsys.readouterr().out.strip()
    assert len(out) <= bench.LINE_BUDGET
    line = json.loads(out)
    assert "roofline" in line and "sv_e2e" not in line and "sv_e2e" in line["roofline"]["sections"]
