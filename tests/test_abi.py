"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/tiddit_hip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from tiddit_amd import build, _native
    build.build()
    return _native


def declared_symbols():
    hdr = open(os.path.join(REPO, "include", "tiddit_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tdt_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(native):
    lib = native.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libtiddit_hip.so does not export " + n
    assert sorted(native.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.tdt_version() >= 100


def test_every_reference_citation_in_header():
    hdr = open(os.path.join(REPO, "include", "tiddit_hip.h")).read()
    for ref in ("tiddit_coverage.pyx:10-21", "tiddit_coverage.pyx:48-74", "__main__.py:229-242", "tiddit_signal.pyx:169-182",
                "tiddit_gc.pyx:14-31", "DBSCAN.py:33-64", "tiddit_cluster.pyx:152-154"):
        assert ref in hdr, ref


def test_no_cpu_fallback_without_gpu(native):
    lib = native.load()
    n = ctypes.c_int(0)
    rc = lib.tdt_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.TdtError):
        native.Context(0)
    from tiddit_amd import DBSCAN
    with pytest.raises(native.TdtError):
        DBSCAN.main([[1, 2], [2, 3], [3, 4]], 5, 2)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "tiddit_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
