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


def test_in_tree_build_is_not_a_measurement_build(native):
    """tdt_build_flags: the product library was compiled with none of the ablation / tunable macros of tools/build_variant.sh"""
    assert native.load().tdt_build_flags() == b""


def test_measurement_build_is_refused_unless_allowed(tmp_path):
    """a library compiled with an ablation macro reports it and tiddit_amd._native refuses to load it (TIDDIT_ALLOW_VARIANT=1 overrides):
    built here from the in-tree objects with ONE translation unit recompiled under a tunable (-DRS_ROUNDS=8)"""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(REPO, "tiddit_amd", "csrc")
    objs = sorted(f for f in os.listdir(csrc) if f.endswith(".o"))
    if not os.path.exists(hipcc) or "tdt_sort.o" not in objs:
        pytest.skip("no hipcc / no in-tree objects")
    var = str(tmp_path / "tdt_sort_var.o")
    so = str(tmp_path / "lib_var.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-DRS_ROUNDS=8", "-I" + os.path.join(REPO, "include"), "-c",
                    os.path.join(csrc, "tdt_sort.hip"), "-o", var], check=True, capture_output=True, timeout=600)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(csrc, o) for o in objs if o != "tdt_sort.o"] + [var, "-o", so, "-lz", "-lpthread", "-ldl"],
                   check=True, capture_output=True, timeout=600)
    code = "import sys; sys.path.insert(0, %r); from tiddit_amd import _native; print(_native.load().tdt_build_flags().decode())" % REPO
    env = dict(os.environ, TIDDIT_HIP_LIB=so)
    env.pop("TIDDIT_ALLOW_VARIANT", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "measurement macros (RS_ROUNDS)" in out.stderr
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, TIDDIT_ALLOW_VARIANT="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split()[-1] == "RS_ROUNDS"


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


class _GatherOp(ctypes.Structure):
    _fields_ = [("root", ctypes.c_int), ("from_send", ctypes.c_int), ("offset_bytes", ctypes.c_size_t), ("nbytes", ctypes.c_size_t)]


def _plan(lib, rank, world, counts, displs, elem, cap=0):
    import numpy as np
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    d = np.ascontiguousarray(displs, dtype=np.uint64)
    ops = (_GatherOp * world)()
    n = ctypes.c_int(0)
    rc = lib.tdt_allgatherv_plan(rank, world, c.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p), elem, cap,
                                 ctypes.cast(ops, ctypes.c_void_p), ctypes.byref(n))
    return rc, [(o.root, o.from_send, o.offset_bytes, o.nbytes) for o in ops[:n.value]]


def test_allgatherv_layout_for_2_to_8_ranks(native):
    """tdt_allgatherv's count / displacement arithmetic without RCCL: the broadcast list of every rank, replayed on host buffers,
    leaves the concatenation of all contributions on every rank — for N = 2..8, empty contributions and gaps included"""
    import numpy as np
    lib = native.load()
    rng = np.random.default_rng(4)
    for world in range(2, 9):
        for elem in (1, 4, 8):
            for trial in range(6):
                counts = rng.integers(0, 50, world)
                counts[rng.integers(0, world)] = 0                         # at least one rank with nothing to send
                if trial == 0:
                    counts[:] = 0
                gap = rng.integers(0, 3, world)                            # displacements need not be dense
                displs = np.concatenate([[0], np.cumsum(counts + gap)[:-1]])
                total = int((displs + counts).max()) * elem
                send = [rng.integers(0, 256, int(counts[r]) * elem, dtype=np.uint8) for r in range(world)]
                recv = [np.full(total, 0xEE, dtype=np.uint8) for _ in range(world)]
                plans = []
                for r in range(world):
                    rc, ops = _plan(lib, r, world, counts, displs, elem, total)
                    assert rc == 0
                    plans.append(ops)
                    assert [o[0] for o in ops] == [q for q in range(world) if counts[q]]          # one broadcast per non-empty rank, rank order
                    assert [o[1] for o in ops] == [int(q == r) for q in range(world) if counts[q]]  # only the root reads its send buffer
                assert all([(o[0], o[2], o[3]) for o in p] == [(o[0], o[2], o[3]) for o in plans[0]] for p in plans)   # same group on all ranks
                for root, _, off, nb in plans[0]:                          # replay: ncclBroadcast(root) delivers the root's bytes everywhere
                    for r in range(world):
                        recv[r][off:off + nb] = send[root]
                for r in range(world):
                    for q in range(world):
                        o = int(displs[q]) * elem
                        assert np.array_equal(recv[r][o:o + int(counts[q]) * elem], send[q])
    # refused: overlapping ranges, ranges beyond the receive buffer, bad rank
    assert _plan(lib, 0, 2, [4, 4], [0, 2], 4)[0] != 0
    assert _plan(lib, 0, 2, [4, 4], [0, 4], 4, cap=31)[0] != 0
    assert _plan(lib, 0, 2, [4, 4], [0, 4], 4, cap=32)[0] == 0
    assert _plan(lib, 2, 2, [4, 4], [0, 4], 4)[0] != 0
