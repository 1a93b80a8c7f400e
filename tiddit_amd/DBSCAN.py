"""Drop-in for ``tiddit.DBSCAN`` (DBSCAN.py) on the MI355X.

``main(data, epsilon, m)`` (:125-129), ``x_coordinate_clustering`` (:33-64) and
``y_coordinate_clustering`` (:66-123) with the reference's arguments and return values: float64
labels in the order of ``data`` (callers sort by posA first, tiddit_cluster.pyx:152), -1 = noise.
All three run the closed form of the reference's run-labelling state machine in HIP
(csrc/tdt_dbscan.hip).  The reference's dead ``generate_clusters`` (:5-31, calls an undefined
function) is not reproduced.
"""
import ctypes

import numpy

from . import _native


def _as_data(data):
    data = numpy.asarray(data)
    if data.ndim != 2:
        raise IndexError("data must be a 2-d array [n, >=1]")
    if not numpy.issubdtype(data.dtype, numpy.integer):
        raise TypeError("tiddit_amd.DBSCAN works on integer coordinates (got dtype %s)" % data.dtype)
    return numpy.ascontiguousarray(data, dtype=numpy.int64)


def _run(data, epsilon, m, mode, ctx=None):
    ctx = ctx or _native.default_context()
    data = _as_data(data)
    n, stride = data.shape
    m = int(m)
    if m < 2:
        if mode == 1 and n - m + 1 <= 0:
            return numpy.full(n, -1.0), -1
        raise ValueError("max() arg is an empty sequence")  # what the reference raises for m == 1
    labels = numpy.empty(n, dtype=numpy.float64)
    last = ctypes.c_int64(-1)
    _native.check(ctx.lib.tdt_dbscan(ctx.handle, _native.ptr(data), n, stride, float(epsilon), m, mode, _native.ptr(labels),
                                     ctypes.byref(last)))
    return labels, int(last.value)


def x_coordinate_clustering(data, epsilon, m):
    return _run(data, epsilon, m, 1)


def y_coordinate_clustering(data, epsilon, m, cluster_id, clusters):
    """Second pass (:66-123) over x labels: ``clusters`` (float64, -1 or cluster numbers) is relabelled in place and returned
    with the final ``cluster_id``, like the reference.  Any label array of the shape ``x_coordinate_clustering`` produces is
    taken as it is (contiguous clusters numbered 0, 1, 2, ... along the array — e.g. labels from another eps / m); it does not
    have to come from this module.  Clusters above 128 members take the ``main`` route, which needs the labels to be the x pass
    of the same ``(data, epsilon, m)``."""
    ctx = _native.default_context()
    data = _as_data(data)
    n, stride = data.shape
    lab = numpy.ascontiguousarray(clusters, dtype=numpy.float64).copy()
    last = ctypes.c_int64(int(cluster_id))
    rc = ctx.lib.tdt_dbscan_y(ctx.handle, _native.ptr(data), n, stride, float(epsilon), int(m), int(cluster_id), _native.ptr(lab), ctypes.byref(last))
    if rc == -6:                                  # TDT_E_UNSUPPORTED: large clusters / m > 64 / another label shape
        xl, xid = _run(data, epsilon, m, 1)
        if xid != cluster_id or not numpy.array_equal(xl, clusters):
            raise NotImplementedError("y_coordinate_clustering: " + ctx.lib.tdt_last_error().decode())
        yl, yid = _run(data, epsilon, m, 0)
        clusters[:] = yl
        return clusters, yid
    _native.check(rc)
    clusters[:] = lab
    return clusters, int(last.value)


def main(data, epsilon, m):
    return _run(data, epsilon, m, 0)[0]
