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
    """Second pass over the labels produced by ``x_coordinate_clustering`` for the same
    (data, epsilon, m); ``clusters`` is updated in place and returned, like the reference."""
    xl, xid = _run(data, epsilon, m, 1)
    if xid != cluster_id or not numpy.array_equal(xl, clusters):
        raise NotImplementedError("y_coordinate_clustering expects the labels/cluster_id returned by "
                                  "x_coordinate_clustering(data, epsilon, m)")
    yl, yid = _run(data, epsilon, m, 0)
    clusters[:] = yl
    return clusters, yid


def main(data, epsilon, m):
    return _run(data, epsilon, m, 0)[0]
