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


def _y_segments(ctx, ydata, seg, keep, nseg, epsilon, m, cluster_id):
    """tdt_dbscan_y_segments: members (y, segment = visiting rank of their label, current label) -> (new labels, cluster_id)"""
    out = numpy.empty(len(ydata), dtype=numpy.float64)
    last = ctypes.c_int64(int(cluster_id))
    _native.check(ctx.lib.tdt_dbscan_y_segments(ctx.handle, _native.ptr(ydata), _native.ptr(seg), _native.ptr(keep), len(ydata), int(nseg),
                                                float(epsilon), int(m), int(cluster_id), _native.ptr(out), ctypes.byref(last)))
    return out, int(last.value)


def y_coordinate_clustering(data, epsilon, m, cluster_id, clusters):
    """Second pass (:66-123) over x labels: ``clusters`` (float64) is relabelled in place and returned with the final
    ``cluster_id``, like the reference.  ANY label array is taken, as the reference takes it (:68-75): labels of another eps / m,
    clusters of any size, a value on several index ranges, sparse / non-integer / negative values, and a ``cluster_id`` below the
    largest label (the ids produced then collide with clusters still to be visited and are merged by the later
    ``clusters == cluster`` masks, :72).

    Three routes, all on the device: (1) labels shaped like this module's own x pass (contiguous clusters 0, 1, 2, ... of at most
    128 members, ``cluster_id`` at or above the largest) -> the tile kernel (``tdt_dbscan_y``); (2) anything else -> one
    ``tdt_dbscan_y_segments`` call over all clusters, in the visiting order of Python's own ``set(clusters)`` (:68); (3) if an id
    produced by (2) could equal a value that was still to be visited, the clusters are replayed one at a time in that order,
    members taken by current value, one ``tdt_dbscan_y_segments`` call per cluster — the reference's O(K*N) loop."""
    ctx = _native.default_context()
    data = _as_data(data)
    n, stride = data.shape
    if int(cluster_id) != cluster_id:
        raise TypeError("cluster_id must be an integer")
    cluster_id = int(cluster_id)
    if int(m) < 2:                                # every window is empty: the reference's max() raises on the first non-empty cluster (:100)
        if (numpy.asarray(clusters) != -1).any():
            raise ValueError("max() arg is an empty sequence")
        return clusters, cluster_id
    lab = numpy.ascontiguousarray(clusters, dtype=numpy.float64).copy()
    last = ctypes.c_int64(cluster_id)
    rc = ctx.lib.tdt_dbscan_y(ctx.handle, _native.ptr(data), n, stride, float(epsilon), int(m), cluster_id, _native.ptr(lab), ctypes.byref(last))
    if rc == 0:
        clusters[:] = lab
        return clusters, int(last.value)
    if rc != -6:                                  # TDT_E_UNSUPPORTED: not the x pass's label shape / large clusters / m > 64
        _native.check(rc)
    lab = numpy.ascontiguousarray(clusters, dtype=numpy.float64).copy()
    order = [v for v in set(clusters) if v != -1 and v == v]            # :68-70, the interpreter's own visiting order (nan never matches :72)
    if not order:
        return clusters, cluster_id
    ycol = numpy.ascontiguousarray(data[:, 1])
    uniq, inv = numpy.unique(lab, return_inverse=True)
    rank_of = {v: r for r, v in enumerate(order)}
    seg_of = numpy.array([rank_of.get(v, -1) for v in uniq.tolist()], dtype=numpy.int32)
    seg = seg_of[inv.reshape(-1)]
    members = numpy.flatnonzero(seg >= 0)
    out, last_id = _y_segments(ctx, numpy.ascontiguousarray(ycol[members]), numpy.ascontiguousarray(seg[members]),
                               numpy.ascontiguousarray(lab[members]), len(order), epsilon, m, cluster_id)
    hi = numpy.array(order, dtype=numpy.float64)
    if not ((hi > cluster_id) & (hi <= last_id)).any():                  # no produced id is a visited value: the clusters were independent
        lab[members] = out
        clusters[:] = lab
        return clusters, last_id
    cid = cluster_id                                                     # literal replay, one visited cluster at a time (:69-122)
    for c in order:
        idx = numpy.flatnonzero(lab == c)
        if not len(idx):
            continue
        out, cid = _y_segments(ctx, numpy.ascontiguousarray(ycol[idx]), numpy.zeros(len(idx), dtype=numpy.int32),
                               numpy.ascontiguousarray(lab[idx]), 1, epsilon, m, cid)
        lab[idx] = out
    clusters[:] = lab
    return clusters, cid


def main(data, epsilon, m):
    return _run(data, epsilon, m, 0)[0]
