// _pycand: the candidate dictionaries of tiddit_cluster.main (tiddit_cluster.pyx:156-254), built with the CPython API.
//
// The return value of tiddit_cluster.main IS a nested Python dictionary — candidates[chrA][chrB][cluster id] = {26 keys, two of them
// dictionaries of eight lists} — so the last step of the clustering stage cannot leave the interpreter.  What can leave it is the
// bytecode: tiddit_amd/tiddit_cluster.py::_native_candidates builds one such dictionary per candidate from the flat member arrays of
// tdt_sigtab_regroup_result (15 µs each: two set() calls, twelve list slices, a 26-key literal; 0.13 s of the 3-Gb job's 2.0 s), this
// module does the same with direct API calls (PyDict_SetItem on interned keys, lists filled from the int32 columns without the
// intermediate .tolist() of the whole column).  Same keys in the same order, same value types (int, "True" / "False" strings, sets of
// str), the same objects shared where the Python shares them ("discordants" IS the set whose copy sits under "sample_discordants").
// tests/test_sigtab_cpu.py compares its output with the Python loop's, key order and value types included.
//
// Host-side glue, no device work; built by tiddit_amd/build.py with gcc (no hipcc needed), loaded by tiddit_cluster when present.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

enum {
    K_signal_type, K_samples, K_sample_discordants, K_sample_splits, K_sample_contigs, K_N_discordants, K_discordants, K_N_splits, K_splits,
    K_N_contigs, K_contigs, K_n_signals, K_posA, K_positions_A, K_start_A, K_end_A, K_posB, K_positions_B, K_start_B, K_end_B,
    P_contigs, P_splits, P_discordants, P_orientation_contigs, P_orientation_splits, P_orientation_discordants, P_start, P_end,
    F_startB, F_endB, F_startA, F_endA, K_COUNT
};
static const char *const key_text[K_COUNT] = {
    "signal_type", "samples", "sample_discordants", "sample_splits", "sample_contigs", "N_discordants", "discordants", "N_splits", "splits",
    "N_contigs", "contigs", "n_signals", "posA", "positions_A", "start_A", "end_A", "posB", "positions_B", "start_B", "end_B",
    "contigs", "splits", "discordants", "orientation_contigs", "orientation_splits", "orientation_discordants", "start", "end",
    "startB", "endB", "startA", "endA"};
static PyObject *keys[K_COUNT];
static PyObject *word[2];               // "False", "True"

// dict[key] = value, taking over the reference to value; -1 on failure (value released either way)
static int put(PyObject *d, int key, PyObject *value) {
    if (!value) return -1;
    const int rc = PyDict_SetItem(d, keys[key], value);
    Py_DECREF(value);
    return rc;
}

static PyObject *ints(const int32_t *p, Py_ssize_t lo, Py_ssize_t hi) {
    PyObject *l = PyList_New(hi - lo);
    if (!l) return NULL;
    for (Py_ssize_t i = lo; i < hi; i++) {
        PyObject *v = PyLong_FromLong(p[i]);
        if (!v) {
            Py_DECREF(l);
            return NULL;
        }
        PyList_SET_ITEM(l, i - lo, v);
    }
    return l;
}

static PyObject *words(const uint8_t *p, Py_ssize_t lo, Py_ssize_t hi) {
    PyObject *l = PyList_New(hi - lo);
    if (!l) return NULL;
    for (Py_ssize_t i = lo; i < hi; i++) {
        PyObject *v = word[p[i] ? 1 : 0];
        Py_INCREF(v);
        PyList_SET_ITEM(l, i - lo, v);
    }
    return l;
}

static PyObject *name_set(PyObject *const *frag, Py_ssize_t lo, Py_ssize_t hi) {
    PyObject *s = PySet_New(NULL);
    if (!s) return NULL;
    for (Py_ssize_t i = lo; i < hi; i++)
        if (PySet_Add(s, frag[i]) < 0) {
            Py_DECREF(s);
            return NULL;
        }
    return s;
}

// {sample: value} (takes over the reference to value)
static PyObject *by_sample(PyObject *sample, PyObject *value) {
    if (!value) return NULL;
    PyObject *d = PyDict_New();
    if (d && PyDict_SetItem(d, sample, value) < 0) Py_CLEAR(d);
    Py_DECREF(value);
    return d;
}

// positions_A / positions_B of one candidate: members [lo, mid) are its discordant pairs, [mid, hi) its split reads
static PyObject *positions(const int32_t *pos, const uint8_t *ori, const int32_t *start, const int32_t *end, Py_ssize_t lo, Py_ssize_t mid, Py_ssize_t hi) {
    PyObject *d = PyDict_New();
    if (!d) return NULL;
    if (put(d, P_contigs, PyList_New(0)) < 0 || put(d, P_splits, ints(pos, mid, hi)) < 0 || put(d, P_discordants, ints(pos, lo, mid)) < 0 ||
        put(d, P_orientation_contigs, PyList_New(0)) < 0 || put(d, P_orientation_splits, words(ori, mid, hi)) < 0 ||
        put(d, P_orientation_discordants, words(ori, lo, mid)) < 0 || put(d, P_start, ints(start, lo, hi)) < 0 || put(d, P_end, ints(end, lo, hi)) < 0)
        Py_CLEAR(d);
    return d;
}

// Counter(values).most_common(1)[0][0] with CPython's tie rule — the value inserted first among those with the largest count (:266-268; the
// product's tiddit_cluster._mode) — over a slice of a column (a candidate has a handful of members: counted pairwise)
static int32_t mode_of(const int32_t *p, Py_ssize_t lo, Py_ssize_t hi) {
    int32_t best = p[lo];
    Py_ssize_t best_n = 0;
    for (Py_ssize_t i = lo; i < hi; i++) {
        int seen = 0;
        for (Py_ssize_t j = lo; j < i && !seen; j++) seen = p[j] == p[i];
        if (seen) continue;
        Py_ssize_t c = 0;
        for (Py_ssize_t j = i; j < hi; j++) c += p[j] == p[i];
        if (c > best_n) best_n = c, best = p[i];
    }
    return best;
}

static int32_t min_of(const int32_t *p, Py_ssize_t lo, Py_ssize_t hi) {
    int32_t v = p[lo];
    for (Py_ssize_t i = lo + 1; i < hi; i++) v = p[i] < v ? p[i] : v;
    return v;
}

static int32_t max_of(const int32_t *p, Py_ssize_t lo, Py_ssize_t hi) {
    int32_t v = p[lo];
    for (Py_ssize_t i = lo + 1; i < hi; i++) v = p[i] > v ? p[i] : v;
    return v;
}

// the breakpoint of one side from the candidate's discordant pairs (tiddit_cluster.pyx:277-330; tiddit_cluster._breakpoints_from_discordants):
// orientation-consistent clusters on BOTH sides take an extreme position, the others the mode.  -> 1 if this side is consistent
static int side_consistent(const uint8_t *ori, Py_ssize_t lo, Py_ssize_t mid, Py_ssize_t *rev, Py_ssize_t *fwd) {
    Py_ssize_t r = 0;
    for (Py_ssize_t i = lo; i < mid; i++) r += ori[i] != 0;
    *rev = r;
    *fwd = (mid - lo) - r;
    return r >= 5 * *fwd || r * 5 <= *fwd;
}

// build(slots, cand, names, startA, endA, startB, endB, posA, posB, oriA, oriB, sample[, is_mp, min_reads]) -> number of candidates
//   slots: list of dict, one per bucket (candidates[chrA][chrB]); cand: int32[n, 4] rows (bucket, cluster id, discordant members, split
//   members), the members of candidate k following those of candidate k - 1 in the member arrays; names: the members' fragment names joined
//   by "\n"; the six int32 columns and the two uint8 orientation columns of tdt_sigtab_regroup_result
static PyObject *build(PyObject *self, PyObject *args) {
    PyObject *slots, *sample;
    Py_buffer cand, names, col[6], ori[2];
    int is_mp = -1;                     // given: the counts, breakpoints and regions of tiddit_cluster._finish_candidates (:256-336) are filled in too
    long min_reads = 0;
    if (!PyArg_ParseTuple(args, "O!y*y*y*y*y*y*y*y*y*y*U|pl", &PyList_Type, &slots, &cand, &names, &col[0], &col[1], &col[2], &col[3], &col[4], &col[5],
                          &ori[0], &ori[1], &sample, &is_mp, &min_reads))
        return NULL;
    PyObject *result = NULL;
    PyObject **frag = NULL;
    Py_ssize_t n_frag = 0;
    const Py_ssize_t n = cand.len / 16, m = col[0].len / 4;
    const int32_t *C = (const int32_t *)cand.buf;
    const int32_t *sA = (const int32_t *)col[0].buf, *eA = (const int32_t *)col[1].buf, *sB = (const int32_t *)col[2].buf, *eB = (const int32_t *)col[3].buf,
                  *pA = (const int32_t *)col[4].buf, *pB = (const int32_t *)col[5].buf;
    const uint8_t *oA = (const uint8_t *)ori[0].buf, *oB = (const uint8_t *)ori[1].buf;
    int bad = cand.len % 16 != 0 || ori[0].len != m || ori[1].len != m;
    for (int k = 0; k < 6; k++) bad = bad || col[k].len != 4 * m;
    if (bad) {
        PyErr_SetString(PyExc_ValueError, "_pycand.build: column lengths disagree");
        goto done;
    }
    // the members' names as str objects, once (the Python path: names.decode().split("\n"))
    frag = (PyObject **)PyMem_Calloc((size_t)(m ? m : 1), sizeof(PyObject *));
    if (!frag) {
        PyErr_NoMemory();
        goto done;
    }
    {
        const char *p = (const char *)names.buf, *end = p + names.len;
        while (n_frag < m) {
            const char *q = memchr(p, '\n', (size_t)(end - p));
            if (!q) q = end;
            if (!(frag[n_frag] = PyUnicode_DecodeUTF8(p, q - p, NULL))) goto done;
            n_frag++;
            if (q == end) break;
            p = q + 1;
        }
        if (n_frag != m) {
            PyErr_SetString(PyExc_ValueError, "_pycand.build: fewer names than members");
            goto done;
        }
    }
    {
        const Py_ssize_t n_slots = PyList_GET_SIZE(slots);
        Py_ssize_t lo = 0;
        for (Py_ssize_t k = 0; k < n; k++) {
            const int32_t bkt = C[4 * k], cid = C[4 * k + 1], nd = C[4 * k + 2], ns = C[4 * k + 3];
            const Py_ssize_t mid = lo + nd, hi = mid + ns;
            if (bkt < 0 || bkt >= n_slots || nd < 0 || ns < 0 || hi > m || !PyDict_Check(PyList_GET_ITEM(slots, bkt))) {
                PyErr_SetString(PyExc_ValueError, "_pycand.build: candidate row out of range");
                goto done;
            }
            PyObject *d = PyDict_New(), *dn = name_set(frag, lo, mid), *sn = name_set(frag, mid, hi);
            int ok = d && dn && sn;
            if (ok) {
                PyObject *samples = PySet_New(NULL);
                if (samples && PySet_Add(samples, sample) < 0) Py_CLEAR(samples);
                Py_INCREF(dn);
                Py_INCREF(sn);
                ok = put(d, K_signal_type, PyDict_New()) == 0 && put(d, K_samples, samples) == 0 &&
                     put(d, K_sample_discordants, by_sample(sample, PySet_New(dn))) == 0 && put(d, K_sample_splits, by_sample(sample, PySet_New(sn))) == 0 &&
                     put(d, K_sample_contigs, by_sample(sample, PySet_New(NULL))) == 0 && put(d, K_N_discordants, PyLong_FromLong(0)) == 0;
                ok = (put(d, K_discordants, dn) == 0) && ok;          // (the references taken above are given up whatever `ok` says)
                ok = ok && put(d, K_N_splits, PyLong_FromLong(0)) == 0;
                ok = (put(d, K_splits, sn) == 0) && ok;
                ok = ok && put(d, K_N_contigs, PyLong_FromLong(0)) == 0 && put(d, K_contigs, PySet_New(NULL)) == 0 &&
                     put(d, K_n_signals, PyLong_FromLong(0)) == 0 && put(d, K_posA, PyLong_FromLong(0)) == 0 &&
                     put(d, K_positions_A, positions(pA, oA, sA, eA, lo, mid, hi)) == 0 && put(d, K_start_A, PyLong_FromLong(0)) == 0 &&
                     put(d, K_end_A, PyLong_FromLong(0)) == 0 && put(d, K_posB, PyLong_FromLong(0)) == 0 &&
                     put(d, K_positions_B, positions(pB, oB, sB, eB, lo, mid, hi)) == 0 && put(d, K_start_B, PyLong_FromLong(0)) == 0 &&
                     put(d, K_end_B, PyLong_FromLong(0)) == 0;
            }
            if (ok && is_mp >= 0) {
                // N_* count distinct fragment names (the sets), not members; split reads decide the breakpoint whenever there are any
                // (:266-268 and :272-274 pick the same mode; a native candidate has no contigs), else the discordant pairs do
                const Py_ssize_t Nd = PySet_GET_SIZE(dn), Ns = PySet_GET_SIZE(sn);
                if (hi == lo) {
                    PyErr_SetString(PyExc_ValueError, "_pycand.build: a candidate without members");
                    ok = 0;
                } else {
                    int32_t a, b;
                    if (Ns) {
                        a = mode_of(pA, mid, hi), b = mode_of(pB, mid, hi);
                    } else {
                        Py_ssize_t revA, fwdA, revB, fwdB;
                        const int cA = side_consistent(oA, lo, mid, &revA, &fwdA), cB = side_consistent(oB, lo, mid, &revB, &fwdB);
                        if (!(cA && cB)) {
                            a = mode_of(pA, lo, mid), b = mode_of(pB, lo, mid);
                        } else {
                            const int a_rev = revA > fwdA, b_rev = revB > fwdB;
                            a = (is_mp ? a_rev : !a_rev) ? max_of(pA, lo, mid) : min_of(pA, lo, mid);
                            b = (is_mp ? b_rev : !b_rev) ? max_of(pB, lo, mid) : min_of(pB, lo, mid);
                        }
                    }
                    (void)min_reads;
                    ok = put(d, K_N_discordants, PyLong_FromSsize_t(Nd)) == 0 && put(d, K_N_splits, PyLong_FromSsize_t(Ns)) == 0 &&
                         put(d, K_N_contigs, PyLong_FromLong(0)) == 0 && put(d, K_posA, PyLong_FromLong(a)) == 0 && put(d, K_posB, PyLong_FromLong(b)) == 0 &&
                         put(d, F_startB, PyLong_FromLong(min_of(sB, lo, hi))) == 0 && put(d, F_endB, PyLong_FromLong(max_of(eB, lo, hi))) == 0 &&
                         put(d, F_startA, PyLong_FromLong(min_of(sA, lo, hi))) == 0 && put(d, F_endA, PyLong_FromLong(max_of(eA, lo, hi))) == 0;
                }
            }
            Py_XDECREF(dn);
            Py_XDECREF(sn);
            if (ok) {
                PyObject *key = PyLong_FromLong(cid);
                ok = key && PyDict_SetItem(PyList_GET_ITEM(slots, bkt), key, d) == 0;
                Py_XDECREF(key);
            }
            Py_XDECREF(d);
            if (!ok) {
                if (!PyErr_Occurred()) PyErr_NoMemory();
                goto done;
            }
            lo = hi;
        }
    }
    result = PyLong_FromSsize_t(n);
done:
    if (frag) {
        for (Py_ssize_t i = 0; i < n_frag; i++) Py_XDECREF(frag[i]);
        PyMem_Free(frag);
    }
    PyBuffer_Release(&cand);
    PyBuffer_Release(&names);
    for (int k = 0; k < 6; k++) PyBuffer_Release(&col[k]);
    PyBuffer_Release(&ori[0]);
    PyBuffer_Release(&ori[1]);
    return result;
}

static PyMethodDef methods[] = {
    {"build", build, METH_VARARGS,
     "build(slots, cand, names, startA, endA, startB, endB, posA, posB, oriA, oriB, sample[, is_mp, min_reads]): slots[bucket][cluster id] = the candidate "
     "dictionary of tiddit_cluster._new_candidate with its members filled in, for every row of cand; with is_mp given also what "
     "tiddit_cluster._finish_candidates adds (counts, breakpoints, regions)"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pycand", "candidate dictionaries of tiddit_cluster.main, built with the CPython API", -1, methods};

PyMODINIT_FUNC PyInit__pycand(void) {
    for (int k = 0; k < K_COUNT; k++)
        if (!(keys[k] = PyUnicode_InternFromString(key_text[k]))) return NULL;
    if (!(word[0] = PyUnicode_InternFromString("False")) || !(word[1] = PyUnicode_InternFromString("True"))) return NULL;
    return PyModule_Create(&module);
}
