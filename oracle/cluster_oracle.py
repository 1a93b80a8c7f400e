"""Per-signal restatement of tiddit_cluster.main (tiddit_cluster.pyx:39-338).  TEST INFRASTRUCTURE ONLY.

Pinned: tests/test_oracle.py runs it on the inputs of tests/golden/cluster.json and tests/golden/sv_e2e.json, whose
expected candidates were produced by the compiled reference module (tests/golden/make_golden.py).  Clustering goes through
the C oracle (oracle.dbscan_main, itself pinned to DBSCAN.py), the rest follows the reference's loops signal by signal:
file parsing (:46-137), per-bucket stable sort + DBSCAN.main + re-sort by signal index (:140-160), regrouping (:161-255) and
the breakpoint choice (:258-336).
"""
from collections import Counter

import numpy

import oracle

_SIDES = ("contigs", "splits", "discordants", "orientation_contigs", "orientation_splits", "orientation_discordants", "start", "end")


def find_discordant_pos(fragment, is_mp):          # :7-37, the eight orientation cases written out
    a_rev, b_rev = fragment[5], fragment[8]
    if is_mp:
        if a_rev == "False" and b_rev == "True":
            return fragment[3], fragment[7]
        if a_rev == "False" and b_rev == "False":
            return fragment[3], fragment[6]
        if a_rev == "True" and b_rev == "True":
            return fragment[4], fragment[7]
        return fragment[4], fragment[6]
    if a_rev == "False" and b_rev == "True":
        return fragment[4], fragment[6]
    if a_rev == "False" and b_rev == "False":
        return fragment[4], fragment[7]
    if a_rev == "True" and b_rev == "True":
        return fragment[3], fragment[6]
    return fragment[3], fragment[7]


def _fresh():                                      # :171-213, same key order
    c = {"signal_type": {}, "samples": set(), "sample_discordants": {}, "sample_splits": {}, "sample_contigs": {},
         "N_discordants": 0, "discordants": set(), "N_splits": 0, "splits": set(), "N_contigs": 0, "contigs": set(),
         "n_signals": 0, "posA": 0}
    c["positions_A"] = {k: [] for k in _SIDES}
    c["start_A"] = 0
    c["end_A"] = 0
    c["posB"] = 0
    c["positions_B"] = {k: [] for k in _SIDES}
    c["start_B"] = 0
    c["end_B"] = 0
    return c


def main(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig, skip_assembly, min_reads):
    discordants, positions = {}, {}
    i = 0
    for sample in samples:
        sources = [("D", "{}_tiddit/discordants_{}.tab"), ("S", "{}_tiddit/splits_{}.tab")]
        if not skip_assembly:
            sources.append(("A", "{}_tiddit/contigs_{}.tab"))
        for kind, pattern in sources:
            for line in open(pattern.format(prefix, sample)):
                content = line.rstrip().split("\t")
                chrA, chrB = content[1], content[2]
                if contig_length[chrA] < min_contig or contig_length[chrB] < min_contig:
                    continue
                positions.setdefault(chrA, {}).setdefault(chrB, [])
                discordants.setdefault(chrA, {}).setdefault(chrB, [])
                if kind == "D":
                    posA, posB = find_discordant_pos(content, is_mp)
                    if int(posA) > contig_length[chrA]:
                        posA = contig_length[chrA]
                        if int(posB) > contig_length[chrB]:
                            posA = contig_length[chrB]                      # :67-70, posA again
                    row = [content[0], sample, "D", posA, content[5], posB, content[8], i,
                           int(content[3]), int(content[4]), int(content[6]), int(content[7])]
                else:
                    posA, posB = content[3], content[5]
                    if int(posA) > contig_length[chrA]:
                        posA = contig_length[chrA]
                    if int(posB) > contig_length[chrB]:
                        posB = contig_length[chrB]
                    row = [content[0], sample, kind, posA, content[4], posB, content[6], i,
                           int(content[7]), int(content[8]), int(content[9]), int(content[10])]
                discordants[chrA][chrB].append(row)
                positions[chrA][chrB].append([int(posA), int(posB), i])
                i += 1

    candidates = {}
    for chrA in chromosomes:
        if chrA not in positions:
            continue
        candidates.setdefault(chrA, {})
        for chrB in chromosomes:
            if chrB not in positions[chrA]:
                continue
            bucket = candidates[chrA].setdefault(chrB, {})
            pts = numpy.array(sorted(positions[chrA][chrB], key=lambda l: l[0]), dtype=numpy.int64)       # :152
            clusters = oracle.dbscan_main(pts, epsilon, m)                                               # :154
            cluster_pos = sorted(([int(p[0]), int(p[1]), int(p[2]), clusters[k]] for k, p in enumerate(pts)), key=lambda l: l[2])
            sig = discordants[chrA][chrB]
            n_ctg_clusters = 0
            for k in range(len(cluster_pos)):
                candidate = int(cluster_pos[k][-1])
                s = sig[k]
                lone = chrA == chrB and s[2] == "A" and (int(s[5]) - int(s[3])) < max_ins_len * 2
                if candidate == -1 and not lone:
                    continue
                elif candidate == -1 and s[2] == "A":
                    candidate = len(cluster_pos) + n_ctg_clusters
                    n_ctg_clusters += 1
                if candidate not in bucket:
                    bucket[candidate] = _fresh()
                c = bucket[candidate]
                if s[1] not in c["samples"]:
                    c["sample_discordants"][s[1]] = set()
                    c["sample_splits"][s[1]] = set()
                    c["sample_contigs"][s[1]] = set()
                c["samples"].add(s[1])
                c["positions_A"]["start"].append(s[8])
                c["positions_A"]["end"].append(s[9])
                c["positions_B"]["start"].append(s[10])
                c["positions_B"]["end"].append(s[11])
                name = {"D": "discordants", "S": "splits"}.get(s[2], "contigs")
                c[name].add(s[0])
                c["positions_A"][name].append(int(s[3]))
                c["positions_A"]["orientation_" + name].append(s[4])
                c["positions_B"][name].append(int(s[5]))
                c["positions_B"]["orientation_" + name].append(s[6])
                c["sample_" + name][s[1]].add(s[0])

    for chrA in candidates:
        for chrB in candidates[chrA]:
            for c in candidates[chrA][chrB].values():
                c["N_discordants"], c["N_splits"], c["N_contigs"] = len(c["discordants"]), len(c["splits"]), len(c["contigs"])
                A, B = c["positions_A"], c["positions_B"]
                top = lambda v: Counter(v).most_common(1)[0][0]
                if c["N_splits"] and min_reads <= c["N_splits"]:
                    c["posA"], c["posB"] = top(A["splits"]), top(B["splits"])
                elif c["N_contigs"]:
                    c["posA"], c["posB"] = top(A["contigs"]), top(B["contigs"])
                elif c["N_splits"]:
                    c["posA"], c["posB"] = top(A["splits"]), top(B["splits"])
                else:
                    reverse_A, forward_A = A["orientation_discordants"].count("True"), A["orientation_discordants"].count("False")
                    reverse_B, forward_B = B["orientation_discordants"].count("True"), B["orientation_discordants"].count("False")
                    if (reverse_A >= 5 * forward_A or reverse_A * 5 <= forward_A) and (reverse_B >= 5 * forward_B or reverse_B * 5 <= forward_B):
                        A_reverse, B_reverse = reverse_A > forward_A, reverse_B > forward_B
                        if is_mp:            # :292-307
                            table = {(True, False): (max, min), (False, True): (min, max), (True, True): (max, max), (False, False): (min, min)}
                        else:                # :309-324
                            table = {(False, True): (max, min), (True, False): (min, max), (False, False): (max, max), (True, True): (min, min)}
                        fa, fb = table[(A_reverse, B_reverse)]
                        c["posA"], c["posB"] = fa(A["discordants"]), fb(B["discordants"])
                    else:
                        c["posA"], c["posB"] = top(A["discordants"]), top(B["discordants"])
                c["startB"], c["endB"] = min(B["start"]), max(B["end"])
                c["startA"], c["endA"] = min(A["start"]), max(A["end"])
    return candidates


def summary(candidates):
    """flat, order-preserving view used for comparisons and fixtures: [chrA, chrB, id, posA, posB, N_d, N_s, N_c, startA, endA, startB, endB]"""
    rows = []
    for chrA in candidates:
        for chrB in candidates[chrA]:
            for cid, c in candidates[chrA][chrB].items():
                rows.append([chrA, chrB, int(cid), int(c["posA"]), int(c["posB"]), c["N_discordants"], c["N_splits"], c["N_contigs"],
                             int(c["startA"]), int(c["endA"]), int(c["startB"]), int(c["endB"])])
    return rows


def canonical(candidates):
    """deterministic text of the WHOLE nested dictionary (sets sorted, dict order kept) for sha256 comparisons"""
    def enc(v):
        if isinstance(v, dict):
            return "{" + ",".join("%s:%s" % (enc(k), enc(x)) for k, x in v.items()) + "}"
        if isinstance(v, (set, frozenset)):
            return "<" + ",".join(sorted(enc(x) for x in v)) + ">"
        if isinstance(v, (list, tuple)):
            return "[" + ",".join(enc(x) for x in v) + "]"
        if isinstance(v, (numpy.integer,)):
            return str(int(v))
        if isinstance(v, (float, numpy.floating)):
            return repr(float(v))
        return repr(v)
    return enc(candidates)
