// Signal tables on the host (no device code, no Python): everything tiddit_signal.main does with the rows its workers return
// (tiddit_signal.pyx:246-326) and everything tiddit_cluster.main does before and after DBSCAN.main with those rows
// (tiddit_cluster.pyx:47-137 the signal table, :156-254 the regrouping into candidates).
//
// The reference passes every discordant / split read through Python lists four times: worker() appends a row, main() merges the
// rows of all contigs into data[chrA][chrB][fragment], writes them as text, and tiddit_cluster.main() parses the text back.  Here the
// selected reads of a device batch (tdt_signal_scan_result: 28-byte field records + the raw BAM records) go ONCE into an
// append-only row log (32 / 96 bytes per row, names in one arena), are merged into per-(chrA,chrB) hash tables keyed by the
// fragment name in first-seen order WHILE the file is still being scanned (the caller's row thread runs beside the device ingest of
// the next batch, without the GIL), and the same tables then produce
//   * the bytes of discordants_{sample}.tab / splits_{sample}.tab (formatted per contig pair on the host thread pool),
//   * the (posA, posB) columns of every (chrA,chrB) bucket with the clip quirk of tiddit_cluster.pyx:67-70, written straight into
//     the caller's (pinned) int32 buffers for tdt_cluster_columns,
//   * and, given the labels, the members of every candidate in the reference's order (candidate = first appearance of its cluster
//     id in signal order) as flat arrays, so that Python only touches the ~10^4 candidates, not the ~10^6 rows.
// The N-rank job routes rows to the owner rank of their chrA (tdt_sigtab_export / _import: the log rows as one blob), so that every
// owner merges, formats, clusters and regroups its own pairs.
//
// Order is the contract.  main() merges results in contig order (header order of the contigs >= min_contig) and, inside a contig, in
// file order.  Rows arrive here in file order; as long as their contig ids never decrease that IS the reference's order and the merge
// is incremental.  The first row that breaks it (an unsorted file) switches the incremental merge of its kind off; finalize then
// merges the whole log in (contig, arrival) order — same result, just not overlapped.
#include "tdt_common.h"

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include <cerrno>
#include <unistd.h>

namespace {

const uint32_t NONE = 0xffffffffu;

struct DRow {              // one discordant read as worker() reports it (tiddit_signal.pyx:214-221): [chrA, chrB, name, start+1, end+1, is_reverse, read contig]
    int32_t tid, mate;     // chrA / chrB follow from the string order of the two contig names
    int32_t start, end;    // reference_start + 1, reference_end + 1
    uint64_t name_off;
    uint16_t name_len;
    uint8_t rev, pad;
    uint32_t pad2;
};
static_assert(sizeof(DRow) == 32, "DRow layout");

struct SRow {              // one split read as SA_analysis returns it (:138-142): [chrA, chrB, name, split_pos, is_reverse, sa_split, sa_minus, startA, endA, startB, endB]
    int32_t tid, a, b;     // contig ids of chrA / chrB; -1 = a name that is not in the header (kept in the arena: other_off / other_len)
    uint16_t name_len;
    uint8_t rev, sa_minus;
    uint64_t name_off;
    uint64_t other_off;
    uint32_t other_len, pad;
    int64_t f[6];          // split_pos, sa_split, startA, endA, startB, endB
};
static_assert(sizeof(SRow) == 88, "SRow layout");

static inline uint32_t hash_bytes(const char *p, size_t n) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (n * 0xff51afd7ed558ccdull);
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0xd6e8feb86659fd93ull;
        h ^= h >> 32;
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w) * 0xd6e8feb86659fd93ull;
    h ^= h >> 32;
    h *= 0xd6e8feb86659fd93ull;
    return (uint32_t)(h >> 32);
}

struct Frag {
    uint32_t hash;
    uint32_t row0;         // first row of the fragment (its name lives there)
    uint32_t row1;         // discordants: the second read; splits: the LAST row of the chain
    uint32_t count;
};

struct FragTable {         // fragment name -> Frag, fragments in first-seen order (a Python dict's order)
    std::vector<Frag> frags;
    std::vector<uint32_t> slot;      // frag index + 1, open addressing
    uint32_t mask = 0;

    void clear() {
        frags.clear();
        slot.clear();
        mask = 0;
    }
    void grow() {
        const size_t cap = slot.empty() ? 16 : slot.size() * 2;
        slot.assign(cap, 0u);
        mask = (uint32_t)(cap - 1);
        for (uint32_t i = 0; i < frags.size(); i++) {
            uint32_t s = frags[i].hash & mask;
            while (slot[s]) s = (s + 1) & mask;
            slot[s] = i + 1;
        }
    }
    // -> index of the fragment named [p, p+len); *fresh when it was appended by this call (its row0 is then still to be set)
    template <class NameOf>
    uint32_t find_or_add(const char *p, uint16_t len, uint32_t h, NameOf name_of, bool *fresh) {
        if (frags.size() * 2 >= slot.size()) grow();
        uint32_t s = h & mask;
        while (slot[s]) {
            const Frag &f = frags[slot[s] - 1];
            if (f.hash == h) {
                uint16_t l2;
                const char *q = name_of(f.row0, &l2);
                if (l2 == len && memcmp(q, p, len) == 0) {
                    *fresh = false;
                    return slot[s] - 1;
                }
            }
            s = (s + 1) & mask;
        }
        frags.push_back(Frag{h, NONE, NONE, 0});
        slot[s] = (uint32_t)frags.size();
        *fresh = true;
        return (uint32_t)frags.size() - 1;
    }
};

struct PairTab {
    int32_t a, b;
    FragTable d, s;
};

struct Segment {
    int32_t a, b;
    int64_t off, len, rows;
};

struct Sig {               // one row of tiddit_cluster.main's signal table (:72, :102)
    int32_t posA, posB, sA, eA, sB, eB;
    uint32_t row;          // the dlog / slog row that carries the fragment's name
    uint8_t kind, oriA, oriB, pad;      // kind 0 = "D", 1 = "S"
};

static inline size_t fmt_i64(long long v, char *p) {
    char t[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do {
        t[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    size_t k = 0;
    if (v < 0) p[k++] = '-';
    for (int i = 0; i < n; i++) p[k++] = t[n - 1 - i];
    return k;
}

static inline size_t fmt_bool(bool v, char *p) {
    if (v) {
        memcpy(p, "True", 4);
        return 4;
    }
    memcpy(p, "False", 5);
    return 5;
}

}  // namespace

struct tdt_sigtab {
    int n = 0;
    int64_t min_contig = 0;
    std::vector<std::string> names;
    std::vector<int64_t> length;
    std::vector<uint8_t> kept;               // contig >= min_contig: main()'s `chromosomes`
    std::vector<int32_t> name_rank;          // position of the contig's name in string order (tiddit_signal.pyx:214: `if mate < chrom`)
    std::unordered_map<std::string, int> id_of;
    std::vector<char> arena;
    std::vector<DRow> dlog;
    std::vector<SRow> slog;
    std::vector<uint32_t> snext;             // splits: next row of the same fragment (the `+=` of :282 as a chain)
    size_t d_merged = 0, s_merged = 0;
    int32_t d_last = -1, s_last = -1;
    bool d_ordered = true, s_ordered = true, dirty = false, finalized = false;
    // finalize() / format() are reached from two threads of one job (the writer thread's pwrite and the clustering's cluster_table):
    // they are idempotent once done, and the first caller does the work under this lock
    std::recursive_mutex mu;
    std::unordered_map<uint64_t, uint32_t> pair_of;
    std::vector<std::unique_ptr<PairTab>> pairs;
    PairTab *last_pair = nullptr;
    std::vector<std::string> clips;          // per contig: the clip FASTA entries of its reads in file order
    // after finalize
    std::vector<uint32_t> order;             // pairs in (chrA, chrB) header order = the nesting of main()'s dictionaries
    std::string text[2];
    std::vector<Segment> seg[2];
    std::vector<int64_t> blk_off[2], blk_len[2];   // per chrA: its rows' bytes inside text[k] (the pairs of one chrA are contiguous)
    bool formatted = false;
    // clustering
    std::vector<Sig> sigs;
    std::vector<int64_t> bucket_off;
    std::vector<int32_t> bucket_a, bucket_b;
    // regroup result
    std::vector<int32_t> cand;               // 4 per candidate: bucket, cluster id, discordant members, split members
    std::vector<uint32_t> mem_all, mem_kind; // signal indices, candidate-major: in signal order / discordants then splits

    const char *dname(uint32_t r, uint16_t *len) const {
        *len = dlog[r].name_len;
        return arena.data() + dlog[r].name_off;
    }
    const char *sname(uint32_t r, uint16_t *len) const {
        *len = slog[r].name_len;
        return arena.data() + slog[r].name_off;
    }
    PairTab &pair(int a, int b) {
        if (last_pair && last_pair->a == a && last_pair->b == b) return *last_pair;
        const uint64_t key = (uint64_t)(uint32_t)a * (uint64_t)n + (uint32_t)b;
        auto it = pair_of.find(key);
        if (it == pair_of.end()) {
            pairs.emplace_back(new PairTab());
            pairs.back()->a = a;
            pairs.back()->b = b;
            it = pair_of.emplace(key, (uint32_t)pairs.size() - 1).first;
        }
        last_pair = pairs[it->second].get();
        return *last_pair;
    }
    uint64_t put(const void *p, size_t len) {
        const uint64_t off = arena.size();
        arena.insert(arena.end(), (const char *)p, (const char *)p + len);
        return off;
    }

    // tiddit_signal.pyx:266-272
    int merge_d(uint32_t r) {
        const DRow &x = dlog[r];
        if (x.tid < 0 || x.tid >= n || !kept[x.tid]) return TDT_OK;         // (worker only runs on the contigs >= min_contig, :250-259)
        if (x.mate < 0 || x.mate >= n) {
            tdt_set_error("signal tables: mate contig id %d outside the header's %d contigs", x.mate, n);
            return TDT_E_RANGE;
        }
        const bool mate_first = name_rank[x.mate] < name_rank[x.tid];
        const int a = mate_first ? x.mate : x.tid, b = mate_first ? x.tid : x.mate;
        if (!kept[a]) return TDT_OK;                                         // `if not signal[0] in data: continue`
        PairTab &p = pair(a, b);
        bool fresh;
        const char *nm = arena.data() + x.name_off;
        const uint32_t h = hash_bytes(nm, x.name_len);
        const uint32_t fi = p.d.find_or_add(nm, x.name_len, h, [this](uint32_t row, uint16_t *l) { return dname(row, l); }, &fresh);
        Frag &f = p.d.frags[fi];
        if (fresh) f.row0 = r;
        else if (f.count == 1) f.row1 = r;
        f.count++;
        return TDT_OK;
    }
    // :274-282
    int merge_s(uint32_t r) {
        const SRow &x = slog[r];
        if (x.tid < 0 || x.tid >= n || !kept[x.tid]) return TDT_OK;
        if (x.a < 0 || !kept[x.a]) return TDT_OK;                            // `if not signal[0] in splits: continue`
        if (x.b < 0) {                                                       // splits[chrA][chrB] with a chrB the header does not have: KeyError
            tdt_set_error("%.*s", (int)x.other_len, arena.data() + x.other_off);
            return TDT_E_KEY;
        }
        PairTab &p = pair(x.a, x.b);
        bool fresh;
        const char *nm = arena.data() + x.name_off;
        const uint32_t h = hash_bytes(nm, x.name_len);
        const uint32_t fi = p.s.find_or_add(nm, x.name_len, h, [this](uint32_t row, uint16_t *l) { return sname(row, l); }, &fresh);
        Frag &f = p.s.frags[fi];
        if (fresh) f.row0 = r;
        else snext[f.row1] = r;
        f.row1 = r;
        f.count++;
        return TDT_OK;
    }
    int merge_new() {
        std::lock_guard<std::recursive_mutex> hold(mu);
        formatted = false;
        finalized = false;
        dirty = true;
        if (d_ordered)
            for (; d_merged < dlog.size(); d_merged++) {
                const int rc = merge_d((uint32_t)d_merged);
                if (rc) return rc;
            }
        if (s_ordered)
            for (; s_merged < slog.size(); s_merged++) {
                const int rc = merge_s((uint32_t)s_merged);
                if (rc) return rc;
            }
        return TDT_OK;
    }
    void note_d(int32_t tid) {
        if (tid < d_last) d_ordered = false;
        d_last = tid;
    }
    void note_s(int32_t tid) {
        if (tid < s_last) s_ordered = false;
        s_last = tid;
    }
    int finalize() {
        std::lock_guard<std::recursive_mutex> hold(mu);
        if (!dirty && finalized) return TDT_OK;            // (a table without pairs is final too: `order` stays empty)
        if (dlog.size() > 0xfffffff0ull || slog.size() > 0xfffffff0ull) {
            tdt_set_error("signal tables: more than 2^32 rows");
            return TDT_E_UNSUPPORTED;
        }
        if (!d_ordered && d_merged != dlog.size()) {                         // an unsorted file: the whole log again, contig by contig (:262-272)
            for (auto &p : pairs) p->d.clear();
            std::vector<uint32_t> idx(dlog.size());
            for (uint32_t i = 0; i < idx.size(); i++) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [this](uint32_t x, uint32_t y) { return dlog[x].tid < dlog[y].tid; });
            for (uint32_t r : idx) {
                const int rc = merge_d(r);
                if (rc) return rc;
            }
            d_merged = dlog.size();
        }
        if (!s_ordered && s_merged != slog.size()) {
            for (auto &p : pairs) p->s.clear();
            std::fill(snext.begin(), snext.end(), NONE);
            std::vector<uint32_t> idx(slog.size());
            for (uint32_t i = 0; i < idx.size(); i++) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [this](uint32_t x, uint32_t y) { return slog[x].tid < slog[y].tid; });
            for (uint32_t r : idx) {
                const int rc = merge_s(r);
                if (rc) return rc;
            }
            s_merged = slog.size();
        }
        order.resize(pairs.size());
        for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [this](uint32_t x, uint32_t y) {
            return pairs[x]->a != pairs[y]->a ? pairs[x]->a < pairs[y]->a : pairs[x]->b < pairs[y]->b;
        });
        dirty = false;
        finalized = true;
        formatted = false;
        return TDT_OK;
    }

    // the two reads of a written discordant fragment in the order of :300-318
    bool d_written(const PairTab &p, const Frag &f, const DRow **first, const DRow **second) const {
        if (f.count < 2) return false;
        const DRow *x = &dlog[f.row0], *y = &dlog[f.row1];
        // chrA == chrB: QUIRK (:307) — the comparison is between the two reads' contig names, which are equal: never swapped
        if (p.a != p.b && x->tid != p.a) std::swap(x, y);
        *first = x;
        *second = y;
        return true;
    }
    void format_pair(const PairTab &p, int kind, std::string &out, int64_t *rows) const {
        char line[256];
        const std::string &A = names[p.a], &B = names[p.b];
        *rows = 0;
        if (kind == 0) {
            for (const Frag &f : p.d.frags) {
                const DRow *x, *y;
                if (!d_written(p, f, &x, &y)) continue;
                const DRow &o = dlog[f.row0];
                out.append(arena.data() + o.name_off, o.name_len);
                out.push_back('\t');
                out += A;
                out.push_back('\t');
                out += B;
                char *q = line;
                for (const DRow *r : {x, y}) {
                    *q++ = '\t';
                    q += fmt_i64(r->start, q);
                    *q++ = '\t';
                    q += fmt_i64(r->end, q);
                    *q++ = '\t';
                    q += fmt_bool(r->rev, q);
                }
                *q++ = '\n';
                out.append(line, (size_t)(q - line));
                ++*rows;
            }
        } else {
            for (const Frag &f : p.s.frags) {
                const SRow &o = slog[f.row0];
                out.append(arena.data() + o.name_off, o.name_len);
                out.push_back('\t');
                out += A;
                out.push_back('\t');
                out += B;
                for (uint32_t r = f.row0; r != NONE; r = snext[r]) {
                    const SRow &s = slog[r];
                    char *q = line;
                    *q++ = '\t';
                    q += fmt_i64(s.f[0], q);
                    *q++ = '\t';
                    q += fmt_bool(s.rev, q);
                    *q++ = '\t';
                    q += fmt_i64(s.f[1], q);
                    *q++ = '\t';
                    q += fmt_bool(s.sa_minus, q);
                    for (int k = 2; k < 6; k++) {
                        *q++ = '\t';
                        q += fmt_i64(s.f[k], q);
                    }
                    out.append(line, (size_t)(q - line));
                }
                out.push_back('\n');
                ++*rows;
            }
        }
    }
    int format() {
        std::lock_guard<std::recursive_mutex> hold(mu);
        int rc = finalize();
        if (rc) return rc;
        if (formatted) return TDT_OK;
        const size_t np = order.size();
        std::vector<std::string> part[2];
        std::vector<int64_t> rows[2];
        for (int k = 0; k < 2; k++) {
            part[k].resize(np);
            rows[k].assign(np, 0);
        }
        // the big pairs first (chr1-chr1 holds 8 % of a human genome's rows): dynamic assignment over the host threads
        std::vector<uint32_t> by_size(np);
        for (uint32_t i = 0; i < np; i++) by_size[i] = i;
        std::sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) {
            const size_t sx = pairs[order[x]]->d.frags.size() + pairs[order[x]]->s.frags.size(), sy = pairs[order[y]]->d.frags.size() + pairs[order[y]]->s.frags.size();
            return sx != sy ? sx > sy : x < y;
        });
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= np) break;
                const uint32_t i = by_size[j];
                const PairTab &p = *pairs[order[i]];
                part[0][i].reserve(p.d.frags.size() * 64);
                format_pair(p, 0, part[0][i], &rows[0][i]);
                part[1][i].reserve(p.s.frags.size() * 96);
                format_pair(p, 1, part[1][i], &rows[1][i]);
            }
        };
        size_t total_frags = 0;
        for (auto &p : pairs) total_frags += p->d.frags.size() + p->s.frags.size();
        int threads = total_frags < 20000 ? 1 : std::min<int>(tdt_host_thread_count(), 16);
        if (threads <= 1) work();
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; t++) pool.emplace_back(work);
            for (auto &t : pool) t.join();
        }
        for (int k = 0; k < 2; k++) {
            size_t total = 0;
            for (auto &s : part[k]) total += s.size();
            text[k].clear();
            text[k].reserve(total);
            seg[k].clear();
            blk_off[k].assign((size_t)n, 0);
            blk_len[k].assign((size_t)n, 0);
            for (size_t i = 0; i < np; i++) {
                if (!rows[k][i]) continue;
                const int a = pairs[order[i]]->a;
                seg[k].push_back(Segment{a, pairs[order[i]]->b, (int64_t)text[k].size(), (int64_t)part[k][i].size(), rows[k][i]});
                if (!blk_len[k][(size_t)a]) blk_off[k][(size_t)a] = (int64_t)text[k].size();
                blk_len[k][(size_t)a] += (int64_t)part[k][i].size();
                text[k] += part[k][i];
            }
        }
        formatted = true;
        return TDT_OK;
    }
};

static tdt_sigtab *as_tab(void *t) { return (tdt_sigtab *)t; }

extern "C" int tdt_sigtab_create(const char *names, const int64_t *lengths, int n_contigs, int64_t min_contig, void **out) {
    if (!out || n_contigs < 0 || (n_contigs && (!names || !lengths))) {
        tdt_set_error("tdt_sigtab_create: bad argument");
        return TDT_E_ARG;
    }
    std::unique_ptr<tdt_sigtab> t(new tdt_sigtab());
    t->n = n_contigs;
    t->min_contig = min_contig;
    const char *p = names;
    for (int i = 0; i < n_contigs; i++) {
        t->names.emplace_back(p);
        p += t->names.back().size() + 1;
        t->length.push_back(lengths[i]);
        t->kept.push_back(lengths[i] >= min_contig);
        t->id_of.emplace(t->names.back(), i);          // (a name twice in the header: the first id, like list.index)
    }
    std::vector<int32_t> idx((size_t)n_contigs);
    for (int i = 0; i < n_contigs; i++) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return t->names[(size_t)x] < t->names[(size_t)y]; });   // bytes of UTF-8 order like code points
    t->name_rank.assign((size_t)n_contigs, 0);
    int32_t rank = 0;
    for (int i = 0; i < n_contigs; i++) {
        if (i && t->names[(size_t)idx[(size_t)i]] != t->names[(size_t)idx[(size_t)i - 1]]) rank++;
        t->name_rank[(size_t)idx[(size_t)i]] = rank;
    }
    t->clips.resize((size_t)n_contigs);
    *out = t.release();
    return TDT_OK;
}

extern "C" void tdt_sigtab_destroy(void *t) { delete as_tab(t); }

// The selected reads of one batch, in file order (the arrays of tdt_signal_scan_result).  Clip entries (action bit 2), discordant
// rows (bit 8) and split rows (bit 4) are appended and merged.  A split read whose SA tag is outside what tdt_split_fields handles
// stops the call: *stopped = its index; the caller runs the literal SA_analysis on it (and raises what that raises), hands the row
// over with tdt_sigtab_add_split_row and calls again with resume = index + 1.  *stopped = n_sel when the batch is done.
extern "C" int tdt_sigtab_add(void *t_, const void *meta_, const uint32_t *raw_end, const uint8_t *raw, size_t n_sel, size_t raw_len, int min_q,
                              size_t resume, size_t *stopped) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !stopped || (n_sel && (!meta_ || !raw_end || !raw)) || resume > n_sel) {
        tdt_set_error("tdt_sigtab_add: bad argument");
        return TDT_E_ARG;
    }
    static const char SEQ[] = "=ACMGRSVTWYHKDBN";
    const uint8_t *meta = (const uint8_t *)meta_;
    *stopped = n_sel;
    for (size_t r = 0; r < n_sel && resume == 0; r++) {
        const uint8_t *m = meta + r * 28;
        const uint8_t action = m[26];
        if (!(action & 10)) continue;
        int32_t tid, pos, end, mate;
        uint16_t flag;
        memcpy(&tid, m + 4, 4);
        memcpy(&pos, m + 8, 4);
        memcpy(&end, m + 12, 4);
        memcpy(&mate, m + 16, 4);
        memcpy(&flag, m + 24, 2);
        const size_t rec0 = r ? raw_end[r - 1] : 0u, rec1 = raw_end[r];
        if (rec1 > raw_len || rec0 + 36 > rec1) {
            tdt_set_error("tdt_sigtab_add: record %zu outside the raw bytes", r);
            return TDT_E_ARG;
        }
        const uint8_t *rec = raw + rec0 + 4;                     // behind block_size: the 32 fixed bytes, then the name (NUL included)
        const int l_name = rec[8];
        const int nl = l_name ? l_name - 1 : 0;
        if (rec0 + 36 + (size_t)l_name > rec1) {
            tdt_set_error("tdt_sigtab_add: record %zu: name outside the record", r);
            return TDT_E_ARG;
        }
        if (tid < 0 || tid >= t->n) {
            tdt_set_error("tdt_sigtab_add: record %zu on contig id %d of %d", r, tid, t->n);
            return TDT_E_RANGE;
        }
        if (action & 2) {                                        // :192-197  ">{name}|{contig}|{pos+1}\n{sequence}\n"
            uint16_t n_cig;
            int32_t lseq;
            memcpy(&n_cig, rec + 12, 2);
            memcpy(&lseq, rec + 16, 4);
            if (lseq < 0) lseq = 0;
            const uint8_t *sq = rec + 32 + l_name + 4 * (size_t)n_cig;
            if (sq + ((size_t)lseq + 1) / 2 > raw + rec1) {
                tdt_set_error("tdt_sigtab_add: record %zu: sequence outside the record", r);
                return TDT_E_ARG;
            }
            std::string &c = t->clips[(size_t)tid];
            char num[24];
            c.push_back('>');
            c.append((const char *)rec + 32, (size_t)nl);
            c.push_back('|');
            c += t->names[(size_t)tid];
            c.push_back('|');
            c.append(num, fmt_i64((long long)pos + 1, num));
            c.push_back('\n');
            const size_t at = c.size();
            c.resize(at + (size_t)lseq + 1);
            char *q = &c[at];
            for (int i = 0; i < lseq; i++) q[i] = SEQ[(i & 1) ? (sq[i >> 1] & 0xf) : (sq[i >> 1] >> 4)];
            q[lseq] = '\n';
        }
        if (action & 8) {                                        // :204-221
            DRow d;
            d.tid = tid;
            d.mate = mate;
            d.start = pos + 1;
            d.end = end + 1;
            d.name_off = t->put(rec + 32, (size_t)nl);
            d.name_len = (uint16_t)nl;
            d.rev = (flag & 0x10) != 0;
            d.pad = 0;
            d.pad2 = 0;
            t->note_d(tid);
            t->dlog.push_back(d);
        }
    }
    for (size_t r = resume; r < n_sel; r++) {
        const uint8_t *m = meta + r * 28;
        if (!(m[26] & 4)) continue;
        TdtSplitOut o;
        tdt_split_one(meta, raw_end, raw, raw_len, (uint32_t)r, min_q, o);
        if (o.status == 0) continue;                             // SA mapQ below min_q: no row (:40-41)
        if (o.status != 1) {
            const int rc = t->merge_new();
            if (rc) return rc;
            *stopped = r;
            return TDT_OK;
        }
        int32_t tid;
        memcpy(&tid, m + 4, 4);
        if (tid < 0 || tid >= t->n) {
            tdt_set_error("tdt_sigtab_add: record %zu on contig id %d of %d", r, tid, t->n);
            return TDT_E_RANGE;
        }
        const size_t rec0 = r ? raw_end[r - 1] : 0u;
        const uint8_t *rec = raw + rec0 + 4;
        const int nl = rec[8] ? rec[8] - 1 : 0;
        const std::string sa_chr((const char *)raw + o.chr_off, o.chr_len);
        const std::string &chrom = t->names[(size_t)tid];
        auto it = t->id_of.find(sa_chr);
        const int sa_id = it == t->id_of.end() ? -1 : it->second;
        SRow s;
        memset(&s, 0, sizeof s);
        s.tid = tid;
        s.rev = o.is_reverse;
        s.sa_minus = o.sa_minus;
        // :118-140 — chrA is the smaller NAME; on one contig the smaller position comes first; the two orientations stay where they are
        if (sa_chr < chrom) {
            s.a = sa_id;
            s.b = tid;
            s.f[0] = o.sa_split; s.f[1] = o.split_pos; s.f[2] = o.seg_start; s.f[3] = o.seg_end; s.f[4] = o.read_start; s.f[5] = o.read_end;
        } else if (sa_chr == chrom && o.sa_split < o.split_pos) {
            s.a = tid;
            s.b = sa_id;
            s.f[0] = o.sa_split; s.f[1] = o.split_pos; s.f[2] = o.seg_start; s.f[3] = o.seg_end; s.f[4] = o.read_start; s.f[5] = o.read_end;
        } else {
            s.a = tid;
            s.b = sa_id;
            s.f[0] = o.split_pos; s.f[1] = o.sa_split; s.f[2] = o.read_start; s.f[3] = o.read_end; s.f[4] = o.seg_start; s.f[5] = o.seg_end;
        }
        s.name_off = t->put(rec + 32, (size_t)nl);
        s.name_len = (uint16_t)nl;
        if (sa_id < 0) {
            s.other_off = t->put(sa_chr.data(), sa_chr.size());
            s.other_len = (uint32_t)sa_chr.size();
        }
        t->note_s(tid);
        t->slog.push_back(s);
        t->snext.push_back(NONE);
    }
    return t->merge_new();
}

// One split row built by the caller (the literal SA_analysis, for tags tdt_split_fields does not take): the read's contig id, the two
// contig NAMES as the row has them, the fragment name, six = {split_pos, sa_split, startA, endA, startB, endB} in the row's order.
extern "C" int tdt_sigtab_add_split_row(void *t_, int tid, const char *chrA, const char *chrB, const char *qname, const int64_t *six, int is_reverse,
                                        int sa_minus) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !chrA || !chrB || !qname || !six || tid < 0 || tid >= t->n || strlen(qname) > 0xffff) {
        tdt_set_error("tdt_sigtab_add_split_row: bad argument");
        return TDT_E_ARG;
    }
    SRow s;
    memset(&s, 0, sizeof s);
    s.tid = tid;
    s.rev = is_reverse != 0;
    s.sa_minus = sa_minus != 0;
    auto ia = t->id_of.find(chrA), ib = t->id_of.find(chrB);
    s.a = ia == t->id_of.end() ? -1 : ia->second;
    s.b = ib == t->id_of.end() ? -1 : ib->second;
    if (s.a < 0 && s.b < 0) {
        tdt_set_error("tdt_sigtab_add_split_row: neither contig name is in the header");
        return TDT_E_ARG;
    }
    for (int k = 0; k < 6; k++) s.f[k] = six[k];
    s.name_len = (uint16_t)strlen(qname);
    s.name_off = t->put(qname, s.name_len);
    if (s.a < 0 || s.b < 0) {
        const char *o = s.a < 0 ? chrA : chrB;
        s.other_len = (uint32_t)strlen(o);
        s.other_off = t->put(o, s.other_len);
    }
    t->note_s(tid);
    t->slog.push_back(s);
    t->snext.push_back(NONE);
    return t->merge_new();
}

// clip FASTA bytes of contig `tid` appended behind what is there (the N-rank job: another rank's share of the contig, in rank order)
extern "C" int tdt_sigtab_add_clips(void *t_, int tid, const char *bytes, size_t len) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || tid < 0 || tid >= t->n || (len && !bytes)) {
        tdt_set_error("tdt_sigtab_add_clips: bad argument");
        return TDT_E_ARG;
    }
    t->clips[(size_t)tid].append(bytes, len);
    return TDT_OK;
}

extern "C" int tdt_sigtab_clips(void *t_, int tid, const char **ptr, size_t *len) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || tid < 0 || tid >= t->n || !ptr || !len) {
        tdt_set_error("tdt_sigtab_clips: bad argument");
        return TDT_E_ARG;
    }
    *ptr = t->clips[(size_t)tid].data();
    *len = t->clips[(size_t)tid].size();
    return TDT_OK;
}

// out[0..8) = discordant rows, split rows, contig pairs, discordant rows merged incrementally (1) or at finalize (0), the same for
// splits, arena bytes, clip bytes, reserved
extern "C" int tdt_sigtab_stats(void *t_, int64_t *out) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !out) {
        tdt_set_error("tdt_sigtab_stats: bad argument");
        return TDT_E_ARG;
    }
    size_t cb = 0;
    for (auto &c : t->clips) cb += c.size();
    out[0] = (int64_t)t->dlog.size();
    out[1] = (int64_t)t->slog.size();
    out[2] = (int64_t)t->pairs.size();
    out[3] = t->d_ordered;
    out[4] = t->s_ordered;
    out[5] = (int64_t)t->arena.size();
    out[6] = (int64_t)cb;
    out[7] = 0;
    return TDT_OK;
}

// ---- the row log as one blob: 32-byte header {magic, n discordant rows, n split rows, name bytes}, DRow[], SRow[], names (offsets
// relative to the names block).  owner == NULL: every row (what worker() returned, contig by contig: the caller filters on tid);
// else only the rows whose chrA is a contig >= min_contig owned by rank `dest` (owner[contig id] = rank) — the rows main()'s merge
// would skip are not sent.  Two-call protocol (out == NULL: size).
extern "C" int tdt_sigtab_export(void *t_, const int32_t *owner, int dest, void *out, size_t cap, size_t *need) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !need) {
        tdt_set_error("tdt_sigtab_export: bad argument");
        return TDT_E_ARG;
    }
    auto d_goes = [&](const DRow &x) {
        if (!owner) return true;
        if (x.tid < 0 || x.tid >= t->n || x.mate < 0 || x.mate >= t->n) return true;      // (reported by the merge of whoever gets it)
        const int a = t->name_rank[(size_t)x.mate] < t->name_rank[(size_t)x.tid] ? x.mate : x.tid;
        return t->kept[(size_t)a] && owner[a] == dest;
    };
    auto s_goes = [&](const SRow &x) {
        if (!owner) return true;
        return x.a >= 0 && t->kept[(size_t)x.a] && owner[x.a] == dest;
    };
    uint64_t nd = 0, ns = 0, nb = 0;
    for (const DRow &x : t->dlog)
        if (d_goes(x)) {
            nd++;
            nb += x.name_len;
        }
    for (const SRow &x : t->slog)
        if (s_goes(x)) {
            ns++;
            nb += (uint64_t)x.name_len + x.other_len;
        }
    const size_t total = 32 + nd * sizeof(DRow) + ns * sizeof(SRow) + nb;
    *need = total;
    if (!out) return TDT_OK;
    if (cap < total) {
        tdt_set_error("tdt_sigtab_export: buffer of %zu bytes, %zu needed", cap, total);
        return TDT_E_ARG;
    }
    uint64_t head[4] = {0x3142415447495354ull, nd, ns, nb};
    memcpy(out, head, 32);
    DRow *dr = (DRow *)((char *)out + 32);
    SRow *sr = (SRow *)((char *)out + 32 + nd * sizeof(DRow));
    char *nm = (char *)out + 32 + nd * sizeof(DRow) + ns * sizeof(SRow);
    uint64_t o = 0;
    for (const DRow &x : t->dlog)
        if (d_goes(x)) {
            DRow y = x;
            y.name_off = o;
            memcpy(nm + o, t->arena.data() + x.name_off, x.name_len);
            o += x.name_len;
            memcpy(dr++, &y, sizeof y);
        }
    for (const SRow &x : t->slog)
        if (s_goes(x)) {
            SRow y = x;
            y.name_off = o;
            memcpy(nm + o, t->arena.data() + x.name_off, x.name_len);
            o += x.name_len;
            y.other_off = o;
            if (x.other_len) memcpy(nm + o, t->arena.data() + x.other_off, x.other_len);
            o += x.other_len;
            memcpy(sr++, &y, sizeof y);
        }
    return TDT_OK;
}

// rows of another table (tdt_sigtab_export of a table over the SAME header) appended behind this table's rows
extern "C" int tdt_sigtab_import(void *t_, const void *blob, size_t len) {
    tdt_sigtab *t = as_tab(t_);
    uint64_t head[4];
    if (!t || !blob || len < 32) {
        tdt_set_error("tdt_sigtab_import: bad argument");
        return TDT_E_ARG;
    }
    memcpy(head, blob, 32);
    const uint64_t nd = head[1], ns = head[2], nb = head[3];
    if (head[0] != 0x3142415447495354ull || nd > len / sizeof(DRow) || ns > len / sizeof(SRow) || 32 + nd * sizeof(DRow) + ns * sizeof(SRow) + nb != len) {
        tdt_set_error("tdt_sigtab_import: not a row blob (or truncated)");
        return TDT_E_ARG;
    }
    const char *dr = (const char *)blob + 32, *sr = dr + nd * sizeof(DRow), *nm = sr + ns * sizeof(SRow);
    const uint64_t base = t->put(nm, nb);
    for (uint64_t i = 0; i < nd; i++) {
        DRow x;
        memcpy(&x, dr + i * sizeof(DRow), sizeof x);
        if (x.name_off + x.name_len > nb) {
            tdt_set_error("tdt_sigtab_import: a name outside the blob");
            return TDT_E_ARG;
        }
        x.name_off += base;
        t->note_d(x.tid);
        t->dlog.push_back(x);
    }
    for (uint64_t i = 0; i < ns; i++) {
        SRow x;
        memcpy(&x, sr + i * sizeof(SRow), sizeof x);
        if (x.name_off + x.name_len > nb || x.other_off + x.other_len > nb || x.a >= t->n || x.b >= t->n) {
            tdt_set_error("tdt_sigtab_import: a row outside the blob or the header");
            return TDT_E_ARG;
        }
        x.name_off += base;
        x.other_off += base;
        t->note_s(x.tid);
        t->slog.push_back(x);
        t->snext.push_back(NONE);
    }
    return t->merge_new();
}

// ---- the two .tab tables as text (tiddit_signal.pyx:298-326), contig pairs in the order of main()'s nested dictionaries
extern "C" int tdt_sigtab_format(void *t_, size_t *n_segments_disc, size_t *n_segments_split) {
    tdt_sigtab *t = as_tab(t_);
    if (!t) {
        tdt_set_error("tdt_sigtab_format: bad argument");
        return TDT_E_ARG;
    }
    const int rc = t->format();
    if (rc) return rc;
    if (n_segments_disc) *n_segments_disc = t->seg[0].size();
    if (n_segments_split) *n_segments_split = t->seg[1].size();
    return TDT_OK;
}

// kind 0 = discordants, 1 = splits.  *ptr stays valid until the table changes.  segments (may be NULL): 5 int64 per contig pair that
// has rows — chrA id, chrB id, offset, length, rows.
extern "C" int tdt_sigtab_text(void *t_, int kind, const char **ptr, size_t *len, int64_t *segments) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || (kind != 0 && kind != 1) || !ptr || !len) {
        tdt_set_error("tdt_sigtab_text: bad argument");
        return TDT_E_ARG;
    }
    const int rc = t->format();
    if (rc) return rc;
    *ptr = t->text[kind].data();
    *len = t->text[kind].size();
    if (segments)
        for (size_t i = 0; i < t->seg[kind].size(); i++) {
            const Segment &s = t->seg[kind][i];
            int64_t *o = segments + 5 * i;
            o[0] = s.a; o[1] = s.b; o[2] = s.off; o[3] = s.len; o[4] = s.rows;
        }
    return TDT_OK;
}

// Bytes per contig of one output: what 0 = the rows of discordants_{sample}.tab whose chrA is the contig, 1 = the same for
// splits_{sample}.tab, 2 = the contig's clip FASTA.  A file is these blocks in header order (:298-332), so with the sizes of every
// rank known a rank can place its own blocks: tdt_sigtab_pwrite writes one block at `offset` of the open file `fd`.
extern "C" int tdt_sigtab_sizes(void *t_, int what, int64_t *out) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || what < 0 || what > 2 || (t->n && !out)) {
        tdt_set_error("tdt_sigtab_sizes: bad argument");
        return TDT_E_ARG;
    }
    if (what == 2) {
        for (int i = 0; i < t->n; i++) out[i] = (int64_t)t->clips[(size_t)i].size();
        return TDT_OK;
    }
    const int rc = t->format();
    if (rc) return rc;
    for (int i = 0; i < t->n; i++) out[i] = t->blk_len[what][(size_t)i];
    return TDT_OK;
}

extern "C" int tdt_sigtab_pwrite(void *t_, int what, int contig, int fd, int64_t offset) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || what < 0 || what > 2 || contig < 0 || contig >= t->n || fd < 0 || offset < 0) {
        tdt_set_error("tdt_sigtab_pwrite: bad argument");
        return TDT_E_ARG;
    }
    const char *p;
    size_t len;
    if (what == 2) {
        p = t->clips[(size_t)contig].data();
        len = t->clips[(size_t)contig].size();
    } else {
        const int rc = t->format();
        if (rc) return rc;
        p = t->text[what].data() + t->blk_off[what][(size_t)contig];
        len = (size_t)t->blk_len[what][(size_t)contig];
    }
    while (len) {
        const ssize_t w = pwrite(fd, p, len, (off_t)offset);
        if (w <= 0) {
            tdt_set_error("tdt_sigtab_pwrite: write failed (%s)", strerror(errno));
            return TDT_E_ARG;
        }
        p += w;
        len -= (size_t)w;
        offset += w;
    }
    return TDT_OK;
}

// ---- tiddit_cluster.main's signal table (tiddit_cluster.pyx:47-105) from the merged tables: per (chrA,chrB) bucket — both contigs
// >= min_contig — the written discordant fragments in file order, then the split fragments; posA / posB as find_discordant_pos (:7-37)
// picks them, clipped to the contig lengths with the QUIRK of :67-70 (posB is never clipped, posA takes chrB's length).  Buckets in
// header order of (chrA, chrB): the order of the loops at :140-147.
extern "C" int tdt_sigtab_cluster_table(void *t_, int is_mp, int64_t min_contig, size_t *n_signals, int *n_buckets) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !n_signals || !n_buckets) {
        tdt_set_error("tdt_sigtab_cluster_table: bad argument");
        return TDT_E_ARG;
    }
    int rc = t->finalize();
    if (rc) return rc;
    t->sigs.clear();
    t->bucket_off.assign(1, 0);
    t->bucket_a.clear();
    t->bucket_b.clear();
    bool range_ok = true;
    auto fits = [&](int64_t v) {
        if (v < -(1ll << 31) || v >= (1ll << 31)) range_ok = false;
        return (int32_t)v;
    };
    for (uint32_t pi : t->order) {
        const PairTab &p = *t->pairs[pi];
        const int64_t lenA = t->length[(size_t)p.a], lenB = t->length[(size_t)p.b];
        if (lenA < min_contig || lenB < min_contig) continue;                // :52
        const size_t before = t->sigs.size();
        for (const Frag &f : p.d.frags) {
            const DRow *x, *y;
            if (!t->d_written(p, f, &x, &y)) continue;
            int64_t posA, posB;                                              // :7-37 on the row [.., startA 3, endA 4, revA 5, startB 6, endB 7, revB 8]
            if (is_mp) {
                if (!x->rev && y->rev) { posA = x->start; posB = y->end; }
                else if (!x->rev && !y->rev) { posA = x->start; posB = y->start; }
                else if (x->rev && y->rev) { posA = x->end; posB = y->end; }
                else { posA = x->end; posB = y->start; }
            } else {
                if (!x->rev && y->rev) { posA = x->end; posB = y->start; }
                else if (!x->rev && !y->rev) { posA = x->end; posB = y->end; }
                else if (x->rev && y->rev) { posA = x->start; posB = y->start; }
                else { posA = x->start; posB = y->end; }
            }
            if (posA > lenA) {
                posA = lenA;
                if (posB > lenB) posA = lenB;                                // QUIRK (:67-70)
            }
            Sig s{fits(posA), fits(posB), x->start, x->end, y->start, y->end, f.row0, 0, x->rev, y->rev, 0};
            t->sigs.push_back(s);
        }
        for (const Frag &f : p.s.frags) {
            const SRow &r = t->slog[f.row0];                                 // (the text row's first eight fields: a longer row's tail is never read, :96)
            const int64_t posA = r.f[0] > lenA ? lenA : r.f[0], posB = r.f[1] > lenB ? lenB : r.f[1];
            Sig s{fits(posA), fits(posB), fits(r.f[2]), fits(r.f[3]), fits(r.f[4]), fits(r.f[5]), f.row0, 1, r.rev, r.sa_minus, 0};
            t->sigs.push_back(s);
        }
        if (t->sigs.size() != before) {
            t->bucket_off.push_back((int64_t)t->sigs.size());
            t->bucket_a.push_back(p.a);
            t->bucket_b.push_back(p.b);
        }
    }
    if (!range_ok) {
        t->sigs.clear();
        tdt_set_error("tdt_sigtab_cluster_table: a coordinate outside int32");
        return TDT_E_UNSUPPORTED;
    }
    *n_signals = t->sigs.size();
    *n_buckets = (int)t->bucket_a.size();
    return TDT_OK;
}

// posA / posB: n_signals int32 each (pinned memory: they go to tdt_cluster_columns as they are); bucket_off: n_buckets + 1;
// bucket_a / bucket_b: contig ids of every bucket
extern "C" int tdt_sigtab_cluster_columns(void *t_, int32_t *posA, int32_t *posB, int64_t *bucket_off, int32_t *bucket_a, int32_t *bucket_b) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !bucket_off || (t->sigs.size() && (!posA || !posB)) || (t->bucket_a.size() && (!bucket_a || !bucket_b))) {
        tdt_set_error("tdt_sigtab_cluster_columns: bad argument");
        return TDT_E_ARG;
    }
    for (size_t i = 0; i < t->sigs.size(); i++) {
        posA[i] = t->sigs[i].posA;
        posB[i] = t->sigs[i].posB;
    }
    memcpy(bucket_off, t->bucket_off.data(), t->bucket_off.size() * 8);
    if (!t->bucket_a.empty()) {
        memcpy(bucket_a, t->bucket_a.data(), t->bucket_a.size() * 4);
        memcpy(bucket_b, t->bucket_b.data(), t->bucket_b.size() * 4);
    }
    return TDT_OK;
}

// ---- tiddit_cluster.pyx:156-254 without the per-row Python: labels[i] = cluster of signal i (-1 = noise), as tdt_cluster_columns
// returns them.  A candidate is a (bucket, cluster id); candidates of a bucket in the order their id first appears in signal order
// (the insertion order of candidates[chrA][chrB], which later numbers the VCF records).
extern "C" int tdt_sigtab_regroup(void *t_, const int32_t *labels, size_t *n_candidates, size_t *n_members, size_t *name_bytes) {
    tdt_sigtab *t = as_tab(t_);
    if (!t || !n_candidates || !n_members || !name_bytes || (t->sigs.size() && !labels)) {
        tdt_set_error("tdt_sigtab_regroup: bad argument");
        return TDT_E_ARG;
    }
    t->cand.clear();
    t->mem_all.clear();
    t->mem_kind.clear();
    std::vector<int32_t> slot;                   // cluster id -> candidate index of this bucket, -1
    size_t names = 0;
    for (size_t b = 0; b + 1 < t->bucket_off.size(); b++) {
        const size_t lo = (size_t)t->bucket_off[b], hi = (size_t)t->bucket_off[b + 1];
        int32_t top = -1;
        for (size_t i = lo; i < hi; i++) top = std::max(top, labels[i]);
        if (top < 0) continue;
        slot.assign((size_t)top + 1, -1);
        const size_t c0 = t->cand.size() / 4;
        for (size_t i = lo; i < hi; i++) {
            const int32_t l = labels[i];
            if (l < 0) continue;
            if (slot[(size_t)l] < 0) {
                slot[(size_t)l] = (int32_t)(t->cand.size() / 4 - c0);
                const int32_t row[4] = {(int32_t)b, l, 0, 0};
                t->cand.insert(t->cand.end(), row, row + 4);
            }
            t->cand[(c0 + (size_t)slot[(size_t)l]) * 4 + 2 + t->sigs[i].kind]++;
        }
        const size_t nc = t->cand.size() / 4 - c0;
        std::vector<uint32_t> start(nc + 1, 0);
        for (size_t c = 0; c < nc; c++) start[c + 1] = start[c] + (uint32_t)(t->cand[(c0 + c) * 4 + 2] + t->cand[(c0 + c) * 4 + 3]);
        const size_t m0 = t->mem_all.size();
        t->mem_all.resize(m0 + start[nc]);
        t->mem_kind.resize(m0 + start[nc]);
        std::vector<uint32_t> at_all(start.begin(), start.end() - 1), at_d(start.begin(), start.end() - 1), at_s(nc);
        for (size_t c = 0; c < nc; c++) at_s[c] = start[c] + (uint32_t)t->cand[(c0 + c) * 4 + 2];
        for (size_t i = lo; i < hi; i++) {
            const int32_t l = labels[i];
            if (l < 0) continue;
            const size_t c = (size_t)slot[(size_t)l];
            t->mem_all[m0 + at_all[c]++] = (uint32_t)i;
            if (t->sigs[i].kind == 0) t->mem_kind[m0 + at_d[c]++] = (uint32_t)i;
            else t->mem_kind[m0 + at_s[c]++] = (uint32_t)i;
            names += (t->sigs[i].kind == 0 ? t->dlog[t->sigs[i].row].name_len : t->slog[t->sigs[i].row].name_len) + 1u;
        }
    }
    *n_candidates = t->cand.size() / 4;
    *n_members = t->mem_all.size();
    *name_bytes = names;
    return TDT_OK;
}

// cand: 4 int32 per candidate (bucket, cluster id, discordant members, split members).  Members are candidate-major.  In signal
// order: startA, endA, startB, endB (the `start` / `end` lists of positions_A / positions_B, :216-219).  Discordants first, then
// splits, each in signal order: posA, posB, the two orientations (0 / 1) and the fragment names ('\n' behind each).
extern "C" int tdt_sigtab_regroup_result(void *t_, int32_t *cand, int32_t *sA, int32_t *eA, int32_t *sB, int32_t *eB, int32_t *kposA, int32_t *kposB,
                                         uint8_t *koriA, uint8_t *koriB, char *names) {
    tdt_sigtab *t = as_tab(t_);
    const size_t m = t ? t->mem_all.size() : 0;
    if (!t || (t->cand.size() && !cand) || (m && (!sA || !eA || !sB || !eB || !kposA || !kposB || !koriA || !koriB || !names))) {
        tdt_set_error("tdt_sigtab_regroup_result: bad argument");
        return TDT_E_ARG;
    }
    if (!t->cand.empty()) memcpy(cand, t->cand.data(), t->cand.size() * 4);
    char *q = names;
    for (size_t k = 0; k < m; k++) {
        const Sig &s = t->sigs[t->mem_all[k]];
        sA[k] = s.sA;
        eA[k] = s.eA;
        sB[k] = s.sB;
        eB[k] = s.eB;
        const Sig &g = t->sigs[t->mem_kind[k]];
        kposA[k] = g.posA;
        kposB[k] = g.posB;
        koriA[k] = g.oriA;
        koriB[k] = g.oriB;
        uint16_t len;
        const char *nm = g.kind == 0 ? t->dname(g.row, &len) : t->sname(g.row, &len);
        memcpy(q, nm, len);
        q += len;
        *q++ = '\n';
    }
    return TDT_OK;
}
