// Host-side BGZF inflate (no device code): the ingest stage in front of tdt_bam_decode.  The reference reads BAM through
// pysam/htslib (`pysam.AlignmentFile(..., threads=n)`, tiddit_signal.pyx:159, __main__.py:224), which inflates BGZF blocks on
// a small worker pool behind a per-read iterator.  Here a whole span of blocks is scanned (header hops only) and then
// inflated in parallel straight into the caller's record buffer — every block knows its output offset from ISIZE — with
// CRC32 and ISIZE verified per block, so the decode stage sees one contiguous byte stream.
#include "tdt_common.h"

#include <atomic>
#include <thread>
#include <vector>
#include <zlib.h>

static std::atomic<int> g_host_threads{0};

int tdt_host_thread_count() {
    int t = g_host_threads.load();
    if (t <= 0) {
        const char *e = getenv("TIDDIT_HOST_THREADS");
        t = e ? atoi(e) : 0;
        if (t <= 0) {
            t = (int)std::thread::hardware_concurrency();
            if (t > 64) t = 64;
            if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // container CPU quota: more threads than that only throttle
                long long quota = 0, period = 0;
                if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
                    const int q = (int)((quota + period - 1) / period);
                    if (q >= 1 && q < t) t = q;
                }
                fclose(f);
            }
        }
        if (t < 1) t = 1;
    }
    return t;
}

extern "C" int tdt_host_threads(int n) {
    const int prev = tdt_host_thread_count();
    if (n > 0) g_host_threads.store(n);
    return prev;
}

static inline uint16_t bz_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t bz_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// One header hop.  -> 1 block complete (sets total size, payload offset/size, isize), 0 need more bytes, <0 malformed.
int tdt_bz_hop(const uint8_t *p, size_t avail, size_t *bsize, size_t *pay_off, size_t *pay_len, uint32_t *isize) {
    if (avail < 18) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
    const size_t xlen = bz_u16(p + 10);
    if (avail < 12 + xlen) return 0;
    long bs = -1;
    for (size_t o = 0; o + 4 <= xlen;) {
        const uint8_t *x = p + 12 + o;
        const size_t slen = bz_u16(x + 2);
        if (x[0] == 66 && x[1] == 67 && slen == 2 && o + 6 <= xlen) bs = (long)bz_u16(x + 4) + 1;
        o += 4 + slen;
    }
    if (bs < 0 || (size_t)bs < 12 + xlen + 8) return -1;
    if (avail < (size_t)bs) return 0;
    *bsize = (size_t)bs;
    *pay_off = 12 + xlen;
    *pay_len = (size_t)bs - xlen - 20;
    *isize = bz_u32(p + bs - 4);
    return 1;
}

extern "C" int tdt_bgzf_scan(const uint8_t *comp, size_t len, size_t max_out, size_t *n_blocks, size_t *consumed, size_t *produced) {
    if ((!comp && len) || !n_blocks || !consumed || !produced) {
        tdt_set_error("tdt_bgzf_scan: bad argument");
        return TDT_E_ARG;
    }
    size_t o = 0, out = 0, nb = 0;
    while (o < len) {
        size_t bs, po, pl;
        uint32_t isz;
        const int r = tdt_bz_hop(comp + o, len - o, &bs, &po, &pl, &isz);
        if (r < 0) {
            tdt_set_error("tdt_bgzf_scan: not a BGZF block at offset %zu", o);
            return TDT_E_ARG;
        }
        if (r == 0 || out + isz > max_out) break;
        o += bs;
        out += isz;
        nb++;
    }
    *n_blocks = nb;
    *consumed = o;
    *produced = out;
    return TDT_OK;
}

struct BzBlock {
    size_t in_off, in_len, out_off;
    uint32_t isize, crc;
};

extern "C" int tdt_bgzf_inflate(const uint8_t *comp, size_t len, uint8_t *out, size_t out_len, int threads) {
    if ((!comp && len) || (!out && out_len)) {
        tdt_set_error("tdt_bgzf_inflate: bad argument");
        return TDT_E_ARG;
    }
    std::vector<BzBlock> blocks;
    size_t o = 0, uo = 0;
    while (o < len) {
        size_t bs, po, pl;
        uint32_t isz;
        if (tdt_bz_hop(comp + o, len - o, &bs, &po, &pl, &isz) != 1) {
            tdt_set_error("tdt_bgzf_inflate: input is not a whole number of BGZF blocks (offset %zu)", o);
            return TDT_E_ARG;
        }
        blocks.push_back(BzBlock{o + po, pl, uo, isz, bz_u32(comp + o + bs - 8)});
        o += bs;
        uo += isz;
    }
    if (uo != out_len) {
        tdt_set_error("tdt_bgzf_inflate: blocks inflate to %zu bytes, caller gave %zu", uo, out_len);
        return TDT_E_ARG;
    }
    if (threads <= 0) threads = tdt_host_thread_count();
    if ((size_t)threads > blocks.size()) threads = blocks.size() ? (int)blocks.size() : 1;
    std::atomic<size_t> next{0};
    std::atomic<long> bad{-1};
    auto work = [&]() {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) {
            bad.store(-2);
            return;
        }
        for (;;) {
            const size_t b0 = next.fetch_add(16);                       // 16 blocks (~1 MB of output) per grab
            if (b0 >= blocks.size() || bad.load() != -1) break;
            const size_t b1 = b0 + 16 < blocks.size() ? b0 + 16 : blocks.size();
            for (size_t b = b0; b < b1; b++) {
                const BzBlock &B = blocks[b];
                if (B.isize == 0) continue;
                inflateReset(&zs);
                zs.next_in = const_cast<Bytef *>(comp + B.in_off);
                zs.avail_in = (uInt)B.in_len;
                zs.next_out = out + B.out_off;
                zs.avail_out = B.isize;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), out + B.out_off, B.isize) != B.crc) {
                    bad.store((long)b);
                    break;
                }
            }
        }
        inflateEnd(&zs);
    };
    if (threads == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (bad.load() != -1) {
        tdt_set_error("tdt_bgzf_inflate: block %ld failed to inflate or its CRC32/ISIZE does not match", bad.load());
        return TDT_E_ARG;
    }
    return TDT_OK;
}
