// Native BAM decoder of libmdx.so (host side of the boundary): BGZF blocks are
// inflated on several host threads and every record is unpacked straight into the SoA columns of
// mdx_batch — the counterpart of iterating a pysam.AlignmentFile (mapdamage/reader.py:38,83-96,
// 121-132) without pysam.  Pure host code (zlib); no HIP calls.
#include "../../include/mdx.h"
#ifndef MDX_HOST_ONLY      // (tools/sanitize/run_bamio.sh builds the host decoder alone, without the HIP runtime)
#include "mdx_internal.h"
#endif
#include "mdx_crc32.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <memory>
#include <vector>

// vector whose resize() leaves new elements uninitialised (the decoder overwrites every one of them; a
// value-initialising resize would touch hundreds of megabytes on one thread first).  Large blocks come straight
// from mmap with MADV_HUGEPAGE: the decoder first-touches and later returns hundreds of megabytes at a time, and
// with 4 KiB pages both are per-page work under the process-wide mmap lock (releasing the 800 MB inflated stream
// of a 4 M-record file held that lock for 100 ms, stalling whoever allocated next).
// Released blocks are kept (up to kPoolBytes) and handed out again: the chunked decoder allocates and releases the
// same few hundred megabytes every chunk, and a recycled block needs neither page faults nor an munmap.
class BigBlocks {
  public:
    static constexpr size_t kHuge = (size_t)2 << 20, kPoolBytes = (size_t)3 << 30, kPoolBlock = (size_t)1 << 30;
    static BigBlocks &get() { static BigBlocks *pool = new BigBlocks(); return *pool; }   // never destroyed: detached
                                                                                          // threads may still free
    void *take(size_t bytes) {
        const size_t need = (bytes + kHuge - 1) / kHuge * kHuge;
        {
            std::lock_guard<std::mutex> guard(mu_);
            size_t best = idle_.size();
            for (size_t i = 0; i < idle_.size(); i++)
                if (idle_[i].second >= need && idle_[i].second <= need + need / 2 &&
                    (best == idle_.size() || idle_[i].second < idle_[best].second)) best = i;
            if (best != idle_.size()) {
                void *p = idle_[best].first;
                live_[p] = idle_[best].second;
                idle_bytes_ -= idle_[best].second;
                idle_.erase(idle_.begin() + (std::ptrdiff_t)best);
                return p;
            }
        }
        void *p = mmap(nullptr, need, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(p, need, MADV_HUGEPAGE);
        std::lock_guard<std::mutex> guard(mu_);
        live_[p] = need;
        return p;
    }
    void give(void *p) {
        size_t span = 0;
        {
            std::lock_guard<std::mutex> guard(mu_);
            auto it = live_.find(p);
            if (it == live_.end()) return;
            span = it->second;
            live_.erase(it);
            if (span <= kPoolBlock && idle_bytes_ + span <= kPoolBytes) {
                idle_.emplace_back(p, span);
                idle_bytes_ += span;
                return;
            }
        }
        (void)munmap(p, span);
    }

  private:
    std::mutex mu_;
    std::unordered_map<void *, size_t> live_;
    std::vector<std::pair<void *, size_t>> idle_;
    size_t idle_bytes_ = 0;
};

template <class T>
struct no_init_alloc {
    typedef T value_type;
    static constexpr size_t kBig = (size_t)4 << 20;
    template <class U> struct rebind { typedef no_init_alloc<U> other; };
    no_init_alloc() = default;
    template <class U> no_init_alloc(const no_init_alloc<U> &) {}
    T *allocate(size_t n) {
        if (n * sizeof(T) < kBig) return static_cast<T *>(::operator new(n * sizeof(T)));
        return static_cast<T *>(BigBlocks::get().take(n * sizeof(T)));
    }
    void deallocate(T *p, size_t n) noexcept {
        if (n * sizeof(T) < kBig) ::operator delete(p);
        else BigBlocks::get().give(p);
    }
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new ((void *)p) U(std::forward<Args>(args)...); }
    template <class U> bool operator==(const no_init_alloc<U> &) const { return true; }
    template <class U> bool operator!=(const no_init_alloc<U> &) const { return false; }
};
typedef std::vector<uint8_t, no_init_alloc<uint8_t>> raw_bytes;

struct mdx_bam {
    std::string error;
    std::string header_text;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lengths;
    // SoA columns
    std::vector<uint16_t> flag, lib;
    std::vector<int32_t> tid, pos, tlen, mtid, mpos, rg_index;
    std::vector<uint32_t> cigar_off, cigar, seq_off, qname_off;
    raw_bytes seq, qual;
    std::string qnames;
    std::vector<std::string> rg_names;
    std::vector<uint8_t> has_mr;
    std::vector<uint8_t> qmin;       // lowest quality of each record (0xFF: no bases)
    // the chunk's encoded records as they stood in the file (mdx_bam_stream_keep_raw): rewriting a BAM preserves every
    // byte it does not change.  rec_off[i] = offset of record i's block_size field, rec_off[n] = end
    raw_bytes raw;
    std::vector<uint64_t> rec_off;
    // returning hundreds of megabytes of touched pages to the system takes tens of milliseconds: the inflated
    // stream of mdx_bam_read is released on this thread while the caller already works on the columns
    std::thread reaper;
    ~mdx_bam() { if (reaper.joinable()) reaper.join(); }
};


namespace { struct MappedFile; }

// Streaming decode: the file is consumed a slab of BGZF blocks at a time, so host memory is bounded by the
// chunk size and the caller can tabulate chunk k while chunk k+1 is being decoded.
struct mdx_bam_stream {
    mdx_bam head;                    // header text + reference dictionary (no records); also carries the error text
    MappedFile *file = nullptr;
    size_t coff = 0;                 // compressed offset of the first block not inflated yet
    int threads = 1;
    bool eof = false;
    bool keep_raw = false;           // chunks keep their encoded records (mdx_bam_raw)
    raw_bytes pending;               // inflated bytes not unpacked yet (a partial record at most, between calls)
    std::vector<size_t> hints;       // offsets into `pending` where a BGZF block began (unpack_records)
    size_t inflated = 0;             // uncompressed bytes inflated so far
    size_t header_bytes = 0;         // uncompressed size of the BAM header (set by mdx_bam_open)
    size_t skip = 0;                 // inflated bytes to drop in front of the next record (mdx_bam_seek)
};

namespace {

struct Block { size_t in_off, in_size, out_off, out_size; uint32_t crc; };   // crc: CRC32 of the inflated bytes (gzip trailer)

// The compressed file, mapped read-only: the inflating threads read the page cache directly (an fread of the
// whole file into a buffer first was a single-threaded copy, a quarter of the decode time on a 64-thread host).
struct MappedFile {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;                     // kept open: the device decode's host threads read their blocks with pread()
    bool open(const char *path, std::string &err) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); fd = -1; err = std::string("not a regular file: ") + path; return false; }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); fd = -1; n = 0; err = std::string("cannot map ") + path; return false; }
            (void)madvise(m, n, MADV_SEQUENTIAL);
            p = (const uint8_t *)m;
        }
        return true;
    }
    void close() { if (p) munmap((void *)p, n); p = nullptr; n = 0; if (fd >= 0) ::close(fd); fd = -1; }
    ~MappedFile() { close(); }
    size_t size() const { return n; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
};

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const uint8_t *p) { return (int32_t)rd32(p); }

// `partial`: the buffer may end inside a block (streaming); the scan stops there and `consumed` says how far
// it got.  Otherwise a truncated block is an error.
// `from`/`want`: scan from that offset and stop after about `want` compressed bytes (streaming).
bool scan_blocks(const MappedFile &file, std::vector<Block> &blocks, size_t &total, std::string &err,
                 size_t from = 0, size_t want = ~(size_t)0, size_t *consumed = nullptr) {
    size_t off = from;
    const bool partial = false;      // a mapped file is all there: a block running past its end is corrupt
    total = 0;
    if (consumed) *consumed = from;
    while (off < file.size() && off - from < want) {
        if (partial && off + 18 > file.size()) break;
        if (off + 18 > file.size() || file[off] != 0x1f || file[off + 1] != 0x8b || !(file[off + 3] & 4)) {
            err = "not a BGZF-compressed file";
            return false;
        }
        const size_t xlen = rd16(&file[off + 10]);
        size_t x = off + 12, xend = x + xlen;
        if (partial && xend > file.size()) break;
        size_t bsize = 0;
        while (x + 4 <= xend && xend <= file.size()) {
            const size_t slen = rd16(&file[x + 2]);
            if (x + 4 + slen > xend) break;                     // (a subfield may not run past the extra field)
            if (file[x] == 'B' && file[x + 1] == 'C' && slen == 2) bsize = (size_t)rd16(&file[x + 4]) + 1;
            x += 4 + slen;
        }
        if (partial && bsize && bsize >= xlen + 20 && off + bsize > file.size()) break;
        if (!bsize || off + bsize > file.size() || bsize < xlen + 20) { err = "corrupt BGZF block"; return false; }
        Block b;
        b.in_off = off + 12 + xlen;
        b.in_size = bsize - xlen - 20;
        b.out_size = rd32(&file[off + bsize - 4]);
        b.crc = rd32(&file[off + bsize - 8]);
        // ISIZE comes from the file: a BGZF block inflates to at most 64 KiB (SAM specification 4.1)
        if (b.out_size > 65536) { err = "corrupt BGZF block (ISIZE beyond 64 KiB)"; return false; }
        b.out_off = total;
        total += b.out_size;
        blocks.push_back(b);
        off += bsize;
        if (consumed) *consumed = off;
    }
    return true;
}

// CRC-32 of the gzip trailer, eight bytes per step (slicing-by-8: eight 256-entry tables; zlib 1.2.11's crc32 does
// four per step at 0.95 GB/s on this host, a quarter of the time its inflate takes for the same block)
struct Crc8 {
    uint32_t t[8][256];
    Crc8() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) t[k][i] = t[0][t[k - 1][i] & 0xFFu] ^ (t[k - 1][i] >> 8);
    }
};
uint32_t crc32_fast(const uint8_t *p, size_t n) {
    static const Crc8 tab;
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        std::memcpy(&lo, p, 4); std::memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = tab.t[7][lo & 0xFFu] ^ tab.t[6][(lo >> 8) & 0xFFu] ^ tab.t[5][(lo >> 16) & 0xFFu] ^ tab.t[4][lo >> 24] ^
            tab.t[3][hi & 0xFFu] ^ tab.t[2][(hi >> 8) & 0xFFu] ^ tab.t[1][(hi >> 16) & 0xFFu] ^ tab.t[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = tab.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// (the gzip trailer's CRC32 is checked like its ISIZE: htslib, behind pysam, refuses a block whose bytes do not match)
bool inflate_block(const uint8_t *src, size_t n, uint8_t *dst, size_t m, uint32_t crc) {
    if (m == 0) return crc == 0;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)n;
    zs.next_out = dst;
    zs.avail_out = (uInt)m;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0 && crc32_fast(dst, m) == crc;
}

template <class F>
void parallel_for(size_t n, int threads, F body) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    std::vector<std::thread> pool;
    std::atomic<size_t> next{0};
    const size_t chunk = std::max<size_t>(1, n / ((size_t)threads * 8));
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&]() {
            for (;;) {
                const size_t lo = next.fetch_add(chunk);
                if (lo >= n) break;
                const size_t hi = std::min(n, lo + chunk);
                for (size_t i = lo; i < hi; i++) body(i);
            }
        });
    for (auto &th : pool) th.join();
}

// A pool of worker threads that lives as long as its owner (the device decode hands a share of every slab's BGZF blocks to
// the host: starting a hundred threads per slab would cost what they save).  run(n, body): body(i) for i in [0, n), taken in
// ascending chunks; returns at once, wait() joins the job.
struct WorkerPool {
    // a job lives as long as somebody holds it: a thread that wakes up late finds the items of *its* job all taken and
    // goes back to sleep — nobody waits for threads, only for items
    struct Job {
        std::function<void(size_t)> body;
        size_t n = 0, chunk = 1;
        std::atomic<size_t> next{0}, done{0};
    };
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    std::shared_ptr<Job> job;
    size_t generation = 0;
    bool stop = false;
    explicit WorkerPool(int count) {
        for (int t = 0; t < count; t++)
            threads.emplace_back([this]() {
                size_t seen = 0;
                for (;;) {
                    std::shared_ptr<Job> j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || generation != seen; });
                        if (stop) return;
                        seen = generation;
                        j = job;
                    }
                    for (;;) {
                        const size_t lo = j->next.fetch_add(j->chunk);
                        if (lo >= j->n) break;
                        const size_t hi = std::min(j->n, lo + j->chunk);
                        for (size_t i = lo; i < hi; i++) j->body(i);
                        j->done.fetch_add(hi - lo, std::memory_order_release);
                    }
                }
            });
    }
    void run(size_t count, size_t chunk_, std::function<void(size_t)> f) {
        auto j = std::make_shared<Job>();
        j->body = std::move(f); j->n = count; j->chunk = chunk_ ? chunk_ : 1;
        std::lock_guard<std::mutex> lk(mu);
        job = j; generation++;
        cv.notify_all();
    }
    // every item of the last job done (the caller's thread helps itself to items meanwhile)
    void wait() {
        std::shared_ptr<Job> j;
        { std::lock_guard<std::mutex> lk(mu); j = job; }
        if (!j) return;
        for (;;) {
            const size_t lo = j->next.fetch_add(j->chunk);
            if (lo >= j->n) break;
            const size_t hi = std::min(j->n, lo + j->chunk);
            for (size_t i = lo; i < hi; i++) j->body(i);
            j->done.fetch_add(hi - lo, std::memory_order_release);
        }
        while (j->done.load(std::memory_order_acquire) < j->n) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
};

// BAM magic, header text and reference dictionary at the start of the uncompressed stream.  Returns 0 and the
// offset of the first record, 1 when `partial` and the header is not complete yet, -1 on a corrupt header.
int parse_header(mdx_bam *b, const uint8_t *data, size_t total, bool partial, size_t *first_record) {
    b->header_text.clear(); b->ref_names.clear(); b->ref_lengths.clear();
    if (total < 12) { if (partial) return 1; b->error = "not a BAM file"; return -1; }
    if (std::memcmp(data, "BAM\1", 4) != 0) { b->error = "not a BAM file"; return -1; }
    size_t off = 4;
    const int32_t l_text = rdi32(&data[off]);
    off += 4;
    if (l_text < 0) { b->error = "corrupt BAM header"; return -1; }
    if (off + (size_t)l_text + 4 > total) { if (partial) return 1; b->error = "corrupt BAM header"; return -1; }
    b->header_text.assign((const char *)&data[off], strnlen((const char *)&data[off], (size_t)l_text));
    off += l_text;
    const int32_t n_ref = rdi32(&data[off]);
    off += 4;
    for (int32_t i = 0; i < n_ref; i++) {
        if (off + 4 > total) { if (partial) return 1; b->error = "corrupt BAM header"; return -1; }
        const int32_t l_name = rdi32(&data[off]);
        if (l_name < 1) { b->error = "corrupt BAM header"; return -1; }
        if (off + 8 + (size_t)l_name > total) { if (partial) return 1; b->error = "corrupt BAM header"; return -1; }
        b->ref_names.emplace_back((const char *)&data[off + 4], (size_t)l_name - 1);
        b->ref_lengths.push_back(rdi32(&data[off + 4 + l_name]));
        off += 8 + l_name;
    }
    *first_record = off;
    return 0;
}

// One step of the record chain at `off`: 1 = a complete record (sizes returned), 0 = data ends inside it
// (or fewer than 4 bytes are left), -1 = corrupt.
// A CIGAR of more than 65 535 operations does not fit the record's 16-bit count: the SAM specification (section 4.2.2) stores
// it in a `CG:B,I` tag and leaves the placeholder `<l_seq>S<reference length>N` in the CIGAR field; htslib, behind the
// reference's pysam (read.cigar, align.py:76-88), puts the real operations back when it reads the record.  Returns the tag's
// operations (and their number) if record r — its fixed part, bs bytes — is such a record, else null.
inline const uint8_t *long_cigar(const uint8_t *r, size_t bs, uint32_t *count) {
    const uint32_t l_name = r[8], n_cig = rd16(r + 12);
    const int32_t l_seq = rdi32(r + 16);
    if (n_cig != 2 || l_seq < 0) return nullptr;
    const uint8_t *p = r + 32 + l_name;
    if (32 + (size_t)l_name + 8 > bs) return nullptr;
    if (rd32(p) != (((uint32_t)l_seq << 4) | 4u) || (rd32(p + 4) & 15u) != 3u) return nullptr;
    p += 8 + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    const uint8_t *end = r + bs;
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        if (ty == 'Z' || ty == 'H') {
            const uint8_t *z = (const uint8_t *)std::memchr(p, 0, (size_t)(end - p));
            if (!z) return nullptr;
            p = z + 1;
        } else if (ty == 'A' || ty == 'c' || ty == 'C') p += 1;
        else if (ty == 's' || ty == 'S') p += 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') p += 4;
        else if (ty == 'B') {
            if (p + 5 > end) return nullptr;
            const uint8_t sub = p[0];
            const uint32_t cnt = rd32(p + 1);
            const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if ((size_t)cnt * w > (size_t)(end - p - 5)) return nullptr;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') { *count = cnt; return p + 5; }
            p += 5 + (size_t)cnt * w;
        } else return nullptr;
    }
    return nullptr;
}

inline int record_at(const uint8_t *data, size_t off, size_t total, uint32_t *n_cig, uint32_t *l_seq, uint32_t *l_qname,
                     size_t *next) {
    if (off + 4 > total) return 0;
    const int32_t bs = rdi32(&data[off]);
    if (bs < 32) return -1;
    if (off + 4 + (size_t)bs > total) return 0;
    const uint8_t *r = &data[off + 4];
    const uint32_t l_name = r[8];
    const int32_t ls = rdi32(r + 16);
    *n_cig = rd16(r + 12);
    if (ls < 0 || 32 + l_name + 4 * (size_t)*n_cig + ((size_t)ls + 1) / 2 + (size_t)ls > (size_t)bs) return -1;
    *l_seq = (uint32_t)ls;
    *l_qname = l_name ? l_name - 1 : 0;
    *next = off + 4 + (size_t)bs;
    // (the real CIGAR of a record that keeps it in its CG tag)
    if (*n_cig == 2) { uint32_t cnt = 0; if (long_cigar(r, (size_t)bs, &cnt)) *n_cig = cnt; }
    return 1;
}

struct ScanSegment {
    size_t begin = 0, end = 0;       // the chain is followed from `begin` until it reaches or passes `end`
    size_t landed = 0;               // where it left the segment
    std::vector<size_t> rec;
    std::vector<uint32_t> n_cig, l_seq, l_qname;
    uint64_t cig = 0, seq = 0, name = 0;
    int state = 1;                   // as record_at at the point the segment stopped (1: ran to `end`)
};

// Records of data[off, total) -> the SoA columns of `b`.  `partial`: data may end inside a record; `consumed`
// is the offset of the first byte not unpacked.  `hints`: ascending offsets into `data` where a record probably
// starts (the first byte of each BGZF block: htslib closes a block early rather than split a record,
// bgzf_flush_try in bam_write1).  The record chain is a linked list, so the scan that sizes the columns is
// serial by nature (half of the decode time on a 64-thread host); with hints, segments of the chain are followed
// in parallel from hinted starts and accepted only if every segment's chain lands exactly on the start of the
// next one - otherwise the serial scan runs as if there had been no hints.  Same result either way.
template <class Lap>
int unpack_records(mdx_bam *b, const uint8_t *data, size_t off, size_t total, int threads, bool partial,
                   size_t *consumed, Lap lap, const std::vector<size_t> *hints = nullptr) {
    std::vector<size_t> rec;
    std::vector<uint32_t> coff{0}, soff{0}, noff{0};
    bool scanned = false;
    // MDX_BAM_PARALLEL_SCAN_MIN: smallest input (bytes) worth the parallel scan; tests set it to 0
    static const size_t scan_min = std::getenv("MDX_BAM_PARALLEL_SCAN_MIN")
                                       ? (size_t)std::strtoull(std::getenv("MDX_BAM_PARALLEL_SCAN_MIN"), nullptr, 10)
                                       : ((size_t)8 << 20);
    if (hints && threads > 1 && !hints->empty() && total - std::min(off, total) >= scan_min) {
        // pass 1, speculative: one segment per ~1/(4 threads) of the data, cut at hinted offsets
        const size_t want = (size_t)threads * 4, stride = (total - off) / want + 1;
        std::vector<ScanSegment> seg(1);
        seg[0].begin = off;
        for (auto it = std::upper_bound(hints->begin(), hints->end(), off); it != hints->end() && *it < total;) {
            seg.back().end = *it;
            seg.emplace_back();
            seg.back().begin = *it;
            it = std::lower_bound(it + 1, hints->end(), *it + stride);
        }
        seg.back().end = total;
        parallel_for(seg.size(), threads, [&](size_t i) {
            ScanSegment &g = seg[i];
            const size_t guess = (g.end - g.begin) / 128 + 16;
            g.rec.reserve(guess); g.n_cig.reserve(guess); g.l_seq.reserve(guess); g.l_qname.reserve(guess);
            size_t o = g.begin, next = 0;
            uint32_t nc = 0, ls = 0, ln = 0;
            while (o < g.end && (g.state = record_at(data, o, total, &nc, &ls, &ln, &next)) == 1) {
                g.rec.push_back(o + 4); g.n_cig.push_back(nc); g.l_seq.push_back(ls); g.l_qname.push_back(ln);
                g.cig += nc; g.seq += ls; g.name += ln;
                o = next;
            }
            g.landed = o;
        });
        bool good = true;
        for (size_t i = 0; i < seg.size() && good; i++) {
            const bool last = i + 1 == seg.size();
            if (!last) good = seg[i].state == 1 && seg[i].landed == seg[i + 1].begin;
            else good = seg[i].state == 1 || (seg[i].state == 0 && (partial || seg[i].landed + 4 > total));
        }
        uint64_t n_all = 0, cig = 0, sq = 0, nm = 0;
        std::vector<uint64_t> base(seg.size() * 4);
        for (size_t i = 0; i < seg.size(); i++) {
            base[4 * i] = n_all; base[4 * i + 1] = cig; base[4 * i + 2] = sq; base[4 * i + 3] = nm;
            n_all += seg[i].rec.size(); cig += seg[i].cig; sq += seg[i].seq; nm += seg[i].name;
        }
        if (good && sq <= 0xFFFFFFFFull && cig <= 0xFFFFFFFFull && nm <= 0xFFFFFFFFull) {
            rec.resize(n_all); coff.resize(n_all + 1); soff.resize(n_all + 1); noff.resize(n_all + 1);
            parallel_for(seg.size(), threads, [&](size_t i) {
                const ScanSegment &g = seg[i];
                size_t k = base[4 * i];
                uint32_t c = (uint32_t)base[4 * i + 1], q = (uint32_t)base[4 * i + 2], m = (uint32_t)base[4 * i + 3];
                for (size_t j = 0; j < g.rec.size(); j++, k++) {
                    rec[k] = g.rec[j];
                    c += g.n_cig[j]; q += g.l_seq[j]; m += g.l_qname[j];
                    coff[k + 1] = c; soff[k + 1] = q; noff[k + 1] = m;
                }
            });
            off = seg.back().landed;
            scanned = true;
        }
    }
    // pass 1 (sequential): record starts and the prefix sums that size the ragged columns
    while (!scanned) {
        uint32_t n_cig = 0, l_seq = 0, l_qname = 0;
        size_t next = 0;
        const int state = record_at(data, off, total, &n_cig, &l_seq, &l_qname, &next);
        if (state < 0 || (state == 0 && !partial && off + 4 <= total)) { b->error = "corrupt BAM record"; return MDX_ERR_ARG; }
        if (state == 0) break;       // the record continues in the next chunk (or fewer than 4 bytes are left)
        const uint64_t ns = (uint64_t)soff.back() + (uint64_t)l_seq;
        if (ns > 0xFFFFFFFFull) { b->error = "more than 4 Gbases in one file: split it"; return MDX_ERR_ARG; }
        rec.push_back(off + 4);
        coff.push_back(coff.back() + n_cig);
        soff.push_back((uint32_t)ns);
        noff.push_back(noff.back() + l_qname);
        off = next;
    }
    const size_t n = rec.size();
    *consumed = off;
    lap(scanned ? "scan (par.)" : "record scan");
    b->flag.resize(n); b->lib.assign(n, 0); b->tid.resize(n); b->pos.resize(n); b->tlen.resize(n);
    b->mtid.resize(n); b->mpos.resize(n); b->rg_index.assign(n, -1); b->has_mr.assign(n, 0); b->qmin.resize(n);
    b->cigar_off = coff; b->seq_off = soff; b->qname_off = noff;
    b->cigar.resize(coff.back()); b->seq.resize((size_t)soff.back() + 64); b->qual.resize((size_t)soff.back() + 64);
    b->qnames.resize(noff.back());
    // read-group ids -> small integers (header order is resolved by the caller): a thread looks its last
    // id up first, the shared map only when the id changes
    std::mutex rg_mu;
    std::unordered_map<std::string, int32_t> rg_map;
    lap("allocate");
    static const char DEC[17] = "=ACMGRSVTWYHKDBN";
    // pass 2 (parallel): unpack every record into the columns
    parallel_for(n, threads, [&](size_t i) {
        const uint8_t *r = &data[rec[i]];
        const size_t bs = (size_t)rdi32(r - 4);
        const uint32_t l_name = r[8], n_cig = rd16(r + 12);
        const int32_t l_seq = rdi32(r + 16);
        b->tid[i] = rdi32(r); b->pos[i] = rdi32(r + 4); b->flag[i] = (uint16_t)(rd16(r + 14) & 0x3FFFu);   // bits 14, 15: MDX_FLAG_HAS_QUAL / _QUAL_ABOVE_MIN, hints, never the file's
        b->mtid[i] = rdi32(r + 20); b->mpos[i] = rdi32(r + 24); b->tlen[i] = rdi32(r + 28);
        const uint8_t *p = r + 32;
        if (l_name) std::memcpy(&b->qnames[noff[i]], p, l_name - 1);
        p += l_name;
        {
            uint32_t n_long = 0;
            const uint8_t *cg = n_cig == 2 ? long_cigar(r, bs, &n_long) : nullptr;
            if (cg) for (uint32_t k = 0; k < n_long; k++) b->cigar[coff[i] + k] = rd32(cg + 4 * k);
            else for (uint32_t k = 0; k < n_cig; k++) b->cigar[coff[i] + k] = rd32(p + 4 * k);
        }
        p += 4 * (size_t)n_cig;
        uint8_t *s = &b->seq[soff[i]];
        for (int32_t k = 0; k < l_seq; k++) {
            const uint8_t byte = p[k >> 1];
            s[k] = (uint8_t)DEC[(k & 1) ? (byte & 15) : (byte >> 4)];
        }
        p += ((size_t)l_seq + 1) / 2;
        std::memcpy(&b->qual[soff[i]], p, (size_t)l_seq);
        {
            uint8_t lowest = 0xFF;
            for (int32_t k = 0; k < l_seq; k++) lowest = p[k] < lowest ? p[k] : lowest;
            b->qmin[i] = lowest;
        }
        p += l_seq;
        const uint8_t *end = r + bs;
        while (p + 3 <= end) {
            const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
            p += 3;
            if (t0 == 'M' && t1 == 'R') b->has_mr[i] = 1;
            if (ty == 'Z' || ty == 'H') {
                const uint8_t *z = (const uint8_t *)std::memchr(p, 0, (size_t)(end - p));
                if (!z) break;
                if (t0 == 'R' && t1 == 'G' && ty == 'Z') {
                    static thread_local const mdx_bam *owner = nullptr;
                    static thread_local std::string last;
                    static thread_local int32_t last_id = -1;
                    const size_t len = (size_t)(z - p);
                    if (owner != b || last.size() != len || std::memcmp(last.data(), p, len) != 0) {
                        last.assign((const char *)p, len);
                        owner = b;
                        std::lock_guard<std::mutex> guard(rg_mu);
                        auto it = rg_map.find(last);
                        if (it == rg_map.end()) {
                            it = rg_map.emplace(last, (int32_t)b->rg_names.size()).first;
                            b->rg_names.push_back(last);
                        }
                        last_id = it->second;
                    }
                    b->rg_index[i] = last_id;
                }
                p = z + 1;
            } else if (ty == 'A' || ty == 'c' || ty == 'C') p += 1;
            else if (ty == 's' || ty == 'S') p += 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') p += 4;
            else if (ty == 'B') {
                if (p + 5 > end) break;
                const uint8_t sub = p[0];
                const uint32_t cnt = rd32(p + 1);
                const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                p += 5 + (size_t)cnt * w;
            } else break;
        }
    });
    lap("unpack");
    return MDX_OK;
}

// Reads about `want` compressed bytes, inflates every complete block among them and appends the result to
// s->pending.  Returns false on an I/O or format error (text in s->head.error).
bool stream_fill(mdx_bam_stream *s, size_t want) {
    if (s->eof) return true;
    std::vector<Block> blocks;
    size_t total = 0, consumed = s->coff;
    if (!scan_blocks(*s->file, blocks, total, s->head.error, s->coff, want, &consumed)) return false;
    const size_t base = s->pending.size();
    s->pending.resize(base + total);
    std::atomic<bool> ok{true};
    const MappedFile &file = *s->file;
    parallel_for(blocks.size(), s->threads, [&](size_t i) {
        const Block &k = blocks[i];
        if (!inflate_block(&file[k.in_off], k.in_size, &s->pending[base + k.out_off], k.out_size, k.crc)) ok = false;
    });
    if (!ok) { s->head.error = "inflate failed (corrupt BGZF block: DEFLATE stream, ISIZE or CRC32)"; return false; }
    for (const Block &k : blocks) s->hints.push_back(base + k.out_off);
    s->coff = consumed;
    if (s->coff >= file.size()) s->eof = true;
    return true;
}

}  // namespace

extern "C" {

int mdx_bam_read(const char *path, int threads, mdx_bam **out) {
    try {
        if (!path || !out) return MDX_ERR_ARG;
        mdx_bam *b = new (std::nothrow) mdx_bam();
        if (!b) return MDX_ERR_ARG;
        *out = b;
        // MDX_BAM_TIMING=1: stage times on stderr
        const bool timing = std::getenv("MDX_BAM_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "mdx_bam_read %-12s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        MappedFile file;
        if (!file.open(path, b->error)) return MDX_ERR_ARG;
        lap("file map");
        std::vector<Block> blocks;
        size_t total = 0;
        if (!scan_blocks(file, blocks, total, b->error)) return MDX_ERR_ARG;
        lap("block scan");
        raw_bytes data(total + 8);
        std::atomic<bool> ok{true};
        parallel_for(blocks.size(), threads, [&](size_t i) {
            const Block &k = blocks[i];
            if (!inflate_block(&file[k.in_off], k.in_size, &data[k.out_off], k.out_size, k.crc)) ok = false;
        });
        if (!ok) { b->error = "inflate failed (corrupt BGZF block: DEFLATE stream, ISIZE or CRC32)"; return MDX_ERR_ARG; }
        lap("inflate");
        file.close();
    
        size_t off = 0, used = 0;
        if (parse_header(b, data.data(), total, false, &off) != 0) return MDX_ERR_ARG;
        std::vector<size_t> hints(blocks.size());
        for (size_t i = 0; i < blocks.size(); i++) hints[i] = blocks[i].out_off;
        const int rc = unpack_records(b, data.data(), off, total, threads, false, &used, lap, &hints);
        b->reaper = std::thread([](raw_bytes d) { raw_bytes().swap(d); }, std::move(data));
        lap("release");
        return rc;
    
    } catch (const std::exception &e) {
        if (out && *out) (*out)->error = std::string("mdx_bam_read: ") + e.what();
        return MDX_ERR_ARG;
    } catch (...) {
        if (out && *out) (*out)->error = "mdx_bam_read: unknown failure";
        return MDX_ERR_ARG;
    }
}

void mdx_bam_free(mdx_bam *b) {
    if (!b) return;
    // a large handle (its columns are hundreds of megabytes) is torn down off the caller's thread: in the chunked
    // pipeline this call sits between two tabulations
    if (b->seq.size() >= ((size_t)64 << 20)) std::thread([b]() { delete b; }).detach();
    else delete b;
}

const char *mdx_bam_error(const mdx_bam *b) { return b ? b->error.c_str() : "null handle"; }

const char *mdx_bam_header_text(const mdx_bam *b) { return b ? b->header_text.c_str() : ""; }

int32_t mdx_bam_n_ref(const mdx_bam *b) { return b ? (int32_t)b->ref_names.size() : 0; }

const char *mdx_bam_ref_name(const mdx_bam *b, int32_t i) {
    return (b && i >= 0 && (size_t)i < b->ref_names.size()) ? b->ref_names[i].c_str() : "";
}

int64_t mdx_bam_ref_length(const mdx_bam *b, int32_t i) {
    return (b && i >= 0 && (size_t)i < b->ref_lengths.size()) ? b->ref_lengths[i] : -1;
}

int mdx_bam_batch(const mdx_bam *b, mdx_batch *view, const int32_t **mtid, const int32_t **mpos,
                  const int32_t **rg_index, const uint8_t **has_mr) {
    if (!b || !view) return MDX_ERR_ARG;
    view->n_reads = (int64_t)b->flag.size();
    view->n_cigar = (int64_t)b->cigar.size();
    view->n_bases = b->seq_off.empty() ? 0 : (int64_t)b->seq_off.back();
    view->flag = b->flag.data(); view->lib = b->lib.data(); view->tid = b->tid.data(); view->pos = b->pos.data();
    view->tlen = b->tlen.data(); view->cigar_off = b->cigar_off.data(); view->cigar = b->cigar.data();
    view->seq_off = b->seq_off.data(); view->seq = b->seq.data(); view->qual = b->qual.data();
    view->seq_format = MDX_SEQ_ASCII; view->reserved = 0; view->lowq = nullptr; view->libsort = nullptr;
    if (mtid) *mtid = b->mtid.data();
    if (mpos) *mpos = b->mpos.data();
    if (rg_index) *rg_index = b->rg_index.data();
    if (has_mr) *has_mr = b->has_mr.data();
    return MDX_OK;
}

const uint8_t *mdx_bam_qmin(const mdx_bam *b) { return b ? b->qmin.data() : nullptr; }

int32_t mdx_bam_n_rg(const mdx_bam *b) { return b ? (int32_t)b->rg_names.size() : 0; }

const char *mdx_bam_rg_name(const mdx_bam *b, int32_t i) {
    return (b && i >= 0 && (size_t)i < b->rg_names.size()) ? b->rg_names[i].c_str() : "";
}

const char *mdx_bam_qnames(const mdx_bam *b, const uint32_t **offsets) {
    if (!b) return "";
    if (offsets) *offsets = b->qname_off.data();
    return b->qnames.data();
}

int mdx_bam_open(const char *path, int threads, mdx_bam_stream **out) {
    try {
        if (!path || !out) return MDX_ERR_ARG;
        mdx_bam_stream *s = new (std::nothrow) mdx_bam_stream();
        if (!s) return MDX_ERR_ARG;
        *out = s;
        s->threads = threads < 1 ? 1 : threads;
        s->file = new (std::nothrow) MappedFile();
        if (!s->file || !s->file->open(path, s->head.error)) return MDX_ERR_ARG;
        if (s->file->size() == 0) s->eof = true;
        // the header may span several blocks: inflate until it parses (one block's worth first — the header of most files
        // — then a megabyte at a time: the device decode path opens the file for its header alone, and a megabyte of blocks
        // inflated for nothing was 2 of its 2.5 ms)
        for (size_t want = 65536;; want = (size_t)1 << 20) {
            if (!stream_fill(s, want)) return MDX_ERR_ARG;
            size_t first = 0;
            const int rc = parse_header(&s->head, s->pending.data(), s->pending.size(), !s->eof, &first);
            if (rc < 0) return MDX_ERR_ARG;
            if (rc == 0) {
                s->header_bytes = first;
                s->pending.erase(s->pending.begin(), s->pending.begin() + (std::ptrdiff_t)first);
                size_t kept = 0;
                for (size_t h : s->hints) if (h >= first) s->hints[kept++] = h - first;
                s->hints.resize(kept);
                return MDX_OK;
            }
        }
    
    } catch (const std::exception &e) {
        if (out && *out) (*out)->head.error = std::string("mdx_bam_open: ") + e.what();
        return MDX_ERR_ARG;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

const mdx_bam *mdx_bam_stream_header(const mdx_bam_stream *s) { return s ? &s->head : nullptr; }

int mdx_bam_next(mdx_bam_stream *s, int64_t chunk_bytes, mdx_bam **out) {
    try {
        if (!s || !out || !s->file) return MDX_ERR_ARG;
        *out = nullptr;
        size_t limit = chunk_bytes < 64 ? 64 : (size_t)chunk_bytes;       // uncompressed BAM bytes per chunk
        const bool timing = std::getenv("MDX_BAM_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "mdx_bam_next %-12s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        for (;;) {
            // BGZF members hold at most 64 KiB each; BAM compresses about 3-4x
            while (!s->eof && s->pending.size() < limit + s->skip)
                if (!stream_fill(s, std::max<size_t>(limit / 4, (size_t)1 << 16))) return MDX_ERR_ARG;
            if (s->skip) {
                // (mdx_bam_seek: the tail of a record that began in front of the block the stream was pointed at)
                if (s->pending.size() < s->skip) { s->head.error = "mdx_bam_seek: offset beyond the data"; return MDX_ERR_ARG; }
                s->pending.erase(s->pending.begin(), s->pending.begin() + (std::ptrdiff_t)s->skip);
                size_t kept = 0;
                for (size_t h : s->hints) if (h >= s->skip) s->hints[kept++] = h - s->skip;
                s->hints.resize(kept);
                s->skip = 0;
            }
            lap("inflate");
            const size_t total = std::min(limit, s->pending.size());
            const bool partial = !(s->eof && total == s->pending.size());
            if (!partial && total < 4) return MDX_OK;                      // end of file: *out stays NULL
            mdx_bam *b = new (std::nothrow) mdx_bam();
            if (!b) return MDX_ERR_ARG;
            b->header_text = s->head.header_text;
            b->ref_names = s->head.ref_names;
            b->ref_lengths = s->head.ref_lengths;
            size_t used = 0;
            const int rc = unpack_records(b, s->pending.data(), 0, total, s->threads, partial, &used, lap, &s->hints);
            if (rc != MDX_OK) { s->head.error = b->error; delete b; return rc; }
            if (s->keep_raw && !b->flag.empty()) {
                b->raw.assign(s->pending.begin(), s->pending.begin() + (std::ptrdiff_t)used);
                b->rec_off.resize(b->flag.size() + 1);
                size_t o = 0;
                for (size_t i = 0; i < b->flag.size(); i++) { b->rec_off[i] = o; o += 4 + (size_t)rd32(&b->raw[o]); }
                b->rec_off[b->flag.size()] = o;
            }
            s->pending.erase(s->pending.begin(), s->pending.begin() + (std::ptrdiff_t)used);
            {
                size_t kept = 0;
                for (size_t h : s->hints) if (h >= used) s->hints[kept++] = h - used;
                s->hints.resize(kept);
            }
            lap("carry");
            if (!b->flag.empty()) { *out = b; return MDX_OK; }
            delete b;
            if (!partial) return MDX_OK;
            limit *= 2;                                                    // a record larger than the chunk: widen
        }
    
    } catch (const std::exception &e) {
        if (s) s->head.error = std::string("mdx_bam_next: ") + e.what();
        return MDX_ERR_ARG;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

int mdx_bam_seek(mdx_bam_stream *s, int64_t comp_off, int64_t phase) {
    try {
        if (!s || !s->file || comp_off < 0 || phase < 0 || (size_t)comp_off > s->file->size()) return MDX_ERR_ARG;
        s->pending.clear();
        s->hints.clear();
        s->coff = (size_t)comp_off;
        s->eof = s->coff >= s->file->size();
        s->skip = (size_t)phase;
        return MDX_OK;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

int mdx_bam_stream_keep_raw(mdx_bam_stream *s, int on) {
    if (!s) return MDX_ERR_ARG;
    s->keep_raw = on != 0;
    return MDX_OK;
}

int mdx_bam_raw(const mdx_bam *b, const uint8_t **data, const uint64_t **rec_off) {
    if (!b || b->rec_off.empty()) return MDX_ERR_STATE;
    if (data) *data = b->raw.data();
    if (rec_off) *rec_off = b->rec_off.data();
    return MDX_OK;
}

int mdx_bam_patch_rescaled(const mdx_bam *b, const uint8_t *qual_out, const float *mr, const uint8_t *rescaled,
                           uint8_t *out, int64_t out_cap, int64_t *out_len) {
    if (!b || b->rec_off.empty() || !qual_out || !mr || !rescaled || !out || !out_len) return MDX_ERR_ARG;
    const size_t n = b->flag.size();
    size_t o = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t r0 = (size_t)b->rec_off[i], sz = (size_t)b->rec_off[i + 1] - r0;     // 4 + block_size
        const size_t need = sz + (rescaled[i] ? 7 : 0);
        if ((int64_t)(o + need) > out_cap) return MDX_ERR_ARG;
        std::memcpy(out + o, &b->raw[r0], sz);
        if (rescaled[i]) {
            const uint8_t *rec = &b->raw[r0];
            const size_t l_read_name = rec[12], n_cigar = rd16(rec + 16), l_seq = rd32(rec + 20);
            const size_t qoff = 4 + 32 + l_read_name + 4 * n_cigar + (l_seq + 1) / 2;
            if (qoff + l_seq > sz) return MDX_ERR_ARG;
            std::memcpy(out + o + qoff, qual_out + b->seq_off[i], l_seq);
            out[o + sz] = 'M'; out[o + sz + 1] = 'R'; out[o + sz + 2] = 'f';
            std::memcpy(out + o + sz + 3, &mr[i], 4);
            const uint32_t bs = (uint32_t)(sz - 4 + 7);
            std::memcpy(out + o, &bs, 4);
        }
        o += need;
    }
    *out_len = (int64_t)o;
    return MDX_OK;
}

// float("%.5f" % x) of rescale.py:275-276, element by element: the MR sum printed with five decimals and read back, then the
// 32-bit value an MR:f tag holds.  (Python formats with correctly rounded decimal conversion in both directions; so does
// glibc.)  n values on `threads` threads; NaN (a record written back unchanged) -> 0.
int mdx_mr_round(const double *mr_raw, int64_t n, float *out, int32_t threads) {
    if (n < 0 || (n > 0 && (!mr_raw || !out))) return MDX_ERR_ARG;
    try {
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : 1, n / 4096 + 1));
        auto work = [&](int64_t lo, int64_t hi) {
            char buf[512];
            for (int64_t i = lo; i < hi; i++) {
                const double x = mr_raw[i];
                if (x != x) { out[i] = 0.f; continue; }
                std::snprintf(buf, sizeof buf, "%.5f", x);
                out[i] = (float)std::strtod(buf, nullptr);
            }
        };
        if (nt == 1) { work(0, n); return MDX_OK; }
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto &th : pool) th.join();
        return MDX_OK;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

void mdx_bam_close(mdx_bam_stream *s) {
    if (!s) return;
    delete s->file;
    delete s;
}


#ifndef MDX_HOST_ONLY
namespace {
// Threads this process may keep busy inflating: half of the hardware threads, 128 at most — and not more than the CPU time
// the control group grants (cpu.max: "quota period"; MDX_CPU_MAX_FILE names another file, for the tests): threads beyond the
// quota use it up in a fraction of the period and then the whole process stands still for the rest of it (a pod with 256
// hardware threads and 16 CPUs' worth of quota: slabs took 60 ms now and then instead of 5 with 128 threads) — divided by the
// ranks of this node (LOCAL_WORLD_SIZE, as torchrun sets it; SURVEY 8e: one process per GPU, and all of them inflate at the
// same time: eight ranks of 14 threads each on a quota of 16 CPUs are that stall again).  MDX_GBAM_HOST_THREADS overrides.
int host_thread_budget() {
    const unsigned hc = std::thread::hardware_concurrency();
    int want_threads = (int)(hc > 8 ? hc / 2 : (hc ? hc : 4));
    if (want_threads > 128) want_threads = 128;
    const char *cpu_max = std::getenv("MDX_CPU_MAX_FILE");
    if (FILE *fh = std::fopen(cpu_max && *cpu_max ? cpu_max : "/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long period = 0;
        if (std::fscanf(fh, "%63s %ld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
            const long cpus = std::atol(q) / period;
            if (cpus >= 1 && want_threads > (int)cpus - 2) want_threads = (int)std::max<long>(1, cpus - 2);
        }
        std::fclose(fh);
    }
    if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int ranks = std::atoi(e);
        if (ranks > 1) want_threads = std::max(1, want_threads / ranks);
    }
    if (const char *e = std::getenv("MDX_GBAM_HOST_THREADS")) want_threads = std::max(1, std::atoi(e));
    return want_threads;
}
// the process's pool of inflating threads (host_thread_budget() of them, counted when the pool starts) and the buffer they
// inflate into: both outlive a file — a hundred threads take milliseconds to start and to join, and a buffer of a few hundred
// megabytes as long to fault in
WorkerPool *host_pool() {
    static std::mutex mu;
    static std::unique_ptr<WorkerPool> pool;
    std::lock_guard<std::mutex> lk(mu);
    if (!pool) pool.reset(new WorkerPool(host_thread_budget()));
    return pool.get();
}
// (pinned: a copy out of pageable memory has the runtime pin and unpin the pages it reads, under the address space's lock
// — with a hundred threads faulting pages in at the same time, slabs took 50 ms now and then instead of 5)
std::mutex g_hbuf_mu;
uint8_t *g_hbuf = nullptr;
size_t g_hbuf_cap = 0;
std::atomic<double> g_host_share{0.12};      // what the last file's slabs settled on: where the next file starts
void host_buffer_take(uint8_t *&p, size_t &cap) {
    std::lock_guard<std::mutex> lk(g_hbuf_mu);
    p = g_hbuf; cap = g_hbuf_cap; g_hbuf = nullptr; g_hbuf_cap = 0;
}
void host_buffer_give(uint8_t *&p, size_t &cap) {
    std::lock_guard<std::mutex> lk(g_hbuf_mu);
    if (p && cap > g_hbuf_cap) { std::swap(p, g_hbuf); std::swap(cap, g_hbuf_cap); }
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
}
// The read-group tables of a handle (names, offsets, libraries) in one small device allocation that outlives the handle:
// three hipMalloc / hipMemcpy pairs at configure and three hipFree at close were 4 ms of an 8 M-record file's 63.
struct RgBuf { int device; void *p; size_t cap; };
std::mutex g_rgbuf_mu;
std::vector<RgBuf> g_rgbufs;
void *rg_buffer_take(int device, size_t need, size_t &cap) {
    {
        std::lock_guard<std::mutex> lk(g_rgbuf_mu);
        for (size_t i = 0; i < g_rgbufs.size(); i++)
            if (g_rgbufs[i].device == device && g_rgbufs[i].cap >= need) {
                void *p = g_rgbufs[i].p; cap = g_rgbufs[i].cap;
                g_rgbufs.erase(g_rgbufs.begin() + (long)i);
                return p;
            }
    }
    void *p = nullptr;
    cap = std::max(need, (size_t)64 << 10);
    if (hipMalloc(&p, cap) != hipSuccess) { cap = 0; return nullptr; }
    return p;
}
// ... and the three buffers a handle sends the next slab's compressed bytes ahead into (mdx_gbam::pf_buf)
std::vector<RgBuf> g_pfbufs;
void pf_buffer_take(int device, size_t need, void *&p, size_t &cap) {
    std::lock_guard<std::mutex> lk(g_rgbuf_mu);
    for (size_t i = 0; i < g_pfbufs.size(); i++)
        if (g_pfbufs[i].device == device && g_pfbufs[i].cap >= need) {
            p = g_pfbufs[i].p; cap = g_pfbufs[i].cap;
            g_pfbufs.erase(g_pfbufs.begin() + (long)i);
            return;
        }
}
void pf_buffer_give(int device, void *p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_rgbuf_mu);
    if (g_pfbufs.size() < 6) g_pfbufs.push_back(RgBuf{device, p, cap});
    else (void)hipFree(p);
}
// ... and a handle's second arena (the first one stays with its context): the largest spare one of the device
std::vector<RgBuf> g_arenas;
void arena_take(int device, void *&p, size_t &cap) {
    std::lock_guard<std::mutex> lk(g_rgbuf_mu);
    long best = -1;
    for (size_t i = 0; i < g_arenas.size(); i++)
        if (g_arenas[i].device == device && (best < 0 || g_arenas[i].cap > g_arenas[(size_t)best].cap)) best = (long)i;
    if (best < 0) return;
    p = g_arenas[(size_t)best].p; cap = g_arenas[(size_t)best].cap;
    g_arenas.erase(g_arenas.begin() + best);
}
void arena_give(int device, void *p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_rgbuf_mu);
    if (g_arenas.size() < 2) g_arenas.push_back(RgBuf{device, p, cap});
    else (void)hipFree(p);
}
// ... and the 64 pinned bytes a handle's CRC verdict lands in
std::vector<int *> g_pins;
int *pin_take() {
    {
        std::lock_guard<std::mutex> lk(g_rgbuf_mu);
        if (!g_pins.empty()) { int *p = g_pins.back(); g_pins.pop_back(); return p; }
    }
    int *p = nullptr;
    if (hipHostMalloc((void **)&p, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pin_give(int *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_rgbuf_mu);
    if (g_pins.size() < 16) g_pins.push_back(p);
    else (void)hipHostFree(p);
}
void rg_buffer_give(int device, void *p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_rgbuf_mu);
    if (g_rgbufs.size() < 16) g_rgbufs.push_back(RgBuf{device, p, cap});
    else (void)hipFree(p);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// GPU-side decode (mdx_gbam.hip): the compressed file goes to HBM a slab of BGZF blocks at a time, is inflated and
// unpacked there, and the batch columns never exist on the host.
struct mdx_gbam {
    mdx_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
    mdx_bam_stream *hs = nullptr;        // header (parsed on the host) and the mapped file
    std::vector<Block> blocks;           // the BGZF blocks found so far (the file is scanned as the slabs need it:
                                         // walking the block headers of a large file touches every page of it once)
    size_t scanned = 0, scanned_out = 0; // compressed offset behind the last block found, inflated bytes in front of it
    // blocks up to compressed offset `upto` (or the end of the file); false: a corrupt block header
    bool scan_to(size_t upto) {
        const MappedFile &f = *hs->file;
        while (scanned < f.size() && scanned < upto) {
            std::vector<Block> more;
            size_t total = 0, consumed = scanned;
            if (!scan_blocks(f, more, total, error, scanned, std::max<size_t>(upto - scanned, (size_t)1 << 20), &consumed)) return false;
            if (consumed == scanned) break;
            for (Block &b : more) { b.out_off += scanned_out; blocks.push_back(b); }
            scanned = consumed; scanned_out += total;
        }
        return true;
    }
    bool whole_file_scanned() const { return scanned >= hs->file->size(); }
    size_t next_block = 0;               // first block not decoded yet
    // Records may straddle BGZF blocks and slabs: a slab's batch holds the records that START in it (the blocks behind it
    // are inflated as far as its last record reaches), and `phase` is where the first record of the next slab starts,
    // counted from that slab's first inflated byte; unknown (phase_known false) behind mdx_gbam_skip, where the device
    // scan's guess stands in — and next_verified says whether that guess, seen from this side, is the true offset.
    size_t phase = 0;
    bool phase_known = true, next_verified = true;
    int fixups = 0;                      // segments whose guessed first record the chain did not confirm (rescanned)
    int slabs_done = 0;
    int64_t view_reads = 0;          // records of the view mdx_gbam_next handed out last (mdx_gbam_view_flags)
    // The host's share of a slab's inflate (round 4): the device inflater is bound by instruction issue — more wavefronts
    // do not help it — while the host's cores idle: the last `host_share` of a slab's inflated bytes are inflated by a
    // pool of host threads into `hbuf` and copied to their place in HBM on a stream of their own, in pieces, under the
    // device's inflate of the rest (whose compressed bytes alone are uploaded).  Adjusted slab by slab to whoever
    // finished first; MDX_GBAM_HOST_SHARE in the environment fixes it (0: the device inflates everything).
    double host_share = 0.12;
    bool host_share_fixed = false;
    WorkerPool *pool = nullptr;          // (the process's: host_pool())
    uint8_t *hbuf = nullptr;             // pinned host memory (the process's: host_buffer_take / _give)
    size_t hbuf_cap = 0;
    hipStream_t copy_stream = nullptr;
    // Round 6: the slabs in a pipeline of two.  A slab's life has two halves: A — its compressed bytes in HBM, the inflate
    // launched on a stream of its own (infl_stream), the host's share inflated and copied beside it — and B — CRC, record
    // scan, the chain walk on the host, unpack (the context's stream), the view handed out, the caller's tabulation.  A call
    // does A of the slab behind the one it hands out before it does B of that one: the device goes from one slab's inflate
    // straight into the next one's (they queue on infl_stream) while B and the tabulation of the slab in front run beside it
    // on the context's stream, and the host's threads are never idle between two slabs.  Two arenas, used in turns: A of slab
    // k + 2 writes where slab k lived — behind an event that says the caller's tabulation of slab k is done (ev_free).
    // MDX_GBAM_NO_LOOKAHEAD=1: one slab at a time (A/B runs; MDX_BAM_TIMING and a handle that skips slabs — a run over several
    // GPUs, mdx_gbam_skip — do the same).
    hipStream_t infl_stream = nullptr;
    hipEvent_t ev_free = nullptr;
    bool no_ahead = false;
    // the CRC of the device-inflated blocks runs on the copy stream, under the scan, the chain walk and the unpack of the same
    // slab; pin_bad = where its verdict lands (pinned: an asynchronous copy)
    int *pin_bad = nullptr;
    std::string error;
    bool want_qual = false, want_mate = false;
    int minqual = 0;                     // --min-basequal on the device path (mdx_gbam_set_min_basequal)
    int seq_format = MDX_SEQ_ASCII;      // form of the seq column handed out (mdx_gbam_set_seq_format)
    bool no_qual_seen = false;           // a counted record without qualities has come by
    // read groups
    std::vector<uint8_t> rg_names;
    std::vector<uint32_t> rg_off;
    std::vector<int32_t> lib_of_rg;
    int lib_default = -1;
    void *d_rg_names = nullptr, *d_rg_off = nullptr, *d_lib_of_rg = nullptr;    // (parts of d_rg: rg_buffer_take)
    void *d_rg = nullptr;
    size_t d_rg_cap = 0;
    // The next slab's compressed bytes, uploaded under this slab's inflate (the device's part of them: the host's share
    // reads the file itself) into one of three buffers of the handle's own — two slabs' inflates may be reading theirs while
    // the third fills; pf_cur = the buffer that holds bytes [pf_in0, pf_in0 + pf_bytes) of the file (-1: none); a slab notes
    // the one its inflate reads (Slab::pf_used; -1: its arena's).  MDX_GBAM_NO_PREFETCH=1: off (A/B runs).
    void *pf_buf[3] = {nullptr, nullptr, nullptr};
    size_t pf_cap[3] = {0, 0, 0};
    int pf_cur = -1;
    size_t pf_in0 = 0, pf_bytes = 0;
    void *d_crc_tables = nullptr;        // mdx_crc32::Tables
    struct Buf { void *p = nullptr; size_t cap = 0; };
    // one slab in flight: what half A leaves for half B, and its device buffers (all but `arena` point into it)
    struct Slab {
        bool ready = false;              // half A done for the slab that starts at block b0 with this chunk_bytes / ahead
        int64_t chunk_bytes = 0;
        size_t b0 = 0, b1 = 0, b2 = 0;   // the slab's blocks [b0, b1), inflated [b0, b2)
        size_t slab_bytes = 0, unc_bytes = 0, out0 = 0, in0 = 0, nh = 0, ahead = 0;
        size_t rec_cap = 0, cig_cap = 0, seq_cap = 0;
        bool more_file = false;
        std::vector<uint32_t> blk, crcs;
        int pf_used = -1;
        hipEvent_t ev_infl0 = nullptr, ev_infl = nullptr;   // around the device's inflate (timed: the host's share is set by them)
        double t_host = 0, frac_host = 0;                    // the host's share of this slab: milliseconds, fraction of the bytes
        bool timed = false;
        Buf comp, blk_d, crc, status, unc, cnt, pre, rec_off, flag, lib, tid, pos, tlen, mtid, mpos,
            cigar_off, cigar, seq_off, seq, qual, small, info, forced, arena;
    } slab[2];
    int cur = 0;                         // the slab the next call hands out
    int view_slab = 0;                   // ... and the one the last call did (mdx_gbam_view_flags)
    mdx_batch last_view{};               // ... its view (mdx_gbam_rescale_slab)
    const int32_t *last_mtid = nullptr, *last_mpos = nullptr;
    bool reserve(Buf &b, size_t bytes) {
        if (bytes <= b.cap) return true;
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&b.p, want) != hipSuccess) { error = "out of device memory"; return false; }
        b.cap = want;
        return true;
    }
    // everything enqueued by this handle has run (the context's stream too)
    bool drain() {
        bool ok = true;
        if (infl_stream) ok = hipStreamSynchronize(infl_stream) == hipSuccess && ok;
        if (copy_stream) ok = hipStreamSynchronize(copy_stream) == hipSuccess && ok;
        if (stream) ok = hipStreamSynchronize(stream) == hipSuccess && ok;
        return ok;
    }
};

int mdx_gbam_open(mdx_ctx *ctx, const char *path, mdx_gbam **out) {
    try {
        if (!ctx || !path || !out) return MDX_ERR_ARG;
        mdx_gbam *g = new (std::nothrow) mdx_gbam();
        if (!g) return MDX_ERR_ARG;
        *out = g;
        g->ctx = ctx;
        void *st = nullptr;
        if (mdx_ctx_stream(ctx, &st, &g->device) != MDX_OK) { g->error = "no context"; return MDX_ERR_ARG; }
        g->stream = (hipStream_t)st;
        int rc = mdx_bam_open(path, 4, &g->hs);
        if (rc != MDX_OK) { g->error = g->hs ? g->hs->head.error : "cannot open"; return rc; }
        // the first slab starts with the block that holds the first record (htslib flushes the header into blocks of its
        // own; other writers let it share a block with records): `phase` bytes into it
        size_t k = 0;
        for (;;) {
            while (k < g->blocks.size() && g->blocks[k].out_off + g->blocks[k].out_size <= g->hs->header_bytes) k++;
            if (k < g->blocks.size() || g->whole_file_scanned()) break;
            if (!g->scan_to(g->scanned + ((size_t)1 << 20))) return MDX_ERR_ARG;
        }
        g->next_block = k;
        g->phase = k < g->blocks.size() ? g->hs->header_bytes - g->blocks[k].out_off : 0;
        g->host_share = g_host_share.load();
        if (const char *e = std::getenv("MDX_GBAM_HOST_SHARE")) {
            g->host_share = std::min(0.9, std::max(0.0, std::atof(e)));
            g->host_share_fixed = true;
        }
        if (const char *e = std::getenv("MDX_GBAM_NO_LOOKAHEAD")) g->no_ahead = *e && *e != '0';
        if (hipSetDevice(g->device) != hipSuccess || mdx_k_gbam_prepare() != hipSuccess) { g->error = "HIP set-up failed"; return MDX_ERR_HIP; }
        {
            static mdx_crc32::Tables tables;
            static std::once_flag once;
            std::call_once(once, [] { mdx_crc32::make_tables(tables); });
            // (what the previous file of this context left behind: its arena and the tables; the second arena from the process's
            // spare buffers)
            void *ev = nullptr;
            mdx_ctx_scratch_take(ctx, &g->slab[0].arena.p, &g->slab[0].arena.cap, &g->d_crc_tables, (void **)&g->copy_stream, &ev);
            if (ev) (void)hipEventDestroy((hipEvent_t)ev);
            arena_take(g->device, g->slab[1].arena.p, g->slab[1].arena.cap);
            if (!g->d_crc_tables && (hipMalloc(&g->d_crc_tables, sizeof(tables)) != hipSuccess ||
                hipMemcpy(g->d_crc_tables, &tables, sizeof(tables), hipMemcpyHostToDevice) != hipSuccess)) { g->error = "HIP set-up failed"; return MDX_ERR_HIP; }
            if (!g->copy_stream && hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess) { g->error = "HIP set-up failed"; return MDX_ERR_HIP; }
            if (hipStreamCreateWithFlags(&g->infl_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&g->ev_free, hipEventDisableTiming) != hipSuccess) { g->error = "HIP set-up failed"; return MDX_ERR_HIP; }
            for (auto &s : g->slab)
                if (hipEventCreate(&s.ev_infl0) != hipSuccess || hipEventCreate(&s.ev_infl) != hipSuccess) { g->error = "HIP set-up failed"; return MDX_ERR_HIP; }
        }
        return MDX_OK;
    } catch (const std::exception &e) {
        if (out && *out) (*out)->error = std::string("mdx_gbam_open: ") + e.what();
        return MDX_ERR_ARG;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

const mdx_bam *mdx_gbam_header(const mdx_gbam *g) { return (g && g->hs) ? &g->hs->head : nullptr; }
const char *mdx_gbam_error(const mdx_gbam *g) { return g ? g->error.c_str() : "null handle"; }

int mdx_gbam_configure(mdx_gbam *g, int32_t n_rg, const char *const *rg_ids, const int32_t *lib_of_rg, int32_t lib_default,
                       int want_qual, int want_mate) {
    try {
        if (!g || n_rg < 0 || (n_rg > 0 && (!rg_ids || !lib_of_rg))) return MDX_ERR_ARG;
        g->rg_names.clear(); g->rg_off.assign(1, 0); g->lib_of_rg.clear();
        for (int i = 0; i < n_rg; i++) {
            const size_t len = std::strlen(rg_ids[i]);
            g->rg_names.insert(g->rg_names.end(), (const uint8_t *)rg_ids[i], (const uint8_t *)rg_ids[i] + len);
            g->rg_off.push_back((uint32_t)g->rg_names.size());
            g->lib_of_rg.push_back(lib_of_rg[i]);
        }
        g->lib_default = lib_default;
        g->want_qual = want_qual != 0; g->want_mate = want_mate != 0;
        if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
        g->d_rg_names = g->d_rg_off = g->d_lib_of_rg = nullptr;
        if (n_rg > 0) {
            // one allocation, one copy: [offsets][libraries][names]
            const size_t b_off = g->rg_off.size() * 4, b_lib = g->lib_of_rg.size() * 4, b_nm = g->rg_names.size() + 1;
            const size_t need = b_off + b_lib + b_nm;
            if (g->d_rg && g->d_rg_cap < need) { rg_buffer_give(g->device, g->d_rg, g->d_rg_cap); g->d_rg = nullptr; g->d_rg_cap = 0; }
            if (!g->d_rg) g->d_rg = rg_buffer_take(g->device, need, g->d_rg_cap);
            if (!g->d_rg) { g->error = "out of device memory"; return MDX_ERR_HIP; }
            std::vector<uint8_t> img(need, 0);
            std::memcpy(img.data(), g->rg_off.data(), b_off);
            std::memcpy(img.data() + b_off, g->lib_of_rg.data(), b_lib);
            std::memcpy(img.data() + b_off + b_lib, g->rg_names.data(), g->rg_names.size());
            if (hipMemcpy(g->d_rg, img.data(), need, hipMemcpyHostToDevice) != hipSuccess) { g->error = "upload of the read-group tables failed"; return MDX_ERR_HIP; }
            g->d_rg_off = g->d_rg; g->d_lib_of_rg = (char *)g->d_rg + b_off; g->d_rg_names = (char *)g->d_rg + b_off + b_lib;
        }
        return MDX_OK;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

// Compressed bytes of the slab that starts at block b0: the rest of the file in as many slabs as `chunk_bytes` asks for, all
// of one size — a last slab of a few blocks pays the fixed times of a whole one, and the compressed bytes sent ahead hide
// best behind an inflate of their own size (an 8 M-record file of 385 MB: two slabs of 193 MB, 61 ms, against 256 + 129 MB,
// 63-66 ms).  A function of the file's size, the slab's first block and chunk_bytes only: the ranks of a multi-GPU run, which
// step over each other's slabs (mdx_gbam_skip), agree on the borders.
static size_t slab_want(const mdx_gbam *g, size_t b0, int64_t chunk_bytes) {
    size_t want = chunk_bytes < 65536 ? 65536 : (size_t)chunk_bytes;
    const size_t fsz = g->hs->file->size();
    const size_t in0 = b0 < g->blocks.size() ? (size_t)g->blocks[b0].in_off : g->scanned;
    if (fsz > in0) {
        const size_t rem = fsz - in0, n = (rem + want - 1) / want;
        if (n > 1) want = (rem + n - 1) / n;
        if (want < 65536) want = 65536;
    }
    return want;
}
// ... and the block behind its last one
static size_t slab_end_block(const mdx_gbam *g, size_t b0, size_t want, size_t *slab_bytes_out) {
    size_t b1 = b0, slab_bytes = 0;
    const size_t in0 = g->blocks[b0].in_off;
    while (b1 < g->blocks.size() && (b1 == b0 || (g->blocks[b1].in_off - in0 < want && slab_bytes + g->blocks[b1].out_size < 0xE0000000ull))) {
        slab_bytes += g->blocks[b1].out_size;
        b1++;
    }
    if (slab_bytes_out) *slab_bytes_out = slab_bytes;
    return b1;
}

namespace {
// MDX_BAM_TRACE=1: when this thread reached each point of a call, without waiting for anything it would not wait for
// anyway — one line per call on stderr
struct GbamTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    std::vector<std::pair<const char *, double>> marks;
    GbamTrace() : on(std::getenv("MDX_BAM_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) { if (on) marks.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
    ~GbamTrace() {
        if (!on || marks.empty()) return;
        std::string line = "mdx_gbam_next trace (ms):";
        char buf[96];
        for (auto &m : marks) { std::snprintf(buf, sizeof buf, " %s %.2f", m.first, m.second); line += buf; }
        std::fprintf(stderr, "%s\n", line.c_str());
    }
};
// MDX_BAM_TIMING=1: stage times on stderr (each lap waits for the device; no slab is prepared ahead then)
struct GbamLaps {
    mdx_gbam *g;
    bool on;
    std::chrono::steady_clock::time_point t_last;
    explicit GbamLaps(mdx_gbam *g_) : g(g_), on(std::getenv("MDX_BAM_TIMING") != nullptr), t_last(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        (void)g->drain();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "mdx_gbam_next %-12s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    }
};
const size_t kHostMinBlocks = [] { const char *e = std::getenv("MDX_GBAM_HOST_MIN_BLOCKS"); return e ? (size_t)std::max(2, std::atoi(e)) : (size_t)512; }();
const int kNoBad = 0x7FFFFFFF;
}  // namespace

// The next slab's block headers and the device's part of its compressed bytes, sent ahead: the walk over the headers touches a
// page of the file per block, the copy out of the file's (pageable) mapping keeps this thread busy for as long as it takes —
// both while the device and the host's threads inflate the slab in hand.
static int gbam_send_ahead(mdx_gbam *g, const mdx_gbam::Slab &s, size_t want, GbamTrace &tr) {
    const size_t b1 = s.b1;
    const size_t in1 = g->blocks[s.b2 - 1].in_off + g->blocks[s.b2 - 1].in_size;
    if (!g->scan_to(in1 + want + 65536)) return MDX_ERR_ARG;
    tr.mark("headers");
    static const bool no_prefetch = [] { const char *e = std::getenv("MDX_GBAM_NO_PREFETCH"); return e && *e && *e != '0'; }();
    if (no_prefetch || b1 >= g->blocks.size()) return MDX_OK;
    const size_t pwant = slab_want(g, b1, s.chunk_bytes);
    size_t pbytes = 0;
    const size_t p1 = slab_end_block(g, b1, pwant, &pbytes);
    const size_t pin0 = g->blocks[b1].in_off;
    // (all of the slab's own blocks but the host's share of the inflated bytes, as its half A will split them)
    size_t upto = p1;
    if (g->host_share > 0.0 && p1 - b1 >= kHostMinBlocks) {
        const size_t target = (size_t)((double)pbytes * g->host_share);
        size_t acc = 0;
        while (upto > b1 + 1 && acc + g->blocks[upto - 1].out_size <= target) { acc += g->blocks[upto - 1].out_size; upto--; }
    }
    const size_t bytes = (upto < g->blocks.size() ? g->blocks[upto].in_off : g->blocks[upto - 1].in_off + g->blocks[upto - 1].in_size) - pin0;
    // a buffer neither of the two slabs in flight reads
    int k = 0;
    while (k < 3 && (k == g->slab[0].pf_used || k == g->slab[1].pf_used)) k++;
    if (k >= 3) return MDX_OK;
    const size_t cap_want = pwant + ((size_t)16 << 20);
    if (g->pf_cap[k] < cap_want) {
        if (g->pf_buf[k]) (void)hipFree(g->pf_buf[k]);
        g->pf_buf[k] = nullptr; g->pf_cap[k] = 0;
        pf_buffer_take(g->device, cap_want, g->pf_buf[k], g->pf_cap[k]);
        if (!g->pf_buf[k]) {
            if (hipMalloc(&g->pf_buf[k], cap_want) == hipSuccess) g->pf_cap[k] = cap_want;
            else (void)hipGetLastError();
        }
    }
    g->pf_cur = -1;
    if (g->pf_cap[k] >= bytes + 64 && bytes > 0 &&
        hipMemcpyAsync(g->pf_buf[k], g->hs->file->p + pin0, bytes, hipMemcpyHostToDevice, g->copy_stream) == hipSuccess) {
        g->pf_cur = k; g->pf_in0 = pin0; g->pf_bytes = bytes;
    }
    tr.mark("sent-ahead");
    return MDX_OK;
}

// Half A of the slab that starts at block b0, into `s`: the slab's and the blocks behind it that its last record may reach
// into (`ahead` inflated bytes of them), the device's part of the compressed bytes in HBM, its inflate launched on
// infl_stream, the host's part inflated by the pool and copied into place on copy_stream.
static int gbam_half_a(mdx_gbam *g, mdx_gbam::Slab &s, size_t b0, int64_t chunk_bytes, size_t ahead, GbamLaps &laps, GbamTrace &tr) {
    s.ready = false;
    s.pf_used = -1;
    const size_t want = slab_want(g, b0, chunk_bytes);
    if (!g->scan_to((size_t)g->blocks[b0].in_off + want + 65536)) return MDX_ERR_ARG;
    const size_t b1 = slab_end_block(g, b0, want, &s.slab_bytes);
    const size_t in0 = g->blocks[b0].in_off, out0 = g->blocks[b0].out_off;
    size_t b2 = b1, unc_bytes = s.slab_bytes;
    while (unc_bytes - s.slab_bytes < ahead && unc_bytes < 0xF0000000ull) {
        if (b2 >= g->blocks.size()) {
            if (g->whole_file_scanned()) break;
            if (!g->scan_to(g->scanned + ((size_t)4 << 20))) return MDX_ERR_ARG;
            continue;
        }
        unc_bytes += g->blocks[b2].out_size;
        b2++;
    }
    s.chunk_bytes = chunk_bytes; s.b0 = b0; s.b1 = b1; s.b2 = b2; s.unc_bytes = unc_bytes; s.out0 = out0; s.in0 = in0; s.ahead = ahead;
    s.more_file = b2 < g->blocks.size() || !g->whole_file_scanned();
    const size_t nba = b2 - b0;               // blocks inflated: the slab's and those ahead
    const size_t in1 = g->blocks[b2 - 1].in_off + g->blocks[b2 - 1].in_size;
    const size_t comp_bytes = in1 - in0;
    std::vector<uint32_t> &blk = s.blk;
    blk.resize(4 * nba); s.crcs.resize(nba);
    for (size_t i = 0; i < nba; i++) {
        const Block &b = g->blocks[b0 + i];
        s.crcs[i] = b.crc;
        blk[4 * i] = (uint32_t)(b.in_off - in0); blk[4 * i + 1] = (uint32_t)b.in_size;
        blk[4 * i + 2] = (uint32_t)(b.out_off - out0); blk[4 * i + 3] = (uint32_t)b.out_size;
    }
    // upper bounds of the columns from the inflated size: a record is at least 36 bytes, and holds its bases
    // twice over (4 bits + a quality byte each): l_seq <= 2/3 of its size
    s.rec_cap = unc_bytes / 36 + 2; s.cig_cap = unc_bytes / 4 + 2; s.seq_cap = unc_bytes + 64;
    // the host's share: the blocks [nh, nba), the last `host_share` of the inflated bytes (slabs of a few hundred blocks
    // are the device's alone; MDX_GBAM_HOST_MIN_BLOCKS: tests put small files through the host's share)
    size_t nh = nba;
    if (g->host_share > 0.0 && nba >= kHostMinBlocks) {
        const size_t target = (size_t)((double)unc_bytes * g->host_share);
        size_t acc = 0;
        while (nh > 1 && acc + g->blocks[b0 + nh - 1].out_size <= target) { acc += g->blocks[b0 + nh - 1].out_size; nh--; }
        if (nba - nh < std::min<size_t>(64, kHostMinBlocks / 2)) nh = nba;
    }
    s.nh = nh;
    const size_t head_comp = nh == nba ? comp_bytes : (size_t)blk[4 * nh];        // compressed bytes the device needs
    // the arena is written behind its last readers: the caller's tabulation of the slab that lived here (enqueued on the
    // context's stream before this call)
    hipStream_t si = g->infl_stream;
    if (hipEventRecord(g->ev_free, g->stream) != hipSuccess || hipStreamWaitEvent(si, g->ev_free, 0) != hipSuccess ||
        hipStreamWaitEvent(g->copy_stream, g->ev_free, 0) != hipSuccess) return MDX_ERR_HIP;
    // one allocation for everything (twenty hipMalloc / hipFree pairs were a tenth of a small file's time)
    {
        struct Want { mdx_gbam::Buf *b; size_t bytes; };
        const Want wants[] = {
            {&s.comp, comp_bytes + 64}, {&s.blk_d, nba * 16}, {&s.crc, nba * 4}, {&s.status, nba * 4}, {&s.unc, unc_bytes + 64}, {&s.cnt, nba * 16},
            {&s.pre, nba * 16}, {&s.info, nba * 16}, {&s.forced, nba * 4}, {&s.small, 64}, {&s.rec_off, s.rec_cap * 4}, {&s.flag, s.rec_cap * 2},
            {&s.lib, s.rec_cap * 2}, {&s.tid, s.rec_cap * 4}, {&s.pos, s.rec_cap * 4}, {&s.tlen, s.rec_cap * 4}, {&s.cigar_off, s.rec_cap * 4},
            {&s.seq_off, s.rec_cap * 4}, {&s.cigar, s.cig_cap * 4}, {&s.seq, s.seq_cap}, {&s.qual, g->want_qual ? s.seq_cap : 0},
            {&s.mtid, g->want_mate ? s.rec_cap * 4 : 0}, {&s.mpos, g->want_mate ? s.rec_cap * 4 : 0}};
        size_t total_bytes = 0;
        for (const Want &w : wants) total_bytes += (w.bytes + 255) & ~(size_t)255;
        // (an arena that must grow is released first: everything that reads it must have run)
        if (total_bytes > s.arena.cap && !g->drain()) return MDX_ERR_HIP;
        if (!g->reserve(s.arena, total_bytes)) return MDX_ERR_HIP;
        size_t at = 0;
        for (const Want &w : wants) { w.b->p = w.bytes ? (char *)s.arena.p + at : nullptr; w.b->cap = 0; at += (w.bytes + 255) & ~(size_t)255; }
    }
    laps.lap("allocate");
    tr.mark("arena");
    (void)hipEventRecord(s.ev_infl0, si);
    // (the bytes an earlier call sent ahead, if they are these: what is missing of them is sent behind)
    const uint8_t *comp_dev = (const uint8_t *)s.comp.p;
    if (g->pf_cur >= 0 && g->pf_in0 == in0 && g->pf_bytes > 0 && g->pf_cap[g->pf_cur] >= head_comp + 64) {
        const int k = g->pf_cur;
        if (hipStreamSynchronize(g->copy_stream) != hipSuccess) return MDX_ERR_HIP;
        if (g->pf_bytes < head_comp &&
            hipMemcpyAsync((char *)g->pf_buf[k] + g->pf_bytes, g->hs->file->p + in0 + g->pf_bytes, head_comp - g->pf_bytes, hipMemcpyHostToDevice, si) != hipSuccess) {
            g->error = "upload failed"; return MDX_ERR_HIP;
        }
        comp_dev = (const uint8_t *)g->pf_buf[k];
        s.pf_used = k;
    }
    g->pf_cur = -1;
    if ((s.pf_used < 0 && hipMemcpyAsync(s.comp.p, g->hs->file->p + in0, head_comp, hipMemcpyHostToDevice, si) != hipSuccess) ||
        hipMemcpyAsync(s.blk_d.p, blk.data(), nba * 16, hipMemcpyHostToDevice, si) != hipSuccess ||
        hipMemcpyAsync(s.crc.p, s.crcs.data(), nba * 4, hipMemcpyHostToDevice, si) != hipSuccess) { g->error = "upload failed"; return MDX_ERR_HIP; }
    int *d_bad_crc = (int *)((char *)s.small.p + 40);
    (void)hipMemcpyAsync(d_bad_crc, &kNoBad, 4, hipMemcpyHostToDevice, si);
    (void)hipMemsetAsync(s.forced.p, 0xFF, nba * 4, si);
    // (a slab whose compressed bytes were not sent ahead — a file's first: the host's threads are about to read the file, and
    // the runtime, copying from pageable memory, works on the same address space: one after the other)
    if (nh < nba && s.pf_used < 0 && !std::getenv("MDX_GBAM_NO_UPLOAD_SYNC") && hipStreamSynchronize(si) != hipSuccess) return MDX_ERR_HIP;
    laps.lap("upload");
    tr.mark("uploaded");
    mdx_k_gbam_inflate(comp_dev, (const uint4 *)s.blk_d.p, (int)nh, (uint8_t *)s.unc.p, (int *)s.status.p, si);
    (void)hipEventRecord(s.ev_infl, si);
    s.timed = false;
    if (nh < nba) {
        // ---- the host's blocks: inflated (and CRC-checked) by the pool into hbuf, piece by piece; this thread copies
        // every finished piece to its place in `unc` on the copy stream, under the device's inflate
        if (!g->pool) g->pool = host_pool();
        if (!g->hbuf) host_buffer_take(g->hbuf, g->hbuf_cap);
        const size_t nt = nba - nh, tail0 = blk[4 * nh + 2], tail_bytes = unc_bytes - tail0;
        if (g->hbuf_cap < tail_bytes + 64) {
            // (room for the largest share of a slab like this one at a share that has grown by three quarters: the share
            // moves by halves towards the balance of the two rates, and pinned memory costs a tenth of a second per 600 MB to
            // map; the copies out of the old buffer must have run)
            if (hipStreamSynchronize(g->copy_stream) != hipSuccess) return MDX_ERR_HIP;
            if (g->hbuf) (void)hipHostFree(g->hbuf);
            g->hbuf = nullptr;
            g->hbuf_cap = std::max(tail_bytes + 64, (size_t)((double)unc_bytes * std::min(0.46, std::max(0.2, 1.75 * g->host_share))) + ((size_t)1 << 20));
            const auto t_p = std::chrono::steady_clock::now();
            if (hipHostMalloc((void **)&g->hbuf, g->hbuf_cap, hipHostMallocDefault) != hipSuccess) { g->hbuf_cap = 0; g->error = "out of pinned host memory"; return MDX_ERR_HIP; }
            if (laps.on || tr.on) std::fprintf(stderr, "mdx_gbam_next pinned buffer of %.0f MB: %.1f ms\n", g->hbuf_cap / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p).count());
        }
        // pieces of about 8 MiB of inflated bytes; done[j] counts the finished blocks of piece j
        std::vector<size_t> piece_lo;          // first block (relative to nh) of every piece, and the end
        {
            size_t acc = 0;
            piece_lo.push_back(0);
            for (size_t i = 0; i < nt; i++) {
                acc += blk[4 * (nh + i) + 3];
                if (acc >= ((size_t)8 << 20) && i + 1 < nt) { piece_lo.push_back(i + 1); acc = 0; }
            }
            piece_lo.push_back(nt);
        }
        const size_t np = piece_lo.size() - 1;
        std::vector<uint32_t> piece_of(nt);
        for (size_t j = 0; j < np; j++) for (size_t i = piece_lo[j]; i < piece_lo[j + 1]; i++) piece_of[i] = (uint32_t)j;
        std::unique_ptr<std::atomic<uint32_t>[]> done(new std::atomic<uint32_t>[np]);
        for (size_t j = 0; j < np; j++) done[j] = 0;
        std::atomic<int> bad_block{-1};
        std::vector<int32_t> host_status(nt);
        const MappedFile &file = *g->hs->file;
        // (a copy: the walk over the next slab's headers, below, appends to g->blocks while the pool reads these)
        const std::vector<Block> host_blocks(g->blocks.begin() + (long)(b0 + nh), g->blocks.begin() + (long)(b0 + nba));
        const Block *const bl = host_blocks.data();
        uint8_t *const hb = g->hbuf;
        // (the pinned buffer is the previous slab's too: its copies must have left it)
        if (hipStreamSynchronize(g->copy_stream) != hipSuccess) return MDX_ERR_HIP;
        const auto t_h0 = std::chrono::steady_clock::now();
        g->pool->run(nt, 4, [&, bl, hb, tail0, out0](size_t i) {
            const Block &k = bl[i];
            const size_t off = (size_t)(k.out_off - out0) - tail0;
            // (the block's compressed bytes through pread() into a buffer of the thread's own: a hundred threads faulting
            // pages of the file's mapping in — under the address space's lock, next to whatever else maps and unmaps —
            // stalled for tens of milliseconds now and then)
            thread_local std::vector<uint8_t> mine;
            const uint8_t *src = &file[k.in_off];
            if (file.fd >= 0 && k.in_size) {
                if (mine.size() < k.in_size) mine.resize(std::max<size_t>(k.in_size, 80 << 10));
                size_t got = 0;
                while (got < k.in_size) {
                    const ssize_t r = pread(file.fd, mine.data() + got, k.in_size - got, (off_t)(k.in_off + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got == k.in_size) src = mine.data();
            }
            const bool ok = k.out_size == 0 || inflate_block(src, k.in_size, hb + off, k.out_size, k.crc);
            host_status[i] = ok ? (int32_t)k.out_size : -1;
            if (!ok) { int expect = -1; bad_block.compare_exchange_strong(expect, (int)i); }
            done[piece_of[i]].fetch_add(1, std::memory_order_release);
        });
        // (this thread has nothing to copy for a while: the next slab's headers and compressed bytes meanwhile)
        int rc_ahead = MDX_OK;
        if (!laps.on) rc_ahead = gbam_send_ahead(g, s, want, tr);
        bool copy_failed = false;
        double t_waited = 0, t_copied = 0;
        for (size_t j = 0; j < np; j++) {
            const auto t_a = std::chrono::steady_clock::now();
            const uint32_t need = (uint32_t)(piece_lo[j + 1] - piece_lo[j]);
            while (done[j].load(std::memory_order_acquire) < need) std::this_thread::sleep_for(std::chrono::microseconds(30));
            const auto t_b = std::chrono::steady_clock::now();
            const size_t lo = (size_t)blk[4 * (nh + piece_lo[j]) + 2] - tail0;
            const size_t hi = piece_lo[j + 1] < nt ? (size_t)blk[4 * (nh + piece_lo[j + 1]) + 2] - tail0 : tail_bytes;
            if (hi > lo && hipMemcpyAsync((char *)s.unc.p + tail0 + lo, hb + lo, hi - lo, hipMemcpyHostToDevice, g->copy_stream) != hipSuccess)
                copy_failed = true;
            const auto t_c = std::chrono::steady_clock::now();
            t_waited += std::chrono::duration<double, std::milli>(t_b - t_a).count();
            t_copied += std::chrono::duration<double, std::milli>(t_c - t_b).count();
        }
        g->pool->wait();
        tr.mark("host-share");
        s.t_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h0).count();
        s.frac_host = (double)tail_bytes / (double)unc_bytes;
        s.timed = true;
        if (laps.on)
            std::fprintf(stderr, "mdx_gbam_next host share  %.2f: %zu of %zu blocks, %.1f MB inflated on %zu threads; waited for them %.2f ms, copies %.2f ms, all %.2f ms; device %s\n",
                         g->host_share, nt, nba, tail_bytes / 1e6, g->pool->threads.size(), t_waited, t_copied, s.t_host,
                         hipEventQuery(s.ev_infl) == hipSuccess ? "done" : "still inflating");
        if (copy_failed || hipMemcpyAsync((int32_t *)s.status.p + nh, host_status.data(), nt * 4, hipMemcpyHostToDevice, g->copy_stream) != hipSuccess ||
            hipStreamSynchronize(g->copy_stream) != hipSuccess) { g->error = "upload of the host-inflated blocks failed"; return MDX_ERR_HIP; }
        tr.mark("host-copied");
        if (bad_block.load() >= 0) {
            g->error = "corrupt BGZF block " + std::to_string(b0 + nh + (size_t)bad_block.load()) + " (DEFLATE stream, ISIZE or CRC32)";
            return MDX_ERR_ARG;
        }
        if (rc_ahead != MDX_OK) return rc_ahead;
    } else if (!laps.on) {
        const int rc_ahead = gbam_send_ahead(g, s, want, tr);
        if (rc_ahead != MDX_OK) return rc_ahead;
    }
    laps.lap("inflate");
    s.ready = true;
    return MDX_OK;
}

// The share that would have let both sides finish together, from the two rates of a slab whose device inflate has run — the
// device's inflate of its blocks, the host's inflate and copies of the others — approached by halves.
static void gbam_adapt_share(mdx_gbam *g, mdx_gbam::Slab &s) {
    if (g->host_share_fixed || !s.timed) return;
    s.timed = false;
    float t_dev = 0.f;
    if (hipEventQuery(s.ev_infl) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventElapsedTime(&t_dev, s.ev_infl0, s.ev_infl) != hipSuccess || t_dev <= 0.1f || s.t_host <= 0.1) { (void)hipGetLastError(); return; }
    const double f = s.frac_host;
    const double r_host = f / s.t_host, r_dev = (1.0 - f) / (double)t_dev;
    const double balanced = r_host / (r_host + r_dev);
    // (aiming a little below the balance: a host that finishes early costs nothing, one that finishes late costs its lateness)
    g->host_share = std::min(0.45, std::max(0.04, 0.5 * g->host_share + 0.5 * 0.85 * balanced));
    g_host_share.store(g->host_share);
}

// Half B of a slab whose half A is done: CRC (beside, on the copy stream), the record chains of its segments and their check on
// the host, unpack into the columns; the view.  *grow: the slab's last record reaches further than the blocks inflated behind
// it (half A again, with more of them).
static int gbam_half_b(mdx_gbam *g, mdx_gbam::Slab &s, mdx_batch *view, const int32_t **d_mtid, const int32_t **d_mpos, bool *grow_out,
                       GbamLaps &laps, GbamTrace &tr) {
    *grow_out = false;
    hipStream_t st = g->stream;
    const size_t b0 = s.b0, b1 = s.b1, nb = s.b1 - s.b0, nba = s.b2 - s.b0, nh = s.nh, unc_bytes = s.unc_bytes, slab_bytes = s.slab_bytes;
    const std::vector<uint32_t> &blk = s.blk;
    const bool more_file = s.more_file;
    int *d_bad_crc = (int *)((char *)s.small.p + 40);
    // the context's stream and the copy stream go on behind the device's inflate
    if (hipStreamWaitEvent(st, s.ev_infl, 0) != hipSuccess) return MDX_ERR_HIP;
    // (the CRC beside what follows, on the copy stream; its verdict is looked at before the slab is handed out.
    // MDX_BAM_TIMING: in line, so that its lap is its own)
    bool crc_aside = !laps.on;
    if (crc_aside) {
        if (!g->pin_bad) g->pin_bad = pin_take();
        if (!g->pin_bad) crc_aside = false;
    }
    if (crc_aside) {
        *g->pin_bad = kNoBad;
        if (hipStreamWaitEvent(g->copy_stream, s.ev_infl, 0) != hipSuccess) return MDX_ERR_HIP;
        mdx_k_gbam_crc((const uint8_t *)s.unc.p, (const uint4 *)s.blk_d.p, (const uint32_t *)s.crc.p, g->d_crc_tables, (int)nh, d_bad_crc, g->copy_stream);
        if (hipMemcpyAsync(g->pin_bad, d_bad_crc, 4, hipMemcpyDeviceToHost, g->copy_stream) != hipSuccess) return MDX_ERR_HIP;
    } else
    mdx_k_gbam_crc((const uint8_t *)s.unc.p, (const uint4 *)s.blk_d.p, (const uint32_t *)s.crc.p, g->d_crc_tables, (int)nh, d_bad_crc, st);
    laps.lap("crc32");
    // the chains of the segments (every block inflated is one: those ahead of the slab say whether the next slab's
    // first record can be found without this one), then their check on the host
    const int n_ref = (int)g->hs->head.ref_names.size();
    const uint32_t start0 = g->phase_known ? (uint32_t)g->phase : 0xFFFFFFFFu;
    mdx_k_gbam_scan((const uint8_t *)s.unc.p, (const uint4 *)s.blk_d.p, (const int *)s.status.p, (int)nba, 0, nullptr, start0,
                    (uint32_t)unc_bytes, n_ref, (uint4 *)s.info.p, (uint4 *)s.cnt.p, st);
    laps.lap("scan");
    std::vector<uint32_t> info(4 * nba), cnt(4 * nba);
    int bad_crc = kNoBad;
    if (hipMemcpyAsync(info.data(), s.info.p, nba * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(cnt.data(), s.cnt.p, nba * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (!crc_aside && hipMemcpyAsync(&bad_crc, d_bad_crc, 4, hipMemcpyDeviceToHost, st) != hipSuccess) || hipStreamSynchronize(st) != hipSuccess) {
        g->error = std::string("GPU decode failed: ") + hipGetErrorString(hipGetLastError());
        return MDX_ERR_HIP;
    }
    tr.mark("scanned");
    gbam_adapt_share(g, s);
    for (size_t i = 0; i < nba; i++)
        if ((int32_t)info[4 * i + 2] == -2) {
            int stv = 0;
            (void)hipMemcpy(&stv, (const int *)s.status.p + i, 4, hipMemcpyDeviceToHost);
            g->error = "corrupt BGZF block " + std::to_string(b0 + i) + " (inflate code " + std::to_string(stv) + ")";
            return MDX_ERR_ARG;
        }
    if (bad_crc != kNoBad) { g->error = "corrupt BGZF block " + std::to_string(b0 + (size_t)bad_crc) + " (CRC32)"; return MDX_ERR_ARG; }
    // (crc_aside: the same check behind the unpack's launch, below)
    auto crc_verdict = [&]() -> bool {
        if (!crc_aside) return true;
        if (hipStreamSynchronize(g->copy_stream) != hipSuccess) { g->error = "GPU decode failed (CRC32 pass)"; return false; }
        if (*g->pin_bad != kNoBad) { g->error = "corrupt BGZF block " + std::to_string(b0 + (size_t)*g->pin_bad) + " (CRC32)"; return false; }
        return true;
    };
    // The walk: `at` = where the next record starts, known exactly; the segment that holds it must have begun its chain
    // there — if its guess was another offset it is scanned again from the right one — and says where the chain lands.
    const size_t slab_end = slab_bytes;
    std::vector<uint8_t> used(nba, 0);
    size_t at = 0, seg = 0;
    bool have = g->phase_known;
    if (have) at = g->phase;
    else {
        // behind a skipped slab: the first guess of this slab stands (its neighbour checks it: next_verified)
        for (size_t i = 0; i < nb && !have; i++)
            if (info[4 * i + 2] != 1u) { have = true; at = info[4 * i]; }
        if (!have) at = slab_end;              // no record starts in this slab
    }
    bool grow = false;
    while (at < slab_end) {
        while (seg < nba && (size_t)blk[4 * seg + 2] + blk[4 * seg + 3] <= at) seg++;
        if (seg >= nb) break;
        if (info[4 * seg] != (uint32_t)at || info[4 * seg + 2] == 1u) {
            // (rare: a guess that was not the record's start, or a segment whose first record begins behind a long one)
            const int32_t f = (int32_t)at;
            if (hipMemcpyAsync((int32_t *)s.forced.p + seg, &f, 4, hipMemcpyHostToDevice, st) != hipSuccess) return MDX_ERR_HIP;
            mdx_k_gbam_scan((const uint8_t *)s.unc.p, (const uint4 *)s.blk_d.p, (const int *)s.status.p, (int)seg + 1, (int)seg,
                            (const int *)s.forced.p, start0, (uint32_t)unc_bytes, n_ref, (uint4 *)s.info.p, (uint4 *)s.cnt.p, st);
            if (hipMemcpyAsync(&info[4 * seg], (const uint4 *)s.info.p + seg, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(&cnt[4 * seg], (const uint4 *)s.cnt.p + seg, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return MDX_ERR_HIP;
            g->fixups++;
        }
        const int32_t stv = (int32_t)info[4 * seg + 2];
        if (stv == -1) { g->error = "corrupt BAM record in BGZF block " + std::to_string(b0 + seg); return MDX_ERR_ARG; }
        if (stv == 3) { g->error = "a record that keeps its CIGAR in a CG tag (more than 65 535 operations): the host decoder's"; return MDX_ERR_UNSUPPORTED; }
        if (stv == 2 && cnt[4 * seg] == 0 && info[4 * seg + 1] == (uint32_t)at && !more_file) {
            g->error = "truncated BAM file: the last record is incomplete"; return MDX_ERR_ARG;
        }
        used[seg] = 1;
        const size_t land = info[4 * seg + 1];
        if (stv == 2 && land < slab_end) {
            // the chain stopped at a record that starts in the slab and is not complete in what was inflated
            if (!more_file) { g->error = "truncated BAM file: the last record is incomplete"; return MDX_ERR_ARG; }
            grow = true;
            break;
        }
        if (land <= at) { g->error = "corrupt BAM records"; return MDX_ERR_ARG; }
        at = land;
    }
    if (grow) {
        // (the CRC pass of this attempt reads the bytes the next one writes)
        if (crc_aside && hipStreamSynchronize(g->copy_stream) != hipSuccess) return MDX_ERR_HIP;
        if (s.ahead >= ((size_t)1 << 30)) { g->error = "a BAM record of more than a gigabyte"; return MDX_ERR_UNSUPPORTED; }
        *grow_out = true;
        return MDX_OK;
    }
    // where the next slab's first record starts — and whether a process that has not seen this slab would find it: the
    // segment ahead that holds it must have guessed exactly that offset (mdx_gbam_skip refuses to go on otherwise)
    const size_t next_phase = at > slab_end ? at - slab_end : 0;
    {
        size_t sg = nb;
        while (sg < nba && (size_t)blk[4 * sg + 2] + blk[4 * sg + 3] <= at) sg++;
        if (b1 >= g->blocks.size() && g->whole_file_scanned()) g->next_verified = true;      // nothing follows
        else {
            // (the rank behind takes the first segment of its slab that has any guess: a segment in front of the right one —
            // inside a record that straddles whole blocks — must not have produced one)
            bool clean = true;
            for (size_t i = nb; i < sg && i < nba; i++) clean = clean && info[4 * i + 2] == 1u;
            g->next_verified = clean && sg < nba && info[4 * sg + 2] != 1u && info[4 * sg] == (uint32_t)at;
        }
    }
    // prefix sums over the segments that hold the chain
    std::vector<uint32_t> pre(4 * nb);
    unsigned long long tot[3] = {0, 0, 0};
    for (size_t i = 0; i < nb; i++) {
        if (!used[i]) { cnt[4 * i] = cnt[4 * i + 1] = cnt[4 * i + 2] = 0; }
        pre[4 * i] = (uint32_t)tot[0]; pre[4 * i + 1] = (uint32_t)tot[1]; pre[4 * i + 2] = (uint32_t)tot[2]; pre[4 * i + 3] = info[4 * i];
        tot[0] += cnt[4 * i]; tot[1] += cnt[4 * i + 1]; tot[2] += cnt[4 * i + 2];
    }
    if (tot[0] > s.rec_cap - 2 || tot[1] > s.cig_cap - 2 || tot[2] > s.seq_cap - 64 || tot[2] > 0xFFFFFFFFull) { g->error = "corrupt BAM records"; return MDX_ERR_ARG; }
    if (hipMemcpyAsync(s.pre.p, pre.data(), nb * 16, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(s.cnt.p, cnt.data(), nb * 16, hipMemcpyHostToDevice, st) != hipSuccess) return MDX_ERR_HIP;
    laps.lap("chains");
    MdxGbamCols c{};
    c.flag = (uint16_t *)s.flag.p; c.lib = (uint16_t *)s.lib.p; c.tid = (int32_t *)s.tid.p; c.pos = (int32_t *)s.pos.p;
    c.tlen = (int32_t *)s.tlen.p; c.mtid = g->want_mate ? (int32_t *)s.mtid.p : nullptr; c.mpos = g->want_mate ? (int32_t *)s.mpos.p : nullptr;
    c.cigar_off = (uint32_t *)s.cigar_off.p; c.cigar = (uint32_t *)s.cigar.p; c.seq_off = (uint32_t *)s.seq_off.p;
    c.seq = (uint8_t *)s.seq.p; c.qual = g->want_qual ? (uint8_t *)s.qual.p : nullptr;
    c.rg_names = (const uint8_t *)g->d_rg_names; c.rg_off = (const uint32_t *)g->d_rg_off; c.lib_of_rg = (const int32_t *)g->d_lib_of_rg;
    c.n_rg = (int)g->lib_of_rg.size(); c.lib_default = g->lib_default;
    uint32_t *d_counters = (uint32_t *)((char *)s.small.p + 48);
    c.minqual = g->want_qual ? g->minqual : 0; c.counters = d_counters;
    if (c.minqual > 0 && hipMemsetAsync(d_counters, 0, 8, st) != hipSuccess) return MDX_ERR_HIP;
    c.seq_packed = g->seq_format == MDX_SEQ_4BIT ? 1 : 0;
    // (the unpack kernel ORs the nibbles of a record into the column: zeroed first, with the dword behind the last base)
    if (c.seq_packed && hipMemsetAsync(c.seq, 0, (size_t)(tot[2] + 1) / 2 + 8, st) != hipSuccess) return MDX_ERR_HIP;
    // (--min-basequal: the mask goes into the nibbles — MDX_SEQ_4BITQ, the packed masked kernel's input)
    c.fold = (c.minqual > 0 && c.seq_packed) ? 1 : 0;
    mdx_k_gbam_unpack((const uint8_t *)s.unc.p, (const uint4 *)s.pre.p, (const uint4 *)s.cnt.p, (int)nb,
                      (uint32_t)tot[0], (uint32_t)tot[1], (uint32_t)tot[2], (uint32_t *)s.rec_off.p, c, st);
    if (hipGetLastError() != hipSuccess) { g->error = "GPU unpack launch failed"; return MDX_ERR_HIP; }
    tr.mark("unpack-enq");
    if (!crc_verdict()) return g->error.find("corrupt") != std::string::npos ? MDX_ERR_ARG : MDX_ERR_HIP;
    laps.lap("unpack");
    view->n_reads = (int64_t)tot[0]; view->n_cigar = (int64_t)tot[1]; view->n_bases = (int64_t)tot[2];
    view->flag = c.flag; view->lib = c.lib; view->tid = c.tid; view->pos = c.pos; view->tlen = c.tlen;
    view->cigar_off = c.cigar_off; view->cigar = c.cigar; view->seq_off = c.seq_off; view->seq = c.seq; view->qual = c.qual;
    view->seq_format = c.fold ? MDX_SEQ_4BITQ : g->seq_format; view->reserved = 0; view->lowq = nullptr; view->libsort = nullptr;
    if (c.minqual > 0) {
        uint32_t counters[2] = {0, 0};
        if (hipMemcpyAsync(counters, d_counters, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return MDX_ERR_HIP;
        if (counters[0]) g->no_qual_seen = true;
        // nothing in this slab can be masked: the unmasked kernel (no nibble of the column is a complement)
        if (counters[1] == 0) { view->qual = nullptr; view->seq_format = g->seq_format; }
    }
    if (d_mtid) *d_mtid = c.mtid;
    if (d_mpos) *d_mpos = c.mpos;
    g->last_view = *view; g->last_mtid = c.mtid; g->last_mpos = c.mpos;
    g->next_block = b1;
    g->phase = next_phase;
    g->phase_known = have;
    g->slabs_done++;
    g->view_reads = view->n_reads;
    s.ready = false;
    return MDX_OK;
}

int mdx_gbam_next(mdx_gbam *g, int64_t chunk_bytes, mdx_batch *view, const int32_t **d_mtid, const int32_t **d_mpos) {
    try {
        if (!g || !view) return MDX_ERR_ARG;
        std::memset(view, 0, sizeof(*view));
        if (d_mtid) *d_mtid = nullptr;
        if (d_mpos) *d_mpos = nullptr;
        // (the block headers of this slab, unless an earlier call has already walked them)
        if (!g->scan_to((g->next_block < g->blocks.size() ? g->blocks[g->next_block].in_off : g->scanned) + slab_want(g, g->next_block, chunk_bytes) + 65536)) return MDX_ERR_ARG;
        if (g->next_block >= g->blocks.size()) {                          // end of file: an empty view
            if (g->phase_known && g->phase != 0) { g->error = "truncated BAM file: the last record is incomplete"; return MDX_ERR_ARG; }
            return MDX_OK;
        }
        if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
        // (tests of the caller's fallback: MDX_GBAM_FAIL_AT=k makes the k-th slab decoded by this handle fail)
        if (const char *fail = std::getenv("MDX_GBAM_FAIL_AT")) {
            if (g->slabs_done == std::atoi(fail)) { g->error = "MDX_GBAM_FAIL_AT"; return MDX_ERR_UNSUPPORTED; }
        }
        GbamLaps laps(g);
        GbamTrace tr;
        mdx_gbam::Slab &s = g->slab[g->cur], &n = g->slab[g->cur ^ 1];
        // ... `ahead` inflated bytes of the blocks behind the slab, more if its last record turns out to be longer (half A of the
        // whole slab again: reads of a quarter of a megabyte are rare)
        size_t ahead = (size_t)256 << 10;
        for (;;) {
            if (!(s.ready && s.b0 == g->next_block && s.chunk_bytes == chunk_bytes && s.ahead == ahead)) {
                const int rc = gbam_half_a(g, s, g->next_block, chunk_bytes, ahead, laps, tr);
                if (rc != MDX_OK) { (void)g->drain(); return rc; }
            }
            tr.mark("A");
            // the slab behind it, ahead of time: its inflate queues behind this one's, the host's threads go on with its share
            if (!g->no_ahead && !laps.on && s.b1 < g->blocks.size() && !(n.ready && n.b0 == s.b1 && n.chunk_bytes == chunk_bytes)) {
                const std::string keep = g->error;
                if (gbam_half_a(g, n, s.b1, chunk_bytes, (size_t)256 << 10, laps, tr) != MDX_OK) {
                    // (whatever is wrong with that slab is reported by the call that hands it out)
                    (void)g->drain();
                    n.ready = false; g->pf_cur = -1; g->error = keep;
                }
                tr.mark("A-next");
            }
            bool grow = false;
            const int rc = gbam_half_b(g, s, view, d_mtid, d_mpos, &grow, laps, tr);
            if (rc != MDX_OK) { (void)g->drain(); s.ready = n.ready = false; return rc; }
            if (!grow) break;
            if (!g->drain()) return MDX_ERR_HIP;
            s.ready = false;
            ahead *= 8;
        }
        g->view_slab = g->cur;
        g->cur ^= 1;
        tr.mark("end");
        return MDX_OK;
    } catch (const std::exception &e) {
        if (g) g->error = std::string("mdx_gbam_next: ") + e.what();
        return MDX_ERR_ARG;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

int mdx_gbam_inflate_blocks(mdx_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const uint32_t *blk, int32_t n_blocks,
                            uint8_t *out, int64_t out_bytes, int32_t *status, const uint32_t *want_crc, uint8_t *crc_ok) {
    if (!ctx || !comp || !blk || !out || !status || n_blocks < 0 || comp_bytes < 0 || out_bytes < 0) return MDX_ERR_ARG;
    if (n_blocks == 0) return MDX_OK;
    for (int32_t b = 0; b < n_blocks; b++) {
        const uint32_t *e = blk + 4 * (size_t)b;
        if ((int64_t)e[0] + e[1] > comp_bytes || e[3] > 65536u || (int64_t)e[2] + e[3] > out_bytes) return MDX_ERR_ARG;
    }
    void *st_ = nullptr;
    int device = 0;
    if (mdx_ctx_stream(ctx, &st_, &device) != MDX_OK) return MDX_ERR_ARG;
    hipStream_t st = (hipStream_t)st_;
    if (hipSetDevice(device) != hipSuccess || mdx_k_gbam_prepare() != hipSuccess) return MDX_ERR_HIP;
    static mdx_crc32::Tables tables;
    static std::once_flag once;
    std::call_once(once, [] { mdx_crc32::make_tables(tables); });
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    uint32_t *d_blk = nullptr, *d_crc = nullptr;
    int *d_status = nullptr, *d_bad = nullptr;
    void *d_tab = nullptr;
    int rc = MDX_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess) rc = MDX_ERR_HIP; return e == hipSuccess; };
    // (one block per launch of the CRC kernel would be simplest to read back; instead every block is checked on its own
    // by giving the kernel one block at a time only when a check fails — the common case is one launch)
    if (ok(hipMalloc((void **)&d_comp, (size_t)comp_bytes + 64)) && ok(hipMalloc((void **)&d_out, (size_t)out_bytes + 64)) &&
        ok(hipMalloc((void **)&d_blk, (size_t)n_blocks * 16)) && ok(hipMalloc((void **)&d_status, (size_t)n_blocks * 4)) &&
        ok(hipMalloc((void **)&d_crc, (size_t)n_blocks * 4)) && ok(hipMalloc((void **)&d_bad, 4)) && ok(hipMalloc(&d_tab, sizeof(tables))) &&
        ok(hipMemcpyAsync(d_comp, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st)) &&
        ok(hipMemcpyAsync(d_blk, blk, (size_t)n_blocks * 16, hipMemcpyHostToDevice, st)) &&
        ok(hipMemcpyAsync(d_tab, &tables, sizeof(tables), hipMemcpyHostToDevice, st)) &&
        ok(hipMemsetAsync(d_out, 0xEE, (size_t)out_bytes + 64, st))) {
        mdx_k_gbam_inflate(d_comp, (const uint4 *)d_blk, n_blocks, d_out, d_status, st);
        if (ok(hipGetLastError()) && ok(hipMemcpyAsync(status, d_status, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, st)) &&
            ok(hipMemcpyAsync(out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st)) && want_crc && crc_ok) {
            // the CRC kernel reports the lowest failing block of a launch: launch it over the blocks behind each failure
            ok(hipMemcpyAsync(d_crc, want_crc, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
            int32_t from = 0;
            for (int32_t b = 0; b < n_blocks; b++) crc_ok[b] = 1;
            while (rc == MDX_OK && from < n_blocks) {
                const int no_bad = 0x7FFFFFFF;
                int bad = no_bad;
                if (!ok(hipMemcpyAsync(d_bad, &no_bad, 4, hipMemcpyHostToDevice, st))) break;
                mdx_k_gbam_crc(d_out, (const uint4 *)d_blk + from, d_crc + from, d_tab, n_blocks - from, d_bad, st);
                if (!ok(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) break;
                if (bad == no_bad) break;
                crc_ok[from + bad] = 0;
                from += bad + 1;
            }
        }
    }
    for (void *p : {(void *)d_comp, (void *)d_out, (void *)d_blk, (void *)d_status, (void *)d_crc, (void *)d_bad, d_tab}) if (p) (void)hipFree(p);
    return rc;
}

int mdx_gbam_set_min_basequal(mdx_gbam *g, int32_t minqual) {
    if (!g || minqual < 0 || minqual > 93) return MDX_ERR_ARG;
    // (the views carry the threshold in their nibbles — MDX_SEQ_4BITQ — and the packed masked kernel reads nothing else: a
    // decoder at one threshold in front of a context at another would be counted with the wrong mask, and nothing would notice)
    if (minqual != 0 && minqual != mdx_ctx_minqual(g->ctx)) {
        g->error = "mdx_gbam_set_min_basequal: " + std::to_string(minqual) + " is not the --min-basequal of the context the file was opened on (" +
                   std::to_string(mdx_ctx_minqual(g->ctx)) + ")";
        return MDX_ERR_ARG;
    }
    g->minqual = minqual;
    return MDX_OK;
}

int mdx_gbam_skip(mdx_gbam *g, int64_t chunk_bytes) {
    // the slab mdx_gbam_next would take now, left undecoded (same borders: the ranks of a multi-GPU run agree on them).
    // Where the first record behind it starts is then the device scan's guess — which the process that decodes the slab
    // in front checks against the truth (next_verified): if this process is that one and the guess would be wrong, it
    // says so here, before anybody counts a record twice or not at all.
    try {
        if (!g) return MDX_ERR_ARG;
        // (a handle that steps over slabs prepares none ahead of time: the slab behind the one in hand is another rank's; what
        // the first call has prepared is dropped)
        if (!g->no_ahead) {
            g->no_ahead = true;
            if (g->slab[0].ready || g->slab[1].ready) {
                (void)hipSetDevice(g->device);
                if (!g->drain()) return MDX_ERR_HIP;
                g->slab[0].ready = g->slab[1].ready = false;
            }
        }
        const size_t want = slab_want(g, g->next_block, chunk_bytes);
        if (!g->scan_to((g->next_block < g->blocks.size() ? g->blocks[g->next_block].in_off : g->scanned) + want + 65536)) return MDX_ERR_ARG;
        if (g->next_block >= g->blocks.size()) return MDX_OK;
        if (g->phase_known && !g->next_verified) {
            g->error = "the first record behind BGZF block " + std::to_string(g->next_block) + " cannot be found without the slab in front of it";
            return MDX_ERR_UNSUPPORTED;
        }
        const size_t b0 = g->next_block, in0 = g->blocks[b0].in_off;
        size_t b1 = b0, unc_bytes = 0;
        while (b1 < g->blocks.size() && (b1 == b0 || (g->blocks[b1].in_off - in0 < want && unc_bytes + g->blocks[b1].out_size < 0xE0000000ull))) {
            unc_bytes += g->blocks[b1].out_size;
            b1++;
        }
        g->next_block = b1;
        g->phase_known = false;
        g->phase = 0;
        return MDX_OK;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

int mdx_gbam_fixups(const mdx_gbam *g) { return g ? g->fixups : 0; }

int mdx_gbam_view_flags(mdx_gbam *g, uint16_t *flags, int64_t n) {
    if (!g || n < 0 || n != g->view_reads || (n > 0 && !flags)) return MDX_ERR_ARG;
    if (n == 0) return MDX_OK;
    if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
    if (hipMemcpyAsync(flags, g->slab[g->view_slab].flag.p, (size_t)n * 2, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        hipStreamSynchronize(g->stream) != hipSuccess) { g->error = "copy of the flag column failed"; return MDX_ERR_HIP; }
    return MDX_OK;
}

int mdx_gbam_view_set_flags(mdx_gbam *g, const uint16_t *flags, int64_t n) {
    if (!g || n < 0 || n != g->view_reads || (n > 0 && !flags)) return MDX_ERR_ARG;
    if (n == 0) return MDX_OK;
    if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
    // (synchronous: the caller's buffer is free again when the call returns)
    if (hipMemcpyAsync(g->slab[g->view_slab].flag.p, flags, (size_t)n * 2, hipMemcpyHostToDevice, g->stream) != hipSuccess ||
        hipStreamSynchronize(g->stream) != hipSuccess) { g->error = "copy of the flag column failed"; return MDX_ERR_HIP; }
    return MDX_OK;
}

int mdx_gbam_tell(const mdx_gbam *g, int64_t *comp_off, int64_t *phase) {
    if (!g || !comp_off || !phase || !g->hs) return MDX_ERR_ARG;
    if (!g->phase_known) return MDX_ERR_STATE;
    // (a Block knows where its DEFLATE payload starts: the block itself starts where the one in front ends, 8 bytes of
    // CRC32 and ISIZE behind that one's payload)
    const size_t k = g->next_block;
    *comp_off = k >= g->blocks.size() ? (int64_t)g->scanned : (k == 0 ? 0 : (int64_t)(g->blocks[k - 1].in_off + g->blocks[k - 1].in_size + 8));
    *phase = (int64_t)g->phase;
    return MDX_OK;
}

int mdx_gbam_set_seq_format(mdx_gbam *g, int32_t seq_format) {
    if (!g || (seq_format != MDX_SEQ_ASCII && seq_format != MDX_SEQ_4BIT)) return MDX_ERR_ARG;
    g->seq_format = seq_format;
    return MDX_OK;
}

int mdx_gbam_missing_qualities(const mdx_gbam *g) { return (g && g->no_qual_seen) ? 1 : 0; }

int mdx_gbam_at_end(const mdx_gbam *g) { return (!g || (g->next_block >= g->blocks.size() && g->whole_file_scanned())) ? 1 : 0; }

void mdx_gbam_close(mdx_gbam *g) {
    if (!g) return;
    const bool timing = std::getenv("MDX_BAM_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "mdx_gbam_close %-11s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    (void)hipSetDevice(g->device);
    (void)g->drain();
    lap("sync");
    // (arenas, CRC tables, copy stream and the host's buffer stay with the context / the process for the next file: the
    // context must still be alive — it owns the stream this handle works on)
    host_buffer_give(g->hbuf, g->hbuf_cap);
    for (auto &sl : g->slab) {
        if (sl.ev_infl0) (void)hipEventDestroy(sl.ev_infl0);
        if (sl.ev_infl) (void)hipEventDestroy(sl.ev_infl);
    }
    if (g->ev_free) (void)hipEventDestroy(g->ev_free);
    if (g->infl_stream) (void)hipStreamDestroy(g->infl_stream);
    pin_give(g->pin_bad);
    // (the larger arena to the context, the other one to the process's spare buffers)
    {
        const int big = g->slab[1].arena.cap > g->slab[0].arena.cap ? 1 : 0;
        mdx_ctx_scratch_give(g->ctx, g->slab[big].arena.p, g->slab[big].arena.cap, g->d_crc_tables, g->copy_stream, nullptr);
        arena_give(g->device, g->slab[big ^ 1].arena.p, g->slab[big ^ 1].arena.cap);
    }
    g->copy_stream = nullptr;
    rg_buffer_give(g->device, g->d_rg, g->d_rg_cap);
    for (int k = 0; k < 3; k++) pf_buffer_give(g->device, g->pf_buf[k], g->pf_cap[k]);
    lap("device");
    // (unmapping the file — a few hundred thousand touched pages — is 3-4 ms of an 8 M-record file's 63; done behind the
    // caller's back it holds the address space's lock against the next file's mmap and page faults, which then wait as long)
    if (g->hs) mdx_bam_close(g->hs);
    lap("file");
    delete g;
    lap("handle");
}

// The BGZF writer on the device (include/mdx.h mdx_bgzf_deflate): `data` cut into members of 0xFF00 bytes, every member in
// pieces, a lane per piece (mdx_gbam.hip), at most kBgzfBatch members per launch (their working memory: 300 KB each); the
// members are put together on the device — header, pieces, CRC32, ISIZE — side by side and copied back.  The buffers stay with
// the process (a few gigabytes of HBM: allocating them was a sixth of a call).
namespace {
struct BgzfBuffers {
    int device = -1, cap_members = 0;
    uint8_t *d_in = nullptr, *d_slots = nullptr, *d_out = nullptr;
    uint32_t *d_sizes = nullptr;
    unsigned long long *d_off = nullptr;
    void *d_scratch = nullptr, *d_tab = nullptr;
    void release() {
        for (void *p : {(void *)d_in, (void *)d_slots, (void *)d_out, (void *)d_sizes, (void *)d_off, d_scratch, d_tab}) if (p) (void)hipFree(p);
        d_in = d_slots = d_out = nullptr; d_sizes = nullptr; d_off = nullptr; d_scratch = d_tab = nullptr; cap_members = 0; device = -1;
    }
};
std::mutex g_bgzf_mu;
BgzfBuffers g_bgzf;
}  // namespace

// (d_data: the stream is in HBM already — no copy in; else `data` on the host)
static int bgzf_deflate_impl(hipStream_t st, int device, const uint8_t *data, const uint8_t *d_data, int64_t n, uint8_t *out, int64_t out_cap,
                             int64_t *out_len) {
    *out_len = 0;
    if (n == 0) return MDX_OK;
    if (hipSetDevice(device) != hipSuccess) return MDX_ERR_HIP;
    static mdx_crc32::Tables tables;
    static std::once_flag once;
    std::call_once(once, [] { mdx_crc32::make_tables(tables); });
    const int64_t total_members = (n + 0xFF00 - 1) / 0xFF00;
    const int kBgzfBatch = 8192;
    // (buffers for as many members as this call has, never fewer than a fifth of a launch's — a process that writes a header
    // first and a gigabyte behind it allocates once)
    const int batch = (int)std::min<int64_t>(std::max<int64_t>(total_members, kBgzfBatch / 4), kBgzfBatch);
    const int pieces = mdx_k_bgzf_pieces();
    int rc = MDX_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess) { rc = MDX_ERR_HIP; (void)hipGetLastError(); } return e == hipSuccess; };
    std::lock_guard<std::mutex> lk(g_bgzf_mu);
    BgzfBuffers &B = g_bgzf;
    try {
        if (B.device != device || B.cap_members < batch) {
            B.release();
            if (ok(hipMalloc((void **)&B.d_in, (size_t)batch * 0xFF00 + 64)) && ok(hipMalloc((void **)&B.d_slots, (size_t)batch * pieces * mdx_k_bgzf_slot_bytes())) &&
                ok(hipMalloc((void **)&B.d_out, (size_t)batch * 65536)) && ok(hipMalloc((void **)&B.d_sizes, (size_t)batch * pieces * 4)) &&
                ok(hipMalloc((void **)&B.d_off, (size_t)batch * 8)) && ok(hipMalloc(&B.d_scratch, mdx_k_bgzf_scratch_bytes(batch))) &&
                ok(hipMalloc(&B.d_tab, sizeof(tables))) && ok(hipMemcpy(B.d_tab, &tables, sizeof(tables), hipMemcpyHostToDevice))) {
                B.device = device; B.cap_members = batch;
            } else { B.release(); return MDX_ERR_HIP; }
        }
        std::vector<uint32_t> sizes((size_t)batch * (size_t)pieces);
        std::vector<unsigned long long> offs((size_t)batch);
        int64_t written = 0;
        for (int64_t b0 = 0; b0 < total_members && rc == MDX_OK; b0 += batch) {
            const int nb = (int)std::min<int64_t>(batch, total_members - b0);
            const int64_t lo = b0 * 0xFF00, bytes = std::min<int64_t>(n - lo, (int64_t)nb * 0xFF00);
            const uint8_t *src = d_data ? d_data + lo : B.d_in;
            if (!d_data && !ok(hipMemcpyAsync(B.d_in, data + lo, (size_t)bytes, hipMemcpyHostToDevice, st))) break;
            mdx_k_bgzf_deflate(src, bytes, nb, B.d_slots, B.d_sizes, B.d_scratch, st);
            if (!ok(hipGetLastError()) || !ok(hipMemcpyAsync(sizes.data(), B.d_sizes, (size_t)nb * pieces * 4, hipMemcpyDeviceToHost, st)) ||
                !ok(hipStreamSynchronize(st))) break;
            unsigned long long at = 0;
            for (int b = 0; b < nb; b++) {
                unsigned long long body = 0;
                for (int q = 0; q < pieces; q++) body += sizes[(size_t)b * pieces + q];
                if (body == 0 || body + 26 > 65536u) { rc = MDX_ERR_ARG; break; }
                offs[(size_t)b] = at; at += body + 26;
            }
            if (rc != MDX_OK) break;
            if (written + (int64_t)at > out_cap) { rc = MDX_ERR_ARG; break; }
            if (!ok(hipMemcpyAsync(B.d_off, offs.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st))) break;
            mdx_k_bgzf_gather(src, bytes, B.d_slots, B.d_sizes, B.d_off, nb, B.d_tab, B.d_out, st);
            if (!ok(hipGetLastError()) || !ok(hipMemcpyAsync(out + written, B.d_out, (size_t)at, hipMemcpyDeviceToHost, st)) ||
                !ok(hipStreamSynchronize(st))) break;
            written += (int64_t)at;
        }
        if (rc == MDX_OK) *out_len = written;
    } catch (...) {
        rc = MDX_ERR_ARG;
    }
    return rc;
}

int mdx_bgzf_deflate(mdx_ctx *ctx, const uint8_t *data, int64_t n, uint8_t *out, int64_t out_cap, int64_t *out_len) {
    if (!ctx || n < 0 || (n > 0 && !data) || !out || !out_len) return MDX_ERR_ARG;
    void *st_ = nullptr;
    int device = 0;
    if (mdx_ctx_stream(ctx, &st_, &device) != MDX_OK) return MDX_ERR_ARG;
    return bgzf_deflate_impl((hipStream_t)st_, device, data, nullptr, n, out, out_cap, out_len);
}

// The slab mdx_gbam_next handed out last, written back (include/mdx.h): patch list -> QUAL fields of the inflated records in
// HBM, sizes -> host -> offsets (a prefix sum of a few million numbers), records + MR tags to their places in an output stream
// in HBM, that stream through the device's BGZF writer; only the compressed members come back.
int mdx_gbam_write_rescaled(mdx_gbam *g, const uint64_t *d_patch, int64_t patch_cap, int32_t n_parts, const uint64_t *d_n_patch,
                            const float *mr, const uint8_t *rescaled, uint8_t *out, int64_t out_cap, int64_t *out_len, int64_t *mr_clash) {
    if (!g || !out || !out_len || patch_cap < 0 || (n_parts > 0 && (!d_patch || !d_n_patch))) return MDX_ERR_ARG;
    *out_len = 0;
    if (mr_clash) *mr_clash = -1;
    const int64_t n = g->view_reads;
    if (n == 0) return MDX_OK;
    if (!mr || !rescaled) return MDX_ERR_ARG;
    try {
        if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
        hipStream_t st = g->stream;
        mdx_gbam::Slab &s = g->slab[g->view_slab];
        uint8_t *d_resc = nullptr, *d_stream = nullptr;
        float *d_mr = nullptr;
        uint32_t *d_sizes = nullptr;
        unsigned long long *d_off = nullptr;
        int *d_clash = nullptr;
        int rc = MDX_OK;
        auto ok = [&](hipError_t e) { if (e != hipSuccess) { rc = MDX_ERR_HIP; g->error = std::string("write_rescaled: ") + hipGetErrorString(e); (void)hipGetLastError(); } return e == hipSuccess; };
        std::vector<uint32_t> sizes((size_t)n);
        std::vector<unsigned long long> offs((size_t)n);
        const int no_clash = 0x7FFFFFFF;
        int clash = no_clash;
        if (ok(hipMalloc((void **)&d_resc, (size_t)n)) && ok(hipMalloc((void **)&d_mr, (size_t)n * 4)) && ok(hipMalloc((void **)&d_sizes, (size_t)n * 4)) &&
            ok(hipMalloc((void **)&d_off, (size_t)n * 8)) && ok(hipMalloc((void **)&d_clash, 4)) &&
            ok(hipMemcpyAsync(d_resc, rescaled, (size_t)n, hipMemcpyHostToDevice, st)) && ok(hipMemcpyAsync(d_mr, mr, (size_t)n * 4, hipMemcpyHostToDevice, st)) &&
            ok(hipMemcpyAsync(d_clash, &no_clash, 4, hipMemcpyHostToDevice, st))) {
            if (n_parts > 0)
                mdx_k_gbam_patch_qual((uint8_t *)s.unc.p, (const uint32_t *)s.rec_off.p, (const uint32_t *)s.seq_off.p, (uint32_t)n,
                                      (const unsigned long long *)d_patch, (const unsigned long long *)d_n_patch, patch_cap, n_parts, st);
            mdx_k_gbam_out_sizes((const uint8_t *)s.unc.p, (const uint32_t *)s.rec_off.p, d_resc, (uint32_t)n, d_sizes, st);
            if (ok(hipGetLastError()) && ok(hipMemcpyAsync(sizes.data(), d_sizes, (size_t)n * 4, hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st))) {
                unsigned long long total = 0;
                for (int64_t i = 0; i < n; i++) { offs[(size_t)i] = total; total += sizes[(size_t)i]; }
                if (ok(hipMalloc((void **)&d_stream, (size_t)total + 64)) && ok(hipMemcpyAsync(d_off, offs.data(), (size_t)n * 8, hipMemcpyHostToDevice, st))) {
                    mdx_k_gbam_write_back((const uint8_t *)s.unc.p, (const uint32_t *)s.rec_off.p, d_off, d_resc, d_mr, (uint32_t)n, d_stream, d_clash, st);
                    if (ok(hipGetLastError()) && ok(hipMemcpyAsync(&clash, d_clash, 4, hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st))) {
                        if (clash != no_clash) {
                            if (mr_clash) *mr_clash = clash;
                            g->error = "record " + std::to_string(clash) + " of the slab is to be rescaled and has an MR tag already";
                            rc = MDX_ERR_BAD_READ;
                        } else {
                            rc = bgzf_deflate_impl(st, g->device, nullptr, d_stream, (int64_t)total, out, out_cap, out_len);
                            if (rc != MDX_OK) g->error = "the BGZF writer failed (output buffer too small?)";
                        }
                    }
                }
            }
        }
        for (void *p : {(void *)d_resc, (void *)d_mr, (void *)d_sizes, (void *)d_off, (void *)d_clash, (void *)d_stream}) if (p) (void)hipFree(p);
        return rc;
    } catch (const std::exception &e) {
        g->error = std::string("mdx_gbam_write_rescaled: ") + e.what();
        return MDX_ERR_ARG;
    }
}

// the name of record `index` of that slab (for the message of rescale.py:277-278); buf holds cap bytes
int mdx_gbam_record_name(mdx_gbam *g, int64_t index, char *buf, int32_t cap) {
    if (!g || !buf || cap < 1 || index < 0 || index >= g->view_reads) return MDX_ERR_ARG;
    buf[0] = 0;
    if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
    mdx_gbam::Slab &s = g->slab[g->view_slab];
    uint32_t off = 0;
    uint8_t head[12];
    if (hipMemcpy(&off, (const uint32_t *)s.rec_off.p + index, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(head, (const uint8_t *)s.unc.p + off, 12, hipMemcpyDeviceToHost) != hipSuccess) return MDX_ERR_HIP;
    const int len = std::min<int>(head[8], cap - 1);
    if (len > 0 && hipMemcpy(buf, (const uint8_t *)s.unc.p + off + 32, (size_t)len, hipMemcpyDeviceToHost) != hipSuccess) return MDX_ERR_HIP;
    buf[len] = 0;
    return MDX_OK;
}

// One call per slab of a --rescale-only pass on the device (rescale.py:285-365): the view mdx_gbam_next handed out last (it
// must bring qualities, mate columns and an ASCII seq column) through the context's rescale kernels in patch mode, MR rounded
// on the host (rescale.py:275-276, mdx_mr_round), the records written back (mdx_gbam_write_rescaled).
int mdx_gbam_rescale_slab(mdx_gbam *g, uint8_t *out, int64_t out_cap, int64_t *out_len, int64_t *counts, int64_t *mr_clash) {
    if (!g || !out || !out_len) return MDX_ERR_ARG;
    *out_len = 0;
    if (mr_clash) *mr_clash = -1;
    const int64_t n = g->view_reads;
    if (n == 0) return MDX_OK;
    const mdx_batch &v = g->last_view;
    if (!v.qual || !g->last_mtid || !g->last_mpos || v.seq_format != MDX_SEQ_ASCII) {
        g->error = "mdx_gbam_rescale_slab: the handle must be configured with qualities and mate columns, the seq column ASCII";
        return MDX_ERR_STATE;
    }
    try {
        if (hipSetDevice(g->device) != hipSuccess) return MDX_ERR_HIP;
        hipStream_t st = g->stream;
        const int parts = 256;
        const int64_t cap = std::max<int64_t>(4096, 8 * n / parts);
        uint64_t *d_patch = nullptr, *d_count = nullptr;
        double *d_mr = nullptr;
        uint8_t *d_status = nullptr;
        int rc = MDX_OK;
        auto ok = [&](hipError_t e) { if (e != hipSuccess) { rc = MDX_ERR_HIP; g->error = std::string("rescale_slab: ") + hipGetErrorString(e); (void)hipGetLastError(); } return e == hipSuccess; };
        std::vector<double> mr_raw((size_t)n);
        std::vector<uint8_t> status((size_t)n), rescaled((size_t)n);
        std::vector<float> mr((size_t)n);
        std::vector<uint64_t> cnt((size_t)parts);
        if (ok(hipMalloc((void **)&d_count, (size_t)parts * 8)) && ok(hipMalloc((void **)&d_mr, (size_t)n * 8)) && ok(hipMalloc((void **)&d_status, (size_t)n))) {
            // (eight entries of room per record, in 256 parts: this workload's records change 0.11 bytes each; a part that
            // overflows is an error — a second launch would count the records into the summary twice)
            if (ok(hipMalloc((void **)&d_patch, (size_t)parts * (size_t)cap * 8))) {
                rc = mdx_rescale_patches_device(g->ctx, &v, g->last_mtid, g->last_mpos, d_patch, cap, parts, d_count, d_mr, d_status);
                if (rc != MDX_OK) g->error = std::string("rescale kernels: ") + mdx_last_error(g->ctx);
                else if (ok(hipMemcpyAsync(cnt.data(), d_count, (size_t)parts * 8, hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st))) {
                    uint64_t fullest = 0;
                    for (uint64_t c_ : cnt) fullest = std::max(fullest, c_);
                    if ((int64_t)fullest > cap) { g->error = "rescale_slab: more rescaled bytes than the list holds"; rc = MDX_ERR_ARG; }
                }
            }
        }
        if (rc == MDX_OK) {
            int64_t bad = -1;
            rc = mdx_sync(g->ctx, &bad);
            if (rc != MDX_OK) { g->error = std::string("rescale kernels: ") + mdx_last_error(g->ctx); if (mr_clash && rc == MDX_ERR_BAD_READ) *mr_clash = -2 - bad; }
        }
        if (rc == MDX_OK && ok(hipMemcpyAsync(mr_raw.data(), d_mr, (size_t)n * 8, hipMemcpyDeviceToHost, st)) &&
            ok(hipMemcpyAsync(status.data(), d_status, (size_t)n, hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st))) {
            for (int64_t i = 0; i < n; i++) {
                const uint8_t stv = status[(size_t)i];
                rescaled[(size_t)i] = (stv == 2 || stv == 3) ? 1 : 0;
                if (counts && stv < 5) counts[stv]++;
            }
            (void)mdx_mr_round(mr_raw.data(), n, mr.data(), host_thread_budget());
            rc = mdx_gbam_write_rescaled(g, d_patch, cap, parts, d_count, mr.data(), rescaled.data(), out, out_cap, out_len, mr_clash);
        }
        for (void *p : {(void *)d_patch, (void *)d_count, (void *)d_mr, (void *)d_status}) if (p) (void)hipFree(p);
        return rc;
    } catch (const std::exception &e) {
        g->error = std::string("mdx_gbam_rescale_slab: ") + e.what();
        return MDX_ERR_ARG;
    }
}

int mdx_host_threads(void) { return host_thread_budget(); }
int mdx_host_pool_threads(void) { return (int)host_pool()->threads.size(); }

int mdx_warm(int32_t device, int64_t pinned_bytes) {
    // what a process pays once, whichever file comes first: the device's context, the decode kernels' code object, the
    // inflating threads, and the pinned buffer the host's share of a slab is inflated into (a few hundred megabytes: tens of
    // milliseconds to fault in and map) — on a thread of the caller's choice, beside whatever else the start of a run does
    try {
        // (MDX_INIT_TRACE=1: what each step took, on stderr)
        const bool trace = std::getenv("MDX_INIT_TRACE") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "mdx_warm %-16s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) { (void)hipGetLastError(); return MDX_ERR_HIP; }
        lap("context");
        if (mdx_k_gbam_prepare() != hipSuccess) { (void)hipGetLastError(); return MDX_ERR_HIP; }
        lap("decode kernels");
        (void)host_pool();
        lap("pool");
        if (pinned_bytes > 0) {
            uint8_t *p = nullptr;
            size_t cap = 0;
            host_buffer_take(p, cap);
            if (cap < (size_t)pinned_bytes) {
                if (p) (void)hipHostFree(p);
                p = nullptr; cap = 0;
                if (hipHostMalloc((void **)&p, (size_t)pinned_bytes, hipHostMallocDefault) == hipSuccess) cap = (size_t)pinned_bytes;
                else { (void)hipGetLastError(); p = nullptr; }
            }
            host_buffer_give(p, cap);
        }
        lap("pinned buffer");
        return MDX_OK;
    } catch (...) {
        return MDX_ERR_ARG;
    }
}

#endif  // MDX_HOST_ONLY

}  // extern "C"
