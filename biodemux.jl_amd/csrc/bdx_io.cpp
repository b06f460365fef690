// bdx_io.cpp — native host-side FASTQ reader / packer and in-order demultiplexing writer
// (libbdx_io.so, plain C ABI, no GPU code).  SURVEY.md §8(f) rank 1: the callers either side
// of the hot path.  Behaviour follows the reference's reader_task / writer_task
// (BioDemuX.jl src/core.jl:43-110, :118-224):
//   * a record is four readline() calls; readline strips "\n" and a preceding "\r"; a truncated
//     last record is padded with empty lines (readline at EOF returns "");
//   * records of one output file keep input order; files are opened in append mode; a path
//     ending in ".gz" (or gzip forced) is written through zlib;
//   * trimming applies to R1's sequence and quality only, clamped to the sequence length
//     (core.jl:162-173).
// The reference does this with one reader and one writer task; here line indexing, packing and
// the per-file gather are spread over host threads so the 1 G reads/s kernel is not starved more
// than the file system dictates.
#include <cerrno>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

// ---- persistent worker pool --------------------------------------------------------------------------------------
// The parallel sections of the reader (line index, packer) and of the writer (sizes, gather / iovecs, files) run on a
// pool that belongs to the CALLING thread (the pipeline's reader and writer threads each keep their own for the whole
// run) instead of starting and joining a set of std::threads per section: a 10 M-read run has ~140 sections per stage,
// i.e. ~2 200 thread starts at a few tens of microseconds each on the coordinating thread.  Tasks are claimed with an
// atomic counter; the caller works too.
class WorkPool {
  public:
    ~WorkPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_start_.notify_all();
        for (auto &t : th_) t.join();
    }
    template <class F>
    void run(int n, F &&f) {
        if (n <= 1) {
            if (n == 1) f(0);
            return;
        }
        while ((int)th_.size() < n - 1) th_.emplace_back([this]() { worker(); });
        std::function<void(int)> job = std::ref(f);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job;
            njobs_ = n;
            next_.store(0, std::memory_order_relaxed);
            pending_ = n;
            ++gen_;
        }
        cv_start_.notify_all();
        work(job, n);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this]() { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void work(const std::function<void(int)> &job, int n) {
        int finished = 0;
        for (;;) {
            const int i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) break;
            job(i);
            ++finished;
        }
        if (finished) {
            std::lock_guard<std::mutex> lk(mu_);
            pending_ -= finished;
            if (pending_ == 0) cv_done_.notify_all();
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_start_.wait(lk, [&]() { return stop_ || (gen_ != seen && job_ != nullptr); });
                if (stop_) return;
                seen = gen_;
                job = job_;
                n = njobs_;
            }
            // (the job object lives until pending_ reaches 0: a worker that arrives after the last task was claimed only
            // reads the counter)
            int finished = 0;
            for (;;) {
                const int i = next_.fetch_add(1, std::memory_order_relaxed);
                if (i >= n) break;
                (*job)(i);
                ++finished;
            }
            if (finished) {
                std::lock_guard<std::mutex> lk(mu_);
                pending_ -= finished;
                if (pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<int> next_{0};
    int njobs_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

template <class F>
static void parallel_for(int n, F &&f) {
    static thread_local WorkPool pool;
    pool.run(n, std::forward<F>(f));
}

extern "C" {

struct bdx_fq_file {
    const uint8_t *data = nullptr;
    int64_t size = 0;  // plain files: file size; .gz: bytes inflated so far once `state` != 0
    bool mapped = false;
    // .gz input: a background thread inflates into an address range reserved up
    // front (MAP_NORESERVE, so pages cost memory only once written) while the index / pack / classify /
    // write stages already work on the records that are there.  `avail` only grows; bytes below it
    // are final.
    bool streaming = false;
    size_t reserve = 0;
    std::thread inflater;
    std::atomic<int64_t> avail{0};
    std::atomic<int> state{0};    // 0 inflating, 1 complete, -1 failed
    std::atomic<bool> cancel{false};
    std::mutex mu;
    std::condition_variable cv;
    std::string err;
    double avg_record = 330.0;    // bytes per record, learned from the batches indexed so far
    bool parallel_members = false;  // the input was a size-tagged member chain inflated block-parallel
};


static thread_local std::string g_io_err;
const char *bdx_io_last_error(void) { return g_io_err.c_str(); }

// Compressed size of the gzip member that starts at c[0] if its header says so: BGZF ('B','C': 16-bit total
// size - 1) or this library's own tag ('D','X': 32-bit total size).  0: untagged (a plain gzip stream).
static size_t tagged_member_size(const uint8_t *c, size_t left) {
    if (left < 18 || c[0] != 0x1f || c[1] != 0x8b || c[2] != 8 || !(c[3] & 4)) return 0;
    const size_t xlen = (size_t)c[10] | ((size_t)c[11] << 8);
    if (12 + xlen > left) return 0;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
        const size_t len = (size_t)c[o + 2] | ((size_t)c[o + 3] << 8);
        if (o + 4 + len > 12 + xlen) return 0;
        if (c[o] == 'B' && c[o + 1] == 'C' && len == 2) return ((size_t)c[o + 4] | ((size_t)c[o + 5] << 8)) + 1;
        if (c[o] == 'D' && c[o + 1] == 'X' && len == 4)
            return (size_t)c[o + 4] | ((size_t)c[o + 5] << 8) | ((size_t)c[o + 6] << 16) | ((size_t)c[o + 7] << 24);
        o += 4 + len;
    }
    return 0;
}

// Block-parallel inflate of a size-tagged member chain (BGZF or this library's own output): the members'
// uncompressed sizes (ISIZE, the last 4 bytes of a member) give every member its place in the output, so T
// threads inflate them independently; `avail` advances over the finished prefix.  Returns the bytes produced
// and stops (`*stopped_at` = compressed offset) at the first member without a tag — the caller continues
// there with the serial stream.  -1 on a corrupt member.
static int64_t inflate_tagged_chain(bdx_fq_file *f, const uint8_t *comp, size_t csize, uint8_t *dst, size_t reserve,
                                    int nthreads, size_t *stopped_at) {
    struct Mem { size_t coff, clen, uoff, ulen; };
    size_t coff = 0, uoff = 0;
    int64_t result = 0;
    const size_t WAVE = 4096;  // members walked, inflated and published per round (bounds the bookkeeping)
    while (coff < csize && !f->cancel.load(std::memory_order_relaxed)) {
        std::vector<Mem> ms;
        while (coff < csize && ms.size() < WAVE) {
            const size_t clen = tagged_member_size(comp + coff, csize - coff);
            if (clen < 18 || coff + clen > csize) break;
            const uint8_t *t = comp + coff + clen - 4;
            const size_t ulen = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
            if (uoff + ulen + (1u << 22) > reserve) return -2;
            ms.push_back(Mem{coff, clen, uoff, ulen});
            coff += clen;
            uoff += ulen;
        }
        if (ms.empty()) break;
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        std::vector<std::atomic<char>> done(ms.size());
        for (auto &d : done) d.store(0);
        std::mutex pub;
        size_t published = 0;
        const int T = std::max(1, std::min<int>(nthreads, (int)ms.size()));
                parallel_for(T, [&](const int t) {
            (void)t;
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= ms.size() || bad.load() || f->cancel.load(std::memory_order_relaxed)) break;
                    const Mem &m = ms[i];
                    z_stream zs;
                    memset(&zs, 0, sizeof(zs));
                    if (inflateInit2(&zs, 15 + 16) != Z_OK) {
                        bad.store(1);
                        break;
                    }
                    zs.next_in = (Bytef *)(comp + m.coff);
                    zs.avail_in = (uInt)m.clen;
                    zs.next_out = dst + m.uoff;
                    zs.avail_out = (uInt)m.ulen;
                    const int rc = m.ulen ? inflate(&zs, Z_FINISH) : inflate(&zs, Z_FINISH);
                    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0 && zs.total_out == m.ulen;
                    inflateEnd(&zs);
                    if (!ok) {
                        bad.store(1);
                        break;
                    }
                    done[i].store(1, std::memory_order_release);
                    std::lock_guard<std::mutex> lk(pub);  // advance `avail` over the finished prefix
                    while (published < ms.size() && done[published].load(std::memory_order_acquire)) ++published;
                    if (published > 0) {
                        const Mem &l = ms[published - 1];
                        f->avail.store((int64_t)(l.uoff + l.ulen), std::memory_order_release);
                        f->cv.notify_all();
                    }
                }
            });
        if (bad.load()) return -1;
        result = (int64_t)uoff;
    }
    *stopped_at = coff;
    return result;
}

// Opens a FASTQ file: plain files are mmap'ed, ".gz" (case-insensitive, fileio.jl:78) is inflated
// into memory with zlib.  Returns 0 on success.
int32_t bdx_fq_open_mt(const char *path, int32_t nthreads_arg, bdx_fq_file **out) {
    *out = nullptr;
    auto *f = new bdx_fq_file();
    std::string p(path), low(p);
    std::transform(low.begin(), low.end(), low.begin(), ::tolower);
    const bool gz = low.size() >= 3 && low.compare(low.size() - 3, 3, ".gz") == 0;
    if (gz) {
        gzFile g = gzopen(path, "rb");
        if (!g) {
            g_io_err = "cannot open " + p;
            delete f;
            return -1;
        }
        gzbuffer(g, 1 << 20);
        struct stat st;
        const int64_t csize = stat(path, &st) == 0 ? (int64_t)st.st_size : 0;
        // FASTQ deflates 3..10x; reserve 64x the compressed size (at least 1 GiB) of address space
        f->reserve = (size_t)std::min<int64_t>(std::max<int64_t>(csize * 64, (int64_t)1 << 30), (int64_t)1 << 40);
        void *m = mmap(nullptr, f->reserve, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) {
            g_io_err = "cannot reserve address space for " + p;
            gzclose(g);
            delete f;
            return -1;
        }
        f->data = (const uint8_t *)m;
        f->streaming = true;
        const int nthreads = nthreads_arg > 0 ? nthreads_arg : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        f->inflater = std::thread([f, g0 = g, p, csize, nthreads]() {
            gzFile g = g0;
            uint8_t *dst = (uint8_t *)f->data;
            size_t used = 0;
            int st = 1;
            // size-tagged member chains (BGZF, this library's own .gz output) are inflated block-parallel;
            // an untagged member (an ordinary gzip stream) — at the start or anywhere later — is read serially
            if (csize > 0) {
                int fd = open(p.c_str(), O_RDONLY);
                void *cm = fd >= 0 ? mmap(nullptr, (size_t)csize, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
                if (fd >= 0) close(fd);
                if (cm != MAP_FAILED) {
                    size_t stopped = 0;
                    const int64_t got = tagged_member_size((const uint8_t *)cm, (size_t)csize)
                                            ? inflate_tagged_chain(f, (const uint8_t *)cm, (size_t)csize, dst, f->reserve, nthreads, &stopped)
                                            : 0;
                    munmap(cm, (size_t)csize);
                    if (got < 0) {
                        f->err = got == -2 ? "inflated size of " + p + " exceeds 64x its compressed size" : "corrupt gzip member in " + p;
                        st = -1;
                    } else {
                        used = (size_t)got;
                        f->parallel_members = got > 0;
                        if (stopped >= (size_t)csize) {
                            st = 2;  // everything was tagged: done
                        } else if (stopped > 0) {
                            // the rest is an ordinary gzip stream: a serial reader positioned at that member
                            gzclose(g);
                            int fd2 = open(p.c_str(), O_RDONLY);
                            g = (fd2 >= 0 && lseek(fd2, (off_t)stopped, SEEK_SET) == (off_t)stopped) ? gzdopen(fd2, "rb") : nullptr;
                            if (!g) {
                                if (fd2 >= 0) close(fd2);
                                f->err = "cannot continue reading " + p;
                                st = -1;
                            } else {
                                gzbuffer(g, 1 << 20);
                            }
                        }
                    }
                }
            }
            while (st == 1 && !f->cancel.load(std::memory_order_relaxed)) {
                if (f->reserve - used < (1u << 22)) {
                    f->err = "inflated size of " + p + " exceeds 64x its compressed size";
                    st = -1;
                    break;
                }
                const int got = gzread(g, dst + used, 1u << 22);
                if (got < 0) {
                    f->err = "gzread failed on " + p;
                    st = -1;
                    break;
                }
                if (got == 0) break;
                used += (size_t)got;
                f->avail.store((int64_t)used, std::memory_order_release);
                f->cv.notify_all();
            }
            if (g) gzclose(g);
            if (st == 2) st = 1;
            {
                std::lock_guard<std::mutex> lk(f->mu);
                f->size = (int64_t)used;
                f->state.store(st, std::memory_order_release);
            }
            f->cv.notify_all();
        });
    } else {
        int fd = open(path, O_RDONLY);
        if (fd < 0) {
            g_io_err = "cannot open " + p;
            delete f;
            return -1;
        }
        struct stat st;
        if (fstat(fd, &st) != 0) {
            g_io_err = "fstat failed on " + p;
            close(fd);
            delete f;
            return -1;
        }
        f->size = (int64_t)st.st_size;
        if (f->size > 0) {
            void *m = mmap(nullptr, (size_t)f->size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                g_io_err = "mmap failed on " + p;
                close(fd);
                delete f;
                return -1;
            }
            madvise(m, (size_t)f->size, MADV_SEQUENTIAL);
            f->data = (const uint8_t *)m;
            f->mapped = true;
        }
        close(fd);
    }
    *out = f;
    return 0;
}

int32_t bdx_fq_open(const char *path, bdx_fq_file **out) { return bdx_fq_open_mt(path, 0, out); }

void bdx_fq_close(bdx_fq_file *f) {
    if (!f) return;
    if (f->streaming) {
        f->cancel.store(true);
        if (f->inflater.joinable()) f->inflater.join();
        if (f->data) munmap((void *)f->data, f->reserve);
    } else if (f->mapped && f->data) {
        munmap((void *)f->data, (size_t)f->size);
    }
    delete f;
}

// Blocks until at least `target` bytes are there or the stream has ended; returns the bytes available
// and whether that is the whole file (*final).  Plain files: immediately.
static int64_t wait_available(bdx_fq_file *f, int64_t target, bool *final, bool *failed) {
    *failed = false;
    if (!f->streaming) {
        *final = true;
        return f->size;
    }
    std::unique_lock<std::mutex> lk(f->mu);
    f->cv.wait_for(lk, std::chrono::milliseconds(50), [&]() { return f->state.load() != 0 || f->avail.load() >= target; });
    while (f->state.load() == 0 && f->avail.load() < target) f->cv.wait_for(lk, std::chrono::milliseconds(50));
    const int st = f->state.load(std::memory_order_acquire);
    *final = st != 0;
    *failed = st < 0;
    return st != 0 ? f->size : f->avail.load(std::memory_order_acquire);
}

// 1 when the .gz input turned out to be a size-tagged member chain that was inflated block-parallel
int32_t bdx_fq_parallel_inflate(bdx_fq_file *f) {
    bool fin, bad;
    (void)wait_available(f, INT64_MAX, &fin, &bad);
    return f->parallel_members ? 1 : 0;
}

const uint8_t *bdx_fq_data(const bdx_fq_file *f) { return f->data; }
// Total size (for .gz: blocks until the stream is completely inflated).
int64_t bdx_fq_size(bdx_fq_file *f) {
    bool fin, bad;
    int64_t s = wait_available(f, INT64_MAX, &fin, &bad);
    return bad ? -1 : s;
}
// The records below `upto` have been written out: give their pages back (streamed .gz only).
void bdx_fq_release(bdx_fq_file *f, int64_t upto) {
    if (!f->streaming || upto <= 0) return;
    const int64_t page = 1 << 12;
    const int64_t n = (upto / page) * page;
    if (n > 0) madvise((void *)f->data, (size_t)n, MADV_DONTNEED);
}

// Line index of the bytes [start, size): fills line_off[k] / line_len[k] for up to 4*max_reads
// lines (len excludes "\n" and a preceding "\r").  Returns the number of RECORDS (a trailing
// partial record counts; its missing lines get offset = size, len = 0) and stores the cursor
// after the last consumed line in *next.  nthreads chunks scan for '\n' in parallel.
// `final` == false (streamed input, more bytes will follow): only complete records are returned.
static int64_t index_range(const uint8_t *d, const int64_t size, const bool final, int64_t start, int64_t max_reads,
                           int64_t *line_off, int32_t *line_len, int64_t *next, int32_t nthreads) {
    if (start >= size || max_reads <= 0) {
        *next = start;
        return 0;
    }
    const int64_t want = 4 * max_reads;
    // Two phases per scanned region, both spread over the threads: (1) every thread notes the newlines of its slice
    // (32-bit offsets from the slice start: half the memory traffic of 64-bit positions); (2) after a prefix sum over
    // the slices' counts every thread turns its own newlines straight into line_off / line_len entries — no merged
    // position vector and no serial pass over millions of lines.  Regions are scanned until enough lines are there.
    int64_t nlines = 0;          // lines written so far
    int64_t cur = start;         // start of the next line
    int64_t scanned_to = start;
    int64_t region = std::min<int64_t>(size - start, std::max<int64_t>(1 << 20, max_reads * 400));
    while (nlines < want && scanned_to < size) {
        // (newline positions are 32-bit offsets from the slice start: a slice stays below 4 GiB whatever `region` has
        // doubled to on input with very long lines)
        const int64_t hi = std::min(size, scanned_to + std::min<int64_t>(region, (int64_t)std::max(1, nthreads) * 0xFFFF0000LL));
        const int T = std::max(1, std::min<int>(nthreads, (int)std::min<int64_t>((hi - scanned_to) >> 20, 1 << 20) + 1));
        const int64_t span = (hi - scanned_to + T - 1) / T;
        // (the slices' newline vectors are kept by the calling thread across calls — the pipeline's reader thread indexes every
        // batch: a fresh 28 MB of them per 2^19-read batch would be page-faulted in again each time)
        static thread_local std::vector<std::vector<uint32_t>> pos_keep;
        if (pos_keep.size() < (size_t)T) pos_keep.resize((size_t)T);
        std::vector<std::vector<uint32_t>> &pos = pos_keep;
        std::vector<int64_t> cnt((size_t)T, 0);
        {
                        parallel_for(T, [&](const int t) {
                (void)t;
                    const int64_t a = std::min(hi, scanned_to + span * t), b = std::min(hi, a + span);
                    // (a slice has at most one newline per byte; FASTQ lines are long, so a bound that is usually
                    // generous is tried first and the slice is rescanned with the exact count if it does not hold)
                    int64_t cap = (b - a) / 24 + 64;
                    for (;;) {
                        std::vector<uint32_t> &v = pos[(size_t)t];
                        if ((int64_t)v.size() < cap) v.resize((size_t)cap);
                        cap = (int64_t)v.size();
                        const uint8_t *p = d + a, *e = d + b;
                        int64_t n = 0;
                        bool fits = true;
                        while (p < e) {
                            const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
                            if (!q) break;
                            if (n < cap) v[(size_t)n] = (uint32_t)(q - (d + a));
                            else fits = false;
                            ++n;
                            p = q + 1;
                        }
                        if (fits) {
                            cnt[(size_t)t] = n;
                            break;
                        }
                        cap = n;
                    }
                });
        }
        // prefix over the slices: first line index and the end of the last line before every slice
        std::vector<int64_t> first((size_t)T + 1, 0), prev_end((size_t)T, 0);
        int64_t pe = cur;  // start of the line that the first newline of the region terminates
        for (int t = 0; t < T; ++t) {
            first[(size_t)t + 1] = first[(size_t)t] + cnt[(size_t)t];
            prev_end[(size_t)t] = pe;
            if (cnt[(size_t)t] > 0) {
                const int64_t a = std::min(hi, scanned_to + span * t);
                pe = a + (int64_t)pos[(size_t)t][(size_t)cnt[(size_t)t] - 1] + 1;
            }
        }
        const int64_t room = want - nlines;
        const int64_t take = std::min<int64_t>(first[(size_t)T], room);
        {
                        parallel_for(T, [&](const int t) {
                (void)t;
                    const int64_t a = std::min(hi, scanned_to + span * t);
                    int64_t lcur = prev_end[(size_t)t];
                    const int64_t k0 = first[(size_t)t], k1 = std::min<int64_t>(first[(size_t)t + 1], take);
                    for (int64_t k = k0; k < k1; ++k) {
                        const int64_t e = a + (int64_t)pos[(size_t)t][(size_t)(k - k0)];
                        int64_t len = e - lcur;
                        line_off[nlines + k] = lcur;
                        if (len > 0 && d[e - 1] == '\r') len -= 1;
                        line_len[nlines + k] = (int32_t)len;
                        lcur = e + 1;
                    }
                });
        }
        if (take > 0) {
            // the cursor after the last line taken
            int t_last = 0;
            while (t_last + 1 < T && first[(size_t)t_last + 1] < take) ++t_last;
            const int64_t a = std::min(hi, scanned_to + span * t_last);
            cur = a + (int64_t)pos[(size_t)t_last][(size_t)(take - 1 - first[(size_t)t_last])] + 1;
        }
        nlines += take;
        scanned_to = hi;
        region *= 2;
    }
    if (!final) {  // the rest of a record may still be on its way: only complete records
        const int64_t drop = nlines % 4;
        if (drop) {
            nlines -= drop;
            cur = line_off[nlines];
        }
    }
    // data not terminated by '\n': the rest is one more line (readline at EOF)
    if (final && nlines < want && scanned_to >= size && cur < size) {
        int64_t len = size - cur;
        line_off[nlines] = cur;
        if (len > 0 && d[size - 1] == '\r') len -= 1;
        line_len[nlines] = (int32_t)len;
        nlines += 1;
        cur = size;
    }
    const int64_t nrec = (nlines + 3) / 4;
    for (int64_t k = nlines; k < 4 * nrec; ++k) {  // pad a truncated last record with empty lines
        line_off[k] = size;
        line_len[k] = 0;
    }
    *next = cur;
    return nrec;
}

int64_t bdx_fq_index(bdx_fq_file *f, int64_t start, int64_t max_reads, int64_t *line_off, int32_t *line_len,
                     int64_t *next, int32_t nthreads) {
    if (!f->streaming) return index_range(f->data, f->size, true, start, max_reads, line_off, line_len, next, nthreads);
    double want_bytes = (double)max_reads * f->avg_record * 1.02 + 4096.0;
    for (;;) {
        bool fin, bad;
        const int64_t have = wait_available(f, start + (int64_t)want_bytes, &fin, &bad);
        if (bad) {
            g_io_err = f->err;
            *next = start;
            return -1;
        }
        const int64_t n = index_range(f->data, have, fin, start, max_reads, line_off, line_len, next, nthreads);
        if (n >= max_reads || fin) {
            if (n > 0) f->avg_record = (double)(*next - start) / (double)n;
            return n;
        }
        want_bytes = want_bytes * 1.25 + 65536.0;  // records longer than estimated: wait for more
    }
}

// Packs the sequence lines (line 1 of every record) into the C-ABI chunk layout:
// seq_off[0] = 0, seq_off[i+1] = seq_off[i] + len_i; bytes copied in parallel.
void bdx_fq_pack(const bdx_fq_file *f, const int64_t *line_off, const int32_t *line_len, int64_t nrec,
                 uint8_t *seq_bytes, int64_t *seq_off, int32_t nthreads) {
    seq_off[0] = 0;
    for (int64_t i = 0; i < nrec; ++i) seq_off[i + 1] = seq_off[i] + line_len[4 * i + 1];
    const int T = std::max(1, std::min<int>(nthreads, (int)(nrec >> 14) + 1));
        const int64_t per = (nrec + T - 1) / T;
    parallel_for(T, [=](const int t) {
        (void)t;
            const int64_t a = per * t, b = std::min(nrec, a + per);
            for (int64_t i = a; i < b; ++i)
                memcpy(seq_bytes + seq_off[i], f->data + line_off[4 * i + 1], (size_t)line_len[4 * i + 1]);
        });
}

int64_t bdx_fq_seq_bytes(const int32_t *line_len, int64_t nrec) {
    int64_t s = 0;
    for (int64_t i = 0; i < nrec; ++i) s += line_len[4 * i + 1];
    return s;
}

// One output stream of a batch: records of file `src` (with its line tables) are appended to
// class_paths[cls[i]] in input order.  trim != 0: keep_start/keep_end (1-based inclusive, -1 =
// untrimmed) are applied to sequence and quality (core.jl:162-173).  force_gzip: config.gzip_output.
// bdx_fq_demux_write_range: only the records whose class lies in [class_lo, class_hi) are written — several writer
// threads can then share a batch (each output file belongs to exactly one of them, so per-file order stays input
// order): a file is written by one thread at a time whatever the file system, and the file of the unmatched reads
// is a tenth of the bytes.
static int32_t demux_write_impl(const bdx_fq_file *src, const int64_t *line_off, const int32_t *line_len, int64_t nrec,
                                const int32_t *cls, int32_t n_classes, const char *const *class_paths,
                                const int32_t *keep_start, const int32_t *keep_end, int32_t trim, int32_t force_gzip,
                                int32_t nthreads, int32_t class_lo, int32_t class_hi);

int32_t bdx_fq_demux_write(const bdx_fq_file *src, const int64_t *line_off, const int32_t *line_len, int64_t nrec,
                           const int32_t *cls, int32_t n_classes, const char *const *class_paths,
                           const int32_t *keep_start, const int32_t *keep_end, int32_t trim, int32_t force_gzip,
                           int32_t nthreads) {
    return demux_write_impl(src, line_off, line_len, nrec, cls, n_classes, class_paths, keep_start, keep_end, trim, force_gzip, nthreads, 0,
                            n_classes);
}

int32_t bdx_fq_demux_write_range(const bdx_fq_file *src, const int64_t *line_off, const int32_t *line_len, int64_t nrec,
                                 const int32_t *cls, int32_t n_classes, const char *const *class_paths,
                                 const int32_t *keep_start, const int32_t *keep_end, int32_t trim, int32_t force_gzip,
                                 int32_t nthreads, int32_t class_lo, int32_t class_hi) {
    return demux_write_impl(src, line_off, line_len, nrec, cls, n_classes, class_paths, keep_start, keep_end, trim, force_gzip, nthreads,
                            class_lo, class_hi);
}

static int32_t demux_write_impl(const bdx_fq_file *src, const int64_t *line_off, const int32_t *line_len, int64_t nrec,
                                const int32_t *cls, int32_t n_classes, const char *const *class_paths,
                                const int32_t *keep_start, const int32_t *keep_end, int32_t trim, int32_t force_gzip,
                                int32_t nthreads, int32_t class_lo, int32_t class_hi) {
    const uint8_t *d = src->data;
    const auto mine = [=](int32_t c) { return c >= class_lo && c < class_hi; };
    if (nrec <= 0) return 0;
    static const bool timing = getenv("BDX_IO_TIMING") != nullptr;  // developer switch: phase times of the writer on stderr
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tm0 = now();
    double tm1 = 0, tm2 = 0, tm3 = 0;
    // Plain files are written in one of two ways, chosen per class and batch:
    //  * few bytes (the barcode files: ~3 MB each per 2^20-read batch): the untouched records are handed to the kernel
    //    as iovecs pointing INTO the input mapping — one copy (input pages -> the file's pages), no gather buffer;
    //  * many bytes (the file of the unmatched reads: a tenth of a batch), trimmed / "\r"-stripped records, gzip: the
    //    records are gathered into a heap buffer by all threads, and the buffer goes out with ONE write (a writev over
    //    130 000 small iovecs is bound by the per-segment cost of the kernel's copy loop, and a file is written by one
    //    thread at a time whatever the file system).
    // Per-file order is input order in both: positions come from a prefix sum over the threads' record slices.
    const int T = std::max(1, std::min<int>(nthreads, (int)(nrec >> 13) + 1));
    const int64_t per = (nrec + T - 1) / T;
    // record i: trimmed range (1-based inclusive a..b within the sequence; b = -1: untrimmed)
    const auto trimmed = [&](int64_t i, int64_t &a, int64_t &b) {
        if (!trim || keep_start[i] == -1) {
            a = 1;
            b = -1;
            return;
        }
        const int64_t slen = line_len[4 * i + 1];
        a = std::max<int64_t>(keep_start[i], 1);
        b = std::min<int64_t>(keep_end[i], slen);
        if (a > b) {
            a = 1;
            b = 0;
        }
    };
    const auto out_bytes = [&](int64_t i) -> int64_t {
        int64_t sl = line_len[4 * i + 1], ql = line_len[4 * i + 3];
        int64_t a, b;
        trimmed(i, a, b);
        if (b != -1) {
            sl = b - a + 1;
            const int64_t qb = std::min<int64_t>(b, ql);  // the quality string is sliced with the same range, clamped to its own length
            ql = qb >= a ? qb - a + 1 : 0;
        }
        return line_len[4 * i] + sl + line_len[4 * i + 2] + ql + 4;
    };
    // one run of the input ("h\ns\np\nq\n", nothing stripped)?
    // (the terminator of the last line: lines are split at '\n' only, so when the NEXT record's first line starts one byte
    // behind this record's last line that byte is the '\n' — read off the line table, which the pass streams through anyway;
    // looking at the byte itself costs a cache miss into the input per record: 40 of the writer's 220 ms per 10 M reads.
    // The batch's last record has no successor in the table and looks.)
    const auto contiguous = [&](int64_t i) -> bool {
        const int64_t o0 = line_off[4 * i], o1 = line_off[4 * i + 1], o2 = line_off[4 * i + 2], o3 = line_off[4 * i + 3];
        if (!(o1 == o0 + line_len[4 * i] + 1 && o2 == o1 + line_len[4 * i + 1] + 1 && o3 == o2 + line_len[4 * i + 2] + 1)) return false;
        const int64_t e3 = o3 + line_len[4 * i + 3];
        if (i + 1 < nrec && line_off[4 * i + 4] > e3) return line_off[4 * i + 4] == e3 + 1;
        return e3 < src->size && d[e3] == '\n';
    };
    // pass 1: bytes and records per class and thread; does every record of a class qualify for the iovec form?
    std::vector<std::vector<int64_t>> tbytes((size_t)T, std::vector<int64_t>((size_t)n_classes, 0));
    std::vector<std::vector<int64_t>> trecs((size_t)T, std::vector<int64_t>((size_t)n_classes, 0));
    std::vector<std::vector<char>> tplain((size_t)T, std::vector<char>((size_t)n_classes, 1));
    std::atomic<int> bad_class{0};
    {
                parallel_for(T, [&](const int t) {
            (void)t;
                const int64_t a0 = per * t, b0 = std::min(nrec, a0 + per);
                auto &h = tbytes[(size_t)t];
                auto &r = trecs[(size_t)t];
                auto &pl = tplain[(size_t)t];
                for (int64_t i = a0; i < b0; ++i) {
                    const int32_t c = cls[i];
                    if (c < 0 || c >= n_classes) {
                        bad_class.store(1);
                        return;
                    }
                    if (!mine(c)) continue;
                    h[(size_t)c] += out_bytes(i);
                    r[(size_t)c] += 1;
                    if (pl[(size_t)c]) {
                        int64_t a, b;
                        trimmed(i, a, b);
                        if (b != -1 || !contiguous(i)) pl[(size_t)c] = 0;
                    }
                }
            });
    }
    if (bad_class.load()) {
        g_io_err = "class index out of range";
        return -1;
    }
    tm1 = now();
    std::vector<int64_t> csize((size_t)n_classes, 0), crecs((size_t)n_classes, 0);
    std::vector<char> cplain((size_t)n_classes, 1);
    for (int c = 0; c < n_classes; ++c)
        for (int t = 0; t < T; ++t) {
            const int64_t v = tbytes[(size_t)t][(size_t)c], r = trecs[(size_t)t][(size_t)c];
            tbytes[(size_t)t][(size_t)c] = csize[(size_t)c];  // -> where thread t's records of class c start (bytes / records)
            trecs[(size_t)t][(size_t)c] = crecs[(size_t)c];
            csize[(size_t)c] += v;
            crecs[(size_t)c] += r;
            cplain[(size_t)c] = cplain[(size_t)c] && tplain[(size_t)t][(size_t)c];
        }
    std::vector<int> todo;
    for (int c = 0; c < n_classes; ++c)
        if (csize[(size_t)c] > 0) todo.push_back(c);
    struct Dest {
        uint8_t *base = nullptr;      // gather buffer, or
        struct iovec *iov = nullptr;  // the records as iovecs into the input
        bool gz = false;
        int fail = 0;
    };
    std::vector<Dest> dest((size_t)n_classes);
    // (gather buffers and iovec arrays are kept by the calling thread — the pipeline's writer thread — across batches: 7 MB of
    // iovecs and the 21 MB buffer of the unmatched reads per 2^19-read batch would be allocated and page-faulted in again)
    static thread_local std::vector<std::vector<uint8_t>> heap_keep;
    static thread_local std::vector<std::vector<struct iovec>> iov_keep;
    if (heap_keep.size() < (size_t)n_classes) heap_keep.resize((size_t)n_classes);
    if (iov_keep.size() < (size_t)n_classes) iov_keep.resize((size_t)n_classes);
    static const int64_t IOV_LIMIT = getenv("BDX_IO_IOV_LIMIT") ? atoll(getenv("BDX_IO_IOV_LIMIT")) : (int64_t)8 << 20;  // classes with more bytes than this in a batch are gathered
    for (int c : todo) {
        Dest &ds = dest[(size_t)c];
        std::string low(class_paths[c]);
        std::transform(low.begin(), low.end(), low.begin(), ::tolower);
        ds.gz = force_gzip || (low.size() >= 3 && low.compare(low.size() - 3, 3, ".gz") == 0);
        if (!ds.gz && cplain[(size_t)c] && csize[(size_t)c] <= IOV_LIMIT) {
            std::vector<struct iovec> &v = iov_keep[(size_t)c];
            if (v.size() < (size_t)crecs[(size_t)c]) v.resize((size_t)crecs[(size_t)c]);
            ds.iov = v.data();
        } else {
            std::vector<uint8_t> &v = heap_keep[(size_t)c];
            if (v.size() < (size_t)csize[(size_t)c]) v.resize((size_t)csize[(size_t)c]);
            ds.base = v.data();
        }
    }
    const bool any_fail = false;
    tm2 = now();
    // pass 2: the gather (heap classes) / the iovecs (the others)
    {
                parallel_for(T, [&](const int t) {
            (void)t;
                const int64_t a0 = per * t, b0 = std::min(nrec, a0 + per);
                auto &pos = tbytes[(size_t)t];
                auto &rpos = trecs[(size_t)t];
                for (int64_t i = a0; i < b0; ++i) {
                    const int32_t c = cls[i];
                    if (!mine(c)) continue;
                    Dest &ds = dest[(size_t)c];
                    if (ds.iov) {
                        struct iovec &v = ds.iov[(size_t)rpos[(size_t)c]++];
                        v.iov_base = (void *)(d + line_off[4 * i]);
                        v.iov_len = (size_t)(line_off[4 * i + 3] + line_len[4 * i + 3] + 1 - line_off[4 * i]);
                        continue;
                    }
                    uint8_t *o = ds.base + pos[(size_t)c];
                    const int64_t hl = line_len[4 * i], pl = line_len[4 * i + 2];
                    int64_t so = line_off[4 * i + 1], sl = line_len[4 * i + 1];
                    int64_t qo = line_off[4 * i + 3], ql = line_len[4 * i + 3];
                    int64_t a, b;
                    trimmed(i, a, b);
                    if (b == -1 && contiguous(i)) {  // one run of the input: one copy
                        const int64_t len = hl + sl + pl + ql + 4;
                        memcpy(o, d + line_off[4 * i], (size_t)len);
                        pos[(size_t)c] += len;
                        continue;
                    }
                    if (b != -1) {
                        so += a - 1;
                        sl = b - a + 1;
                        const int64_t qb = std::min<int64_t>(b, ql);
                        qo += a - 1;
                        ql = qb >= a ? qb - a + 1 : 0;
                    }
                    memcpy(o, d + line_off[4 * i], (size_t)hl);
                    o += hl;
                    *o++ = '\n';
                    memcpy(o, d + so, (size_t)sl);
                    o += sl;
                    *o++ = '\n';
                    memcpy(o, d + line_off[4 * i + 2], (size_t)pl);
                    o += pl;
                    *o++ = '\n';
                    memcpy(o, d + qo, (size_t)ql);
                    o += ql;
                    *o++ = '\n';
                    pos[(size_t)c] += hl + sl + pl + ql + 4;
                }
            });
    }
    tm3 = now();
    // pass 3: gzip output — every 4 MiB piece of every class is deflated as a gzip member of its own by whichever thread
    // is free (members concatenate to a valid .gz; one big class — `unknown` — would otherwise serialise the whole batch
    // behind one zlib stream), then the members of a file are appended in order.
    struct Member {
        int c;
        size_t off, len;
        std::vector<uint8_t> out;
        int bad = 0;
    };
    std::vector<Member> members;
    if (!any_fail)
        for (int c : todo)
            if (dest[(size_t)c].gz) {
                const size_t total = (size_t)csize[(size_t)c], piece = (size_t)1 << 22;
                for (size_t off = 0; off < total; off += piece) {
                    Member mb;
                    mb.c = c;
                    mb.off = off;
                    mb.len = std::min(piece, total - off);
                    members.push_back(std::move(mb));
                }
            }
    if (!members.empty()) {
        std::atomic<size_t> nextm{0};
        const int TM = std::max(1, std::min<int>(nthreads, (int)members.size()));
        parallel_for(TM, [&](const int) {
                for (;;) {
                    const size_t i = nextm.fetch_add(1);
                    if (i >= members.size()) break;
                    Member &mb = members[i];
                    z_stream zs;
                    memset(&zs, 0, sizeof(zs));
                    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                        mb.bad = 1;
                        continue;
                    }
                    // Every member carries its own compressed size in a gzip "extra" subfield ('D','X', 4 bytes LE;
                    // the BGZF idea with a 32-bit size): any gzip reader skips it, bdx_fq_open uses it to walk
                    // the member chain and inflate the members in parallel.
                    static unsigned char extra[8] = {'D', 'X', 4, 0, 0, 0, 0, 0};
                    gz_header hd;
                    memset(&hd, 0, sizeof(hd));
                    hd.os = 255;
                    hd.extra = extra;
                    hd.extra_len = 8;
                    deflateSetHeader(&zs, &hd);
                    mb.out.resize(deflateBound(&zs, (uLong)mb.len) + 64);
                    zs.next_in = (Bytef *)(dest[(size_t)mb.c].base + mb.off);
                    zs.avail_in = (uInt)mb.len;
                    zs.next_out = mb.out.data();
                    zs.avail_out = (uInt)mb.out.size();
                    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) mb.bad = 1;
                    mb.out.resize(mb.out.size() - zs.avail_out);
                    deflateEnd(&zs);
                    // header: 10 fixed bytes, XLEN (2), then the subfield SI1 SI2 LEN(2) DATA(4) -> data at 16..19
                    if (!mb.bad && mb.out.size() >= 20 && (mb.out[3] & 4) && mb.out[12] == 'D' && mb.out[13] == 'X') {
                        const uint32_t cs = (uint32_t)mb.out.size();
                        for (int b = 0; b < 4; ++b) mb.out[16 + b] = (uint8_t)(cs >> (8 * b));
                    } else {
                        mb.bad = 1;
                    }
                }
            });
    }
    // finish: the plain files (largest first: the unmatched reads' buffer is one long write) and the members of the
    // gzip files (consecutive in `members`), each file appended by one thread
    {
        std::vector<int> order(todo);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return csize[(size_t)x] > csize[(size_t)y]; });
        std::atomic<size_t> nextc{0};
        const int TW = std::max(1, std::min<int>(nthreads, (int)order.size()));
        parallel_for(TW, [&](const int) {
                for (;;) {
                    const size_t k = nextc.fetch_add(1);
                    if (k >= order.size()) break;
                    const int c = order[k];
                    Dest &ds = dest[(size_t)c];
                    if (ds.gz) {
                        FILE *fp = fopen(class_paths[c], "ab");
                        if (!fp) {
                            ds.fail = 1;
                            continue;
                        }
                        for (const Member &mb : members)
                            if (mb.c == c && (mb.bad || fwrite(mb.out.data(), 1, mb.out.size(), fp) != mb.out.size())) ds.fail = 1;
                        if (fclose(fp) != 0) ds.fail = 1;
                        continue;
                    }
                    const int fd = open(class_paths[c], O_WRONLY | O_APPEND | O_CREAT, 0644);
                    if (fd < 0) {
                        ds.fail = 1;
                        continue;
                    }
                    if (ds.iov) {
                        struct iovec *v = ds.iov;
                        int64_t left = crecs[(size_t)c];
                        while (left > 0 && !ds.fail) {
                            const int n = (int)std::min<int64_t>(left, 1024);  // IOV_MAX
                            ssize_t w = writev(fd, v, n);
                            if (w < 0 && errno == EINTR) continue;  // interrupted before anything went out: again
                            if (w <= 0) {  // (0 bytes of a non-empty request would spin forever)
                                ds.fail = 1;
                                break;
                            }
                            int done = 0;  // (a short write: skip what went out, retry the rest)
                            while (done < n && w >= (ssize_t)v[done].iov_len) {
                                w -= (ssize_t)v[done].iov_len;
                                ++done;
                            }
                            if (done < n && w > 0) {
                                v[done].iov_base = (char *)v[done].iov_base + w;
                                v[done].iov_len -= (size_t)w;
                            }
                            v += done;
                            left -= done;
                        }
                    } else {
                        const uint8_t *p = ds.base;
                        int64_t left = csize[(size_t)c];
                        while (left > 0) {
                            const ssize_t w = write(fd, p, (size_t)left);
                            if (w < 0 && errno == EINTR) continue;
                            if (w <= 0) {
                                ds.fail = 1;
                                break;
                            }
                            p += w;
                            left -= w;
                        }
                    }
                    if (close(fd) != 0) ds.fail = 1;
                }
            });
    }
    if (timing)
        fprintf(stderr, "[bdx_io] writer: sizes %.1f ms, buffers %.1f ms, gather / iovecs %.1f ms, deflate + files %.1f ms (%lld records)\n",
                (tm1 - tm0) * 1e3, (tm2 - tm1) * 1e3, (tm3 - tm2) * 1e3, (now() - tm3) * 1e3, (long long)nrec);
    for (int c : todo)
        if (dest[(size_t)c].fail) {
            g_io_err = std::string("cannot write ") + class_paths[c];
            return -1;
        }
    return 0;
}

}  // extern "C"
