// kmerust_host.cpp -- reader, builder, writers and KMIX index above the C ABI.  See kmerust_host.h
// for the reference lines each piece mirrors.  Counting is always the HIP path (kh_*).
#include "kmerust_host.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace kmerust {

// =============================================================================================
// format / input
// =============================================================================================
static std::string lower_ext(const std::string &name) {
    const size_t slash = name.find_last_of('/');
    const std::string base = slash == std::string::npos ? name : name.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return "";
    std::string e = base.substr(dot + 1);
    for (char &c : e) c = (char)std::tolower((unsigned char)c);
    return e;
}

static std::string strip_ext(const std::string &name) {
    const size_t slash = name.find_last_of('/');
    const size_t dot = name.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return name;
    return name.substr(0, dot);
}

SequenceFormat format_from_extension(const std::string &path) {
    std::string ext = lower_ext(path);
    if (ext == "gz") ext = lower_ext(strip_ext(path));  // strip .gz, look at the real extension
    if (ext == "fq" || ext == "fastq") return SequenceFormat::Fastq;
    return SequenceFormat::Fasta;  // known FASTA extensions and anything unknown
}

SequenceFormat resolve_format(SequenceFormat f, const std::string *path) {
    if (f != SequenceFormat::Auto) return f;
    return path ? format_from_extension(*path) : SequenceFormat::Fasta;
}

const char *format_name(SequenceFormat f) {
    switch (f) {
    case SequenceFormat::Auto: return "auto";
    case SequenceFormat::Fasta: return "fasta";
    default: return "fastq";
    }
}

// =============================================================================================
// reader
// =============================================================================================
namespace {

// Buffered line reader over zlib (reads plain files transparently) or stdin.
class LineSource {
public:
    explicit LineSource(const std::string &path) : path_(path) {
        if (is_stdin_path(path)) {
            gz_ = gzdopen(0, "rb");
        } else {
            gz_ = gzopen(path.c_str(), "rb");
        }
        if (!gz_) throw Error("failed to read sequence file '" + path + "': " + std::strerror(errno));
        gzbuffer(gz_, 1u << 20);
        buf_.resize(1u << 22);
    }
    ~LineSource() {
        if (gz_) gzclose(gz_);
    }
    // Next line without its terminator ('\n', optional '\r' and trailing blanks removed).
    bool next(std::string &line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos_ == len_) {
                const int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
                if (n < 0) {
                    int en = 0;
                    const char *msg = gzerror(gz_, &en);
                    throw Error("failed to decompress gzip file '" + path_ + "': " + (msg ? msg : "read error"));
                }
                if (n == 0) break;
                pos_ = 0;
                len_ = (size_t)n;
            }
            const char *p = buf_.data() + pos_;
            const char *nl = (const char *)memchr(p, '\n', len_ - pos_);
            any = true;
            if (nl) {
                line.append(p, (size_t)(nl - p));
                pos_ = (size_t)(nl - buf_.data()) + 1;
                break;
            }
            line.append(p, len_ - pos_);
            pos_ = len_;
        }
        if (!any) return false;
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        return true;
    }

private:
    std::string path_;
    gzFile gz_ = nullptr;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
};

struct BatchBuilder {
    Batch b;
    bool want_qual;
    size_t limit;
    const BatchSink &sink;
    uint64_t total_records = 0;
    BatchBuilder(bool wq, size_t lim, const BatchSink &s) : want_qual(wq), limit(lim), sink(s) {}
    void end_record() {
        b.bases.push_back('\n');
        if (want_qual) b.qual.push_back('\n');
        b.records++;
        total_records++;
        if (b.bases.size() >= limit) flush();
    }
    void flush() {
        if (b.records) sink(b);
        b.bases.clear();
        b.qual.clear();
        b.records = 0;
    }
};

}  // namespace

uint64_t read_sequences(const std::string &path, SequenceFormat fmt, bool want_qual, size_t batch_bytes,
                        const BatchSink &sink) {
    const SequenceFormat f = resolve_format(fmt, is_stdin_path(path) ? nullptr : &path);
    LineSource src(path);
    const bool fastq = f == SequenceFormat::Fastq;
    if (!fastq) want_qual = false;  // FASTA carries no qualities (reader.rs:68-79)
    BatchBuilder bb(want_qual, batch_bytes, sink);
    std::string line;
    if (!fastq) {
        bool in_record = false;
        while (src.next(line)) {
            if (!line.empty() && line[0] == '>') {
                if (in_record) bb.end_record();
                in_record = true;
            } else if (!in_record) {
                if (line.empty()) continue;
                throw Error("failed to parse sequence record: expected '>' at record start");
            } else {
                bb.b.bases.insert(bb.b.bases.end(), line.begin(), line.end());
            }
        }
        if (in_record) bb.end_record();
    } else {
        bool have = src.next(line);
        while (have) {
            if (line.empty()) {  // blank lines between records
                have = src.next(line);
                continue;
            }
            if (line[0] != '@') throw Error("failed to parse sequence record: expected '@' at record start");
            const size_t seq_start = bb.b.bases.size();
            bool plus = false;
            while ((have = src.next(line))) {
                if (!line.empty() && line[0] == '+') {
                    plus = true;
                    break;
                }
                bb.b.bases.insert(bb.b.bases.end(), line.begin(), line.end());
            }
            if (!plus) throw Error("failed to parse sequence record: incomplete FASTQ record (no '+' line)");
            const size_t seq_len = bb.b.bases.size() - seq_start;
            size_t got = 0;
            while (got < seq_len && (have = src.next(line))) {
                if (want_qual) bb.b.qual.insert(bb.b.qual.end(), line.begin(), line.end());
                got += line.size();
            }
            if (got != seq_len)
                throw Error("failed to parse sequence record: unequal length of sequence and qualities");
            bb.end_record();
            have = src.next(line);
        }
    }
    bb.flush();
    return bb.total_records;
}

// =============================================================================================
// counting session over the C ABI
// =============================================================================================
Timing &timing() {
    static Timing t;
    return t;
}

// The command line is one count per process: releasing tens of gigabytes of device memory allocation by allocation
// before exiting (0.03 - 0.27 s measured) buys nothing.  Library users (KmerCounter in a long-lived process) never set this.
bool &leak_at_exit() {
    static bool v = false;
    return v;
}

namespace {
double wall_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Lap {  // adds the scope's wall time to one field of timing()
    double &dst;
    const double t0;
    explicit Lap(double &d) : dst(d), t0(wall_s()) {}
    ~Lap() { dst += wall_s() - t0; }
};
}  // namespace

struct Session {
    // One context per device.  More than one device (KmerCounter::devices, `kmerust --gpus N`): a kh_group --
    // chunks of whole records go to the devices in turn, every device counts its share into its own table, and
    // kh_group_merge (RCCL all-to-all of region segments + LDS merge, inside the library) turns the N tables into
    // one table sharded by hash range; results are then read shard by shard.
    std::vector<kh_ctx *> ctxs;
    kh_group *group = nullptr;
    kh_ctx *ctx = nullptr;  // ctxs[0]
    uint32_t k;
    size_t next_ctx = 0;
    // input_bytes: the size of the (plain) file about to be counted, 0 = unknown -- kh_config::input_mib
    Session(const KmerCounter &kc, bool use_qual, uint64_t input_bytes = 0) : k((uint32_t)kc.k_) {
        if (!kc.k_set_) throw Error("k-mer length not set");
        Lap lap(timing().create_s);
        kh_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.struct_size = sizeof(cfg);
        cfg.k = (uint32_t)kc.k_;
        cfg.min_quality = use_qual ? kc.min_quality_ : -1;
        cfg.device = kc.device_;
        cfg.capacity_hint = kc.capacity_hint_;
        cfg.input_mib = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (input_bytes + (1u << 20) - 1) >> 20);
        // (a chunk's record scan runs beside the next chunk's transfer; a refusal then comes one call late -- count_file_text
        //  starts the whole file over on a refusal anyway)
        if (const char *e = getenv("KMERUST_DEFER_SCAN"); !(e && e[0] == '0')) cfg.flags |= KH_FLAG_DEFER_TEXT_SCAN;
        if (kc.devices_.size() > 1) {
            std::vector<int32_t> devs(kc.devices_.begin(), kc.devices_.end());
            const int rc = kh_group_create(&group, &cfg, devs.data(), (uint32_t)devs.size());
            if (rc != KH_OK) throw Error(std::string("kh_group_create: ") + kh_strerror(rc));
            for (uint32_t i = 0; i < kh_group_size(group); ++i) ctxs.push_back(kh_group_ctx(group, i));
        } else {
            if (kc.devices_.size() == 1) cfg.device = kc.devices_[0];
            kh_ctx *c = nullptr;
            check_on(nullptr, kh_create(&c, &cfg), "kh_create");
            ctxs.push_back(c);
        }
        ctx = ctxs[0];
    }
    ~Session() {
        Lap lap(timing().destroy_s);
        if (leak_at_exit() && !group) return;  // (the process is about to end: the driver reclaims everything at once)
        if (group) kh_group_destroy(group);
        else if (ctx) kh_destroy(ctx);
    }
    static void check_on(kh_ctx *c, int rc, const char *what) {
        if (rc == KH_OK) return;
        std::string msg = std::string(what) + ": " + kh_strerror(rc);
        if (c && kh_last_error(c)[0]) msg += std::string(" (") + kh_last_error(c) + ")";
        throw Error(msg);
    }
    void check(int rc, const char *what) const { check_on(ctx, rc, what); }
    void reset_all() {
        for (kh_ctx *c : ctxs) check_on(c, kh_reset(c), "kh_reset");
    }
    // Uncompressed or gzip files: the TEXT goes to the device in chunks of whole records and the
    // records are found there (kh_push_text).  Returns false -- with the table(s) reset -- when the
    // device scanner refuses the layout; the line parser below then takes the file (and reports
    // malformed input with the reference's messages).
    bool count_file_text(const std::string &path, SequenceFormat fmt) {
        const bool fastq = fmt == SequenceFormat::Fastq;
        // plain files are read() straight into the chunk buffer; gzip (magic 1f 8b) goes through zlib
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error("failed to read sequence file '" + path + "': " + std::strerror(errno));
        unsigned char magic[2] = {0, 0};
        const bool is_gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        gzFile gz = is_gz ? gzdopen(fd, "rb") : nullptr;
        if (is_gz && !gz) {
            close(fd);
            throw Error("failed to read sequence file '" + path + "': " + std::strerror(errno));
        }
        struct Closer {
            gzFile g;
            int fd;
            ~Closer() {
                if (g) gzclose(g);  // closes fd too
                else close(fd);
            }
        } closer{gz, fd};
        if (gz) gzbuffer(gz, 1u << 20);
        auto read_some = [&](uint8_t *dst, size_t want) -> size_t {  // 0 = end of file
            if (gz) {
                const int n = gzread(gz, dst, (unsigned)std::min<size_t>(want, 1u << 30));
                if (n < 0) {
                    int en = 0;
                    const char *msg = gzerror(gz, &en);
                    throw Error("failed to decompress gzip file '" + path + "': " + (msg ? msg : "read error"));
                }
                return (size_t)n;
            }
            for (;;) {
                const ssize_t n = read(fd, dst, std::min<size_t>(want, (size_t)1 << 30));
                if (n >= 0) return (size_t)n;
                if (errno != EINTR) throw Error("failed to read sequence file '" + path + "': " + std::strerror(errno));
            }
        };
        // One worker per device: the reader fills a chunk buffer while the devices scan and count the previous
        // chunks (with one device this is the plain read -> push loop on the caller's thread).
        const size_t ndev = ctxs.size();
        struct Worker {
            std::thread th;
            std::mutex m;
            std::condition_variable cv;
            std::vector<uint8_t> job;  // whole records handed over by the reader
            bool has_job = false, quit = false;
            int rc = KH_OK;
            kh_ctx *ctx = nullptr;  // the device context this worker pushes into: its kh_last_error explains `rc`
        };
        std::vector<std::unique_ptr<Worker>> workers;
        const int text_fmt = fastq ? KH_TEXT_FASTQ : KH_TEXT_FASTA;
        if (ndev > 1)
            for (size_t d = 0; d < ndev; ++d) {
                workers.emplace_back(new Worker());
                Worker *w = workers.back().get();
                kh_ctx *c = ctxs[d];
                w->ctx = c;
                w->th = std::thread([w, c, text_fmt] {
                    for (;;) {
                        std::unique_lock<std::mutex> lk(w->m);
                        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
                        if (!w->has_job) return;
                        lk.unlock();
                        const int rc = kh_push_text(c, w->job.data(), w->job.size(), text_fmt);
                        lk.lock();
                        if (rc != KH_OK && w->rc == KH_OK) w->rc = rc;
                        w->has_job = false;
                        w->cv.notify_all();
                    }
                });
            }
        kh_ctx *failed_ctx = nullptr;       // the context whose push failed (the message must be ITS kh_last_error, not device 0's)
        auto stop_workers = [&]() -> int {  // waits for the queued chunks; first error of any device
            int rc = KH_OK;
            for (auto &w : workers) {
                {
                    std::unique_lock<std::mutex> lk(w->m);
                    w->cv.wait(lk, [&] { return !w->has_job; });
                    w->quit = true;
                    w->cv.notify_all();
                    if (rc == KH_OK && w->rc != KH_OK) {
                        rc = w->rc;
                        failed_ctx = w->ctx;
                    }
                }
                w->th.join();
            }
            workers.clear();
            return rc;
        };
        struct Joiner {  // (exceptions from the reader must not leave threads behind)
            std::function<int()> &f;
            ~Joiner() { f(); }
        };
        std::function<int()> stopper = stop_workers;
        Joiner joiner{stopper};

        // (128 MiB chunks.  The chunk buffers are pinned, and pinning runs at ~5 GB/s: two 256 MiB buffers cost ~0.1 s of every
        //  run, two of 128 MiB half that.  Measured on a 31.6 GB FASTQ (tools/cli_s100m_probe.py): the pushes take 0.75-0.78 s at
        //  64, 128 and 256 MiB alike -- 41 GB/s, what one host buffer DMAs at; a chunk's scan runs beside the next chunk's
        //  transfer (KH_FLAG_DEFER_TEXT_SCAN) -- so the smaller buffers win: 1.21 s against 1.32 s of wall time.  The size of a
        //  chunk no longer decides the size of a counting batch: since round 4 the library accumulates the scanned chunks on the
        //  device and counts tens of GB at a time -- input.hip, scan_text.)
        const size_t chunk = text_chunk_bytes();
        // The chunk buffer is PINNED memory (kh_host_alloc): kh_push_text then DMAs from it -- no staging memcpy inside the
        // library -- and a plain file is read into it by several pread() calls side by side (one read() moves ~6 GB/s
        // out of the page cache, less than the device scans and counts).  If pinned memory cannot be had the buffer is
        // ordinary memory and everything still works, through the library's staging.
        struct ChunkBuf {
            uint8_t *p = nullptr;
            size_t cap = 0;
            bool pinned = false;
            ~ChunkBuf() { release(); }
            void release() {
                if (p) {
                    if (pinned) (void)kh_host_free(p);
                    else free(p);
                }
                p = nullptr;
                cap = 0;
            }
            void reserve(size_t n, size_t keep) {  // grows to n bytes, keeping the first `keep`
                if (n <= cap) return;
                void *q = nullptr;
                Lap lap(timing().buffers_s);
                bool pin = kh_host_alloc(&q, n) == KH_OK;
                if (!pin) q = malloc(n);
                if (!q) throw Error("out of memory for the text chunk buffer");
                if (keep) memcpy(q, p, keep);
                release();
                p = (uint8_t *)q;
                cap = n;
                pinned = pin;
            }
            uint8_t *data() { return p; }
            size_t size() const { return cap; }
        } buf, buf2;  // (buf2: the chunk being filled while `buf` is pushed -- plain files on one device, KMERUST_PIPELINED_READER=0)
        size_t have = 0;
        bool eof = false, pushed = false;
        // plain files: where the next byte comes from, and how many there are (pread needs no shared cursor)
        off_t file_pos = 0, file_size = 0;
        bool can_pread = false;
        if (!gz) {
            struct stat sb;
            if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
                file_size = sb.st_size;
                can_pread = true;
            }
        }
        auto read_parallel = [&](uint8_t *dst, size_t want) -> size_t {  // 0 = end of file
            const size_t left = (size_t)(file_size - file_pos);
            want = std::min(want, left);
            if (want == 0) return 0;
            const size_t min_part = (size_t)8 << 20;
            const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(read_threads(), want / min_part));
            const size_t per = (want + parts - 1) / parts;
            std::vector<std::thread> th;
            std::vector<int> errs(parts, 0);
            auto part = [&](unsigned i) {
                size_t off = (size_t)i * per, end = std::min(want, off + per);
                while (off < end) {
                    const ssize_t n = pread(fd, dst + off, end - off, file_pos + (off_t)off);
                    if (n > 0) off += (size_t)n;
                    else if (n == 0 || errno != EINTR) {
                        errs[i] = n == 0 ? EIO : errno;  // (the file shrank under us, or a real error)
                        return;
                    }
                }
            };
            try {
                for (unsigned i = 1; i < parts; ++i) th.emplace_back(part, i);
            } catch (...) {  // (no thread: the caller's thread reads those parts too)
                for (auto &t : th) t.join();
                for (unsigned i = (unsigned)th.size() + 1; i < parts; ++i) part(i);
                th.clear();
            }
            part(0);
            for (auto &t : th) t.join();
            for (int e : errs)
                if (e) throw Error("failed to read sequence file '" + path + "': " + std::strerror(e));
            file_pos += (off_t)want;
            return want;
        };
        auto refuse = [&]() {
            (void)stop_workers();
            if (pushed) reset_all();
            return false;
        };
        // ---- one device, a plain file: a READER THREAD runs ahead of the pushes through three chunk buffers (round 5) ----
        // Round 4 read chunk i + 1 beside the push of chunk i and joined the two after every chunk: an iteration took the longer
        // of the two plus the hand-over (thread start and join, the search for the record boundary, the tail's copy) -- 3.35 ms
        // per 128 MiB where the transfer alone takes 2.3 (tools/ubench/h2d_probe.hip: 57 GB/s from such buffers, 50 with twelve
        // threads writing the neighbouring buffer).  Now nothing but the queue sits between two kh_push_text calls; the reader
        // cuts the chunks (the boundary search, the carry of the tail) and pins the second and third buffer itself.
        if (ndev == 1 && can_pread && pipelined_reader()) {
            constexpr int NB = 3;
            ChunkBuf bufs[NB];
            bufs[0].reserve(chunk, 0);
            struct Item {
                int idx;
                size_t cut;
                bool last;
            };
            std::mutex qm;
            std::condition_variable qcv;
            std::deque<Item> ready;
            bool busy[NB] = {false, false, false};
            bool stop = false, too_long = false;
            std::exception_ptr rd_err;
            std::thread reader([&] {
                try {
                    size_t fill = 0;
                    int cur = 0;
                    bool at_end = false;
                    for (;;) {
                        while (!at_end && fill < bufs[cur].size()) {
                            const double t0 = wall_s();
                            const size_t n = read_parallel(bufs[cur].data() + fill, bufs[cur].size() - fill);
                            timing().read_s += wall_s() - t0;
                            timing().bytes_read += n;
                            if (n == 0) at_end = true;
                            fill += n;
                        }
                        size_t cut = fill;
                        if (!at_end) {
                            cut = fastq ? fastq_cut(bufs[cur].data(), fill) : fasta_cut(bufs[cur].data(), fill);
                            if (cut == 0) {  // no record boundary in a whole chunk: grow and read on
                                if (bufs[cur].size() >= (size_t)8 << 30) {
                                    std::lock_guard<std::mutex> lk(qm);
                                    too_long = true;
                                    qcv.notify_all();
                                    return;
                                }
                                bufs[cur].reserve(bufs[cur].size() * 2, fill);
                                continue;
                            }
                        }
                        const int nxt = (cur + 1) % NB;
                        if (!at_end) {  // the tail behind the last whole record opens the next chunk
                            {
                                std::unique_lock<std::mutex> lk(qm);
                                qcv.wait(lk, [&] { return !busy[nxt] || stop; });
                                if (stop) return;
                            }
                            bufs[nxt].reserve(std::max(chunk, bufs[cur].size()), 0);  // (a buffer that had to grow for a long record: its tail can be as long)
                            memcpy(bufs[nxt].data(), bufs[cur].data() + cut, fill - cut);
                        }
                        {
                            std::lock_guard<std::mutex> lk(qm);
                            busy[cur] = true;
                            ready.push_back(Item{cur, cut, at_end});
                            qcv.notify_all();
                        }
                        if (at_end) return;
                        fill -= cut;
                        cur = nxt;
                    }
                } catch (...) {
                    std::lock_guard<std::mutex> lk(qm);
                    rd_err = std::current_exception();
                    qcv.notify_all();
                }
            });
            auto end_reader = [&]() {
                {
                    std::lock_guard<std::mutex> lk(qm);
                    stop = true;
                    qcv.notify_all();
                }
                if (reader.joinable()) reader.join();
            };
            try {
                for (;;) {
                    Item it{0, 0, true};
                    {
                        std::unique_lock<std::mutex> lk(qm);
                        qcv.wait(lk, [&] { return !ready.empty() || rd_err || too_long; });
                        if (ready.empty()) break;  // (the reader gave up: its reason is looked at below)
                        it = ready.front();
                        ready.pop_front();
                    }
                    if (it.cut) {
                        timing().chunks++;
                        int rc;
                        {
                            Lap lap(timing().push_s);
                            rc = kh_push_text(ctx, bufs[it.idx].data(), it.cut, text_fmt);
                        }
                        if (rc == KH_ERR_FORMAT) {
                            end_reader();
                            return refuse();
                        }
                        check(rc, "kh_push_text");
                        pushed = true;
                    }
                    {
                        std::lock_guard<std::mutex> lk(qm);
                        busy[it.idx] = false;
                        qcv.notify_all();
                    }
                    if (it.last) break;
                }
            } catch (...) {
                end_reader();
                throw;
            }
            end_reader();
            if (rd_err) std::rethrow_exception(rd_err);
            if (too_long) return refuse();
            eof = true;  // (skips the loop below: on to kh_finish)
        }
        if (!eof) buf.reserve(chunk, 0);
        while (!eof) {
            while (have < buf.size()) {
                    size_t n;
                {
                    Lap lap(timing().read_s);
                    n = can_pread ? read_parallel(buf.data() + have, buf.size() - have) : read_some(buf.data() + have, buf.size() - have);
                    timing().bytes_read += n;
                }
                if (n == 0) {
                    eof = true;
                    break;
                }
                have += n;
            }
            size_t cut = have;
            if (!eof) {
                cut = fastq ? fastq_cut(buf.data(), have) : fasta_cut(buf.data(), have);
                if (cut == 0) {  // no record boundary in a whole chunk: grow and read on
                    if (buf.size() >= (size_t)8 << 30) return refuse();
                    buf.reserve(buf.size() * 2, have);
                    continue;
                }
            }
            if (cut) timing().chunks++;
            if (cut && ndev == 1 && can_pread && !eof) {
                // The next chunk is read (tail of this one first) while the device takes this one: kh_push_text blocks for
                // the transfer, the record scan and the counting of the chunk.
                buf2.reserve(buf.size(), 0);
                const size_t tail = have - cut;
                size_t got = 0;
                std::exception_ptr rd_err;
                double rd_s = 0;
                std::thread reader([&] {
                    const double t0 = wall_s();
                    try {
                        memcpy(buf2.data(), buf.data() + cut, tail);
                        got = read_parallel(buf2.data() + tail, buf2.size() - tail);
                    } catch (...) {
                        rd_err = std::current_exception();
                    }
                    rd_s = wall_s() - t0;
                });
                int rc;
                {
                    Lap lap(timing().push_s);
                    rc = kh_push_text(ctx, buf.data(), cut, text_fmt);
                }
                reader.join();
                timing().read_s += rd_s;  // (overlapped with push_s: the two no longer add up to the wall time)
                timing().bytes_read += got;
                if (rd_err) std::rethrow_exception(rd_err);
                if (rc == KH_ERR_FORMAT) return refuse();
                check(rc, "kh_push_text");
                pushed = true;
                std::swap(buf.p, buf2.p);
                std::swap(buf.cap, buf2.cap);
                std::swap(buf.pinned, buf2.pinned);
                have = tail + got;
                if (got == 0) eof = true;  // (what is left in the buffer is pushed by the next round, as the last chunk)
                if (eof && have == 0) break;
                if (eof) {  // the last chunk: whole records up to the end of the file
                    timing().chunks++;
                    Lap lap(timing().push_s);
                    const int rc2 = kh_push_text(ctx, buf.data(), have, text_fmt);
                    if (rc2 == KH_ERR_FORMAT) return refuse();
                    check(rc2, "kh_push_text");
                    have = 0;
                    break;
                }
                continue;
            }
            if (cut && ndev == 1) {
                Lap lap(timing().push_s);
                const int rc = kh_push_text(ctx, buf.data(), cut, text_fmt);
                if (rc == KH_ERR_FORMAT) return refuse();
                check(rc, "kh_push_text");
                pushed = true;
            } else if (cut) {
                Worker *w = workers[next_ctx++ % ndev].get();
                std::unique_lock<std::mutex> lk(w->m);
                w->cv.wait(lk, [&] { return !w->has_job; });
                if (w->rc == KH_ERR_FORMAT) {
                    lk.unlock();
                    return refuse();
                }
                if (w->rc != KH_OK) {
                    const int rc = w->rc;
                    kh_ctx *const wc = w->ctx;
                    lk.unlock();
                    (void)stop_workers();
                    check_on(wc, rc, "kh_push_text");
                }
                w->job.assign(buf.data(), buf.data() + cut);
                w->has_job = true;
                w->cv.notify_all();
                pushed = true;
            }
            memmove(buf.data(), buf.data() + cut, have - cut);
            have -= cut;
        }
        int rc = stop_workers();
        // the last chunk's scan (KH_FLAG_DEFER_TEXT_SCAN) has not given its verdict yet: kh_finish brings it
        for (size_t i = 0; i < ctxs.size() && rc == KH_OK; ++i) {
            Lap lap(timing().finish_s);
            rc = kh_finish(ctxs[i], nullptr);
            if (rc != KH_OK && rc != KH_ERR_FORMAT) failed_ctx = ctxs[i];
        }
        if (rc == KH_ERR_FORMAT) {
            if (pushed) reset_all();
            return false;
        }
        check_on(failed_ctx ? failed_ctx : ctx, rc, "kh_push_text");
        return true;
    }
    static unsigned read_threads() {
        const char *e = getenv("KMERUST_READ_THREADS");
        if (e && atoi(e) > 0) return (unsigned)atoi(e);
        // (a chunk's read runs beside the previous chunk's push, ~10 ms per 256 MiB since the library counts one text under the
        //  next one's copy: four pread()s side by side take as long, eight leave a margin -- measured, profiles/README.md r03b)
        const unsigned hw = std::thread::hardware_concurrency();
        return std::max(1u, std::min(12u, hw ? hw : 1u));
    }
    static bool pipelined_reader() {  // KMERUST_PIPELINED_READER=0: round 4's read-beside-push loop (A/B)
        const char *e = getenv("KMERUST_PIPELINED_READER");
        return !(e && e[0] == '0');
    }
    static size_t text_chunk_bytes() {
        const char *e = getenv("KMERUST_TEXT_CHUNK_KB");  // tests use small chunks to exercise the cuts
        return (e && atol(e) > 0 ? (size_t)atol(e) : (size_t)128 << 10) << 10;  // (128 MiB: see count_file_text)
    }
    // Largest p > 0 with a record starting at p ('>' at a line start); 0 if none.
    static size_t fasta_cut(const uint8_t *b, size_t n) {
        for (size_t p = n; p-- > 1;)
            if (b[p] == '>' && b[p - 1] == '\n') return p;
        return 0;
    }
    // Largest p > 0 where a complete 4-line record provably starts: '@' at a line start, '+' two
    // lines on, and |seq| == |qual| (a quality line may itself start with '@').
    static size_t fastq_cut(const uint8_t *b, size_t n) {
        auto line_end = [&](size_t from) -> size_t {  // index of the '\n' ending the line at `from`, or n
            const void *q = from < n ? memchr(b + from, '\n', n - from) : nullptr;
            return q ? (size_t)((const uint8_t *)q - b) : n;
        };
        for (size_t p = n; p-- > 1;) {
            if (b[p] != '@' || b[p - 1] != '\n') continue;
            const size_t e0 = line_end(p);
            if (e0 >= n) continue;
            const size_t e1 = line_end(e0 + 1);
            if (e1 >= n) continue;
            const size_t e2 = line_end(e1 + 1);
            if (e2 >= n || b[e1 + 1] != '+') continue;
            const size_t e3 = line_end(e2 + 1);
            if (e3 >= n) continue;
            size_t ls = e1 - (e0 + 1), lq = e3 - (e2 + 1);
            if (ls && b[e1 - 1] == '\r') --ls;
            if (lq && b[e3 - 1] == '\r') --lq;
            if (ls == lq) return p;
        }
        return 0;
    }
    void count_file(const std::string &path, SequenceFormat fmt, bool want_qual) {
        const char *hp = getenv("KMERUST_HOST_PARSE");
        const bool host_only = is_stdin_path(path) || (hp && hp[0] && hp[0] != '0');
        timing().text_path = !host_only;
        if (host_only || !count_file_text(path, resolve_format(fmt, &path))) {
            timing().text_path = false;
            const size_t batch = ctxs.size() > 1 ? (size_t)64 << 20 : (size_t)512 << 20;
            read_sequences(path, fmt, want_qual, batch, [&](const Batch &b) {
                kh_ctx *c = ctxs[next_ctx++ % ctxs.size()];
                check_on(c, kh_push(c, b.bases.data(), want_qual && !b.qual.empty() ? b.qual.data() : nullptr, b.bases.size()),
                         "kh_push");
            });
        }
        Lap lap(timing().finish_s);
        for (kh_ctx *c : ctxs) check_on(c, kh_finish(c, nullptr), "kh_finish");
        if (group) {
            const int rc = kh_group_merge(group, nullptr);
            if (rc != KH_OK) {
                std::string msg = std::string("kh_group_merge: ") + kh_strerror(rc);
                for (kh_ctx *c : ctxs)
                    if (kh_last_error(c)[0]) msg += std::string(" (") + kh_last_error(c) + ")";
                throw Error(msg);
            }
        }
    }
    PackedCounts result(uint64_t min_count) {
        Lap lap(timing().result_s);
        PackedCounts pc;
        pc.k = k;
        std::vector<uint64_t> ns(ctxs.size(), 0);
        uint64_t total = 0;
        for (size_t i = 0; i < ctxs.size(); ++i) {
            check_on(ctxs[i], kh_result_size(ctxs[i], min_count, &ns[i]), "kh_result_size");
            total += ns[i];
        }
        pc.keys.resize(total);
        pc.counts.resize(total);
        uint64_t off = 0;
        for (size_t i = 0; i < ctxs.size(); ++i) {  // the shards' key sets are disjoint: concatenation is the map
            uint64_t got = 0;
            check_on(ctxs[i], kh_result_copy(ctxs[i], pc.keys.data() + off, pc.counts.data() + off, ns[i], min_count, &got),
                     "kh_result_copy");
            off += got;
        }
        pc.keys.resize(off);
        pc.counts.resize(off);
        return pc;
    }
    std::vector<std::pair<uint64_t, uint64_t>> histogram(uint64_t min_count) {
        Lap lap(timing().result_s);
        std::map<uint64_t, uint64_t> sum;  // a histogram of disjoint shards is element-wise additive
        for (kh_ctx *c : ctxs) {
            uint64_t cap = 1u << 16;
            for (;;) {
                std::vector<uint64_t> cc(cap), f(cap);
                uint64_t n = 0;
                const int rc = kh_histogram(c, min_count, cc.data(), f.data(), cap, &n);
                if (rc == KH_ERR_RANGE) {
                    cap *= 16;
                    continue;
                }
                check_on(c, rc, "kh_histogram");
                if (ctxs.size() == 1) {
                    std::vector<std::pair<uint64_t, uint64_t>> h(n);
                    for (uint64_t i = 0; i < n; ++i) h[i] = {cc[i], f[i]};
                    return h;
                }
                for (uint64_t i = 0; i < n; ++i) sum[cc[i]] += f[i];
                break;
            }
        }
        return std::vector<std::pair<uint64_t, uint64_t>>(sum.begin(), sum.end());
    }
};

KmerCounter &KmerCounter::k(size_t kk) {
    if (kk < 1 || kk > 32) throw KmerLengthError(kk);  // KmerLength::new, src/kmer.rs:100-110
    k_ = kk;
    k_set_ = true;
    return *this;
}

// size of a plain (uncompressed, regular) input file; 0 for stdin, gzip and anything stat() does not know
static uint64_t plain_file_bytes(const std::string &path) {
    if (is_stdin_path(path) || lower_ext(path) == "gz") return 0;
    struct stat sb;
    return (stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) ? (uint64_t)sb.st_size : 0;
}

static bool wants_quality(const KmerCounter &, SequenceFormat resolved, int min_quality, const std::string &path) {
    // run.rs:543: both Some; the CLI path for stdin ignores -Q (src/main.rs:145-152, run.rs:195-197)
    return min_quality >= 0 && resolved == SequenceFormat::Fastq && !is_stdin_path(path);
}

PackedCounts KmerCounter::count_packed(const std::string &path, bool apply_min_count) const {
    const SequenceFormat f = resolve_format(input_format_, is_stdin_path(path) ? nullptr : &path);
    const bool q = wants_quality(*this, f, min_quality_, path);
    Session s(*this, q, plain_file_bytes(path));
    s.count_file(path, f, q);
    return s.result(apply_min_count ? min_count_ : 1);
}

std::unordered_map<std::string, uint64_t> KmerCounter::count(const std::string &path) const {
    const PackedCounts pc = count_packed(path, true);
    std::unordered_map<std::string, uint64_t> m;
    m.reserve(pc.keys.size());
    for (size_t i = 0; i < pc.keys.size(); ++i) m.emplace(unpack_to_string(pc.keys[i], pc.k), pc.counts[i]);
    return m;
}

std::vector<std::pair<uint64_t, uint64_t>> KmerCounter::histogram(const std::string &path) const {
    const SequenceFormat f = resolve_format(input_format_, is_stdin_path(path) ? nullptr : &path);
    const bool q = wants_quality(*this, f, min_quality_, path);
    Session s(*this, q, plain_file_bytes(path));
    s.count_file(path, f, q);
    return s.histogram(min_count_);  // computed on the device, after the min_count filter (run.rs:447-450)
}

void KmerCounter::count_to_writer(const std::string &path, FILE *out) const {
    if (format_ == OutputFormat::Histogram) {
        write_histogram(out, histogram(path));
        return;
    }
    write_counts(out, count_packed(path, true), format_, 1);
}

void KmerCounter::run(const std::string &path) const { count_to_writer(path, stdout); }

// =============================================================================================
// output
// =============================================================================================
std::string unpack_to_string(uint64_t bits, uint32_t k) {
    std::string s(k, 'A');
    kh_unpack(bits, k, reinterpret_cast<uint8_t *>(&s[0]));
    return s;
}

void write_histogram(FILE *out, const std::vector<std::pair<uint64_t, uint64_t>> &hist) {
    Lap lap(timing().write_s);
    for (const auto &cf : hist) fprintf(out, "%llu\t%llu\n", (unsigned long long)cf.first, (unsigned long long)cf.second);
    fflush(out);
}

namespace {
// BufWriter of run.rs:446: lines are formatted into an own buffer and handed to fwrite in 1 MiB pieces
// (no setvbuf: the C standard leaves it undefined once the stream has been used).
struct OutBuf {
    FILE *out;
    std::string buf;
    explicit OutBuf(FILE *o) : out(o) { buf.reserve((1u << 20) + 256); }
    void put(const char *s, size_t n) {
        buf.append(s, n);
        if (buf.size() >= (1u << 20)) flush();
    }
    void put(const char *s) { put(s, strlen(s)); }
    void put_u64(unsigned long long v) {
        char t[24];
        int n = 0;
        do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        char r[24];
        for (int i = 0; i < n; ++i) r[i] = t[n - 1 - i];
        put(r, (size_t)n);
    }
    void flush() {
        if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), out);
        buf.clear();
    }
    ~OutBuf() {
        flush();
        fflush(out);
    }
};
}  // namespace

void write_counts(FILE *out, const PackedCounts &pc, OutputFormat fmt, uint64_t min_count) {
    OutBuf w(out);
    char kmer[33];
    auto each = [&](const std::function<void(const char *, unsigned long long, bool)> &fn) {
        bool first = true;
        for (size_t i = 0; i < pc.keys.size(); ++i) {
            if (pc.counts[i] < min_count) continue;
            kh_unpack(pc.keys[i], pc.k, reinterpret_cast<uint8_t *>(kmer));
            kmer[pc.k] = 0;
            fn(kmer, (unsigned long long)pc.counts[i], first);
            first = false;
        }
        return first;  // true if nothing was written
    };
    switch (fmt) {
    case OutputFormat::Fasta:  // ">{count}\n{kmer}\n"  (run.rs:453-456)
        each([&](const char *km, unsigned long long c, bool) {
            w.put(">", 1); w.put_u64(c); w.put("\n", 1); w.put(km, pc.k); w.put("\n", 1);
        });
        break;
    case OutputFormat::Tsv:  // "{kmer}\t{count}\n"  (run.rs:458-461)
        each([&](const char *km, unsigned long long c, bool) {
            w.put(km, pc.k); w.put("\t", 1); w.put_u64(c); w.put("\n", 1);
        });
        break;
    case OutputFormat::Json: {  // serde_json::to_writer_pretty of Vec<{kmer,count}> + newline (run.rs:463-470)
        bool any = false;
        each([&](const char *km, unsigned long long c, bool first) {
            w.put(first ? "[\n" : ",\n");
            w.put("  {\n    \"kmer\": \""); w.put(km, pc.k); w.put("\",\n    \"count\": "); w.put_u64(c); w.put("\n  }");
            any = true;
        });
        w.put(any ? "\n]\n" : "[]\n");
        break;
    }
    case OutputFormat::Histogram: {  // host fallback: count-of-counts after the filter (run.rs:471-481)
        std::map<uint64_t, uint64_t> h;
        for (size_t i = 0; i < pc.keys.size(); ++i)
            if (pc.counts[i] >= min_count) h[pc.counts[i]]++;
        for (const auto &cf : h) { w.put_u64(cf.first); w.put("\t", 1); w.put_u64(cf.second); w.put("\n", 1); }
        break;
    }
    }
}

// =============================================================================================
// KMIX index
// =============================================================================================
uint32_t crc32_ieee(const uint8_t *data, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

static bool ends_with_gz(const std::string &p) { return lower_ext(p) == "gz"; }

static void put_le64(std::vector<uint8_t> &v, uint64_t x) {
    for (int i = 0; i < 8; ++i) v.push_back((uint8_t)(x >> (8 * i)));
}
static uint64_t get_le64(const uint8_t *p) {
    uint64_t x = 0;
    for (int i = 0; i < 8; ++i) x |= (uint64_t)p[i] << (8 * i);
    return x;
}

void save_index(const PackedCounts &pc, const std::string &path) {
    std::vector<uint8_t> v;
    v.reserve(18 + pc.keys.size() * 16);
    v.insert(v.end(), {'K', 'M', 'I', 'X'});  // MAGIC
    v.push_back(1);                           // VERSION
    v.push_back((uint8_t)pc.k);
    put_le64(v, pc.keys.size());
    for (size_t i = 0; i < pc.keys.size(); ++i) {
        put_le64(v, pc.keys[i]);
        put_le64(v, pc.counts[i]);
    }
    const uint32_t crc = crc32_ieee(v.data(), v.size());  // of everything before it
    for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(crc >> (8 * i)));
    bool ok;
    if (ends_with_gz(path)) {
        gzFile g = gzopen(path.c_str(), "wb");
        ok = g != nullptr;
        size_t off = 0;
        while (ok && off < v.size()) {
            const unsigned chunk = (unsigned)std::min<size_t>(v.size() - off, 1u << 30);
            ok = gzwrite(g, v.data() + off, chunk) == (int)chunk;
            off += chunk;
        }
        if (g) ok = (gzclose(g) == Z_OK) && ok;
    } else {
        FILE *f = fopen(path.c_str(), "wb");
        ok = f && fwrite(v.data(), 1, v.size(), f) == v.size();
        if (f) ok = (fclose(f) == 0) && ok;
    }
    if (!ok) throw Error("failed to write index file '" + path + "': " + std::strerror(errno));
}

PackedCounts load_index(const std::string &path) {
    auto bad = [&](const std::string &details) { return Error("invalid index file '" + path + "': " + details); };
    gzFile g = gzopen(path.c_str(), "rb");  // reads plain files transparently
    if (!g) throw Error("failed to read index file '" + path + "': " + std::strerror(errno));
    std::vector<uint8_t> data;
    std::vector<uint8_t> buf(1u << 20);
    for (;;) {
        const int n = gzread(g, buf.data(), (unsigned)buf.size());
        if (n < 0) {
            gzclose(g);
            throw Error("failed to read index file '" + path + "': decompression error");
        }
        if (n == 0) break;
        data.insert(data.end(), buf.begin(), buf.begin() + n);
    }
    gzclose(g);
    if (data.size() < 18) throw bad("file too small");
    if (memcmp(data.data(), "KMIX", 4) != 0) throw bad("invalid magic bytes (not a kmerust index file)");
    const size_t body = data.size() - 4;
    uint32_t stored = 0;
    for (int i = 0; i < 4; ++i) stored |= (uint32_t)data[body + i] << (8 * i);
    const uint32_t computed = crc32_ieee(data.data(), body);
    if (computed != stored) {
        char m[96];
        snprintf(m, sizeof m, "checksum mismatch (expected 0x%x, got 0x%x)", stored, computed);  // {:#x}
        throw bad(m);
    }
    if (data[4] != 1) throw bad("unsupported version " + std::to_string(data[4]));
    const uint32_t k = data[5];
    if (k < 1 || k > 32)
        throw bad("invalid k-mer length: k-mer length " + std::to_string(k) + " is out of range: must be between 1 and 32");
    const uint64_t count = get_le64(&data[6]);
    const size_t have = body - 14;
    if (count > (SIZE_MAX >> 5) || have != count * 16)
        throw bad("data size mismatch (expected " + std::to_string(count * 16) + " bytes, got " + std::to_string(have) + " bytes)");
    PackedCounts pc;
    pc.k = k;
    pc.keys.resize(count);
    pc.counts.resize(count);
    for (uint64_t i = 0; i < count; ++i) {
        pc.keys[i] = get_le64(&data[14 + 16 * i]);
        pc.counts[i] = get_le64(&data[14 + 16 * i + 8]);
    }
    return pc;
}

}  // namespace kmerust
