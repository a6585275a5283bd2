// nh_run.hip -- whole-run entry nh_run(): files in, kraken2-compatible files out.
//
// Replaces, for the caller at /root/reference/src/main.rs:270, what the kraken2 subprocess does
// with the argv of src/main.rs:210-267: read 1-2 FASTQ/FASTA inputs (plain/gzip/bzip2), classify
// every fragment, write the kept fragments UNCOMPRESSED to kraken_out.fq or kraken_out_1.fq +
// kraken_out_2.fq (src/main.rs:252-256,308-309,333), optionally the per-read kraken output
// (--output) and return the counts nohuman scrapes from stderr (src/lib.rs:61-97).
// kraken2 units restated: classify.cc ProcessFiles / output formatting.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "nh_codec.h"
#include "nh_fastx.h"
#include "nh_gunzip.h"
#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {

// Output file written with writev(): a batch's output is a list of spans, either slices of the raw
// input text (kept records that need no reformatting -- the common case, no copy) or pieces of a
// scratch buffer (reformatted records, kraken output lines).
// What one batch puts into one output file: slices of the batch's raw text (the normal case: nothing is copied) and of
// a scratch string (reformatted records), in output order
struct Spans {
    std::vector<struct iovec> iov;
    std::vector<char> is_scratch;  // per span: iov_base is an offset into scratch
    std::string scratch;
    void clear() {
        iov.clear();
        is_scratch.clear();
        scratch.clear();
    }
};

struct OutFile {
    int fd = -1;
    std::string path;
    Spans sp;  // the batch being formatted (the writer moves it into the flusher's job and gets a cleared one back)
    std::string &scratch = sp.scratch;
    std::unique_ptr<StreamEncoder> enc;  // set: the spans go through a streaming encoder
    int open(const char *p, int codec = NH_CODEC_NONE, unsigned codec_threads = 1, int device = -1) {
        path = p;
        fd = ::open(p, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        if (fd < 0) return set_error(NH_EIO, "cannot create %s", p);
        if (codec != NH_CODEC_NONE) {
            enc.reset(make_encoder(codec, fd, codec_threads ? codec_threads : 1, p, device));
            if (!enc) return NH_EINVAL;  // message set by make_encoder
        }
        return NH_OK;
    }
    void add_raw(const char *p, size_t n) {
        if (!n) return;
        std::vector<struct iovec> &iov = sp.iov;
        if (!iov.empty() && !sp.is_scratch.back() && (const char *)iov.back().iov_base + iov.back().iov_len == p) {
            iov.back().iov_len += n;
            return;
        }
        iov.push_back({(void *)p, n});
        sp.is_scratch.push_back(0);
    }
    // the caller appended [from, scratch.size()) to scratch
    void add_scratch(size_t from) {
        const size_t n = sp.scratch.size() - from;
        if (!n) return;
        std::vector<struct iovec> &iov = sp.iov;
        if (!iov.empty() && sp.is_scratch.back() && (size_t)iov.back().iov_base + iov.back().iov_len == from) {
            iov.back().iov_len += n;
            return;
        }
        iov.push_back({(void *)from, n});
        sp.is_scratch.push_back(1);
    }
    int flush() { return flush(sp); }
    int flush(Spans &x) {  // write all spans of x, then forget them
        std::vector<struct iovec> &iov = x.iov;
        for (size_t i = 0; i < iov.size(); i++)
            if (x.is_scratch[i]) iov[i].iov_base = (void *)(x.scratch.data() + (size_t)iov[i].iov_base);
        size_t i = 0;
        int rc = NH_OK;
        if (enc) {
            for (; i < iov.size() && rc == NH_OK; i++) rc = enc->write(iov[i].iov_base, iov[i].iov_len);
            i = iov.size();
            const int src = enc->settle();  // the batch's buffers (host and device) go back to the pipeline after this
            if (rc == NH_OK) rc = src;
        }
        while (i < iov.size()) {
            const int cnt = (int)std::min<size_t>(iov.size() - i, 512);
            ssize_t w = ::writev(fd, &iov[i], cnt);
            if (w < 0) {
                if (errno == EINTR) continue;
                rc = set_error(NH_EIO, "write error on %s", path.c_str());
                break;
            }
            size_t left = (size_t)w;  // consume whole spans, shorten a partly written one
            while (i < iov.size() && left >= iov[i].iov_len) left -= iov[i++].iov_len;
            if (left) {
                iov[i].iov_base = (char *)iov[i].iov_base + left;
                iov[i].iov_len -= left;
            }
        }
        x.clear();
        return rc;
    }
    int close() {
        int rc = NH_OK;
        if (enc) {
            rc = enc->finish();
            enc.reset();
        }
        if (fd >= 0 && ::close(fd) != 0 && rc == NH_OK) rc = set_error(NH_EIO, "write error on %s", path.c_str());
        fd = -1;
        return rc;
    }
    ~OutFile() {
        enc.reset();
        if (fd >= 0) ::close(fd);
    }
};

// kraken2 TrimPairInfo: drop a trailing /1 or /2 from ids longer than two characters
static void trim_pair_info(std::string &id) {
    size_t sz = id.size();
    if (sz <= 2) return;
    if (id[sz - 2] == '/' && (id[sz - 1] == '1' || id[sz - 1] == '2')) id.erase(sz - 2);
}

// kraken2 AddHitlistString
static void append_hitlist(std::string &dst, const Engine *e, const uint32_t *taxa, uint64_t n) {
    if (n == 0) {
        dst += "0:0";
        return;
    }
    char tmp[64];
    uint64_t i = 0;
    bool first = true;
    while (i < n) {
        uint32_t t = taxa[i];
        uint64_t j = i;
        while (j < n && taxa[j] == t) j++;
        if (t == TAXON_MATE_BORDER) {
            for (uint64_t r = i; r < j; r++) {
                if (!first) dst += ' ';
                dst += "|:|";
                first = false;
            }
        } else {
            if (!first) dst += ' ';
            if (t == TAXON_AMBIGUOUS)
                snprintf(tmp, sizeof tmp, "A:%llu", (unsigned long long)(j - i));
            else
                snprintf(tmp, sizeof tmp, "%llu:%llu", (unsigned long long)e->external[t],
                         (unsigned long long)(j - i));
            dst += tmp;
            first = false;
        }
        i = j;
    }
}

// kraken2 reports.cc ReportKrakenStyle / KrakenReportDFS (reached through
// nohuman's -r/--kraken-report, /root/reference/src/main.rs:97-99,226-228).  Lowest-confidence part
// of the restatement: the line format is "%6.2f\tclade\ttaxon\trank\ttaxid\t<2*depth spaces>name".
struct TaxoView {
    uint64_t n = 0;
    const uint8_t *nodes = nullptr;
    const char *names = nullptr, *ranks = nullptr;
    uint64_t field(uint64_t i, int k) const {
        uint64_t v;
        memcpy(&v, nodes + 56 * i + 8 * k, 8);
        return v;
    }
};

static void report_dfs(FILE *f, const TaxoView &t, const std::vector<uint64_t> &clade,
                       const std::vector<uint64_t> &own, uint64_t total, uint64_t id, char rank_code,
                       int rank_depth, int depth) {
    if (clade[id] == 0) return;  // clades absent from the sample are not printed
    const std::string rank = t.ranks + t.field(id, 4);
    if (rank == "superkingdom" || rank == "domain") { rank_code = 'D'; rank_depth = 0; }
    else if (rank == "kingdom") { rank_code = 'K'; rank_depth = 0; }
    else if (rank == "phylum") { rank_code = 'P'; rank_depth = 0; }
    else if (rank == "class") { rank_code = 'C'; rank_depth = 0; }
    else if (rank == "order") { rank_code = 'O'; rank_depth = 0; }
    else if (rank == "family") { rank_code = 'F'; rank_depth = 0; }
    else if (rank == "genus") { rank_code = 'G'; rank_depth = 0; }
    else if (rank == "species") { rank_code = 'S'; rank_depth = 0; }
    else rank_depth++;
    std::string rank_str(1, rank_code);
    if (rank_depth != 0) rank_str += std::to_string(rank_depth);
    fprintf(f, "%6.2f\t%llu\t%llu\t%s\t%llu\t", 100.0 * (double)clade[id] / (double)total,
            (unsigned long long)clade[id], (unsigned long long)own[id], rank_str.c_str(),
            (unsigned long long)t.field(id, 5));
    for (int i = 0; i < depth; i++) fputs("  ", f);
    fprintf(f, "%s\n", t.names + t.field(id, 3));
    const uint64_t first = t.field(id, 1), cnt = t.field(id, 2);
    std::vector<uint64_t> kids(cnt);
    for (uint64_t i = 0; i < cnt; i++) kids[i] = first + i;
    std::stable_sort(kids.begin(), kids.end(),
                     [&](uint64_t x, uint64_t y) { return clade[x] > clade[y]; });
    for (uint64_t c : kids) report_dfs(f, t, clade, own, total, c, rank_code, rank_depth, depth + 1);
}

static int write_report(const Engine *e, const char *path, const std::vector<uint64_t> &own,
                        uint64_t total, uint64_t unclassified) {
    TaxoView t;
    const uint8_t *img = e->taxo_image.data();
    memcpy(&t.n, img + 8, 8);
    uint64_t name_len;
    memcpy(&name_len, img + 16, 8);
    t.nodes = img + 32;
    t.names = (const char *)(img + 32 + 56 * t.n);
    t.ranks = t.names + name_len;
    std::vector<uint64_t> clade(own);
    for (uint64_t i = t.n; i-- > 1;)  // children have larger ids than their parents
        if (clade[i]) clade[t.field(i, 0)] += clade[i];
    FILE *f = fopen(path, "w");
    if (!f) return set_error(NH_EIO, "cannot create %s", path);
    if (unclassified != 0)
        fprintf(f, "%6.2f\t%llu\t%llu\tU\t0\tunclassified\n",
                100.0 * (double)unclassified / (double)total, (unsigned long long)unclassified,
                (unsigned long long)unclassified);
    if (t.n > 1 && total) report_dfs(f, t, clade, own, total, 1, 'R', -1, 0);
    if (fclose(f) != 0) return set_error(NH_EIO, "write error on %s", path);
    return NH_OK;
}

// ---- the pipeline -----------------------------------------------------------------------------------
// reader thread(s): inflate + parse one input file each into HalfBatches (bounded queue)
// main thread:      pair the halves, copy their raw text (page-locked batch buffers) to the device as it
//                   is and classify the sequences in place: (start, length) per sequence; H2D + classify +
//                   D2H asynchronously on one of two stream slots per device (batch b -> device
//                   b mod G, database replicated per device: SURVEY.md section 8e)
// writer thread:    format and write the outputs of finished batches in input order
static size_t estimate_batch_frags(const char *path, size_t *mean_record_bytes) {
    BlockReader r;
    std::string err;
    *mean_record_bytes = 0;
    if (r.open(path, err) != 0) return 1u << 18;
    HalfBatch hb;
    r.next_batch(hb, 256, 64u << 20);
    if (hb.recs.empty()) return 1u << 18;
    uint64_t sum = 0;
    for (const RecRef &x : hb.recs) sum += x.slen;
    const uint64_t mean = sum / hb.recs.size() + 1;
    const RecRef &lastr = hb.recs.back();
    *mean_record_bytes = (size_t)((lastr.q + lastr.qlen + 2 > lastr.s + lastr.slen + 2 ? lastr.q + lastr.qlen + 2
                                                                                   : lastr.s + lastr.slen + 2) /
                                  hb.recs.size() + 1);
    uint64_t frags = (96ull << 20) / mean;
    if (frags < 256) frags = 256;
    if (frags > (1u << 18)) frags = 1u << 18;
    return (size_t)frags;
}

template <class T>
class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T &&v) {
        std::unique_lock<std::mutex> lk(mu_);
        not_full_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return;
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    bool pop(T &out) {  // false when closed and drained
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
        not_empty_.notify_all();
        not_full_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

// What the two input files contribute to one launch: records [off, off + n) of each half.  Normally a half is used whole
// (both readers cut at the same record counts); where a reader had to cut by TEXT (a batch's record text is addressed with 32
// bits), or the two files came from different readers after a handover, the halves differ in length and the longer one is used
// in parts: the halves are shared, and go back to their pool (device-born ones: to their reader) when the last part is through.
// A shared half is READ-ONLY once its first part has been queued (the writer used to fetch a device-born half's
// text into hb.text in place while the main thread read its size for the next part and the flusher of the part before still
// held spans into the old buffer): the halves' text lengths are taken when the half arrives (len1 / len2), and a part that
// must have the bytes on the host after all (a kept record that is rewritten: CRLF, "+id") fetches them into a buffer of its own.
struct Batch {
    std::shared_ptr<HalfBatch> h1, h2;
    size_t off1 = 0, off2 = 0;
    size_t len1 = 0, len2 = 0;  // bytes of text of the two halves (the whole half's, also for a part)
    std::unique_ptr<char[]> fetch1, fetch2;  // this part's own host copy of a half's text (see above), else null
    size_t n = 0;
    int slot = -1;  // stream slot that carries its results
    int dev_index = 0;  // index of the slot's device among the run's engines
};

struct Slot {  // pinned host + device buffers of one in-flight batch
    Engine *e = nullptr;
    hipStream_t stream = nullptr;
    int work_slot = 0;
    // The sequences are classified IN PLACE: the raw text of the batch (both mate files' record text, as
    // read) is copied to the device as it is, and the kernel gets (start, length) of every sequence.
    uint64_t *h_off = nullptr;   // n_seq starts into d_text
    uint32_t *h_len = nullptr;   // n_seq lengths
    nh_result *h_res = nullptr;
    uint32_t *h_taxa = nullptr;
    uint64_t *h_taxa_off = nullptr;
    int *h_flag = nullptr;  // the engine's sticky error bits as of the end of this slot's batch (copied on its stream)
    void *d_text = nullptr, *d_off = nullptr, *d_len = nullptr, *d_res = nullptr, *d_taxa = nullptr, *d_taxa_off = nullptr;
    size_t cap_text = 0, cap_frag = 0, cap_taxa = 0;
    uint64_t n_taxa = 0;
    bool busy = false;
};

static int slot_reserve(Slot &s, size_t ntext, size_t nfrag, size_t ntaxa) {
    if (dev_set(s.e->device) != hipSuccess) return set_error(NH_EDEVICE, "hipSetDevice failed");
    auto grow = [](size_t need) { return need + need / 4 + 4096; };
    if (!s.h_flag) {
        if (host_malloc((void **)&s.h_flag, 64, hipHostMallocDefault) != hipSuccess) return set_error(NH_EOOM, "cannot allocate batch buffers");
        *s.h_flag = 0;
    }
    if (ntext + 64 > s.cap_text) {
        if (s.d_text) (void)hipFree(s.d_text);
        s.cap_text = grow(ntext + 64);
        if (dev_malloc(&s.d_text, s.cap_text) != hipSuccess)
            return set_error(NH_EOOM, "cannot allocate batch buffers (%zu bytes)", s.cap_text);
    }
    if (nfrag > s.cap_frag) {
        for (void *p : {(void *)s.h_off, (void *)s.h_len, (void *)s.h_res, (void *)s.h_taxa_off})
            if (p) (void)hipHostFree(p);
        for (void *p : {s.d_off, s.d_len, s.d_res, s.d_taxa_off})
            if (p) (void)hipFree(p);
        s.cap_frag = grow(nfrag);
        const size_t ns = 2 * s.cap_frag + 2;
        if (host_malloc((void **)&s.h_off, ns * 8, hipHostMallocDefault) != hipSuccess ||
            host_malloc((void **)&s.h_len, ns * 4, hipHostMallocDefault) != hipSuccess ||
            host_malloc((void **)&s.h_res, s.cap_frag * sizeof(nh_result), hipHostMallocDefault) != hipSuccess ||
            host_malloc((void **)&s.h_taxa_off, (s.cap_frag + 1) * 8, hipHostMallocDefault) != hipSuccess ||
            dev_malloc(&s.d_off, ns * 8) != hipSuccess || dev_malloc(&s.d_len, ns * 4) != hipSuccess ||
            dev_malloc(&s.d_res, s.cap_frag * sizeof(nh_result)) != hipSuccess ||
            dev_malloc(&s.d_taxa_off, (s.cap_frag + 1) * 8) != hipSuccess)
            return set_error(NH_EOOM, "cannot allocate batch buffers (%zu fragments)", s.cap_frag);
    }
    if (ntaxa > s.cap_taxa) {
        if (s.h_taxa) (void)hipHostFree(s.h_taxa);
        if (s.d_taxa) (void)hipFree(s.d_taxa);
        s.cap_taxa = grow(ntaxa);
        if (host_malloc((void **)&s.h_taxa, s.cap_taxa * 4, hipHostMallocDefault) != hipSuccess ||
            dev_malloc(&s.d_taxa, s.cap_taxa * 4) != hipSuccess)
            return set_error(NH_EOOM, "cannot allocate k-mer taxa buffers");
    }
    return NH_OK;
}

static void slot_free(Slot &s) {
    if (!s.e) return;
    (void)dev_set(s.e->device);
    for (void *p : {(void *)s.h_off, (void *)s.h_len, (void *)s.h_res, (void *)s.h_taxa, (void *)s.h_taxa_off, (void *)s.h_flag})
        if (p) (void)hipHostFree(p);
    for (void *p : {s.d_text, s.d_off, s.d_len, s.d_res, s.d_taxa, s.d_taxa_off})
        if (p) (void)hipFree(p);
    if (s.stream) (void)hipStreamDestroy(s.stream);
}

// page-locked text buffers of the batches (RawBuf allocator): the H2D copy of a batch's raw text runs
// asynchronously straight from where the reader inflated / read it
// A host that cannot page-lock that much (memlock / cgroup limits) gets pageable memory instead -- the
// asynchronous copy accepts it and merely stages it itself; a 64-byte header says which kind a block is.
static std::atomic<int> g_pageable_batches{0};
// Page-locking a batch buffer costs ~40 ms and releasing it as much again: a run of a few seconds spent 0.4-0.7 s of its wall
// time on its dozen buffers.  Released buffers are kept for the process's next run (at most PINNED_KEEP of them; nh_close
// of an engine empties the store, NOHUMAN_PINNED_CACHE=0 turns it off).
static constexpr size_t PINNED_KEEP = 16;
static std::mutex g_pinned_mu;
static std::vector<void *> g_pinned_store;  // block starts (header: [0] kind, [1] usable bytes)
static bool pinned_cache_on() {
    static const bool on = !(getenv("NOHUMAN_PINNED_CACHE") && getenv("NOHUMAN_PINNED_CACHE")[0] == '0');
    return on;
}
static void *pinned_alloc(size_t n) {
    void *p = nullptr;
    static const bool no_pin = getenv("NOHUMAN_NO_PINNED") != nullptr;  // test knob: exercise the fallback
    if (!no_pin && pinned_cache_on() && n >= ((size_t)8u << 20)) {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        for (size_t i = 0; i < g_pinned_store.size(); i++) {
            const uint64_t cap = ((uint64_t *)g_pinned_store[i])[1];
            if (cap >= n && cap <= n + n / 2) {
                p = g_pinned_store[i];
                g_pinned_store.erase(g_pinned_store.begin() + (long)i);
                return (char *)p + 64;
            }
        }
    }
    if (!no_pin && host_malloc(&p, n + 64, hipHostMallocPortable) == hipSuccess) {
        ((uint64_t *)p)[0] = 1;
        ((uint64_t *)p)[1] = n;
        return (char *)p + 64;
    }
    (void)hipGetLastError();
    if (posix_memalign(&p, 64, n + 64) != 0) return nullptr;
    ((uint64_t *)p)[0] = 2;
    ((uint64_t *)p)[1] = n;
    g_pageable_batches++;
    return (char *)p + 64;
}
static void pinned_free(void *q) {
    void *p = (char *)q - 64;
    if (*(uint64_t *)p == 1) {
        if (pinned_cache_on() && ((uint64_t *)p)[1] >= ((size_t)8u << 20)) {
            std::lock_guard<std::mutex> lk(g_pinned_mu);
            if (g_pinned_store.size() < PINNED_KEEP) {
                g_pinned_store.push_back(p);
                return;
            }
        }
        (void)hipHostFree(p);
    } else {
        free(p);
    }
}
void run_cache_trim() {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    for (void *p : g_pinned_store) (void)hipHostFree(p);
    g_pinned_store.clear();
}

// folds the counters of one run into the engine's running totals (nh_stats_get)
__global__ void k_add_counters(unsigned long long *dst, const unsigned long long *src) {
    atomicAdd(&dst[threadIdx.x], src[threadIdx.x]);
}

struct RunState {
    const nh_run_args *a;
    std::vector<Engine *> engines;
    bool paired, want_k;
    // results of the run
    uint64_t total = 0, classified = 0, total_bases = 0;
    std::vector<uint64_t> dev_counts;  // per device {fragments, classified, bases, 0} as the writer saw them (checker)
    std::vector<uint64_t *> d_run_counters;  // per device: the counters the classify kernels of THIS run add to (HBM)
    std::vector<uint64_t> call_counts;
    // first error of any thread
    std::mutex err_mu;
    int err_code = NH_OK;
    std::string err_msg;
    void fail(int code, const std::string &msg) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (err_code == NH_OK) {
            err_code = code;
            err_msg = msg;
        }
    }
    bool failed() {
        std::lock_guard<std::mutex> lk(err_mu);
        return err_code != NH_OK;
    }
};

// Recycles batch buffers (~100 MB each) between the writer and the readers, so that steady state
// neither allocates nor page-faults.
class BatchPool {
public:
    ~BatchPool() {
        if (filler_.joinable()) filler_.join();
    }
    std::unique_ptr<HalfBatch> get() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!free_.empty()) {
                std::unique_ptr<HalfBatch> hb = std::move(free_.back());
                free_.pop_back();
                return hb;
            }
        }
        return make();
    }
    void put(std::unique_ptr<HalfBatch> hb) {
        if (!hb) return;
        if (hb->release) {  // a batch born on the GPU: its text buffer goes back to the device reader
            hb->release();
            hb->release = nullptr;
        }
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(std::move(hb));
    }
    // Page-locking ~100 MB takes tens of milliseconds: a helper makes the first buffers while the reader
    // already fills the ones it has, instead of the reader stopping for each of them.
    void start_prefill(int n) {
        filler_ = std::thread([this, n] {
            for (int i = 0; i < n && !stop_.load(); i++) put(make());
        });
    }
    void stop() { stop_.store(true); }
    size_t first_reserve = 0;

private:
    std::unique_ptr<HalfBatch> make() {
        std::unique_ptr<HalfBatch> hb(new HalfBatch());
        hb->text.set_allocator(pinned_alloc, pinned_free);
        if (first_reserve) hb->text.reserve(first_reserve);  // one allocation instead of a dozen growing ones
        return hb;
    }
    std::mutex mu_;
    std::vector<std::unique_ptr<HalfBatch>> free_;
    std::thread filler_;
    std::atomic<bool> stop_{false};
};

struct StageClock {  // NOHUMAN_TRACE=1: where the wall time of a run goes, per thread
    std::atomic<uint64_t> ns[12];
    StageClock() {
        for (auto &x : ns) x = 0;
    }
    static uint64_t now() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                   std::chrono::steady_clock::now().time_since_epoch())
            .count();
    }
};
enum { ST_READ1 = 0, ST_READ2, ST_RPUSH1, ST_RPUSH2, ST_MPOP, ST_MSLOT, ST_MGATHER, ST_MLAUNCH, ST_WPOP, ST_WSYNC, ST_WFORMAT, ST_WWRITE };

// the host reader's loop: batches of an opened BlockReader into the queue, behind `skip` records that are read and dropped
// (the reader on the GPU handed them out before it handed the file over)
static void host_read(BlockReader &r, BoundedQueue<std::unique_ptr<HalfBatch>> *out, RunState *rs, BatchPool *pool, StageClock *clk, int which,
                      size_t batch_frags, size_t batch_text, uint64_t skip) {
    while (skip) {
        std::unique_ptr<HalfBatch> hb = pool->get();
        r.next_batch(*hb, (size_t)std::min<uint64_t>(skip, batch_frags), batch_text);
        if (!hb->error.empty()) {
            rs->fail(NH_EIO, hb->error);
            return;
        }
        const size_t n = hb->recs.size();
        const bool eof = hb->eof;
        pool->put(std::move(hb));
        skip -= std::min<uint64_t>(skip, n);
        if (eof || rs->failed()) {  // (the file ends inside what the device reader read: it cannot be the same file)
            if (skip) rs->fail(NH_EIO, "the input changed while it was read");
            return;
        }
    }
    for (;;) {
        std::unique_ptr<HalfBatch> hb = pool->get();
        hb->recs.reserve(batch_frags);
        uint64_t t0 = StageClock::now();
        r.next_batch(*hb, batch_frags, batch_text);
        uint64_t t1 = StageClock::now();
        clk->ns[ST_READ1 + which] += t1 - t0;
        if (!hb->error.empty()) {
            rs->fail(NH_EIO, hb->error);
            break;
        }
        const bool eof = hb->eof;
        out->push(std::move(hb));
        clk->ns[ST_RPUSH1 + which] += StageClock::now() - t1;
        if (eof || rs->failed()) break;
    }
}

// Which reader a gzip input starts on when NOHUMAN_GZ_READER does not say.  Measured (tools/reader_choice.py,
// profiles/r05_reader_choice.txt; run time with the reader on the GPU / with the host reader on 16 cores):
//   outputs gzip-encoded on the GPU (the text never leaves HBM): 0.45-0.94 at every size from 50 MB to 1 GB a file, paired
//     and single-end, and 0.79-0.92 on long reads                                                      -> the GPU
//   short reads, only classified reads kept / --output wanted (the text is fetched, little is written): 0.33-0.78 -> the GPU
//   outputs written by the host (plain, bzip2, xz, zstd: every kept byte crosses PCIe and goes through
//     writev): 0.78-0.98 below 128 MiB a file, 1.09-1.21 from there to 1 GiB, 0.88 at 4 GiB             -> the host in between
//   long reads (batches few and large: nothing flows before the first 512 MiB piece is through) with outputs written by the
//     host or nothing kept: 1.14-1.26                                                                  -> the host
static bool device_reader_pays(const char *path, size_t mean_record_bytes, bool host_text_wanted, bool bulk_by_host) {
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if (const char *e = getenv("NOHUMAN_GZDEV_MIN_BYTES")) return (uint64_t)st.st_size >= (uint64_t)atoll(e);  // tuning / test knob
    if (!host_text_wanted) return true;
    if (mean_record_bytes > 4096) return false;
    // bulk_by_host: most of the input is expected to leave through the host (the default mode keeps the non-human reads; a
    // --keep-human-reads run keeps few).  Between 128 MiB and 2 GiB a file -- one to four pieces, little to pipeline -- the
    // host reader is the faster one then; above, the GPU reader's pieces overlap the writes (50 M pairs: 25.5 against 22.3)
    const uint64_t sz = (uint64_t)st.st_size;
    return !(bulk_by_host && sz >= ((uint64_t)128u << 20) && sz < ((uint64_t)2u << 30));
}

static void reader_main(const char *path, BoundedQueue<std::unique_ptr<HalfBatch>> *out, RunState *rs,
                        BatchPool *pool, StageClock *clk, int which, size_t batch_frags, size_t batch_text,
                        unsigned gz_threads, std::vector<int> gz_devices, bool device_reader, unsigned hybrid_threads) {
    const int gz_device = gz_devices.empty() ? -1 : gz_devices[0];
    // gzip FASTQ: the whole reader on the GPU (inflate, record index; the text stays in HBM) where that pays (the caller
    // decides: device_reader) or NOHUMAN_GZ_READER says "device"; "host": the host decoders; "device-text": inflate on the
    // GPU, records parsed on the host (BlockReader)
    uint64_t skip = 0;  // records the reader on the GPU handed out before it handed the file over
    {
        const char *how = getenv("NOHUMAN_GZ_READER");
        const bool named = how && !strcmp(how, "device");
        if (gz_device >= 0 && device_reader) {
            DevFastqReader dr;
            std::string derr;
            // (the run's devices, this file's first: piece i of the stream is inflated, indexed and classified on device i mod G)
            const int orc = dr.open(path, gz_devices.data(), (int)gz_devices.size(), derr, hybrid_threads);
            if (orc < 0) {
                if (named) {  // asked for by name: no silent change of reader
                    rs->fail(NH_EIO, derr);
                    out->close();
                    return;
                }
                fprintf(stderr, "nohuman: WARN %s: the gzip reader on GPU %d could not be set up (%s); reading on the host\n", path, gz_device,
                        derr.c_str());
            }
            bool fell_back = orc != 0;
            while (!fell_back) {
                std::unique_ptr<HalfBatch> hb = pool->get();
                uint64_t t0 = StageClock::now();
                // (paired inputs: both files' readers cut at the same record counts; single-end batches are cut by text as well)
                const int rc = dr.next_batch(*hb, batch_frags, rs->paired ? 0 : batch_text);
                uint64_t t1 = StageClock::now();
                clk->ns[ST_READ1 + which] += t1 - t0;
                if (rc == 1) {
                    // no four-line FASTQ: the host parser reads the file (FASTA, wrapped lines, its error messages) -- or the
                    // reader on the GPU could not go on: the host reader does, behind the records already handed out
                    pool->put(std::move(hb));
                    if (!dr.handover_reason().empty()) {
                        if (named) {
                            rs->fail(NH_EIO, dr.handover_reason());
                            break;
                        }
                        skip = dr.records_handed();
                        fprintf(stderr, "nohuman: WARN %s: the gzip reader on GPU %d stopped (%s); the host reader goes on from record %llu\n", path,
                                gz_device, dr.handover_reason().c_str(), (unsigned long long)skip);
                    }
                    fell_back = true;
                    break;
                }
                if (!hb->error.empty()) {
                    rs->fail(NH_EIO, hb->error);
                    break;
                }
                const bool eof = hb->eof;
                out->push(std::move(hb));
                clk->ns[ST_RPUSH1 + which] += StageClock::now() - t1;
                if (eof || rs->failed()) break;
            }
            if (!fell_back) {
                out->close();
                dr.close();  // (waits for the batches still in the pipeline: they point into its buffers)
                return;
            }
            if (skip == 0) dr.close();
            else {
                // the batches it handed out are still in the pipeline and point into its buffers: the host reader starts now, the
                // device reader's buffers go when they are through (close() waits for them)
                BlockReader r;
                std::string err;
                if (r.open(path, err, gz_threads, -1) != 0) {
                    rs->fail(NH_EIO, err);
                    out->close();
                    dr.close();
                    return;
                }
                host_read(r, out, rs, pool, clk, which, batch_frags, batch_text, skip);
                out->close();
                dr.close();
                return;
            }
        }
    }
    BlockReader r;
    std::string err;
    // (the GPU inflates for the host parser only when asked for by name: "device-text", or "device" and the text is no FASTQ)
    const char *how = getenv("NOHUMAN_GZ_READER");
    if (r.open(path, err, gz_threads, how && (!strcmp(how, "device-text") || !strcmp(how, "device")) ? gz_device : -1) != 0) {
        rs->fail(NH_EIO, err);
        out->close();
        return;
    }
    host_read(r, out, rs, pool, clk, which, batch_frags, batch_text, 0);
    out->close();
}

static inline void put_record(std::string &dst, const char *text, int format, const RecRef &r, const char *suffix) {
    dst.append(text + r.h, r.hlen);
    if (suffix) dst += suffix;
    dst += '\n';
    dst.append(text + r.s, r.slen);
    if (format == FMT_FASTQ) {
        dst += "\n+\n";
        dst.append(text + r.q, r.qlen);
    }
    dst += '\n';
}

// decide + format one finished batch into the span lists of the output files (runs on the writer
// thread, batches arrive in input order)
static void format_batch(RunState *rs, const Batch &b, const Slot &s, OutFile &o1, OutFile &o2, OutFile &ok) {
    const nh_run_args *a = rs->a;
    const Engine *e = s.e;
    const bool keep_class = a->keep_human != 0;
    const char *t1 = b.fetch1 ? b.fetch1.get() : b.h1->text.data();
    const char *t2 = !rs->paired ? nullptr : b.fetch2 ? b.fetch2.get() : b.h2->text.data();
    char tmp[128];
    uint64_t bases = 0, classified = 0;
    const RecRef *const R1 = b.h1->recs.data() + b.off1;
    const RecRef *const R2 = rs->paired ? b.h2->recs.data() + b.off2 : nullptr;
    for (size_t i = 0; i < b.n; i++) {
        const RecRef &r1 = R1[i];
        const uint32_t call = s.h_res[i].call;
        const bool is_class = call != 0;
        classified += is_class;
        rs->call_counts[call] += is_class;
        bases += r1.slen + (rs->paired ? R2[i].slen : 0);
        if (is_class == keep_class) {
            if (!is_class && r1.raw_end) {
                o1.add_raw(t1 + r1.h, r1.raw_end - r1.h);
            } else {
                const char *suffix = nullptr;
                if (is_class) {
                    snprintf(tmp, sizeof tmp, " kraken:taxid|%llu", (unsigned long long)e->external[call]);
                    suffix = tmp;
                }
                const size_t from = o1.scratch.size();
                put_record(o1.scratch, t1, b.h1->format, r1, suffix);
                o1.add_scratch(from);
            }
            if (rs->paired) {
                const RecRef &r2 = R2[i];
                if (!is_class && r2.raw_end) {
                    o2.add_raw(t2 + r2.h, r2.raw_end - r2.h);
                } else {
                    const size_t from = o2.scratch.size();
                    put_record(o2.scratch, t2, b.h2->format, r2, is_class ? tmp : nullptr);
                    o2.add_scratch(from);
                }
            }
        }
        if (rs->want_k) {
            const uint64_t ext = is_class ? e->external[call] : 0;
            std::string &bufk = ok.scratch;
            std::string id(t1 + r1.h + 1, r1.idlen);
            if (rs->paired) trim_pair_info(id);
            bufk += is_class ? "C\t" : "U\t";
            bufk += id;
            snprintf(tmp, sizeof tmp, "\t%llu\t", (unsigned long long)ext);
            bufk += tmp;
            if (rs->paired)
                snprintf(tmp, sizeof tmp, "%u|%u\t", r1.slen, R2[i].slen);
            else
                snprintf(tmp, sizeof tmp, "%u\t", r1.slen);
            bufk += tmp;
            append_hitlist(bufk, e, s.h_taxa + s.h_taxa_off[i], s.h_taxa_off[i + 1] - s.h_taxa_off[i]);
            bufk += '\n';
        }
    }
    if (rs->want_k) ok.add_scratch(0);
    rs->total += b.n;
    rs->classified += classified;
    rs->total_bases += bases;
    uint64_t *dc = &rs->dev_counts[4 * (size_t)b.dev_index];
    dc[0] += b.n;
    dc[1] += classified;
    dc[2] += bases;
}

int run_engines(const std::vector<Engine *> &engines, const nh_run_args *a, nh_stats *stats) {
    if (!a || !a->in1 || !a->out1) return set_error(NH_EINVAL, "nh_run: in1 and out1 are required");
    if (engines.empty()) return set_error(NH_EINVAL, "nh_run: no engine");
    RunState rs;
    rs.a = a;
    rs.engines = engines;
    rs.paired = a->in2 != nullptr;
    if (rs.paired && !a->out2) return set_error(NH_EINVAL, "nh_run: paired input needs out2");
    if (!(a->confidence >= 0.0 && a->confidence <= 1.0))
        return set_error(NH_EINVAL, "Confidence score must be in the closed interval [0, 1]");
    rs.want_k = a->kraken_output && a->kraken_output[0] && strcmp(a->kraken_output, "/dev/null") != 0;
    rs.call_counts.assign(engines[0]->external.size(), 0);
    rs.dev_counts.assign(4 * engines.size(), 0);
    for (const char *p : {a->in1, a->in2}) {  // fail on unreadable inputs before creating outputs
        if (!p) continue;
        FILE *f = fopen(p, "rb");
        if (!f) return set_error(NH_EIO, "cannot open %s", p);
        fclose(f);
    }
    // outputs are created with O_TRUNC before the first input byte is read: an output that IS an input
    // (same device and inode) would be emptied -- refuse (the CLI host stages its outputs and renames)
    for (const char *o : {a->out1, a->out2, rs.want_k ? a->kraken_output : nullptr, a->report}) {
        struct stat so;
        if (!o || !o[0] || stat(o, &so) != 0 || !S_ISREG(so.st_mode)) continue;
        for (const char *p : {a->in1, a->in2}) {
            struct stat si;
            if (p && stat(p, &si) == 0 && si.st_dev == so.st_dev && si.st_ino == so.st_ino)
                return set_error(NH_EINVAL, "nh_run: output %s is the input %s", o, p);
        }
    }
    OutFile o1, o2, ok;
    int rc;
    if (a->out_codec < NH_CODEC_NONE || a->out_codec > NH_CODEC_ZSTD)
        return set_error(NH_EINVAL, "nh_run: unknown out_codec %d", a->out_codec);
    // gzip output is encoded on the GPU (nh_deflate.hip); with two devices each mate file has its own
    // (the two are opened side by side: a GPU encoder takes 70 ms to set up -- page-locked buffers, a trial of its prices)
    {
        int rc2 = NH_OK;
        std::string err2;
        std::thread t2;
        if (rs.paired)
            t2 = std::thread([&] {
                rc2 = o2.open(a->out2, a->out_codec, a->codec_threads, engines[engines.size() > 1 ? 1 : 0]->device);
                if (rc2) err2 = g_last_error;
            });
        rc = o1.open(a->out1, a->out_codec, a->codec_threads, engines[0]->device);
        if (t2.joinable()) t2.join();
        if (rc) return rc;
        if (rc2) return set_error(rc2, "%s", err2.c_str());
    }
    if (rs.want_k && (rc = ok.open(a->kraken_output))) return rc;

    // fragments per batch: ~96 MB of sequence, at most 262144; both readers cut at the same record
    // count so that paired batches stay aligned (a byte budget only cuts single-end batches)
    size_t mean_rec = 0;
    size_t BATCH_FRAGS = estimate_batch_frags(a->in1, &mean_rec);
    if (const char *env = getenv("NOHUMAN_BATCH_FRAGS")) {  // tuning / test knob
        const long v = atol(env);
        if (v > 0) BATCH_FRAGS = (size_t)v;
    }
    // byte budget of one reader's batch: single-end batches are cut by it; paired batches are cut by record
    // count (both readers at the same count), and their budget only keeps the text of the two halves together
    // below the 4 GB a batch's 32-bit sequence positions can address
    size_t BATCH_TEXT = rs.paired ? (size_t)0x7F000000u : (size_t)(512u << 20);
    if (const char *env = getenv("NOHUMAN_BATCH_TEXT")) {  // test knob: batches cut by text at small scale (paired: the halves then differ in length)
        const long v = atol(env);
        if (v > 0) BATCH_TEXT = (size_t)v;
    }
    const int G = (int)engines.size();
    const int mates = rs.paired ? 2 : 1;
    // Batches of the reader on the GPU keep their text in HBM.  The host needs the bytes for plain outputs and the host's
    // codecs, for the ids of --output lines and for the suffix of classified-out headers; gzip outputs encoded on the GPU
    // take the kept records from HBM (a record that needs reformatting -- CRLF, "+id" -- makes the writer fetch its batch).
    const bool host_text_wanted = !(a->out_codec == NH_CODEC_GZIP && o1.enc && o1.enc->takes_device_spans() &&
                                    (!rs.paired || (o2.enc && o2.enc->takes_device_spans()))) ||
                                  rs.want_k || a->keep_human != 0 || G > 1;  // (G > 1: a batch's slot and its file's encoder may sit on different GPUs)
    const uint32_t flags = rs.paired ? NH_FLAG_PAIRED : 0;
    auto t0 = std::chrono::steady_clock::now();

    rs.d_run_counters.assign((size_t)G, nullptr);
    auto free_run_counters = [&] {
        for (int g = 0; g < G; g++)
            if (rs.d_run_counters[g]) {
                (void)dev_set(engines[g]->device);
                (void)hipFree(rs.d_run_counters[g]);
                rs.d_run_counters[g] = nullptr;
            }
    };
    for (int g = 0; g < G; g++) {
        const size_t nb = (CNT_N + 12) * sizeof(uint64_t);  // (+12: the words of the instrumented kernel variant)
        if (dev_set(engines[g]->device) != hipSuccess || dev_malloc((void **)&rs.d_run_counters[g], nb) != hipSuccess ||
            hipMemset(rs.d_run_counters[g], 0, nb) != hipSuccess) {
            free_run_counters();
            return set_error(NH_EDEVICE, "cannot allocate the run's counters on device %d", engines[g]->device);
        }
    }
    // stream slots a device: two keep the classifier busy; a third lets the copy of a batch's text back to the host (batches
    // born on the GPU, outputs written by the host) run while the writer is still busy with the batch before; a fourth is
    // held by the batch the flusher is writing
    int NS = 4;
    if (const char *env = getenv("NOHUMAN_SLOTS")) NS = std::max(1, std::min(8, atoi(env)));
    std::vector<Slot> slots((size_t)NS * G);
    for (int i = 0; i < NS * G; i++) {
        slots[i].e = engines[i / NS];
        slots[i].work_slot = i % NS;
        if (dev_set(slots[i].e->device) != hipSuccess ||
            hipStreamCreateWithFlags(&slots[i].stream, hipStreamNonBlocking) != hipSuccess) {
            for (auto &s : slots) slot_free(s);
            free_run_counters();
            return set_error(NH_EDEVICE, "cannot create streams");
        }
    }

    BoundedQueue<std::unique_ptr<HalfBatch>> q1(3), q2(3);
    BatchPool pool1, pool2;
    int prefill = 4 + 2 * G;
    if (mean_rec) {  // size the page-locked text buffers once: a batch of records plus one read chunk
        size_t want = BATCH_FRAGS * mean_rec + BATCH_FRAGS * mean_rec / 8 + (8u << 20);
        if (want > BATCH_TEXT + (8u << 20)) want = BATCH_TEXT + (8u << 20);
        // small inputs: no more (and no larger) buffers than the text they can hold, compressed 12:1 at most
        struct stat st1;
        if (stat(a->in1, &st1) == 0 && S_ISREG(st1.st_mode)) {
            const uint64_t est = (uint64_t)st1.st_size * 12 + (1u << 20);
            if (est < want) want = (size_t)est;
            const uint64_t nb = est / want + 1;
            if (nb < (uint64_t)prefill) prefill = (int)nb;
        }
        if (want < (1ull << 31)) pool1.first_reserve = pool2.first_reserve = want;
        if (getenv("NOHUMAN_TRACE"))
            fprintf(stderr, "[nohuman trace] batch: %zu fragments, text buffers of %zu bytes reserved at once, %d prefilled per file\n",
                    BATCH_FRAGS, pool1.first_reserve, prefill);
    }
    int prefill1 = prefill, prefill2 = prefill;
    // which reader takes a gzip input: NOHUMAN_GZ_READER by name ("device", "host", "device-text"), else by the input
    // (device_reader_pays); paired files go the same way (the smaller file decides)
    bool dev_reader1 = false, dev_reader2 = false, split_readers = false;
    {
        const char *how = getenv("NOHUMAN_GZ_READER");
        const bool off = how && (!strcmp(how, "host") || !strcmp(how, "device-text"));
        const bool named = how && !strcmp(how, "device");
        const bool by_host = a->keep_human == 0 && !(a->out_codec == NH_CODEC_GZIP && o1.enc && o1.enc->takes_device_spans() &&
                                                     (!rs.paired || (o2.enc && o2.enc->takes_device_spans())));
        dev_reader1 = !off && dev_gunzip_wants(a->in1) && (named || device_reader_pays(a->in1, mean_rec, host_text_wanted, by_host));
        dev_reader2 = rs.paired && !off && dev_gunzip_wants(a->in2) && (named || device_reader_pays(a->in2, mean_rec, host_text_wanted, by_host));
        if (rs.paired && !named && dev_reader1 != dev_reader2 && dev_gunzip_wants(a->in1) && dev_gunzip_wants(a->in2)) dev_reader1 = dev_reader2 = false;
        // "split": the two mate files on the two KINDS of reader at once -- file 1 inflated and indexed on the GPU, file 2 by all
        // the host's inflate workers (round 6: the GPU's codec kernels and the host's cores idle in turn otherwise)
        split_readers = rs.paired && how && !strcmp(how, "split") && dev_gunzip_wants(a->in1) && dev_gunzip_wants(a->in2);
        if (split_readers) dev_reader1 = true, dev_reader2 = false;
        if (getenv("NOHUMAN_TRACE"))
            fprintf(stderr, "[nohuman trace] gzip reader: %s%s%s\n", dev_reader1 ? "GPU" : "host", rs.paired ? (dev_reader2 ? " / GPU" : " / host") : "",
                    how ? " (NOHUMAN_GZ_READER)" : "");
        // batches of the reader on the GPU carry no host text unless an output needs it: no page-locked buffers made ahead for them
        if (dev_reader1 && !host_text_wanted) pool1.first_reserve = 0, prefill1 = 0;
        if (dev_reader2 && !host_text_wanted) pool2.first_reserve = 0, prefill2 = 0;
    }
    if (prefill1) pool1.start_prefill(prefill1);
    if (rs.paired && prefill2) pool2.start_prefill(prefill2);
    StageClock clk;
    // gzip inputs are inflated by `threads` workers in all, shared between the files
    unsigned gz_threads = (a->threads ? a->threads : 1) / (unsigned)mates;
    if (gz_threads < 1) gz_threads = 1;
    if (gz_threads > 16) gz_threads = 16;  // beyond that the record parser of the file is the limit
    unsigned gz_threads2 = gz_threads;
    if (split_readers) gz_threads2 = std::min(16u, std::max(1u, a->threads ? a->threads : 1u));  // (file 1's reader is on the GPU: every worker to file 2)
    // gzip inputs are read on the GPU (nh_gunzip.hip): file 1 on the first device, file 2 on the second where there is one
    // ... piece by piece over the run's devices (the compressed ranges are the shards): file 1 starts on the
    // first device, file 2 on the second, so the two files' pieces of the same moment sit on different GPUs
    std::vector<int> devs1, devs2;
    for (int g = 0; g < G; g++) {
        devs1.push_back(engines[(size_t)g]->device);
        devs2.push_back(engines[(size_t)((g + 1) % G)]->device);
    }
    if (getenv("NOHUMAN_GZ_SHARD") && getenv("NOHUMAN_GZ_SHARD")[0] == '0') devs1.resize(1), devs2.resize(1);  // (A / B: a file's reader on one device)
    if (const char *e = getenv("NOHUMAN_GZ_LANES")) {  // tuning knob: several lanes of the reader on ONE device (pieces decoded ahead there)
        const int n = atoi(e);
        while (G == 1 && (int)devs1.size() < n && n <= 4) devs1.push_back(devs1[0]), devs2.push_back(devs2[0]);
    }
    // The hybrid reader (round 6, nh_gunzip.hip): the odd cells of each input's (then alternating) piece grid inflated by host workers
    // beside the GPU's.  OFF unless asked for (NOHUMAN_GZ_HYBRID=n: n workers per file; 1: the run's threads less four, shared between
    // the files): built for the runs whose kept text is re-encoded on the GPU -- there the chip's codec kernels, inflate and deflate in
    // turn, are what the run waits for -- and measured there: 30.8 -> 24.7 Mreads/s on 40 M pairs (profiles/r06_hybrid.txt): a host
    // piece's in-order part (stitching, CRC, copy, upload: 0.2 s) holds the file's stream up five times a file, and pieces decoded
    // ahead lose the in-order reader's staged input.  Correct and covered; what would make it pay is written down there.
    unsigned hybrid_threads = 0;
    if (const char *e = getenv("NOHUMAN_GZ_HYBRID")) {
        const int v = atoi(e);
        const unsigned T = a->threads ? a->threads : 1;
        const unsigned want = v <= 0 ? 0u : v == 1 ? std::max(1u, (T > 4 ? T - 4 : 1u) / (unsigned)mates) : (unsigned)v;
        hybrid_threads = G == 1 ? std::min(16u, want) : 0u;  // (one device: the reader's lanes over several are untested with it)
        if (getenv("NOHUMAN_TRACE") && (dev_reader1 || dev_reader2))
            fprintf(stderr, "[nohuman trace] gzip reader: hybrid %s (%u host workers per file)\n", hybrid_threads ? "on" : "off", hybrid_threads);
    }
    std::thread t1(reader_main, a->in1, &q1, &rs, &pool1, &clk, 0, BATCH_FRAGS, BATCH_TEXT, gz_threads, devs1, dev_reader1, hybrid_threads);
    std::thread t2;
    if (rs.paired)
        t2 = std::thread(reader_main, a->in2, &q2, &rs, &pool2, &clk, 1, BATCH_FRAGS, BATCH_TEXT, gz_threads2, devs2, dev_reader2, hybrid_threads);

    // writer: consumes batches in order; each arrives after its stream was synchronised.  Two stages: the WRITER
    // waits for the batch's stream, decides and formats (span lists, nothing is copied); the FLUSHER writes the spans
    // of the batch before while the writer is on the next one -- a helper takes the second mate file, so that both files
    // are written at the same time -- and only then gives the batch's buffers and its stream slot back.  (One thread did
    // both in round 3: 35.5 GB of kept text leave through writev() at 12 GB/s, 2.9 s, and the 1.3 s of waiting and
    // formatting came on top, profiles/r04_e2e.txt.)
    struct FlushJob {
        Batch b;
        Spans s1, s2, sk;
        bool valid = false;  // (false: the run had failed when the batch arrived -- only its buffers go back)
    };
    BoundedQueue<Batch> wq((size_t)(NS * G));
    BoundedQueue<std::unique_ptr<FlushJob>> fq(1);
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    std::mutex w2_mu;
    std::condition_variable w2_cv;
    int w2_state = 0;  // 0 idle, 1 flush requested, 2 done, -1 quit
    Spans *w2_spans = nullptr;
    int w2_rc = NH_OK;
    std::string w2_err;
    std::thread tw2;
    if (rs.paired)
        tw2 = std::thread([&] {
            for (;;) {
                std::unique_lock<std::mutex> lk(w2_mu);
                w2_cv.wait(lk, [&] { return w2_state == 1 || w2_state == -1; });
                if (w2_state == -1) return;
                Spans *sp2 = w2_spans;
                lk.unlock();
                int frc = o2.flush(*sp2);
                lk.lock();
                w2_rc = frc;
                if (frc) w2_err = g_last_error;
                w2_state = 2;
                w2_cv.notify_all();
            }
        });
    std::thread tf([&] {
        std::unique_ptr<FlushJob> j;
        while (fq.pop(j)) {
            Slot &s = slots[j->b.slot];
            if (j->valid && !rs.failed()) {
                uint64_t c3 = StageClock::now();
                (void)dev_set(s.e->device);
                const Batch &b = j->b;
                const size_t base2w = (b.len1 + 8 + 255) & ~(size_t)255;
                // the batch's raw text is still in the slot's device buffer: an encoder on that GPU takes
                // the kept records from there instead of a second trip over PCIe
                // (a part that fetched its own host copy has its spans there: they are staged like any host memory)
                if (o1.enc) o1.enc->map_device(b.h1->text.data(), b.len1, s.d_text, s.e->device, b.h1->host_text_valid);
                if (rs.paired && o2.enc)
                    o2.enc->map_device(b.h2->text.data(), b.len2, (const char *)s.d_text + base2w, s.e->device, b.h2->host_text_valid);
                if (rs.paired) {
                    std::lock_guard<std::mutex> lk(w2_mu);
                    w2_spans = &j->s2;
                    w2_state = 1;
                    w2_cv.notify_all();
                }
                int wrc = o1.flush(j->s1);
                if (!wrc && rs.want_k) wrc = ok.flush(j->sk);
                if (rs.paired) {
                    std::unique_lock<std::mutex> lk(w2_mu);
                    w2_cv.wait(lk, [&] { return w2_state == 2; });
                    w2_state = 0;
                    if (!wrc && w2_rc) wrc = set_error(w2_rc, "%s", w2_err.c_str());
                }
                clk.ns[ST_WWRITE] += StageClock::now() - c3;
                if (wrc) rs.fail(wrc, g_last_error);
            }
            j->b.h1.reset();  // (the last part of a half gives it back to its pool: the deleter of take())
            j->b.h2.reset();
            {
                std::lock_guard<std::mutex> lk(slot_mu);
                s.busy = false;
            }
            slot_cv.notify_all();
        }
    });
    std::thread tw([&] {
        Batch b;
        for (;;) {
            uint64_t c0 = StageClock::now();
            if (!wq.pop(b)) break;
            uint64_t c1 = StageClock::now();
            clk.ns[ST_WPOP] += c1 - c0;
            Slot &s = slots[b.slot];
            std::unique_ptr<FlushJob> j(new FlushJob());
            if (!rs.failed()) {
                (void)dev_set(s.e->device);
                hipError_t he = hipStreamSynchronize(s.stream);
                uint64_t c2 = StageClock::now();
                clk.ns[ST_WSYNC] += c2 - c1;
                int wrc = NH_OK;
                if (he != hipSuccess) wrc = set_error(NH_EDEVICE, "classify: %s", hipGetErrorString(he));
                // (the error bits came with the results on the slot's stream: a copy of their own here would queue behind
                // whatever else the device is running -- the gzip reader's kernels -- once a batch)
                if (!wrc && *s.h_flag) wrc = check_error_flag(s.e);
                if (!wrc) {
                    const size_t base2w = (b.len1 + 8 + 255) & ~(size_t)255;
                    // a batch whose text is only in HBM, and a kept record that must be rewritten (CRLF, "+id" line): fetch it
                    // (into a buffer of this part: the half itself may be in use by other parts -- Batch)
                    auto need_fetch = [&](const HalfBatch &hb) {
                        if (hb.host_text_valid) return false;
                        const size_t off = &hb == b.h1.get() ? b.off1 : b.off2;
                        for (size_t i = 0; i < b.n; i++)
                            if (!hb.recs[off + i].raw_end && (s.h_res[i].call != 0) == (a->keep_human != 0)) return true;
                        return false;
                    };
                    for (int m = 0; m < (rs.paired ? 2 : 1) && !wrc; m++) {
                        const HalfBatch &hb = m ? *b.h2 : *b.h1;
                        if (need_fetch(hb)) {
                            const size_t L = m ? b.len2 : b.len1;
                            std::unique_ptr<char[]> &dst = m ? b.fetch2 : b.fetch1;
                            dst.reset(new (std::nothrow) char[L + 64]);
                            if (!dst) wrc = set_error(NH_EOOM, "out of memory");
                            if (!wrc && hipMemcpy(dst.get(), (const char *)s.d_text + (m ? base2w : 0), L, hipMemcpyDeviceToHost) != hipSuccess)
                                wrc = set_error(NH_EDEVICE, "fetching a batch's text from the device failed");
                        }
                    }
                }
                if (!wrc) {
                    format_batch(&rs, b, s, o1, o2, ok);
                    clk.ns[ST_WFORMAT] += StageClock::now() - c2;
                    j->s1 = std::move(o1.sp);
                    j->s2 = std::move(o2.sp);
                    j->sk = std::move(ok.sp);
                    o1.sp.clear();
                    o2.sp.clear();
                    ok.sp.clear();
                    j->valid = true;
                }
                if (wrc) rs.fail(wrc, g_last_error);
            }
            j->b = std::move(b);
            fq.push(std::move(j));
        }
        fq.close();
    });

    // main: pair halves, stage, launch
    uint64_t batch_no = 0;
    std::vector<uint64_t> home_turn((size_t)G, 0);
    // the half of each file in hand and how many of its records have gone out (a half is normally used whole: Batch)
    std::shared_ptr<HalfBatch> c1, c2;
    size_t p1 = 0, p2 = 0, cl1 = 0, cl2 = 0;  // (cl: the half's bytes of text, read once, when it arrives)
    auto take = [](BoundedQueue<std::unique_ptr<HalfBatch>> &q, BatchPool *pool, std::shared_ptr<HalfBatch> &c, size_t &pos, size_t &len) {
        std::unique_ptr<HalfBatch> u;
        if (!q.pop(u)) return false;
        c = std::shared_ptr<HalfBatch>(u.release(), [pool](HalfBatch *h) { pool->put(std::unique_ptr<HalfBatch>(h)); });
        pos = 0;
        len = c->text.size();
        return true;
    };
    for (;;) {
        if (rs.failed()) break;
        Batch b;
        uint64_t m0 = StageClock::now();
        if (!c1 && !take(q1, &pool1, c1, p1, cl1)) break;
        if (rs.paired && !c2 && !take(q2, &pool2, c2, p2, cl2)) break;
        // kraken2 reads the files in lockstep and stops at the shorter one.  Both readers cut batches at the same record
        // counts, so the halves normally pair up whole; where they do not (see Batch) the shorter one decides and the rest of
        // the longer one pairs with the other file's next half -- no read is dropped, no run stopped
        b.h1 = c1;
        b.off1 = p1;
        b.len1 = cl1;
        b.n = c1->recs.size() - p1;
        if (rs.paired) {
            b.h2 = c2;
            b.off2 = p2;
            b.len2 = cl2;
            b.n = std::min(b.n, c2->recs.size() - p2);
            p2 += b.n;
        }
        p1 += b.n;
        // a half that is used up is let go of HERE, not when the next one arrives: a reader on the GPU may need its buffer to
        // produce that next one (a piece smaller than a batch hands out nothing until the piece behind it is decoded)
        const bool done1 = p1 == c1->recs.size(), done2 = rs.paired && p2 == c2->recs.size();
        const bool last = (done1 && c1->eof) || (done2 && c2->eof);
        if (done1) c1.reset();
        if (done2) c2.reset();
        if (b.n > 0) {
            // the slot: in turn over all devices' slots -- but a batch born on a GPU (the gzip reader there) is classified
            // on THAT device, in that device's slots in turn, unless they are all busy and another device has a free one
            // (then its text crosses xGMI once); the writer takes the batches in the order they are pushed, whatever the slot
            int si = (int)(batch_no % (uint64_t)(NS * G));
            uint64_t m1 = StageClock::now();
            clk.ns[ST_MPOP] += m1 - m0;
            {
                int home = -1;
                if (G > 1 && b.h1->dev_text)
                    for (int g = 0; g < G; g++)
                        if (engines[(size_t)g]->device == b.h1->dev_device) {
                            home = g;
                            break;
                        }
                std::unique_lock<std::mutex> lk(slot_mu);
                if (home >= 0) {
                    si = home * NS + (int)(home_turn[(size_t)home]++ % (uint64_t)NS);
                    if (slots[(size_t)si].busy)
                        for (int k = 0; k < NS * G; k++)
                            if (!slots[(size_t)k].busy) {
                                si = k;
                                break;
                            }
                }
                slot_cv.wait(lk, [&] { return !slots[(size_t)si].busy; });
                slots[(size_t)si].busy = true;
            }
            Slot &s = slots[si];
            uint64_t m2 = StageClock::now();
            clk.ns[ST_MSLOT] += m2 - m1;
            b.slot = si;
            b.dev_index = si / NS;
            // (start, length) of every sequence inside the raw text: text of file 1 at byte 0 of the device
            // buffer, text of file 2 behind it
            const size_t len1 = b.len1, len2 = rs.paired ? b.len2 : 0;
            const size_t base2 = (len1 + 8 + 255) & ~(size_t)255;
            const size_t ntext = rs.paired ? base2 + len2 : len1;
            if (ntext >= (1ull << 32)) {
                rs.fail(NH_EINVAL, "a batch of more than 4 GB of record text (its offsets are 32-bit): paired records of several kilobases each are not supported");
                wq.push(std::move(b));
                break;
            }
            uint64_t ntaxa = 0;
            const uint64_t k = s.e->info.k;
            if (rs.want_k) {
                for (size_t i = 0; i < b.n; i++) {
                    const uint64_t l1 = b.h1->recs[b.off1 + i].slen;
                    ntaxa += l1 >= k ? l1 - k + 1 : 0;
                    if (rs.paired) {
                        const uint64_t l2 = b.h2->recs[b.off2 + i].slen;
                        ntaxa += (l2 >= k ? l2 - k + 1 : 0) + 1;
                    }
                }
            }
            rc = slot_reserve(s, ntext, b.n, ntaxa + 1);
            if (rc) {
                rs.fail(rc, g_last_error);
                wq.push(std::move(b));
                break;
            }
            dev_check_ptr(s.d_text, s.e->device, "nh_run, a slot's text buffer");
            uint64_t nbases = 0, toff = 0;
            for (size_t i = 0; i < b.n; i++) {
                const RecRef &r1 = b.h1->recs[b.off1 + i];
                s.h_off[i * mates] = r1.s;
                s.h_len[i * mates] = r1.slen;
                nbases += r1.slen;
                if (rs.want_k) {
                    s.h_taxa_off[i] = toff;
                    toff += r1.slen >= k ? r1.slen - k + 1 : 0;
                }
                if (rs.paired) {
                    const RecRef &r2 = b.h2->recs[b.off2 + i];
                    s.h_off[i * mates + 1] = base2 + r2.s;
                    s.h_len[i * mates + 1] = r2.slen;
                    nbases += r2.slen;
                    if (rs.want_k) toff += (r2.slen >= k ? r2.slen - k + 1 : 0) + 1;
                }
            }
            if (rs.want_k) s.h_taxa_off[b.n] = toff;
            s.n_taxa = toff;
            // H2D, classify, D2H: all asynchronous on the slot's stream
            uint64_t m3 = StageClock::now();
            clk.ns[ST_MGATHER] += m3 - m2;
            hipError_t he = dev_set(s.e->device);
            {   // test knob of the checker itself: the launch goes out under the run's FIRST device whatever slot carries it
                static const bool brk = getenv("NOHUMAN_DEBUG_DEVICE_BREAK") != nullptr;
                if (brk) he = dev_set(engines[0]->device);
            }
            // the batch's text: from the host as it was read, or -- a batch of the reader on the GPU -- from HBM to HBM (from
            // another device's memory: over xGMI); the bytes then go to the host only if an output written there needs them
            auto stage_text = [&](HalfBatch &hb, size_t at, size_t len) -> hipError_t {
                if (!len) return hipSuccess;
                if (!hb.dev_text) return hipMemcpyAsync((char *)s.d_text + at, hb.text.data(), len, hipMemcpyHostToDevice, s.stream);
                hipError_t e2 = dev_copy_between((char *)s.d_text + at, s.e->device, hb.dev_text, hb.dev_device, len, s.stream);
                if (e2 == hipSuccess && host_text_wanted && !hb.host_text_valid) {  // (a half used in parts is fetched by its first)
                    hb.text.clear();  // (a batch born on the GPU comes with a token buffer: the real one only where it is needed)
                    if (!hb.text.reserve(len + 64)) return hipErrorOutOfMemory;
                    hb.text.set_size(len);
                    e2 = hipMemcpyAsync(hb.text.data(), (char *)s.d_text + at, len, hipMemcpyDeviceToHost, s.stream);
                    hb.host_text_valid = true;  // (once the stream has been synchronised: the writer does that first)
                }
                return e2;
            };
            if (he == hipSuccess) he = stage_text(*b.h1, 0, len1);
            if (he == hipSuccess && rs.paired) he = stage_text(*b.h2, base2, len2);
            if (he == hipSuccess)
                he = hipMemcpyAsync(s.d_off, s.h_off, b.n * mates * 8, hipMemcpyHostToDevice, s.stream);
            if (he == hipSuccess)
                he = hipMemcpyAsync(s.d_len, s.h_len, b.n * mates * 4, hipMemcpyHostToDevice, s.stream);
            if (he == hipSuccess && rs.want_k)
                he = hipMemcpyAsync(s.d_taxa_off, s.h_taxa_off, (b.n + 1) * 8, hipMemcpyHostToDevice, s.stream);
            if (he != hipSuccess) {
                rs.fail(NH_EDEVICE, std::string("H2D: ") + hipGetErrorString(he));
            } else {
                rc = classify_device(s.e, s.d_text, s.d_off, b.n,
                                     flags | (nbases / b.n > 2000 ? NH_FLAG_LONG : 0u), a->confidence, s.d_res,
                                     rs.want_k ? s.d_taxa : nullptr, rs.want_k ? s.d_taxa_off : nullptr,
                                     rs.d_run_counters[(size_t)(si / NS)], s.stream, s.d_len, ntext);
                if (rc) rs.fail(rc, g_last_error);
            }
            if (!rs.failed()) {
                he = hipMemcpyAsync(s.h_res, s.d_res, b.n * sizeof(nh_result), hipMemcpyDeviceToHost, s.stream);
                if (he == hipSuccess)
                    he = hipMemcpyAsync(s.h_flag, s.e->d_error + LAUNCH_SLOTS, sizeof(int), hipMemcpyDeviceToHost, s.stream);
                if (he == hipSuccess && rs.want_k && toff)
                    he = hipMemcpyAsync(s.h_taxa, s.d_taxa, toff * 4, hipMemcpyDeviceToHost, s.stream);
                if (he != hipSuccess) rs.fail(NH_EDEVICE, std::string("D2H: ") + hipGetErrorString(he));
            }
            batch_no++;
            wq.push(std::move(b));
            clk.ns[ST_MLAUNCH] += StageClock::now() - m3;
        }
        if (last) break;
    }
    // shut the pipeline down (also on errors): unblock readers, drain the writer
    c1.reset();  // (a reader on the GPU waits in close() for every batch it handed out)
    c2.reset();
    pool1.stop();
    pool2.stop();
    q1.close();
    q2.close();
    wq.close();
    {   // what the readers had queued is dropped (a batch born on the GPU gives its text buffer back: its reader waits for that)
        std::unique_ptr<HalfBatch> drop;
        while (q1.pop(drop)) drop.reset();
        while (q2.pop(drop)) drop.reset();
    }
    t1.join();
    if (t2.joinable()) t2.join();
    tw.join();
    tf.join();
    if (tw2.joinable()) {
        {
            std::lock_guard<std::mutex> lk(w2_mu);
            w2_state = -1;
        }
        w2_cv.notify_all();
        tw2.join();
    }
    for (auto &s : slots) slot_free(s);
    if (const char *tr = getenv("NOHUMAN_TRACE")) {
        if (tr[0] == '1') {
            static const char *names[] = {"read1", "read2", "rpush1", "rpush2", "main.pop", "main.slot", "main.gather",
                                          "main.launch", "wr.pop", "wr.sync", "wr.format", "wr.write"};
            fprintf(stderr, "[nohuman trace] wall %.3fs |",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            for (int i = 0; i < 12; i++) fprintf(stderr, " %s %.3f", names[i], (double)clk.ns[i].load() * 1e-9);
            fprintf(stderr, "\n");
            if (g_pageable_batches.load())
                fprintf(stderr, "[nohuman trace] %d batch buffers could not be page-locked (pageable memory used)\n",
                        g_pageable_batches.load());
        }
    }
    if (rs.err_code != NH_OK) {
        free_run_counters();
        return set_error(rs.err_code, "%s", rs.err_msg.c_str());
    }
    if (dev_debug()) {  // NOHUMAN_DEBUG_DEVICE: a launch, copy or allocation under the wrong device fails the run that met it
        const std::string v = dev_violation(true);
        if (!v.empty()) {
            free_run_counters();
            return set_error(NH_EDEVICE, "device discipline: %s", v.c_str());
        }
    }

    // The run's only exchange step: the counters the classify kernels kept in each
    // device's HBM are summed by ONE all-reduce over RCCL, reduced where they lie.  The sums the writer kept
    // on the host are the checker.  A collective that cannot run at G > 1 is reported on stderr (one WARN
    // line; the host sums stand) and fails the run under NOHUMAN_RCCL=strict; NOHUMAN_RCCL=0 skips the
    // collective, =1 also runs it for a single device.
    {
        std::vector<uint64_t> rows(4 * (size_t)G, 0);
        uint64_t dsum[4] = {0, 0, 0, 0};
        hipError_t he = hipSuccess;
        for (int g = 0; g < G && he == hipSuccess; g++) {
            he = dev_set(engines[g]->device);
            if (he == hipSuccess) he = hipMemcpy(&rows[4 * g], rs.d_run_counters[g], 32, hipMemcpyDeviceToHost);
            for (int i = 0; i < 4; i++) dsum[i] += rows[4 * g + i];
            if (he == hipSuccess) {  // the engine's running totals (nh_stats_get) include this run
                hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(4), 0, engines[g]->stream,
                                   (unsigned long long *)engines[g]->d_counters, (const unsigned long long *)rs.d_run_counters[g]);
                he = hipStreamSynchronize(engines[g]->stream);
            }
        }
        int crc = NH_OK;
        if (he != hipSuccess) crc = set_error(NH_EDEVICE, "reading the run's counters: %s", hipGetErrorString(he));
        else if (dsum[CNT_FRAGMENTS] != rs.total || dsum[CNT_CLASSIFIED] != rs.classified || dsum[CNT_BASES] != rs.total_bases)
            crc = set_error(NH_EDEVICE, "the devices counted %llu fragments / %llu classified / %llu bases, the writer %llu / %llu / %llu",
                            (unsigned long long)dsum[0], (unsigned long long)dsum[1], (unsigned long long)dsum[2],
                            (unsigned long long)rs.total, (unsigned long long)rs.classified, (unsigned long long)rs.total_bases);
        const char *env = getenv("NOHUMAN_RCCL");
        const bool strict = env && strcmp(env, "strict") == 0;
        const bool want = env ? env[0] != '0' : G > 1;
        if (!crc && want && (G > 1 || env)) {
            std::vector<int> ids;
            for (Engine *e : engines) ids.push_back(e->device);
            std::vector<const uint64_t *> src(rs.d_run_counters.begin(), rs.d_run_counters.end());
            std::string backend;
            const int arc = allreduce_counters(ids.data(), G, rows.data(), backend, src.data());
            if (arc == NH_OK) {
                for (int g = 0; g < G && !crc; g++)
                    if (rows[4 * g] != rs.total || rows[4 * g + 1] != rs.classified || rows[4 * g + 2] != rs.total_bases)
                        crc = set_error(NH_EDEVICE, "count all-reduce disagrees with the host-side sum on device %d", ids[g]);
                if (getenv("NOHUMAN_TRACE")) fprintf(stderr, "[nohuman trace] counters reduced by %s\n", backend.c_str());
            } else if (strict) {
                crc = arc;
            } else {
                fprintf(stderr, "nohuman: WARN count all-reduce over %d devices did not run (%s); the counts were summed on the host\n",
                        G, g_last_error.c_str());
            }
        }
        free_run_counters();
        if (crc) return crc;
    }

    if (a->report && a->report[0] &&
        (rc = write_report(engines[0], a->report, rs.call_counts, rs.total, rs.total - rs.classified)))
        return rc;
    {  // (side by side as well: the last chunks of two GPU encoders, their buffers' release)
        int rc2 = NH_OK;
        std::string err2;
        std::thread t2;
        if (rs.paired)
            t2 = std::thread([&] {
                rc2 = o2.close();
                if (rc2) err2 = g_last_error;
            });
        rc = o1.close();
        if (t2.joinable()) t2.join();
        if (rc) return rc;
        if (rc2) return set_error(rc2, "%s", err2.c_str());
    }
    if (rs.want_k && (rc = ok.close())) return rc;
    if (stats) {
        nh_stats st;
        memset(&st, 0, sizeof st);
        st.total_sequences = rs.total;
        st.classified = rs.classified;
        st.unclassified = rs.total - rs.classified;
        st.total_bases = rs.total_bases;
        st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        *stats = st;
    }
    return NH_OK;
}

int run_engine(Engine *e, const nh_run_args *a, nh_stats *stats) {
    std::vector<Engine *> v{e};
    return run_engines(v, a, stats);
}

}  // namespace nh

extern "C" {

int nh_run_engine(nh_engine *e, const nh_run_args *args, nh_stats *stats) {
    if (!e) return nh::set_error(NH_EINVAL, "null engine");
    return nh::run_engine((nh::Engine *)e, args, stats);
}

int nh_fastx_scan(const char *path, uint64_t *n_records, uint64_t *n_bases, uint64_t *digest) {
    if (!path || !n_records || !n_bases || !digest) return nh::set_error(NH_EINVAL, "null argument");
    nh::BlockReader r;
    std::string err;
    if (r.open(path, err) != 0) return nh::set_error(NH_EIO, "%s", err.c_str());
    uint64_t n = 0, nb = 0, h = 0xcbf29ce484222325ull;
    auto mix = [&](const char *p, size_t len) {
        for (size_t i = 0; i < len; i++) h = (h ^ (unsigned char)p[i]) * 0x100000001b3ull;
        h = (h ^ 0) * 0x100000001b3ull;
    };
    size_t scan_batch = 4096;
    if (const char *env = getenv("NOHUMAN_SCAN_BATCH")) {  // test knob
        const long v = atol(env);
        if (v > 0) scan_batch = (size_t)v;
    }
    nh::HalfBatch hb;
    for (;;) {
        r.next_batch(hb, scan_batch, 64u << 20);
        if (!hb.error.empty()) return nh::set_error(NH_EIO, "%s", hb.error.c_str());
        for (const nh::RecRef &x : hb.recs) {
            n++;
            nb += x.slen;
            mix(hb.text.data() + x.h, x.hlen);
            mix(hb.text.data() + x.s, x.slen);
            mix(hb.text.data() + x.q, x.qlen);
        }
        if (hb.eof) break;
    }
    *n_records = n;
    *n_bases = nb;
    *digest = h;
    return NH_OK;
}

// Whole run on one or several devices: the database is loaded into every device's HBM, batches go
// round-robin, outputs stay in input order, the counts are summed on the host.
int nh_run(const nh_run_args *args, nh_stats *stats) {
    if (!args || !args->db_dir) return nh::set_error(NH_EINVAL, "nh_run: db_dir is required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nh::set_error(NH_EDEVICE, "no HIP device available");
    ndev = nh::dev_count();
    std::vector<int> devs;
    if (args->device_ids && args->n_devices > 0)
        devs.assign(args->device_ids, args->device_ids + args->n_devices);
    else
        for (int i = 0; i < (args->n_devices > 0 ? args->n_devices : ndev); i++) devs.push_back(i);
    // one replica of the database per device, loaded at the same time (each device has its own PCIe
    // link; the file comes from the page cache after the first reader)
    const bool trace = getenv("NOHUMAN_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    std::vector<nh::Engine *> opened(devs.size(), nullptr);
    std::vector<int> rcs(devs.size(), NH_OK);
    std::vector<std::string> errs(devs.size());
    {
        std::vector<std::thread> loaders;
        for (size_t i = 0; i < devs.size(); i++)
            loaders.emplace_back([&, i] {
                rcs[i] = nh::open_dir(args->db_dir, devs[i], &opened[i]);
                if (rcs[i]) errs[i] = nh::g_last_error;  // thread-local: carry it over
            });
        for (auto &t : loaders) t.join();
    }
    std::vector<nh::Engine *> engines;
    int rc = NH_OK;
    for (size_t i = 0; i < devs.size(); i++) {
        if (opened[i]) engines.push_back(opened[i]);
        if (rcs[i] && !rc) rc = nh::set_error(rcs[i], "%s", errs[i].c_str());
    }
    const double t_load = since(t_begin);
    const auto t_run = std::chrono::steady_clock::now();
    if (!rc) rc = nh::run_engines(engines, args, stats);
    const double s_run = since(t_run);
    std::string keep = nh::g_last_error;
    const auto t_close = std::chrono::steady_clock::now();
    for (nh::Engine *e : engines) nh::destroy(e);
    if (trace)
        fprintf(stderr, "[nohuman trace] nh_run: database into HBM %.3f s, the run %.3f s, closing (HBM and page-locked buffers given back) %.3f s\n", t_load,
                s_run, since(t_close));
    if (rc) nh::g_last_error = keep;
    return rc;
}

}  // extern "C"
