// nh_fastx.cpp -- see nh_fastx.h.  Record semantics follow SURVEY.md A.6 (kraken2 seqreader.cc).
#include "nh_fastx.h"
#include "nh_gunzip.h"
#include "nh_inflate.h"

#include <ctype.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace nh {

// One outstanding src.read() at a time, executed by a helper thread: the reader thread parses the stretch it
// already has while the next one is inflated / read straight into the same batch buffer (the copy out of the
// decoder's chunk buffers and the record parse each stream the whole text once; serialised on one thread they
// were the longest stage of a gzip run).
class ReadAhead {
public:
    explicit ReadAhead(ByteSource *src) : src_(src), th_([this] { loop(); }) {}
    ~ReadAhead() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void issue(uint8_t *dst, size_t cap) {
        std::lock_guard<std::mutex> lk(mu_);
        dst_ = dst;
        cap_ = cap;
        state_ = 1;
        cv_.notify_all();
    }
    bool pending() {
        std::lock_guard<std::mutex> lk(mu_);
        return state_ != 0;
    }
    long wait() {  // result of the issued read
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return state_ == 2; });
        state_ = 0;
        return n_;
    }

private:
    void loop() {
        for (;;) {
            uint8_t *d;
            size_t c;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || state_ == 1; });
                if (quit_) return;
                d = dst_;
                c = cap_;
            }
            const long n = src_->read(d, c);
            {
                std::lock_guard<std::mutex> lk(mu_);
                n_ = n;
                state_ = 2;
            }
            cv_.notify_all();
        }
    }
    ByteSource *src_;
    std::mutex mu_;
    std::condition_variable cv_;
    uint8_t *dst_ = nullptr;
    size_t cap_ = 0;
    long n_ = 0;
    int state_ = 0;  // 0 idle, 1 issued, 2 done
    bool quit_ = false;
    std::thread th_;
};

// bzip2 inputs (kraken2's wrapper pipes them through `bzip2 -dc`): decoded in process with the
// system's libbz2.so.1 -- the image has the library but not bzlib.h.  Concatenated streams are decoded
// one after the other as bzip2 -dc does; a truncated or damaged file is an ERROR, never a short read.
class Bz2Source {
public:
    struct Stream {
        char *next_in;
        unsigned avail_in, total_in_lo32, total_in_hi32;
        char *next_out;
        unsigned avail_out, total_out_lo32, total_out_hi32;
        void *state;
        void *(*bzalloc)(void *, int, int);
        void (*bzfree)(void *, void *);
        void *opaque;
    };
    ~Bz2Source() {
        if (live_) end_(&bs_);
        if (fd_ >= 0) ::close(fd_);
    }
    // libbz2 is resolved once per process (never unloaded)
    struct Api {
        int (*init)(Stream *, int, int) = nullptr;
        int (*run)(Stream *) = nullptr;
        int (*end)(Stream *) = nullptr;
        Api() {
            void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) return;
            init = (int (*)(Stream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
            run = (int (*)(Stream *))dlsym(h, "BZ2_bzDecompress");
            end = (int (*)(Stream *))dlsym(h, "BZ2_bzDecompressEnd");
        }
    };
    int open(const char *path, std::string &err) {
        static const Api api;
        init_ = api.init;
        run_ = api.run;
        end_ = api.end;
        if (!init_ || !run_ || !end_) {
            err = std::string("bzip2 input needs libbz2.so.1, which could not be loaded (") + path + ")";
            return -1;
        }
        fd_ = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd_ < 0) {
            err = std::string("cannot open ") + path;
            return -1;
        }
        path_ = path;
        in_.resize(1u << 20);
        return 0;
    }
    long read(uint8_t *buf, size_t cap, std::string &err) {
        size_t got = 0;
        while (got < cap && !done_) {
            if (bs_.avail_in == 0 && !in_eof_) {
                ssize_t n;
                do n = ::read(fd_, in_.data(), in_.size());
                while (n < 0 && errno == EINTR);
                if (n < 0) {
                    err = "read error on " + path_;
                    return -1;
                }
                if (n == 0) in_eof_ = true;
                bs_.next_in = in_.data();
                bs_.avail_in = (unsigned)n;
            }
            if (!live_) {
                if (bs_.avail_in == 0 && in_eof_) {  // clean end between streams
                    done_ = true;
                    break;
                }
                char *ni = bs_.next_in;
                const unsigned ai = bs_.avail_in;
                memset(&bs_, 0, sizeof bs_);
                bs_.next_in = ni;
                bs_.avail_in = ai;
                if (init_(&bs_, 0, 0) != 0) {
                    err = "bzip2: cannot start the decoder";
                    return -1;
                }
                live_ = true;
            }
            bs_.next_out = (char *)buf + got;
            const size_t room = cap - got < (1u << 30) ? cap - got : (1u << 30);
            bs_.avail_out = (unsigned)room;
            const int r = run_(&bs_);
            got += room - bs_.avail_out;
            if (r == 4) {  // BZ_STREAM_END: another stream may follow
                end_(&bs_);
                live_ = false;
                streams_++;
            } else if (r == -5 && streams_ > 0) {
                // BZ_DATA_ERROR_MAGIC at the start of a follow-on stream: bytes behind the last stream that
                // are no bzip2 stream.  `bzip2 -dc` (kraken2's pipe) warns "trailing garbage after EOF
                // ignored" and delivers the data: so does this reader
                fprintf(stderr, "nohuman: %s: trailing garbage after the last bzip2 stream ignored\n", path_.c_str());
                end_(&bs_);
                live_ = false;
                done_ = true;
            } else if (r != 0) {
                err = "bzip2: damaged input " + path_;
                return -1;
            } else if (bs_.avail_in == 0 && in_eof_ && bs_.avail_out != 0 && streams_ > 0 && bs_.total_in_lo32 < 4 &&
                       bs_.total_in_hi32 == 0 && bs_.total_out_lo32 == 0) {
                // the file ends inside what could have been the magic of another stream: trailing garbage too
                fprintf(stderr, "nohuman: %s: trailing garbage after the last bzip2 stream ignored\n", path_.c_str());
                end_(&bs_);
                live_ = false;
                done_ = true;
            } else if (bs_.avail_in == 0 && in_eof_ && bs_.avail_out != 0) {
                err = "bzip2: unexpected end of " + path_;
                return -1;
            }
        }
        return (long)got;
    }

private:
    int (*init_)(Stream *, int, int) = nullptr;
    int (*run_)(Stream *) = nullptr;
    int (*end_)(Stream *) = nullptr;
    Stream bs_ = {};
    std::vector<char> in_;
    std::string path_;
    int fd_ = -1;
    bool live_ = false, in_eof_ = false, done_ = false;
    unsigned streams_ = 0;  // complete streams decoded so far
};

// nh_internal.h (nh_engine.hip): allocations give back what the process keeps between runs before they fail; devices are logical
hipError_t dev_malloc(void **p, size_t bytes);
hipError_t dev_set(int ldev);

// gzip decoded on a GPU: DevGunzip (nh_gunzip.hip) leaves a piece of text in device memory, read() fetches it from
// there into the caller's buffer (the page-locked text buffer of a batch: one copy over PCIe, no inflate on the host).
// Two text buffers: a helper thread has the GPU decode the next piece while this one is being fetched.
class DevGzSource {
public:
    ~DevGzSource() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
        gz_.close();
        if (device_ >= 0) (void)dev_set(device_);
        for (Buf &b : buf_)
            if (b.d) (void)hipFree(b.d);
        if (stream_) (void)hipStreamDestroy(stream_);
        if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    }
    int open(const char *path, int device, std::string &err) {
        device_ = device;
        if (dev_set(device) != hipSuccess) {
            err = "hipSetDevice failed";
            return -1;
        }
        if (gz_.open(path, device, 0, 0, err) != 0) return -1;
        // text of a piece: 256 MiB of gzip at up to 10 : 1 (a piece that would not fit is cut down by the reader); small
        // files get small buffers
        room_ = (size_t)2560u << 20;
        struct stat st;
        if (stat(path, &st) == 0 && (uint64_t)st.st_size * 16 + ((size_t)64u << 20) < room_) room_ = (size_t)st.st_size * 16 + ((size_t)64u << 20);
        if (const char *e = getenv("NOHUMAN_GZDEV_ROOM")) room_ = std::max<size_t>((size_t)atoll(e), (size_t)1u << 20);
        for (;;) {
            const bool ok = dev_malloc((void **)&buf_[0].d, room_ + 64) == hipSuccess && dev_malloc((void **)&buf_[1].d, room_ + 64) == hipSuccess;
            if (ok) break;
            (void)hipGetLastError();
            for (Buf &b : buf_) {
                if (b.d) (void)hipFree(b.d);
                b.d = nullptr;
            }
            if (room_ <= ((size_t)128u << 20)) {
                err = "the gzip reader's text buffers cannot be had";
                return -1;
            }
            room_ /= 2;
        }
        if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking) != hipSuccess) {
            err = "cannot create streams";
            return -1;
        }
        th_ = std::thread([this] { produce(); });
        return 0;
    }
    long read(uint8_t *buf, size_t cap, std::string &err) {
        if (dev_set(device_) != hipSuccess) {
            err = "hipSetDevice failed";
            return -1;
        }
        size_t got = 0;
        while (got < cap) {
            Buf &b = buf_[take_];
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return b.full || done_; });
                if (!b.full) {  // the stream has ended (or failed)
                    if (!perr_.empty()) {
                        err = perr_;
                        return -1;
                    }
                    break;
                }
            }
            const size_t k = std::min(cap - got, b.len - b.off);
            if (hipMemcpyAsync(buf + got, b.d + b.off, k, hipMemcpyDeviceToHost, copy_stream_) != hipSuccess ||
                hipStreamSynchronize(copy_stream_) != hipSuccess) {
                err = "gzip reader: fetching the text from the device failed";
                return -1;
            }
            b.off += k;
            got += k;
            if (b.off == b.len) {
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    b.full = false;
                }
                cv_.notify_all();
                take_ ^= 1;
            }
        }
        return (long)got;
    }

private:
    struct Buf {
        uint8_t *d = nullptr;
        size_t len = 0, off = 0;
        bool full = false;
    };
    void produce() {
        int fill = 0;
        for (;;) {
            Buf &b = buf_[fill];
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !b.full || stop_; });
                if (stop_) return;
            }
            const long n = gz_.next(b.d, room_, stream_);
            std::lock_guard<std::mutex> lk(mu_);
            if (n <= 0) {
                if (n < 0) perr_ = gz_.error();
                done_ = true;
                cv_.notify_all();
                return;
            }
            b.len = (size_t)n;
            b.off = 0;
            b.full = true;
            cv_.notify_all();
            fill ^= 1;
        }
    }
    DevGunzip gz_;
    int device_ = -1;
    Buf buf_[2];
    size_t room_ = 0;
    int take_ = 0;
    hipStream_t stream_ = nullptr, copy_stream_ = nullptr;
    std::thread th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false, done_ = false;
    std::string perr_;
};

ByteSource::~ByteSource() { close(); }

int ByteSource::open(const char *path, std::string &err, unsigned gz_threads, int device) {
    close();
    FILE *f = fopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path;
        return -1;
    }
    unsigned char magic[3] = {0, 0, 0};
    size_t got = fread(magic, 1, 3, f);
    fclose(f);
    if (got == 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h') {
        bz_ = new Bz2Source();
        if (bz_->open(path, err) != 0) {
            delete bz_;
            bz_ = nullptr;
            return -1;
        }
        return 0;
    }
    if (!(got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {  // plain text: no inflate layer
        fd_ = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd_ < 0) {
            err = std::string("cannot open ") + path;
            return -1;
        }
        (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
        return 0;
    }
    {   // the reader on the GPU (default wherever a device is at hand; NOHUMAN_GZ_READER=host keeps the host decoders)
        const char *how = getenv("NOHUMAN_GZ_READER");
        if (device >= 0 && !(how && !strcmp(how, "host")) && dev_gunzip_wants(path)) {
            dgz_ = new DevGzSource();
            std::string derr;
            if (dgz_->open(path, device, derr) == 0) return 0;
            delete dgz_;
            dgz_ = nullptr;
            if (how && !strcmp(how, "device")) {  // asked for by name: no silent change of reader
                err = derr;
                return -1;
            }
            fprintf(stderr, "nohuman: WARN %s: the gzip reader on GPU %d could not be set up (%s); inflating on the host\n", path, device,
                    derr.c_str());
        }
    }
    if (const char *env = getenv("NOHUMAN_GZ_THREADS")) gz_threads = (unsigned)atoi(env);  // 0 = zlib
    if (gz_threads > 0) {
        size_t chunk = 0;
        if (const char *env = getenv("NOHUMAN_GZ_CHUNK")) chunk = (size_t)atol(env);  // test knob
        pgz_ = new ParallelGunzip();
        std::string perr;
        if (pgz_->open(path, gz_threads, chunk, perr) == 0) return 0;
        delete pgz_;  // e.g. not a regular file: zlib takes it
        pgz_ = nullptr;
    }
    gzFile g = gzopen(path, "rb");
    if (!g) {
        err = std::string("cannot open ") + path;
        return -1;
    }
    gzbuffer(g, 1u << 20);
    gz_ = g;
    return 0;
}

long ByteSource::read(uint8_t *buf, size_t cap) {
    if (dgz_) return dgz_->read(buf, cap, pgz_error_);
    if (pgz_) {
        long n = pgz_->read(buf, cap);
        if (n < 0) pgz_error_ = pgz_->error();
        return n;
    }
    if (fd_ >= 0) {
        size_t got = 0;
        while (got < cap) {  // fill the request like gzread / fread do
            ssize_t n = ::read(fd_, buf + got, cap - got);
            if (n < 0) {
                if (errno == EINTR) continue;
                return -1;
            }
            if (n == 0) break;
            got += (size_t)n;
        }
        return (long)got;
    }
    if (gz_) {
        int n = gzread((gzFile)gz_, buf, (unsigned)cap);
        if (n < 0) {
            int code = 0;
            const char *msg = gzerror((gzFile)gz_, &code);
            pgz_error_ = std::string("gzip: ") + (msg ? msg : "read error");
        }
        return n;
    }
    if (bz_) return bz_->read(buf, cap, pgz_error_);
    return -1;
}

void ByteSource::close() {
    if (gz_) gzclose((gzFile)gz_);
    delete bz_;
    bz_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    delete pgz_;
    pgz_ = nullptr;
    delete dgz_;
    dgz_ = nullptr;
    fd_ = -1;
    gz_ = nullptr;
}

int FastxReader::open(const char *path, std::string &err) {
    buf_.resize(4u << 20);
    pos_ = len_ = 0;
    eof_ = false;
    format_ = FMT_AUTO;
    return src_.open(path, err);
}

int FastxReader::peek() {
    if (pos_ == len_) {
        if (eof_) return -1;
        long n = src_.read(buf_.data(), buf_.size());
        if (n <= 0) {
            eof_ = true;
            return -1;
        }
        pos_ = 0;
        len_ = (size_t)n;
    }
    return buf_[pos_];
}

// std::getline semantics: false only if no character could be extracted
bool FastxReader::getline(std::string &line) {
    line.clear();
    bool any = false;
    for (;;) {
        if (peek() < 0) return any;
        any = true;
        const uint8_t *p = buf_.data() + pos_;
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', len_ - pos_);
        if (nl) {
            line.append((const char *)p, (size_t)(nl - p));
            pos_ += (size_t)(nl - p) + 1;
            return true;
        }
        line.append((const char *)p, len_ - pos_);
        pos_ = len_;
    }
}

static void strip(std::string &s) {
    while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
}

int FastxReader::next(SeqRecord &rec, std::string &err) {
    if (!getline(line_)) return 0;
    strip(line_);
    if (format_ == FMT_AUTO) {
        if (!line_.empty() && line_[0] == '@')
            format_ = FMT_FASTQ;
        else if (!line_.empty() && line_[0] == '>')
            format_ = FMT_FASTA;
        else {
            err = "sequence reader - unrecognized file format";
            return -1;
        }
    }
    rec.format = format_;
    if (format_ == FMT_FASTQ) {
        if (line_.empty()) return 0;  // an empty line may end the file
        if (line_[0] != '@') {
            err = "malformed FASTQ file (exp. '@', saw \"" + line_ + "\"), aborting";
            return -1;
        }
    } else {
        if (line_.empty() || line_[0] != '>') {
            err = "malformed FASTA file (exp. '>', saw \"" + line_ + "\"), aborting";
            return -1;
        }
    }
    rec.header = line_;
    if (line_.size() <= 1) return 0;
    size_t ws = line_.find_first_of(" \t\r", 1);
    rec.id.assign(line_, 1, ws == std::string::npos ? std::string::npos : ws - 1);
    if (format_ == FMT_FASTQ) {
        if (!getline(line_)) return 0;
        strip(line_);
        rec.seq = line_;
        if (!getline(line_)) return 0;  // '+' line, discarded
        if (!getline(line_)) return 0;
        strip(line_);
        rec.quals = line_;
    } else {
        rec.quals.clear();
        rec.seq.clear();
        for (;;) {
            int c = peek();
            if (c < 0 || c == '>') break;
            if (!getline(line_)) break;
            strip(line_);
            rec.seq += line_;
        }
    }
    return 1;
}

// ------------------------------------------------------------------------------------------------
int BlockReader::open(const char *path, std::string &err, unsigned gz_threads, int device) {
    tail_.clear();
    eof_ = false;
    format_ = FMT_AUTO;
    fa_resume_rec_ = (size_t)-1;
    chunk_ = 4u << 20;
    if (const char *env = getenv("NOHUMAN_READ_CHUNK")) {  // test knob: records across read boundaries
        const long v = atol(env);
        if (v > 0) chunk_ = (size_t)v;
    }
    return src_.open(path, err, gz_threads, device);
}

static inline const char *rstrip(const char *b, const char *e) {
    while (e > b && isspace((unsigned char)e[-1])) e--;
    return e;
}

namespace {
struct Line {
    const char *b, *e;  // [b, e) without the newline
    size_t next;        // offset of the byte after the line
    bool term;          // ended by a newline (false: ended by the end of the input)
};
}  // namespace

// std::getline semantics on text[p..len): 1 = a line, 0 = the line is not complete yet (more input
// to come), -1 = end of input and not a single character left
static inline int get_line(const char *t, size_t p, size_t len, bool eof, Line &L) {
    const char *nl = (const char *)memchr(t + p, '\n', len - p);
    if (nl) {
        L.b = t + p;
        L.e = nl;
        L.next = (size_t)(nl - t) + 1;
        L.term = true;
        return 1;
    }
    if (!eof) return 0;
    if (p == len) return -1;
    L.b = t + p;
    L.e = t + len;
    L.next = len;
    L.term = false;
    return 1;
}

int BlockReader::parse_one(HalfBatch &hb, size_t &pos, size_t len) {
    char *t = hb.text.data();
    Line h;
    int g = get_line(t, pos, len, eof_, h);
    if (g == 0) return 0;
    if (g < 0) {
        hb.eof = true;
        return -1;
    }
    const char *he = rstrip(h.b, h.e);
    if (format_ == FMT_AUTO) {
        if (he > h.b && *h.b == '@')
            format_ = FMT_FASTQ;
        else if (he > h.b && *h.b == '>')
            format_ = FMT_FASTA;
        else {
            hb.error = "sequence reader - unrecognized file format";
            return -2;
        }
        hb.format = format_;
    }
    if (format_ == FMT_FASTQ) {
        if (he == h.b) {  // an empty line may end the file
            hb.eof = true;
            return -1;
        }
        if (*h.b != '@') {
            hb.error = "malformed FASTQ file (exp. '@', saw \"" + std::string(h.b, he) + "\"), aborting";
            return -2;
        }
    } else if (he == h.b || *h.b != '>') {
        hb.error = "malformed FASTA file (exp. '>', saw \"" + std::string(h.b, he) + "\"), aborting";
        return -2;
    }
    if (he - h.b <= 1) {
        hb.eof = true;
        return -1;
    }
    RecRef r;
    r.h = (uint32_t)(h.b - t);
    r.hlen = (uint32_t)(he - h.b);
    const char *ws = h.b + 1;
    while (ws < he && *ws != ' ' && *ws != '\t' && *ws != '\r') ws++;
    r.idlen = (uint32_t)(ws - h.b - 1);
    bool canon = h.term && he == h.e;
    if (format_ == FMT_FASTQ) {
        Line s, pl, q;
        for (Line *L : {&s, &pl, &q}) {  // sequence, '+' line (discarded), qualities
            g = get_line(t, L == &s ? h.next : L == &pl ? s.next : pl.next, len, eof_, *L);
            if (g == 0) return 0;
            if (g < 0) {  // the file ends inside the record: kraken2 drops it
                hb.eof = true;
                return -1;
            }
        }
        const char *se = rstrip(s.b, s.e), *qe = rstrip(q.b, q.e);
        r.s = (uint32_t)(s.b - t);
        r.slen = (uint32_t)(se - s.b);
        r.q = (uint32_t)(q.b - t);
        r.qlen = (uint32_t)(qe - q.b);
        canon = canon && se == s.e && qe == q.e && q.term && pl.e - pl.b == 1 && *pl.b == '+';
        r.raw_end = canon ? (uint32_t)q.next : 0;
        pos = q.next;
    } else {
        // pass 1: find the end of the record (next line starting with '>' or the end of input)
        size_t p = fa_resume_rec_ == pos ? fa_resume_ : h.next;
        for (;;) {
            if (p == len) {
                if (eof_) break;
                fa_resume_rec_ = pos;
                fa_resume_ = p;
                return 0;
            }
            if (t[p] == '>') break;
            Line L;
            if (get_line(t, p, len, eof_, L) == 0) {
                fa_resume_rec_ = pos;
                fa_resume_ = p;
                return 0;
            }
            p = L.next;
        }
        fa_resume_rec_ = (size_t)-1;
        // pass 2: join the sequence lines in place
        char *dst = t + h.next;
        size_t nlines = 0;
        bool plain = true;
        for (size_t x = h.next; x < p;) {
            Line L;
            get_line(t, x, p, true, L);
            const char *le = rstrip(L.b, L.e);
            const size_t n = (size_t)(le - L.b);
            if (dst != L.b) memmove(dst, L.b, n);
            dst += n;
            plain = plain && le == L.e && L.term;
            nlines++;
            x = L.next;
        }
        r.s = (uint32_t)h.next;
        r.slen = (uint32_t)(dst - (t + h.next));
        r.q = r.s + r.slen;
        r.qlen = 0;
        r.raw_end = canon && plain && nlines == 1 ? (uint32_t)p : 0;
        pos = p;
    }
    hb.recs.push_back(r);
    return 1;
}

BlockReader::~BlockReader() { close(); }

void BlockReader::close() {
    delete ahead_;  // joins the helper: no read may be in flight when the source goes away
    ahead_ = nullptr;
    src_.close();
}

void BlockReader::next_batch(HalfBatch &hb, size_t max_recs, size_t max_text) {
    const size_t CHUNK = chunk_;
    const size_t LIMIT = 0xFFF00000ull;  // offsets are 32-bit
    hb.reset();
    hb.format = format_;
    if (max_text > (3ull << 30)) max_text = 3ull << 30;
    if (!hb.text.reserve(tail_.size() + CHUNK)) {
        hb.error = "out of memory";
        return;
    }
    if (tail_.size()) memcpy(hb.text.data(), tail_.data(), tail_.size());
    hb.text.set_size(tail_.size());
    tail_.clear();
    if (fa_resume_rec_ != (size_t)-1) fa_resume_rec_ = (size_t)-1;  // offsets changed: rescan
    if (!ahead_) ahead_ = new ReadAhead(&src_);
    size_t pos = 0;
    bool inflight = false;
    // the next stretch of input is read into the buffer (by the helper) while this thread parses what is there
    auto issue = [&]() -> bool {
        if (eof_ || inflight) return true;
        if (hb.text.size() + CHUNK > LIMIT) {
            hb.error = "sequence record larger than 4 GB";
            return false;
        }
        if (!hb.text.reserve(hb.text.size() + CHUNK)) {  // (never while a read is in flight: it may move the buffer)
            hb.error = "out of memory";
            return false;
        }
        ahead_->issue((uint8_t *)hb.text.data() + hb.text.size(), CHUNK);
        inflight = true;
        return true;
    };
    auto collect = [&]() -> bool {  // waits for the read in flight and appends what it brought
        if (!inflight) return true;
        inflight = false;
        const long n = ahead_->wait();
        if (n < 0) {
            hb.error = src_.error().empty() ? std::string("read error on input file") : src_.error();
            return false;
        }
        if (n == 0)
            eof_ = true;
        else
            hb.text.set_size(hb.text.size() + (size_t)n);
        return true;
    };
    for (;;) {
        if (!issue()) break;
        int r = 1;
        const size_t avail = hb.text.size();  // (the helper writes beyond it)
        while (hb.recs.size() < max_recs && pos < max_text) {
            r = parse_one(hb, pos, avail);
            if (r != 1) break;
        }
        if (r < 0) {  // end of input or malformed
            if (inflight) (void)ahead_->wait();
            hb.text.set_size(pos);
            return;
        }
        if (r == 1) break;  // batch full
        if (!collect()) return;  // more input needed
    }
    if (!hb.error.empty()) {
        if (inflight) (void)ahead_->wait();
        return;
    }
    if (!collect()) return;
    // keep what follows the last complete record for the next batch
    if (!tail_.append(hb.text.data() + pos, hb.text.size() - pos)) {
        hb.error = "out of memory";
        return;
    }
    hb.text.set_size(pos);
}

}  // namespace nh
