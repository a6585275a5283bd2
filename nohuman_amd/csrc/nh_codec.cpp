// nh_codec.cpp -- output compression stage of the nohuman host.
//
// Mirrors CompressionFormat::compress (/root/reference/src/compression.rs:182-200): gzip with
// `threads` workers (gzip_compress, compression.rs:214-233, gzp's block-parallel encoder at the
// default level), bzip2 single-threaded through libbz2 (compression.rs:202-212), xz multi-threaded
// through liblzma (compression.rs:235-252), zstd through libzstd -- every one a StreamEncoder
// (nh_codec.h), so that nh_run can feed kept records to it directly.  Parity target is the decompressed content and the container magic
// (compression.rs:282-288), not byte-identical streams.
//
// gzip: the input is cut into 512 KiB blocks; each worker deflates one block as a raw deflate stream
// primed with the previous block's last 32 KiB as dictionary and closed with a sync flush (byte
// aligned, not final); the blocks are written in order inside ONE gzip member whose CRC-32 is
// combined from the per-block CRCs.  Any gzip reader sees an ordinary single-member file.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "nh_codec.h"
#include "nh_inflate.h"
#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {

namespace {

const size_t GZ_BLOCK = 512u << 10;
const size_t GZ_DICT = 32u << 10;

struct GzJob {
    std::vector<unsigned char> in;    // [dict | block]
    size_t dict_len = 0, len = 0;
    std::vector<unsigned char> out;
    size_t out_len = 0;
    uint32_t crc = 0;
    bool done = false, failed = false;
};

struct GzShared {
    std::mutex mu;
    std::condition_variable work_cv, done_cv;
    std::deque<GzJob *> pending;
    bool quit = false;
    int level = 6;
};

void gz_worker(GzShared *sh) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    bool init = false;
    for (;;) {
        GzJob *j;
        {
            std::unique_lock<std::mutex> lk(sh->mu);
            sh->work_cv.wait(lk, [&] { return sh->quit || !sh->pending.empty(); });
            if (sh->pending.empty()) break;
            j = sh->pending.front();
            sh->pending.pop_front();
        }
        bool ok = true;
        if (!init) {
            ok = deflateInit2(&zs, sh->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
            init = ok;
        } else {
            ok = deflateReset(&zs) == Z_OK;
        }
        const unsigned char *block = j->in.data() + j->dict_len;
        if (ok && j->dict_len) ok = deflateSetDictionary(&zs, j->in.data(), (uInt)j->dict_len) == Z_OK;
        if (ok) {
            j->out.resize(deflateBound(&zs, (uLong)j->len) + 64);
            zs.next_in = (Bytef *)block;
            zs.avail_in = (uInt)j->len;
            zs.next_out = j->out.data();
            zs.avail_out = (uInt)j->out.size();
            const int rc = deflate(&zs, Z_SYNC_FLUSH);
            ok = rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0;
            j->out_len = j->out.size() - zs.avail_out;
            j->crc = crc32_fast(0, (const uint8_t *)block, j->len);  // (carry-less multiply where the CPU has it: 5x zlib's)
        }
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            j->failed = !ok;
            j->done = true;
        }
        sh->done_cv.notify_all();
    }
    if (init) deflateEnd(&zs);
}

bool write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
        ssize_t w = ::write(fd, c, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += w;
        n -= (size_t)w;
    }
    return true;
}

long read_full(int fd, void *p, size_t n) {
    size_t got = 0;
    while (got < n) {
        ssize_t r = ::read(fd, (char *)p + got, n - got);
        if (r < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (r == 0) break;
        got += (size_t)r;
    }
    return (long)got;
}

// ---- gzip: block-parallel, streaming ------------------------------------------------------------------
class GzipEncoder : public StreamEncoder {
public:
    GzipEncoder(int fd, unsigned threads, const char *name) : fd_(fd), name_(name) {
        if (threads < 1) threads = 1;
        for (unsigned i = 0; i < threads; i++) pool_.emplace_back(gz_worker, &sh_);
        max_inflight_ = 2 * (size_t)threads + 2;
        static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};  // no name, no mtime, unix
        if (!write_all(fd_, header, sizeof header)) rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
        crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
    }
    ~GzipEncoder() override {
        while (!inflight_.empty()) retire_head();  // workers still hold pointers
        {
            std::lock_guard<std::mutex> lk(sh_.mu);
            sh_.quit = true;
        }
        sh_.work_cv.notify_all();
        for (auto &t : pool_) t.join();
    }
    int write(const void *p, size_t n) override {
        const unsigned char *c = (const unsigned char *)p;
        while (n && rc_ == NH_OK) {
            if (!cur_) begin_block();
            const size_t room = GZ_BLOCK - cur_->len;
            const size_t take = n < room ? n : room;
            memcpy(cur_->in.data() + cur_->dict_len + cur_->len, c, take);
            cur_->len += take;
            c += take;
            n -= take;
            if (cur_->len == GZ_BLOCK) submit();
        }
        return rc_;
    }
    int finish() override {
        if (cur_ && cur_->len) submit();
        while (!inflight_.empty()) retire_head();
        if (rc_ != NH_OK) return rc_;
        unsigned char tail[10] = {0x03, 0x00};  // final block: fixed Huffman, end-of-block only
        for (int i = 0; i < 4; i++) {
            tail[2 + i] = (unsigned char)(crc_ >> (8 * i));
            tail[6 + i] = (unsigned char)((uint32_t)total_ >> (8 * i));
        }
        if (!write_all(fd_, tail, sizeof tail)) rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
        return rc_;
    }

private:
    void begin_block() {
        if (!spare_.empty()) {
            cur_ = std::move(spare_.back());
            spare_.pop_back();
        } else {
            cur_.reset(new GzJob());
        }
        cur_->done = cur_->failed = false;
        cur_->dict_len = dict_.size();
        cur_->len = 0;
        cur_->in.resize(cur_->dict_len + GZ_BLOCK);
        if (cur_->dict_len) memcpy(cur_->in.data(), dict_.data(), cur_->dict_len);
    }
    void submit() {
        GzJob *j = cur_.get();
        const size_t have = j->dict_len + j->len, keep = have < GZ_DICT ? have : GZ_DICT;
        dict_.assign(j->in.data() + have - keep, j->in.data() + have);
        {
            std::lock_guard<std::mutex> lk(sh_.mu);
            sh_.pending.push_back(j);
        }
        sh_.work_cv.notify_one();
        inflight_.push_back(std::move(cur_));
        while (inflight_.size() >= max_inflight_) retire_head();
    }
    void retire_head() {  // wait for the oldest block and write it
        GzJob *j = inflight_.front().get();
        {
            std::unique_lock<std::mutex> lk(sh_.mu);
            sh_.done_cv.wait(lk, [&] { return j->done; });
        }
        if (rc_ == NH_OK) {
            if (j->failed)
                rc_ = set_error(NH_EIO, "deflate failed on %s", name_.c_str());
            else if (!write_all(fd_, j->out.data(), j->out_len))
                rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
        }
        crc_ = (uint32_t)crc32_combine(crc_, j->crc, (z_off_t)j->len);
        total_ += j->len;
        spare_.push_back(std::move(inflight_.front()));
        inflight_.pop_front();
    }
    int fd_;
    std::string name_;
    GzShared sh_;
    std::vector<std::thread> pool_;
    std::deque<std::unique_ptr<GzJob>> inflight_;
    std::vector<std::unique_ptr<GzJob>> spare_;
    std::unique_ptr<GzJob> cur_;
    std::vector<unsigned char> dict_;
    size_t max_inflight_ = 4;
    uint32_t crc_ = 0;
    uint64_t total_ = 0;
    int rc_ = NH_OK;
};

class PlainEncoder : public StreamEncoder {
public:
    PlainEncoder(int fd, const char *name) : fd_(fd), name_(name) {}
    int write(const void *p, size_t n) override {
        return write_all(fd_, p, n) ? NH_OK : set_error(NH_EIO, "write error on %s", name_.c_str());
    }
    int finish() override { return NH_OK; }

private:
    int fd_;
    std::string name_;
};

// ---- zstd through the system's libzstd.so.1 (the image has the library but not its header; the few
// entry points of the stable streaming API are declared here) -- zstd_compress of the reference:
// default level 3, `threads` workers, frame checksum (compression.rs:254-268)
struct ZstdApi {
    struct InBuf { const void *src; size_t size, pos; };
    struct OutBuf { void *dst; size_t size, pos; };
    void *(*createCCtx)(void) = nullptr;
    size_t (*freeCCtx)(void *) = nullptr;
    size_t (*setParameter)(void *, int, int) = nullptr;
    size_t (*compressStream2)(void *, OutBuf *, InBuf *, int) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char *(*getErrorName)(size_t) = nullptr;
    bool ok = false;
    ZstdApi() {
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        createCCtx = (void *(*)(void))dlsym(h, "ZSTD_createCCtx");
        freeCCtx = (size_t(*)(void *))dlsym(h, "ZSTD_freeCCtx");
        setParameter = (size_t(*)(void *, int, int))dlsym(h, "ZSTD_CCtx_setParameter");
        compressStream2 = (size_t(*)(void *, OutBuf *, InBuf *, int))dlsym(h, "ZSTD_compressStream2");
        isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        getErrorName = (const char *(*)(size_t))dlsym(h, "ZSTD_getErrorName");
        ok = createCCtx && freeCCtx && setParameter && compressStream2 && isError && getErrorName;
    }
};
const ZstdApi &zstd_api() {
    static const ZstdApi Z;
    return Z;
}

class ZstdEncoder : public StreamEncoder {
public:
    ZstdEncoder(int fd, unsigned threads, const char *name) : fd_(fd), name_(name), obuf_(1u << 20) {
        const ZstdApi &Z = zstd_api();
        enum { C_LEVEL = 100, C_CHECKSUM = 201, C_WORKERS = 400 };
        cctx_ = Z.createCCtx();
        if (!cctx_) {
            rc_ = set_error(NH_EOOM, "ZSTD_createCCtx failed");
            return;
        }
        (void)Z.setParameter(cctx_, C_LEVEL, 3);
        (void)Z.setParameter(cctx_, C_CHECKSUM, 1);
        if (threads > 1) (void)Z.setParameter(cctx_, C_WORKERS, (int)threads);  // ignored by single-threaded builds
    }
    ~ZstdEncoder() override {
        if (cctx_) zstd_api().freeCCtx(cctx_);
    }
    int write(const void *p, size_t n) override { return pump(p, n, false); }
    int finish() override { return pump(nullptr, 0, true); }

private:
    int pump(const void *p, size_t n, bool last) {
        const ZstdApi &Z = zstd_api();
        enum { E_CONTINUE = 0, E_END = 2 };
        ZstdApi::InBuf in = {p, n, 0};
        while (rc_ == NH_OK) {
            ZstdApi::OutBuf out = {obuf_.data(), obuf_.size(), 0};
            const size_t left = Z.compressStream2(cctx_, &out, &in, last ? E_END : E_CONTINUE);
            if (Z.isError(left)) {
                rc_ = set_error(NH_EIO, "zstd: %s", Z.getErrorName(left));
                break;
            }
            if (out.pos && !write_all(fd_, obuf_.data(), out.pos)) {
                rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
                break;
            }
            if (last ? left == 0 : in.pos == in.size) break;
        }
        return rc_;
    }
    int fd_;
    std::string name_;
    std::vector<char> obuf_;
    void *cctx_ = nullptr;
    int rc_ = NH_OK;
};

// ---- bzip2 through libbz2.so.1 (bzip2_compress of the reference, compression.rs:202-212: one thread,
// bzip2::Compression::default() = block size 6).  The image has the library but not bzlib.h.
struct Bz2Api {
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
    int (*init)(Stream *, int, int, int) = nullptr;
    int (*compress)(Stream *, int) = nullptr;
    int (*end)(Stream *) = nullptr;
    bool ok = false;
    Bz2Api() {
        void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        init = (int (*)(Stream *, int, int, int))dlsym(h, "BZ2_bzCompressInit");
        compress = (int (*)(Stream *, int))dlsym(h, "BZ2_bzCompress");
        end = (int (*)(Stream *))dlsym(h, "BZ2_bzCompressEnd");
        ok = init && compress && end;
    }
};
const Bz2Api &bz2_api() {
    static const Bz2Api B;
    return B;
}

class Bzip2Encoder : public StreamEncoder {
public:
    Bzip2Encoder(int fd, const char *name) : fd_(fd), name_(name), obuf_(1u << 20) {
        memset(&bs_, 0, sizeof bs_);
        if (bz2_api().init(&bs_, 6, 0, 0) != 0) rc_ = set_error(NH_EOOM, "BZ2_bzCompressInit failed");
        else live_ = true;
    }
    ~Bzip2Encoder() override {
        if (live_) bz2_api().end(&bs_);
    }
    int write(const void *p, size_t n) override {
        const char *c = (const char *)p;
        while (n && rc_ == NH_OK) {  // avail_in is 32-bit
            const size_t take = n < (1u << 30) ? n : (1u << 30);
            pump(c, take, 0 /* BZ_RUN */);
            c += take;
            n -= take;
        }
        return rc_;
    }
    int finish() override { return pump(nullptr, 0, 2 /* BZ_FINISH */); }

private:
    int pump(const char *p, size_t n, int action) {
        bs_.next_in = (char *)p;
        bs_.avail_in = (unsigned)n;
        while (rc_ == NH_OK) {
            bs_.next_out = obuf_.data();
            bs_.avail_out = (unsigned)obuf_.size();
            const int r = bz2_api().compress(&bs_, action);
            if (r < 0) {
                rc_ = set_error(NH_EIO, "bzip2 compressor failed on %s (%d)", name_.c_str(), r);
                break;
            }
            const size_t got = obuf_.size() - bs_.avail_out;
            if (got && !write_all(fd_, obuf_.data(), got)) {
                rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
                break;
            }
            if (action == 0 ? bs_.avail_in == 0 : r == 4 /* BZ_STREAM_END */) break;
        }
        return rc_;
    }
    int fd_;
    std::string name_;
    std::vector<char> obuf_;
    Bz2Api::Stream bs_;
    bool live_ = false;
    int rc_ = NH_OK;
};

// ---- xz through liblzma.so.5 (xz_compress of the reference, compression.rs:235-252: multi-threaded
// stream encoder, preset 6, CRC64).  The image has the library but not lzma.h.
struct LzmaApi {
    struct Stream {
        const unsigned char *next_in;
        size_t avail_in;
        uint64_t total_in;
        unsigned char *next_out;
        size_t avail_out;
        uint64_t total_out;
        const void *allocator;
        void *internal;
        void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
        uint64_t reserved_int1, reserved_int2;
        size_t reserved_int3, reserved_int4;
        int reserved_enum1, reserved_enum2;
    };
    struct Mt {
        uint32_t flags, threads;
        uint64_t block_size;
        uint32_t timeout, preset;
        const void *filters;
        int check;
        int reserved_enum1, reserved_enum2, reserved_enum3;
        uint32_t reserved_int1, reserved_int2, reserved_int3, reserved_int4;
        uint64_t reserved_int5, reserved_int6, reserved_int7, reserved_int8;
        void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    };
    int (*encoder_mt)(Stream *, const Mt *) = nullptr;
    int (*code)(Stream *, int) = nullptr;
    void (*end)(Stream *) = nullptr;
    bool ok = false;
    LzmaApi() {
        void *h = dlopen("liblzma.so.5", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        encoder_mt = (int (*)(Stream *, const Mt *))dlsym(h, "lzma_stream_encoder_mt");
        code = (int (*)(Stream *, int))dlsym(h, "lzma_code");
        end = (void (*)(Stream *))dlsym(h, "lzma_end");
        ok = encoder_mt && code && end;
    }
};
const LzmaApi &lzma_api() {
    static const LzmaApi X;
    return X;
}

class XzEncoder : public StreamEncoder {
public:
    XzEncoder(int fd, unsigned threads, const char *name) : fd_(fd), name_(name), obuf_(1u << 20) {
        memset(&ls_, 0, sizeof ls_);
        LzmaApi::Mt mt;
        memset(&mt, 0, sizeof mt);
        mt.threads = threads ? threads : 1;
        mt.preset = 6;
        mt.check = 4;  // LZMA_CHECK_CRC64
        const int r = lzma_api().encoder_mt(&ls_, &mt);
        if (r != 0) rc_ = set_error(NH_EOOM, "lzma_stream_encoder_mt failed (%d)", r);
        else live_ = true;
    }
    ~XzEncoder() override {
        if (live_) lzma_api().end(&ls_);
    }
    int write(const void *p, size_t n) override { return pump(p, n, 0 /* LZMA_RUN */); }
    int finish() override { return pump(nullptr, 0, 3 /* LZMA_FINISH */); }

private:
    int pump(const void *p, size_t n, int action) {
        ls_.next_in = (const unsigned char *)p;
        ls_.avail_in = n;
        while (rc_ == NH_OK) {
            ls_.next_out = (unsigned char *)obuf_.data();
            ls_.avail_out = obuf_.size();
            const int r = lzma_api().code(&ls_, action);
            if (r != 0 && r != 1) {  // LZMA_OK, LZMA_STREAM_END
                rc_ = set_error(NH_EIO, "xz compressor failed on %s (%d)", name_.c_str(), r);
                break;
            }
            const size_t got = obuf_.size() - ls_.avail_out;
            if (got && !write_all(fd_, obuf_.data(), got)) {
                rc_ = set_error(NH_EIO, "write error on %s", name_.c_str());
                break;
            }
            if (action == 0 ? (ls_.avail_in == 0 && ls_.avail_out != 0) : r == 1) break;
        }
        return rc_;
    }
    int fd_;
    std::string name_;
    std::vector<char> obuf_;
    LzmaApi::Stream ls_;
    bool live_ = false;
    int rc_ = NH_OK;
};

}  // namespace

StreamEncoder *make_encoder(int codec, int fd, unsigned threads, const char *name, int device) {
    switch (codec) {
        case NH_CODEC_NONE:
            return new PlainEncoder(fd, name);
        case NH_CODEC_GZIP: {
            const char *how = getenv("NOHUMAN_GZIP");
            if (device >= 0 && !(how && !strcmp(how, "host"))) {
                StreamEncoder *e = make_gpu_gzip_encoder(fd, device, name);
                if (e) return e;
                // The encoder's buffers (about 1 GiB of HBM, 0.5 GiB page-locked) could not be had -- a large database
                // or a small host.  Loud, never silent: one WARN line and the host encoder (the same gzip container,
                // zlib blocks), or a failed run when the GPU encoder was asked for by name (NOHUMAN_GZIP=gpu).
                const std::string why = g_last_error;
                if (how && !strcmp(how, "gpu")) {
                    set_error(NH_EDEVICE, "%s (NOHUMAN_GZIP=gpu: no fallback to the host encoder)", why.c_str());
                    return nullptr;
                }
                fprintf(stderr, "nohuman: WARN gzip output %s: the GPU encoder could not be set up (%s); encoding on the host\n",
                        name ? name : "", why.c_str());
            }
            return new GzipEncoder(fd, threads, name);
        }
        case NH_CODEC_ZSTD:
            if (!zstd_api().ok) {
                set_error(NH_EINVAL, "Zstd output is not available: libzstd.so.1 could not be loaded");
                return nullptr;
            }
            return new ZstdEncoder(fd, threads, name);
        case NH_CODEC_BZIP2:
            if (!bz2_api().ok) {
                set_error(NH_EINVAL, "Bzip2 output is not available: libbz2.so.1 could not be loaded");
                return nullptr;
            }
            return new Bzip2Encoder(fd, name);
        case NH_CODEC_XZ:
            if (!lzma_api().ok) {
                set_error(NH_EINVAL, "Xz output is not available: liblzma.so.5 could not be loaded");
                return nullptr;
            }
            return new XzEncoder(fd, threads, name);
        default:
            set_error(NH_EINVAL, "unknown codec %d", codec);
            return nullptr;
    }
}

int compress_file(const char *in, const char *out, int codec, unsigned threads, int device = -1) {
    if (!in || !out) return set_error(NH_EINVAL, "nh_compress_file: null path");
    if (codec != NH_CODEC_NONE && codec != NH_CODEC_GZIP && codec != NH_CODEC_BZIP2 && codec != NH_CODEC_XZ &&
        codec != NH_CODEC_ZSTD)
        return set_error(NH_EINVAL, "nh_compress_file: unknown codec %d", codec);
    int fin = ::open(in, O_RDONLY | O_CLOEXEC);
    if (fin < 0) return set_error(NH_EIO, "cannot open %s", in);
    int fout = ::open(out, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fout < 0) {
        ::close(fin);
        return set_error(NH_EIO, "cannot create %s", out);
    }
    int rc = NH_OK;
    {
        std::unique_ptr<StreamEncoder> enc(make_encoder(codec, fout, threads, out, device));
        if (!enc) {
            rc = NH_EINVAL;  // message set by make_encoder
        } else {
            std::vector<char> buf(4u << 20);
            for (;;) {
                const long n = read_full(fin, buf.data(), buf.size());
                if (n < 0) rc = set_error(NH_EIO, "read error on %s", in);
                if (n <= 0) break;
                if ((rc = enc->write(buf.data(), (size_t)n))) break;
            }
            if (rc == NH_OK) rc = enc->finish();
        }
    }
    if (::close(fout) != 0 && rc == NH_OK) rc = set_error(NH_EIO, "write error on %s", out);
    ::close(fin);
    return rc;
}

}  // namespace nh

extern "C" int nh_gunzip_file(const char *in, const char *out, uint32_t threads, uint64_t chunk_bytes,
                              uint64_t *stats3) {
    if (!in || !out) return nh::set_error(NH_EINVAL, "nh_gunzip_file: null path");
    nh::ParallelGunzip gz;
    std::string err;
    if (gz.open(in, threads, (size_t)chunk_bytes, err) != 0) return nh::set_error(NH_EIO, "%s", err.c_str());
    int fout = ::open(out, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fout < 0) return nh::set_error(NH_EIO, "cannot create %s", out);
    std::vector<uint8_t> buf(8u << 20);
    int rc = NH_OK;
    for (;;) {
        const long n = gz.read(buf.data(), buf.size());
        if (n < 0) {
            rc = nh::set_error(NH_EIO, "%s", gz.error().c_str());
            break;
        }
        if (n == 0) break;
        if (!nh::write_all(fout, buf.data(), (size_t)n)) {
            rc = nh::set_error(NH_EIO, "write error on %s", out);
            break;
        }
    }
    if (::close(fout) != 0 && rc == NH_OK) rc = nh::set_error(NH_EIO, "write error on %s", out);
    if (stats3) gz.stats(&stats3[0], &stats3[1], &stats3[2]);
    return rc;
}

// Test hook (not part of the ABI in include/nohuman_engine.h, like nh_debug_sched): the file decoded as a CHAIN OF RANGES, the way
// the hybrid reader of nh_gunzip.hip uses RangeGunzip -- cell k = the bytes [k, k + 1) * cell_bytes of the file, each by a fresh
// RangeGunzip whose chunks are all decoded before the stream's position and window at the cell are handed to it; every
// `host_every`-th cell that way, the cells between by the sequential decoder (standing in for the GPU's pieces, which end at the
// first block boundary behind their cell too).  The members' CRC-32 / ISIZE are checked here from the stretches the ranges report.
// stats4: {cells by RangeGunzip, chunks accepted, chunks rejected, bytes decoded in order}.  Needs no GPU.
extern "C" int nh_debug_gunzip_ranges(const char *in, const char *out, uint32_t threads, uint64_t cell_bytes, uint64_t chunk_bytes,
                                      uint32_t host_every, uint64_t *stats4) {
    if (!in || !out || !cell_bytes) return nh::set_error(NH_EINVAL, "nh_debug_gunzip_ranges: bad argument");
    int fd = ::open(in, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return nh::set_error(NH_EIO, "cannot open %s", in);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 18) {
        ::close(fd);
        return nh::set_error(NH_EIO, "not a regular gzip file: %s", in);
    }
    const size_t size = (size_t)st.st_size;
    const uint8_t *base = (const uint8_t *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == (const uint8_t *)MAP_FAILED) return nh::set_error(NH_EIO, "cannot map %s", in);
    int rc = NH_OK;
    int fout = ::open(out, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fout < 0) rc = nh::set_error(NH_EIO, "cannot create %s", out);
    bool trunc = false;
    const uint8_t *body = rc ? nullptr : nh::gzip_member_body(base, base + size, &trunc);
    if (!rc && !body) rc = nh::set_error(NH_EIO, "not in gzip format: %s", in);
    uint64_t P = body ? (uint64_t)(body - base) * 8 : 0;
    std::vector<uint8_t> window(32768, 0), wafter(32768);
    uint32_t run_crc = 0;
    uint64_t run_len = 0, n_host = 0, acc = 0, rej = 0, gap = 0;
    bool ended = false;
    if (!host_every) host_every = 1;
    auto member = [&](uint32_t crc, uint64_t len, bool end, uint32_t want_crc, uint32_t want_isize) {
        run_crc = nh::crc32_join(run_crc, crc, len);
        run_len += len;
        if (!end) return true;
        const bool ok = run_crc == want_crc && (uint32_t)run_len == want_isize;
        run_crc = 0;
        run_len = 0;
        return ok;
    };
    for (uint64_t turn = 0; !rc && !ended; turn++) {
        const uint64_t cell = (P >> 3) / cell_bytes, lo = cell * cell_bytes, hi = lo + cell_bytes;
        uint64_t end_bit = 0;
        if (turn % host_every == 0) {
            nh::RangeGunzip rg;
            rg.start(base, size, lo, hi, threads, (size_t)chunk_bytes);
            rg.wait_speculated();
            // (deflate expands 1032 : 1 at most; the buffer is not touched beyond what is written)
            const size_t cap = (size_t)std::min<uint64_t>((std::min<uint64_t>(hi, size) - lo + 65536) * 1040, (uint64_t)1 << 33);
            std::unique_ptr<uint8_t[]> text(new uint8_t[cap]);
            std::vector<nh::GzSeg> segs;
            const long n = rg.finish(P, window.data(), text.get(), cap, &end_bit, &ended, wafter.data(), segs);
            if (n < 0) {
                rc = nh::set_error(NH_EIO, "%s", rg.error().c_str());
                break;
            }
            uint64_t a = 0, r = 0, g = 0;
            rg.stats(&a, &r, &g);
            acc += a, rej += r, gap += g, n_host++;
            uint64_t sum = 0;
            for (const nh::GzSeg &sg : segs) {
                if (!member(sg.crc, sg.len, sg.member_end, sg.want_crc, sg.want_isize)) rc = nh::set_error(NH_EIO, "gzip: crc error");
                sum += sg.len;
            }
            if (!rc && sum != (uint64_t)n) rc = nh::set_error(NH_EIO, "the stretches do not add up to the text");
            if (!rc && !nh::write_all(fout, text.get(), (size_t)n)) rc = nh::set_error(NH_EIO, "write error on %s", out);
            window.swap(wafter);
        } else {  // (a piece of the other kind: in order, to the first block boundary behind the cell)
            std::vector<uint8_t> o;
            std::vector<nh::GzMemberEnd> members;
            std::string err;
            if (nh::inflate_from(base, base + size, P, hi * 8, window.data(), window.size(), o, members, &end_bit, &ended, err) != 0) {
                rc = nh::set_error(NH_EIO, "%s", err.c_str());
                break;
            }
            uint64_t a = 0;
            for (const nh::GzMemberEnd &m : members) {
                if (!member(nh::crc32_fast(0, o.data() + a, (size_t)(m.out_pos - a)), m.out_pos - a, true, m.crc, m.isize))
                    rc = nh::set_error(NH_EIO, "gzip: crc error");
                a = m.out_pos;
            }
            if (o.size() > a) member(nh::crc32_fast(0, o.data() + a, o.size() - (size_t)a), o.size() - a, false, 0, 0);
            if (!rc && !nh::write_all(fout, o.data(), o.size())) rc = nh::set_error(NH_EIO, "write error on %s", out);
            if (o.size() >= window.size()) memcpy(window.data(), o.data() + o.size() - window.size(), window.size());
            else if (!o.empty()) {
                memmove(window.data(), window.data() + o.size(), window.size() - o.size());
                memcpy(window.data() + window.size() - o.size(), o.data(), o.size());
            }
        }
        if (!rc && !ended && end_bit <= P) rc = nh::set_error(NH_EIO, "the stream did not advance");
        P = end_bit;
    }
    if (!rc && run_len) rc = nh::set_error(NH_EIO, "gzip: unexpected end of file");
    if (fout >= 0 && ::close(fout) != 0 && rc == NH_OK) rc = nh::set_error(NH_EIO, "write error on %s", out);
    munmap((void *)base, size);
    if (stats4) stats4[0] = n_host, stats4[1] = acc, stats4[2] = rej, stats4[3] = gap;
    return rc;
}

extern "C" int nh_compress_file(const char *in, const char *out, int codec, uint32_t threads) {
    return nh::compress_file(in, out, codec, threads);
}

extern "C" int nh_compress_file_device(const char *in, const char *out, int codec, uint32_t threads, int32_t device) {
    if (device < 0) return nh::set_error(NH_EINVAL, "nh_compress_file_device: no device");
    return nh::compress_file(in, out, codec, threads, device);
}
