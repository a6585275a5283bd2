// nh_codec.cpp -- output compression stage of the nohuman host.
//
// Mirrors CompressionFormat::compress (/root/reference/src/compression.rs:182-200): gzip with
// `threads` workers (gzip_compress, compression.rs:214-233, gzp's block-parallel encoder at the
// default level), bzip2 single-threaded (compression.rs:202-212), xz multi-threaded
// (compression.rs:235-252).  Parity target is the decompressed content and the container magic
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
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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
            j->crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), block, (uInt)j->len);
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

int gzip_parallel(int fin, int fout, unsigned threads, const char *in_name, const char *out_name) {
    if (threads < 1) threads = 1;
    GzShared sh;
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < threads; i++) pool.emplace_back(gz_worker, &sh);
    auto stop = [&] {
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.quit = true;
        }
        sh.work_cv.notify_all();
        for (auto &t : pool) t.join();
    };
    static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};  // no name, no mtime, unix
    int rc = NH_OK;
    if (!write_all(fout, header, sizeof header)) rc = set_error(NH_EIO, "write error on %s", out_name);
    std::deque<std::unique_ptr<GzJob>> inflight;
    std::vector<std::unique_ptr<GzJob>> spare;
    std::vector<unsigned char> dict;
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    uint64_t total = 0;
    const size_t max_inflight = 2 * (size_t)threads + 2;
    bool eof = false;
    auto retire_head = [&]() {  // wait for the oldest block and write it
        GzJob *j = inflight.front().get();
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.done_cv.wait(lk, [&] { return j->done; });
        }
        if (j->failed)
            rc = set_error(NH_EIO, "deflate failed on %s", in_name);
        else if (!write_all(fout, j->out.data(), j->out_len))
            rc = set_error(NH_EIO, "write error on %s", out_name);
        crc = (uint32_t)crc32_combine(crc, j->crc, (z_off_t)j->len);
        total += j->len;
        spare.push_back(std::move(inflight.front()));
        inflight.pop_front();
    };
    while (!eof && rc == NH_OK) {
        std::unique_ptr<GzJob> j;
        if (!spare.empty()) {
            j = std::move(spare.back());
            spare.pop_back();
        } else {
            j.reset(new GzJob());
        }
        j->done = j->failed = false;
        j->dict_len = dict.size();
        j->in.resize(j->dict_len + GZ_BLOCK);
        if (j->dict_len) memcpy(j->in.data(), dict.data(), j->dict_len);
        const long n = read_full(fin, j->in.data() + j->dict_len, GZ_BLOCK);
        if (n < 0) {
            rc = set_error(NH_EIO, "read error on %s", in_name);
            break;
        }
        if (n == 0) break;
        eof = (size_t)n < GZ_BLOCK;
        j->len = (size_t)n;
        const size_t have = j->dict_len + j->len, keep = have < GZ_DICT ? have : GZ_DICT;
        dict.assign(j->in.data() + have - keep, j->in.data() + have);
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.pending.push_back(j.get());
        }
        sh.work_cv.notify_one();
        inflight.push_back(std::move(j));
        while (inflight.size() >= max_inflight && rc == NH_OK) retire_head();
    }
    while (!inflight.empty()) retire_head();  // also on errors: workers still hold pointers
    stop();
    if (rc != NH_OK) return rc;
    unsigned char tail[10] = {0x03, 0x00};  // final block: fixed Huffman, end-of-block only
    for (int i = 0; i < 4; i++) {
        tail[2 + i] = (unsigned char)(crc >> (8 * i));
        tail[6 + i] = (unsigned char)((uint32_t)total >> (8 * i));
    }
    if (!write_all(fout, tail, sizeof tail)) return set_error(NH_EIO, "write error on %s", out_name);
    return NH_OK;
}

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

int zstd_file(int fin, int fout, unsigned threads, const char *in_name, const char *out_name) {
    static const ZstdApi Z;
    if (!Z.ok) return set_error(NH_EINVAL, "Zstd output is not available: libzstd.so.1 could not be loaded");
    enum { C_LEVEL = 100, C_CHECKSUM = 201, C_WORKERS = 400, E_CONTINUE = 0, E_END = 2 };
    void *cctx = Z.createCCtx();
    if (!cctx) return set_error(NH_EOOM, "ZSTD_createCCtx failed");
    (void)Z.setParameter(cctx, C_LEVEL, 3);
    (void)Z.setParameter(cctx, C_CHECKSUM, 1);
    if (threads > 1) (void)Z.setParameter(cctx, C_WORKERS, (int)threads);  // ignored by single-threaded builds
    std::vector<char> ibuf(1u << 20), obuf(1u << 20);
    int rc = NH_OK;
    for (bool last = false; !last && rc == NH_OK;) {
        const long n = read_full(fin, ibuf.data(), ibuf.size());
        if (n < 0) {
            rc = set_error(NH_EIO, "read error on %s", in_name);
            break;
        }
        last = (size_t)n < ibuf.size();
        ZstdApi::InBuf in = {ibuf.data(), (size_t)n, 0};
        for (;;) {
            ZstdApi::OutBuf out = {obuf.data(), obuf.size(), 0};
            const size_t left = Z.compressStream2(cctx, &out, &in, last ? E_END : E_CONTINUE);
            if (Z.isError(left)) {
                rc = set_error(NH_EIO, "zstd: %s", Z.getErrorName(left));
                break;
            }
            if (out.pos && !write_all(fout, obuf.data(), out.pos)) {
                rc = set_error(NH_EIO, "write error on %s", out_name);
                break;
            }
            if (last ? left == 0 : in.pos == in.size) break;
        }
    }
    Z.freeCCtx(cctx);
    return rc;
}

std::string shell_quote(const char *s) {
    std::string q = "'";
    for (; *s; s++) q += *s == '\'' ? std::string("'\\''") : std::string(1, *s);
    return q + "'";
}

}  // namespace

int compress_file(const char *in, const char *out, int codec, unsigned threads) {
    if (!in || !out) return set_error(NH_EINVAL, "nh_compress_file: null path");
    if (codec != NH_CODEC_NONE && codec != NH_CODEC_GZIP && codec != NH_CODEC_BZIP2 && codec != NH_CODEC_XZ &&
        codec != NH_CODEC_ZSTD)
        return set_error(NH_EINVAL, "nh_compress_file: unknown codec %d", codec);
    int fin = ::open(in, O_RDONLY | O_CLOEXEC);
    if (fin < 0) return set_error(NH_EIO, "cannot open %s", in);
    int rc = NH_OK;
    if (codec == NH_CODEC_NONE || codec == NH_CODEC_GZIP || codec == NH_CODEC_ZSTD) {
        int fout = ::open(out, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        if (fout < 0) {
            ::close(fin);
            return set_error(NH_EIO, "cannot create %s", out);
        }
        if (codec == NH_CODEC_GZIP) {
            rc = gzip_parallel(fin, fout, threads, in, out);
        } else if (codec == NH_CODEC_ZSTD) {
            rc = zstd_file(fin, fout, threads, in, out);
        } else {
            std::vector<char> buf(4u << 20);
            for (;;) {
                const long n = read_full(fin, buf.data(), buf.size());
                if (n < 0) rc = set_error(NH_EIO, "read error on %s", in);
                if (n <= 0) break;
                if (!write_all(fout, buf.data(), (size_t)n)) {
                    rc = set_error(NH_EIO, "write error on %s", out);
                    break;
                }
            }
        }
        if (::close(fout) != 0 && rc == NH_OK) rc = set_error(NH_EIO, "write error on %s", out);
    } else {
        // libbz2 / liblzma headers are not in this image; their command-line tools are
        std::string cmd = codec == NH_CODEC_BZIP2 ? std::string("bzip2 -c")
                                                  : "xz -6 -c -T" + std::to_string(threads ? threads : 1);
        cmd += " < " + shell_quote(in) + " > " + shell_quote(out);
        const int st = system(cmd.c_str());
        if (st != 0) rc = set_error(NH_EIO, "the %s compressor failed on %s", codec == NH_CODEC_BZIP2 ? "bzip2" : "xz", out);
    }
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

extern "C" int nh_compress_file(const char *in, const char *out, int codec, uint32_t threads) {
    return nh::compress_file(in, out, codec, threads);
}
