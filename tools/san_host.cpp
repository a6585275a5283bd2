// san_host.cpp -- sanitizer harness for the host-side input / output code (no GPU): built by tests/test_sanitizers.py with
// -fsanitize=address,undefined and again with -fsanitize=thread from THIS file, tools/san_stubs.cpp (stand-ins for the
// library's own GPU-side classes, which a build without device code cannot hold) and the product sources nh_inflate.cpp,
// nh_fastx.cpp, nh_codec.cpp as they are.
//   san_host gunzip file.gz threads chunk_bytes   ParallelGunzip against zlib byte for byte, then the block reader over the file
//   san_host gzip file threads                    the host gzip encoder (make_encoder, the pool of block workers) on the file's
//                                                 bytes, written in odd-sized pieces; zlib inflates the result back to the input
//   san_host ranges file.gz threads cell chunk every   RangeGunzip (the hybrid reader's host lane, round 6): the file as a chain of cells,
//                                                 every `every`-th by a fresh RangeGunzip decoded ahead and stitched, against zlib
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include <memory>

#include "nh_codec.h"
#include "nh_fastx.h"
#include "nh_inflate.h"
#include "nohuman_engine.h"

static int gzip_mode(const char *path, unsigned threads) {
    std::vector<uint8_t> data;
    {
        FILE *f = fopen(path, "rb");
        if (!f) return 2;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
        fclose(f);
    }
    const std::string out = std::string(path) + ".san.gz";
    const int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return 2;
    std::unique_ptr<nh::StreamEncoder> enc(nh::make_encoder(NH_CODEC_GZIP, fd, threads, out.c_str(), -1));
    if (!enc) {
        printf("no encoder: %s\n", nh_last_error());
        return 1;
    }
    size_t pos = 0, step = 1;
    int rc = 0;
    while (pos < data.size() && rc == 0) {  // pieces of growing odd sizes: block boundaries fall everywhere
        const size_t n = std::min(data.size() - pos, step);
        rc = enc->write(data.data() + pos, n);
        pos += n;
        step = step * 3 + 7;
        if (step > (5u << 20)) step = 1;
    }
    if (rc == 0) rc = enc->settle();
    if (rc == 0) rc = enc->finish();
    enc.reset();
    close(fd);
    if (rc != 0) {
        printf("encoder failed: %s\n", nh_last_error());
        return 1;
    }
    std::vector<uint8_t> back;
    gzFile g = gzopen(out.c_str(), "rb");
    if (!g) return 1;
    std::vector<uint8_t> buf(1 << 20);
    for (;;) {
        const int n = gzread(g, buf.data(), (unsigned)buf.size());
        if (n < 0) {
            gzclose(g);
            printf("zlib cannot read the encoder's output\n");
            return 1;
        }
        if (n == 0) break;
        back.insert(back.end(), buf.begin(), buf.begin() + n);
    }
    gzclose(g);
    unlink(out.c_str());
    if (back != data) {
        printf("MISMATCH: %zu bytes in, %zu back\n", data.size(), back.size());
        return 1;
    }
    printf("%zu bytes encoded on %u threads and inflated back\n", data.size(), threads);
    return 0;
}

extern "C" int nh_debug_gunzip_ranges(const char *in, const char *out, uint32_t threads, uint64_t cell_bytes, uint64_t chunk_bytes,
                                      uint32_t host_every, uint64_t *stats4);

static int ranges_mode(const char *path, unsigned threads, uint64_t cell, uint64_t chunk, unsigned every) {
    std::vector<uint8_t> ref;
    bool ref_ok = true;
    {
        gzFile g = gzopen(path, "rb");
        if (!g) return 2;
        std::vector<uint8_t> buf(1 << 20);
        for (;;) {
            int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n < 0) { ref_ok = false; break; }
            if (n == 0) break;
            ref.insert(ref.end(), buf.begin(), buf.begin() + n);
        }
        int err = 0;
        gzerror(g, &err);
        if (err != Z_OK && err != Z_STREAM_END) ref_ok = false;
        gzclose(g);
    }
    const std::string out = std::string(path) + ".ranges.out";
    uint64_t st[4] = {0, 0, 0, 0};
    const int rc = nh_debug_gunzip_ranges(path, out.c_str(), threads, cell, chunk, every, st);
    std::vector<uint8_t> got;
    if (rc == 0) {
        FILE *f = fopen(out.c_str(), "rb");
        if (!f) return 1;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) got.insert(got.end(), buf, buf + n);
        fclose(f);
    }
    unlink(out.c_str());
    if (rc == 0 && ref_ok && got != ref) {
        printf("MISMATCH: %zu vs %zu bytes\n", got.size(), ref.size());
        return 1;
    }
    printf("%zu bytes, ranges %s, zlib %s, %llu cells by RangeGunzip, %llu chunks accepted\n", got.size(), rc == 0 ? "ok" : nh_last_error(),
           ref_ok ? "ok" : "error", (unsigned long long)st[0], (unsigned long long)st[1]);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && !strcmp(argv[1], "gzip")) return gzip_mode(argv[2], (unsigned)atoi(argv[3]));
    if (argc >= 7 && !strcmp(argv[1], "ranges"))
        return ranges_mode(argv[2], (unsigned)atoi(argv[3]), (uint64_t)atoll(argv[4]), (uint64_t)atoll(argv[5]), (unsigned)atoi(argv[6]));
    if (argc < 5 || strcmp(argv[1], "gunzip")) return 2;
    const char *path = argv[2];
    const unsigned threads = (unsigned)atoi(argv[3]);
    const size_t chunk = (size_t)atol(argv[4]);
    std::vector<uint8_t> ref;
    bool ref_ok = true;
    {
        gzFile g = gzopen(path, "rb");
        if (!g) return 2;
        std::vector<uint8_t> buf(1 << 20);
        for (;;) {
            int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n < 0) { ref_ok = false; break; }
            if (n == 0) break;
            ref.insert(ref.end(), buf.begin(), buf.begin() + n);
        }
        int err = 0;
        gzerror(g, &err);
        if (err != Z_OK && err != Z_STREAM_END) ref_ok = false;
        gzclose(g);
    }
    nh::ParallelGunzip pg;
    std::string err;
    if (pg.open(path, threads, chunk, err) != 0) {
        printf("open failed: %s (zlib %s)\n", err.c_str(), ref_ok ? "ok" : "failed");
        return 0;
    }
    std::vector<uint8_t> got, buf(777777);
    bool ok = true;
    for (;;) {
        long n = pg.read(buf.data(), buf.size());
        if (n < 0) { ok = false; break; }
        if (n == 0) break;
        got.insert(got.end(), buf.begin(), buf.begin() + n);
    }
    if (ok && ref_ok && got != ref) {
        printf("MISMATCH: %zu vs %zu bytes\n", got.size(), ref.size());
        return 1;
    }
    if (ok && !ref_ok) printf("note: zlib reports an error, the decoder did not\n");
    // the block reader over the same file (any content: errors are fine, crashes are not)
    nh::BlockReader r;
    size_t recs = 0;
    if (r.open(path, err, threads) == 0) {
        nh::HalfBatch hb;
        for (;;) {
            r.next_batch(hb, 1000, 1u << 20);
            recs += hb.recs.size();
            if (!hb.error.empty() || hb.eof) break;
        }
    }
    printf("%zu bytes, decoder %s, zlib %s, %zu records\n", got.size(), ok ? "ok" : pg.error().c_str(),
           ref_ok ? "ok" : "error", recs);
    return 0;
}
