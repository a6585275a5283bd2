// asan_host.cpp -- sanitizer harness for the host-side input code (no GPU, no HIP): decodes a gzip
// file with ParallelGunzip and compares every byte with zlib, then runs the block reader over it.
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -Inohuman_amd/csrc tools/asan_host.cpp \
//       nohuman_amd/csrc/nh_inflate.cpp nohuman_amd/csrc/nh_fastx.cpp -lz -lpthread -o /tmp/asan_host
//   /tmp/asan_host file.gz threads chunk_bytes
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#include "nh_fastx.h"
#include "nh_inflate.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const char *path = argv[1];
    const unsigned threads = (unsigned)atoi(argv[2]);
    const size_t chunk = (size_t)atol(argv[3]);
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
