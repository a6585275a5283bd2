// reader_bench.cpp -- host-only timing of the input side of nh_run (no GPU): BlockReader batches
// (inflate + in-place parse) per second, with the share of the parse alone.
//   g++ -O3 -std=c++17 -Inohuman_amd/csrc tools/reader_bench.cpp nohuman_amd/csrc/nh_inflate.cpp \
//       nohuman_amd/csrc/nh_fastx.cpp -lz -lpthread -o /tmp/reader_bench
//   /tmp/reader_bench file.fq[.gz] gz_threads [batch_records=262144]
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>

#include "nh_fastx.h"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const unsigned threads = (unsigned)atoi(argv[2]);
    const size_t batch = argc > 3 ? (size_t)atol(argv[3]) : 262144;
    for (int rep = 0; rep < 3; rep++) {
        nh::BlockReader r;
        std::string err;
        if (r.open(argv[1], err, threads) != 0) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        nh::HalfBatch hb;
        size_t n = 0, bytes = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            r.next_batch(hb, batch, (size_t)-1);
            if (!hb.error.empty()) {
                fprintf(stderr, "%s\n", hb.error.c_str());
                return 1;
            }
            n += hb.recs.size();
            bytes += hb.text.size();
            if (hb.eof) break;
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%zu records, %.1f MB text in %.3f s = %.0f MB/s, %.2f Mrecords/s (gz threads %u)\n", n, bytes / 1e6, dt,
               bytes / 1e6 / dt, n / 1e6 / dt, threads);
    }
    return 0;
}
