// nh_inflate.h -- gzip input decoder of the host pipeline ("FASTQ/FASTA reader
// + gzip-overlapped input"; kraken2 gets gzip inputs through `gzip -dc`, here the bytes come from an
// in-process decoder that uses several cores on ONE gzip stream).
//
// Why: the classify kernel takes ~760 M reads/s, zlib inflates ~3 M reads/s of FASTQ per file.  A
// deflate stream has no index, so it cannot simply be cut into pieces; the decoder below cuts the
// compressed file into chunks anyway and lets each worker
//   1. search its chunk for the first position that parses as a non-final dynamic-Huffman block
//      header (all the consistency rules of RFC 1951 3.2.7 hold there by chance about once in 10^7+
//      bit positions),
//   2. decode from there without knowing the 32 KiB of history before it: output symbols are 16 bits
//      wide, a back-reference into the unknown history yields a marker 0x8000|window_index; once 32 KiB
//      of output are free of markers the worker drops to the ordinary byte decoder,
//   3. stop at the first block boundary in the next chunk that passes the same header test.
// The consumer stitches the chunks in order: a chunk is accepted only if it starts exactly where the
// accepted stream ended (so a false positive of the search can cost time, never correctness), its
// markers are replaced from the now known window, and each gzip member's CRC-32 and length are checked
// like gzip does.  Gaps (stretches with no dynamic block header, rejected chunks) are decoded in order
// by the consumer.  The technique is the one published for pugz / rapidgzip; the code is written
// from the deflate specification.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace nh {

class GunzipImpl;

// Decompresses a gzip file (one or many members) from a memory-mapped regular file.
class ParallelGunzip {
public:
    ParallelGunzip();
    ~ParallelGunzip();
    ParallelGunzip(const ParallelGunzip &) = delete;
    ParallelGunzip &operator=(const ParallelGunzip &) = delete;
    // threads = worker threads (>= 1); chunk_bytes = compressed bytes per chunk (0 = default)
    int open(const char *path, unsigned threads, size_t chunk_bytes, std::string &err);
    // fills up to cap bytes; returns the number of bytes, 0 at the end, -1 on error (see error())
    long read(uint8_t *dst, size_t cap);
    const std::string &error() const;
    void close();
    // statistics for tests / tracing: chunks decoded speculatively and accepted, chunks rejected or
    // without a block start, bytes the consumer had to decode itself
    void stats(uint64_t *accepted, uint64_t *rejected, uint64_t *gap_bytes) const;

private:
    GunzipImpl *impl_;
};

// CRC-32 (gzip polynomial), slicing-by-16; crc = 0 to start
uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n);

}  // namespace nh
