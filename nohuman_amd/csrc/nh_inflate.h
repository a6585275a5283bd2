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
#include <vector>

namespace nh {

// ---- pieces of the host decoder that the device reader (nh_gunzip.hip) falls back on ------------------------------
struct GzMemberEnd {
    uint64_t out_pos;  // bytes of output before the end of the member
    uint32_t crc, isize;
};
// First byte of the deflate data of the gzip member whose header starts at p: nullptr when [p, end) does not start
// with a gzip member header (gzip ignores such trailing bytes), *truncated set when the header runs past `end`.
const uint8_t *gzip_member_body(const uint8_t *p, const uint8_t *end, bool *truncated);
// Decodes the gzip file image [base, end) from bit position from_bit -- a block boundary -- with the up to 32 KiB of
// text before it (window_len 0 at a member start) to the first block boundary at or behind stop_bit, walking over
// member trailers and headers, or to the end of the stream -- or to the first block boundary with out_limit bytes of
// text.  0, or -1 with err set (damaged data: the messages of the host reader).
int inflate_from(const uint8_t *base, const uint8_t *end, uint64_t from_bit, uint64_t stop_bit, const uint8_t *window,
                 size_t window_len, std::vector<uint8_t> &out, std::vector<GzMemberEnd> &members, uint64_t *end_bit,
                 bool *stream_end, std::string &err, size_t out_limit = (size_t)-1);
// crc of A || B from crc(A), crc(B) and the length of B (GF(2) shift; what zlib calls crc32_combine)
uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

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

// A RANGE of a gzip stream on the host's cores, for the reader on the GPU (nh_gunzip.hip, round 6: the hybrid reader -- while
// the GPU's codec kernels are the run's bottleneck the host's cores inflate some of the stream's cells): start() sets `threads`
// workers on the bytes [lo, hi) of a gzip file image, every chunk decoded speculatively at once (16-bit symbols, markers for the
// unknown window); finish() -- called when the stream, decoded elsewhere up to here, has reached the range -- stitches the
// chunks from the stream's position and window, replaces the markers and puts the text into the caller's buffer.  The range
// ends where ParallelGunzip's chunks end: at the first block boundary at or behind `hi` that passes the block-header test.
struct GzSeg {  // a stretch of the range's text that belongs to one gzip member
    uint64_t len;
    uint32_t crc;  // CRC-32 of the stretch
    bool member_end;
    uint32_t want_crc, want_isize;  // the member's trailer (member_end)
};
class RangeGunzip {
public:
    RangeGunzip();
    ~RangeGunzip();
    RangeGunzip(const RangeGunzip &) = delete;
    RangeGunzip &operator=(const RangeGunzip &) = delete;
    int start(const uint8_t *base, size_t size, uint64_t lo_byte, uint64_t hi_byte, unsigned threads, size_t chunk_bytes);
    void wait_speculated();  // every chunk has been tried (needs neither the stream's position nor its window)
    // -1: error(); else the bytes of text written to dst
    long finish(uint64_t from_bit, const uint8_t *window, uint8_t *dst, size_t cap, uint64_t *end_bit, bool *stream_end,
                uint8_t *window_after, std::vector<GzSeg> &segs);
    const std::string &error() const;
    void stats(uint64_t *accepted, uint64_t *rejected, uint64_t *gap_bytes) const;
    void close();

private:
    GunzipImpl *impl_;
};

// CRC-32 (gzip polynomial), slicing-by-16; crc = 0 to start
uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n);

}  // namespace nh
