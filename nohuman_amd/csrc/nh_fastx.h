// nh_fastx.h -- FASTQ/FASTA record reader with kraken2's record semantics (SURVEY.md A.6;
// kraken2 seqreader.cc BatchSequenceReader, external to /root/reference, reached through the
// input paths nohuman passes verbatim at /root/reference/src/main.rs:267).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

namespace nh {

enum SeqFormat { FMT_AUTO = 0, FMT_FASTA = 1, FMT_FASTQ = 2 };

struct SeqRecord {
    SeqFormat format = FMT_AUTO;
    std::string header;  // whole header line incl. '@' / '>', trailing whitespace stripped
    std::string id;      // text after the first char up to the first space / tab / CR
    std::string seq;
    std::string quals;
};

// Byte source: plain, gzip (own multi-threaded decoder or zlib) or bzip2 (libbz2; kraken2's wrapper pipes
// such inputs through `bzip2 -dc`).
class ParallelGunzip;
class Bz2Source;
class DevGzSource;  // gzip decoded on a GPU (nh_gunzip.h), the text fetched from there

class ByteSource {
public:
    ~ByteSource();
    // gz_threads > 0: gzip files are decoded by the in-process multi-threaded decoder
    // (nh_inflate.h) on that many workers; 0: zlib on the calling thread.
    // device >= 0: gzip files are decoded on that GPU (nh_gunzip.h) unless NOHUMAN_GZ_READER=host
    int open(const char *path, std::string &err, unsigned gz_threads = 0, int device = -1);
    // fills up to cap bytes; returns bytes read, 0 at EOF, -1 on error
    long read(uint8_t *buf, size_t cap);
    void close();
    const std::string &error() const { return pgz_error_; }

private:
    void *gz_ = nullptr;   // gzFile
    Bz2Source *bz_ = nullptr;  // bzip2 (libbz2 by dlopen)
    int fd_ = -1;          // plain text
    ParallelGunzip *pgz_ = nullptr;
    DevGzSource *dgz_ = nullptr;
    std::string pgz_error_;
};

class FastxReader {
public:
    int open(const char *path, std::string &err);
    // 1 = record read, 0 = end of input, -1 = malformed input (err set)
    int next(SeqRecord &rec, std::string &err);
    void close() { src_.close(); }

private:
    bool getline(std::string &line);
    int peek();
    ByteSource src_;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    SeqFormat format_ = FMT_AUTO;
    std::string line_;
};

// ---- block reader: whole batches of records as slices of the raw input text -----------------------
struct RecRef {  // offsets into HalfBatch::text
    uint32_t h, hlen;  // header line (with '@' / '>'), trailing whitespace stripped
    uint32_t idlen;    // id = text[h+1 .. h+1+idlen)
    uint32_t s, slen;  // sequence
    uint32_t q, qlen;  // qualities (FASTQ)
    // != 0: text[h .. raw_end) is byte for byte what the record looks like when written back
    // ("header\nseq\n+\nquals\n" / "header\nseq\n"), so kept records can be written without a copy
    uint32_t raw_end;
};

// Growable byte buffer that never zero-fills (batches are ~100 MB and recycled between reads).  The
// memory can come from a caller-supplied allocator: nh_run hands out page-locked memory, so that the raw
// text of a batch goes to the device by an asynchronous copy straight from here.
class RawBuf {
public:
    typedef void *(*AllocFn)(size_t);
    typedef void (*FreeFn)(void *);
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { release(); }
    // only before the first allocation; alloc returns nullptr on failure
    void set_allocator(AllocFn a, FreeFn f) {
        if (!p_) {
            alloc_ = a;
            free_ = f;
        }
    }
    char *data() { return p_; }
    const char *data() const { return p_; }
    size_t size() const { return len_; }
    size_t capacity() const { return cap_; }
    void set_size(size_t n) { len_ = n; }
    void clear() { len_ = 0; }
    bool reserve(size_t n) {  // false = out of memory
        if (n <= cap_) return true;
        size_t c = cap_ ? cap_ : 4096;
        while (c < n) c += c / 2 + 4096;
        char *q;
        if (alloc_) {
            q = (char *)alloc_(c);
            if (!q) return false;
            if (len_) memcpy(q, p_, len_);
            if (p_) free_(p_);
        } else {
            q = (char *)realloc(p_, c);
            if (!q) return false;
        }
        p_ = q;
        cap_ = c;
        return true;
    }
    bool append(const char *b, size_t n) {
        if (!reserve(len_ + n)) return false;
        if (n) memcpy(p_ + len_, b, n);
        len_ += n;
        return true;
    }

private:
    void release() {
        if (p_) {
            if (alloc_) free_(p_);
            else free(p_);
        }
        p_ = nullptr;
    }
    char *p_ = nullptr;
    size_t len_ = 0, cap_ = 0;
    AllocFn alloc_ = nullptr;
    FreeFn free_ = nullptr;
};

struct HalfBatch {  // the records one input file contributes to a batch
    RawBuf text;
    std::vector<RecRef> recs;
    SeqFormat format = FMT_AUTO;
    bool eof = false;
    std::string error;  // non-empty: malformed input
    // A batch born on a GPU (DevFastqReader, nh_gunzip.h): its text lies at dev_text in the memory of device dev_device,
    // `text` has its size but holds the bytes only once host_text_valid says so (fetched for outputs that need them);
    // release() hands the device memory back to the reader when the batch is done.
    const uint8_t *dev_text = nullptr;
    int dev_device = -1;
    bool host_text_valid = true;
    std::function<void()> release;
    void reset() {
        if (release) {
            release();
            release = nullptr;
        }
        dev_text = nullptr;
        dev_device = -1;
        host_text_valid = true;
        text.clear();
        recs.clear();
        format = FMT_AUTO;
        eof = false;
        error.clear();
    }
    ~HalfBatch() {
        if (release) release();
    }
};

// Same record semantics as FastxReader (kraken2 seqreader.cc).  The input is read
// (inflated) straight into the batch's text buffer and parsed in place: a RecRef is a set of offsets
// into the raw bytes, nothing is copied per record (multi-line FASTA sequences are joined in place).
class ReadAhead;  // helper thread that fills the next stretch of a batch's text while the previous one is parsed

class BlockReader {
public:
    BlockReader() = default;
    BlockReader(const BlockReader &) = delete;
    BlockReader &operator=(const BlockReader &) = delete;
    ~BlockReader();
    int open(const char *path, std::string &err, unsigned gz_threads = 0, int device = -1);
    // parses up to max_recs records (or about max_text bytes) into hb (which is reset first);
    // sets hb.eof at end of input.  Text offsets are 32-bit: max_text is clamped below 2^32.
    void next_batch(HalfBatch &hb, size_t max_recs, size_t max_text);
    void close();

private:
    // one record at text[pos..len): 1 parsed, 0 more input needed, -1 end of input reached
    // (hb.eof set), -2 malformed (hb.error set)
    int parse_one(HalfBatch &hb, size_t &pos, size_t len);
    ByteSource src_;
    RawBuf tail_;       // bytes after the last complete record of the previous batch
    bool eof_ = false;  // the source is drained
    SeqFormat format_ = FMT_AUTO;
    size_t chunk_ = 4u << 20;  // bytes per read of the source
    size_t fa_resume_ = 0;  // FASTA: the scan of the record at `pos` may resume here (long records)
    size_t fa_resume_rec_ = (size_t)-1;
    ReadAhead *ahead_ = nullptr;
};

}  // namespace nh
