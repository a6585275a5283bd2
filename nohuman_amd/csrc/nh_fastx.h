// nh_fastx.h -- FASTQ/FASTA record reader with kraken2's record semantics (SURVEY.md A.6;
// kraken2 seqreader.cc BatchSequenceReader, external to /root/reference, reached through the
// input paths nohuman passes verbatim at /root/reference/src/main.rs:267).
#pragma once
#include <stdint.h>
#include <stdio.h>

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

// Byte source: plain, gzip (zlib) or bzip2 (through `bzip2 -dc`, as kraken2's wrapper does).
class ByteSource {
public:
    ~ByteSource();
    int open(const char *path, std::string &err);
    // fills up to cap bytes; returns bytes read, 0 at EOF, -1 on error
    long read(uint8_t *buf, size_t cap);
    void close();

private:
    void *gz_ = nullptr;   // gzFile
    FILE *pipe_ = nullptr; // bzip2 -dc
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

// ---- block reader: whole batches of records as slices of one text buffer ------------------------
struct RecRef {  // offsets into HalfBatch::text
    uint32_t h, hlen;  // header line (with '@' / '>'), trailing whitespace stripped
    uint32_t idlen;    // id = text[h+1 .. h+1+idlen)
    uint32_t s, slen;  // sequence
    uint32_t q, qlen;  // qualities (FASTQ)
};

struct HalfBatch {  // the records one input file contributes to a batch
    std::vector<char> text;
    std::vector<RecRef> recs;
    SeqFormat format = FMT_AUTO;
    bool eof = false;
    std::string error;  // non-empty: malformed input
};

// Same record semantics as FastxReader (kraken2 seqreader.cc), but parses straight
// out of the inflate buffer into one contiguous text block per batch -- no per-record allocation.
class BlockReader {
public:
    int open(const char *path, std::string &err);
    // appends up to max_recs records (or max_text bytes) to hb; sets hb.eof at end of input
    void next_batch(HalfBatch &hb, size_t max_recs, size_t max_text);
    void close() { src_.close(); }

private:
    bool fill();                                   // read more bytes; false at end of input
    bool line(const char *&b, const char *&e);     // next line [b,e) without '\n'; false at EOF
    ByteSource src_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    SeqFormat format_ = FMT_AUTO;
    std::string carry_;  // FASTA: header line already consumed while joining the previous record
    bool have_carry_ = false;
};

}  // namespace nh
