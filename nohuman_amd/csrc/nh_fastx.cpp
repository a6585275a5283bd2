// nh_fastx.cpp -- see nh_fastx.h.  Record semantics follow SURVEY.md A.6 (kraken2 seqreader.cc).
#include "nh_fastx.h"

#include <ctype.h>
#include <string.h>
#include <zlib.h>

namespace nh {

ByteSource::~ByteSource() { close(); }

int ByteSource::open(const char *path, std::string &err) {
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
        // kraken2's wrapper pipes bzip2 inputs through `bzip2 -dc`
        std::string cmd = "bzip2 -dc '";
        for (const char *p = path; *p; p++) {
            if (*p == '\'')
                cmd += "'\\''";
            else
                cmd += *p;
        }
        cmd += "'";
        pipe_ = popen(cmd.c_str(), "r");
        if (!pipe_) {
            err = std::string("cannot run bzip2 -dc on ") + path;
            return -1;
        }
        return 0;
    }
    gzFile g = gzopen(path, "rb");  // transparent for plain files
    if (!g) {
        err = std::string("cannot open ") + path;
        return -1;
    }
    gzbuffer(g, 1u << 20);
    gz_ = g;
    return 0;
}

long ByteSource::read(uint8_t *buf, size_t cap) {
    if (gz_) {
        int n = gzread((gzFile)gz_, buf, (unsigned)cap);
        return n;
    }
    if (pipe_) {
        size_t n = fread(buf, 1, cap, pipe_);
        if (n == 0 && ferror(pipe_)) return -1;
        return (long)n;
    }
    return -1;
}

void ByteSource::close() {
    if (gz_) gzclose((gzFile)gz_);
    if (pipe_) pclose(pipe_);
    gz_ = nullptr;
    pipe_ = nullptr;
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
int BlockReader::open(const char *path, std::string &err) {
    buf_.resize(8u << 20);
    pos_ = len_ = 0;
    eof_ = false;
    format_ = FMT_AUTO;
    have_carry_ = false;
    return src_.open(path, err);
}

bool BlockReader::fill() {
    if (eof_) return false;
    if (pos_ > 0) {  // keep the unread tail at the front
        memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
        len_ -= pos_;
        pos_ = 0;
    }
    if (len_ == buf_.size()) buf_.resize(buf_.size() * 2);  // a line longer than the buffer
    long n = src_.read((uint8_t *)buf_.data() + len_, buf_.size() - len_);
    if (n <= 0) {
        eof_ = true;
        return false;
    }
    len_ += (size_t)n;
    return true;
}

// std::getline semantics: false only if not a single character could be extracted
bool BlockReader::line(const char *&b, const char *&e) {
    for (;;) {
        const char *p = buf_.data() + pos_;
        const char *nl = (const char *)memchr(p, '\n', len_ - pos_);
        if (nl) {
            b = p;
            e = nl;
            pos_ = (size_t)(nl - buf_.data()) + 1;
            return true;
        }
        if (!fill()) {  // end of input: the rest (if any) is the last line
            if (pos_ == len_) return false;
            b = buf_.data() + pos_;
            e = buf_.data() + len_;
            pos_ = len_;
            return true;
        }
    }
}

static inline const char *rstrip(const char *b, const char *e) {
    while (e > b && isspace((unsigned char)e[-1])) e--;
    return e;
}

void BlockReader::next_batch(HalfBatch &hb, size_t max_recs, size_t max_text) {
    hb.format = format_;
    while (hb.recs.size() < max_recs && hb.text.size() < max_text) {
        const char *b, *e;
        std::string hdr;
        if (have_carry_) {
            hdr.swap(carry_);
            have_carry_ = false;
        } else {
            if (!line(b, e)) {
                hb.eof = true;
                return;
            }
            hdr.assign(b, rstrip(b, e));
        }
        if (format_ == FMT_AUTO) {
            if (!hdr.empty() && hdr[0] == '@')
                format_ = FMT_FASTQ;
            else if (!hdr.empty() && hdr[0] == '>')
                format_ = FMT_FASTA;
            else {
                hb.error = "sequence reader - unrecognized file format";
                return;
            }
            hb.format = format_;
        }
        if (format_ == FMT_FASTQ) {
            if (hdr.empty()) {  // an empty line may end the file
                hb.eof = true;
                return;
            }
            if (hdr[0] != '@') {
                hb.error = "malformed FASTQ file (exp. '@', saw \"" + hdr + "\"), aborting";
                return;
            }
        } else if (hdr.empty() || hdr[0] != '>') {
            hb.error = "malformed FASTA file (exp. '>', saw \"" + hdr + "\"), aborting";
            return;
        }
        if (hdr.size() <= 1) {
            hb.eof = true;
            return;
        }
        RecRef r;
        r.h = (uint32_t)hb.text.size();
        r.hlen = (uint32_t)hdr.size();
        size_t ws = hdr.find_first_of(" \t\r", 1);
        r.idlen = (uint32_t)((ws == std::string::npos ? hdr.size() : ws) - 1);
        hb.text.insert(hb.text.end(), hdr.begin(), hdr.end());
        if (format_ == FMT_FASTQ) {
            if (!line(b, e)) {
                hb.text.resize(r.h);
                hb.eof = true;
                return;
            }
            const char *se = rstrip(b, e);
            r.s = (uint32_t)hb.text.size();
            r.slen = (uint32_t)(se - b);
            hb.text.insert(hb.text.end(), b, se);
            const char *pb, *pe;
            if (!line(pb, pe) || !line(b, e)) {  // '+' line (discarded), qualities
                hb.text.resize(r.h);
                hb.eof = true;
                return;
            }
            const char *qe = rstrip(b, e);
            r.q = (uint32_t)hb.text.size();
            r.qlen = (uint32_t)(qe - b);
            hb.text.insert(hb.text.end(), b, qe);
        } else {
            r.s = (uint32_t)hb.text.size();
            for (;;) {  // join sequence lines up to the next header
                if (pos_ == len_ && !fill()) break;
                if (buf_[pos_] == '>') break;
                if (!line(b, e)) break;
                hb.text.insert(hb.text.end(), b, rstrip(b, e));
            }
            r.slen = (uint32_t)(hb.text.size() - r.s);
            r.q = r.s + r.slen;
            r.qlen = 0;
        }
        hb.recs.push_back(r);
    }
}

}  // namespace nh
