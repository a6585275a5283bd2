// nh_gunzip.h -- the gzip READER on the GPU (the reference hands .gz inputs straight to the
// path, /root/reference/src/main.rs:267, where kraken2's wrapper pipes them through `gzip -dc`).
//
// The compressed bytes go to the device as they are read; there
//   k_search3   finds, per stretch of the file, the first bit position that parses as a non-final dynamic-Huffman block
//               header (the seam test of nh_inflate.cpp plus a trial walk of the block's first tokens),
//   k_inflate3  decodes from every start found to the next one, a wavefront per chunk, to 16-bit symbols -- what a match
//               copies out of the unknown 32 KiB before the chunk is a marker 0x8000 | index --, walking over member
//               trailers and headers,
//   k_scan_local / k_scan_groups / k_scan_windows   give every chunk its window by a prefix scan over the chunks' index maps,
//   k_resolve   turns the symbols into the text, at its place in the caller's device buffer,
//   k_crc       computes the CRC-32 of every chunk's pieces (the host joins them per member and checks CRC and ISIZE
//               like gzip does).
// A false positive of the search (the chunk before does not end where the next one starts) is re-decoded from the true
// end; what the device cannot decode (a block longer than the look-ahead, text beyond 16 : 1 per chunk, more than four
// members ending in one chunk, damage) is decoded by the host decoder (nh_inflate.cpp: inflate_from) from the same bit
// position and window -- loudly: one line on stderr per file and a count in the statistics; damaged data fails there
// with the host reader's messages.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace nh {

struct DevGunzipStats {
    uint64_t segments = 0, chunks = 0, redecoded = 0;  // segments decoded, chunks in them, chunks decoded again (broken seam)
    uint64_t fallback_segments = 0;                    // segments the host decoder took over
    uint64_t members = 0;
    uint64_t text_bytes = 0, gzip_bytes = 0;
    double ms_search = 0, ms_decode = 0, ms_scan = 0, ms_resolve = 0, ms_crc = 0;  // HIP events, NOHUMAN_TRACE / stats only
    double s_host_copy = 0, s_wait = 0;
};

class DevGunzipImpl;

// One gzip file (one or many members) decoded on a GPU, piece by piece, into device memory.
class DevGunzip {
public:
    DevGunzip();
    ~DevGunzip();
    DevGunzip(const DevGunzip &) = delete;
    DevGunzip &operator=(const DevGunzip &) = delete;
    // seg_bytes / stretch_bytes: compressed bytes per piece / per chunk (0 = defaults; test knobs NOHUMAN_GZDEV_SEG,
    // NOHUMAN_GZDEV_STRETCH).  -1 with err set when the file is no regular gzip file or the buffers cannot be had.
    int open(const char *path, int device, size_t seg_bytes, size_t stretch_bytes, std::string &err);
    // Decodes the next piece of the stream to d_dst (device memory, `room` bytes, 16-byte aligned) on `stream` and
    // waits for it: the number of text bytes (0: the stream has ended), or -1 with error() set.  A piece that would
    // not fit `room` is cut down; `room` must hold at least 64 MiB.
    long next(void *d_dst, size_t room, hipStream_t stream);
    // Pieces spread over several devices: add_device() gives the reader a buffer set on one more device (the index of the
    // set, the first device being 0; -1 with err set); next_on(set, ...) decodes the next piece THERE (d_dst and stream on
    // that device) -- the stream's state goes from piece to piece, the 32 KiB window by a device-to-device copy.
    int add_device(int device, std::string &err);
    long next_on(int set, void *d_dst, size_t room, hipStream_t stream);
    uint64_t pieces_of(int set) const;
    // Pieces decoded AHEAD of the stream (what makes the reader scale with the devices): the file is a grid of cells of one
    // piece's size; decode_ahead(set, cell) runs the first phase of a cell's piece -- search, decode to symbols, chain check:
    // everything that needs neither the window nor the stream's state -- on that set, from any thread, while the stream is
    // elsewhere.  take(set, use_ahead) is the stream's next piece on that set, in stream order: the decoded cell if it starts
    // exactly where the stream stands (windows, text, CRCs are done now), else a piece decoded in order from the stream's
    // position to the end of its cell.  ahead_ok(): the reader's shape allows it.
    bool ahead_ok() const;
    uint64_t cell_of_position() const;
    uint64_t cells() const;
    void decode_ahead(int set, uint64_t cell, hipStream_t stream);
    long take(int set, bool use_ahead, void *d_dst, size_t room, hipStream_t stream);
    // The hybrid reader's host lane (round 6): cells of the grid inflated by the host's cores (set_host_threads(n > 0) enables it)
    // while the GPU decodes others -- decode_ahead_host(cell) from any thread, take_host() in stream order like take(); the
    // piece's text ends up in device memory like any other's.  host_ok(room): the reader's shape and the text a cell is
    // expected to hold allow it.
    void set_host_threads(unsigned n);
    bool host_ok(size_t room) const;
    void decode_ahead_host(uint64_t cell);
    long take_host(bool use_ahead, void *d_dst, size_t room, hipStream_t stream);
    uint64_t host_pieces() const;
    bool ended() const;
    const std::string &error() const;
    // error() is a failed END-TO-END check of the decode (a member's CRC-32 or ISIZE, the stream's end), not a resource or a
    // "cannot go on here" condition: text handed out before it may be wrong
    bool integrity_failure() const;
    const DevGunzipStats &stats() const;
    void close();

private:
    DevGunzipImpl *impl_;
};

// The whole reader of a gzip FASTQ file on the GPU: DevGunzip's text stays in HBM, a newline scan and a record kernel
// index it there (kraken2's record semantics, SURVEY.md A.6: header verbatim with trailing whitespace stripped, id up to
// the first blank, '+' line dropped, a record the file ends inside is dropped, an empty header line ends the input), and
// what the host gets is the record table -- the batch's text is copied from HBM to HBM where the classifier reads it,
// and only kept records of outputs that are written by the host ever cross PCIe.
struct HalfBatch;
class DevFastqImpl;
class DevFastqReader {
public:
    DevFastqReader();
    ~DevFastqReader();
    DevFastqReader(const DevFastqReader &) = delete;
    DevFastqReader &operator=(const DevFastqReader &) = delete;
    // 0 ok; -1 error (err); 1: not a file this reader takes (no regular gzip file): use BlockReader.
    // devices: the GPUs of the run, this file's first one first -- piece i of the stream is inflated and indexed on
    // devices[i mod n], and its batches say so (HalfBatch::dev_device: nh_run classifies them where they were born).
    // host_threads > 0: the HYBRID reader -- one more lane whose cells of the stream are inflated by that many workers on the
    // host's cores while the devices decode the others (nh_run asks for it where the GPU's codec kernels are the run's
    // bottleneck: gzip outputs encoded on the GPU); the text of every piece ends up on a device either way.
    int open(const char *path, const int *devices, int n_devices, std::string &err, unsigned host_threads = 0);
    int open(const char *path, int device, std::string &err) { return open(path, &device, 1, err, 0); }
    // The next batch.  max_text == 0 (paired inputs: both files' readers must cut at the same records): exactly max_recs
    // records, fewer only at the end of the input (hb.eof).  max_text > 0 (single-end): up to max_recs records and cut after
    // the record that reaches max_text bytes (BlockReader::next_batch's rule), and a piece hands out every complete record
    // it holds -- so a file of short reads followed by long ones never has to carry more than one record from piece to
    // piece.  The same max_recs / max_text in every call.
    // 0 ok (hb.error set on malformed input -- kraken2's message).
    // 1: this reader hands the file over to BlockReader: either the text is no four-line FASTQ (FASTA, wrapped sequences;
    //    nothing handed out yet) or the device reader could not go on (a batch larger than the room it keeps in front of a
    //    piece, buffers that cannot be had, a stream its decoder gives up on: handover_reason()).  records_handed() records
    //    have been handed out -- whole batches in paired mode --; BlockReader skips that many and reads on (reader_main, nh_run.hip).
    int next_batch(HalfBatch &hb, size_t max_recs, size_t max_text = 0);
    uint64_t records_handed() const;
    const std::string &handover_reason() const;  // empty: "not ours" (no FASTQ)
    void close();  // waits until every batch handed out has been released

private:
    DevFastqImpl *impl_;
};

// Is the file one the device reader takes (a regular gzip file; BGZF as well: its chunks' starts come from the members' headers)?
bool dev_gunzip_wants(const char *path);

}  // namespace nh
