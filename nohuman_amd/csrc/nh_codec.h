// nh_codec.h -- streaming output encoders: the writer of nh_run feeds the kept
// records straight into one of these, so that the compress stage of the reference
// (/root/reference/src/compression.rs:182-268, called at src/main.rs:342-368) needs no temporary
// uncompressed file.
#pragma once
#include <stddef.h>

namespace nh {

class StreamEncoder {
public:
    virtual ~StreamEncoder() {}
    // appends n bytes to the stream; NH_OK or an error (set_error)
    virtual int write(const void *p, size_t n) = 0;
    // flushes everything and writes the container's trailer; the file descriptor stays open
    virtual int finish() = 0;
    // Optional: the host bytes [host, host + len) also lie at `dev` in the memory of GPU `device` until the next
    // settle() returns -- an encoder that works on that GPU may take spans of them from there.
    // host_valid false: the bytes exist ONLY at `dev` (a batch of the reader on the GPU); spans of such a range are never
    // read from the host.  Only encoders that answer takes_device_spans() may be handed such ranges.
    virtual void map_device(const void *host, size_t len, const void *dev, int device, bool host_valid = true) {
        (void)host, (void)len, (void)dev, (void)device, (void)host_valid;
    }
    virtual bool takes_device_spans() const { return false; }
    // every byte handed to write() so far has been taken over: the caller's memory (host and mapped) is free again
    virtual int settle() { return 0; }
};

// codec: nh_codec of the C ABI.  The encoder writes to fd (not closed by it); `name` only labels errors.
// Returns nullptr with the error set when the codec's library cannot be loaded.
// device >= 0: gzip is encoded on that GPU (nh_deflate.hip) unless NOHUMAN_GZIP=host asks for the host encoder.
StreamEncoder *make_encoder(int codec, int fd, unsigned threads, const char *name, int device = -1);
// the GPU gzip encoder itself (nullptr with the error set when its buffers cannot be had)
StreamEncoder *make_gpu_gzip_encoder(int fd, int device, const char *name);

}  // namespace nh
