// fqz_stream.hip — compress.Compress / compress.Decompress (internal/compress/compress.go:125-192, 558-604) as a streaming
// host pipeline over the device-resident batch codec.
//
// The reference runs a producer goroutine (parser), W worker goroutines (block codec) and an ordered collector joined by
// channels (compress.go:240-278, 365-403).  Here the block codec is the GPU, and what has to overlap is the host link:
//   feeder thread     source -> pinned buffer -> H2D          (produceCompressJobs   compress.go:303-363)
//   calling thread    assemble the batch, launch + finish the device pipeline      (runCompressionWorker :280-301)
//   drainer thread    D2H -> sink, in batch order              (collectAndWriteResults :365-403)
// over three slots (each with its own child context = own stream and workspaces), so that the H2D of slice k+1, the
// kernels of slice k and the D2H of slice k-1 run at the same time and memory stays bounded by the slots whatever the
// size of the input.  A batch is the unconsumed tail of the previous one (a partial block) plus the new slice; when not
// even one 100 000-record block fits (long reads) the batch simply keeps growing with the next slice.
#include "fqz_ctx.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

int fqz_enc_launch(fqz_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint32_t rpb, int qual_encoding, uint32_t flags, uint8_t *d_out, size_t out_cap,
                   hipStream_t stream);

namespace {

const int NS = 3; // slots in flight

size_t slice_bytes()
{
    const char *e = getenv("FQZ_SLICE_KB"); // (tests use small slices to drive many batches through small inputs)
    const size_t kb = e ? (size_t)strtoull(e, nullptr, 10) : 0;
    return kb ? kb << 10 : (size_t)128 << 20;
}

struct Io { // the two ends of a stream job: either plain memory (copied to / from the device directly) or callbacks
    const uint8_t *mem_in = nullptr; size_t mem_in_n = 0, mem_in_pos = 0;
    fqz_read_fn rd = nullptr; void *rd_user = nullptr;
    uint8_t *mem_out = nullptr; size_t mem_out_cap = 0; bool count_only = false;
    fqz_write_fn wr = nullptr; void *wr_user = nullptr;
    size_t written = 0;
    // reads exactly n bytes unless the source ends: returns bytes read or < 0
    long read_full(uint8_t *dst, size_t n)
    {
        size_t got = 0;
        if (mem_in || !rd) {
            got = mem_in_n - mem_in_pos < n ? mem_in_n - mem_in_pos : n;
            if (got) memcpy(dst, mem_in + mem_in_pos, got);
            mem_in_pos += got;
            return (long)got;
        }
        while (got < n) {
            long r = rd(rd_user, dst + got, n - got);
            if (r < 0) return FQZ_E_IO;
            if (r == 0) break;
            got += (size_t)r;
        }
        return (long)got;
    }
};

enum { ST_FREE = 0, ST_LOADED = 1, ST_DONE = 2 };

struct Slot {
    int state = ST_FREE;
    fqz_ctx *lane = nullptr;
    DevBuf &d_new;      // the slice as it came over the link   } kept by the context across calls (pinned allocations
    PinnedBuf &h_in, &h_out; //                                     } of this size cost tens of milliseconds)
    Slot(DevBuf &a, PinnedBuf &b, PinnedBuf &c) : d_new(a), h_in(b), h_out(c) {}
    size_t n_new = 0;
    bool eof = false;
    // filled by the calling thread for the drainer
    const uint8_t *d_res = nullptr;
    size_t res_len = 0;
    bool header_first = false;
    uint8_t header[FQZ_FILE_HEADER_SIZE];
};

struct Pipe {
    std::mutex mu;
    std::condition_variable cv;
    Slot slot[NS];
    explicit Pipe(fqz_ctx *ctx) : slot{Slot(ctx->sl_new[0], ctx->sl_hin[0], ctx->sl_hout[0]), Slot(ctx->sl_new[1], ctx->sl_hin[1], ctx->sl_hout[1]),
                                       Slot(ctx->sl_new[2], ctx->sl_hin[2], ctx->sl_hout[2])} {}
    int err = 0;
    bool in_done = false;   // the feeder has handed over its last slice
    long n_batches = -1;    // total number of batches, known once the calling thread has seen the end
    void fail(int rc) { std::lock_guard<std::mutex> g(mu); if (!err) err = rc; cv.notify_all(); }
    // waits until slot k has state `want` (or an error / the end of the job); returns false to stop
    bool wait_state(int k, int want)
    {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return err || slot[k].state == want; });
        return !err;
    }
    void set_state(int k, int st) { std::lock_guard<std::mutex> g(mu); slot[k].state = st; cv.notify_all(); }
};

int ensure_lanes(fqz_ctx *ctx)
{
    while ((int)ctx->lanes.size() < NS) {
        fqz_ctx *c = nullptr;
        int rc = fqz_ctx_create(ctx->device, &c);
        if (rc) return rc;
        ctx->lanes.push_back(c);
    }
    return FQZ_OK;
}

// drainer: batches leave in order (collectAndWriteResults compress.go:365-403)
void drain_loop(Pipe &P, Io &io, int device)
{
    (void)hipSetDevice(device);
    for (long k = 0;; k++) {
        {
            std::unique_lock<std::mutex> g(P.mu);
            P.cv.wait(g, [&] { return P.err || P.slot[k % NS].state == ST_DONE || (P.n_batches >= 0 && k >= P.n_batches); });
            if (P.err || (P.n_batches >= 0 && k >= P.n_batches)) return;
        }
        Slot &s = P.slot[k % NS];
        int rc = FQZ_OK;
        auto put = [&](const uint8_t *host, const uint8_t *dev, size_t n) {
            if (!n || rc) return;
            if (io.count_only) { io.written += n; return; }
            if (io.mem_out) {
                if (io.written + n > io.mem_out_cap) { rc = FQZ_E_DST_SMALL; return; }
                if (host) memcpy(io.mem_out + io.written, host, n);
                else if (hipMemcpy(io.mem_out + io.written, dev, n, hipMemcpyDeviceToHost) != hipSuccess) rc = FQZ_E_HIP;
                io.written += n;
                return;
            }
            const uint8_t *src = host;
            if (!src) {
                if (s.h_out.ensure(n)) { rc = FQZ_E_NOMEM; return; }
                if (hipMemcpy(s.h_out.p, dev, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = FQZ_E_HIP; return; }
                src = s.h_out.as<uint8_t>();
            }
            if (io.wr(io.wr_user, src, n)) { rc = FQZ_E_IO; return; }
            io.written += n;
        };
        if (s.header_first) put(s.header, nullptr, FQZ_FILE_HEADER_SIZE);
        put(nullptr, s.d_res, s.res_len);
        if (rc) { P.fail(rc); return; }
        P.set_state((int)(k % NS), ST_FREE);
    }
}

} // namespace

// ===========================================================================
// compress.Compress
// ===========================================================================
static int compress_job(fqz_ctx *ctx, Io &io, const fqz_options *opts)
{
    fqz_options o = {FQZ_DEFAULT_BLOCK_SIZE, 0};                       // compress.go:126-128 (nil opts)
    if (opts) o = *opts;
    if (!o.block_size) o.block_size = FQZ_DEFAULT_BLOCK_SIZE;          // compress.go:129-131
    const uint32_t rpb = FQZ_DEFAULT_BLOCK_SIZE;                       // batches are always 100 000 records (compress.go:48-52, App. B-4)
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_lanes(ctx);
    if (rc) return rc;
    Pipe P(ctx);
    for (int i = 0; i < NS; i++) P.slot[i].lane = ctx->lanes[i];
    const size_t slice = slice_bytes();
    const int device = ctx->device;

    // feeder (produceCompressJobs compress.go:303-363, minus the parsing: the GPU indexes the text itself)
    std::thread feeder([&] {
        (void)hipSetDevice(device);
        for (long k = 0;; k++) {
            Slot &s = P.slot[k % NS];
            if (!P.wait_state((int)(k % NS), ST_FREE)) return;
            if (s.d_new.ensure(slice + 64)) { P.fail(FQZ_E_NOMEM); return; }
            size_t n = 0;
            if (io.mem_in || !io.rd) { // memory source: straight over the link
                n = io.mem_in_n - io.mem_in_pos < slice ? io.mem_in_n - io.mem_in_pos : slice;
                if (n && hipMemcpy(s.d_new.p, io.mem_in + io.mem_in_pos, n, hipMemcpyHostToDevice) != hipSuccess) { P.fail(FQZ_E_HIP); return; }
                io.mem_in_pos += n;
                s.eof = io.mem_in_pos == io.mem_in_n;
            } else {
                if (s.h_in.ensure(slice)) { P.fail(FQZ_E_NOMEM); return; }
                long r = io.read_full(s.h_in.as<uint8_t>(), slice);
                if (r < 0) { P.fail((int)r); return; }
                n = (size_t)r;
                s.eof = n < slice;
                if (n && hipMemcpy(s.d_new.p, s.h_in.p, n, hipMemcpyHostToDevice) != hipSuccess) { P.fail(FQZ_E_HIP); return; }
            }
            s.n_new = n;
            const bool last = s.eof;
            P.set_state((int)(k % NS), ST_LOADED);
            if (last) return;
        }
    });
    std::thread drainer([&] { drain_loop(P, io, device); });

    // calling thread: the batch = [unconsumed tail of the previous batch | new slice] -> device pipeline
    int enc = FQZ_DETECT_ENCODING;                                      // decided on the first batch (compress.go:146-154)
    uint8_t flags = 0;
    bool first = true;
    const uint8_t *carry = nullptr; // lives in the previous lane's text buffer
    size_t carry_len = 0;
    long k = 0;
    for (;; k++) {
        Slot &s = P.slot[k % NS];
        if (!P.wait_state((int)(k % NS), ST_LOADED)) break;
        fqz_ctx *lane = s.lane;
        const size_t n_text = carry_len + s.n_new;
        const bool final_batch = s.eof;
        if (n_text >= 0x7FFFFFFFull) { P.fail(FQZ_E_TOO_LARGE); break; } // one device pass addresses < 2 GiB of text: a block that large cannot be encoded
        if ((rc = lane->d_in.ensure(n_text + 64))) { P.fail(rc); break; }
        uint8_t *d_text = lane->d_in.as<uint8_t>();
        hipError_t he = hipSuccess;
        if (carry_len) he = hipMemcpyAsync(d_text, carry, carry_len, hipMemcpyDeviceToDevice, lane->stream);
        if (he == hipSuccess && s.n_new) he = hipMemcpyAsync(d_text + carry_len, s.d_new.p, s.n_new, hipMemcpyDeviceToDevice, lane->stream);
        if (he != hipSuccess) { P.fail(fqz_set_hip_error(he, "hipMemcpyAsync(batch)")); break; }
        const size_t cap = fqz_encode_bound_blocks(n_text, rpb);
        if ((rc = lane->d_out.ensure(cap + 64))) { P.fail(rc); break; }
        fqz_batch_result res;
        for (int attempt = 0;; attempt++) {
            rc = fqz_enc_launch(lane, d_text, n_text, rpb, enc, final_batch ? FQZ_BATCH_FINAL : 0, lane->d_out.as<uint8_t>(), cap, lane->stream);
            if (!rc) rc = fqz_enc_finish(lane, &res, nullptr, nullptr, 0);
            if (rc == FQZ_E_TOO_LARGE && attempt < 3) continue; // the context has resized itself (very short lines): same launch again
            break;
        }
        if (rc) { P.fail(rc); break; }                                  // "parsing FASTQ: ..." / "compressing block: ..."
        s.header_first = false;
        if (first && (final_batch || res.n_blocks)) { // block 0 is in this batch: its records decided the encoding (a batch that held less than one block decides nothing)
            enc = res.qual_encoding;
            if (enc == FQZ_ENCODING_PHRED64) flags |= FQZ_FLAG_PHRED64; // compress.go:162-164
            fqz_file_header fh = {FQZ_VERSION2, o.block_size, flags};  // compress.go:157-161
            fqz_write_file_header(&fh, s.header);
            s.header_first = true;
            first = false;
        }
        s.d_res = lane->d_out.as<uint8_t>();
        s.res_len = res.out_len;
        // what was not consumed (less than one block, unless nothing fitted: then the whole batch) opens the next batch
        const size_t used = final_batch ? n_text : (size_t)res.consumed;
        carry = d_text + used;
        carry_len = n_text - used;
        P.set_state((int)(k % NS), ST_DONE);
        if (final_batch) { k++; break; }
    }
    {
        std::lock_guard<std::mutex> g(P.mu);
        P.n_batches = k;
        P.cv.notify_all();
    }
    feeder.join();
    drainer.join();
    return P.err;
}

// ===========================================================================
// compress.Decompress
// ===========================================================================
static int decompress_job(fqz_ctx *ctx, Io &io, const fqz_decompress_options *opts)
{
    (void)opts;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_lanes(ctx);
    if (rc) return rc;
    uint8_t hdr[FQZ_FILE_HEADER_SIZE];
    long got = io.read_full(hdr, FQZ_FILE_HEADER_SIZE);
    if (got < 0) return (int)got;
    fqz_file_header fh;
    rc = fqz_read_file_header(hdr, (size_t)got, &fh);                   // compress.go:567-570
    if (rc) return rc;
    if (fh.version != FQZ_VERSION1 && fh.version != FQZ_VERSION2) return FQZ_E_FILE_VERSION; // compress.go:571-573
    const int enc = (fh.flags & FQZ_FLAG_PHRED64) ? FQZ_ENCODING_PHRED64 : FQZ_ENCODING_PHRED33; // compress.go:576-579
    Pipe P(ctx);
    for (int i = 0; i < NS; i++) P.slot[i].lane = ctx->lanes[i];
    const size_t slice = slice_bytes() / 4; // compressed bytes per batch: the text is ~4x larger
    const int device = ctx->device;
    const uint32_t hs = fh.version == FQZ_VERSION1 ? 32 : 36;

    // feeder: whole blocks (readNextDecompressJob compress.go:721-758) into a slot, then over the link
    std::thread feeder([&] {
        (void)hipSetDevice(device);
        std::vector<uint8_t> pending; // a block header read ahead of a full slot
        for (long k = 0;; k++) {
            Slot &s = P.slot[k % NS];
            if (!P.wait_state((int)(k % NS), ST_FREE)) return;
            size_t n = 0;
            bool eof = false;
            if (io.mem_in || !io.rd) { // walk the block headers in place, then one copy
                const uint8_t *p = io.mem_in + io.mem_in_pos;
                const size_t left = io.mem_in_n - io.mem_in_pos;
                while (n < left) {
                    fqz_block_header bh;
                    int h = fqz_read_block_header(p + n, left - n, fh.version, &bh);
                    if (h < 0) { P.fail(h); return; }                   // "reading block header: unexpected EOF"
                    unsigned long long pay = (unsigned long long)bh.seq_size + bh.qual_size + bh.header_size + bh.plus_size + bh.npos_size + bh.lengths_size;
                    if (pay > left - n - (size_t)h) { P.fail(FQZ_E_READ_DATA); return; } // compress.go:732
                    const size_t blk = (size_t)h + (size_t)pay;
                    if (n && n + blk > slice) break;
                    n += blk;
                }
                if (n >= 0x7FFFFFFFull) { P.fail(FQZ_E_TOO_LARGE); return; }
                if (s.d_new.ensure(n + 64)) { P.fail(FQZ_E_NOMEM); return; }
                if (n && hipMemcpy(s.d_new.p, p, n, hipMemcpyHostToDevice) != hipSuccess) { P.fail(FQZ_E_HIP); return; }
                io.mem_in_pos += n;
                eof = io.mem_in_pos == io.mem_in_n;
            } else {
                size_t cap = slice + (64u << 20);
                if (s.h_in.ensure(cap)) { P.fail(FQZ_E_NOMEM); return; }
                uint8_t *h_in = s.h_in.as<uint8_t>();
                for (;;) {
                    uint8_t bhb[36];
                    if (!pending.empty()) { memcpy(bhb, pending.data(), hs); pending.clear(); }
                    else {
                        long r = io.read_full(bhb, hs);
                        if (r < 0) { P.fail((int)r); return; }
                        if (r == 0) { eof = true; break; }              // clean EOF at a block boundary (compress.go:614-617)
                        if ((uint32_t)r < hs) { P.fail(FQZ_E_SHORT); return; }
                    }
                    fqz_block_header bh;
                    (void)fqz_read_block_header(bhb, hs, fh.version, &bh);
                    const unsigned long long pay = (unsigned long long)bh.seq_size + bh.qual_size + bh.header_size + bh.plus_size + bh.npos_size + bh.lengths_size;
                    const size_t blk = hs + (size_t)pay;
                    if (n && n + blk > slice) { pending.assign(bhb, bhb + hs); break; }
                    if (n + blk >= 0x7FFFFFFFull) { P.fail(FQZ_E_TOO_LARGE); return; }
                    if (n + blk > cap) { // one huge block: grow the pinned buffer (keeps what is already there)
                        PinnedBuf bigger;
                        if (bigger.ensure(n + blk)) { P.fail(FQZ_E_NOMEM); return; }
                        memcpy(bigger.p, h_in, n);
                        s.h_in.release();
                        s.h_in.p = bigger.p; s.h_in.cap = bigger.cap;
                        h_in = s.h_in.as<uint8_t>();
                        cap = s.h_in.cap;
                    }
                    memcpy(h_in + n, bhb, hs);
                    long r = io.read_full(h_in + n + hs, (size_t)pay);
                    if (r < 0) { P.fail((int)r); return; }
                    if ((unsigned long long)r < pay) { P.fail(FQZ_E_READ_DATA); return; }
                    n += blk;
                }
                if (s.d_new.ensure(n + 64)) { P.fail(FQZ_E_NOMEM); return; }
                if (n && hipMemcpy(s.d_new.p, h_in, n, hipMemcpyHostToDevice) != hipSuccess) { P.fail(FQZ_E_HIP); return; }
            }
            s.n_new = n;
            s.eof = eof;
            P.set_state((int)(k % NS), ST_LOADED);
            if (eof) return;
        }
    });
    std::thread drainer([&] { drain_loop(P, io, device); });
    long k = 0;
    for (;; k++) {
        Slot &s = P.slot[k % NS];
        if (!P.wait_state((int)(k % NS), ST_LOADED)) break;
        fqz_ctx *lane = s.lane;
        s.header_first = false;
        s.res_len = 0;
        if (s.n_new) {
            fqz_batch_result res;
            lane->dec.skip_assemble = io.count_only; // only the size is wanted: everything but the text assembly
            rc = fqz_dec_launch(lane, s.d_new.as<uint8_t>(), s.n_new, fh.version, enc, nullptr, 0, lane->stream);
            if (!rc) rc = fqz_dec_finish(lane, &res);
            lane->dec.skip_assemble = false;
            if (rc) { P.fail(rc); break; }
            s.d_res = lane->dec.d_out;
            s.res_len = res.out_len;
        }
        const bool last = s.eof;
        P.set_state((int)(k % NS), ST_DONE);
        if (last) { k++; break; }
    }
    {
        std::lock_guard<std::mutex> g(P.mu);
        P.n_batches = k;
        P.cv.notify_all();
    }
    feeder.join();
    drainer.join();
    return P.err;
}

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" int fqz_compress_stream(fqz_ctx *ctx, fqz_read_fn rd, void *rd_user, fqz_write_fn wr, void *wr_user, const fqz_options *opts)
{
    if (!ctx || !rd || !wr) return FQZ_E_ARG;
    Io io;
    io.rd = rd; io.rd_user = rd_user; io.wr = wr; io.wr_user = wr_user;
    return compress_job(ctx, io, opts);
}

extern "C" int fqz_decompress_stream(fqz_ctx *ctx, fqz_read_fn rd, void *rd_user, fqz_write_fn wr, void *wr_user, const fqz_decompress_options *opts)
{
    if (!ctx || !rd || !wr) return FQZ_E_ARG;
    Io io;
    io.rd = rd; io.rd_user = rd_user; io.wr = wr; io.wr_user = wr_user;
    return decompress_job(ctx, io, opts);
}

extern "C" int fqz_compress(fqz_ctx *ctx, const uint8_t *fastq, size_t n, uint8_t *out, size_t out_cap, size_t *out_len, const fqz_options *opts)
{
    if (!ctx || (!fastq && n) || !out || !out_len) return FQZ_E_ARG;
    *out_len = 0;
    Io io;
    static const uint8_t nothing = 0;
    io.mem_in = fastq ? fastq : &nothing; io.mem_in_n = n;
    io.mem_out = out; io.mem_out_cap = out_cap;
    int rc = compress_job(ctx, io, opts);
    if (rc) return rc;
    *out_len = io.written;
    return FQZ_OK;
}

extern "C" int fqz_decompress(fqz_ctx *ctx, const uint8_t *fqz, size_t n, uint8_t *out, size_t out_cap, size_t *out_len, const fqz_decompress_options *opts)
{
    if (!ctx || !fqz || !out_len) return FQZ_E_ARG;
    *out_len = 0;
    Io io;
    io.mem_in = fqz; io.mem_in_n = n;
    io.mem_out = out; io.mem_out_cap = out_cap; io.count_only = out == nullptr;
    int rc = decompress_job(ctx, io, opts);
    if (rc) return rc;
    *out_len = io.written;
    return FQZ_OK;
}

namespace {
struct Grow { uint8_t *p = nullptr; size_t n = 0, cap = 0; };
int grow_write(void *u, const uint8_t *src, size_t n)
{
    Grow *g = (Grow *)u;
    if (g->n + n > g->cap) {
        size_t want = g->cap ? g->cap * 2 : (1u << 20);
        while (want < g->n + n) want *= 2;
        uint8_t *q = (uint8_t *)realloc(g->p, want);
        if (!q) return 1;
        g->p = q; g->cap = want;
    }
    memcpy(g->p + g->n, src, n);
    g->n += n;
    return 0;
}
long file_read(void *u, uint8_t *dst, size_t cap) { size_t r = fread(dst, 1, cap, (FILE *)u); return ferror((FILE *)u) ? -1 : (long)r; }
int file_write(void *u, const uint8_t *src, size_t n) { return fwrite(src, 1, n, (FILE *)u) != n; }
} // namespace

// The decoded size is only known after the lengths streams have been decoded, so a caller that wants the text in one call
// lets the library allocate it (free with fqz_buffer_free).  One decode, no sizing pass.
extern "C" int fqz_decompress_alloc(fqz_ctx *ctx, const uint8_t *fqz, size_t n, uint8_t **out, size_t *out_len, const fqz_decompress_options *opts)
{
    if (!ctx || !fqz || !out || !out_len) return FQZ_E_ARG;
    *out = nullptr; *out_len = 0;
    Io io;
    Grow g;
    io.mem_in = fqz; io.mem_in_n = n;
    io.wr = grow_write; io.wr_user = &g;
    int rc = decompress_job(ctx, io, opts);
    if (rc) { free(g.p); return rc == FQZ_E_IO ? FQZ_E_NOMEM : rc; }
    *out = g.p ? g.p : (uint8_t *)malloc(1);
    *out_len = g.n;
    return FQZ_OK;
}
extern "C" void fqz_buffer_free(uint8_t *p) { free(p); }

// ---- file forms (cmd/fqpack/main.go:190-203 execute): 1 MiB-buffered stdio at both ends, nothing held but the slots
extern "C" int fqz_compress_file(fqz_ctx *ctx, const char *in_path, const char *out_path, const fqz_options *opts)
{
    if (!ctx || !in_path || !out_path) return FQZ_E_ARG;
    FILE *fi = strcmp(in_path, "-") ? fopen(in_path, "rb") : stdin;
    if (!fi) return FQZ_E_IO;
    FILE *fo = strcmp(out_path, "-") ? fopen(out_path, "wb") : stdout;
    if (!fo) { if (fi != stdin) fclose(fi); return FQZ_E_IO; }
    setvbuf(fi, nullptr, _IOFBF, 1 << 20);
    setvbuf(fo, nullptr, _IOFBF, 1 << 20);
    int rc = fqz_compress_stream(ctx, file_read, fi, file_write, fo, opts);
    if (fflush(fo) && !rc) rc = FQZ_E_IO;
    if (fi != stdin) fclose(fi);
    if (fo != stdout && fclose(fo) && !rc) rc = FQZ_E_IO;
    return rc;
}

extern "C" int fqz_decompress_file(fqz_ctx *ctx, const char *in_path, const char *out_path, const fqz_decompress_options *opts)
{
    if (!ctx || !in_path || !out_path) return FQZ_E_ARG;
    FILE *fi = strcmp(in_path, "-") ? fopen(in_path, "rb") : stdin;
    if (!fi) return FQZ_E_IO;
    FILE *fo = strcmp(out_path, "-") ? fopen(out_path, "wb") : stdout;
    if (!fo) { if (fi != stdin) fclose(fi); return FQZ_E_IO; }
    setvbuf(fi, nullptr, _IOFBF, 1 << 20);
    setvbuf(fo, nullptr, _IOFBF, 1 << 20);
    int rc = fqz_decompress_stream(ctx, file_read, fi, file_write, fo, opts);
    if (fflush(fo) && !rc) rc = FQZ_E_IO;
    if (fi != stdin) fclose(fi);
    if (fo != stdout && fclose(fo) && !rc) rc = FQZ_E_IO;
    return rc;
}
