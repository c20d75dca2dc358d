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
#include <functional>
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
    const uint8_t *dev_in = nullptr; size_t dev_in_n = 0, dev_in_pos = 0; // memory source, first part: already on the device (fqz_compress_multi)
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
    std::vector<uint64_t> boff; // decompression: where the feeder found the block headers of the slice (a hint for the device, fqz_dec_launch)
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
// One shard of a multi-device job (fqz_compress_multi): the encoding is the file's, decided by the shard that holds block 0
struct ShardRole {
    int explicit_enc = -1;                 // >= 0: encode with this FQZ_ENCODING_*, write no file header
    std::function<void(int)> on_enc;       // the shard with block 0 reports what it detected
    std::vector<std::pair<uint64_t, uint32_t>> *table = nullptr; // block table wanted: (offset in this shard's output, records) of its blocks
};

// block table of a version-3 file (include/fqz.h): appended behind the last block
static void put_u32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
static void put_u64(std::vector<uint8_t> &v, uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i))); }
static std::vector<uint8_t> block_table_bytes(const std::vector<std::pair<uint64_t, uint32_t>> &t, uint64_t at)
{
    std::vector<uint8_t> v;
    put_u32(v, FQZ_BLOCK_TABLE_MARK);
    v.insert(v.end(), {'F', 'Q', 'Z', 'X'});
    put_u32(v, (uint32_t)t.size());
    for (const auto &e : t) { put_u64(v, e.first); put_u32(v, e.second); }
    put_u64(v, at);
    v.insert(v.end(), {'F', 'Q', 'Z', 'X'});
    return v;
}


static int compress_job(fqz_ctx *ctx, Io &io, const fqz_options *opts, const ShardRole *role = nullptr)
{
    fqz_options o = {FQZ_DEFAULT_BLOCK_SIZE, 0, 0, 0};                  // compress.go:126-128 (nil opts)
    if (opts) o = *opts;
    if (o.container_version && o.container_version != FQZ_VERSION2 && o.container_version != FQZ_VERSION3) return FQZ_E_FILE_VERSION;
    const bool v3 = o.container_version == FQZ_VERSION3;
    const bool want_table = o.block_index != 0;
    if (want_table && !v3) return FQZ_E_ARG; // (a version-2 file is a plain chain of blocks: the stock reader would trip over a table)
    std::vector<std::pair<uint64_t, uint32_t>> table_own, &table = (role && role->table) ? *role->table : table_own;
    uint64_t out_pos = 0; // bytes this job has produced so far
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
                const size_t nd = io.dev_in_n - io.dev_in_pos < slice ? io.dev_in_n - io.dev_in_pos : slice; // resident part first
                // (a device-to-device hipMemcpy may return before the copy has run; the batch is put together on the lane's own stream,
                //  which does not wait for the null stream: the copy is made on this thread's stream and waited for.  Without the wait a
                //  shard of fqz_compress_multi now and then read its slice half copied - a parse error once in a few hundred runs)
                if (nd && (hipMemcpyAsync(s.d_new.p, io.dev_in + io.dev_in_pos, nd, hipMemcpyDeviceToDevice, hipStreamPerThread) != hipSuccess ||
                           hipStreamSynchronize(hipStreamPerThread) != hipSuccess)) { P.fail(FQZ_E_HIP); return; }
                io.dev_in_pos += nd;
                const size_t nh = io.mem_in_n - io.mem_in_pos < slice - nd ? io.mem_in_n - io.mem_in_pos : slice - nd;
                if (nh && hipMemcpy(s.d_new.as<uint8_t>() + nd, io.mem_in + io.mem_in_pos, nh, hipMemcpyHostToDevice) != hipSuccess) { P.fail(FQZ_E_HIP); return; }
                io.mem_in_pos += nh;
                n = nd + nh;
                s.eof = io.mem_in_pos == io.mem_in_n && io.dev_in_pos == io.dev_in_n;
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
    if (role && role->explicit_enc >= 0) { enc = role->explicit_enc; first = false; }
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
        std::vector<uint64_t> boff, blen;
        if (want_table) { boff.resize(n_text / (6ull * rpb) + 2); blen.resize(boff.size()); } // (a record is at least six bytes)
        for (int attempt = 0;; attempt++) {
            rc = fqz_enc_launch(lane, d_text, n_text, rpb, enc, (final_batch ? FQZ_BATCH_FINAL : 0u) | (v3 ? FQZ_BATCH_V3 : 0u), lane->d_out.as<uint8_t>(), cap, lane->stream);
            if (!rc) rc = fqz_enc_finish(lane, &res, want_table ? boff.data() : nullptr, want_table ? blen.data() : nullptr, boff.size());
            if (rc == FQZ_E_TOO_LARGE && attempt < 3) continue; // the context has resized itself (very short lines): same launch again
            break;
        }
        if (rc) { P.fail(rc); break; }                                  // "parsing FASTQ: ..." / "compressing block: ..."
        s.header_first = false;
        if (first && (final_batch || res.n_blocks)) { // block 0 is in this batch: its records decided the encoding (a batch that held less than one block decides nothing)
            enc = res.qual_encoding;
            if (enc == FQZ_ENCODING_PHRED64) flags |= FQZ_FLAG_PHRED64; // compress.go:162-164
            fqz_file_header fh = {(uint8_t)(v3 ? FQZ_VERSION3 : FQZ_VERSION2), o.block_size, flags};  // compress.go:157-161
            fqz_write_file_header(&fh, s.header);
            s.header_first = true;
            first = false;
            if (role && role->on_enc) role->on_enc(enc);
        }
        if (s.header_first) out_pos += FQZ_FILE_HEADER_SIZE;
        if (want_table)
            for (uint32_t b = 0; b < res.n_blocks; b++)
                table.emplace_back(out_pos + boff[b], b + 1 < res.n_blocks ? rpb : res.n_records - rpb * (res.n_blocks - 1));
        out_pos += res.out_len;
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
    if (!P.err && want_table && !(role && role->table)) { // everything is out: the table goes behind the last block
        const std::vector<uint8_t> t = block_table_bytes(table, out_pos);
        if (io.count_only) io.written += t.size();
        else if (io.mem_out) {
            if (io.written + t.size() > io.mem_out_cap) return FQZ_E_DST_SMALL;
            memcpy(io.mem_out + io.written, t.data(), t.size());
            io.written += t.size();
        } else {
            if (io.wr(io.wr_user, t.data(), t.size())) return FQZ_E_IO;
            io.written += t.size();
        }
    }
    return P.err;
}

// ===========================================================================
// compress.Decompress
// ===========================================================================
// body_of: the source starts at a block header and belongs to a file with this header (a shard of fqz_decompress_multi)
static int decompress_job(fqz_ctx *ctx, Io &io, const fqz_decompress_options *opts, const fqz_file_header *body_of = nullptr)
{
    (void)opts;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_lanes(ctx);
    if (rc) return rc;
    fqz_file_header fh;
    if (body_of) fh = *body_of;
    else {
        uint8_t hdr[FQZ_FILE_HEADER_SIZE];
        long got = io.read_full(hdr, FQZ_FILE_HEADER_SIZE);
        if (got < 0) return (int)got;
        rc = fqz_read_file_header(hdr, (size_t)got, &fh);               // compress.go:567-570
        if (rc) return rc;
    }
    if (fh.version != FQZ_VERSION1 && fh.version != FQZ_VERSION2 && fh.version != FQZ_VERSION3) return FQZ_E_FILE_VERSION; // compress.go:571-573
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
                bool table = false;
                s.boff.clear();
                while (n < left) {
                    if (fh.version == FQZ_VERSION3 && left - n >= 8 && !memcmp(p + n, "\xFF\xFF\xFF\xFF" "FQZX", 8)) { table = true; break; } // the block table: the chain ends here
                    fqz_block_header bh;
                    int h = fqz_read_block_header(p + n, left - n, fh.version, &bh);
                    if (h < 0) { P.fail(h); return; }                   // "reading block header: unexpected EOF"
                    unsigned long long pay = (unsigned long long)bh.seq_size + bh.qual_size + bh.header_size + bh.plus_size + bh.npos_size + bh.lengths_size;
                    if (pay > left - n - (size_t)h) { P.fail(FQZ_E_READ_DATA); return; } // compress.go:732
                    const size_t blk = (size_t)h + (size_t)pay;
                    if (n && n + blk > slice) break;
                    s.boff.push_back(n);
                    n += blk;
                }
                if (n >= 0x7FFFFFFFull) { P.fail(FQZ_E_TOO_LARGE); return; }
                if (s.d_new.ensure(n + 64)) { P.fail(FQZ_E_NOMEM); return; }
                if (n && hipMemcpy(s.d_new.p, p, n, hipMemcpyHostToDevice) != hipSuccess) { P.fail(FQZ_E_HIP); return; }
                io.mem_in_pos += n;
                if (table) io.mem_in_pos = io.mem_in_n;
                eof = io.mem_in_pos == io.mem_in_n;
            } else {
                size_t cap = slice + (64u << 20);
                if (s.h_in.ensure(cap)) { P.fail(FQZ_E_NOMEM); return; }
                uint8_t *h_in = s.h_in.as<uint8_t>();
                s.boff.clear();
                for (;;) {
                    uint8_t bhb[36];
                    if (!pending.empty()) { memcpy(bhb, pending.data(), hs); pending.clear(); }
                    else {
                        long r = io.read_full(bhb, hs);
                        if (r < 0) { P.fail((int)r); return; }
                        if (r == 0) { eof = true; break; }              // clean EOF at a block boundary (compress.go:614-617)
                        // (a version-3 file whose only content is the block table: 24 bytes, fewer than a block header)
                        if ((uint32_t)r < hs && !(fh.version == FQZ_VERSION3 && r >= 8 && !memcmp(bhb, "\xFF\xFF\xFF\xFF" "FQZX", 8))) { P.fail(FQZ_E_SHORT); return; }
                    }
                    if (fh.version == FQZ_VERSION3 && !memcmp(bhb, "\xFF\xFF\xFF\xFF" "FQZX", 8)) { eof = true; break; } // the block table: the chain ends here (the rest is not read)
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
                    s.boff.push_back(n);
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
            lane->dec.hint_off = s.boff.data(); lane->dec.hint_n = s.boff.size(); // (the feeder walked the block headers already)
            rc = fqz_dec_launch(lane, s.d_new.as<uint8_t>(), s.n_new, fh.version, enc, nullptr, 0, lane->stream);
            if (!rc) rc = fqz_dec_finish(lane, &res);
            lane->dec.hint_off = nullptr; lane->dec.hint_n = 0;
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

// ===========================================================================
// compress.Compress over several devices: the worker pool of the reference (W goroutines pulling batches, ordered
// collector; compress.go:240-278, 365-403) becomes one host thread + one context per device.  Blocks are independent and
// the file is their concatenation, so every device gets a contiguous range of whole blocks:
//   phase A  each thread brings its byte range of the text onto its device and counts the lines in it
//   host     prefix sum of the counts -> for every device the first line that starts a block (a multiple of
//            4 x 100 000 lines: the parser takes exactly four lines per record, fqparser/parser.go:136-184) at or behind
//            its range; the tile that holds it is fetched and searched on the host
//   phase B  each thread runs the streaming pipeline above on its shard, fed from the resident range (plus, from the
//            host, the part of its last block that lies in the next range); the shard with block 0 detects the quality
//            encoding, the others wait for it (FlagPhred64 is a property of the file, compress.go:146-164)
//   host     the shards' outputs are concatenated in order (sizes are on the host after every finish)
// ===========================================================================
namespace {
const uint32_t MC_TILE = 1u << 16;
__global__ __launch_bounds__(256) void k_multi_count(const uint8_t *text, size_t n, uint32_t *counts)
{
    __shared__ uint32_t sh[4];
    const size_t base = (size_t)blockIdx.x * MC_TILE;
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x * 16; i < MC_TILE; i += 256 * 16) {
        const size_t off = base + i;
        if (off + 16 <= n) {
            const uint4 v = *(const uint4 *)(text + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 4; k++) {
                const uint32_t x = w[k] ^ 0x0A0A0A0Au;
                c += __popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u);
            }
        } else {
            for (size_t q = off; q < n && q < off + 16; q++) c += text[q] == '\n';
        }
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

struct Shard {
    int device = 0;
    fqz_ctx *ctx = nullptr;
    size_t r0 = 0, r1 = 0;             // byte range resident on the device
    DevBuf text, counts;
    std::vector<uint32_t> h_counts;    // newlines per MC_TILE of the range
    unsigned long long lines = 0;
    size_t a = 0, b = 0;               // the shard: text bytes [a, b), whole blocks
    std::vector<uint8_t> out;
    std::vector<std::pair<uint64_t, uint32_t>> table; // its blocks: (offset in `out`, records)
    int rc = 0;
};

int grow_vec(void *u, const uint8_t *src, size_t n)
{
    std::vector<uint8_t> *v = (std::vector<uint8_t> *)u;
    try { v->insert(v->end(), src, src + n); } catch (...) { return 1; }
    return 0;
}
} // namespace

extern "C" int fqz_compress_multi(const int *devices, int n_devices, const uint8_t *fastq, size_t n, uint8_t *out, size_t out_cap, size_t *out_len,
                                  const fqz_options *opts)
{
    if (!devices || n_devices < 1 || n_devices > 64 || (!fastq && n) || !out || !out_len) return FQZ_E_ARG;
    *out_len = 0;
    const uint32_t rpb = FQZ_DEFAULT_BLOCK_SIZE;
    const int N = n_devices;
    const bool want_table = opts && opts->block_index;
    if (want_table && opts->container_version != FQZ_VERSION3) return FQZ_E_ARG;
    std::vector<Shard> sh((size_t)N);
    // byte ranges: equal parts, cut at multiples of the counting tile
    for (int d = 0; d < N; d++) {
        sh[d].device = devices[d];
        sh[d].r0 = d ? sh[d - 1].r1 : 0;
        size_t r1 = d + 1 == N ? n : (size_t)((unsigned long long)n * (unsigned)(d + 1) / (unsigned)N) / MC_TILE * MC_TILE;
        sh[d].r1 = r1 < sh[d].r0 ? sh[d].r0 : r1;
    }
    // ---- phase A
    {
        std::vector<std::thread> th;
        for (int d = 0; d < N; d++)
            th.emplace_back([&, d] {
                Shard &s = sh[d];
                if (hipSetDevice(s.device) != hipSuccess) { s.rc = FQZ_E_HIP; return; }
                if ((s.rc = fqz_ctx_create(s.device, &s.ctx))) return;
                const size_t len = s.r1 - s.r0, tiles = (len + MC_TILE - 1) / MC_TILE;
                if ((s.rc = s.text.ensure(len + 64)) || (s.rc = s.counts.ensure(4 * (tiles + 1)))) return;
                const size_t step = (size_t)256 << 20;
                for (size_t o = 0; o < len; o += step) { // (pageable source: the copy of one slice overlaps the count of the one before)
                    const size_t m = len - o < step ? len - o : step;
                    if (hipMemcpyAsync(s.text.as<uint8_t>() + o, fastq + s.r0 + o, m, hipMemcpyHostToDevice, s.ctx->stream) != hipSuccess) { s.rc = FQZ_E_HIP; return; }
                }
                if (tiles) hipLaunchKernelGGL(k_multi_count, dim3((unsigned)tiles), dim3(256), 0, s.ctx->stream, s.text.as<uint8_t>(), len, s.counts.as<uint32_t>());
                s.h_counts.resize(tiles);
                if (tiles && hipMemcpyAsync(s.h_counts.data(), s.counts.p, 4 * tiles, hipMemcpyDeviceToHost, s.ctx->stream) != hipSuccess) { s.rc = FQZ_E_HIP; return; }
                if (hipStreamSynchronize(s.ctx->stream) != hipSuccess) { s.rc = FQZ_E_HIP; return; }
                for (uint32_t c : s.h_counts) s.lines += c;
            });
        for (auto &t : th) t.join();
    }
    int rc = 0;
    for (int d = 0; d < N && !rc; d++) rc = sh[d].rc;
    // ---- shard starts: the first block boundary at or behind the start of every range
    if (!rc) {
        std::vector<unsigned long long> L((size_t)N + 1, 0); // lines in front of range d
        for (int d = 0; d < N; d++) L[d + 1] = L[d] + sh[d].lines;
        const unsigned long long per_block = 4ull * rpb;
        size_t prev = 0;
        for (int d = 0; d < N && !rc; d++) {
            size_t a = 0;
            if (d) {
                const unsigned long long T = (L[d] + per_block - 1) / per_block * per_block; // line T starts a block
                if (T == 0) a = 0;
                else if (T > L[N]) a = n;
                else { // one byte behind newline number T - 1 (0-based), which lies in the range e with L[e] <= T - 1 < L[e + 1]
                    int e = 0; // (from the first range: when L[d] is itself a block boundary, newline T - 1 lies in an EARLIER range)
                    while (e + 1 < N && L[e + 1] <= T - 1) e++;
                    unsigned long long j = T - 1 - L[e];
                    size_t tile = 0;
                    while (tile < sh[e].h_counts.size() && j >= sh[e].h_counts[tile]) { j -= sh[e].h_counts[tile]; tile++; }
                    if (tile >= sh[e].h_counts.size()) { rc = FQZ_E_HIP; break; } // (cannot happen: the counts say the newline is there)
                    const size_t t0 = tile * (size_t)MC_TILE, tl = sh[e].r1 - sh[e].r0 - t0 < MC_TILE ? sh[e].r1 - sh[e].r0 - t0 : MC_TILE;
                    const uint8_t *p = fastq + sh[e].r0 + t0; // (the text is on the host as well: search the tile there)
                    size_t q = 0;
                    for (; q < tl; q++)
                        if (p[q] == '\n' && j-- == 0) break;
                    a = sh[e].r0 + t0 + q + 1;
                }
                if (a < prev) a = prev;
            }
            sh[d].a = a;
            prev = a;
        }
        for (int d = 0; d < N; d++) sh[d].b = d + 1 < N ? sh[d + 1].a : n;
    }
    // ---- phase B
    if (!rc) {
        std::mutex mu;
        std::condition_variable cv;
        int file_enc = -1; // -1 not known yet, -2 the first shard failed
        int lead = 0;      // the shard that holds block 0: it detects the encoding and writes the file header (shards in front of it are empty)
        for (int d = 0; d < N; d++)
            if (sh[d].a < sh[d].b) { lead = d; break; }
        std::vector<std::thread> th;
        for (int d = 0; d < N; d++)
            th.emplace_back([&, d] {
                Shard &s = sh[d];
                (void)hipSetDevice(s.device);
                ShardRole role;
                if (want_table) role.table = &s.table;
                if (d == lead) role.on_enc = [&](int e) { std::lock_guard<std::mutex> g(mu); file_enc = e; cv.notify_all(); };
                else {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return file_enc != -1; });
                    if (file_enc < 0) { s.rc = 0; return; } // (the error is the first shard's)
                    role.explicit_enc = file_enc;
                    if (s.a >= s.b) return; // nothing left for this device
                }
                Io io;
                static const uint8_t nothing = 0;
                // the resident part of [a, b), then what lies behind the range on the host
                if (s.a < s.r0) { s.rc = FQZ_E_ARG; return; } // (starts are never in front of their range)
                const size_t dev_end = s.b < s.r1 ? s.b : s.r1, ha = s.a > s.r1 ? s.a : s.r1;
                if (s.a < dev_end) { io.dev_in = s.text.as<uint8_t>() + (s.a - s.r0); io.dev_in_n = dev_end - s.a; }
                io.mem_in = s.b > ha ? fastq + ha : &nothing;
                io.mem_in_n = s.b > ha ? s.b - ha : 0;
                io.wr = grow_vec; io.wr_user = &s.out;
                s.rc = compress_job(s.ctx, io, opts, &role);
                if (d == lead) { std::lock_guard<std::mutex> g(mu); if (file_enc == -1) { file_enc = -2; cv.notify_all(); } }
            });
        for (auto &t : th) t.join();
        for (int d = 0; d < N && !rc; d++) rc = sh[d].rc; // the error of the earliest shard, like the ordered collector
    }
    for (int d = 0; d < N; d++)
        if (sh[d].ctx) {
            (void)hipSetDevice(sh[d].device);
            sh[d].text.release(); sh[d].counts.release();
            fqz_ctx_destroy(sh[d].ctx);
        }
    if (rc) return rc;
    size_t total = 0;
    std::vector<std::pair<uint64_t, uint32_t>> table;
    for (int d = 0; d < N; d++) {
        for (const auto &e : sh[d].table) table.emplace_back(total + e.first, e.second); // (offsets in the shard's output -> in the file)
        total += sh[d].out.size();
    }
    std::vector<uint8_t> tb;
    if (want_table) tb = block_table_bytes(table, total);
    if (total + tb.size() > out_cap) return FQZ_E_DST_SMALL;
    size_t pos = 0;
    for (int d = 0; d < N; d++) { if (!sh[d].out.empty()) memcpy(out + pos, sh[d].out.data(), sh[d].out.size()); pos += sh[d].out.size(); }
    if (!tb.empty()) memcpy(out + pos, tb.data(), tb.size());
    *out_len = total + tb.size();
    return FQZ_OK;
}

// compress.Decompress over several devices: the block headers are walked on the host (readNextDecompressJob,
// compress.go:721-758), every device takes a contiguous range of whole blocks of about the same compressed size through the
// streaming pipeline above, and the texts are concatenated in order.  Byte-identical to fqz_decompress on one device.
extern "C" int fqz_decompress_multi(const int *devices, int n_devices, const uint8_t *fqz, size_t n, uint8_t **out, size_t *out_len,
                                    const fqz_decompress_options *opts)
{
    if (!devices || n_devices < 1 || n_devices > 64 || !fqz || !out || !out_len) return FQZ_E_ARG;
    *out = nullptr; *out_len = 0;
    fqz_file_header fh;
    int rc = fqz_read_file_header(fqz, n, &fh);
    if (rc) return rc;
    if (fh.version != FQZ_VERSION1 && fh.version != FQZ_VERSION2 && fh.version != FQZ_VERSION3) return FQZ_E_FILE_VERSION;
    // block boundaries
    std::vector<size_t> starts;
    size_t pos = FQZ_FILE_HEADER_SIZE;
    size_t body_end = n; // (a version-3 file may end with its block table: the blocks end in front of it)
    while (pos < n) {
        if (fh.version == FQZ_VERSION3 && n - pos >= 8 && !memcmp(fqz + pos, "\xFF\xFF\xFF\xFF" "FQZX", 8)) { body_end = pos; break; }
        fqz_block_header bh;
        const int h = fqz_read_block_header(fqz + pos, n - pos, fh.version, &bh);
        if (h < 0) return h;
        const unsigned long long pay = (unsigned long long)bh.seq_size + bh.qual_size + bh.header_size + bh.plus_size + bh.npos_size + bh.lengths_size;
        if (pay > n - pos - (size_t)h) return FQZ_E_READ_DATA;
        starts.push_back(pos);
        pos += (size_t)h + (size_t)pay;
    }
    starts.push_back(body_end);
    const size_t nblk = starts.size() - 1;
    const int N = n_devices;
    struct DShard { size_t a = 0, b = 0; fqz_ctx *ctx = nullptr; std::vector<uint8_t> out; int rc = 0; };
    std::vector<DShard> sh((size_t)N);
    { // contiguous ranges of whole blocks: a device takes the blocks that end at or before its share of the compressed bytes
      // (none, when a block is larger than a share), the last device the rest
        size_t k = 0;
        for (int d = 0; d < N; d++) {
            sh[d].a = starts[k];
            if (d + 1 == N) k = nblk;
            else {
                const size_t want = FQZ_FILE_HEADER_SIZE + (size_t)((unsigned long long)(body_end - FQZ_FILE_HEADER_SIZE) * (unsigned)(d + 1) / (unsigned)N);
                while (k < nblk && starts[k + 1] <= want) k++;
            }
            sh[d].b = starts[k];
        }
    }
    std::vector<std::thread> th;
    for (int d = 0; d < N; d++)
        th.emplace_back([&, d] {
            DShard &s = sh[d];
            if (s.a >= s.b) return;
            if (hipSetDevice(devices[d]) != hipSuccess) { s.rc = FQZ_E_HIP; return; }
            if ((s.rc = fqz_ctx_create(devices[d], &s.ctx))) return;
            Io io;
            io.mem_in = fqz + s.a; io.mem_in_n = s.b - s.a;
            io.wr = grow_vec; io.wr_user = &s.out;
            s.rc = decompress_job(s.ctx, io, opts, &fh);
        });
    for (auto &t : th) t.join();
    for (int d = 0; d < N; d++)
        if (sh[d].ctx) { (void)hipSetDevice(devices[d]); fqz_ctx_destroy(sh[d].ctx); }
    for (int d = 0; d < N; d++) if (sh[d].rc) return sh[d].rc; // the error of the earliest shard
    size_t total = 0;
    for (int d = 0; d < N; d++) total += sh[d].out.size();
    uint8_t *buf = (uint8_t *)malloc(total ? total : 1);
    if (!buf) return FQZ_E_NOMEM;
    size_t at = 0;
    for (int d = 0; d < N; d++) { if (!sh[d].out.empty()) memcpy(buf + at, sh[d].out.data(), sh[d].out.size()); at += sh[d].out.size(); }
    *out = buf; *out_len = total;
    return FQZ_OK;
}

// The block table of a version-3 file (include/fqz.h), read from the end of the file.  Host only.
extern "C" int fqz_read_block_table(const uint8_t *fqz, size_t n, uint64_t *off, uint32_t *n_records, size_t cap, size_t *n_blocks)
{
    if (!fqz || !n_blocks) return FQZ_E_ARG;
    *n_blocks = 0;
    fqz_file_header fh;
    int rc = fqz_read_file_header(fqz, n, &fh);
    if (rc) return rc;
    auto u32 = [&](size_t p) { return (uint32_t)fqz[p] | ((uint32_t)fqz[p + 1] << 8) | ((uint32_t)fqz[p + 2] << 16) | ((uint32_t)fqz[p + 3] << 24); };
    auto u64 = [&](size_t p) { return (uint64_t)u32(p) | ((uint64_t)u32(p + 4) << 32); };
    if (fh.version != FQZ_VERSION3 || n < FQZ_FILE_HEADER_SIZE + 24 || memcmp(fqz + n - 4, "FQZX", 4)) return FQZ_E_ARG;
    const uint64_t at = u64(n - 12);
    if (at < FQZ_FILE_HEADER_SIZE || at > n - 24 || u32((size_t)at) != FQZ_BLOCK_TABLE_MARK || memcmp(fqz + at + 4, "FQZX", 4)) return FQZ_E_ARG;
    const uint32_t nb = u32((size_t)at + 8);
    if ((uint64_t)nb * 12 + 24 != n - at) return FQZ_E_ARG;
    *n_blocks = nb;
    if (!off && !n_records) return FQZ_OK;
    if (nb > cap) return FQZ_E_DST_SMALL;
    for (uint32_t b = 0; b < nb; b++) {
        const uint64_t o = u64((size_t)at + 12 + 12ull * b);
        if (o < FQZ_FILE_HEADER_SIZE || o + 36 > at) return FQZ_E_ARG; // (every entry points at a block header in front of the table)
        if (off) off[b] = o;
        if (n_records) n_records[b] = u32((size_t)at + 12 + 12ull * b + 8);
    }
    return FQZ_OK;
}
