// fqz_api.hip — the extern "C" surface of libfqzhip (include/fqz.h).
// Host logic mirrors internal/compress/compress.go (Compress :125, Decompress :558)
// and internal/fqformat/container.go; all codec arithmetic runs in HIP kernels.
#include "fqz_ctx.h"
#include "fqz_device.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

// ===========================================================================
// errors
// ===========================================================================
static thread_local char g_hip_err[256] = "";

int fqz_set_hip_error(hipError_t e, const char *what)
{
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) return FQZ_E_NO_DEVICE;
    if (e == hipErrorOutOfMemory) return FQZ_E_NOMEM;
    return FQZ_E_HIP;
}

extern "C" const char *fqz_last_hip_error(void) { return g_hip_err; }
extern "C" const char *fqz_version(void) { return "libfqzhip 0.1.0 gfx950"; }

extern "C" const char *fqz_strerror(int code)
{
    switch (code) {
    case FQZ_OK: return "ok";
    case FQZ_E_SHORT: return "unexpected EOF";
    case FQZ_E_MAGIC: return "invalid magic bytes: not an FQZ file";
    case FQZ_E_BLOCK_VERSION: return "unsupported block header version";
    case FQZ_E_FILE_VERSION: return "unsupported file version";
    case FQZ_E_HDR_AT: return "invalid FASTQ: header line must start with @";
    case FQZ_E_SEP_PLUS: return "invalid FASTQ: separator line must start with +";
    case FQZ_E_LEN_MISMATCH: return "invalid FASTQ: sequence and quality lengths must match";
    case FQZ_E_LONG_N: return "sequence has ambiguous bases beyond position 65536; N-position tracking is limited to 65536 bp";
    case FQZ_E_TRUNC_HEADER: return "truncated header data";
    case FQZ_E_TRUNC_PLUS: return "truncated plus-line payload data";
    case FQZ_E_TRUNC_SEQ: return "truncated sequence data";
    case FQZ_E_TRUNC_QUAL: return "truncated quality data";
    case FQZ_E_TRUNC_LEN: return "truncated length data";
    case FQZ_E_TRUNC_NPOS: return "truncated N position data";
    case FQZ_E_ENTROPY: return "decompressing stream: invalid or unsupported zstd frame";
    case FQZ_E_READ_DATA: return "reading compressed data: unexpected EOF";
    case FQZ_E_NOMEM: return "out of memory";
    case FQZ_E_DST_SMALL: return "destination buffer too small";
    case FQZ_E_FIELD_WRAP: return "header, plus-line payload or N count exceeds 65535 (u16 field would wrap)";
    case FQZ_E_NPOS_RANGE: return "N position beyond read length";
    case FQZ_E_HIP: return "HIP runtime error";
    case FQZ_E_NO_DEVICE: return "no HIP device available (libfqzhip has no CPU fallback)";
    case FQZ_E_ARG: return "invalid argument";
    case FQZ_E_TOO_LARGE: return "batch too large for one device pass";
    case FQZ_E_IO: return "I/O error";
    case FQZ_E_CHECKSUM: return "decompressing stream: CRC check failed";
    default: return "unknown error";
    }
}

// ===========================================================================
// context
// ===========================================================================
extern "C" int fqz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int fqz_ctx_create(int device, fqz_ctx **out)
{
    if (!out) return FQZ_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fqz_set_hip_error(e, "hipGetDeviceCount");
    if (n <= 0) { snprintf(g_hip_err, sizeof g_hip_err, "no HIP device"); return FQZ_E_NO_DEVICE; }
    if (device < 0 || device >= n) return FQZ_E_ARG;
    HIP_TRY(hipSetDevice(device));
    fqz_ctx *c = new (std::nothrow) fqz_ctx();
    if (!c) return FQZ_E_NOMEM;
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fqz_set_hip_error(e, "hipStreamCreate"); }
    *out = c;
    return FQZ_OK;
}

extern "C" void fqz_ctx_destroy(fqz_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    EncState &e = c->enc;
    DevBuf *eb[] = {&e.info, &e.tile_cnt, &e.ls, &e.E, &e.plans, &e.arena, &e.npos, &e.slots, &e.csize, &e.stamps, &e.lf, &e.zstate, &e.scan_state, &e.gmap, &e.xmap, &e.hside,
                    &e.segmeta, &e.seg, &e.hslots};
    for (DevBuf *b : eb) b->release();
    e.h_info.release(); e.h_plans.release(); e.h_front.release();
    if (e.ev_front) (void)hipEventDestroy(e.ev_front);
    if (e.ev_layout) (void)hipEventDestroy(e.ev_layout);
    for (int i = 0; i < 2; i++) if (c->half[i]) { fqz_ctx_destroy(c->half[i]); c->half[i] = nullptr; }
    (void)hipSetDevice(c->device);
    if (c->ev_half) (void)hipEventDestroy(c->ev_half);
    if (e.side) { (void)hipStreamSynchronize(e.side); (void)hipStreamDestroy(e.side); (void)hipEventDestroy(e.ev_fork); (void)hipEventDestroy(e.ev_join);
                  (void)hipStreamSynchronize(e.side2); (void)hipStreamDestroy(e.side2); (void)hipEventDestroy(e.ev_join2);
                  (void)hipStreamSynchronize(e.side3); (void)hipStreamDestroy(e.side3); (void)hipEventDestroy(e.ev_join3); (void)hipEventDestroy(e.ev_npos); (void)hipEventDestroy(e.ev_gmap); }
    DecState &d = c->dec;
    DevBuf *db[] = {&d.info, &d.blocks, &d.chunks, &d.frames, &d.streams, &d.rec, &d.partials, &d.tables, &d.lz_scratch};
    for (DevBuf *b : db) b->release();
    d.h_info.release(); d.h_blocks.release();
    d.hint.release(); d.h_hint.release();
    if (d.side) { (void)hipStreamSynchronize(d.side); (void)hipStreamDestroy(d.side); (void)hipEventDestroy(d.ev_fork); (void)hipEventDestroy(d.ev_join);
                  (void)hipStreamSynchronize(d.side2); (void)hipStreamDestroy(d.side2); (void)hipEventDestroy(d.ev_join2); (void)hipEventDestroy(d.ev_x); (void)hipEventDestroy(d.ev_joinx);
                  (void)hipEventDestroy(d.ev_huf); (void)hipEventDestroy(d.ev_seq); }
    c->prof.collect();
    for (hipEvent_t ev : c->prof.pool) (void)hipEventDestroy(ev);
    for (fqz_ctx *lane : c->lanes) fqz_ctx_destroy(lane);
    c->lanes.clear();
    (void)hipSetDevice(c->device);
    c->d_in.release(); c->d_out.release(); c->h_stage.release();
    for (int i = 0; i < 3; i++) { c->sl_new[i].release(); c->sl_hin[i].release(); c->sl_hout[i].release(); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ===========================================================================
// fqformat (container.go) — host side, byte-exact
// ===========================================================================
static inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline uint32_t get32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

extern "C" void fqz_write_file_header(const fqz_file_header *h, uint8_t out[10])
{
    out[0] = 'F'; out[1] = 'Q'; out[2] = 'Z'; out[3] = 0;
    out[4] = h->version;
    put32(out + 5, h->block_size);
    out[9] = h->flags;
}

extern "C" int fqz_read_file_header(const uint8_t *in, size_t n, fqz_file_header *h)
{
    if (n < 4) return FQZ_E_SHORT;
    if (!(in[0] == 'F' && in[1] == 'Q' && in[2] == 'Z' && in[3] == 0)) return FQZ_E_MAGIC;
    if (n < 10) return FQZ_E_SHORT;
    h->version = in[4];
    h->block_size = get32(in + 5);
    h->flags = in[9];
    return FQZ_OK;
}

extern "C" int fqz_write_block_header(const fqz_block_header *b, uint8_t version, uint8_t *out)
{
    if (version == FQZ_VERSION1) {
        const uint32_t v[8] = {b->num_records, b->seq_size, b->qual_size, b->header_size, b->npos_size, b->lengths_size,
                               b->original_seq_size, b->original_qual_size};
        for (int i = 0; i < 8; i++) put32(out + 4 * i, v[i]);
        return 32;
    }
    if (version == FQZ_VERSION2 || version == FQZ_VERSION3) {
        const uint32_t v[9] = {b->num_records, b->seq_size, b->qual_size, b->header_size, b->plus_size, b->npos_size,
                               b->lengths_size, b->original_seq_size, b->original_qual_size};
        for (int i = 0; i < 9; i++) put32(out + 4 * i, v[i]);
        return 36;
    }
    return FQZ_E_BLOCK_VERSION;
}

extern "C" int fqz_read_block_header(const uint8_t *in, size_t n, uint8_t version, fqz_block_header *b)
{
    memset(b, 0, sizeof *b);
    if (version == FQZ_VERSION1) {
        if (n < 32) return FQZ_E_SHORT;
        b->num_records = get32(in); b->seq_size = get32(in + 4); b->qual_size = get32(in + 8); b->header_size = get32(in + 12);
        b->npos_size = get32(in + 16); b->lengths_size = get32(in + 20); b->original_seq_size = get32(in + 24);
        b->original_qual_size = get32(in + 28);
        return 32;
    }
    if (version == FQZ_VERSION2 || version == FQZ_VERSION3) {
        if (n < 36) return FQZ_E_SHORT;
        b->num_records = get32(in); b->seq_size = get32(in + 4); b->qual_size = get32(in + 8); b->header_size = get32(in + 12);
        b->plus_size = get32(in + 16); b->npos_size = get32(in + 20); b->lengths_size = get32(in + 24);
        b->original_seq_size = get32(in + 28); b->original_qual_size = get32(in + 32);
        return 36;
    }
    return FQZ_E_BLOCK_VERSION;
}

// ===========================================================================
// bounds
// ===========================================================================
extern "C" size_t fqz_entropy_bound(size_t n)
{
    if (!n) return 0;
    // FQZ-H2 payload: index frame (24 + per zstd block its size, 3 bytes, and its entry points), per 64 KiB group a frame header and
    // a checksum, per block a header
    const size_t chunks = (n + FQZ_CHUNK - 1) / FQZ_CHUNK, groups = (chunks + FQZ_GROUP - 1) / FQZ_GROUP;
    return 24 + (3 + 2 * FQZ_ENT) * chunks + 11 * groups + n + 3 * chunks;
}

extern "C" size_t fqz_encode_bound_blocks(size_t n_bytes, uint32_t rpb)
{
    // pre-entropy bytes <= 2*text + small per-record terms (each record has >= 6 text bytes), every chunk stored raw,
    // per zstd block 6 + 24 bytes (header + index entry + entry points), per 64 KiB group 11; per block of records a 36-byte header and six
    // payloads with a 24-byte index, a short last group and a short last chunk each
    if (!rpb) rpb = FQZ_DEFAULT_BLOCK_SIZE;
    const size_t blocks = n_bytes / 6 / rpb + 2;
    return 2 * n_bytes + (n_bytes / 1024 + 64) * 128 + blocks * (36 + 6 * (24 + 11 + 6 + 2 * FQZ_ENT) + 3 * 4) + n_bytes / 32 /* record samples: 12 bytes per 64 records */ + 4096;
}
extern "C" size_t fqz_encode_bound(size_t n_bytes) { return fqz_encode_bound_blocks(n_bytes, FQZ_DEFAULT_BLOCK_SIZE); }

// ===========================================================================
// device-resident batches
// ===========================================================================
static hipStream_t pick_stream(fqz_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }

extern "C" int fqz_encode_batch_launch(fqz_ctx *ctx, const uint8_t *d_fastq, size_t n_bytes, uint32_t records_per_block, int qual_encoding,
                                       uint32_t flags, uint8_t *d_out, size_t out_cap, void *stream)
{
    if (!ctx || (!d_fastq && n_bytes) || !d_out) return FQZ_E_ARG;
    return fqz_enc_launch(ctx, d_fastq, n_bytes, records_per_block, qual_encoding, flags, d_out, out_cap, pick_stream(ctx, stream));
}

extern "C" int fqz_encode_batch_finish(fqz_ctx *ctx, fqz_batch_result *res, uint64_t *block_off, uint64_t *block_len, size_t max_blocks)
{
    if (!ctx) return FQZ_E_ARG;
    return fqz_enc_finish(ctx, res, block_off, block_len, max_blocks);
}

// A large batch as TWO HALVES IN FLIGHT (VERDICT r2 #2d: "in-batch overlap").  One batch at a time leaves the chip idle at both
// ends of every kernel and in the latency-bound stretches (line index scan, plans, sequence chains, layout): with three batches in
// flight the same pipeline runs 15 % faster (bench.py `pipelined`).  So a batch of FQZ_HALVES_MIN bytes or more is encoded as two:
//   half A = the first n/2 bytes, as a non-final batch: whole 100 000-record blocks only; its front end (line index, record table,
//            plan) reports how many bytes those blocks consume, which encoding block 0 has, and any format error;
//   half B = everything behind that, starting at a block boundary by construction, launched as soon as A's front end has reported
//            (one host round trip, ~0.3 ms into the batch) on its own context and streams; its blocks are laid out behind A's in
//            the same output buffer (k_layout takes A's byte count from device memory).
// Blocks are independent and keep their order: the bytes are those of the one-piece encode (tests/test_gpu_fullsize.py).
// Opt-in (FQZ_BATCH_HALVES, or FQZ_ENC_HALVES=1 in the environment) - measured SLOWER than one piece, see experiments/README.md:
// the host needs ~0.5 ms to queue one half's ~50 launches and event operations, so the second half's tail reaches the device late.
// Anything unusual (lines too short for the single-pass index, capacity relaunches, errors) is
// redone in one piece.
#define FQZ_HALVES_MIN (192ull << 20)
static int encode_batch_halves(fqz_ctx *ctx, const uint8_t *d_fastq, size_t n_bytes, uint32_t rpb, int qual_encoding, uint32_t flags, uint8_t *d_out, size_t out_cap,
                               fqz_batch_result *res, uint64_t *block_off, uint64_t *block_len, size_t max_blocks, hipStream_t caller)
{
    HIP_TRY(hipSetDevice(ctx->device));
    for (int i = 0; i < 2; i++)
        if (!ctx->half[i]) { int rc = fqz_ctx_create(ctx->device, &ctx->half[i]); if (rc) return rc; }
    if (!ctx->ev_half) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_half, hipEventDisableTiming));
    fqz_ctx *A = ctx->half[0], *B = ctx->half[1];
    A->prof.on = B->prof.on = ctx->prof.on; A->prof.dominant_only = B->prof.dominant_only = ctx->prof.dominant_only; A->prof.dominant = B->prof.dominant = ctx->prof.dominant;
    // behind whatever the caller has queued
    HIP_TRY(hipEventRecord(ctx->ev_half, caller));
    HIP_TRY(hipStreamWaitEvent(A->stream, ctx->ev_half, 0));
    HIP_TRY(hipStreamWaitEvent(B->stream, ctx->ev_half, 0));
    const size_t P = (n_bytes / 2) & ~(size_t)15;
    EncLaunchExtra xa;
    xa.want_front = true; xa.want_layout_event = true;
    // A's front end, then (one host round trip) B's, then the long tails of launches: A's, B's
    xa.phase = 1;
    int rc = fqz_enc_launch_ex(A, d_fastq, P, rpb, qual_encoding, flags & ~FQZ_BATCH_FINAL, d_out, out_cap, A->stream, &xa);
    if (rc) return rc;
    xa.phase = 2;
    fqz_batch_result ra, rb;
    memset(&ra, 0, sizeof ra); memset(&rb, 0, sizeof rb);
    if (hipEventSynchronize(A->enc.ev_front) != hipSuccess) return FQZ_E_HIP;
    const EncInfo front = *A->enc.h_front.as<EncInfo>();
    if (front.status) { // a format error in the first half, or a capacity to grow: its own finish says which
        if ((rc = fqz_enc_launch_ex(A, d_fastq, P, rpb, qual_encoding, flags & ~FQZ_BATCH_FINAL, d_out, out_cap, A->stream, &xa))) return rc;
        rc = fqz_enc_finish(A, &ra, nullptr, nullptr, 0);
        if (res) *res = ra;
        return rc ? rc : FQZ_E_HIP;
    }
    const size_t c1 = front.n_blocks ? (size_t)front.consumed : 0, base2 = c1 & ~(size_t)15;
    const int enc2 = (qual_encoding == FQZ_DETECT_ENCODING && front.n_blocks) ? (front.qual_off == 64 ? FQZ_ENCODING_PHRED64 : FQZ_ENCODING_PHRED33) : qual_encoding;
    EncLaunchExtra xb;
    xb.skip = (uint32_t)(c1 & 15);
    xb.prev_info = A->enc.info.as<EncInfo>();
    xb.prev_layout = A->enc.ev_layout;
    xb.phase = 1;
    const int rcb1 = fqz_enc_launch_ex(B, d_fastq + base2, n_bytes - base2, rpb, enc2, flags, d_out, out_cap, B->stream, &xb);
    if ((rc = fqz_enc_launch_ex(A, d_fastq, P, rpb, qual_encoding, flags & ~FQZ_BATCH_FINAL, d_out, out_cap, A->stream, &xa))) return rc;
    xb.phase = 2;
    rc = rcb1 ? rcb1 : fqz_enc_launch_ex(B, d_fastq + base2, n_bytes - base2, rpb, enc2, flags, d_out, out_cap, B->stream, &xb);
    std::vector<uint64_t> boA(block_off ? max_blocks : 0), blA(block_len ? max_blocks : 0), boB(boA.size()), blB(blA.size());
    int rca = fqz_enc_finish(A, &ra, block_off ? boA.data() : nullptr, block_len ? blA.data() : nullptr, max_blocks);
    int rcb = rc ? rc : fqz_enc_finish(B, &rb, block_off ? boB.data() : nullptr, block_len ? blB.data() : nullptr, max_blocks);
    // the children's kernel times count as the parent's
    for (fqz_ctx *h : {A, B}) {
        h->prof.collect();
        for (const ProfTotal &t : h->prof.totals) {
            bool found = false;
            for (ProfTotal &u : ctx->prof.totals) if (u.name == t.name) { u.ms += t.ms; u.calls += t.calls; found = true; break; }
            if (!found) ctx->prof.totals.push_back(t);
        }
        h->prof.totals.clear();
    }
    if (rca || rcb) {
        if (res) { *res = rca ? ra : rb; if (!rca && rb.status && rb.error_record) res->error_record = rb.error_record + ra.n_records; }
        return rca ? rca : rcb;
    }
    if (res) {
        memset(res, 0, sizeof *res);
        res->n_records = ra.n_records + rb.n_records;
        res->n_blocks = ra.n_blocks + rb.n_blocks;
        res->consumed = base2 + rb.consumed;
        res->out_len = ra.out_len + rb.out_len;
        res->qual_encoding = ra.n_blocks ? ra.qual_encoding : rb.qual_encoding;
        res->n_chunks = ra.n_chunks + rb.n_chunks;
        for (int s = 0; s < FQZ_NS; s++) { res->stream_raw[s] = ra.stream_raw[s] + rb.stream_raw[s]; res->stream_comp[s] = ra.stream_comp[s] + rb.stream_comp[s]; }
    }
    if ((block_off || block_len) && (size_t)ra.n_blocks + rb.n_blocks > max_blocks) return FQZ_E_DST_SMALL;
    for (uint32_t b = 0; b < ra.n_blocks; b++) { if (block_off) block_off[b] = boA[b]; if (block_len) block_len[b] = blA[b]; }
    for (uint32_t b = 0; b < rb.n_blocks; b++) { if (block_off) block_off[ra.n_blocks + b] = boB[b]; if (block_len) block_len[ra.n_blocks + b] = blB[b]; }
    return FQZ_OK;
}

extern "C" int fqz_encode_batch_dev(fqz_ctx *ctx, const uint8_t *d_fastq, size_t n_bytes, uint32_t records_per_block, int qual_encoding,
                                    uint32_t flags, uint8_t *d_out, size_t out_cap, fqz_batch_result *res, uint64_t *block_off,
                                    uint64_t *block_len, size_t max_blocks, void *stream)
{
    static const bool halves_env = getenv("FQZ_ENC_HALVES") && atoi(getenv("FQZ_ENC_HALVES"));
    static const bool halves_off = (getenv("FQZ_ENC_SEG") && atoi(getenv("FQZ_ENC_SEG"))) || (getenv("FQZ_DBG_STAMPS") && atoi(getenv("FQZ_DBG_STAMPS")));
    if (ctx && d_fastq && d_out && (halves_env || (flags & FQZ_BATCH_HALVES)) && !halves_off && !ctx->half_off && n_bytes >= FQZ_HALVES_MIN && n_bytes < 0x7FFFFFFFull &&
        !(flags & FQZ_BATCH_SEG) && !ctx->enc.in_flight && !(((uintptr_t)d_fastq & 15) || ((uintptr_t)d_out & 15))) {
        flags &= ~FQZ_BATCH_HALVES;
        const int rc = encode_batch_halves(ctx, d_fastq, n_bytes, records_per_block, qual_encoding, flags, d_out, out_cap, res, block_off, block_len, max_blocks, pick_stream(ctx, stream));
        // FQZ_E_TOO_LARGE: a context wants to resize (very short lines, many headers chunks): such batches are encoded in one piece below
        if (rc != FQZ_E_TOO_LARGE) return rc;
        ctx->half_off = true;
    }
    flags &= ~FQZ_BATCH_HALVES;
    for (int attempt = 0; attempt < 4; attempt++) {
        int rc = fqz_encode_batch_launch(ctx, d_fastq, n_bytes, records_per_block, qual_encoding, flags, d_out, out_cap, stream);
        if (rc) return rc;
        rc = fqz_encode_batch_finish(ctx, res, block_off, block_len, max_blocks);
        // very short lines: a tile-local line slot overflowed (finish() switched the context to the two-pass index) and / or
        // there are more lines than the optimistic line-table capacity (finish() recorded the exact need): run again
        if (rc == FQZ_E_TOO_LARGE && attempt < 3) continue;
        return rc;
    }
    return FQZ_E_TOO_LARGE;
}

extern "C" int fqz_decode_batch_launch(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding,
                                       uint8_t *d_out, size_t out_cap, void *stream)
{
    if (!ctx || (!d_blocks && n_bytes)) return FQZ_E_ARG;
    return fqz_dec_launch(ctx, d_blocks, n_bytes, version, qual_encoding, d_out, out_cap, pick_stream(ctx, stream));
}
extern "C" int fqz_decode_batch_finish(fqz_ctx *ctx, fqz_batch_result *res)
{
    if (!ctx) return FQZ_E_ARG;
    return fqz_dec_finish(ctx, res);
}
extern "C" int fqz_decode_batch_dev(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding,
                                    uint8_t *d_out, size_t out_cap, fqz_batch_result *res, void *stream)
{
    int rc = fqz_decode_batch_launch(ctx, d_blocks, n_bytes, version, qual_encoding, d_out, out_cap, stream);
    if (rc) return rc;
    return fqz_decode_batch_finish(ctx, res);
}

extern "C" int fqz_decode_batch_dev_hint(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding, uint8_t *d_out, size_t out_cap,
                                         fqz_batch_result *res, const uint64_t *block_off, size_t n_blocks, void *stream)
{
    if (!ctx) return FQZ_E_ARG;
    ctx->dec.hint_off = block_off; ctx->dec.hint_n = block_off ? n_blocks : 0;
    const int rc = fqz_decode_batch_dev(ctx, d_blocks, n_bytes, version, qual_encoding, d_out, out_cap, res, stream);
    ctx->dec.hint_off = nullptr; ctx->dec.hint_n = 0;
    return rc;
}

// Diagnostic only (FQZ_DBG_STAMPS=1): s_memtime stamps of k_entropy's phases, 16 x u64 per chunk.
extern "C" int fqz_debug_get_stamps(fqz_ctx *ctx, unsigned long long *out, size_t max_chunks, size_t *n_chunks)
{
    if (!ctx || !out || !n_chunks) return FQZ_E_ARG;
    return fqz_enc_get_stamps(ctx, out, max_chunks, n_chunks);
}

extern "C" int fqz_debug_get_streams(fqz_ctx *ctx, uint32_t block, uint8_t *streams[6], size_t stream_len[6])
{
    if (!ctx || !stream_len) return FQZ_E_ARG;
    return fqz_enc_get_streams(ctx, block, streams, stream_len);
}

// ===========================================================================
// host-buffer entry points: H2D -> device pipeline -> D2H
// ===========================================================================
static int stage_in(fqz_ctx *ctx, const uint8_t *src, size_t n)
{
    int rc = ctx->d_in.ensure(n + 64);
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(ctx->d_in.p, src, n, hipMemcpyHostToDevice, ctx->stream));
    return FQZ_OK;
}

// Encodes `n` bytes of FASTQ (whole records) as blocks of rpb records into ctx->d_out.
static int encode_staged(fqz_ctx *ctx, const uint8_t *fastq, size_t n, uint32_t rpb, int qual_encoding, uint32_t flags,
                         fqz_batch_result *res, std::vector<uint64_t> *offs, std::vector<uint64_t> *lens)
{
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_in(ctx, fastq, n);
    if (rc) return rc;
    size_t cap = fqz_encode_bound_blocks(n, rpb);
    if ((rc = ctx->d_out.ensure(cap + 64))) return rc;
    size_t max_blocks = n / 4 / (rpb ? rpb : 1) + 2;
    if (offs) { offs->assign(max_blocks, 0); lens->assign(max_blocks, 0); }
    return fqz_encode_batch_dev(ctx, ctx->d_in.as<uint8_t>(), n, rpb, qual_encoding, flags, ctx->d_out.as<uint8_t>(), cap, res,
                                offs ? offs->data() : nullptr, lens ? lens->data() : nullptr, max_blocks, ctx->stream);
}

extern "C" int fqz_encode_block(fqz_ctx *ctx, const uint8_t *fastq, size_t n_bytes, int qual_encoding, uint8_t *out, size_t out_cap,
                                size_t *out_len, uint32_t *n_records)
{
    if (!ctx || (!fastq && n_bytes) || !out || !out_len) return FQZ_E_ARG;
    *out_len = 0;
    fqz_batch_result res;
    // one block, whatever its record count: rpb = max so that everything lands in block 0
    int rc = encode_staged(ctx, fastq, n_bytes, 0x7FFFFFFFu, qual_encoding, FQZ_BATCH_FINAL, &res, nullptr, nullptr);
    if (rc) return rc;
    if (n_records) *n_records = res.n_records;
    if (res.out_len > out_cap) return FQZ_E_DST_SMALL;
    if (res.out_len) HIP_TRY(hipMemcpy(out, ctx->d_out.p, res.out_len, hipMemcpyDeviceToHost));
    *out_len = res.out_len;
    return FQZ_OK;
}

static int decode_staged(fqz_ctx *ctx, const uint8_t *blocks, size_t n, uint8_t version, int qual_encoding, uint8_t *out, size_t out_cap,
                         size_t *out_len, bool size_only)
{
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_in(ctx, blocks, n);
    if (rc) return rc;
    // FASTQ text is bounded by the pre-entropy sizes announced in the frames; the decoder sizes its own
    // staging buffer and reports the exact length
    fqz_batch_result res;
    ctx->dec.skip_assemble = size_only; // the size is known once the lengths / headers / plus streams are decoded and summed: no text is assembled
    rc = fqz_dec_launch(ctx, ctx->d_in.as<uint8_t>(), n, version, qual_encoding, nullptr, 0, ctx->stream);
    if (!rc) rc = fqz_dec_finish(ctx, &res);
    ctx->dec.skip_assemble = false;
    if (rc) return rc;
    *out_len = res.out_len;
    if (size_only) return FQZ_OK;
    if (res.out_len > out_cap) return FQZ_E_DST_SMALL;
    if (res.out_len) HIP_TRY(hipMemcpy(out, ctx->d_out.p, res.out_len, hipMemcpyDeviceToHost));
    return FQZ_OK;
}

extern "C" int fqz_decode_block(fqz_ctx *ctx, const uint8_t *block, size_t n, uint8_t version, int qual_encoding, uint8_t *out,
                                size_t out_cap, size_t *out_len)
{
    if (!ctx || !block || !out_len) return FQZ_E_ARG;
    return decode_staged(ctx, block, n, version, qual_encoding, out, out_cap, out_len, false);
}

extern "C" int fqz_decode_block_size(fqz_ctx *ctx, const uint8_t *block, size_t n, uint8_t version, size_t *out_len)
{
    if (!ctx || !block || !out_len) return FQZ_E_ARG;
    return decode_staged(ctx, block, n, version, FQZ_ENCODING_PHRED33, nullptr, 0, out_len, true);
}

// ===========================================================================
// internal/encoder primitive mirrors: small kernels built from the same device helpers
// ===========================================================================
__global__ __launch_bounds__(256) void k_prim_pack(const uint8_t *seq, uint32_t n, uint8_t *packed, uint16_t *npos, uint32_t *n_npos)
{
    // sequence.go:139-184; single workgroup, positions kept ascending through a running ballot offset
    __shared__ uint32_t sh[4];
    __shared__ uint32_t run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    uint32_t nq = (n + 3) >> 2;
    for (uint32_t base = 0; base < nq; base += 256) {
        uint32_t i = base + threadIdx.x, inv = 0, cnt = 0;
        if (i < nq) {
            uint32_t have = n - 4 * i < 4 ? n - 4 * i : 4, x = 0;
            for (uint32_t j = 0; j < have; j++) x |= (uint32_t)seq[4 * i + j] << (8 * j);
            uint32_t valid = acgt_mask(x);
            uint32_t in_read = have < 4 ? 0x80808080u >> (8 * (4 - have)) : 0x80808080u;
            packed[i] = (uint8_t)pack4(x, valid);
            inv = ~valid & in_read;
            for (uint32_t j = 0; j < 4; j++) if ((inv & (0x80u << (8 * j))) && 4 * i + j < FQZ_MAX_SEQUENCE_LENGTH) cnt++;
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan_256(cnt, sh, &tot) + run;
        for (uint32_t j = 0; j < 4; j++)
            if ((inv & (0x80u << (8 * j))) && 4 * i + j < FQZ_MAX_SEQUENCE_LENGTH) npos[ex++] = (uint16_t)(4 * i + j);
        __syncthreads();
        if (threadIdx.x == 0) run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_npos = run;
}

__global__ __launch_bounds__(256) void k_prim_unpack(const uint8_t *packed, const uint16_t *npos, uint32_t n_npos, uint32_t seq_len,
                                                     uint8_t *seq, int *err)
{
    // sequence.go:188-223
    for (uint32_t i = threadIdx.x; i < seq_len; i += 256) {
        uint32_t c = (packed[i >> 2] >> ((i & 3) << 1)) & 3;
        seq[i] = (uint8_t)((0x54474341u >> (8 * c)) & 0xFF); // "ACGT"
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_npos; k += 256) {
        if (npos[k] >= seq_len) *err = 1; else seq[npos[k]] = 'N';
    }
}

__global__ __launch_bounds__(256) void k_prim_add(uint8_t *q, uint32_t n, uint32_t delta)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) q[i] = (uint8_t)(q[i] + delta);
}

__global__ __launch_bounds__(256) void k_prim_delta_enc(const uint8_t *in, uint8_t *out, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = (uint8_t)(in[i] - (i ? in[i - 1] : 0));
}

__global__ __launch_bounds__(256) void k_prim_delta_dec(uint8_t *q, uint32_t n)
{
    // quality.go:107-118: inclusive prefix sum mod 256; single workgroup, carry across strips
    __shared__ uint32_t sh[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n ? q[i] : 0, tot;
        uint32_t ex = block_excl_scan_256(v, sh, &tot);
        if (i < n) q[i] = (uint8_t)(carry + ex + v);
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_prim_min(const uint8_t *q, uint32_t n, uint32_t *mn)
{
    uint32_t m = 255;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = q[i] < m ? q[i] : m;
    m = wave_min(m);
    if ((threadIdx.x & 63) == 0) atomicMin(mn, m);
}

struct Scratch { // small RAII device scratch for the primitive calls
    void *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess ? 0 : FQZ_E_NOMEM; }
};

extern "C" int fqz_pack_bases(fqz_ctx *ctx, const uint8_t *seq, size_t n, uint8_t *packed, uint16_t *npos, size_t *n_npos)
{
    if (!ctx || (!seq && n) || n > 0x7FFFFFFFu || !n_npos) return FQZ_E_ARG;
    *n_npos = 0;
    if (!n) return FQZ_OK; // sequence.go:141-143
    HIP_TRY(hipSetDevice(ctx->device));
    size_t pl = (n + 3) / 4, lim = n < FQZ_MAX_SEQUENCE_LENGTH ? n : FQZ_MAX_SEQUENCE_LENGTH;
    Scratch s;
    size_t o_p = fqz_align_up(n, 16), o_n = o_p + fqz_align_up(pl, 16), o_c = o_n + fqz_align_up(2 * lim, 16);
    if (s.alloc(o_c + 16)) return FQZ_E_NOMEM;
    uint8_t *d = (uint8_t *)s.p;
    HIP_TRY(hipMemcpyAsync(d, seq, n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_prim_pack, dim3(1), dim3(256), 0, ctx->stream, d, (uint32_t)n, d + o_p, (uint16_t *)(d + o_n), (uint32_t *)(d + o_c));
    uint32_t cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, d + o_c, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(packed, d + o_p, pl, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (cnt && npos) HIP_TRY(hipMemcpy(npos, d + o_n, 2ull * cnt, hipMemcpyDeviceToHost));
    *n_npos = cnt;
    return FQZ_OK;
}

extern "C" int fqz_unpack_bases(fqz_ctx *ctx, const uint8_t *packed, const uint16_t *npos, size_t n_npos, size_t seq_len, uint8_t *seq)
{
    if (!ctx || seq_len > 0x7FFFFFFFu || (seq_len && (!packed || !seq)) || (n_npos && !npos)) return FQZ_E_ARG;
    if (!seq_len) return FQZ_OK; // sequence.go:189-191
    HIP_TRY(hipSetDevice(ctx->device));
    size_t pl = (seq_len + 3) / 4;
    Scratch s;
    size_t o_n = fqz_align_up(pl, 16), o_s = o_n + fqz_align_up(2 * n_npos, 16), o_e = o_s + fqz_align_up(seq_len, 16);
    if (s.alloc(o_e + 16)) return FQZ_E_NOMEM;
    uint8_t *d = (uint8_t *)s.p;
    HIP_TRY(hipMemcpyAsync(d, packed, pl, hipMemcpyHostToDevice, ctx->stream));
    if (n_npos) HIP_TRY(hipMemcpyAsync(d + o_n, npos, 2 * n_npos, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d + o_e, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_prim_unpack, dim3(1), dim3(256), 0, ctx->stream, d, (const uint16_t *)(d + o_n), (uint32_t)n_npos, (uint32_t)seq_len,
                       d + o_s, (int *)(d + o_e));
    int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, d + o_e, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(seq, d + o_s, seq_len, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return err ? FQZ_E_NPOS_RANGE : FQZ_OK;
}

extern "C" int fqz_detect_encoding(fqz_ctx *ctx, const uint8_t *quals, const uint64_t *offsets, size_t n, int *encoding)
{
    if (!ctx || !encoding || (n && !offsets)) return FQZ_E_ARG;
    *encoding = FQZ_ENCODING_PHRED33;
    size_t total = n ? (size_t)(offsets[n] - offsets[0]) : 0;
    if (!total) return FQZ_OK; // quality.go:36-39
    if (total > 0x7FFFFFFFu) return FQZ_E_TOO_LARGE;
    HIP_TRY(hipSetDevice(ctx->device));
    Scratch s;
    size_t o_m = fqz_align_up(total, 16);
    if (s.alloc(o_m + 16)) return FQZ_E_NOMEM;
    uint8_t *d = (uint8_t *)s.p;
    HIP_TRY(hipMemcpyAsync(d, quals + offsets[0], total, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d + o_m, 0xFF, 4, ctx->stream));
    uint32_t grid = (uint32_t)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_prim_min, dim3(grid), dim3(256), 0, ctx->stream, d, (uint32_t)total, (uint32_t *)(d + o_m));
    uint32_t mn = 255;
    HIP_TRY(hipMemcpyAsync(&mn, d + o_m, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    mn &= 0xFF; // quality.go:43-48
    *encoding = (mn >= 64 && mn != 255) ? FQZ_ENCODING_PHRED64 : FQZ_ENCODING_PHRED33;
    return FQZ_OK;
}

static int prim_inplace(fqz_ctx *ctx, uint8_t *q, size_t n, int which, uint32_t arg)
{
    if (!ctx || (!q && n) || n > 0x7FFFFFFFu) return FQZ_E_ARG;
    if (!n) return FQZ_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    Scratch s;
    size_t o_b = fqz_align_up(n, 16);
    if (s.alloc(2 * o_b + 16)) return FQZ_E_NOMEM;
    uint8_t *d = (uint8_t *)s.p;
    HIP_TRY(hipMemcpyAsync(d, q, n, hipMemcpyHostToDevice, ctx->stream));
    uint32_t grid = (uint32_t)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    uint8_t *res = d;
    if (which == 0) hipLaunchKernelGGL(k_prim_add, dim3(grid), dim3(256), 0, ctx->stream, d, (uint32_t)n, arg);
    else if (which == 1) { hipLaunchKernelGGL(k_prim_delta_enc, dim3(grid), dim3(256), 0, ctx->stream, d, d + o_b, (uint32_t)n); res = d + o_b; }
    else hipLaunchKernelGGL(k_prim_delta_dec, dim3(1), dim3(256), 0, ctx->stream, d, (uint32_t)n);
    HIP_TRY(hipMemcpyAsync(q, res, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FQZ_OK;
}

extern "C" int fqz_normalize_quality(fqz_ctx *ctx, uint8_t *q, size_t n, int enc)
{
    return prim_inplace(ctx, q, n, 0, 256u - (enc == FQZ_ENCODING_PHRED64 ? 64u : 33u));
}
extern "C" int fqz_denormalize_quality(fqz_ctx *ctx, uint8_t *q, size_t n, int enc)
{
    return prim_inplace(ctx, q, n, 0, enc == FQZ_ENCODING_PHRED64 ? 64u : 33u);
}
extern "C" int fqz_delta_encode(fqz_ctx *ctx, uint8_t *q, size_t n) { return prim_inplace(ctx, q, n, 1, 0); }
extern "C" int fqz_delta_decode(fqz_ctx *ctx, uint8_t *q, size_t n) { return prim_inplace(ctx, q, n, 2, 0); }

// ===========================================================================
// entropy stage alone
// ===========================================================================
int fqz_enc_entropy_only(fqz_ctx *ctx, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap, size_t *payload_off, size_t *out_len, hipStream_t st);
int fqz_dec_entropy_only(fqz_ctx *ctx, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap, size_t *out_len, hipStream_t st);

extern "C" int fqz_entropy_encode(fqz_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len)
{
    if (!ctx || (!src && n) || !out_len) return FQZ_E_ARG;
    *out_len = 0;
    if (!n) return FQZ_OK; // empty stream -> 0-byte payload
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_in(ctx, src, n);
    if (rc) return rc;
    size_t bound = fqz_entropy_bound(n) + 64;
    if ((rc = ctx->d_out.ensure(bound + 64))) return rc;
    size_t got = 0, off = 0;
    rc = fqz_enc_entropy_only(ctx, ctx->d_in.as<uint8_t>(), n, ctx->d_out.as<uint8_t>(), bound, &off, &got, ctx->stream);
    if (rc) return rc;
    if (got > cap) return FQZ_E_DST_SMALL;
    HIP_TRY(hipMemcpy(dst, ctx->d_out.as<uint8_t>() + off, got, hipMemcpyDeviceToHost));
    *out_len = got;
    return FQZ_OK;
}

extern "C" int fqz_entropy_decode(fqz_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len)
{
    if (!ctx || (!src && n) || !out_len) return FQZ_E_ARG;
    *out_len = 0;
    if (!n) return FQZ_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_in(ctx, src, n);
    if (rc) return rc;
    if ((rc = ctx->d_out.ensure(cap + 64))) return rc;
    size_t got = 0;
    rc = fqz_dec_entropy_only(ctx, ctx->d_in.as<uint8_t>(), n, ctx->d_out.as<uint8_t>(), cap, &got, ctx->stream);
    if (rc) return rc;
    if (got) HIP_TRY(hipMemcpy(dst, ctx->d_out.p, got, hipMemcpyDeviceToHost));
    *out_len = got;
    return FQZ_OK;
}

// ===========================================================================
// per-kernel timing
// ===========================================================================
extern "C" int fqz_profile_enable(fqz_ctx *ctx, int on)
{
    if (!ctx) return FQZ_E_ARG;
    ctx->prof.on = on != 0;
    ctx->prof.dominant_only = on == 2 || on == 3; // 2: k_entropy only, 3: k_rans only (container version 3)
    ctx->prof.dominant = on == 3 ? "k_rans" : "k_entropy";
    return FQZ_OK;
}
extern "C" int fqz_profile_reset(fqz_ctx *ctx)
{
    if (!ctx) return FQZ_E_ARG;
    ctx->prof.collect();
    ctx->prof.totals.clear();
    return FQZ_OK;
}
extern "C" int fqz_profile_read(fqz_ctx *ctx, char *names, size_t names_cap, double *ms, uint32_t *calls, size_t max_entries, size_t *n_entries)
{
    if (!ctx || !n_entries) return FQZ_E_ARG;
    ctx->prof.collect();
    std::string all;
    size_t n = 0;
    for (const ProfTotal &t : ctx->prof.totals) {
        if (n >= max_entries) break;
        if (n) all += "\n";
        all += t.name;
        if (ms) ms[n] = t.ms;
        if (calls) calls[n] = t.calls;
        n++;
    }
    if (names && names_cap) { size_t k = all.size() < names_cap - 1 ? all.size() : names_cap - 1; memcpy(names, all.data(), k); names[k] = 0; }
    *n_entries = n;
    return FQZ_OK;
}

// ===========================================================================
// synthetic FASTQ — host generator for bench / tests
// ===========================================================================
static inline uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Xoshiro {
    uint64_t s[4];
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    void seed(uint64_t v) { for (int i = 0; i < 4; i++) s[i] = splitmix64(v); }
    inline uint64_t next()
    {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    inline uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

static inline uint8_t *put_dec(uint8_t *p, uint64_t v)
{
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = (uint8_t)tmp[--k];
    return p;
}

extern "C" int fqz_synth_fastq(const fqz_synth_params *pp, uint64_t n_records, uint8_t *out, size_t cap, size_t *out_len, uint64_t *n_written)
{
    if (!pp || !out || !out_len) return FQZ_E_ARG;
    fqz_synth_params p = *pp;
    if (!p.min_len && !p.max_len) p.min_len = p.max_len = 150;
    if (p.max_len < p.min_len) return FQZ_E_ARG;
    if (p.phred != 33 && p.phred != 64) p.phred = 33;
    const size_t worst = 2ull * p.max_len + 160;
    uint8_t *w = out, *end = out + cap;
    uint64_t r = 0;
    static const uint8_t lv4[4] = {37, 25, 11, 2}; // F : , #  (Phred+33)
    for (; r < n_records; r++) {
        if ((size_t)(end - w) < worst) break;
        uint64_t i = p.first_record + r;
        Xoshiro g;
        g.seed(p.seed * 0x9E3779B97F4A7C15ull + i); // per-record stream: shards generate identical records independently
        uint32_t L = p.min_len + (p.max_len > p.min_len ? g.below(p.max_len - p.min_len + 1) : 0);
        *w++ = '@';
        memcpy(w, "SIM:1:FCX123:1:", 15); w += 15;
        w = put_dec(w, 1101 + i / 200000); *w++ = ':';
        w = put_dec(w, 1000 + (7919 * i) % 20000); *w++ = ':';
        w = put_dec(w, 1000 + (104729 * i) % 20000);
        memcpy(w, " 1:N:0:ATCACG", 13); w += 13;
        if (p.max_len != p.min_len) { memcpy(w, " length=", 8); w += 8; w = put_dec(w, L); }
        *w++ = '\n';
        // bases: uniform ACGT, N in geometric runs (mean 3) with overall fraction n_permille/1000
        uint32_t nrun = 0;
        for (uint32_t j = 0; j < L;) {
            uint64_t x = g.next();
            for (int k = 0; k < 16 && j < L; k++, j++, x >>= 4) {
                uint8_t b = "ACGT"[x & 3];
                if (p.n_permille) { // a run of 1..5 N (mean 3) starts with probability permille/3000
                    if (nrun) { b = 'N'; nrun--; }
                    else if (g.below(3000) < p.n_permille) { b = 'N'; nrun = g.below(5); }
                }
                *w++ = b;
            }
        }
        *w++ = '\n'; *w++ = '+'; *w++ = '\n';
        if (p.quality_profile == 0) {
            // 4-level binned Markov chain: stay 0.93; from F: ':' 0.05, ',' 0.015, '#' 0.005; back to F with 0.5
            uint32_t cur = 0;
            for (uint32_t j = 0; j < L; j++) {
                uint32_t u = g.below(1000);
                if (u >= 930) {
                    if (cur == 0) cur = u < 980 ? 1 : (u < 995 ? 2 : 3);
                    else cur = (u & 1) ? 0 : 1 + g.below(3);
                }
                *w++ = (uint8_t)(lv4[cur] + p.phred);
            }
        } else {
            // 41-level HiSeq-like random walk: start 34 +- 4, steps -2..+1 weighted to slow decay, floor 2, cap 40
            int q = 30 + (int)g.below(9);
            for (uint32_t j = 0; j < L; j++) {
                uint32_t u = g.below(100);
                q += u < 8 ? -2 : (u < 30 ? -1 : (u < 80 ? 0 : 1));
                if (q < 2) q = 2;
                if (q > 40) q = 40;
                *w++ = (uint8_t)(q + (int)p.phred);
            }
        }
        *w++ = '\n';
    }
    *out_len = (size_t)(w - out);
    if (n_written) *n_written = r;
    return FQZ_OK;
}
