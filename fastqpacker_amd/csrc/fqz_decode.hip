// fqz_decode.hip — device-resident decode pipeline (gfx950 / MI355X).
//
// Replaces, for a run of blocks at once, the reference's
//   ReadBlockHeader + readCompressedStreams   container.go:116-152, compress.go:721-758   (k_dec_blocks)
//   6 x zstd.Decoder.DecodeAll                compress.go:785-814                         (k_dec_frames, k_dec_chunks, k_dec_entropy)
//   blockReader.writeRecord and helpers       compress.go:944-1078                        (k_dec_walk, k_dec_sizes, k_dec_assemble)
//     encoder.AppendUnpackBases               sequence.go:188-223
//     encoder.DeltaDecode + DenormalizeQuality quality.go:66-75, 107-118
// The zstd subset understood here is what libfqzhip (and the oracle) emit: Raw / RLE / Compressed
// blocks whose Compressed blocks carry Huffman, raw or RLE literals and zero sequences.  Frames
// with LZ sequences (files written by the stock encoder) are reported as FQZ_E_ENTROPY.
#include "fqz_ctx.h"
#include "fqz_device.h"
#include "fqz_xxh.h"

#include <stdlib.h>
#include <string.h>
#include <vector>

struct DecBlock {
    uint32_t nrec;
    uint32_t rec_base;              // first record of the block in the batch
    uint32_t pay_off[FQZ_NS];       // payload offsets in d_in
    uint32_t pay_len[FQZ_NS];
    uint32_t raw_len[FQZ_NS];       // pre-entropy bytes (from the frames)
    uint32_t n_chunks[FQZ_NS];
    uint32_t chunk_base[FQZ_NS];
    uint32_t a_off[FQZ_NS];         // arena offset of the decoded stream (16-aligned)
    uint32_t tile_base[3];          // first walk tile of the headers / plus / nPos stream
    uint32_t walk_mode[3];          // 1 = offsets already written by k_dec_walk0
    uint32_t walk_cand[3];          // entry offsets k_dec_walk1 precomputes per tile (64 / 128 / 256, by average record length)
    uint32_t lz[FQZ_NS];            // 0, or 1 + scratch slot: the payload is a foreign frame with LZ sequences (k_dec_lz)
    uint32_t indexed[FQZ_NS];       // the payload starts with our index frame (FQZ-H2): its zstd blocks are located without a walk
    uint32_t n_frames[FQZ_NS];      // zstd frames of the payload (content checksums are verified per frame)
    uint32_t frame_base[FQZ_NS];    // first entry of the payload in the frame table
    uint32_t samp_off[FQZ_NS];      // offset in d_in of the record samples of the payload's index (stream offsets of records 64, 128, ...); 0: none
    uint32_t ent_off[FQZ_NS];       // offset in d_in of the entry points of the payload's index (FQZ_ENT u16 a zstd block); 0: none
    uint32_t seq_scratch;           // arena offset of the scratch of the headers stream's blocks with sequences (DSEQ_STRIDE each; 0: none)
    uint32_t seq_scratch_len;       // the same for the lengths stream (a chunk of equal read lengths is one match)
};

struct DecFrame {
    uint32_t dst_off;   // arena offset of the frame's content
    uint32_t len;       // content bytes
    uint32_t ck_off;    // offset in d_in of the 4-byte Content_Checksum (0: the frame carries none)
    uint32_t stream;    // S_SEQ .. S_LEN (the verification launches take a stream mask)
};

struct DecChunk {
    uint32_t src_off;   // offset of the zstd block CONTENT in d_in
    uint32_t csize;     // content bytes
    uint32_t dst_off;   // arena offset
    uint32_t regen;     // bytes produced
    uint32_t btype;     // 0 raw, 1 RLE, 2 compressed
    uint32_t tree_off;  // treeless literals: offset in d_in of the Huffman tree description of the group's first block ...
    uint32_t tree_len;  // ... and the bytes available there (0 = the block carries its own tree)
    uint32_t stream;    // S_SEQ .. S_LEN: which of the block's six payloads (the launches take a stream mask)
    uint32_t seq_len;   // 0, or the bytes of the block's Sequences_Section (fqz_decode_seq.h): dst_off / regen then describe the
    uint32_t out_off;   // LITERALS (decoded into a scratch area) and out_off / out_len the block's place in the stream
    uint32_t out_len;
    uint32_t ent_off;   // 0, or the offset in d_in of the block's entry points (FQZI index: three per Huffman stream)
};

#define DSEQ_MAX 2048u                                   // sequences per block the fast path takes
#define DSEQ_STRIDE (2u * FQZ_CHUNK + 64u)               // scratch per block: [literals 16 KiB | triples 16 KiB | nseq u32 ...]
#define DSEQ_INVALID 0xFFFFFFFFu

__device__ __forceinline__ uint32_t rd32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t rd24(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
__device__ __forceinline__ uint32_t rd16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }

__device__ __forceinline__ void dec_fail(DecInfo *info, int code) { atomicCAS(&info->status, 0, code); }

// ---------------------------------------------------------------------------
// block headers (container.go:116-152) — one thread walks the chain
// ---------------------------------------------------------------------------
__device__ void dec_blocks_walk(const uint8_t *in, uint32_t n, uint32_t version, DecInfo *info, DecBlock *blocks, uint32_t block_cap)
{
    uint32_t pos = 0, nb = 0, nrec = 0;
    const uint32_t hs = version == FQZ_VERSION1 ? 32 : 36;
    while (pos < n) {
        if (n - pos < hs) { dec_fail(info, FQZ_E_SHORT); break; }
        const uint8_t *h = in + pos;
        uint32_t sz[FQZ_NS];
        // (a hop of the chain is one memory round trip: the seven words it needs are requested together, as unaligned dwords)
        const uint32_t records = load_u32_unaligned(h), f1 = load_u32_unaligned(h + 4), f2 = load_u32_unaligned(h + 8), f3 = load_u32_unaligned(h + 12);
        const uint32_t f4 = load_u32_unaligned(h + 16), f5 = load_u32_unaligned(h + 20), f6 = load_u32_unaligned(h + 24);
        if (version == FQZ_VERSION3 && records == FQZ_BLOCK_TABLE_MARK && f1 == 0x585A5146u) break; // 'FQZX': the block table, the chain ends here
        sz[S_SEQ] = f1; sz[S_QUAL] = f2; sz[S_HDR] = f3;
        if (version == FQZ_VERSION1) { sz[S_PLUS] = 0; sz[S_NPOS] = f4; sz[S_LEN] = f5; }
        else { sz[S_PLUS] = f4; sz[S_NPOS] = f5; sz[S_LEN] = f6; }
        pos += hs;
        // wire order: seq, qual, headers, [plus], nPos, lengths (compress.go:738-751)
        const int order[FQZ_NS] = {S_SEQ, S_QUAL, S_HDR, S_PLUS, S_NPOS, S_LEN};
        bool ok = true;
        uint32_t off[FQZ_NS];
        for (int q = 0; q < FQZ_NS; q++) {
            int s = order[q];
            if (sz[s] > n - pos) { ok = false; break; }
            off[s] = pos;
            pos += sz[s];
        }
        if (!ok) { dec_fail(info, FQZ_E_READ_DATA); break; }
        if (blocks && nb < block_cap) {
            DecBlock *b = &blocks[nb];
            b->nrec = records;
            b->rec_base = nrec;
            for (int s = 0; s < FQZ_NS; s++) { b->pay_off[s] = off[s]; b->pay_len[s] = sz[s]; }
        }
        if ((unsigned long long)nrec + records > 0x7FFFFFFFull) { dec_fail(info, FQZ_E_TOO_LARGE); break; }
        nrec += records;
        nb++;
    }
    info->n_blocks = nb;
    info->n_rec = nrec;
    info->n_groups = (nrec + 63) / 64;
}
__global__ void k_dec_blocks(const uint8_t *in, uint32_t n, uint32_t version, DecInfo *info, DecBlock *blocks, uint32_t block_cap)
{
    if (threadIdx.x || blockIdx.x) return;
    dec_blocks_walk(in, n, version, info, blocks, block_cap);
}

// The same table from a HINT: the caller says where the block headers are (fqz_decode_batch_dev_hint - whoever cut the batch
// out of a file has walked them already, as the reference's reader does, compress.go:721-758).  A thread per block reads its
// header and checks that the block ends where the next one starts (the last one at the end of the input, or at a version-3
// block table); if everything adds up the chain - one dependent memory round trip per block, 1.6 us each - is not walked at all,
// otherwise it is, as if there had been no hint.  One 256-thread workgroup.
__global__ __launch_bounds__(256) void k_dec_blocks_hint(const uint8_t *in, uint32_t n, uint32_t version, DecInfo *info, DecBlock *blocks, uint32_t block_cap,
                                                         const unsigned long long *hint, uint32_t n_hint)
{
    __shared__ uint32_t sh[4];
    const uint32_t t = threadIdx.x, hs = version == FQZ_VERSION1 ? 32 : 36;
    bool bad = n_hint == 0 || n_hint > block_cap;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_hint && !bad; b0 += 256) {
        const uint32_t b = b0 + t;
        uint32_t records = 0;
        if (b < n_hint) {
            const unsigned long long at = hint[b];
            if ((b == 0 && at != 0) || at > n || n - at < hs) bad = true;
            else {
                const uint8_t *h = in + at;
                uint32_t sz[FQZ_NS];
                records = load_u32_unaligned(h);
                sz[S_SEQ] = load_u32_unaligned(h + 4); sz[S_QUAL] = load_u32_unaligned(h + 8); sz[S_HDR] = load_u32_unaligned(h + 12);
                if (version == FQZ_VERSION1) { sz[S_PLUS] = 0; sz[S_NPOS] = load_u32_unaligned(h + 16); sz[S_LEN] = load_u32_unaligned(h + 20); }
                else { sz[S_PLUS] = load_u32_unaligned(h + 16); sz[S_NPOS] = load_u32_unaligned(h + 20); sz[S_LEN] = load_u32_unaligned(h + 24); }
                if (version == FQZ_VERSION3 && records == FQZ_BLOCK_TABLE_MARK) bad = true; // (the table is not a block)
                unsigned long long pos = at + hs;
                const int order[FQZ_NS] = {S_SEQ, S_QUAL, S_HDR, S_PLUS, S_NPOS, S_LEN};
                DecBlock *k = &blocks[b];
                for (int q = 0; q < FQZ_NS; q++) { const int s = order[q]; k->pay_off[s] = (uint32_t)pos; k->pay_len[s] = sz[s]; pos += sz[s]; }
                k->nrec = records;
                if (pos > n) bad = true;
                else if (b + 1 < n_hint) { if (hint[b + 1] != pos) bad = true; }
                else if (pos != n) { // behind the last block: nothing, or the block table of a version-3 file
                    if (!(version == FQZ_VERSION3 && n - pos >= 8 && load_u32_unaligned(in + pos) == FQZ_BLOCK_TABLE_MARK && load_u32_unaligned(in + pos + 4) == 0x585A5146u)) bad = true;
                }
            }
        }
        bad = __syncthreads_or(bad) != 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(records, sh, &tot);
        if (b < n_hint && !bad) blocks[b].rec_base = carry + ex;
        if ((unsigned long long)carry + tot > 0x7FFFFFFFull) bad = true; // (the walk names the error)
        carry += tot;
        __syncthreads();
    }
    bad = __syncthreads_or(bad) != 0;
    if (t) return;
    if (bad) { dec_blocks_walk(in, n, version, info, blocks, block_cap); return; } // (overwrites what the hint wrote)
    info->n_blocks = n_hint;
    info->n_rec = carry;
    info->n_groups = (carry + 63) / 64;
}

// ---------------------------------------------------------------------------
// zstd frame / block walk (RFC 8878 3.1.1): one thread per (block, stream)
// pass 0: count chunks + regenerated bytes; pass 1: write DecChunk entries; pass 2: write + verify the speculative sizes
// ---------------------------------------------------------------------------
__device__ int frame_header(const uint8_t *p, uint32_t n, uint32_t *hdr, long long *fcs, int *checksum)
{
    if (n < 6) return -1;
    if (!(p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD)) return -1;
    uint32_t fhd = p[4];
    int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
    if (fhd & 0x08) return -1;
    *checksum = (fhd >> 2) & 1;
    uint32_t q = 5;
    if (!single) q += 1;
    q += dict == 0 ? 0 : (dict == 1 ? 1 : (dict == 2 ? 2 : 4));
    int fs = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (q + fs > n) return -1;
    *fcs = -1;
    if (fs) {
        unsigned long long v = 0;
        for (int i = 0; i < fs; i++) v |= (unsigned long long)p[q + i] << (8 * i);
        if (fs == 2) v += 256;
        *fcs = (long long)v;
    }
    *hdr = q + fs;
    return 0;
}

// zstd skippable frame (RFC 8878 3.1.2): magic 0x184D2A5? + u32 size; decoders skip it
__device__ __forceinline__ bool skippable_magic(const uint8_t *p) { return (p[0] & 0xF0) == 0x50 && p[1] == 0x2A && p[2] == 0x4D && p[3] == 0x18; }
// FQZ-H2 index in front of a payload: 'FQZI', version 1, stream, pre-entropy length, zstd block count, 3 bytes per block
#define H2_IDX_HDR 24u
// flags bit 0: behind the block sizes, [records u32][stream offset u32 of records 64, 128, ...] (*samples = offset of the first
// one inside the payload, *srec = the record count it was made for)
// flags bit 1: behind those, FQZ_ENT u16 a block: where a decoder may enter the block's Huffman streams (*ent = offset inside the payload)
__device__ __forceinline__ bool h2_index(const uint8_t *p, uint32_t n, uint32_t *raw, uint32_t *nch, uint32_t *samples, uint32_t *srec, uint32_t *ent)
{
    if (n < H2_IDX_HDR || !skippable_magic(p) || p[0] != 0x50) return false;
    if (!(p[8] == 'F' && p[9] == 'Q' && p[10] == 'Z' && p[11] == 'I' && p[12] == 1)) return false;
    const uint32_t sz = rd32(p + 4), r = rd32(p + 16), c = rd32(p + 20), flags = p[14] | ((uint32_t)p[15] << 8);
    if (r == 0 || r > 0x7FFFFFF0u || c != (r + FQZ_CHUNK - 1) / FQZ_CHUNK || (flags & ~3u)) return false;
    unsigned long long want = H2_IDX_HDR - 8 + 3ull * c;
    *samples = 0; *srec = 0; *ent = 0;
    if (flags & 1) {
        if (8ull + want + 4 > n) return false;
        const uint32_t nr = rd32(p + H2_IDX_HDR + 3 * c);
        if (nr <= 64) return false;
        want += 4 + 4ull * ((nr - 1) / 64);
        *samples = H2_IDX_HDR + 3 * c + 4;
        *srec = nr;
    }
    if (flags & 2) {
        if (want + 8ull > 0x7FFFFFF0ull) return false;
        *ent = (uint32_t)(want + 8);
        want += 2ull * FQZ_ENT * c;
    }
    if (sz != want || 8ull + sz > n) return false;
    *raw = r;
    *nch = c;
    return true;
}

// regenerated size of a Compressed block = its literals' Regenerated_Size (there are no sequences)
__device__ int literals_regen(const uint8_t *p, uint32_t n, uint32_t *regen)
{
    if (n < 1) return -1;
    uint32_t type = p[0] & 3, fmt = (p[0] >> 2) & 3;
    if (type <= 1) {
        if (fmt == 0 || fmt == 2) *regen = p[0] >> 3;
        else if (fmt == 1) { if (n < 2) return -1; *regen = (p[0] >> 4) | ((uint32_t)p[1] << 4); }
        else { if (n < 3) return -1; *regen = (p[0] >> 4) | ((uint32_t)p[1] << 4) | ((uint32_t)p[2] << 12); }
        return 0;
    }
    if (type == 3) return -1; // treeless literals: not produced by our encoders
    if (fmt <= 1) { if (n < 3) return -1; *regen = (rd24(p) >> 4) & 0x3FF; }
    else if (fmt == 2) { if (n < 4) return -1; *regen = (rd32(p) >> 4) & 0x3FFF; }
    else { if (n < 5) return -1; *regen = (rd32(p) >> 4) & 0x3FFFF; }
    return 0;
}

// Speculative sizing (saves one of the two sequential frame walks): every payload our encoder writes is ONE frame
// that carries its content size and cuts it into 16 KiB blocks, so the pre-entropy size and the number of zstd
// blocks follow from the frame header alone.  One thread per (block, stream); anything else (no content size,
// oversize) marks the payload with FQZ_SPEC_UNKNOWN and the host takes the general two-walk path.  The single walk
// that follows (pass 2) verifies the guess and asks for the general path if a frame turns out different.
#define FQZ_SPEC_UNKNOWN 0xFFFFFFFFu
#define FQZ_DEC_RETRY_GENERAL (-1000) // internal: never returned to callers
__global__ __launch_bounds__(64) void k_dec_fhdr(const uint8_t *in, DecInfo *info, DecBlock *blocks, uint32_t block_cap)
{
    const uint32_t id = blockIdx.x * 64 + threadIdx.x;
    if (id >= info->n_blocks * FQZ_NS || id >= block_cap * FQZ_NS) return;
    DecBlock *b = &blocks[id / FQZ_NS];
    const int s = id % FQZ_NS;
    const uint32_t n = b->pay_len[s];
    b->indexed[s] = 0;
    b->samp_off[s] = 0;
    b->ent_off[s] = 0;
    if (!n) { b->raw_len[s] = 0; b->n_chunks[s] = 0; b->n_frames[s] = 0; return; }
    uint32_t raw, nch, samples, srec, ent;
    if (h2_index(in + b->pay_off[s], n, &raw, &nch, &samples, &srec, &ent)) { // our own payload: sizes and block places come from its index (k_dec_index)
        b->raw_len[s] = raw;
        b->n_chunks[s] = nch;
        b->n_frames[s] = (nch + FQZ_GROUP - 1) / FQZ_GROUP;
        b->indexed[s] = 1;
        if (samples && srec == b->nrec && (s == S_HDR || s == S_PLUS || s == S_NPOS)) b->samp_off[s] = b->pay_off[s] + samples;
        if (ent) b->ent_off[s] = b->pay_off[s] + ent;
        return;
    }
    uint32_t hdr;
    long long fcs;
    int ck;
    if (frame_header(in + b->pay_off[s], n, &hdr, &fcs, &ck) < 0 || fcs < 0 || fcs > 0x7FFFFFF0ll) { b->raw_len[s] = FQZ_SPEC_UNKNOWN; return; }
    b->raw_len[s] = (uint32_t)fcs;
    b->n_chunks[s] = fcs ? (uint32_t)((fcs + FQZ_CHUNK - 1) / FQZ_CHUNK) : 1u;
    b->n_frames[s] = 1;
}

// One workgroup per (block, stream).  The chain of zstd block headers is sequential (every header tells where the
// next one starts), so the workgroup streams the payload through LDS in fixed 64 KiB windows with coalesced loads
// and the first wave hops through the staged bytes with scalar instructions: an LDS round trip per hop instead of a
// global-memory one.  Two window buffers: while wave 0 walks window w, the other 15 waves load window w + 1.  The
// walking wave itself never has a load in flight: hipcc waits for ALL outstanding memory operations (vmcnt(0)) at
// the back-edge of a loop that carries in-flight load registers, which made every hop wait for its own DecChunk
// store (1000 cycles per zstd block).
#define FRAME_WIN 65536u
#define FRAME_SLACK 64u                       // a parse position may read this far past its window
#define FRAME_BUF (FRAME_WIN + FRAME_SLACK)
#define FRAME_NT 1024u                        // threads per workgroup: the walk is bound by how many loads one workgroup keeps in flight
#define FRAME_LD (FRAME_NT - 64u)              // loader threads: every wave but the first
#define FRAME_V4 ((FRAME_BUF / 16 + FRAME_LD - 1) / FRAME_LD) // uint4 per loader thread per window
__device__ __forceinline__ uint32_t lds24(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

// All loads of a window are issued back to back: no branches between them (out-of-range slots read offset 0 and are
// zeroed by a select), so that a window costs one memory round trip.  The one 16-byte slot that straddles the end of
// the input is patched afterwards by frame_win_tail().
__device__ __forceinline__ void frame_win_load(uint4 v[FRAME_V4], const uint8_t *in, uint32_t in_bytes, unsigned long long wbase, uint32_t t)
{
    const bool tiny = in_bytes < 16;
#pragma unroll
    for (uint32_t u = 0; u < FRAME_V4; u++) {
        const uint32_t i = (t + FRAME_LD * u) * 16;
        const unsigned long long a = wbase + i;
        const bool ok = i < FRAME_BUF && a + 16 <= in_bytes;
        const uint4 x = tiny ? make_uint4(0, 0, 0, 0) : *(const uint4 *)(in + (ok ? a : 0ull));
        v[u] = ok ? x : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void frame_win_store(uint8_t *buf, const uint4 v[FRAME_V4], uint32_t t)
{
#pragma unroll
    for (uint32_t u = 0; u < FRAME_V4; u++) {
        const uint32_t i = (t + FRAME_LD * u) * 16;
        if (i < FRAME_BUF) *(uint4 *)(buf + i) = v[u];
    }
}
// bytes of the window that lie in the last, partial 16-byte slot of the input (call after frame_win_store, before the barrier)
__device__ __forceinline__ void frame_win_tail(uint8_t *buf, const uint8_t *in, uint32_t in_bytes, unsigned long long wbase, uint32_t t)
{
    const unsigned long long a0 = (unsigned long long)(in_bytes & ~15u); // start of the partial slot
    if ((in_bytes & 15u) && a0 >= wbase && a0 - wbase < FRAME_BUF && t < (in_bytes & 15u)) buf[a0 - wbase + t] = in[a0 + t];
}

// walker state, kept in LDS between windows so that inside the walking wave every value is wave-uniform (SGPRs)
struct FrameWalk {
    uint32_t pos, nch, dst, in_frame, ck, state; // state: 0 = go on, 1 = done, 2 = failed
    uint32_t nfr, fstart;                         // frames completed, arena offset of the current frame's content
    uint32_t fcs_lo, fcs_hi, has_fcs, lz, frame_lz;
    uint32_t tree_off, tree_len; // last Huffman tree description seen in the frame (for treeless blocks)
    uint32_t frame_regen_lo, frame_regen_hi, total_lo, total_hi;
    uint32_t fch, rtab;          // FQZ-R1: chunk index at the frame's start; 0, or 1 + the place in the frame of the block that carries the frequency table
};

// rlist (version-3 files, else nullptr): the walk announces the rANS groups it finds to k_dec_rans, as k_dec_index does for indexed payloads
__global__ __launch_bounds__(FRAME_NT) void k_dec_frames(const uint8_t *in, uint32_t in_bytes, DecInfo *info, DecBlock *blocks, DecChunk *chunks, DecFrame *frames, int pass,
                                                         int v3, uint2 *rlist, uint32_t rcap, uint32_t set_chunks = 0xFFFFFFFFu,
                                                         unsigned long long *zero = nullptr, uint32_t n_zero = 0)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[2][FRAME_BUF];
    __shared__ FrameWalk W;
    const uint32_t id = blockIdx.x, t = threadIdx.x;
    // (two chores of the launch that would be a copy and a fill of their own - ~8 us of queue time each between the host's
    //  read-back and the first bulk kernel: the chunk count the host worked out, and the block totals k_dec_sizes adds into)
    if (id == 0) {
        if (t == 0 && set_chunks != 0xFFFFFFFFu) info->n_chunks = set_chunks;
        for (uint32_t i = t; i < n_zero; i += FRAME_NT) zero[i] = 0;
    }
    const uint32_t nb = info->n_blocks;
    if (id >= nb * FQZ_NS || info->status) return;
    DecBlock *b = &blocks[id / FQZ_NS];
    const int s = id % FQZ_NS;
    if (pass == 1 && b->lz[s]) return; // a foreign frame: decoded as a whole by k_dec_lz
    if (pass == 2 && b->indexed[s]) return; // our own payload with its index: k_dec_index places the blocks without a walk
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x))) // wave-uniform values live in SGPRs: the walk runs on the scalar unit
    const uint32_t p0 = UNI(b->pay_off[s]), n = UNI(b->pay_len[s]); // the payload is in[p0, p0 + n); `in` is 16-byte aligned
    DecChunk *out = pass ? chunks + UNI(b->chunk_base[s]) : nullptr;
    DecFrame *fout = pass && frames ? frames + UNI(b->frame_base[s]) : nullptr;
    const uint32_t nfr_expected = UNI(b->n_frames[s]);
    const uint32_t n_expected = UNI(b->n_chunks[s]);
    if (t == 0) {
        FrameWalk z;
        memset(&z, 0, sizeof z);
        z.dst = pass ? b->a_off[s] : 0;
        z.state = n ? 0 : 1;
        W = z;
    }
    __syncthreads();
    const uint32_t abase0 = p0 & ~15u; // window w covers absolute bytes [abase0 + w * FRAME_WIN, + FRAME_BUF)
    uint32_t cur = 0;                  // window that holds the position
    bool primed = false;               // win[cur % 2] holds window cur
    uint32_t state = UNI(W.state);
    const uint32_t lt = t - 64;        // loader index (threads 64..)
    while (state == 0) {
        if (!primed) { // start, or a block jumped over a whole window: load window cur now
            if (t >= 64) {
                uint4 v[FRAME_V4];
                frame_win_load(v, in, in_bytes, (unsigned long long)abase0 + (unsigned long long)cur * FRAME_WIN, lt);
                frame_win_store(win[cur % 2], v, lt);
            }
            __syncthreads();
            frame_win_tail(win[cur % 2], in, in_bytes, (unsigned long long)abase0 + (unsigned long long)cur * FRAME_WIN, t);
            primed = true;
            __syncthreads();
        }
        if (t >= 64) { // the loaders fetch window cur + 1 while wave 0 walks window cur
            uint4 v[FRAME_V4];
            frame_win_load(v, in, in_bytes, (unsigned long long)abase0 + (unsigned long long)(cur + 1) * FRAME_WIN, lt);
            frame_win_store(win[(cur + 1) % 2], v, lt);
        } else {
            // wave 0, every lane with the same values: the chain of block headers is walked with scalar instructions
            uint32_t pos = UNI(W.pos), nch = UNI(W.nch), dst = UNI(W.dst), in_frame = UNI(W.in_frame), ck = UNI(W.ck), st = 0;
            uint32_t nfr = UNI(W.nfr), fstart = UNI(W.fstart), fch = UNI(W.fch), rtab = UNI(W.rtab);
            uint32_t has_fcs = UNI(W.has_fcs), lz = UNI(W.lz), frame_lz = UNI(W.frame_lz), tree_off = UNI(W.tree_off), tree_len = UNI(W.tree_len);
            unsigned long long fcs = ((unsigned long long)UNI(W.fcs_hi) << 32) | UNI(W.fcs_lo);
            unsigned long long frame_regen = ((unsigned long long)UNI(W.frame_regen_hi) << 32) | UNI(W.frame_regen_lo);
            unsigned long long regen_total = ((unsigned long long)UNI(W.total_hi) << 32) | UNI(W.total_lo);
            const uint8_t *wbuf = win[cur % 2];
            const uint32_t wlo = cur * FRAME_WIN; // window start, relative to abase0
            for (;;) {
                const uint32_t rel = p0 - abase0 + pos;           // position relative to abase0
                if (rel - wlo >= FRAME_WIN && pos < n) break;     // the position left this window
                const uint8_t *q = wbuf + (rel - wlo);
                if (!in_frame) {
                    if (pos >= n) { st = 1; break; }
                    if (n - pos >= 8 && UNI(skippable_magic(q))) { // a skippable frame (our index, or anybody's metadata): not content
                        uint32_t ssz;
                        __builtin_memcpy(&ssz, q + 4, 4);
                        ssz = UNI(ssz);
                        if (ssz > n - pos - 8) { st = 2; break; }
                        pos += 8 + ssz;
                        continue;
                    }
                    uint32_t hdr;
                    long long f64;
                    int ck0;
                    if (UNI(frame_header(q, n - pos, &hdr, &f64, &ck0) < 0)) { st = 2; break; }
                    has_fcs = UNI(f64 >= 0);
                    fcs = ((unsigned long long)UNI((uint32_t)((unsigned long long)f64 >> 32)) << 32) | UNI((uint32_t)f64);
                    ck = UNI(ck0);
                    pos += UNI(hdr);
                    frame_regen = 0;
                    in_frame = 1;
                    frame_lz = 0;
                    tree_len = 0;
                    fstart = dst;
                    fch = nch;
                    rtab = 0;
                    continue;
                }
                // block header (3 bytes) and the literals header behind it (<= 5 bytes) in one LDS round trip; everything
                // below is wave-uniform and written with selects, so that a hop is a short run of scalar instructions
                uint32_t h0, h1;
                __builtin_memcpy(&h0, q, 4);
                __builtin_memcpy(&h1, q + 4, 4);
                h0 = UNI(h0);
                h1 = UNI(h1);
                const uint32_t bh = h0 & 0xFFFFFFu;
                const uint32_t last = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
                const uint32_t cpos = pos + 3;
                // Regenerated_Size of the literals section = the block's (there are no sequences); literals_regen() on registers
                const uint32_t lh = (h0 >> 24) | (h1 << 8); // literals header bytes 0..3
                const uint32_t lt2 = lh & 3, fmt = (lh >> 2) & 3, l4 = lh >> 4;
                const uint32_t r_plain = fmt == 1 ? (l4 & 0xFFFu) : (fmt == 3 ? (l4 & 0xFFFFFu) : ((lh & 0xFFu) >> 3)); // raw / RLE literals
                const uint32_t n_plain = fmt == 1 ? 2u : (fmt == 3 ? 3u : 1u);
                const uint32_t r_huf = fmt <= 1 ? (l4 & 0x3FFu) : (fmt == 2 ? (l4 & 0x3FFFu) : (l4 & 0x3FFFFu));
                const uint32_t n_huf = fmt <= 1 ? 3u : (fmt == 2 ? 4u : 5u);
                const uint32_t need = lt2 <= 1 ? n_plain : n_huf;
                const uint32_t csize = type == 1 ? 1u : bs;
                // FQZ-R1 block (type 3, fqz_rans.h): m u16 | tflag | ...; the first one of a frame carries the table, and only that one
                const uint32_t m3 = (h0 >> 24) | ((h1 & 0xFFu) << 8), tflag = (h1 >> 8) & 0xFFu;
                const bool r3_ok = (v3 != 0) & (s == S_QUAL) & (bs >= 3 + 64) & ((tflag & ~1u) == 0) & ((tflag != 0) == (rtab == 0)) & (m3 >= 1) & (m3 <= FQZ_CHUNK);
                uint32_t regen = type == 3 ? m3 : (type == 2 ? (lt2 <= 1 ? r_plain : r_huf) : bs);
                // a block of ours ends right after the literals with Number_of_Sequences = 0; anything longer carries LZ sequences
                const uint32_t c_huf = fmt <= 1 ? ((lh >> 14) & 0x3FFu) : (fmt == 2 ? ((lh >> 18) & 0x3FFFu) : (((lh >> 22) | ((h1 >> 24) << 10)) & 0x3FFFFu));
                const uint32_t lit_total = need + (lt2 == 0 ? r_plain : (lt2 == 1 ? 1u : c_huf));
                // (a treeless block of ours follows a block with a tree in the same frame)
                if (n - pos >= 3 && type == 2 && (lit_total + 1 != bs || (lt2 == 3 && !tree_len))) {
                    // foreign frame.  With the speculative sizes that is a wrong guess; on the general path the payload is
                    // handed to k_dec_lz, which needs ONE frame that states its content size
                    if (pass == 2) { st = 3; break; }
                    // general path: the payload is handed to k_dec_lz as a whole.  What a frame with such blocks regenerates is
                    // only known from its content size, so it has to state one; the walk goes on to add up the frames
                    if (!(pass == 0 && has_fcs && fcs <= 0x7FFFFFF0ull)) { st = 2; break; }
                    lz = 1;
                    frame_lz = 1;
                    regen = 0;
                }
                const bool bad = (n - pos < 3) | ((type == 3) & !r3_ok) | ((type == 2) & ((bs < need) | (bs > 128 * 1024))) | (csize > n - cpos) | (regen > 128 * 1024);
                if (bad) { st = 2; break; }
                if (out && (pass == 1 || nch < n_expected)) {
                    if (t == 0) {
                        DecChunk c;
                        c.src_off = p0 + cpos;
                        c.csize = csize;
                        c.dst_off = dst;
                        c.regen = regen;
                        c.btype = type == 3 ? 4u : type; // (4: k_dec_rans)
                        c.tree_off = (type == 2 && lt2 == 3) ? tree_off : 0;
                        c.tree_len = (type == 2 && lt2 == 3) ? tree_len : 0;
                        c.stream = (uint32_t)s;
                        c.seq_len = 0; c.out_off = dst; c.out_len = regen; c.ent_off = 0;
                        out[nch] = c;
                    }
                    dst += regen;
                }
                if (type == 2 && lt2 == 2) { tree_off = p0 + cpos + need; tree_len = c_huf; } // this block's tree serves the treeless ones after it
                if (type == 3 && tflag) rtab = nch - fch + 1;
                nch++;
                pos = cpos + csize;
                frame_regen += regen;
                if (last) {
                    if (rtab) { // the frame's rANS group: its chunks and the one with the table
                        if (nch - fch > FQZ_GROUP) { st = 2; break; }
                        if (out && rlist && (pass == 1 || nch <= n_expected) && t == 0) {
                            const uint32_t r = atomicAdd(&info->n_rgroups, 1u);
                            if (r < rcap) rlist[r] = make_uint2(UNI(b->chunk_base[s]) + fch, (nch - fch) | ((rtab - 1) << 8));
                        }
                        rtab = 0;
                    }
                    if (ck && n - pos < 4) { st = 2; break; }
                    if (frame_lz) frame_regen = fcs;
                    if (has_fcs && fcs != frame_regen) { st = 2; break; }
                    if (fout && (pass == 1 || nfr < nfr_expected) && t == 0) {
                        DecFrame f;
                        f.dst_off = fstart; f.len = (uint32_t)frame_regen; f.ck_off = ck ? p0 + pos : 0u; f.stream = (uint32_t)s;
                        fout[nfr] = f;
                    }
                    nfr++;
                    if (ck) pos += 4;
                    regen_total += frame_regen;
                    in_frame = 0;
                }
            }
            if (t == 0) {
                W.pos = pos; W.nch = nch; W.dst = dst; W.in_frame = in_frame; W.ck = ck; W.state = st; W.has_fcs = has_fcs; W.lz = lz; W.frame_lz = frame_lz; W.tree_off = tree_off; W.tree_len = tree_len;
                W.nfr = nfr; W.fstart = fstart; W.fch = fch; W.rtab = rtab;
                W.fcs_lo = (uint32_t)fcs; W.fcs_hi = (uint32_t)(fcs >> 32);
                W.frame_regen_lo = (uint32_t)frame_regen; W.frame_regen_hi = (uint32_t)(frame_regen >> 32);
                W.total_lo = (uint32_t)regen_total; W.total_hi = (uint32_t)(regen_total >> 32);
            }
        }
        __syncthreads();
        frame_win_tail(win[(cur + 1) % 2], in, in_bytes, (unsigned long long)abase0 + (unsigned long long)(cur + 1) * FRAME_WIN, t);
        state = UNI(W.state);
        const uint32_t nxt = (p0 - abase0 + UNI(W.pos)) / FRAME_WIN;
        if (nxt != cur + 1) primed = false; // jumped over a window (a block longer than 64 KiB): reload at the new place
        cur = nxt;
        __syncthreads();
    }
    if (t == 0) {
        const unsigned long long regen_total = ((unsigned long long)W.total_hi << 32) | W.total_lo;
        if (W.state == 3) { dec_fail(info, FQZ_DEC_RETRY_GENERAL); return; }
        if (W.state == 2 || W.in_frame) { dec_fail(info, FQZ_E_ENTROPY); return; }
        if (regen_total > 0x7FFFFFFFull) { dec_fail(info, FQZ_E_TOO_LARGE); return; }
        if (W.lz) { b->raw_len[s] = (uint32_t)regen_total; b->n_chunks[s] = 0; b->n_frames[s] = 0; b->lz[s] = 1; return; }
        if (!pass) { b->raw_len[s] = (uint32_t)regen_total; b->n_chunks[s] = W.nch; b->n_frames[s] = W.nfr; }
        if (pass == 2 && (W.nch != n_expected || regen_total != b->raw_len[s] || W.nfr != nfr_expected)) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
    }
}

// ---------------------------------------------------------------------------
// K8 entropy decode: one wave per zstd block
// ---------------------------------------------------------------------------
struct BackBits { // backward bit reader (zstd bitstreams are read from their last byte)
    const uint8_t *p;
    unsigned long long buf; // next bits at the MSB end
    int avail;              // valid bits in buf
    int byte_pos;           // bytes [0, byte_pos) not loaded yet
    long long remaining;    // stream bits not yet consumed (may go negative = over-read)
};
__device__ __forceinline__ void bb_refill(BackBits &b)
{
    while (b.avail <= 32 && b.byte_pos > 0) {
        if (b.byte_pos >= 4) {
            uint32_t w = load_u32_unaligned(b.p + b.byte_pos - 4);
            b.buf |= (unsigned long long)w << (32 - b.avail);
            b.avail += 32;
            b.byte_pos -= 4;
        } else {
            uint32_t w = b.p[b.byte_pos - 1];
            b.buf |= (unsigned long long)w << (56 - b.avail);
            b.avail += 8;
            b.byte_pos -= 1;
        }
    }
}
__device__ __forceinline__ int bb_init(BackBits &b, const uint8_t *p, uint32_t n)
{
    if (!n || !p[n - 1]) return -1;
    b.p = p; b.buf = 0; b.avail = 0; b.byte_pos = (int)n;
    bb_refill(b);
    int pad = 8 - highbit32_d(p[n - 1]); // end mark + zero bits above it
    b.buf <<= pad;
    b.avail -= pad;
    b.remaining = (long long)(n - 1) * 8 + highbit32_d(p[n - 1]);
    return 0;
}
__device__ __forceinline__ uint32_t bb_peek(const BackBits &b, int nb) { return nb ? (uint32_t)(b.buf >> (64 - nb)) : 0u; }
__device__ __forceinline__ void bb_skip(BackBits &b, int nb) { b.buf <<= nb; b.avail -= nb; b.remaining -= nb; }
__device__ __forceinline__ uint32_t bb_read(BackBits &b, int nb)
{
    bb_refill(b);
    uint32_t v = bb_peek(b, nb);
    bb_skip(b, nb);
    return v;
}

// FSE-compressed Huffman weights (RFC 8878 4.2.1.2); one lane; returns count or -1
__device__ int fse_decode_weights_dev(const uint8_t *src, uint32_t n, uint8_t *w, int cap, uint8_t *dsym, uint8_t *dnb, uint16_t *dnew)
{
    if (n < 2) return -1;
    uint32_t bitpos = 0;
    auto peek = [&](int nb) -> uint32_t {
        unsigned long long v = 0;
        for (int i = 0; i < 4; i++) { uint32_t q = (bitpos >> 3) + i; if (q < n) v |= (unsigned long long)src[q] << (8 * i); }
        return (uint32_t)(v >> (bitpos & 7)) & ((1u << nb) - 1);
    };
    int table_log = (int)peek(4) + 5;
    bitpos += 4;
    if (table_log > 6) return -1;
    int table_size = 1 << table_log;
    int remaining = table_size + 1, threshold = table_size, nb = table_log + 1;
    short norm[16];
    int sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= 12) {
        if (prev0) {
            int n0 = sym;
            for (;;) { uint32_t r = peek(2); bitpos += 2; n0 += (int)r; if (r != 3) break; }
            if (n0 > 12) return -1;
            while (sym < n0) norm[sym++] = 0;
        }
        int max = (2 * threshold - 1) - remaining, count;
        uint32_t lowv = peek(nb - 1);
        if ((int)lowv < max) { count = (int)lowv; bitpos += nb - 1; }
        else { count = (int)peek(nb); if (count >= threshold) count -= max; bitpos += nb; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (short)count;
        prev0 = !count;
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (bitpos > 8 * n) return -1;
    }
    if (remaining != 1) return -1;
    int max_sv = sym - 1;
    uint32_t hdr = (bitpos + 7) >> 3;
    if (hdr >= n) return -1;
    {
        uint16_t next[13];
        if (max_sv > 12) return -1; // weights are 0..12
        int high = table_size - 1;
        for (int s = 0; s <= max_sv; s++) {
            if (norm[s] == -1) { dsym[high--] = (uint8_t)s; next[s] = 1; }
            else next[s] = (uint16_t)norm[s];
        }
        int step = (table_size >> 1) + (table_size >> 3) + 3, mask = table_size - 1, pos = 0;
        for (int s = 0; s <= max_sv; s++)
            for (int i = 0; i < norm[s]; i++) {
                dsym[pos] = (uint8_t)s;
                pos = (pos + step) & mask;
                while (pos > high) pos = (pos + step) & mask;
            }
        if (pos != 0) return -1;
        for (int u = 0; u < table_size; u++) {
            int s = dsym[u];
            uint32_t ns = next[s]++;
            dnb[u] = (uint8_t)(table_log - highbit32_d(ns));
            dnew[u] = (uint16_t)((ns << dnb[u]) - (uint32_t)table_size);
        }
    }
    BackBits br;
    if (bb_init(br, src + hdr, n - hdr) < 0) return -1;
    if (br.remaining < 2 * table_log) return -1;
    uint32_t s1 = bb_read(br, table_log), s2 = bb_read(br, table_log);
    int out = 0;
    for (;;) { // FSE_decompress_usingDTable tail semantics: stop when a state update over-reads
        if (out >= cap) return -1;
        w[out++] = dsym[s1];
        if (br.remaining < dnb[s1]) { if (out >= cap) return -1; w[out++] = dsym[s2]; break; }
        s1 = dnew[s1] + bb_read(br, dnb[s1]);
        if (out >= cap) return -1;
        w[out++] = dsym[s2];
        if (br.remaining < dnb[s2]) { if (out >= cap) return -1; w[out++] = dsym[s1]; break; }
        s2 = dnew[s2] + bb_read(br, dnb[s2]);
    }
    return out;
}

// Huffman_Tree_Description -> decode table (sym | nbits<<8); one lane. Returns bytes used or -1.
__device__ int huf_read_table_dev(const uint8_t *src, uint32_t n, uint16_t *dt, int *table_log_out, uint8_t *w, uint8_t *scratch)
{
    if (!n) return -1;
    int nw;
    uint32_t used;
    uint32_t hb = src[0];
    if (hb >= 128) {
        nw = (int)hb - 127;
        used = 1 + (uint32_t)(nw + 1) / 2;
        if (used > n) return -1;
        for (int i = 0; i < nw; i += 2) {
            w[i] = src[1 + i / 2] >> 4;
            if (i + 1 < nw) w[i + 1] = src[1 + i / 2] & 15;
        }
    } else {
        used = 1 + hb;
        if (used > n) return -1;
        nw = fse_decode_weights_dev(src + 1, hb, w, 255, scratch, scratch + 64, (uint16_t *)(scratch + 128));
        if (nw < 0) return -1;
    }
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) { if (w[i] > 12) return -1; total += (1u << w[i]) >> 1; }
    if (!total) return -1;
    int table_log = highbit32_d(total) + 1;
    if (table_log > 12) return -1;
    uint32_t rest = (1u << table_log) - total;
    if (rest & (rest - 1)) return -1;
    w[nw] = (uint8_t)(highbit32_d(rest) + 1);
    nw++;
    uint32_t rank_start[14], rank_cnt[14];
    for (int r = 0; r < 14; r++) rank_start[r] = rank_cnt[r] = 0;
    for (int i = 0; i < nw; i++) rank_cnt[w[i]]++;
    if (rank_cnt[1] < 2 || (rank_cnt[1] & 1)) return -1;
    uint32_t next = 0;
    for (int r = 1; r <= table_log; r++) { rank_start[r] = next; next += rank_cnt[r] << (r - 1); }
    for (int s = 0; s < nw; s++) {
        uint32_t ws = w[s];
        if (!ws) continue;
        uint32_t len = (1u << ws) >> 1, base = rank_start[ws];
        uint16_t e = (uint16_t)(s | ((table_log + 1 - ws) << 8));
        for (uint32_t u = 0; u < len; u++) dt[base + u] = e;
        rank_start[ws] = base + len;
    }
    *table_log_out = table_log;
    return (int)used;
}

__device__ int huf_decode_stream_dev(const uint8_t *src, uint32_t n, const uint16_t *dt, int table_log, uint8_t *dst, uint32_t count)
{
    BackBits br;
    if (bb_init(br, src, n) < 0) return -1;
    for (uint32_t i = 0; i < count; i++) {
        bb_refill(br);
        uint32_t e = dt[bb_peek(br, table_log)];
        dst[i] = (uint8_t)e;
        bb_skip(br, (int)(e >> 8));
    }
    return br.remaining == 0 ? 0 : -1;
}

#include "fqz_decode_lz.h"

#define LZ_SCRATCH (128u * 1024u + 64u) // literals of one zstd block
// foreign payloads (frames with LZ sequences): one wave each, blocks in order
__global__ __launch_bounds__(64) void k_dec_lz(const uint8_t *in, DecInfo *info, const DecBlock *blocks, uint8_t *arena, uint8_t *scratch)
{
    __shared__ LzLds L;
    const uint32_t id = blockIdx.x;
    if (id >= info->n_blocks * FQZ_NS || info->status) return;
    const DecBlock *b = &blocks[id / FQZ_NS];
    const int s = id % FQZ_NS;
    const uint32_t slot = LZU(b->lz[s]);
    if (!slot) return;
    // the frames of the payload one after the other (skippable frames are not content)
    const uint8_t *p = in + LZU(b->pay_off[s]);
    const uint32_t n = LZU(b->pay_len[s]), raw = LZU(b->raw_len[s]);
    uint32_t pos = 0, out = 0;
    int r = 0;
    while (pos < n && r >= 0) {
        if (n - pos >= 8 && LZU(skippable_magic(p + pos))) {
            const uint32_t ssz = LZU(rd32(p + pos + 4));
            if (ssz > n - pos - 8) { r = -1; break; }
            pos += 8 + ssz;
            continue;
        }
        uint32_t used = 0, made = 0;
        r = lz_decode_frame(L, p + pos, n - pos, arena + LZU(b->a_off[s]) + out, raw - out, scratch + (size_t)(slot - 1) * LZ_SCRATCH, &used, &made);
        pos += used;
        out += made;
    }
    if (r >= 0 && out != raw) r = -1;
    if (r < 0) dec_fail(info, r == -2 ? FQZ_E_CHECKSUM : FQZ_E_ENTROPY);
}

// ---------------------------------------------------------------------------
// FQZ-H2 payloads: the index in front of the payload gives the size of every zstd block, so the block table is built
// in parallel instead of by a walk along the chain of block headers.  One workgroup per (block, stream): a scan of the
// sizes places every block; a thread per block then checks that a block of that size really starts there, reads its
// type and finds the Huffman tree a treeless block refers to (an earlier block of its group).  Anything that does not
// add up sends the batch down the general path (the index is a hint, never trusted).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dec_index(const uint8_t *in, DecInfo *info, const DecBlock *blocks, DecChunk *chunks, DecFrame *frames, uint2 *rlist, uint32_t rcap)
{
    __shared__ uint32_t sh[4];
    const uint32_t id = blockIdx.x, t = threadIdx.x;
    if (id >= info->n_blocks * FQZ_NS || info->status) return;
    const DecBlock *b = &blocks[id / FQZ_NS];
    const int s = id % FQZ_NS;
    if (!b->indexed[s]) return;
    const uint32_t p0 = b->pay_off[s], n = b->pay_len[s], raw = b->raw_len[s], nch = b->n_chunks[s];
    const uint8_t *idx = in + p0 + H2_IDX_HDR;
    DecChunk *out = chunks + b->chunk_base[s];
    DecFrame *fout = frames + b->frame_base[s];
    const uint32_t body0 = p0 + 8 + rd32(in + p0 + 4); // first byte behind the index frame
    uint32_t carry = 0;
    bool bad = false;
    for (uint32_t c0 = 0; c0 < nch; c0 += 256) {
        const uint32_t c = c0 + t;
        const uint32_t sz = c < nch ? rd24(idx + 3 * c) : 0u;
        uint32_t tot;
        const uint32_t before = carry + block_excl_scan_256(sz, sh, &tot);
        carry += tot;
        if (c < nch) {
            const uint32_t g = c / FQZ_GROUP, cg = c % FQZ_GROUP;
            const uint32_t M = raw - g * FQZ_GROUP * FQZ_CHUNK < FQZ_GROUP * FQZ_CHUNK ? raw - g * FQZ_GROUP * FQZ_CHUNK : FQZ_GROUP * FQZ_CHUNK;
            const uint32_t fh = M < 256u ? 6u : 7u;
            const unsigned long long pos64 = (unsigned long long)body0 + 11ull * g + fh + before; // the block header
            const uint32_t mk = raw - c * FQZ_CHUNK < FQZ_CHUNK ? raw - c * FQZ_CHUNK : FQZ_CHUNK;
            const bool last_in_group = cg == FQZ_GROUP - 1 || c + 1 == nch;
            if (sz < 4 || pos64 + sz + (last_in_group ? 4 : 0) > (unsigned long long)p0 + n) bad = true;
            else {
                const uint32_t pos = (uint32_t)pos64;
                const uint8_t *q = in + pos;
                const uint32_t bh = rd24(q), last = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
                const uint32_t csize = type == 1 ? 1u : bs;
                DecChunk d;
                d.src_off = pos + 3; d.csize = csize; d.dst_off = b->a_off[s] + c * FQZ_CHUNK; d.regen = mk; d.btype = type;
                d.tree_off = 0; d.tree_len = 0; d.stream = (uint32_t)s;
                d.seq_len = 0; d.out_off = d.dst_off; d.out_len = mk;
                d.ent_off = (type == 2 && b->ent_off[s]) ? b->ent_off[s] + 2u * FQZ_ENT * c : 0u;
                if (3 + csize != sz || last != (last_in_group ? 1u : 0u)) bad = true;
                else if (type == 3) {
                    // FQZ-R1 (version-3 files, fqz_rans.h): m u16 | tflag | [table] | words | 16 states.  The first such block of a
                    // group carries the table (and announces the group to k_dec_rans), the others must find it in front of them
                    if (!rlist || s != S_QUAL || bs < 3 + 64 || rd16(q + 3) != mk || (q[5] & ~1u)) bad = true;
                    else {
                        d.btype = 4;
                        bool first = true; // no type-3 block of this group in front of this one
                        uint32_t back = 0;
                        for (uint32_t j = 1; j <= cg && !bad; j++) {
                            back += rd24(idx + 3 * (c - j));
                            if (back > pos - body0) { bad = true; break; }
                            if (((rd24(in + pos - back) >> 1) & 3) == 3) first = false;
                        }
                        if (first != ((q[5] & 1u) != 0)) bad = true;
                        else if (first) {
                            const uint32_t r = atomicAdd(&info->n_rgroups, 1u);
                            const uint32_t in_group = nch - (c - cg) < FQZ_GROUP ? nch - (c - cg) : FQZ_GROUP;
                            if (r < rcap) rlist[r] = make_uint2(b->chunk_base[s] + c - cg, in_group | (cg << 8));
                            else bad = true;
                        }
                    }
                }
                else if (type != 2) { if (bs != mk) bad = true; }
                else {
                    // Compressed block: literals only (Number_of_Sequences = 0), regenerating exactly the chunk
                    const uint32_t lh = q[3] | ((uint32_t)q[4] << 8) | ((uint32_t)q[5] << 16) | ((uint32_t)q[6] << 24);
                    const uint32_t lt2 = lh & 3, fmt = (lh >> 2) & 3, l4 = lh >> 4;
                    const uint32_t r_plain = fmt == 1 ? (l4 & 0xFFFu) : (fmt == 3 ? (l4 & 0xFFFFFu) : ((lh & 0xFFu) >> 3));
                    const uint32_t n_plain = fmt == 1 ? 2u : (fmt == 3 ? 3u : 1u);
                    const uint32_t r_huf = fmt <= 1 ? (l4 & 0x3FFu) : (fmt == 2 ? (l4 & 0x3FFFu) : (l4 & 0x3FFFFu));
                    const uint32_t n_huf = fmt <= 1 ? 3u : (fmt == 2 ? 4u : 5u);
                    const uint32_t need = lt2 <= 1 ? n_plain : n_huf;
                    const uint32_t c_huf = fmt <= 1 ? ((lh >> 14) & 0x3FFu) : (fmt == 2 ? ((lh >> 18) & 0x3FFFu) : (((lh >> 22) | ((uint32_t)q[7] << 10)) & 0x3FFFFu));
                    const uint32_t lit_total = need + (lt2 == 0 ? r_plain : (lt2 == 1 ? 1u : c_huf));
                    const uint32_t regen = lt2 <= 1 ? r_plain : r_huf;
                    const uint32_t scr = s == S_HDR ? b->seq_scratch : (s == S_LEN ? b->seq_scratch_len : 0u);
                    if (bs >= need && lit_total + 2 <= bs && regen <= mk && scr && q[3 + lit_total] != 0) {
                        // sequences behind the literals (headers model, lengths: fqz_decode_seq.h): the literals go to the block's scratch
                        d.dst_off = scr + c * DSEQ_STRIDE;
                        d.regen = regen;
                        d.seq_len = bs - lit_total;
                    } else if (bs < need || lit_total + 1 != bs || regen != mk) bad = true;
                    if (!bad && lt2 == 3) { // treeless: the tree travels with an earlier Compressed block of the group
                        uint32_t back = 0; // bytes from that block's header to this one's
                        bool found = false;
                        for (uint32_t j = 1; j <= cg && !found; j++) {
                            back += rd24(idx + 3 * (c - j));
                            if (back > pos - body0) { bad = true; break; }
                            const uint8_t *qj = in + pos - back;
                            const uint32_t bhj = rd24(qj);
                            if (((bhj >> 1) & 3) != 2) continue;
                            const uint32_t lhj = qj[3] | ((uint32_t)qj[4] << 8) | ((uint32_t)qj[5] << 16) | ((uint32_t)qj[6] << 24);
                            if ((lhj & 3) != 2) continue;
                            const uint32_t fj = (lhj >> 2) & 3;
                            d.tree_off = pos - back + 3 + (fj <= 1 ? 3u : (fj == 2 ? 4u : 5u));
                            d.tree_len = fj <= 1 ? ((lhj >> 14) & 0x3FFu) : (fj == 2 ? ((lhj >> 18) & 0x3FFFu) : (((lhj >> 22) | ((uint32_t)qj[7] << 10)) & 0x3FFFFu));
                            found = true;
                        }
                        if (!found) bad = true;
                    }
                }
                if (cg == 0 && !bad) { // the group's frame header sits right in front of its first block
                    const uint8_t *f = q - fh;
                    const bool ok = f[0] == 0x28 && f[1] == 0xB5 && f[2] == 0x2F && f[3] == 0xFD &&
                                    (M < 256u ? (f[4] == 0x24 && f[5] == M) : (f[4] == 0x64 && (uint32_t)(f[5] | (f[6] << 8)) == M - 256u));
                    if (!ok) bad = true;
                }
                if (last_in_group && !bad) {
                    DecFrame fr;
                    fr.dst_off = b->a_off[s] + g * FQZ_GROUP * FQZ_CHUNK; fr.len = M; fr.ck_off = pos + sz; fr.stream = (uint32_t)s;
                    fout[g] = fr;
                    if (c + 1 == nch && pos + sz + 4 != p0 + n) bad = true; // the payload ends with the last checksum
                }
                out[c] = d;
            }
        }
        __syncthreads();
    }
    if (bad) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
}

// Content checksums: four lanes per frame, 16 frames per wave; a mismatch fails the decode (the reference's decoder
// verifies them too: zstd.Decoder defaults, compress.go:120-122).
#define DXXH_PER_WAVE 16u // frames per wave: every lane busy; 16 stripes in flight per lane cover the latency
__global__ __launch_bounds__(64) void k_dec_xxh(const uint8_t *in, DecInfo *info, const DecFrame *frames, uint32_t n_frames, const uint8_t *arena, uint32_t stream_mask)
{
    if (info->status) return;
    const uint32_t lane = threadIdx.x, f = blockIdx.x * DXXH_PER_WAVE + (lane >> 2);
    DecFrame fr;
    fr.dst_off = fr.len = fr.ck_off = fr.stream = 0;
    if (f < n_frames && (lane >> 2) < DXXH_PER_WAVE) fr = frames[f];
    const bool on = f < n_frames && (lane >> 2) < DXXH_PER_WAVE && fr.ck_off && ((stream_mask >> fr.stream) & 1u);
    const unsigned long long h = xxh64_quad(arena + fr.dst_off, on ? fr.len : 0u, lane);
    if (on && (lane & 3) == 0 && (uint32_t)h != rd32(in + fr.ck_off)) dec_fail(info, FQZ_E_CHECKSUM);
}

// ---------------------------------------------------------------------------
// Fast path for our own profile: Compressed blocks with 4-stream Huffman literals, table log <= 11.
// One wave decodes 16 consecutive zstd blocks at once: lane = 4 * (block in group) + stream.  Per block
// the wave keeps ~0.8 KiB of LDS: an 8-bit first-level table (symbol | nbits << 8, or an escape), the
// symbols in canonical order, and the rank starts for the codes longer than 8 bits.
// Blocks it handled are marked btype = 3 so that k_dec_entropy (the general path) skips them.
// ---------------------------------------------------------------------------
// Backward bit reader with a two-deep software prefetch queue, branch-light.  At a refill the word merged into
// the bit buffer was loaded two refills ago and the word promoted to the head of the queue one refill ago, so the
// s_waitcnt the compiler places in front of those uses never waits on the load issued in this refill.
// The stream's first bytes are fetched with a full dword load that starts up to 3 bytes BEFORE the stream
// (those bytes are block / literals headers of the same zstd block, always readable) and masked.
struct BackBitsP {
    const uint8_t *p;
    unsigned long long buf; // next bits at the MSB end
    int avail;              // valid bits in buf
    int byte_pos;           // bytes [0, byte_pos) not fetched yet
    uint32_t w0, w1;        // prefetched bits, left-aligned in 32 bits (w0 is merged next)
    int n0, n1;             // how many of them are valid (0 = nothing there)
};
__device__ __forceinline__ void bbp_fetch(BackBitsP &b, uint32_t &w, int &nbits)
{
    const int take = b.byte_pos < 4 ? b.byte_pos : 4;            // 0..4 bytes
    w = load_u32_unaligned(b.p + b.byte_pos - 4);                 // bytes [byte_pos-4, byte_pos); NOT touched until it is merged
    nbits = 8 * take;
    b.byte_pos -= take;
}
__device__ __forceinline__ void bbp_refill(BackBitsP &b)
{
    if (b.avail <= 32) {
        // keep the top n0 bits of w0 (a partial first word also holds bytes that precede the stream)
        const uint32_t keep = b.n0 == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> b.n0);
        b.buf |= (unsigned long long)(b.w0 & keep) << (32 - b.avail);
        b.avail += b.n0;          // n0 == 0 once the stream is exhausted: nothing changes
        b.w0 = b.w1;              // loaded one refill ago
        b.n0 = b.n1;
        bbp_fetch(b, b.w1, b.n1); // in flight until the refill after next
    }
}
// returns the number of payload bits of the stream, or -1
__device__ __forceinline__ int bbp_init(BackBitsP &b, const uint8_t *p, uint32_t n)
{
    if (!n || !p[n - 1]) return -1;
    b.p = p; b.buf = 0; b.avail = 0; b.byte_pos = (int)n;
    bbp_fetch(b, b.w0, b.n0);
    bbp_fetch(b, b.w1, b.n1);
    bbp_refill(b); // first word
    bbp_refill(b); // second word when the first left room (avail <= 32)
    int pad = 8 - highbit32_d(p[n - 1]); // end mark + zero bits above it
    b.buf <<= pad;
    b.avail -= pad;
    return (int)(n - 1) * 8 + highbit32_d(p[n - 1]);
}

#include "fqz_decode_seq.h"
#define FQZ_RANS_DECODER
#include "fqz_rans.h"

#define HG 16           // blocks per wave, four lanes a block (one per Huffman stream)
#define L1_BITS 8
#define L1_ESC 0xFFFFu
struct HufGroupLds {
    uint16_t l1[1 << L1_BITS];  // first level; aliases the weight array while the tables are built
    uint8_t syms[256];          // symbols sorted by (weight, symbol); aliases the FSE scratch
    uint16_t rs[16];            // rs[w]: first TL-bit table index of weight w (rs[TL+1] = 1 << TL)
    uint16_t ss[16];            // ss[w]: first position in syms[] of weight w
    uint32_t cnt[16];           // table build by the block's lanes together: symbols per weight, then the next free place in syms[] per weight; 14 total, 15 bad
};

// (the launch covers the chunks [first, first + count): the host numbers the chunks of the quality streams first, so that the
//  launch for them and the launch for the other streams are both dense)
//
// QL = 4: SIXTEEN lanes a block, four per Huffman stream.  A Huffman stream is one dependent chain - table lookup, shift, next
// symbol - of up to 4096 steps, and a batch of 1 GB has 35 k blocks of them: 140 k lanes on a chip that wants half a million,
// each waiting out every LDS and memory latency alone (k_dec_huf ran at 0.9 waves per SIMD and its duration was that of ONE
// stream, ~0.8 ms).  The FQZI index of our own files therefore carries three ENTRY POINTS per stream (oracle: encode_group_chunks):
// quarter q of a stream starts at bit E_q with symbol e_q and must end exactly at E_q+1 - the index is a hint, never trusted: a
// quarter that does not end where the next one began sends the batch to the general path, which reads every stream from its end
// mark as any zstd decoder does.  A block without entry points (QL = 1 launches; a foreign frame) is decoded by its first quarter.
template <int QL>
__global__ __launch_bounds__(64) void k_dec_huf(const uint8_t *in, DecInfo *info, DecChunk *chunks, uint8_t *arena, int dbg, uint32_t stream_mask,
                                                uint32_t first, uint32_t count)
{
    constexpr uint32_t LPB = 4 * QL, HGQ = 64 / LPB; // lanes per block, blocks per wave
    __shared__ HufGroupLds G[HGQ];
    __shared__ uint32_t obuf[16 * 64]; // per-lane 64-byte output staging, transposed: dword j of lane l at [j*64 + l]
    const uint32_t lane = threadIdx.x, grp = lane / LPB, li = lane % LPB, sub = li / QL, qt = li % QL;
    const uint32_t id = first + blockIdx.x * HGQ + grp;
    if (info->status) return;
    const bool have = id < first + count && id < info->n_chunks;
    DecChunk c;
    c.btype = 0;
    if (have) c = chunks[id];
    if (have && !((stream_mask >> c.stream) & 1u)) c.btype = 0; // another launch's stream: nothing to do here
    HufGroupLds &g = G[grp];
    // ---- per block: parse the literals header and the tree description (one lane per block)
    uint32_t ok = 0, tl = 0, lh = 0, lsize = 0, used = 0, regen = c.regen;
    bool treeless = false, one = false, huf_lit = false; // huf_lit: a Compressed block with Huffman-coded literals - this kernel's business
    const uint8_t *src = in + c.src_off;
    if (have && c.btype == 2 && c.csize >= 5) {
        uint32_t type = src[0] & 3, fmt = (src[0] >> 2) & 3;
        treeless = type == 3 && c.tree_len != 0; // the table comes from the tree of an earlier block of the frame
        if (type == 2 || treeless) {
            one = fmt == 0; // a single stream (fewer than 256 literals: the last block of a stream, a well-matched headers chunk)
            if (fmt <= 1) { lh = 3; lsize = (rd24(src) >> 14) & 0x3FF; }
            else if (fmt == 2) { lh = 4; lsize = rd32(src) >> 18; }
            else { lh = 5; lsize = (rd32(src) >> 22) | ((uint32_t)src[4] << 10); }
            ok = (lh + lsize + (c.seq_len ? c.seq_len : 1u) == c.csize && (c.seq_len || src[lh + lsize] == 0) && regen >= (one ? 1u : 4u)) ? 1u : 0u;
            huf_lit = true;
        }
    }
    if (dbg == 3) return; // timing experiment: the block's literals header only
    if constexpr (QL > 1) {
        // The tree description -> canonical order, by the block's LPB lanes together (one lane alone spends as long on it - a
        // byte, a dependent LDS access at a time - as on a quarter of a stream: 0.2 of the kernel's 0.47 ms):
        //   weights   the direct form (what our encoder writes for <= 128 weights): a lane per byte; FSE-compressed: the leader, serially
        //   counts    per weight with LDS atomics; the leader derives the table log, the implied last weight and the ranges
        //   symbols   sorted by (weight, symbol), LPB at a time: a symbol's place = its weight's next free place + the number of
        //             lanes below it that hold the same weight (a ballot per weight)
        uint8_t *w = (uint8_t *)g.l1;
        const uint8_t *ip = treeless ? in + c.tree_off : src + lh;
        const uint32_t tree_lim = treeless ? c.tree_len : lsize;
        const uint32_t lead = lane - li, bmask = (1u << LPB) - 1u;
        int nw = -1;
        if (ok) {
            const uint32_t hb = ip[0];
            if (hb >= 128) {
                nw = (int)hb - 127;
                used = 1 + (uint32_t)(nw + 1) / 2;
                if (used > tree_lim) nw = -1;
                else for (int k = (int)li; 2 * k < nw; k += (int)LPB) { const uint32_t b = ip[1 + k]; w[2 * k] = (uint8_t)(b >> 4); if (2 * k + 1 < nw) w[2 * k + 1] = (uint8_t)(b & 15); }
            } else {
                used = 1 + hb;
                if (li == 0 && used <= tree_lim) nw = fse_decode_weights_dev(ip + 1, hb, w, 255, g.syms, g.syms + 64, (uint16_t *)(g.syms + 128));
            }
            g.cnt[li] = 0; // (LPB = 16 = the array)
        }
        nw = __shfl(nw, (int)lead, WAVE); // (the FSE form: the leader knows)
        if (dbg == 4) return; // timing experiment: ... and the weights read
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (ok && nw > 0) {
            uint32_t tot = 0, bad = 0;
            for (int i = (int)li; i < nw; i += (int)LPB) { const uint32_t x = w[i]; if (x > 12) bad = 1; else { tot += (1u << x) >> 1; atomicAdd(&g.cnt[x], 1u); } }
            if (tot) atomicAdd(&g.cnt[14], tot);
            if (bad) atomicOr(&g.cnt[15], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t good = 0, table_log = 0;
        if (ok && nw > 0 && li == 0) {
            const uint32_t total = g.cnt[14];
            if (!g.cnt[15] && total) {
                table_log = (uint32_t)highbit32_d(total) + 1;
                const uint32_t rest = (1u << table_log) - total;
                if (table_log <= 11 && !(rest & (rest - 1))) {
                    const uint32_t lw = (uint32_t)highbit32_d(rest) + 1;
                    w[nw] = (uint8_t)lw;
                    g.cnt[lw]++;
                    if (g.cnt[1] >= 2 && !(g.cnt[1] & 1)) {
                        uint32_t r = 0, q = 0;
                        for (uint32_t x = 1; x <= table_log; x++) { const uint32_t n = g.cnt[x]; g.rs[x] = (uint16_t)r; g.ss[x] = (uint16_t)q; g.cnt[x] = q; r += n << (x - 1); q += n; }
                        g.rs[table_log + 1] = (uint16_t)r; // = 1 << table_log for a complete code
                        g.ss[table_log + 1] = (uint16_t)q;
                        good = r == (1u << table_log);
                    }
                }
            }
        }
        good = __shfl(good, (int)lead, WAVE);
        table_log = __shfl(table_log, (int)lead, WAVE);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        ok = ok && good;
        const int nsym = ok ? nw + 1 : 0; // (with the implied last weight)
        static_assert(LPB == 16, "g.cnt is cleared a lane an entry");
        for (int rd = 0; __ballot(rd * (int)LPB < nsym) != 0; rd++) { // (the wave's blocks go round together: the ballots are the wave's)
            const int sy = rd * (int)LPB + (int)li;
            const uint32_t x = sy < nsym ? w[sy] : 0u;
            uint32_t below = 0, same = 0;
            for (uint32_t xx = 1; xx <= 11; xx++) {
                const uint32_t m = (uint32_t)(__ballot(x == xx) >> lead) & bmask;
                if (x == xx) { below = __popc(m & ((1u << li) - 1u)); same = __popc(m); }
            }
            uint32_t at = 0;
            if (x) at = g.cnt[x] + below;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (x) { g.syms[at] = (uint8_t)sy; if (!below) g.cnt[x] += same; } // overwrites the FSE scratch: done with it
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        tl = table_log;
        if (treeless) used = 0; // nothing of this block's literals section is a tree
    } else
    if (ok && li == 0) {
        // weights -> g.l1 (as bytes), FSE scratch -> g.syms
        uint8_t *w = (uint8_t *)g.l1;
        const uint8_t *ip = treeless ? in + c.tree_off : src + lh;
        const uint32_t tree_lim = treeless ? c.tree_len : lsize;
        int nw = -1;
        uint32_t hb = ip[0];
        if (hb >= 128) {
            nw = (int)hb - 127;
            used = 1 + (uint32_t)(nw + 1) / 2;
            if (used > tree_lim) nw = -1;
            else for (int i = 0; i < nw; i += 2) { w[i] = ip[1 + i / 2] >> 4; if (i + 1 < nw) w[i + 1] = ip[1 + i / 2] & 15; }
        } else {
            used = 1 + hb;
            if (used <= tree_lim) nw = fse_decode_weights_dev(ip + 1, hb, w, 255, g.syms, g.syms + 64, (uint16_t *)(g.syms + 128));
        }
        uint32_t good = 0, table_log = 0;
        if (dbg == 4) return; // timing experiment: ... and the weights read
        if (nw > 0) {
            uint32_t total = 0, cnt[14];
            for (int r = 0; r < 14; r++) cnt[r] = 0;
            bool bad = false;
            for (int i = 0; i < nw; i++) { uint32_t x = w[i]; if (x > 12) { bad = true; break; } total += (1u << x) >> 1; cnt[x]++; }
            if (!bad && total) {
                table_log = (uint32_t)highbit32_d(total) + 1;
                uint32_t rest = (1u << table_log) - total;
                if (table_log <= 11 && !(rest & (rest - 1))) {
                    uint32_t lw = (uint32_t)highbit32_d(rest) + 1;
                    w[nw] = (uint8_t)lw;
                    cnt[lw]++;
                    nw++;
                    if (cnt[1] >= 2 && !(cnt[1] & 1)) {
                        uint32_t r = 0, q = 0, pos[14];
                        for (uint32_t x = 1; x <= table_log; x++) { g.rs[x] = (uint16_t)r; g.ss[x] = (uint16_t)q; pos[x] = q; r += cnt[x] << (x - 1); q += cnt[x]; }
                        g.rs[table_log + 1] = (uint16_t)r; // = 1 << table_log for a complete code
                        g.ss[table_log + 1] = (uint16_t)q;
                        good = r == (1u << table_log);
                        if (good) for (int sy = 0; sy < nw; sy++) { uint32_t x = w[sy]; if (x) g.syms[pos[x]++] = (uint8_t)sy; } // overwrites the FSE scratch: done with it
                    }
                }
            }
        }
        ok = good;
        tl = table_log;
        if (treeless) used = 0; // nothing of this block's literals section is a tree
    }
    if (dbg == 2) return; // timing experiment: parse + weights only
    // broadcast the leader's verdict to the lanes of its block
    ok = __shfl(ok, (int)(lane - li), WAVE);
    tl = __shfl(tl, (int)(lane - li), WAVE);
    used = __shfl(used, (int)(lane - li), WAVE);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- first-level table: 256 / LPB entries per lane.  Entry v8 covers the TL-bit indices [v8 << (TL-8), ...) when TL > 8.
    const uint32_t l1b = tl < L1_BITS ? tl : L1_BITS;
    if (ok) {
        for (uint32_t e = li; e < (1u << l1b); e += LPB) {
            uint32_t v = e << (tl - l1b);       // smallest TL-bit index with this prefix
            uint32_t w = 1;
            for (uint32_t x = 2; x <= tl; x++) w += v >= g.rs[x];
            uint32_t nb = tl + 1 - w;
            uint16_t ent = L1_ESC;
            if (nb <= l1b) ent = (uint16_t)(g.syms[g.ss[w] + ((v - g.rs[w]) >> (w - 1))] | (nb << 8));
            g.l1[e] = ent; // the weight bytes this overwrites are no longer needed (syms / rs / ss are final)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- the four streams of every block, QL lanes each
    uint32_t fail = 0;
    if (dbg == 1) return; // timing experiment: tables only
    if (ok) {
        const uint8_t *ip = src + lh + used;
        uint32_t rem = lsize - used;
        uint32_t z1 = 0, z2 = 0, z3 = 0;
        const uint32_t seg = one ? regen : (regen + 3) / 4;
        bool geo = rem >= (one ? 1u : 10u);
        if (geo && !one) {
            z1 = rd16(ip); z2 = rd16(ip + 2); z3 = rd16(ip + 4);
            geo = !(6 + z1 + z2 + z3 >= rem || 3 * seg > regen);
        }
        if (!geo) fail = 1;
        else {
            {
                const uint32_t z4 = rem - 6 - z1 - z2 - z3; // (four streams)
                const uint32_t so = one ? 0u : 6 + (sub > 0 ? z1 : 0) + (sub > 1 ? z2 : 0) + (sub > 2 ? z3 : 0);
                const uint32_t sn = one ? rem : (sub == 0 ? z1 : (sub == 1 ? z2 : (sub == 2 ? z3 : z4)));
                const uint32_t L = one ? regen : (sub < 3 ? seg : regen - 3 * seg); // symbols of the stream
                // my quarter: symbols [e_lo, e_hi), bits [p_end, p_top) of the stream (LSB-first from its first byte; the
                // stream is read from the top down)
                uint32_t e_lo = 0, e_hi = L, p_end = 0;
                int p_top = -1; // -1: from the end mark
                if (QL > 1) {
                    if (c.ent_off && !one) {
                        const uint32_t per = ((L + 63) / 64 + 3) & ~3u, q16 = 16 * per;
                        const uint8_t *ep = in + c.ent_off + 2 * (3 * sub);
                        e_lo = qt * q16 < L ? qt * q16 : L;
                        e_hi = (qt == QL - 1 || (qt + 1) * q16 > L) ? L : (qt + 1) * q16;
                        if (qt) p_top = (int)rd16(ep + 2 * (qt - 1));
                        if (qt < QL - 1) p_end = rd16(ep + 2 * qt);
                    } else if (qt) e_lo = e_hi = L; // no entry points: the first quarter decodes the stream
                }
                uint32_t cnt = e_hi - e_lo;
                const bool idle = (QL > 1 && qt && (!c.ent_off || one)) || (one && sub); // (a single stream: the block's first lane)
                uint8_t *dst = arena + c.dst_off + (one ? 0u : sub * seg) + e_lo;
                // 128-bit bit buffer (hi:lo, next bits at the MSB end of hi), four symbols per iteration.  The loads that
                // top the buffer up are issued at the START of an iteration and merged at its END, so their latency
                // overlaps four symbol decodes and no load is in flight across the loop back-edge (hipcc copies
                // loop-carried registers there and would wait for them with vmcnt(0) on every step).
                const uint8_t *sp = ip + so;
                if (idle) {}
                else if (!sn || !sp[sn - 1]) fail = 1;
                else if (p_top > (int)(8 * (sn - 1)) + highbit32_d(sp[sn - 1]) || (p_top >= 0 && (uint32_t)p_top < p_end)) fail = 2; // (an entry point outside the stream)
                else {
                    if (p_top < 0) p_top = (int)(8 * (sn - 1)) + highbit32_d(sp[sn - 1]); // the end mark: the payload's bits lie below it
                    uint32_t b3 = 0, b2 = 0, b1 = 0, b0 = 0; // the bit buffer as four words (b3 on top): a shift is three v_alignbit + one shift
                    int avail = 0;            // valid bits in b3:b2:b1:b0
                    int byte_pos = (p_top + 7) >> 3; // bytes [0, byte_pos) not fetched yet
                    // a dword that ends at byte_pos: bytes before the stream belong to the same zstd block, always readable
                    auto fetch = [&](uint32_t &w, int &nb) {
                        const int take = byte_pos < 4 ? byte_pos : 4;
                        w = load_u32_unaligned(sp + byte_pos - 4);
                        nb = 8 * take;
                        byte_pos -= take;
                    };
                    auto merge = [&](uint32_t w, int nb) { // append the top nb bits of w below the valid bits
                        const uint32_t keep = nb == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> nb);
                        const unsigned long long v = (unsigned long long)(w & keep) << 32; // left-aligned in 64 bits
                        unsigned long long hi = ((unsigned long long)b3 << 32) | b2, lo = ((unsigned long long)b1 << 32) | b0;
                        if (avail < 64) { hi |= v >> avail; if (avail > 32) lo |= v << (64 - avail); }
                        else lo |= v >> (avail - 64);
                        b3 = (uint32_t)(hi >> 32); b2 = (uint32_t)hi; b1 = (uint32_t)(lo >> 32); b0 = (uint32_t)lo;
                        avail += nb;
                    };
                    auto shift_bits = [&](uint32_t nb) { // 1 <= nb <= 31
                        const uint32_t sh = 32 - nb;
                        b3 = __builtin_amdgcn_alignbit(b3, b2, sh);
                        b2 = __builtin_amdgcn_alignbit(b2, b1, sh);
                        b1 = __builtin_amdgcn_alignbit(b1, b0, sh);
                        b0 <<= nb;
                    };
                    { // prime: four dwords
                        uint32_t w[4]; int nb[4];
                        for (int q = 0; q < 4; q++) fetch(w[q], nb[q]);
                        for (int q = 0; q < 4; q++) merge(w[q], nb[q]);
                        const int pad = 8 * ((p_top + 7) >> 3) - p_top; // what lies above the entry in its byte (the end mark and the zero bits above it)
                        if (pad) shift_bits((uint32_t)pad);
                        avail -= pad;
                    }
                    const uint32_t sh1 = 32 - l1b, sh2 = 32 - tl;
                    const bool aligned = (((uintptr_t)dst) & 15) == 0;
                    uint8_t *ob = (uint8_t *)obuf + (lane << 2);
                    int neg = 0; // sign bit set once the buffer ran dry (a corrupt stream)
                    for (uint32_t i = 0; i < cnt; i += 4) {
                        // top-up loads for this iteration (0, 1 or 2 dwords)
                        uint32_t w0 = 0, w1 = 0;
                        int n0 = 0, n1 = 0;
                        // refills are wave-synchronous: the iteration that issues a load also waits for it (nothing may be in flight
                        // across the back-edge), so lanes top up together, when the first of them could run short (4 x 11 bits per
                        // iteration), instead of one lane or another stalling the wave in every iteration
                        if (__ballot(avail < 88)) {
                            if (avail <= 96) fetch(w0, n0);
                            if (avail <= 64) fetch(w1, n1);
                        }
                        uint32_t acc4 = 0;
                        auto symbol = [&](uint32_t u) {
                            uint32_t e = g.l1[b3 >> sh1];
                            if (e == L1_ESC) { // a code longer than the first level: canonical rank search
                                uint32_t v = b3 >> sh2, w = 1;
                                for (uint32_t x = 2; x <= tl; x++) w += v >= g.rs[x];
                                e = g.syms[g.ss[w] + ((v - g.rs[w]) >> (w - 1))] | ((tl + 1 - w) << 8);
                            }
                            const uint32_t nb = e >> 8; // 1..11
                            shift_bits(nb);
                            avail -= (int)nb;
                            acc4 |= (e & 0xFFu) << (8 * u); // (the iteration's four symbols are one dword of the staging)
                        };
                        // all four symbols exist for every lane except in a stream's last iteration: no per-symbol predicate then
                        if (__builtin_amdgcn_ballot_w64(i + 4 > cnt) == 0) {
#pragma unroll
                            for (uint32_t u = 0; u < 4; u++) symbol(u);
                        } else {
#pragma unroll
                            for (uint32_t u = 0; u < 4; u++)
                                if (i + u < cnt) symbol(u);
                        }
                        obuf[((i & 63) >> 2) * 64 + lane] = acc4; // transposed staging: dword (i / 4) % 16 of this lane - one LDS write for four symbols
                        neg |= avail;
                        if (avail < 0) avail = 0; // corrupt stream: caught below
                        merge(w0, n0);
                        merge(w1, n1);
                        const uint32_t done = i + 4 < cnt ? i + 4 : cnt;
                        if ((done & 63) == 0 || done == cnt) { // flush this lane's staged bytes: whole 64-byte lines when possible
                            const uint32_t base = (done - 1) & ~63u, nbytes = done - base;
                            uint8_t *o = dst + base;
                            if (nbytes == 64 && aligned) {
#pragma unroll
                                for (uint32_t j = 0; j < 4; j++)
                                    *(uint4 *)(o + 16 * j) = make_uint4(obuf[(4 * j) * 64 + lane], obuf[(4 * j + 1) * 64 + lane],
                                                                        obuf[(4 * j + 2) * 64 + lane], obuf[(4 * j + 3) * 64 + lane]);
                            } else {
                                for (uint32_t q = 0; q < nbytes; q++) o[q] = ob[((q >> 2) << 8) + (q & 3)];
                            }
                        }
                    }
                    // every bit of the quarter must have gone into a symbol: the read position - the bytes not fetched plus what
                    // the buffer still holds - is where the next quarter began (the stream's first bit for the last one)
                    if (neg < 0 || 8 * byte_pos + avail != (int)p_end) fail = (QL > 1 && c.ent_off && !one) ? 2u : 1u;
                }
            }
        }
    }
    if (fail) dec_fail(info, fail == 2 ? FQZ_DEC_RETRY_GENERAL : FQZ_E_ENTROPY); // (2: the entry points do not fit the stream - or the stream is corrupt: the general path tells)
    if (ok && li == 0 && !fail) chunks[id].btype = 3; // done: the general kernel skips it
    // an indexed batch (QL > 1) has no fallback kernel behind this one on its critical chain (k_dec_entropy, 145 registers a lane,
    // would wait for the qualities' decode to leave room): Huffman-coded literals this kernel does not take - a tree it refuses -
    // send the batch to the general path, which has it and names the error
    if (QL > 1 && huf_lit && !ok && li == 0) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
}

// FQZ-R1 blocks (version-3 files): a wave per group that k_dec_index listed (fqz_rans.h)
__global__ __launch_bounds__(64) void k_dec_rans(const uint8_t *in, DecInfo *info, const DecChunk *chunks, const uint2 *rlist, uint32_t rcap, uint8_t *arena)
{
    __shared__ __attribute__((aligned(16))) RansDecLds S;
    if (blockIdx.x >= info->n_rgroups || blockIdx.x >= rcap || info->status) return;
    bool failed = false;
    rans_decode_group(S, in, chunks, rlist[blockIdx.x], arena, &failed);
    if (failed) dec_fail(info, FQZ_E_ENTROPY);
}

// One wave per workgroup takes the chunks first + blockIdx.x, + gridDim.x, ...: most chunks of a launch are not its business (another
// launch's stream, done by k_dec_huf), and a workgroup per chunk - tens of thousands that need their LDS only to find that out -
// queue behind whatever holds the chip's LDS at the time (the qualities' Huffman decode: 0.26 - 0.44 ms for this kernel).
// what needs no table: Raw and RLE blocks, and Compressed blocks whose literals are raw or RLE (the lengths chunks: 4 literals and
// a match).  Returns false for Huffman-coded literals (dec_entropy_chunk).
__device__ __forceinline__ bool dec_plain_chunk(const uint8_t *in, DecInfo *info, const DecChunk &c, uint8_t *arena)
{
    const uint8_t *src = in + c.src_off;
    uint8_t *dst = arena + c.dst_off;
    const uint32_t lane = threadIdx.x;
    if (c.btype == 0) { // raw block (2-bit packed bases): 128-bit copies, the source is unaligned
        const uint32_t full = c.regen & ~15u, quads = full & ~4095u;
        for (uint32_t i = lane * 16; i < quads; i += 4096) { // four loads in flight a lane
            const uint4 a = load_u128_unaligned(src + i), b = load_u128_unaligned(src + i + 1024), e = load_u128_unaligned(src + i + 2048), f = load_u128_unaligned(src + i + 3072);
            store_u128_unaligned(dst + i, a); store_u128_unaligned(dst + i + 1024, b); store_u128_unaligned(dst + i + 2048, e); store_u128_unaligned(dst + i + 3072, f);
        }
        for (uint32_t i = quads + lane * 16; i < full; i += 64 * 16) store_u128_unaligned(dst + i, load_u128_unaligned(src + i));
        for (uint32_t i = full + lane; i < c.regen; i += 64) dst[i] = src[i];
        return true;
    }
    if (c.btype == 1) {
        const uint32_t v4 = (uint32_t)src[0] * 0x01010101u, full = c.regen & ~15u;
        for (uint32_t i = lane * 16; i < full; i += 64 * 16) store_u128_unaligned(dst + i, make_uint4(v4, v4, v4, v4));
        for (uint32_t i = full + lane; i < c.regen; i += 64) dst[i] = (uint8_t)v4;
        return true;
    }
    // Compressed block with raw / RLE literals
    const uint32_t n = c.csize, type = src[0] & 3, fmt = (src[0] >> 2) & 3, regen = c.regen;
    if (type <= 1) {
        const uint32_t lh = (fmt == 0 || fmt == 2) ? 1 : (fmt == 1 ? 2 : 3), lsize = type == 0 ? regen : 1;
        if (lh + lsize + (c.seq_len ? c.seq_len : 1u) != n || (!c.seq_len && src[lh + lsize] != 0)) { dec_fail(info, FQZ_E_ENTROPY); return true; }
        if (type == 0) for (uint32_t i = lane; i < regen; i += 64) dst[i] = src[lh + i];
        else { uint8_t v = src[lh]; for (uint32_t i = lane; i < regen; i += 64) dst[i] = v; }
        return true;
    }
    return false;
}
// Compressed block with Huffman-coded literals that k_dec_huf left (one stream; a tree it would not take): literals section + "0 sequences"
__device__ __forceinline__ void dec_entropy_chunk(const uint8_t *in, DecInfo *info, const DecChunk &c, uint8_t *arena, uint16_t *s_dt, uint8_t *s_w, uint8_t *s_scratch, int *s_i)
{
    const uint8_t *src = in + c.src_off;
    uint8_t *dst = arena + c.dst_off;
    const uint32_t lane = threadIdx.x;
    if (c.btype != 2) return;
    uint32_t n = c.csize;
    uint32_t type = src[0] & 3, fmt = (src[0] >> 2) & 3, lh, regen = c.regen, lsize = 0, nstreams = 1;
    if (type <= 1) return; // (dec_plain_chunk)
    if (fmt <= 1) { lh = 3; lsize = (rd24(src) >> 14) & 0x3FF; nstreams = fmt ? 4 : 1; }
    else if (fmt == 2) { lh = 4; lsize = rd32(src) >> 18; nstreams = 4; }
    else { lh = 5; lsize = (rd32(src) >> 22) | ((uint32_t)src[4] << 10); nstreams = 4; }
    if (lh + lsize + (c.seq_len ? c.seq_len : 1u) != n || (!c.seq_len && src[lh + lsize] != 0)) { dec_fail(info, FQZ_E_ENTROPY); return; } // sequences nobody announced
    const uint8_t *ip = src + lh;
    const bool treeless = type == 3; // the table comes from the tree of an earlier block of the frame (DecChunk.tree_off)
    if (treeless && !c.tree_len) { if (lane == 0) dec_fail(info, FQZ_E_ENTROPY); return; }
    if (lane == 0) {
        int tl = 0;
        s_i[1] = treeless ? huf_read_table_dev(in + c.tree_off, c.tree_len, s_dt, &tl, s_w, s_scratch) : huf_read_table_dev(ip, lsize, s_dt, &tl, s_w, s_scratch);
        s_i[0] = tl;
    }
    __syncthreads();
    int used = s_i[1], table_log = s_i[0];
    if (used < 0) { if (lane == 0) dec_fail(info, FQZ_E_ENTROPY); return; }
    if (treeless) used = 0;
    ip += used;
    uint32_t rem = lsize - (uint32_t)used;
    if (nstreams == 1) {
        if (lane == 0 && huf_decode_stream_dev(ip, rem, s_dt, table_log, dst, regen) < 0) dec_fail(info, FQZ_E_ENTROPY);
        return;
    }
    if (rem < 10) { if (lane == 0) dec_fail(info, FQZ_E_ENTROPY); return; }
    uint32_t z1 = rd16(ip), z2 = rd16(ip + 2), z3 = rd16(ip + 4);
    if (6 + z1 + z2 + z3 >= rem) { if (lane == 0) dec_fail(info, FQZ_E_ENTROPY); return; }
    uint32_t z4 = rem - 6 - z1 - z2 - z3, seg = (regen + 3) / 4;
    if (3 * seg > regen) { if (lane == 0) dec_fail(info, FQZ_E_ENTROPY); return; }
    if (lane < 4) { // one lane per stream (the format gives 4-way parallelism per block)
        uint32_t so = 6 + (lane > 0 ? z1 : 0) + (lane > 1 ? z2 : 0) + (lane > 2 ? z3 : 0);
        uint32_t sn = lane == 0 ? z1 : (lane == 1 ? z2 : (lane == 2 ? z3 : z4));
        uint32_t cnt = lane < 3 ? seg : regen - 3 * seg;
        if (huf_decode_stream_dev(ip + so, sn, s_dt, table_log, dst + lane * seg, cnt) < 0) dec_fail(info, FQZ_E_ENTROPY);
    }
}
#define DENT_GRID 4096u
// The two kernels below are what runs on the critical chain while the qualities' Huffman decode (80 VGPRs a wave, five or six waves a
// SIMD) holds most of every register file: a wave of theirs starts only where its own registers still fit.  With 145 VGPRs - the
// fallback decoder's table build - the one kernel that did both jobs waited 0.25 ms for places to run 0.03 ms of copies.  So the
// copies have a kernel of their own (a few registers, no LDS), k_dec_huf takes single-stream blocks too, and the fallback runs on the
// general path only (an indexed batch that needs it is redone there).
__global__ __launch_bounds__(64) void k_dec_plain(const uint8_t *in, DecInfo *info, const DecChunk *chunks, uint8_t *arena, uint32_t stream_mask, uint32_t first, uint32_t count)
{
    if (info->status) return;
    const uint32_t end = first + count < info->n_chunks ? first + count : info->n_chunks;
    for (uint32_t id = first + blockIdx.x; id < end; id += gridDim.x) {
        const DecChunk c = chunks[id];
        if (c.btype >= 3 || !((stream_mask >> c.stream) & 1u)) continue;
        (void)dec_plain_chunk(in, info, c, arena);
    }
}
__global__ __launch_bounds__(64) void k_dec_entropy(const uint8_t *in, DecInfo *info, const DecChunk *chunks, uint8_t *arena, uint32_t stream_mask, uint32_t first, uint32_t count)
{
    __shared__ uint16_t s_dt[4096];
    __shared__ uint8_t s_w[260];
    __shared__ uint8_t s_scratch[256];
    __shared__ int s_i[8]; // 0 table log, 1 tree bytes used (or -1), 2..5 per-stream result
    if (info->status) return;
    const uint32_t end = first + count < info->n_chunks ? first + count : info->n_chunks;
    for (uint32_t id = first + blockIdx.x; id < end; id += gridDim.x) {
        const DecChunk c = chunks[id];
        if (c.btype >= 3 || !((stream_mask >> c.stream) & 1u)) continue; // decoded by k_dec_huf (3) / k_dec_rans (4) / another launch's stream
        dec_entropy_chunk(in, info, c, arena, s_dt, s_w, s_scratch, s_i);
        __syncthreads(); // (one wave: orders its LDS traffic before the next chunk's table)
    }
}

// ---------------------------------------------------------------------------
// per-record offsets inside the length-prefixed streams (headers, plus, nPos):
// record r = [u16 k][k * unit bytes] (compress.go:977-1015, 1055-1078).  The chain through a block is
// sequential, so it is cut into 16 KiB tiles:
//   k_dec_walk0  shortcuts (v1 / all-bare '+', no N anywhere) -> offsets written directly
//   k_dec_walk1  per tile: 256 lanes walk from every entry offset 0..255 -> transfer function F[tile][e]
//   k_dec_walk2  per (block, stream): chain the true entry offsets through F (one hop per tile)
//   k_dec_walk3  per tile: one lane walks from the true entry and writes the record offsets
// ---------------------------------------------------------------------------
#ifndef WALK_TILE
#define WALK_TILE 8192u // A/B on one box, walk1 + walk2 + walk3: 0.53 ms at 8 KiB, 0.63 at 16 KiB, 1.15 at 32 KiB
#endif
#define WALK_ENTRIES 256u
#define WALK_END 0xFFFFFFFFu
struct WalkF { uint32_t exit, cnt; };      // exit: offset into the next tile, or WALK_END when the stream ended
struct WalkEntry { uint32_t e, base; };    // true entry offset (WALK_END = nothing to do) and first record index

__device__ __forceinline__ int walk_stream(int which) { return which == 0 ? S_HDR : (which == 1 ? S_PLUS : S_NPOS); }
__device__ __forceinline__ int walk_err(int which) { return which == 0 ? FQZ_E_TRUNC_HEADER : (which == 1 ? FQZ_E_TRUNC_PLUS : FQZ_E_TRUNC_NPOS); }

// trivial streams need no walk: an absent plus stream, or a stream in which every record is just a zero prefix
// (no plus payloads / no N at all).  One 1024-thread workgroup per (block, stream).
__global__ __launch_bounds__(1024) void k_dec_walk0(const uint8_t *arena, DecInfo *info, DecBlock *blocks, uint32_t *offs, uint32_t ostride)
{
    const uint32_t bidx = blockIdx.x / 3, which = blockIdx.x % 3;
    if (bidx >= info->n_blocks || info->status) return;
    DecBlock *b = &blocks[bidx];
    const int s = walk_stream(which);
    const uint32_t len = b->raw_len[s], nrec = b->nrec, t = threadIdx.x;
    uint32_t *out = offs + (size_t)which * ostride + b->rec_base;
    uint32_t mode = 0; // 0 = walk, 1 = done here
    // every record owns a 2-byte prefix in these streams (compress.go:977-980, 1055-1060): a stream too short for nrec prefixes
    // is truncated whatever it holds (and would otherwise leave the offsets of its records unwritten)
    if ((unsigned long long)len < 2ull * nrec && !(s == S_PLUS && len == 0)) { if (t == 0) dec_fail(info, walk_err((int)which)); return; }
    if (s == S_PLUS && len == 0) { // v1 container or all-bare '+': appendPlusLine's fast path (compress.go:995-999)
        for (uint32_t r = t; r < nrec; r += 1024) out[r] = 0;
        mode = 1;
    } else if ((unsigned long long)len == 2ull * nrec) { // every record is just its prefix, provided all prefixes are zero
        const uint8_t *p = arena + b->a_off[s]; // 16-byte aligned, padded to whole uint4s
        uint32_t bad = 0;
        for (uint32_t i = t * 16; i < len; i += 1024 * 16) {
            const uint4 v = *(const uint4 *)(p + i);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t have = len - i > 4 * k ? len - i - 4 * k : 0;
                bad |= have >= 4 ? w[k] : (have ? w[k] & ((1u << (8 * have)) - 1) : 0u);
            }
        }
        if (!__syncthreads_or(bad != 0)) { for (uint32_t r = t; r < nrec; r += 1024) out[r] = 2 * r; mode = 1; }
        else { if (t == 0) dec_fail(info, walk_err((int)which)); return; } // nrec prefixes fill the stream, so a payload behind any of them runs past its end
    } else if (nrec == 0) mode = 1;
    if (mode == 0 && b->samp_off[s]) mode = 2; // the index carries record samples: k_dec_walk_s
    if (t == 0) b->walk_mode[which] = mode;
}

// Streams whose index carries record samples (the stream offset of every 64th record): a lane walks the 64 records behind its
// sample and must arrive exactly at the next one (the index is a hint: anything that does not add up sends the batch to the
// general path, which walks the chains from the start and names the error).  grid: xper workgroups per (block, stream)
__global__ __launch_bounds__(64) void k_dec_walk_s(const uint8_t *in, const uint8_t *arena, DecInfo *info, const DecBlock *blocks, uint32_t *offs, uint32_t ostride, uint32_t xper)
{
    const uint32_t bw = blockIdx.x / xper, bx = blockIdx.x % xper, bidx = bw / 3, which = bw % 3;
    if (bidx >= info->n_blocks || info->status) return;
    const DecBlock *b = &blocks[bidx];
    if (b->walk_mode[which] != 2) return;
    const int s = walk_stream((int)which);
    const uint32_t unit = s == S_NPOS ? 2 : 1, len = b->raw_len[s], nrec = b->nrec, ngroups = (nrec + 63) / 64;
    const uint32_t g = bx * 64 + threadIdx.x;
    if (g >= ngroups) return;
    const uint8_t *samp = in + b->samp_off[s]; // sample k (record 64 k, k >= 1) at 4 (k - 1)
    uint32_t pos = g ? rd32(samp + 4 * (g - 1)) : 0u;
    const uint32_t want = g + 1 < ngroups ? rd32(samp + 4 * g) : len;
    const uint8_t *p = arena + b->a_off[s];
    uint32_t *out = offs + (size_t)which * ostride + b->rec_base + 64 * g;
    const uint32_t cnt = nrec - 64 * g < 64 ? nrec - 64 * g : 64;
    bool bad = pos > len || want > len;
    for (uint32_t i = 0; i < cnt && !bad; i++) {
        if (pos + 2 > len) { bad = true; break; }
        out[i] = pos;
        pos += 2 + unit * rd16(p + pos);
    }
    if (bad || pos != want) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
}

// tile -> (block, which, first byte of the tile in the stream)
__device__ __forceinline__ bool walk_locate(const DecBlock *blocks, uint32_t nb, uint32_t tile, uint32_t *bidx, uint32_t *which, uint32_t *t0)
{
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (blocks[mid].tile_base[0] <= tile) lo = mid; else hi = mid; }
    const DecBlock *b = &blocks[lo];
    for (int w = 2; w >= 0; w--) {
        uint32_t nt = (b->raw_len[walk_stream(w)] + WALK_TILE - 1) / WALK_TILE;
        if (nt && tile >= b->tile_base[w] && tile < b->tile_base[w] + nt) { *bidx = lo; *which = (uint32_t)w; *t0 = (tile - b->tile_base[w]) * WALK_TILE; return true; }
    }
    return false;
}

__device__ __forceinline__ void walk_load_tile(uint8_t *tile, const uint8_t *p, uint32_t t0, uint32_t len, uint32_t tid, uint32_t nthreads)
{
    // WALK_TILE + 16 bytes, zero beyond the end of the stream (the arena is padded to whole uint4s)
    for (uint32_t i = tid * 16; i < WALK_TILE + 16; i += nthreads * 16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t0 + i < len) v = *(const uint4 *)(p + t0 + i);
        *(uint4 *)(tile + i) = v;
    }
}

__global__ __launch_bounds__(256) void k_dec_walk1(const uint8_t *arena, const DecInfo *info, const DecBlock *blocks, uint32_t n_tiles, WalkF *F)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[WALK_TILE + 16];
    const uint32_t tidx = blockIdx.x;
    if (tidx >= n_tiles || info->status) return;
    uint32_t bidx, which, t0;
    if (!walk_locate(blocks, info->n_blocks, tidx, &bidx, &which, &t0)) return;
    const DecBlock *b = &blocks[bidx];
    if (b->walk_mode[which]) return;
    const int s = walk_stream((int)which);
    const uint32_t unit = s == S_NPOS ? 2 : 1, len = b->raw_len[s];
    walk_load_tile(tile, arena + b->a_off[s], t0, len, threadIdx.x, 256);
    __syncthreads();
    // a tile is entered at an offset below the longest record; records are short, so for most streams the first 64
    // (or 128) entry offsets cover every tile and the other waves leave right after helping with the load
    if (threadIdx.x >= b->walk_cand[which]) return;
    uint32_t pos = threadIdx.x, cnt = 0;
    WalkF f;
    for (;;) {
        if (t0 + pos + 2 > len) { f.exit = WALK_END; break; }           // no further record starts in the stream
        if (pos >= WALK_TILE) { f.exit = pos - WALK_TILE; break; }
        uint16_t k16;
        __builtin_memcpy(&k16, tile + pos, 2); // one (unaligned) LDS read per hop
        const uint32_t k = k16;
        cnt++;
        unsigned long long nxt = (unsigned long long)pos + 2 + (unsigned long long)k * unit;
        if (t0 + nxt > len) { f.exit = WALK_END; break; }               // truncated record: reported by walk3
        pos = (uint32_t)nxt;
    }
    f.cnt = cnt;
    F[(size_t)tidx * WALK_ENTRIES + threadIdx.x] = f;
}

#ifndef W2_BATCH
#define W2_BATCH 8u // (32 measured slower: 0.106 vs 0.077 ms)
#endif
__global__ __launch_bounds__(64) void k_dec_walk2(const uint8_t *arena, DecInfo *info, const DecBlock *blocks, const WalkF *F, WalkEntry *entries)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[WALK_TILE + 16];
    __shared__ WalkF s_f;
    const uint32_t bidx = blockIdx.x / 3, which = blockIdx.x % 3;
    if (bidx >= info->n_blocks || info->status) return;
    const DecBlock *b = &blocks[bidx];
    if (b->walk_mode[which]) return;
    const int s = walk_stream((int)which);
    const uint32_t unit = s == S_NPOS ? 2 : 1, len = b->raw_len[s], nrec = b->nrec, cand = b->walk_cand[which];
    const uint32_t nt = (len + WALK_TILE - 1) / WALK_TILE, tb = b->tile_base[which];
    const uint8_t *p = arena + b->a_off[s];
    // every lane carries the same chain state (the loop is uniform); lane 0 writes.  The chain hops from tile to tile
    // through F[tile][entry offset]: fetched one hop at a time that is a dependent global load per tile, so the first 64
    // entries of W2_BATCH tiles are loaded at once (lane l holds entry l: one memory round trip per batch) and a hop is a
    // cross-lane read.
    uint32_t e = 0, base = 0;
    bool ended = false;
    for (uint32_t k0 = 0; k0 < nt; k0 += W2_BATCH) {
        WalkF row[W2_BATCH];
        WalkEntry mine;
        mine.e = WALK_END;
        mine.base = 0;
#pragma unroll
        for (uint32_t j = 0; j < W2_BATCH; j++) {
            row[j].exit = WALK_END;
            row[j].cnt = 0;
            if (k0 + j < nt) row[j] = F[(size_t)(tb + k0 + j) * WALK_ENTRIES + threadIdx.x]; // (walk_cand >= 64 = the lanes of this wave)
        }
#pragma unroll
        for (uint32_t j = 0; j < W2_BATCH; j++) {
            const uint32_t k = k0 + j;
            if (k >= nt) break;
            WalkEntry we;
            we.e = (ended || base >= nrec) ? WALK_END : e;
            we.base = base;
            if (threadIdx.x == j) mine = we; // stored after the batch: a store per hop would be waited for (vmcnt) by the next hop
            if (we.e == WALK_END) continue;
            if (e >= WALK_TILE) { e -= WALK_TILE; continue; } // a long record covers this whole tile
            WalkF f;
            if (e < 64) { // e is wave-uniform: a v_readlane with the lane in an SGPR, no LDS round trip in the chain
                const int el = __builtin_amdgcn_readfirstlane((int)e);
                f.exit = (uint32_t)__builtin_amdgcn_readlane((int)row[j].exit, el);
                f.cnt = (uint32_t)__builtin_amdgcn_readlane((int)row[j].cnt, el);
            }
            else if (e < cand) f = F[(size_t)(tb + k) * WALK_ENTRIES + e];
            else { // entry beyond the precomputed range (a record longer than cand - 2 bytes precedes): stage the tile and walk it here
                const uint32_t t0 = k * WALK_TILE;
                __syncthreads();
                walk_load_tile(tile, p, t0, len, threadIdx.x, 64);
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t pos = e, cnt = 0;
                    for (;;) {
                        if (t0 + pos + 2 > len) { f.exit = WALK_END; break; }
                        if (pos >= WALK_TILE) { f.exit = pos - WALK_TILE; break; }
                        uint16_t k16;
                        __builtin_memcpy(&k16, tile + pos, 2);
                        cnt++;
                        unsigned long long nxt = (unsigned long long)pos + 2 + (unsigned long long)k16 * unit;
                        if (t0 + nxt > len) { f.exit = WALK_END; break; }
                        pos = (uint32_t)nxt;
                    }
                    f.cnt = cnt;
                    s_f = f;
                }
                __syncthreads();
                f = s_f;
            }
            base += f.cnt;
            if (f.exit == WALK_END) ended = true; else e = f.exit;
        }
        if (threadIdx.x < W2_BATCH && k0 + threadIdx.x < nt) entries[tb + k0 + threadIdx.x] = mine; // lane j: tile k0 + j
    }
    if (base < nrec && threadIdx.x == 0) dec_fail(info, walk_err((int)which)); // the stream ran out before NumRecords records
}

__global__ __launch_bounds__(64) void k_dec_walk3(const uint8_t *arena, DecInfo *info, const DecBlock *blocks, uint32_t n_tiles, const WalkEntry *entries,
                                                  uint32_t *offs, uint32_t ostride)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[WALK_TILE + 16];
    const uint32_t tidx = blockIdx.x;
    if (tidx >= n_tiles || info->status) return;
    uint32_t bidx, which, t0;
    if (!walk_locate(blocks, info->n_blocks, tidx, &bidx, &which, &t0)) return;
    const DecBlock *b = &blocks[bidx];
    if (b->walk_mode[which]) return;
    const WalkEntry we = entries[tidx];
    if (we.e == WALK_END || we.e >= WALK_TILE) return;
    const int s = walk_stream((int)which);
    const uint32_t unit = s == S_NPOS ? 2 : 1, len = b->raw_len[s], nrec = b->nrec;
    walk_load_tile(tile, arena + b->a_off[s], t0, len, threadIdx.x, 64);
    __syncthreads();
    if (threadIdx.x) return;
    uint32_t *out = offs + (size_t)which * ostride + b->rec_base;
    uint32_t pos = we.e, r = we.base;
    while (r < nrec && pos < WALK_TILE) {
        if (t0 + pos + 2 > len) { dec_fail(info, walk_err((int)which)); return; }
        uint16_t k16;
        __builtin_memcpy(&k16, tile + pos, 2);
        const uint32_t k = k16;
        unsigned long long nxt = (unsigned long long)pos + 2 + (unsigned long long)k * unit;
        if (t0 + nxt > len) { dec_fail(info, walk_err((int)which)); return; }
        out[r++] = t0 + pos;
        pos = (uint32_t)nxt;
    }
}

// ---------------------------------------------------------------------------
// per-record sizes (one thread per record) -> scanned into offsets
// cols: 0 packed seq bytes, 1 quality bytes, 2 FASTQ text bytes
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t find_block(const DecBlock *blocks, uint32_t nb, uint32_t r)
{
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (blocks[mid].rec_base <= r) lo = mid; else hi = mid; }
    return lo;
}

// btot[block][3]: 64-bit sums of the three columns per block: the 32-bit columns (and their scans) may wrap for a crafted
// block, the totals that the truncation checks compare may not
// gsum[c][g]: the sum of column c over the records [64 g, 64 g + 64) (as 32-bit values, like the columns).  The offsets that
// k_dec_assemble needs are prefix sums of the columns; only the group sums are scanned by kernels of their own (1/64 of the
// elements: three short launches instead of two passes over 34 MB), a wave of k_dec_assemble scans its 64 records itself.
__global__ __launch_bounds__(256) void k_dec_sizes(const uint8_t *arena, DecInfo *info, const DecBlock *blocks, const uint32_t *offs,
                                                   uint32_t ostride, uint32_t *cols, uint32_t cstride, unsigned long long *btot, uint32_t *gsum, uint32_t gstride)
{
    uint32_t n_rec = info->n_rec, nb = info->n_blocks;
    if (info->status) return;
    const uint32_t lane = threadIdx.x & 63;
    // a workgroup takes a contiguous run of records (almost always inside one block): the 64-bit block totals are kept in
    // registers and flushed with one atomic per wave when the block changes and at the end (an atomic per wave and trip on the
    // three words of a block took 0.4 ms)
    const uint32_t per = ((n_rec + gridDim.x - 1) / gridDim.x + 255) & ~255u, r_begin = blockIdx.x * per;
    const uint32_t r_end = r_begin + per < n_rec ? r_begin + per : n_rec;
    unsigned long long a0 = 0, a1 = 0, a2 = 0; // this lane's share of the totals of block `cur`
    uint32_t cur = 0xFFFFFFFFu;                // (wave-uniform)
    auto flush = [&]() {
        if (cur == 0xFFFFFFFFu) return;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { a0 += __shfl_xor(a0, d, WAVE); a1 += __shfl_xor(a1, d, WAVE); a2 += __shfl_xor(a2, d, WAVE); }
        if (lane == 0 && (a0 | a1 | a2)) { atomicAdd(&btot[3ull * cur], a0); atomicAdd(&btot[3ull * cur + 1], a1); atomicAdd(&btot[3ull * cur + 2], a2); }
        a0 = a1 = a2 = 0;
    };
    for (uint32_t r0 = r_begin; r0 < r_end; r0 += 256) {
        const uint32_t r = r0 + threadIdx.x;
        unsigned long long v0 = 0, v1 = 0, v2 = 0;
        uint32_t bi = 0xFFFFFFFFu;
        if (r < r_end) {
            bi = find_block(blocks, nb, r);
            const DecBlock *b = &blocks[bi];
            uint32_t lr = r - b->rec_base;
            if (4ull * (lr + 1) > b->raw_len[S_LEN]) { dec_fail(info, FQZ_E_TRUNC_LEN); bi = 0xFFFFFFFFu; } // readSeqLength compress.go:1046
            else {
                uint32_t L = *(const uint32_t *)(arena + b->a_off[S_LEN] + 4 * lr);
                // appendSequence / appendQuality (compress.go:1017-1041) check every record against what is left of the stream
                if (((unsigned long long)L + 3) / 4 > b->raw_len[S_SEQ]) { dec_fail(info, FQZ_E_TRUNC_SEQ); L = 0; }
                else if (L > b->raw_len[S_QUAL]) { dec_fail(info, FQZ_E_TRUNC_QUAL); L = 0; }
                uint32_t H = rd16(arena + b->a_off[S_HDR] + offs[r]);
                uint32_t P = b->raw_len[S_PLUS] ? rd16(arena + b->a_off[S_PLUS] + offs[ostride + r]) : 0;
                v0 = (L + 3) >> 2; v1 = L; v2 = (unsigned long long)H + P + 2ull * L + 6;
                cols[r] = (uint32_t)v0;
                cols[cstride + r] = (uint32_t)v1;
                cols[2 * (size_t)cstride + r] = (uint32_t)v2;
            }
        }
        { // the group's sums (a wave's 64 records are one group: r0 is a multiple of 256)
            uint32_t g0 = (uint32_t)v0, g1 = (uint32_t)v1, g2 = (uint32_t)v2;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { g0 += __shfl_xor(g0, d, WAVE); g1 += __shfl_xor(g1, d, WAVE); g2 += __shfl_xor(g2, d, WAVE); }
            const uint32_t rw = r0 + (threadIdx.x & ~63u);
            if (lane == 0 && rw < r_end) { gsum[rw >> 6] = g0; gsum[gstride + (rw >> 6)] = g1; gsum[2 * (size_t)gstride + (rw >> 6)] = g2; }
        }
        // the wave's records share a block almost always; a wave that straddles two blocks adds its lanes one by one
        const unsigned long long act = __ballot(bi != 0xFFFFFFFFu);
        if (!act) continue;
        const uint32_t b_first = (uint32_t)__shfl((int)bi, __ffsll((long long)act) - 1, WAVE);
        if (__ballot(bi != 0xFFFFFFFFu && bi != b_first) == 0) {
            if (b_first != cur) { flush(); cur = b_first; }
            a0 += v0; a1 += v1; a2 += v2;
        } else if (bi != 0xFFFFFFFFu) {
            atomicAdd(&btot[3ull * bi], v0); atomicAdd(&btot[3ull * bi + 1], v1); atomicAdd(&btot[3ull * bi + 2], v2);
        }
    }
    // the end of the run: the four waves usually hold the same block - one atomic per workgroup and column
    __shared__ unsigned long long s_part[4][3];
    __shared__ uint32_t s_cur[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a0 += __shfl_xor(a0, d, WAVE); a1 += __shfl_xor(a1, d, WAVE); a2 += __shfl_xor(a2, d, WAVE); }
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 0) { s_part[wv][0] = a0; s_part[wv][1] = a1; s_part[wv][2] = a2; s_cur[wv] = cur; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 0; w < 4; w++) {
            if (s_cur[w] == 0xFFFFFFFFu) continue;
            unsigned long long t0 = s_part[w][0], t1 = s_part[w][1], t2 = s_part[w][2];
            for (uint32_t w2 = w + 1; w2 < 4; w2++)
                if (s_cur[w2] == s_cur[w]) { t0 += s_part[w2][0]; t1 += s_part[w2][1]; t2 += s_part[w2][2]; s_cur[w2] = 0xFFFFFFFFu; }
            if (t0 | t1 | t2) { atomicAdd(&btot[3ull * s_cur[w]], t0); atomicAdd(&btot[3ull * s_cur[w] + 1], t1); atomicAdd(&btot[3ull * s_cur[w] + 2], t2); }
        }
    }
}

// one workgroup: the 64-bit totals of every block against its stream lengths, their sum against the output capacity
// ... and, per block, the scanned value of the packed-bases and quality columns at its first record (bbase[c][b]: the streams of a
// block start at its own arena offset): the scanned sum of its group plus the records of that group in front of it
__global__ __launch_bounds__(256) void k_dec_check(DecInfo *info, const DecBlock *blocks, const unsigned long long *btot, size_t out_cap, const uint32_t *cols,
                                                   uint32_t cstride, const uint32_t *gsum, uint32_t gstride, uint32_t *bbase, uint32_t bstride)
{
    __shared__ unsigned long long s_sum[256];
    if (info->status) return;
    const uint32_t nb = info->n_blocks;
    unsigned long long mine = 0;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const DecBlock *k = &blocks[b];
        {
            const uint32_t r0 = k->rec_base, g0 = r0 >> 6;
            uint32_t b0 = gsum[g0], b1 = gsum[gstride + g0];
            for (uint32_t r = g0 << 6; r < r0; r++) { b0 += cols[r]; b1 += cols[cstride + r]; }
            bbase[b] = b0; bbase[bstride + b] = b1;
        }
        if (btot[3ull * b] > k->raw_len[S_SEQ]) dec_fail(info, FQZ_E_TRUNC_SEQ);
        if (btot[3ull * b + 1] > k->raw_len[S_QUAL]) dec_fail(info, FQZ_E_TRUNC_QUAL);
        mine += btot[3ull * b + 2];
    }
    s_sum[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 256; i++) tot += s_sum[i];
        info->out_len = tot;
        if (tot > 0xFFFFFFF0ull) dec_fail(info, FQZ_E_TOO_LARGE); // output offsets are 32 bits
        else if (tot > out_cap) dec_fail(info, FQZ_E_DST_SMALL);
    }
}

// ---------------------------------------------------------------------------
// K9-K11: unpack bases + N overlay, quality undelta, FASTQ text assembly (compress.go:944-1044).
// Piece-centric like k_split: a wave owns 64 consecutive records; every lane gathers the metadata of its
// record (stream cursors, lengths, output offset) with coalesced loads; then the header payloads, bases,
// plus payloads and qualities of the 64 records are cut into 16-output-byte pieces and every lane of every
// round produces one piece (one unaligned load, the transform, one unaligned 128-bit store).  The piece ->
// record map is a binary search over a wave scan of the piece counts (ds_bpermute).
// The quality prefix sum restarts per record: pieces of a record sit in consecutive lanes, so an ordinary wave
// scan of the piece totals minus its value at the record's first piece gives the sum of the pieces before;
// a record that continues from the previous round takes the running sum carried over instead.
// ---------------------------------------------------------------------------
#ifndef ASM_G
#define ASM_G 64u // records a wave gathers the metadata of at a time (<= 64)
#endif
// The text of a trip's records is staged in LDS and leaves in aligned 16-byte units, whole lines at a time: a field of a
// record is a run of ~150 bytes, and stored field by field its two end chunks were written to memory twice (WRITE_SIZE
// 1.57 x the text).  Trips whose text does not fit the window (long reads) store directly as before.
#ifndef ASM_W
#define ASM_W 8192u // bytes of text a wave stages (0: never); 23 records of 150 bp (A/B on one box: 4 KiB 0.56 ms, 6 KiB 0.53, 8 KiB 0.50, 12 KiB 0.53, 16 KiB 0.65; direct stores 0.76-0.86)
#endif
#define DRL(v, i) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (i)))
#define DSH(v, i) ((uint32_t)__shfl((int)(v), (int)(i), WAVE))
__global__ __launch_bounds__(256) void k_dec_assemble(const uint8_t *__restrict__ arena, DecInfo *info, const DecBlock *__restrict__ blocks,
                                                      const uint32_t *__restrict__ offs, uint32_t ostride, const uint32_t *__restrict__ cols,
                                                      uint32_t cstride, uint32_t qoff, uint8_t *__restrict__ out, const uint32_t *__restrict__ gsum,
                                                      uint32_t gstride, const uint32_t *__restrict__ bbase, uint32_t bstride)
{
    static_assert(ASM_G == 64, "a wave's records are one group of the scanned sums");
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][ASM_W + 32];
    if (info->status) return;
    const uint32_t n_rec = info->n_rec, nb = info->n_blocks;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6, lane = lane_id();
    const uint32_t n_groups = (n_rec + ASM_G - 1) / ASM_G;
    uint8_t *const stg = s_stage[threadIdx.x >> 6];
    const bool out_aligned = (((uintptr_t)out) & 15) == 0;
    for (uint32_t g = wave; g < n_groups; g += nwaves) {
        const uint32_t r = lane < ASM_G ? g * ASM_G + lane : n_rec;
        uint32_t m_L = 0, m_hdr = 0, m_H = 0, m_seq = 0, m_np = 0, m_nn = 0, m_plus = 0, m_P = 0, m_q = 0;
        size_t m_out = 0;
        // the record's places in the packed bases, the qualities and the text: the scanned sums of its group + a wave scan of the sizes
        uint32_t v0 = 0, v1 = 0, v2 = 0;
        if (r < n_rec) { v0 = cols[r]; v1 = cols[cstride + r]; v2 = cols[2 * (size_t)cstride + r]; }
        const uint32_t e0 = gsum[g] + wave_incl_scan(v0) - v0, e1 = gsum[gstride + g] + wave_incl_scan(v1) - v1;
        const uint32_t e2 = gsum[2 * (size_t)gstride + g] + wave_incl_scan(v2) - v2;
        if (r < n_rec) {
            const uint32_t bi = find_block(blocks, nb, r);
            const DecBlock *b = &blocks[bi];
            m_L = v1;
            m_out = e2;
            m_hdr = b->a_off[S_HDR] + offs[r];
            m_H = rd16(arena + m_hdr);
            m_seq = b->a_off[S_SEQ] + (e0 - bbase[bi]);
            m_np = b->a_off[S_NPOS] + offs[2 * (size_t)ostride + r];
            m_nn = rd16(arena + m_np);
            if (b->raw_len[S_PLUS]) { m_plus = b->a_off[S_PLUS] + offs[ostride + r]; m_P = rd16(arena + m_plus); }
            m_q = b->a_off[S_QUAL] + (e1 - bbase[bstride + bi]);
        }
        const uint32_t o32 = (uint32_t)m_out; // a batch decodes to < 4 GiB (checked on the host before the launch)
        const uint32_t rec_end = o32 + (r < n_rec ? m_H + m_P + 2 * m_L + 6 : 0u);
        const uint32_t n_here = n_rec - g * ASM_G < ASM_G ? n_rec - g * ASM_G : ASM_G;
        // trips over the group's records: as many consecutive records as the staging window holds (one that does not fit alone
        // - a long read - is stored directly, as every record used to be)
        for (uint32_t lo = 0; lo < n_here;) {
        const uint32_t o_lo = (uint32_t)__builtin_amdgcn_readlane((int)o32, (int)lo), bias = o_lo & ~15u;
        const unsigned long long fits = __ballot(lane >= lo && lane < n_here && rec_end - bias <= ASM_W);
        const unsigned long long stop = ~fits & (~0ull << lo);
        uint32_t hi = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
        hi = hi < n_here ? hi : n_here;
        const bool staged = ASM_W && out_aligned && hi > lo;
        if (!(ASM_W && out_aligned)) hi = n_here; // (no staging at all: the whole group in one trip)
        else if (hi == lo) hi = lo + 1;
        const uint32_t o_hi = (uint32_t)__builtin_amdgcn_readlane((int)rec_end, (int)(hi - 1));
        const bool in = lane >= lo && lane < hi; // this lane's record belongs to the trip
#define ASM_PUT(X, W4, NB) do { if (staged) store_piece(stg + ((X) - bias), (W4), (NB)); else store_piece(out + (X), (W4), (NB)); } while (0)
        const uint32_t pq = in ? (m_L + 15) >> 4 : 0u, ph = in ? (m_H + 15) >> 4 : 0u, pp = in ? (m_P + 15) >> 4 : 0u;
        const uint32_t iq = wave_incl_scan(pq), ih = wave_incl_scan(ph), ip = wave_incl_scan(pp);
        const PieceMap pm_iq = piece_map_make(pq, iq), pm_ih = piece_map_make(ph, ih), pm_ip = piece_map_make(pp, ip);
        const uint32_t Tq = DRL(iq, 63), Th = DRL(ih, 63), Tp = DRL(ip, 63);

        // Every phase issues the loads of two rounds of 64 pieces before it uses the first one (0.77 -> 0.70 ms, measured
        // A/B on one box; boxes differ by more than that).
        // ---- header and plus payloads
        auto copy_lines = [&](const PieceMap &pm, uint32_t incl, uint32_t cnt, uint32_t T, uint32_t m_len, uint32_t m_src, uint32_t m_dst) {
            for (uint32_t base = 0; base < T; base += 2 * WAVE) {
                uint4 v[2];
                uint32_t d[2], nb[2];
                bool on[2];
#pragma unroll
                for (uint32_t u = 0; u < 2; u++) {
                    const uint32_t p = base + u * WAVE + lane;
                    on[u] = p < T;
                    if (base + u * WAVE >= T) continue; // (uniform)
                    uint32_t i, k;
                    piece_locate(pm, incl, cnt, on[u] ? p : 0, &i, &k);
                    const uint32_t len = DSH(m_len, i), src = DSH(m_src, i);
                    d[u] = DSH(m_dst, i) + 16 * k;
                    nb[u] = len - 16 * k < 16 ? len - 16 * k : 16;
                    if (on[u]) v[u] = load_u128_unaligned(arena + src + 2 + 16 * k);
                }
#pragma unroll
                for (uint32_t u = 0; u < 2; u++) {
                    if (base + u * WAVE < T && on[u]) {
                        const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        ASM_PUT(d[u], x, nb[u]);
                    }
                }
            }
        };
        copy_lines(pm_ih, ih, ph, Th, m_H, m_hdr, o32 + 1);
        copy_lines(pm_ip, ip, pp, Tp, m_P, m_plus, o32 + 4 + m_H + m_L);
        // ---- bases and qualities share the piece map
        // bases: 4 packed bytes -> 16 ASCII bases (unpackLookupUint32, sequence.go:36-42)
        // quality: inclusive prefix sum mod 256 (DeltaDecode, quality.go:105-118) + offset (DenormalizeQuality)
        {
            struct PieceJob { bool on; uint32_t k, nbytes, dseq, dq, vs, xs[4], xq[4]; uint4 vq; };
            uint32_t carry = 0; // sum of the pieces of the record in progress that earlier rounds handled
            auto fetch = [&](uint32_t p, PieceJob &J) { // every lane of the wave calls this
                J.on = p < Tq;
                uint32_t i;
                piece_locate(pm_iq, iq, pq, J.on ? p : 0, &i, &J.k);
                const uint32_t Li = DSH(m_L, i), Hi = DSH(m_H, i), Pi = DSH(m_P, i), dst = DSH(o32, i);
                const uint32_t ss = DSH(m_seq, i), sq = DSH(m_q, i);
                J.nbytes = Li - 16 * J.k < 16 ? Li - 16 * J.k : 16;
                J.dseq = dst + 2 + Hi + 16 * J.k;
                J.dq = dst + 5 + Hi + Li + Pi + 16 * J.k;
                J.vs = 0;
                J.vq = make_uint4(0, 0, 0, 0);
                if (J.on) {
                    J.vs = load_u32_unaligned(arena + ss + 4 * J.k);
                    J.vq = load_u128_unaligned(arena + sq + 16 * J.k);
                }
            };
            // registers only: vs -> 16 ASCII bases in xs, vq -> the 16 quality bytes in xq
            auto compute = [&](PieceJob &J) {
                const uint32_t k = J.k, nbytes = J.on ? J.nbytes : 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t b8 = (J.vs >> (8 * q)) & 0xFF;
                    const uint32_t nib = (b8 | (b8 << 12)) & 0x000F000Fu;        // bases 0,1 | bases 2,3
                    const uint32_t codes = (nib | (nib << 6)) & 0x03030303u;     // byte j = 2-bit code of base j
                    J.xs[q] = __builtin_amdgcn_perm(0u, 0x54474341u, codes);                      // "ACGT"[code]
                }
                uint32_t x[4] = {0, 0, 0, 0};
                if (J.on) {
                    x[0] = J.vq.x; x[1] = J.vq.y; x[2] = J.vq.z; x[3] = J.vq.w;
                    // bytes past the read must not enter the sums
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (nbytes < 4u * q + 4) x[q] = nbytes > 4u * q ? x[q] & ((1u << (8 * (nbytes - 4u * q))) - 1) : 0u;
                    // inclusive prefix inside the piece
                    uint32_t run = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        x[q] = add_bytes(x[q], x[q] << 8);
                        x[q] = add_bytes(x[q], x[q] << 16);
                        x[q] = add_bytes(x[q], run * 0x01010101u);
                        run = x[q] >> 24;
                    }
                }
                const uint32_t tot = x[3] >> 24;
                const uint32_t incl = wave_incl_scan(tot), excl = incl - tot;
                const uint32_t head_excl = DSH(excl, lane >= k ? lane - k : 0);
                const uint32_t before = (lane >= k ? excl - head_excl : carry + excl) & 0xFF;
                const uint32_t add = ((qoff + before) & 0xFF) * 0x01010101u;
#pragma unroll
                for (int q = 0; q < 4; q++) J.xq[q] = add_bytes(x[q], add);
                carry = DRL(before + tot, 63) & 0xFF; // only read by lanes whose record started before the next round
            };
            // stores last, when every load of the trip has been consumed (stores count in vmcnt like loads: a wait for the
            // second round's loads behind the first round's stores would wait for those too)
            auto commit = [&](PieceJob &J) {
                if (J.on) {
                    ASM_PUT(J.dseq, J.xs, J.nbytes);
                    ASM_PUT(J.dq, J.xq, J.nbytes);
                }
            };
            for (uint32_t base = 0; base < Tq; base += 2 * WAVE) {
                PieceJob A, B;
                fetch(base + lane, A);
                const bool two = base + WAVE < Tq;
                if (two) fetch(base + WAVE + lane, B);
                compute(A);
                if (two) compute(B);
                commit(A);
                if (two) commit(B);
            }
        }
        // ---- the fixed bytes of the record: '@', the three '\n' after header / sequence / plus, '+', the final '\n'.
        // Last, when the lines around them have just been written and are still in L2: stored first (before the pieces),
        // each of these single bytes sent a 128-byte line to HBM on its own (WRITE_SIZE 1.75 x the text).
        if (r < n_rec && in) {
            if (staged) { // (the same six stores; spelled out twice so that each side keeps its address space)
                uint8_t *q = stg + (o32 - bias);
                q[0] = '@'; q[1 + m_H] = '\n'; q[2 + m_H + m_L] = '\n'; q[3 + m_H + m_L] = '+'; q[4 + m_H + m_L + m_P] = '\n'; q[5 + m_H + 2 * m_L + m_P] = '\n';
            } else {
                uint8_t *o = out + m_out;
                o[0] = '@';
                o[1 + m_H] = '\n';
                o[2 + m_H + m_L] = '\n';
                o[3 + m_H + m_L] = '+';
                o[4 + m_H + m_L + m_P] = '\n';
                o[5 + m_H + 2 * m_L + m_P] = '\n';
            }
        }
        if (staged) { // the window leaves: the bytes in front of the first aligned unit, whole 16-byte units, the bytes behind the last
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t a_lo = (o_lo + 15) & ~15u, a_hi = o_hi & ~15u;
            const uint32_t head_end = a_lo < o_hi ? a_lo : o_hi;
            if (lane < head_end - o_lo) out[o_lo + lane] = stg[o_lo - bias + lane];
            for (uint32_t i = a_lo + 16 * lane; i < a_hi; i += 16 * WAVE) *(uint4 *)(out + i) = *(const uint4 *)(stg + (i - bias));
            if (a_hi >= a_lo && lane < o_hi - a_hi) out[a_hi + lane] = stg[a_hi - bias + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- N overlay (rare): after the bases of this group are stored
        unsigned long long todo = __ballot(in && m_nn != 0);
        if (todo) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        while (todo) {
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t L = DRL(m_L, i), nn = DRL(m_nn, i);
            const uint8_t *np = arena + DRL(m_np, i);
            uint8_t *os = out + DSH(o32, i) + 2 + DRL(m_H, i);
            for (uint32_t k = lane; k < nn; k += WAVE) {
                uint32_t pz = rd16(np + 2 + 2 * k);
                if (pz >= L) dec_fail(info, FQZ_E_NPOS_RANGE); else os[pz] = 'N';
            }
        }
        lo = hi;
        } // trips
#undef ASM_PUT
    }
}

// ===========================================================================
// scans (same three-phase scheme as the encoder; kept local to this TU)
// ===========================================================================
#define DSCAN_ITEMS 16
#define DSCAN_TILE (256 * DSCAN_ITEMS)
__global__ __launch_bounds__(256) void k_dscan_reduce(const uint32_t *data, const uint32_t *n_ptr, uint32_t col_stride, uint32_t *partials,
                                                      uint32_t pstride)
{
    __shared__ uint32_t sh[4];
    uint32_t n = *n_ptr, base = blockIdx.x * DSCAN_TILE;
    if (base >= n) return;
    const uint32_t *col = data + (size_t)blockIdx.y * col_stride;
    uint32_t i0 = base + threadIdx.x * DSCAN_ITEMS, s = 0;
#pragma unroll
    for (int k = 0; k < DSCAN_ITEMS; k++) { uint32_t i = i0 + k; s += (i < n) ? col[i] : 0u; }
    uint32_t tot;
    (void)block_excl_scan_256(s, sh, &tot);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * pstride + blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_dscan_top(uint32_t *data, const uint32_t *n_ptr, uint32_t col_stride, uint32_t *partials, uint32_t pstride)
{
    __shared__ uint32_t sh[4];
    uint32_t n = *n_ptr, nwg = (n + DSCAN_TILE - 1) / DSCAN_TILE, carry = 0;
    uint32_t *p = partials + (size_t)blockIdx.x * pstride;
    for (uint32_t b = 0; b < nwg; b += 256) {
        uint32_t i = b + threadIdx.x, v = i < nwg ? p[i] : 0u, tot;
        uint32_t ex = block_excl_scan_256(v, sh, &tot);
        if (i < nwg) p[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) data[(size_t)blockIdx.x * col_stride + n] = carry;
}
__global__ __launch_bounds__(256) void k_dscan_apply(uint32_t *data, const uint32_t *n_ptr, uint32_t col_stride, const uint32_t *partials,
                                                     uint32_t pstride)
{
    __shared__ uint32_t sh[4];
    uint32_t n = *n_ptr, base = blockIdx.x * DSCAN_TILE;
    if (base >= n) return;
    uint32_t *col = data + (size_t)blockIdx.y * col_stride;
    uint32_t i0 = base + threadIdx.x * DSCAN_ITEMS, v[DSCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < DSCAN_ITEMS; k++) { uint32_t i = i0 + k; v[k] = (i < n) ? col[i] : 0u; s += v[k]; }
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(s, sh, &tot) + partials[(size_t)blockIdx.y * pstride + blockIdx.x];
#pragma unroll
    for (int k = 0; k < DSCAN_ITEMS; k++) { uint32_t i = i0 + k; if (i < n) col[i] = ex; ex += v[k]; }
}

// ===========================================================================
// host side
// ===========================================================================
// The pipeline needs two small read-backs (block count, then frame sizes) before the bulk
// kernels can be sized; both happen inside launch.  finish() only waits for the tail.
static int dec_launch(fqz_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t version, int qual_encoding, uint8_t *d_out, size_t out_cap,
                      hipStream_t st, bool general);

int fqz_dec_launch(fqz_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t version, int qual_encoding, uint8_t *d_out, size_t out_cap,
                   hipStream_t st)
{
    const char *fg = getenv("FQZ_DEC_GENERAL"); // test hook (read on every call: tests switch it)
    const bool force_general = fg && atoi(fg);
    return dec_launch(ctx, d_in, n_bytes, version, qual_encoding, d_out, out_cap, st, force_general);
}

static int dec_launch(fqz_ctx *ctx, const uint8_t *d_in, size_t n_bytes, uint8_t version, int qual_encoding, uint8_t *d_out, size_t out_cap,
                      hipStream_t st, bool general)
{
    DecState &d = ctx->dec;
    if (d.in_flight) return FQZ_E_ARG;
    if (version != FQZ_VERSION1 && version != FQZ_VERSION2 && version != FQZ_VERSION3) return FQZ_E_BLOCK_VERSION;
    if (n_bytes >= 0x7FFFFFFFull) return FQZ_E_TOO_LARGE;
    if ((uintptr_t)d_in & 15) return FQZ_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    if ((rc = d.info.ensure(sizeof(DecInfo)))) return rc;
    if ((rc = d.h_info.ensure(sizeof(DecInfo)))) return rc;
    DecInfo *info = d.info.as<DecInfo>();
    DecInfo *hi = d.h_info.as<DecInfo>();
    const uint32_t n = (uint32_t)n_bytes;
    d.d_in = d_in; d.n_bytes = n_bytes; d.stream = st;
    d.version = version; d.qual_encoding = qual_encoding; d.user_out = d_out; d.user_cap = out_cap; d.general = general;

    // ---- block table and payload sizes.  Our own payloads (and any payload whose first frame states its size) need no walk
    //      for that, so the table is filled optimistically for up to cap0 blocks in ONE pass (one read-back instead of two);
    //      a container with more blocks, and the general path, count the blocks first.
    uint32_t nb = 0, n_rec = 0, fgrid = 0;
    DecBlock *blocks = nullptr;
    bool have_table = false;
    if (!general) {
        const uint32_t cap0 = 4096;
        if ((rc = d.blocks.ensure(sizeof(DecBlock) * (size_t)cap0))) return rc;
        if ((rc = d.h_blocks.ensure(sizeof(DecBlock) * (size_t)cap0))) return rc;
        blocks = d.blocks.as<DecBlock>();
        HIP_TRY(hipMemsetAsync(info, 0, sizeof(DecInfo), st));
        HIP_TRY(hipMemsetAsync(blocks, 0, sizeof(DecBlock) * (size_t)cap0, st));
        if (d.hint_off && d.hint_n && d.hint_n <= cap0) { // the caller knows where the blocks start: no walk along the chain
            if ((rc = d.hint.ensure(8 * d.hint_n))) return rc;
            if ((rc = d.h_hint.ensure(8 * d.hint_n))) return rc;
            memcpy(d.h_hint.p, d.hint_off, 8 * d.hint_n);
            HIP_TRY(hipMemcpyAsync(d.hint.p, d.h_hint.p, 8 * d.hint_n, hipMemcpyHostToDevice, st));
            PROF(ctx, st, "k_dec_blocks", hipLaunchKernelGGL(k_dec_blocks_hint, dim3(1), dim3(256), 0, st, d_in, n, (uint32_t)version, info, blocks, cap0,
                                                             d.hint.as<unsigned long long>(), (uint32_t)d.hint_n));
        } else
        PROF(ctx, st, "k_dec_blocks", hipLaunchKernelGGL(k_dec_blocks, dim3(1), dim3(64), 0, st, d_in, n, (uint32_t)version, info, blocks, cap0));
        PROF(ctx, st, "k_dec_fhdr", hipLaunchKernelGGL(k_dec_fhdr, dim3((cap0 * FQZ_NS + 63) / 64), dim3(64), 0, st, d_in, info, blocks, cap0));
        const uint32_t pre = 256; // (the first blocks travel with the counters: one round trip for ordinary batches)
        HIP_TRY(hipMemcpyAsync(hi, info, sizeof(DecInfo), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(d.h_blocks.p, blocks, sizeof(DecBlock) * (size_t)pre, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hi->status) return hi->status;
        nb = hi->n_blocks; n_rec = hi->n_rec;
        if (nb && nb <= cap0) {
            if (nb > pre) {
                HIP_TRY(hipMemcpyAsync(d.h_blocks.as<DecBlock>() + pre, blocks + pre, sizeof(DecBlock) * (size_t)(nb - pre), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            have_table = true;
        }
    } else {
        // ---- pass A: count blocks
        HIP_TRY(hipMemsetAsync(info, 0, sizeof(DecInfo), st));
        PROF(ctx, st, "k_dec_blocks", hipLaunchKernelGGL(k_dec_blocks, dim3(1), dim3(64), 0, st, d_in, n, (uint32_t)version, info, (DecBlock *)nullptr, 0u));
        HIP_TRY(hipMemcpyAsync(hi, info, sizeof(DecInfo), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hi->status) return hi->status;
        nb = hi->n_blocks; n_rec = hi->n_rec;
    }
    d.n_blocks = nb;
    if (!nb) { // empty container body (compress.go:591-593)
        memset(hi, 0, sizeof *hi);
        d.d_out = d_out; d.out_cap = out_cap; d.in_flight = true;
        return FQZ_OK;
    }
    fgrid = nb * FQZ_NS; // one wave per (block, stream)
    if (!have_table) {
        // ---- pass B: block table + frame walk (sizes)
        if ((rc = d.blocks.ensure(sizeof(DecBlock) * (size_t)nb))) return rc;
        if ((rc = d.h_blocks.ensure(sizeof(DecBlock) * (size_t)nb))) return rc;
        blocks = d.blocks.as<DecBlock>();
        HIP_TRY(hipMemsetAsync(info, 0, sizeof(DecInfo), st));
        HIP_TRY(hipMemsetAsync(blocks, 0, sizeof(DecBlock) * (size_t)nb, st));
        PROF(ctx, st, "k_dec_blocks", hipLaunchKernelGGL(k_dec_blocks, dim3(1), dim3(64), 0, st, d_in, n, (uint32_t)version, info, blocks, nb));
        if (general) PROF(ctx, st, "k_dec_frames", hipLaunchKernelGGL(k_dec_frames, dim3(fgrid), dim3(FRAME_NT), 0, st, d_in, n, info, blocks, (DecChunk *)nullptr, (DecFrame *)nullptr, 0, version == FQZ_VERSION3 ? 1 : 0, (uint2 *)nullptr, 0u));
        else PROF(ctx, st, "k_dec_fhdr", hipLaunchKernelGGL(k_dec_fhdr, dim3((fgrid + 63) / 64), dim3(64), 0, st, d_in, info, blocks, nb));
        HIP_TRY(hipMemcpyAsync(hi, info, sizeof(DecInfo), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(d.h_blocks.p, blocks, sizeof(DecBlock) * (size_t)nb, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (hi->status) return hi->status;
    if (!general) { // a payload whose frame header does not tell its size: take the two-walk path
        const DecBlock *hb0 = d.h_blocks.as<DecBlock>();
        for (uint32_t b = 0; b < nb; b++)
            for (int s = 0; s < FQZ_NS; s++)
                if (hb0[b].raw_len[s] == FQZ_SPEC_UNKNOWN) return dec_launch(ctx, d_in, n_bytes, version, qual_encoding, d_out, out_cap, st, true);
    }
    // ---- layout on the host: arena offsets, chunk bases, output bound
    DecBlock *hb = d.h_blocks.as<DecBlock>();
    unsigned long long arena = 0, chunks = 0, out_bound = 0, tiles = 0, n_lz = 0, nframes = 0;
    bool any_indexed = false;
    for (int s = 0; s < FQZ_NS; s++) { hi->stream_raw[s] = 0; hi->stream_comp[s] = 0; }
    // chunk numbers: the quality streams of all blocks first, then the other streams block by block - the two entropy launches
    // (qualities on the side stream, the rest on the caller's) then each cover a dense range
    unsigned long long qchunks = 0;
    for (uint32_t b = 0; b < nb; b++) { hb[b].chunk_base[S_QUAL] = (uint32_t)qchunks; qchunks += hb[b].n_chunks[S_QUAL]; }
    for (uint32_t b = 0; b < nb; b++) {
        for (int s = 0; s < FQZ_NS; s++) {
            hb[b].a_off[s] = (uint32_t)arena;
            arena += ((unsigned long long)hb[b].raw_len[s] + 15 + 16) & ~15ull; // +16: tile loads read whole uint4
            if (s != S_QUAL) { hb[b].chunk_base[s] = (uint32_t)(qchunks + chunks); chunks += hb[b].n_chunks[s]; }
            hb[b].frame_base[s] = (uint32_t)nframes;
            nframes += hb[b].n_frames[s];
            if (general) { hb[b].indexed[s] = 0; hb[b].samp_off[s] = 0; hb[b].ent_off[s] = 0; } // the general path walks every payload and every chain, index or not
            any_indexed |= hb[b].indexed[s] != 0;
            if (hb[b].lz[s]) hb[b].lz[s] = (uint32_t)++n_lz; // 1 + scratch slot
            hi->stream_raw[s] += hb[b].raw_len[s];
            hi->stream_comp[s] += hb[b].pay_len[s];
        }
        {
            const int ws[3] = {S_HDR, S_PLUS, S_NPOS};
            for (int w = 0; w < 3; w++) { hb[b].tile_base[w] = (uint32_t)tiles; tiles += (hb[b].raw_len[ws[w]] + WALK_TILE - 1) / WALK_TILE; hb[b].walk_mode[w] = 0;
                const unsigned long long avg = hb[b].nrec ? hb[b].raw_len[ws[w]] / hb[b].nrec : 0; // bytes per record incl. the 2-byte prefix
                hb[b].walk_cand[w] = avg + 12 <= 64 ? 64u : (avg + 24 <= 128 ? 128u : WALK_ENTRIES); }
        }
        out_bound += (unsigned long long)hb[b].raw_len[S_HDR] + hb[b].raw_len[S_PLUS] + 2ull * hb[b].raw_len[S_QUAL] + 4ull * hb[b].nrec;
    }
    uint32_t max_sgroups = 0; // groups of 64 records of the largest block whose index carries record samples
    for (uint32_t b = 0; b < nb; b++)
        for (int s = 0; s < FQZ_NS; s++)
            if (hb[b].samp_off[s] && (hb[b].nrec + 63) / 64 > max_sgroups) max_sgroups = (hb[b].nrec + 63) / 64;
    bool any_seq = false; // scratch for the blocks with sequences of our own headers streams (fqz_decode_seq.h), behind the streams
    for (uint32_t b = 0; b < nb; b++) {
        hb[b].seq_scratch = 0;
        hb[b].seq_scratch_len = 0;
        if (hb[b].indexed[S_HDR] && hb[b].n_chunks[S_HDR]) {
            hb[b].seq_scratch = (uint32_t)arena;
            arena += (unsigned long long)hb[b].n_chunks[S_HDR] * DSEQ_STRIDE;
            any_seq = true;
        }
        if (hb[b].indexed[S_LEN] && hb[b].n_chunks[S_LEN]) {
            hb[b].seq_scratch_len = (uint32_t)arena;
            arena += (unsigned long long)hb[b].n_chunks[S_LEN] * DSEQ_STRIDE;
            any_seq = true;
        }
    }
    // NumRecords comes from the (untrusted) block header: tie it to the decoded stream sizes before anything is sized from it.
    // Order as blockReader.writeRecord meets them (compress.go:944-975): length, N positions, header.
    for (uint32_t b = 0; b < nb; b++) {
        const unsigned long long nr = hb[b].nrec;
        if (4ull * nr > hb[b].raw_len[S_LEN]) return FQZ_E_TRUNC_LEN;
        if (2ull * nr > hb[b].raw_len[S_NPOS]) return FQZ_E_TRUNC_NPOS;
        if (2ull * nr > hb[b].raw_len[S_HDR]) return FQZ_E_TRUNC_HEADER;
        if (hb[b].raw_len[S_PLUS] && 2ull * nr > hb[b].raw_len[S_PLUS]) return FQZ_E_TRUNC_PLUS;
    }
    chunks += qchunks;
    if (arena > 0xFFFFFFF0ull || chunks > 0x7FFFFFFFull || out_bound > 0xFFFFFFF0ull || tiles > 0x7FFFFFFFull || nframes > 0x7FFFFFFFull) return FQZ_E_TOO_LARGE;
    const uint32_t n_tiles = (uint32_t)tiles;
    const bool skip_assemble = d.skip_assemble;
    if (!d_out && !skip_assemble) { // host-buffer entry points: decode into the context's staging buffer
        if ((rc = ctx->d_out.ensure((size_t)out_bound + 64))) return rc;
        d_out = ctx->d_out.as<uint8_t>();
        out_cap = (size_t)out_bound + 64;
    }
    if (skip_assemble) out_cap = 0xFFFFFFF0ull; // nothing is written: only the total is reported
    d.d_out = d_out; d.out_cap = out_cap;
    const uint32_t n_chunks = (uint32_t)chunks, n_q = (uint32_t)qchunks, n_o = n_chunks - n_q; // all / quality / other chunks
    const uint32_t ostride = n_rec + 1, cstride = n_rec + 1;
    if ((rc = d.streams.ensure((size_t)arena + 64))) return rc;
    if (n_lz && (rc = d.lz_scratch.ensure((size_t)n_lz * LZ_SCRATCH))) return rc;
    if ((rc = d.chunks.ensure(sizeof(DecChunk) * ((size_t)n_chunks + 1)))) return rc;
    const uint32_t n_frames = (uint32_t)nframes;
    if ((rc = d.frames.ensure((sizeof(DecFrame) + sizeof(uint2)) * ((size_t)n_frames + 1)))) return rc; // frames | groups with rANS blocks
    DecFrame *dfr = d.frames.as<DecFrame>();
    uint2 *rlist = version == FQZ_VERSION3 ? (uint2 *)(dfr + (size_t)n_frames + 1) : nullptr; // (version-3 files only: FQZ-R1, fqz_rans.h)
    const uint32_t n_groups = (n_rec + 63) / 64, gstride = n_groups + 2, bstride = nb + 1;
    if ((rc = d.rec.ensure(4ull * (3ull * ostride + 3ull * cstride + 3ull * gstride + 2ull * bstride) + 64))) return rc;
    uint32_t pmax = n_rec / DSCAN_TILE + 2;
    if ((rc = d.partials.ensure(4ull * 3 * pmax + 24ull * ((size_t)nb + 1)))) return rc;
    if ((rc = d.tables.ensure(((size_t)n_tiles + 1) * (WALK_ENTRIES * sizeof(WalkF) + sizeof(WalkEntry))))) return rc;
    WalkF *walkF = d.tables.as<WalkF>();
    WalkEntry *walkE = (WalkEntry *)(walkF + ((size_t)n_tiles + 1) * WALK_ENTRIES);
    DecChunk *dch = d.chunks.as<DecChunk>();
    uint8_t *darena = d.streams.as<uint8_t>();
    uint32_t *offs = d.rec.as<uint32_t>(), *cols = offs + 3ull * ostride, *partials = d.partials.as<uint32_t>();
    uint32_t *gsum = cols + 3ull * cstride, *bbase = gsum + 3ull * gstride; // sums per 64 records (scanned in place), column values at the blocks' first records
    unsigned long long *btot = (unsigned long long *)(partials + ((3ull * pmax + 1) & ~1ull)); // 64-bit column totals per block
    HIP_TRY(hipMemcpyAsync(blocks, hb, sizeof(DecBlock) * (size_t)nb, hipMemcpyHostToDevice, st));
    hi->n_chunks = n_chunks;
    // ---- bulk kernels (the first one also sets info->n_chunks and clears the block totals)
    PROF(ctx, st, "k_dec_frames", hipLaunchKernelGGL(k_dec_frames, dim3(fgrid), dim3(FRAME_NT), 0, st, d_in, n, info, blocks, dch, dfr, general ? 1 : 2, version == FQZ_VERSION3 ? 1 : 0, rlist, n_frames,
                                                     n_chunks, btot, 3u * nb));
    if (any_indexed) PROF(ctx, st, "k_dec_index", hipLaunchKernelGGL(k_dec_index, dim3(fgrid), dim3(256), 0, st, d_in, info, blocks, dch, dfr, rlist, n_frames));
    // The record walks and the size scans below need only the header / plus / nPos / lengths streams, the text assembly
    // at the end needs the bases and qualities too.  The latter are 3/4 of the entropy decode and the walks leave the
    // chip almost empty, so the two run side by side: bases + qualities on the context's side stream, joined before
    // k_dec_assemble (285 -> 295 GB/s over 30 decodes, A/B on one box).
    // (the packed bases are Raw blocks in our files - a plain copy of a tenth of the text, needed by the assembly only: it rides
    //  on the second side stream, behind the sequences' bit streams, off the chain that leads to the record walks)
    const uint32_t early = (1u << S_HDR) | (1u << S_PLUS) | (1u << S_NPOS) | (1u << S_LEN) | (1u << S_SEQ), late = 1u << S_QUAL;
    bool forked = false;
    if (n_chunks) {
        const int dbg = getenv("FQZ_DBG_DEC") ? atoi(getenv("FQZ_DBG_DEC")) : 0;
        if (!d.side) {
            // (a CU mask that keeps the qualities' long Huffman decode off a part of the chip: 406 - 446 against 498 GB/s; the lowest
            //  stream priority for it: 320 against 390 GB/s; holding it back behind the headers chain: 457 against 520 GB/s)
            HIP_TRY(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_joinx, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_join, hipEventDisableTiming));
            HIP_TRY(hipStreamCreateWithFlags(&d.side2, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_join2, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_x, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_huf, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&d.ev_seq, hipEventDisableTiming));
        }
        static const bool dbg_serial = getenv("FQZ_DBG_SERIAL") && atoi(getenv("FQZ_DBG_SERIAL")); // diagnostic runs: one stream (standalone kernel times)
        const hipStream_t sd = dbg_serial ? st : d.side;
        HIP_TRY(hipEventRecord(d.ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(d.side, d.ev_fork, 0));
        const hipStream_t s2 = dbg_serial ? st : d.side2;
        const uint32_t seqm = 1u << S_SEQ;
        HIP_TRY(hipStreamWaitEvent(d.side2, d.ev_fork, 0));
        if (any_seq) { // the sequence bit streams need the input only: a chain of serial steps, beside the literals' Huffman decode
            PROF(ctx, s2, "k_dec_seq_fse", hipLaunchKernelGGL(k_dec_seq_fse, dim3((n_o + 15) / 16), dim3(64), 0, s2, d_in, info, dch, darena, n_q));
            HIP_TRY(hipEventRecord(d.ev_join2, d.side2));
        }
        if (n_o) {
            // (an indexed batch of our own: four lanes per Huffman stream, on the index's entry points)
            if (!general) PROF(ctx, st, "k_dec_huf", hipLaunchKernelGGL(k_dec_huf<4>, dim3((n_o + 3) / 4), dim3(64), 0, st, d_in, info, dch, darena, dbg, early, n_q, n_o));
            else PROF(ctx, st, "k_dec_huf", hipLaunchKernelGGL(k_dec_huf<1>, dim3((n_o + HG - 1) / HG), dim3(64), 0, st, d_in, info, dch, darena, dbg, early, n_q, n_o));
            PROF(ctx, st, "k_dec_plain", hipLaunchKernelGGL(k_dec_plain, dim3(n_o < DENT_GRID ? n_o : DENT_GRID), dim3(64), 0, st, d_in, info, dch, darena, early & ~seqm, n_q, n_o));
            if (general) PROF(ctx, st, "k_dec_entropy", hipLaunchKernelGGL(k_dec_entropy, dim3(n_o < DENT_GRID ? n_o : DENT_GRID), dim3(64), 0, st, d_in, info, dch, darena, early & ~seqm, n_q, n_o));
            // the bases: behind the Huffman launch above (which takes what a foreign file has Huffman-coded of them and marks it done)
            HIP_TRY(hipEventRecord(d.ev_huf, st));
            HIP_TRY(hipStreamWaitEvent(d.side2, d.ev_huf, 0));
            PROF(ctx, s2, "k_dec_plain", hipLaunchKernelGGL(k_dec_plain, dim3(n_o < DENT_GRID ? n_o : DENT_GRID), dim3(64), 0, s2, d_in, info, dch, darena, seqm, n_q, n_o));
            if (general) PROF(ctx, s2, "k_dec_entropy", hipLaunchKernelGGL(k_dec_entropy, dim3(n_o < DENT_GRID ? n_o : DENT_GRID), dim3(64), 0, s2, d_in, info, dch, darena, seqm, n_q, n_o));
        }
        HIP_TRY(hipEventRecord(d.ev_seq, d.side2));
        if (any_seq) { // headers blocks with sequences: their literals are in the scratch now, their triples come from side2
            HIP_TRY(hipStreamWaitEvent(st, d.ev_join2, 0));
            PROF(ctx, st, "k_dec_seq_exec", hipLaunchKernelGGL(k_dec_seq_exec, dim3(n_o), dim3(64), 0, st, info, dch, darena, n_q));
        }
        // (holding the qualities' decode back until the sequence bit streams are through measured 380 against 418 GB/s; the same for
        //  the rANS decode of version 3 held back behind the early streams' Huffman decode: 460 against 543 GB/s)
        if (n_q && rlist) { // version 3: the qualities are rANS blocks, a wave per group (at most one group per frame)
            const uint32_t rg = (n_q + FQZ_GROUP - 1) / FQZ_GROUP + nb < n_frames ? (n_q + FQZ_GROUP - 1) / FQZ_GROUP + nb : n_frames;
            PROF(ctx, sd, "k_dec_rans", hipLaunchKernelGGL(k_dec_rans, dim3(rg ? rg : 1), dim3(64), 0, sd, d_in, info, dch, rlist, n_frames, darena));
        }
        if (n_q) {
            if (!general) PROF(ctx, sd, "k_dec_huf", hipLaunchKernelGGL(k_dec_huf<4>, dim3((n_q + 3) / 4), dim3(64), 0, sd, d_in, info, dch, darena, dbg, late, 0u, n_q));
            else PROF(ctx, sd, "k_dec_huf", hipLaunchKernelGGL(k_dec_huf<1>, dim3((n_q + HG - 1) / HG), dim3(64), 0, sd, d_in, info, dch, darena, dbg, late, 0u, n_q));
            PROF(ctx, sd, "k_dec_plain", hipLaunchKernelGGL(k_dec_plain, dim3(n_q < DENT_GRID ? n_q : DENT_GRID), dim3(64), 0, sd, d_in, info, dch, darena, late, 0u, n_q));
            if (general) PROF(ctx, sd, "k_dec_entropy", hipLaunchKernelGGL(k_dec_entropy, dim3(n_q < DENT_GRID ? n_q : DENT_GRID), dim3(64), 0, sd, d_in, info, dch, darena, late, 0u, n_q));
        }
        HIP_TRY(hipEventRecord(d.ev_join, d.side)); // the qualities are decoded: the text can be assembled
        if (n_frames) { // content checksums of the decoded frames, all on the side stream: nothing needs them before the verdict at the
            // end, so they run beside the text assembly (367 -> 390 GB/s, A/B on one box; the early streams are complete once the
            // sequences have been executed: ev_x)
            HIP_TRY(hipEventRecord(d.ev_x, st));
            PROF(ctx, sd, "k_dec_xxh", hipLaunchKernelGGL(k_dec_xxh, dim3((n_frames + DXXH_PER_WAVE - 1) / DXXH_PER_WAVE), dim3(64), 0, sd, d_in, info, dfr, n_frames, darena, late));
            HIP_TRY(hipStreamWaitEvent(d.side, d.ev_x, 0));
            HIP_TRY(hipStreamWaitEvent(d.side, d.ev_seq, 0));
            PROF(ctx, sd, "k_dec_xxh", hipLaunchKernelGGL(k_dec_xxh, dim3((n_frames + DXXH_PER_WAVE - 1) / DXXH_PER_WAVE), dim3(64), 0, sd, d_in, info, dfr, n_frames, darena, early));
        }
        HIP_TRY(hipEventRecord(d.ev_joinx, d.side));
        forked = true;
    }
    if (n_lz) PROF(ctx, st, "k_dec_lz", hipLaunchKernelGGL(k_dec_lz, dim3(fgrid), dim3(64), 0, st, d_in, info, blocks, darena, d.lz_scratch.as<uint8_t>()));
    PROF(ctx, st, "k_dec_walk0", hipLaunchKernelGGL(k_dec_walk0, dim3(nb * 3), dim3(1024), 0, st, darena, info, blocks, offs, ostride));
    if (max_sgroups) {
        const uint32_t xper = (max_sgroups + 63) / 64;
        PROF(ctx, st, "k_dec_walk_s", hipLaunchKernelGGL(k_dec_walk_s, dim3(xper * nb * 3), dim3(64), 0, st, d_in, darena, info, blocks, offs, ostride, xper));
    }
    // the tiled walks are only for chains that neither a rule (k_dec_walk0) nor record samples (k_dec_walk_s) resolve
    bool need_walks = false;
    for (uint32_t b = 0; b < nb && !need_walks; b++) {
        const int ws[3] = {S_HDR, S_PLUS, S_NPOS};
        for (int w = 0; w < 3; w++) {
            const int s = ws[w];
            const bool by_rule = hb[b].nrec == 0 || (unsigned long long)hb[b].raw_len[s] == 2ull * hb[b].nrec || (s == S_PLUS && hb[b].raw_len[s] == 0);
            if (!by_rule && !hb[b].samp_off[s]) need_walks = true;
        }
    }
    if (n_tiles && need_walks) {
        PROF(ctx, st, "k_dec_walk1", hipLaunchKernelGGL(k_dec_walk1, dim3(n_tiles), dim3(256), 0, st, darena, info, blocks, n_tiles, walkF));
        PROF(ctx, st, "k_dec_walk2", hipLaunchKernelGGL(k_dec_walk2, dim3(nb * 3), dim3(64), 0, st, darena, info, blocks, walkF, walkE));
        PROF(ctx, st, "k_dec_walk3", hipLaunchKernelGGL(k_dec_walk3, dim3(n_tiles), dim3(64), 0, st, darena, info, blocks, n_tiles, walkE, offs, ostride));
    }
    if (n_rec) {
        uint32_t g = (n_rec + 1023) / 1024; // (runs of >= 1024 records: one flush of the block totals per workgroup)
        if (g > 2048) g = 2048;
        PROF(ctx, st, "k_dec_sizes", hipLaunchKernelGGL(k_dec_sizes, dim3(g), dim3(256), 0, st, darena, info, blocks, offs, ostride, cols, cstride, btot, gsum, gstride));
    }
    {
        uint32_t nwg = (n_groups + DSCAN_TILE - 1) / DSCAN_TILE; // (the scans run over the group sums)
        if (!nwg) nwg = 1;
        hipLaunchKernelGGL(k_dscan_reduce, dim3(nwg, 3), dim3(256), 0, st, gsum, &info->n_groups, gstride, partials, pmax);
        hipLaunchKernelGGL(k_dscan_top, dim3(3), dim3(256), 0, st, gsum, &info->n_groups, gstride, partials, pmax);
        hipLaunchKernelGGL(k_dscan_apply, dim3(nwg, 3), dim3(256), 0, st, gsum, &info->n_groups, gstride, partials, pmax);
    }
    PROF(ctx, st, "k_dec_check", hipLaunchKernelGGL(k_dec_check, dim3(1), dim3(256), 0, st, info, blocks, btot, out_cap, cols, cstride, gsum, gstride, bbase, bstride));
    if (n_rec) {
        uint32_t g = ((n_rec + ASM_G - 1) / ASM_G + 3) / 4;
        if (g > 8192) g = 8192;
        if (!g) g = 1;
        uint32_t qoff = qual_encoding == FQZ_ENCODING_PHRED64 ? 64u : 33u;
        if (forked) { HIP_TRY(hipStreamWaitEvent(st, d.ev_join, 0)); HIP_TRY(hipStreamWaitEvent(st, d.ev_seq, 0)); forked = false; }
        if (!skip_assemble) PROF(ctx, st, "k_dec_assemble", hipLaunchKernelGGL(k_dec_assemble, dim3(g), dim3(256), 0, st, darena, info, blocks, offs, ostride, cols, cstride, qoff, d_out, gsum, gstride, bbase, bstride));
    }
    if (forked) { HIP_TRY(hipStreamWaitEvent(st, d.ev_join, 0)); HIP_TRY(hipStreamWaitEvent(st, d.ev_seq, 0)); } // (no records: nothing assembled, still join)
    if (n_chunks) HIP_TRY(hipStreamWaitEvent(st, d.ev_joinx, 0)); // the checksums: before the status is read
    HIP_TRY(hipGetLastError());
    // keep the host-side per-stream totals; status / out_len come back after the tail
    HIP_TRY(hipMemcpyAsync(&hi->status, &info->status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&hi->out_len, &info->out_len, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    hi->n_blocks = nb; hi->n_rec = n_rec;
    d.in_flight = true;
    return FQZ_OK;
}

int fqz_dec_finish(fqz_ctx *ctx, fqz_batch_result *res)
{
    DecState &d = ctx->dec;
    if (!d.in_flight) return FQZ_E_ARG;
    d.in_flight = false;
    HIP_TRY(hipStreamSynchronize(d.stream));
    const DecInfo *hi = d.h_info.as<DecInfo>();
    if (hi->status == FQZ_DEC_RETRY_GENERAL && !d.general) { // a frame was not cut the way our encoder cuts it
        int rc = dec_launch(ctx, d.d_in, d.n_bytes, d.version, d.qual_encoding, d.user_out, d.user_cap, d.stream, true);
        if (rc) return rc;
        return fqz_dec_finish(ctx, res);
    }
    if (res) {
        memset(res, 0, sizeof *res);
        res->n_records = hi->n_rec;
        res->n_blocks = hi->n_blocks;
        res->consumed = d.n_bytes;
        res->out_len = hi->status ? 0 : hi->out_len;
        res->status = hi->status;
        res->n_chunks = hi->n_chunks;
        for (int s = 0; s < FQZ_NS; s++) { res->stream_raw[s] = hi->stream_raw[s]; res->stream_comp[s] = hi->stream_comp[s]; }
    }
    return hi->status;
}

// entropy stage alone: one payload -> its pre-entropy bytes (wraps the payload in a fake block table)
int fqz_dec_entropy_only(fqz_ctx *ctx, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap, size_t *out_len, hipStream_t st)
{
    DecState &d = ctx->dec;
    if (d.in_flight || n >= 0x7FFFFFFFull) return FQZ_E_ARG;
    int rc;
    if ((rc = d.info.ensure(sizeof(DecInfo)))) return rc;
    if ((rc = d.h_info.ensure(sizeof(DecInfo)))) return rc;
    if ((rc = d.blocks.ensure(sizeof(DecBlock)))) return rc;
    if ((rc = d.h_blocks.ensure(sizeof(DecBlock)))) return rc;
    DecInfo *info = d.info.as<DecInfo>(), *hi = d.h_info.as<DecInfo>();
    DecBlock *blocks = d.blocks.as<DecBlock>(), *hb = d.h_blocks.as<DecBlock>();
    memset(hb, 0, sizeof *hb);
    hb->pay_len[S_SEQ] = (uint32_t)n;
    DecInfo z;
    memset(&z, 0, sizeof z);
    z.n_blocks = 1;
    *hi = z;
    HIP_TRY(hipMemcpyAsync(info, hi, sizeof z, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(blocks, hb, sizeof *hb, hipMemcpyHostToDevice, st));
    PROF(ctx, st, "k_dec_frames", hipLaunchKernelGGL(k_dec_frames, dim3(1), dim3(FRAME_NT), 0, st, d_src, (uint32_t)n, info, blocks, (DecChunk *)nullptr, (DecFrame *)nullptr, 0, 0, (uint2 *)nullptr, 0u));
    HIP_TRY(hipMemcpyAsync(hi, info, sizeof z, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hb, blocks, sizeof *hb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hi->status) return hi->status;
    uint32_t raw = hb->raw_len[S_SEQ], nch = hb->n_chunks[S_SEQ];
    if (raw > cap) return FQZ_E_DST_SMALL;
    if (hb->lz[S_SEQ]) { // a frame with LZ sequences (not written by this library): decoded block by block by one wave
        if ((rc = d.lz_scratch.ensure(LZ_SCRATCH))) return rc;
        hb->lz[S_SEQ] = 1;
        hb->a_off[S_SEQ] = 0;
        HIP_TRY(hipMemcpyAsync(blocks, hb, sizeof *hb, hipMemcpyHostToDevice, st));
        PROF(ctx, st, "k_dec_lz", hipLaunchKernelGGL(k_dec_lz, dim3(FQZ_NS), dim3(64), 0, st, d_src, info, blocks, d_dst, d.lz_scratch.as<uint8_t>()));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hi, info, sizeof z, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hi->status) return hi->status;
        *out_len = raw;
        return FQZ_OK;
    }
    if ((rc = d.chunks.ensure(sizeof(DecChunk) * ((size_t)nch + 1)))) return rc;
    const uint32_t nfr = hb->n_frames[S_SEQ];
    if ((rc = d.frames.ensure(sizeof(DecFrame) * ((size_t)nfr + 1)))) return rc;
    for (int s = 1; s < FQZ_NS; s++) { hb->chunk_base[s] = nch; hb->frame_base[s] = nfr; }
    hi->n_chunks = nch;
    HIP_TRY(hipMemcpyAsync(blocks, hb, sizeof *hb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(&info->n_chunks, &hi->n_chunks, 4, hipMemcpyHostToDevice, st));
    PROF(ctx, st, "k_dec_frames", hipLaunchKernelGGL(k_dec_frames, dim3(1), dim3(FRAME_NT), 0, st, d_src, (uint32_t)n, info, blocks, d.chunks.as<DecChunk>(), d.frames.as<DecFrame>(), 1, 0, (uint2 *)nullptr, 0u));
    if (nch) {
        PROF(ctx, st, "k_dec_huf", hipLaunchKernelGGL(k_dec_huf<1>, dim3((nch + HG - 1) / HG), dim3(64), 0, st, d_src, info, d.chunks.as<DecChunk>(), d_dst, 0, 0x3Fu, 0u, nch));
        PROF(ctx, st, "k_dec_plain", hipLaunchKernelGGL(k_dec_plain, dim3(nch < DENT_GRID ? nch : DENT_GRID), dim3(64), 0, st, d_src, info, d.chunks.as<DecChunk>(), d_dst, 0x3Fu, 0u, nch));
        PROF(ctx, st, "k_dec_entropy", hipLaunchKernelGGL(k_dec_entropy, dim3(nch < DENT_GRID ? nch : DENT_GRID), dim3(64), 0, st, d_src, info, d.chunks.as<DecChunk>(), d_dst, 0x3Fu, 0u, nch));
        if (nfr) PROF(ctx, st, "k_dec_xxh", hipLaunchKernelGGL(k_dec_xxh, dim3((nfr + DXXH_PER_WAVE - 1) / DXXH_PER_WAVE), dim3(64), 0, st, d_src, info, d.frames.as<DecFrame>(), nfr, d_dst, 0x3Fu));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hi, info, sizeof z, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hi->status) return hi->status;
    *out_len = raw;
    return FQZ_OK;
}
