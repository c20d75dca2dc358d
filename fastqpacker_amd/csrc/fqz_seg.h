// fqz_seg.h — the FQZ-S1 segment path of the encoder (gfx950 / MI355X): text -> frames in ONE pass over the text.
//
// Replaces, like fqz_encode.hip's group path, the record loop of compressBlockWithBuffers (internal/compress/compress.go:474-520)
// and the six EncodeAll calls (:523-528), but per SEGMENT of a block: the records whose first byte lies in a 64 KiB window of the
// block's text (oracle/fqz_entropy.c, "FQZ-S1", is the specification).  A workgroup takes one segment: it finds the lines of its
// records, splits them into the six pre-entropy stream parts IN LDS (sequence.go:139-184, quality.go:53-103, compress.go:495-519)
// and entropy-codes the quality / plus / nPos / lengths parts from there (fqz_entropy_dev.h); the line index, the record table and
// those streams never exist in HBM.  What goes to HBM besides the zstd blocks: the stream parts once, write-only on this kernel's
// side, for the content checksums (k_xxh) and the headers model (k_hdr_model_seg and the kernels behind it, unchanged).
//
//   k_count_nl + scan        newline counts per 4 KiB tile (the only other pass over the text)
//   k_seg_setup              record / block counts from the line count (parser.go:136-243: four lines a record)
//   k_seg_blocks             first byte of every block (line 4 x 100 000 x b), the end of the batch, the dangling record's checks
//   k_seg_plan               segments per block, their numbering
//   k_seg_table              first record and first byte of every segment
//   k_seg_detect             DetectEncoding over block 0 (quality.go:22-49), when asked for
//   k_seg_encode             the segment workgroup described above
//   k_hdr_model_seg ...      headers: model, sequence sections, literals (fqz_hdrlz.h), per segment
//   k_seg_sizes, k_seg_layout, k_seg_compact   frame sizes -> payload and block offsets -> the blocks in their final place
#pragma once
#include "fqz_seg_entropy.h"

#define SEG_TEXT 65536u     // == FQZO_SEG_TEXT
#define SEG_RMAX 384u       // == FQZO_SEG_RMAX
#define SEG_ARENA 51200u    // == FQZO_SEG_ARENA
#define SEG_SPAN_MAX 131072u // text bytes of a segment's records beyond which it cannot qualify (its streams are at least ~0.6 of the text)
#define SEG_NT 512u      // threads of a segment workgroup
#define SEG_NW (SEG_NT / 64u)
#define SEG_PAGE 1024u      // unit of the slot pool
#define SEG_IDS 12u         // csize / xsum entries of a segment: at most 4 + 5 zstd blocks (51200 bytes in five parts), one id per frame
#define SEG_SLOT_PAGES 60u  // slot pool pages of a segment: the sum of seg_pages over parts that fit SEG_ARENA
#define SEG_ASTRIDE (SEG_ARENA + 96u) // stream arena bytes of a segment
// slot pool pages of a part of M bytes: one chunk -> its worst-case block; several -> FQZ_SLOT apart, as the entropy coder writes them
__host__ __device__ static inline uint32_t seg_pages(uint32_t M, int s)
{
    if (!M) return 0;
    if (s == S_SEQ) return (M + 16 + SEG_PAGE - 1) / SEG_PAGE;
    const uint32_t nch = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
    return ((nch - 1) * FQZ_SLOT + (M - (nch - 1) * FQZ_CHUNK) + 64 + SEG_PAGE - 1) / SEG_PAGE;
}

struct SegInfo {                 // one per segment (+ a closing entry)
    uint32_t text_off;           // first byte of the segment's first record ( == the end of the previous segment's last record)
    uint32_t rec0;               // index of that record in the batch
    uint32_t raw[FQZ_NS];        // pre-entropy bytes of the six stream parts
    uint32_t chunk0[FQZ_NS];     // first entry of the part's zstd blocks in csize[] / of its frame in xsum[] (S_HDR: csize lives in the headers' own table)
    uint32_t slot0[FQZ_NS];      // where those blocks wait: in SEG_PAGE units of the slot pool (S_SEQ: the part's bytes - its blocks are Raw; S_HDR: the
                                 // first headers chunk ordinal - its blocks lie in the headers' slots)
    uint32_t a_off[FQZ_NS];      // the part in the stream arena (S_HDR also read by the headers model)
    uint32_t foff[FQZ_NS];       // the frame inside its payload, behind the index (k_seg_sizes)
    uint32_t flen[FQZ_NS];       // frame bytes
    uint32_t hx;                 // headers chunks 1, 2, ... of the part (rare): their ordinals start here (chunk 0 has the segment's number)
    uint32_t pad_;
};

// one headers chunk of a segment for the headers kernels: where it lies in the stream arena, its records, its ordinal in the side
// buffers, where its zstd block goes (byte offset in the slot pool) and which csize entry it has
struct SegHdrJob { uint32_t a_off, e_off, nrec, c0, mk, chunk, slot_off, cs; };

// ---------------------------------------------------------------------------------------------
// line -> byte offset from the scanned newline counts (one wave; every lane calls and gets the result)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t seg_nl_mask16(const uint4 v)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) m |= ((((zero_bytes(w[k] ^ 0x0A0A0A0Au) >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);
    return m;
}

// first byte of line `line` (0-based); 1 <= line <= tile_off[n_tiles] unless line == 0
__device__ uint32_t seg_line_start(const uint8_t *__restrict__ text, uint32_t n, const uint32_t *__restrict__ tile_off, uint32_t n_tiles, uint32_t line)
{
    if (line == 0) return 0;
    const uint32_t lane = lane_id();
    uint32_t lo = 0, hi = n_tiles; // tile_off[lo] < line <= tile_off[hi]
    while (hi - lo > 1) {
        const uint32_t step = (hi - lo + 63) / 64;
        const uint32_t p = lo + (lane + 1) * step;
        const bool below = p < hi && tile_off[p] < line;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(below)); // the probes are in order: the first cnt of them lie below
        const uint32_t nlo = lo + cnt * step;
        hi = nlo + step < hi ? nlo + step : hi;
        lo = nlo;
    }
    uint32_t k = line - tile_off[lo]; // the k-th newline of tile lo (1-based)
    const uint32_t tb = lo * FQZ_TILE;
    uint32_t res = n;
    bool found = false;
#pragma unroll 1
    for (uint32_t q = 0; q < FQZ_TILE / 1024 && !found; q++) {
        const uint32_t off = tb + q * 1024 + 16 * lane;
        uint32_t m = seg_nl_mask16(load_text16(text, off, n));
        const uint32_t c = __popc(m), incl = wave_incl_scan(c);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (k <= tot) {
            const bool mine = incl - c < k && k <= incl;
            uint32_t j = k - (incl - c); // 1-based inside this lane's mask (meaningful when mine)
            uint32_t pos = 0;
            if (mine) {
                while (--j) m &= m - 1;
                pos = off + (uint32_t)(__ffs(m) - 1) + 1;
            }
            const unsigned long long who = __ballot(mine);
            res = (uint32_t)__shfl((int)pos, (int)(__ffsll((long long)who) - 1), WAVE);
            found = true;
        } else k -= tot;
    }
    return res;
}

// newlines in text[0, x)
__device__ uint32_t seg_lines_before(const uint8_t *__restrict__ text, uint32_t n, const uint32_t *__restrict__ tile_off, uint32_t x)
{
    const uint32_t lane = lane_id(), t = x / FQZ_TILE, tb = t * FQZ_TILE;
    uint32_t c = 0;
#pragma unroll 1
    for (uint32_t q = 0; q < FQZ_TILE / 1024; q++) {
        const uint32_t off = tb + q * 1024 + 16 * lane;
        if (tb + q * 1024 >= x) break; // (wave-uniform)
        uint32_t m = seg_nl_mask16(load_text16(text, off, n));
        if (off + 16 > x) m &= off >= x ? 0u : ((1u << (x - off)) - 1u);
        c += __popc(m);
    }
    return tile_off[t] + wave_sum(c);
}

// ---------------------------------------------------------------------------------------------
// counts and boundaries
// ---------------------------------------------------------------------------------------------
__global__ void k_seg_setup(EncInfo *info, const uint32_t *tile_off, uint32_t n_tiles, uint32_t rec_cap, uint32_t block_cap, uint32_t rpb, uint32_t final_batch)
{
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t n_lines = tile_off[n_tiles];
    info->n_lines = n_lines;
    const uint32_t total = n_lines / 4;
    uint32_t n_rec = final_batch ? total : (total / rpb) * rpb;
    uint32_t n_blocks = (n_rec + rpb - 1) / rpb;
    if (n_rec > rec_cap || n_blocks > block_cap) { info->status = FQZ_E_TOO_LARGE; n_rec = 0; n_blocks = 0; }
    info->n_rec_total = total;
    info->n_rec = n_rec;
    info->n_blocks = n_blocks;
}

// wave b <= n_blocks: bstart[b] = first byte of block b (bstart[n_blocks] = the end of the last block = info->consumed);
// wave n_blocks + 1: the checks the parser still makes on a dangling last record (parser.go:138-165) before it is dropped (:196-199)
__global__ __launch_bounds__(64) void k_seg_blocks(const uint8_t *__restrict__ text, uint32_t n, const uint32_t *__restrict__ tile_off, uint32_t n_tiles, EncInfo *info,
                                                   uint32_t *bstart, uint32_t rpb, uint32_t final_batch)
{
    const uint32_t b = blockIdx.x, nb = info->n_blocks, n_rec = info->n_rec, n_lines = info->n_lines;
    if (b > nb + 1) return;
    if (b <= nb) {
        const uint32_t line = b < nb ? 4 * rpb * b : 4 * n_rec;
        const uint32_t off = seg_line_start(text, n, tile_off, n_tiles, line);
        if (threadIdx.x == 0) {
            bstart[b] = off;
            if (b == nb) info->consumed = final_batch ? n : off;
        }
        return;
    }
    if (!final_batch || info->status) return;
    const uint32_t have = n_lines - 4 * n_rec; // 0..3 complete lines of a record that never ends
    if (have >= 1) {
        const uint32_t s0 = seg_line_start(text, n, tile_off, n_tiles, 4 * n_rec);
        if (threadIdx.x == 0 && !(s0 < n && text[s0] == '@')) report_error(info, n_rec, 0, FQZ_E_HDR_AT);
    }
    if (have >= 3) {
        const uint32_t s2 = seg_line_start(text, n, tile_off, n_tiles, 4 * n_rec + 2);
        if (threadIdx.x == 0 && !(s2 < n && text[s2] == '+')) report_error(info, n_rec, 1, FQZ_E_SEP_PLUS);
    }
}

// one 256-thread workgroup: segments per block and their numbering; plans[b].rec0 / nrec
__global__ __launch_bounds__(256) void k_seg_plan(EncInfo *info, const uint32_t *bstart, BlockPlan *plans, uint32_t *seg_base, uint32_t rpb, uint32_t seg_cap)
{
    __shared__ uint32_t sh[4];
    const uint32_t t = threadIdx.x;
    const uint32_t n_blocks = info->status ? 0u : info->n_blocks, n_rec = info->n_rec;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256) {
        const uint32_t b = b0 + t;
        uint32_t ns = 0;
        if (b < n_blocks) ns = (bstart[b + 1] - bstart[b] + SEG_TEXT - 1) / SEG_TEXT;
        uint32_t tot;
        const uint32_t ex = carry + block_excl_scan_256(ns, sh, &tot);
        if (b < n_blocks) {
            seg_base[b] = ex;
            BlockPlan *p = &plans[b];
            p->rec0 = b * rpb;
            p->nrec = (b + 1) * rpb < n_rec ? rpb : n_rec - b * rpb;
            p->seg_base = ex;
            p->n_seg = ns;
            p->fallback = 0;
        }
        carry += tot;
        __syncthreads();
    }
    if (t == 0) {
        seg_base[n_blocks] = carry;
        info->n_segs = carry;
        info->n_hchunks = carry;            // headers chunk ordinals: a segment's first chunk has its number; further ones are handed out behind
        info->n_xgroups = FQZ_NS * carry;   // frames for the checksums: a list entry per segment and stream (empty ones are skipped)
        if (carry > seg_cap) { info->status = FQZ_E_TOO_LARGE; info->n_segs = 0; info->n_blocks = 0; info->n_rec = 0; info->n_hchunks = 0; info->n_xgroups = 0; }
    }
}

// wave g: segment g = (block b, window s): its first record and that record's first byte
__global__ __launch_bounds__(64) void k_seg_table(const uint8_t *__restrict__ text, uint32_t n, const uint32_t *__restrict__ tile_off, uint32_t n_tiles, const EncInfo *info,
                                                  const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ seg_base, SegInfo *seg, uint32_t rpb)
{
    const uint32_t g = blockIdx.x, n_segs = info->n_segs, nb = info->n_blocks, n_rec = info->n_rec;
    if (g > n_segs || info->status) return;
    if (g == n_segs) {
        if (threadIdx.x == 0) { seg[g].text_off = nb ? bstart[nb] : 0u; seg[g].rec0 = n_rec; }
        return;
    }
    uint32_t lo = 0, hi = nb; // seg_base[lo] <= g < seg_base[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_base[mid] <= g) lo = mid; else hi = mid; }
    const uint32_t b = lo, s = g - seg_base[b];
    uint32_t off, r;
    if (s == 0) { off = bstart[b]; r = b * rpb; }
    else {
        const uint32_t ws = bstart[b] + s * SEG_TEXT; // < bstart[b + 1] <= n
        const uint32_t before = seg_lines_before(text, n, tile_off, ws);
        const uint32_t j = before + (text[ws - 1] != '\n' ? 1u : 0u); // the first line that starts at or behind ws
        const uint32_t r_end = (b + 1) * rpb < n_rec ? (b + 1) * rpb : n_rec;
        r = (j + 3) / 4;
        if (r >= r_end) { r = r_end; off = bstart[b + 1]; }
        else off = seg_line_start(text, n, tile_off, n_tiles, 4 * r);
    }
    if (threadIdx.x == 0) { seg[g].text_off = off; seg[g].rec0 = r; }
}

// ---------------------------------------------------------------------------------------------
// the segment workgroup
// ---------------------------------------------------------------------------------------------
struct SegTabs {
    uint32_t lt[4 * SEG_RMAX + 8];       // line starts relative to the segment's first byte (lt[0] = 0)
    uint16_t L[SEG_RMAX + 8], H[SEG_RMAX + 8], P[SEG_RMAX + 8];
    uint16_t oseq[SEG_RMAX + 8], oqual[SEG_RMAX + 8], ohdr[SEG_RMAX + 8], oplus[SEG_RMAX + 8], onpos[SEG_RMAX + 8]; // offsets of record r in its part ([nrec] = total)
    uint16_t ipq[SEG_RMAX + 8];          // inclusive count of 16-byte quality pieces up to record r
    uint32_t ncnt[SEG_RMAX];             // N bases per record
};
struct SegLds {
    __attribute__((aligned(16))) uint8_t arena[SEG_ARENA + 64];
    union U {
        SegTabs tab;
        EntropyLds ent;
        __device__ U() {}
    } u;
    uint32_t qhist[4][256];  // byte histogram of every 16 KiB chunk of the quality part, counted while it is written
    uint32_t sh[4 * SEG_NW];
    uint32_t reg[FQZ_NS + 1]; // region starts inside the arena
    uint32_t raw[FQZ_NS];
    uint32_t chunk0[FQZ_NS];
    uint32_t slot0[FQZ_NS];
    uint32_t a_off[FQZ_NS];
    uint32_t misc[8];
};

// exclusive scan over the NW * 64 threads of the workgroup of up to four values at once; *tot = totals.  sh: 4 * NW words
template <int NW>
__device__ __forceinline__ void seg_scan4(const uint32_t v[4], uint32_t *sh, uint32_t ex[4], uint32_t tot[4])
{
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint32_t inc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) inc[c] = wave_incl_scan(v[c]);
    __syncthreads();
    if (l == 63) {
#pragma unroll
        for (int c = 0; c < 4; c++) sh[NW * c + w] = inc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t base = 0, t2 = 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NW; k++) { const uint32_t x = sh[NW * c + k]; if (k < w) base += x; t2 += x; }
        ex[c] = base + inc[c] - v[c];
        tot[c] = t2;
    }
}
__device__ __forceinline__ void seg_fallback(EncInfo *info, BlockPlan *plans, uint32_t blk)
{
    atomicOr(&plans[blk].fallback, 1u);
    atomicOr(&info->seg_fallback, 1u);
}

// The lengths part of a segment whose reads all have the same length (FQZ-S1; oracle fqzo_seg_frame, stream 5): one Compressed
// block = Raw literals (the first u32) + ONE sequence {literal length 4, match length M - 4, offset 4} on the predefined FSE
// tables (oracle hdr_write_sequences with n = 1: extra bits of the three codes, then the three initial states, the end mark).
// M < SEG_LEN_MIN is left to the general coder (the oracle judges such a block against its 66-bit bound and stores it Raw).
#define SEG_LEN_MIN 20u
__device__ __forceinline__ void seg_len_frame_block(const uint8_t *part, uint32_t M, uint8_t *slot, uint32_t *cs) // (one thread calls it)
{
    const uint32_t ml = M - 4, mlb = ml - 3;
    const uint32_t mc = mlb < 128 ? (uint32_t)c_hml_code[mlb] : (uint32_t)highbit32_d(mlb) + 36u;
    const uint32_t lc = 4, oc = 2, ofv = 4 + 3;
    unsigned long long acc = 0;
    uint32_t nb = 0;
    // literal length extra bits: none (code 4); match length, offset
    acc |= (unsigned long long)(ml - c_hml_base[mc]) << nb; nb += c_hml_bits[mc];
    acc |= (unsigned long long)(ofv - (1u << oc)) << nb; nb += oc;
    acc |= (unsigned long long)(c_hdr_trans.init[1][mc] & 63u) << nb; nb += 6; // FSE_flushCState: match lengths, offsets, literal lengths
    acc |= (unsigned long long)(c_hdr_trans.init[2][oc] & 31u) << nb; nb += 5;
    acc |= (unsigned long long)(c_hdr_trans.init[0][lc] & 63u) << nb; nb += 6;
    acc |= 1ull << nb; nb += 1;                                                 // end mark
    const uint32_t sbytes = (nb + 7) >> 3, content = 1 + 4 + 2 + sbytes;
    const uint32_t bh = 1u | (2u << 1) | (content << 3);
    slot[0] = (uint8_t)bh; slot[1] = (uint8_t)(bh >> 8); slot[2] = (uint8_t)(bh >> 16);
    slot[3] = (uint8_t)(4u << 3);                                               // Raw_Literals_Block, 1-byte header, 4 literals
    slot[4] = part[0]; slot[5] = part[1]; slot[6] = part[2]; slot[7] = part[3];
    slot[8] = 1;                                                                // Number_of_Sequences
    slot[9] = 0;                                                                // Symbol_Compression_Modes: Predefined x 3
    for (uint32_t i = 0; i < sbytes; i++) slot[10 + i] = (uint8_t)(acc >> (8 * i));
    *cs = 3 + content;
}

// MODE 0: encode.  MODE 1: DetectEncoding (quality.go:22-49) - the minimum quality byte of the segment goes to info->min_qual.
template <int MODE>
__device__ __forceinline__ void seg_workgroup(SegLds &S, const uint8_t *__restrict__ text, uint32_t n_text, EncInfo *info, SegInfo *seg, BlockPlan *plans, uint32_t rpb,
                              uint8_t *__restrict__ sarena, uint8_t *__restrict__ slots, uint32_t *__restrict__ csize, uint32_t *__restrict__ ehbuf,
                              SegHdrJob *__restrict__ jobs, uint32_t hcap, uint4 *__restrict__ hmap, uint4 *__restrict__ xmap, uint32_t g,
                              unsigned long long *stamps = nullptr)
{
    // stamps != nullptr (FQZ_DBG_STAMPS, diagnostic runs only): thread 0 records s_memtime at every phase boundary, 16 words a segment
#define SEG_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)g * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    SEG_STAMP(0);
    SegInfo *sg = &seg[g];
    const uint32_t a = sg->text_off, e = seg[g + 1].text_off, rec0 = sg->rec0, nrec = seg[g + 1].rec0 - rec0;
    const uint32_t blk = rpb ? rec0 / rpb : 0;
    const uint32_t n_segs = info->n_segs;
    SegTabs &T = S.u.tab;
    auto zero_out = [&]() { // nothing to code: the segment's entries in the job and group lists say so
        if (MODE == 0 && t < FQZ_NS) {
            sg->raw[t] = 0; sg->chunk0[t] = 0; sg->slot0[t] = 0; sg->a_off[t] = 0; sg->foff[t] = 0; sg->flen[t] = 0;
            xmap[t * n_segs + g] = make_uint4(0, 0, 0, 0);
            if (t == 0) { sg->hx = 0; hmap[g] = make_uint4(g, 0, 0, 0); SegHdrJob j = {0, 0, 0, 0, 0, g, 0, 0}; jobs[g] = j; }
        }
    };
    if (nrec == 0) { zero_out(); return; }
    if (MODE == 0 && (nrec > SEG_RMAX || e - a > SEG_SPAN_MAX)) { zero_out(); if (t == 0) seg_fallback(info, plans, blk); return; }
    if (MODE == 1 && (nrec > SEG_RMAX || e - a > SEG_SPAN_MAX)) { // (rare) the slow way: every quality byte of the segment's lines 3 mod 4, one thread
        if (t == 0) {
            uint32_t mn = 255, line = 0;
            for (uint32_t p = a; p < e; p++) { const uint32_t c = text[p]; if (c == '\n') line++; else if ((line & 3) == 3 && c != '\r') mn = c < mn ? c : mn; }
            // (a '\r' in front of the newline is stripped by the parser; one inside a quality line is below 59 anyway)
            if (mn < 255) atomicMin(&info->min_qual, mn);
        }
        return;
    }
    // ---- P1a: newline bitmap of [a, e), a u16 per aligned 16-byte piece, in the (still empty) arena
    const uint32_t base = a & ~15u, np = (e - base + 15) >> 4;
    uint16_t *bm = (uint16_t *)S.arena;
    // (eight loads a thread in flight: a load that is waited for right away costs a memory round trip a round)
    for (uint32_t i0 = 0; i0 < ((np + 15) & ~15u) + 16; i0 += 8 * SEG_NT) {
        uint4 v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t i = i0 + u * SEG_NT + t;
            v[u] = i < np ? load_text16(text, base + 16 * i, n_text) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t i = i0 + u * SEG_NT + t;
            uint32_t m = 0;
            if (i < np) {
                const uint32_t off = base + 16 * i;
                m = seg_nl_mask16(v[u]);
                if (off < a) m &= ~((1u << (a - off)) - 1u);
                if (off + 16 > e) m &= (1u << (e - off)) - 1u;
            }
            if (i < ((np + 15) & ~15u) + 16) bm[i] = (uint16_t)m;
        }
    }
    __syncthreads();
    SEG_STAMP(1);
    // ---- P1b: line starts: a thread per 16 pieces (256 text bytes), one scan of the counts per 256 threads
    const uint32_t ng = (np + 15) >> 4;
    uint32_t carry = 0;
    if (t == 0) T.lt[0] = 0;
    for (uint32_t g0 = 0; g0 < ng; g0 += SEG_NT) {
        const uint32_t gi = g0 + t;
        uint32_t w8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (gi < ng) {
            const uint4 x = *(const uint4 *)(S.arena + 32 * gi), y = *(const uint4 *)(S.arena + 32 * gi + 16);
            w8[0] = x.x; w8[1] = x.y; w8[2] = x.z; w8[3] = x.w; w8[4] = y.x; w8[5] = y.y; w8[6] = y.z; w8[7] = y.w;
        }
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) c += __popc(w8[d]);
        const uint32_t v4[4] = {c, 0, 0, 0};
        uint32_t ex[4], tot[4];
        seg_scan4<SEG_NW>(v4, S.sh, ex, tot);
        uint32_t idx = carry + ex[0] + 1; // newline j (1-based) starts line j
#pragma unroll
        for (int d = 0; d < 8; d++) {
            uint32_t w = w8[d];
            while (w) {
                const uint32_t bit = (uint32_t)__ffs(w) - 1;
                w &= w - 1;
                const uint32_t pos = base + 16 * (16 * gi + 2 * d + (bit >> 4)) + (bit & 15) + 1 - a;
                if (idx <= 4 * SEG_RMAX) T.lt[idx] = pos;
                idx++;
            }
        }
        carry += tot[0];
        __syncthreads();
    }
    if (carry != 4 * nrec) { // (cannot happen: the segment table comes from the same newline counts)
        zero_out();
        if (t == 0) { if (MODE == 0) seg_fallback(info, plans, blk); else info->status = FQZ_E_HIP; }
        return;
    }
    SEG_STAMP(2);
    // ---- P2: the records (parser.go:136-183 nextInto): line lengths without '\n' and one '\r', the checks, stream sizes
    uint32_t sz_tot[4] = {0, 0, 0, 0};
    uint32_t pq_min = 0xFFFFFFFFu, pq_max = 0, pq_carry = 0;
    uint32_t p_or = 0; // OR of the plus payload lengths of this thread's records
    for (uint32_t r0 = 0; r0 < nrec; r0 += SEG_NT) {
        const uint32_t r = r0 + t;
        uint32_t v[4] = {0, 0, 0, 0}, L = 0, H = 0, P = 0;
        if (r < nrec) {
            const uint32_t s0 = T.lt[4 * r], s1 = T.lt[4 * r + 1], s2 = T.lt[4 * r + 2], s3 = T.lt[4 * r + 3], s4 = T.lt[4 * r + 4];
            const uint8_t *x = text + a;
            uint32_t l0 = s1 - 1 - s0, l1 = s2 - 1 - s1, l2 = s3 - 1 - s2, l3 = s4 - 1 - s3;
            const uint32_t c0 = x[s0], c2 = x[s2];
            if (l0 && x[s1 - 2] == '\r') l0--;
            if (l1 && x[s2 - 2] == '\r') l1--;
            if (l2 && x[s3 - 2] == '\r') l2--;
            if (l3 && x[s4 - 2] == '\r') l3--;
            const uint32_t gr = rec0 + r;
            if (MODE == 0) {
                if (l0 == 0 || c0 != '@') { report_error(info, gr, 0, FQZ_E_HDR_AT); l0 = 1; }
                if (l2 == 0 || c2 != '+') { report_error(info, gr, 1, FQZ_E_SEP_PLUS); l2 = 1; }
                if (l1 != l3) report_error(info, gr, 2, FQZ_E_LEN_MISMATCH);
            } else { if (l0 == 0) l0 = 1; if (l2 == 0) l2 = 1; }
            H = l0 - 1; P = l2 - 1; L = l1;
            if (MODE == 0 && (H > 65535u || P > 65535u)) { report_error(info, gr, 3, FQZ_E_FIELD_WRAP); H &= 0xFFFF; P &= 0xFFFF; }
            if (MODE == 1) { // quality.go:22-49: the minimum over the quality bytes (of the line as the parser returns it: l3 bytes)
                uint32_t mn = 255;
                for (uint32_t i = 0; i < l3; i++) { const uint32_t q = x[s3 + i]; mn = q < mn ? q : mn; }
                if (mn < 255) atomicMin(&info->min_qual, mn);
            }
            if (L > 0xFFFFu) L = 0xFFFFu; // (cannot qualify: caught by the arena test below)
            T.L[r] = (uint16_t)L; T.H[r] = (uint16_t)H; T.P[r] = (uint16_t)P;
            v[0] = (L + 3) >> 2; v[1] = L; v[2] = 2 + H; v[3] = 2 + P;
        }
        if (MODE == 1) continue;
        uint32_t ex[4], tot[4];
        seg_scan4<SEG_NW>(v, S.sh, ex, tot);
        const uint32_t pq = (L + 15) >> 4;
        if (r < nrec) {
            T.oseq[r] = (uint16_t)(sz_tot[0] + ex[0]); T.oqual[r] = (uint16_t)(sz_tot[1] + ex[1]);
            T.ohdr[r] = (uint16_t)(sz_tot[2] + ex[2]); T.oplus[r] = (uint16_t)(sz_tot[3] + ex[3]);
            T.ncnt[r] = 0;
            p_or |= P;
            pq_min = pq < pq_min ? pq : pq_min; pq_max = pq > pq_max ? pq : pq_max;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) sz_tot[c] += tot[c];
        __syncthreads();
    }
    if (MODE == 1) return;
    SEG_STAMP(3);
    // sizes that cannot be held: the block goes the FQZ-H2 way (FQZ-S1: a segment's six parts, each rounded up to 16, fit SEG_ARENA)
    const uint32_t len_bytes = 4 * nrec;
    const uint32_t r_seq = 0, r_qual = r_seq + ((sz_tot[0] + 15) & ~15u), r_hdr = r_qual + ((sz_tot[1] + 15) & ~15u), r_plus = r_hdr + ((sz_tot[2] + 15) & ~15u),
                   r_len = r_plus + ((sz_tot[3] + 15) & ~15u), r_npos = r_len + ((len_bytes + 15) & ~15u);
    if ((unsigned long long)sz_tot[0] + sz_tot[1] + sz_tot[2] + sz_tot[3] > SEG_ARENA || r_npos + ((2 * nrec + 15) & ~15u) > SEG_ARENA) {
        zero_out();
        if (t == 0) seg_fallback(info, plans, blk);
        return;
    }
    if (t == 0) { T.oseq[nrec] = (uint16_t)sz_tot[0]; T.oqual[nrec] = (uint16_t)sz_tot[1]; T.ohdr[nrec] = (uint16_t)sz_tot[2]; T.oplus[nrec] = (uint16_t)sz_tot[3]; }
    // piece table of the quality / bases lines: uniform (every record the same count: fixed-length reads) or by search over ipq
    {
        const uint32_t mn = wave_min(pq_min), mx = 0xFFFFFFFFu - wave_min(0xFFFFFFFFu - pq_max);
        const unsigned long long anyp = __ballot(p_or != 0);
        __syncthreads(); // (the bitmap in the arena is dead from here on; S.sh free)
        if (lane == 0) { S.sh[wave] = mn; S.sh[SEG_NW + wave] = mx; S.sh[2 * SEG_NW + wave] = anyp ? 1u : 0u; }
        __syncthreads();
        uint32_t gmn = 0xFFFFFFFFu, gmx = 0, gp = 0;
#pragma unroll
        for (uint32_t w2 = 0; w2 < SEG_NW; w2++) { gmn = min(gmn, S.sh[w2]); gmx = max(gmx, S.sh[SEG_NW + w2]); gp |= S.sh[2 * SEG_NW + w2]; }
        pq_min = gmn; pq_max = gmx;
        p_or = gp; // 0: every plus line is bare -> the plus part is all zero bytes
        __syncthreads();
    }
    const uint32_t uni = (pq_min == pq_max && pq_min > 0) ? pq_min : 0u;
    if (!uni) { // inclusive piece counts (records in order; a serial prefix per 256 records through the scan)
        uint32_t pc = 0;
        for (uint32_t r0 = 0; r0 < nrec; r0 += SEG_NT) {
            const uint32_t r = r0 + t;
            const uint32_t pq = r < nrec ? ((uint32_t)T.L[r] + 15) >> 4 : 0u;
            const uint32_t v4[4] = {pq, 0, 0, 0};
            uint32_t ex[4], tot[4];
            seg_scan4<SEG_NW>(v4, S.sh, ex, tot);
            if (r < nrec) T.ipq[r] = (uint16_t)(pc + ex[0] + pq);
            pc += tot[0];
            __syncthreads();
        }
        pq_carry = pc;
    }
    const uint32_t Tq = uni ? uni * nrec : pq_carry;
    const uint32_t qoff = info->qual_off;
    __syncthreads();
    SEG_STAMP(4);
    // ---- P3: split.  Bases and qualities piece by piece (16 text bytes of one line per lane and round, as k_split does);
    //      headers, plus lines and lengths a record per thread.
    for (uint32_t i = t; i < 4 * 256; i += SEG_NT) (&S.qhist[0][0])[i] = 0;
    __syncthreads();
    const float inv_uni = uni ? 1.0f / (float)uni : 0.0f;
#define SEG_SR 4u // rounds of 256 pieces per trip: 2 x SEG_SR loads a thread in flight before the first is used
    for (uint32_t p0 = 0; p0 < Tq; p0 += SEG_SR * SEG_NT) {
        uint32_t x[SEG_SR][4], y[SEG_SR][4], have[SEG_SR], srcq[SEG_SR], dseq[SEG_SR], dqual[SEG_SR], ri[SEG_SR], rk[SEG_SR];
#pragma unroll
        for (uint32_t u = 0; u < SEG_SR; u++) {
            const uint32_t p = p0 + u * SEG_NT + t;
            const bool on = p < Tq;
            uint32_t i = 0, k = 0;
            if (on) {
                if (uni) {
                    uint32_t q = (uint32_t)(((float)p + 0.5f) * inv_uni);
                    if (q * uni > p) q--;
                    if ((q + 1) * uni <= p) q++;
                    i = q; k = p - q * uni;
                } else { // smallest i with ipq[i] > p
                    uint32_t lo = 0, hi = nrec - 1;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)T.ipq[mid] > p) hi = mid; else lo = mid + 1; }
                    i = lo;
                    k = p - ((uint32_t)T.ipq[i] - (((uint32_t)T.L[i] + 15) >> 4));
                }
            }
            ri[u] = i; rk[u] = k;
            have[u] = 0; srcq[u] = 0; dseq[u] = 0; dqual[u] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) x[u][q] = y[u][q] = 0;
            if (on) {
                const uint32_t Li = T.L[i];
                const uint32_t src = a + T.lt[4 * i + 1];
                srcq[u] = a + T.lt[4 * i + 3];
                dseq[u] = r_seq + T.oseq[i] + 4 * k;
                dqual[u] = r_qual + T.oqual[i] + 16 * k;
                have[u] = Li - 16 * k < 16 ? Li - 16 * k : 16;
                load_piece(text, (size_t)src + 16 * k, n_text, x[u]);
                load_piece(text, (size_t)srcq[u] + 16 * k, n_text, y[u]);
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < SEG_SR; u++) {
            const uint32_t p = p0 + u * SEG_NT + t;
            const bool on = p < Tq;
            const uint32_t k = rk[u], i = ri[u];
            // the byte in front of a quality piece is the last byte of the previous lane's piece (same read, piece k - 1); lane 0 fetches it
            const uint32_t left = (uint32_t)__shfl_up((int)(y[u][3] >> 24), 1, WAVE);
            if (on) {
                uint32_t out = 0, nn = 0;
                const uint32_t hv0 = have[u];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t v = x[u][q], in_read = 0x80808080u;
                    if (hv0 < 4u * q + 4) { // bytes past the read pack as 0
                        const uint32_t hv = hv0 > 4u * q ? hv0 - 4u * q : 0;
                        v = hv ? v & ((1u << (8 * hv)) - 1) : 0;
                        in_read = hv ? in_read >> (8 * (4 - hv)) : 0;
                    }
                    const uint32_t vmask = acgt_mask(v);
                    out |= pack4(v, vmask) << (8 * q);
                    nn += __popc(~vmask & in_read);
                }
                uint32_t prev = k ? (lane ? left : (uint32_t)text[srcq[u] + 16 * k - 1]) : qoff;
#pragma unroll
                for (int q = 0; q < 4; q++) { const uint32_t yq = y[u][q]; y[u][q] = sub_bytes(yq, (yq << 8) | (prev & 0xFF)); prev = yq >> 24; }
                const uint32_t nb = (hv0 + 3) >> 2;
                uint8_t *o = S.arena + dseq[u];
                if (nb == 4) store_u32_unaligned(o, out);
                else {
                    if (nb & 2) { const uint16_t v16 = (uint16_t)out; __builtin_memcpy(o, &v16, 2); }
                    if (nb & 1) o[nb & 2] = (uint8_t)(out >> (8 * (nb & 2)));
                }
                store_piece(S.arena + dqual[u], y[u], hv0);
                if (nn) atomicAdd(&T.ncnt[i], nn);
            }
            // ---- histogram of the quality part, per 16 KiB chunk, from the registers: zeros (most delta bytes) are counted with
            //      SWAR compares and added once per wave, the rest a byte at a time with LDS atomics
            {
                const uint32_t hv0 = on ? have[u] : 0u;
                const uint32_t qo = dqual[u] - r_qual, ck = qo >> 14; // (FQZ_CHUNK = 2^14)
                const bool straddle = on && ((qo & (FQZ_CHUNK - 1)) + hv0 > FQZ_CHUNK);
                uint32_t nz = 0;
                if (on && !straddle) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t valid = hv0 >= 4u * q + 4 ? 0x80808080u : (hv0 > 4u * q ? (0x80808080u >> (8 * (4 * q + 4 - hv0))) : 0u);
                        const uint32_t z = zero_bytes(y[u][q]) & valid;
                        nz += __popc(z);
                        uint32_t other = valid & ~z;
                        while (other) {
                            const int bit = __ffs(other) - 1; // 7, 15, 23 or 31
                            other &= other - 1;
                            atomicAdd(&S.qhist[ck][(y[u][q] >> (bit - 7)) & 0xFF], 1u);
                        }
                    }
                } else if (straddle) { // (once per chunk boundary) byte by byte
                    for (uint32_t bi = 0; bi < hv0; bi++) atomicAdd(&S.qhist[(qo + bi) >> 14][(y[u][bi >> 2] >> (8 * (bi & 3))) & 0xFF], 1u);
                }
                // zeros: one add per wave when all its pieces lie in one chunk (nearly always)
                const uint32_t ck0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck);
                const bool mixed = __ballot(on && !straddle && ck != ck0) != 0;
                if (!mixed) {
                    const uint32_t tot = wave_sum(nz);
                    if (lane == 0 && tot) atomicAdd(&S.qhist[ck0 & 3][0], tot);
                } else if (nz) atomicAdd(&S.qhist[ck][0], nz);
            }
        }
    }
    SEG_STAMP(5);
    for (uint32_t r = t; r < nrec; r += SEG_NT) {
        const uint32_t H = T.H[r], P = T.P[r], L = T.L[r];
        *(uint32_t *)(S.arena + r_len + 4 * r) = L; // compress.go:501
        { // headers: [u16 H][H bytes without '@'] (compress.go:514-515), 16 bytes of that image at a time
            const uint32_t src = a + T.lt[4 * r] + 1;
            uint8_t *dst = S.arena + r_hdr + T.ohdr[r];
            const uint32_t img = H + 2;
            for (uint32_t k = 0; 16 * k < img; k++) {
                uint32_t x[4];
                if (k) load_piece(text, (size_t)src + 16 * k - 2, n_text, x);
                else {
                    uint32_t y[4];
                    load_piece(text, src, n_text, y);
                    x[0] = (y[0] << 16) | (H & 0xFFFFu); x[1] = (y[1] << 16) | (y[0] >> 16); x[2] = (y[2] << 16) | (y[1] >> 16); x[3] = (y[3] << 16) | (y[2] >> 16);
                }
                store_piece(dst + 16 * k, x, img - 16 * k < 16 ? img - 16 * k : 16);
            }
        }
        { // plus line: [u16 P][P bytes without '+'] (compress.go:518-519)
            const uint32_t src = a + T.lt[4 * r + 2] + 1;
            uint8_t *dst = S.arena + r_plus + T.oplus[r];
            const uint32_t img = P + 2;
            for (uint32_t k = 0; 16 * k < img; k++) {
                uint32_t x[4];
                if (k) load_piece(text, (size_t)src + 16 * k - 2, n_text, x);
                else {
                    uint32_t y[4] = {0, 0, 0, 0};
                    if (P) load_piece(text, src, n_text, y);
                    x[0] = (y[0] << 16) | (P & 0xFFFFu); x[1] = (y[1] << 16) | (y[0] >> 16); x[2] = (y[2] << 16) | (y[1] >> 16); x[3] = (y[3] << 16) | (y[2] >> 16);
                }
                store_piece(dst + 16 * k, x, img - 16 * k < 16 ? img - 16 * k : 16);
            }
        }
    }
    __syncthreads();
    SEG_STAMP(6);
    // ---- P3b: N positions: [u16 n][n x u16 position] per record (compress.go:477-498); positions below 65536 only - every read
    //      of a qualifying segment is shorter than that
    uint32_t np_tot = 0;
    for (uint32_t r0 = 0; r0 < nrec; r0 += SEG_NT) {
        const uint32_t r = r0 + t;
        const uint32_t nn = r < nrec ? T.ncnt[r] : 0u;
        const uint32_t v4[4] = {r < nrec ? 2 + 2 * nn : 0u, 0, 0, 0};
        uint32_t ex[4], tot[4];
        seg_scan4<SEG_NW>(v4, S.sh, ex, tot);
        if (r < nrec) T.onpos[r] = (uint16_t)((np_tot + ex[0]) & 0xFFFFu);
        if (r < nrec && np_tot + ex[0] > 0xFFFFu) np_tot = 0x10000000u; // (far beyond the arena: caught below)
        np_tot += tot[0];
        __syncthreads();
    }
    if (r_npos + ((np_tot + 15) & ~15u) > SEG_ARENA || np_tot >= 0x10000000u) {
        zero_out();
        if (t == 0) seg_fallback(info, plans, blk);
        return;
    }
    if (t == 0) T.onpos[nrec] = (uint16_t)np_tot;
    for (uint32_t r = t; r < nrec; r += SEG_NT) {
        const uint16_t n16 = (uint16_t)T.ncnt[r];
        *(uint16_t *)(S.arena + r_npos + T.onpos[r]) = n16; // (2-aligned: every record takes an even number of bytes)
    }
    for (uint32_t r = wave; np_tot != 2 * nrec && r < nrec; r += SEG_NT / 64) { // a wave per record that has any: its pieces lane by lane
        const uint32_t nn = T.ncnt[r];
        if (!nn) continue;
        const uint32_t L = T.L[r], src = a + T.lt[4 * r + 1];
        uint8_t *o = S.arena + r_npos + T.onpos[r] + 2;
        uint32_t done = 0;
        for (uint32_t kb = 0; 16 * kb < L; kb += 64) {
            const uint32_t k = kb + lane;
            uint32_t bad16 = 0;
            if (16 * k < L) {
                uint32_t x[4];
                load_piece(text, (size_t)src + 16 * k, n_text, x);
                const uint32_t have = L - 16 * k < 16 ? L - 16 * k : 16;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t inv = ~acgt_mask(x[q]) & 0x80808080u;
                    bad16 |= ((((inv >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * q);
                }
                bad16 &= have >= 16 ? 0xFFFFu : ((1u << have) - 1);
            }
            const uint32_t c = __popc(bad16), incl = wave_incl_scan(c);
            uint32_t rank = done + incl - c, m2 = bad16;
            while (m2) {
                const uint32_t bpos = 16 * k + (uint32_t)(__ffs(m2) - 1);
                m2 &= m2 - 1;
                *(uint16_t *)(o + 2 * rank) = (uint16_t)bpos;
                rank++;
            }
            done += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    // all u32 lengths equal: the lengths part is 4 literal bytes and one match (FQZ-S1)
    uint32_t len_same = 0;
    {
        uint32_t diff = 0;
        for (uint32_t r = t; r < nrec; r += SEG_NT) diff |= (uint32_t)T.L[r] ^ (uint32_t)T.L[0];
        const unsigned long long any = __ballot(diff != 0);
        __syncthreads();
        if (lane == 0) S.sh[wave] = any ? 1u : 0u;
        __syncthreads();
        uint32_t anyd = 0;
#pragma unroll
        for (uint32_t w2 = 0; w2 < SEG_NW; w2++) anyd |= S.sh[w2];
        len_same = !anyd && len_bytes >= SEG_LEN_MIN ? 1u : 0u;
    }
    __syncthreads();
    SEG_STAMP(7);
    // ---- P4: where things go.  Everything a segment owns in HBM sits at a place its number gives (csize / xsum entries, slot pool
    //      pages, stream arena, headers chunk ordinal = g, record-offset table): 15 000 workgroups bumping shared counters spent more
    //      time in those atomics than in the coding.  Only a headers part longer than one chunk (rare) takes its further chunk
    //      ordinals from a counter.
    if (t == 0) {
        const uint32_t raw[FQZ_NS] = {sz_tot[0], sz_tot[1], sz_tot[2], sz_tot[3], np_tot, len_bytes};
        const uint32_t reg[FQZ_NS] = {r_seq, r_qual, r_hdr, r_plus, r_npos, r_len};
        uint32_t id = g * SEG_IDS, pg = g * SEG_SLOT_PAGES, apos = g * SEG_ASTRIDE;
        bool ok = true;
        const uint32_t nch_h = (raw[S_HDR] + FQZ_CHUNK - 1) / FQZ_CHUNK;
        uint32_t ox = 0; // ordinals of the headers chunks 1.. (chunk 0 has ordinal g)
        if (nch_h > 1) { ox = atomicAdd(&info->n_hchunks, nch_h - 1); if (ox + nch_h - 1 > hcap) { ok = false; atomicCAS(&info->status, 0, FQZ_E_TOO_LARGE); } }
        for (int s = 0; s < FQZ_NS; s++) {
            S.raw[s] = raw[s]; S.reg[s] = reg[s];
            S.chunk0[s] = id;
            S.slot0[s] = pg;
            S.a_off[s] = apos;
            id += s == S_SEQ ? 1u : (raw[s] + FQZ_CHUNK - 1) / FQZ_CHUNK;
            pg += seg_pages(raw[s], s);
            apos += (raw[s] + 15) & ~15u;
            sg->raw[s] = raw[s]; sg->chunk0[s] = S.chunk0[s]; sg->slot0[s] = S.slot0[s]; sg->a_off[s] = S.a_off[s]; sg->foff[s] = 0; sg->flen[s] = 0;
        }
        sg->hx = ox;
        S.misc[0] = ok ? 1u : 0u;
        if (ok) {
            for (uint32_t k = 0; k < nch_h; k++) {
                const uint32_t o = k ? ox + k - 1 : g;
                SegHdrJob j;
                j.a_off = S.a_off[S_HDR]; j.e_off = g * (SEG_RMAX + 1); j.nrec = nrec; j.c0 = k * FQZ_CHUNK;
                j.mk = raw[S_HDR] - k * FQZ_CHUNK < FQZ_CHUNK ? raw[S_HDR] - k * FQZ_CHUNK : FQZ_CHUNK;
                j.chunk = o; j.slot_off = S.slot0[S_HDR] * SEG_PAGE + k * FQZ_SLOT; j.cs = S.chunk0[S_HDR] + k;
                jobs[o] = j;
            }
            // (groups: one list entry per segment and stream, empty ones included - k_entropy_hdr_seg / k_xxh skip them)
            hmap[g] = make_uint4(g, S.a_off[S_HDR], raw[S_HDR] | ((uint32_t)S_HDR << 28), ox);
            // (stream-major: the 16 frames a wave of k_xxh hashes side by side are of one stream, so of about one length)
            for (int s = 0; s < FQZ_NS; s++) xmap[s * n_segs + g] = make_uint4(S.chunk0[s], S.a_off[s], raw[s] | ((uint32_t)s << 28), 0u);
        }
    }
    __syncthreads();
    if (!S.misc[0]) return;
    {
        // the six parts lie back to back (each 16-aligned) in LDS in the order seq, qual, hdr, plus, len, npos and in the arena in
        // stream order: region by region, 16 bytes a thread
        for (int s = 0; s < FQZ_NS; s++) {
            const uint32_t nb16 = (S.raw[s] + 15) >> 4;
            uint4 *dst = (uint4 *)(sarena + S.a_off[s]);
            const uint4 *src = (const uint4 *)(S.arena + S.reg[s]);
            for (uint32_t i = t; i < nb16; i += SEG_NT) dst[i] = src[i];
        }
        // packed bases: Raw blocks by definition; their bytes wait in their slots for k_seg_compact
        {
            const uint32_t nb16 = (S.raw[S_SEQ] + 15) >> 4;
            uint4 *dst = (uint4 *)(slots + (size_t)S.slot0[S_SEQ] * SEG_PAGE);
            const uint4 *src = (const uint4 *)(S.arena + S.reg[S_SEQ]);
            for (uint32_t i = t; i < nb16; i += SEG_NT) dst[i] = src[i];
        }
        if (S.raw[S_HDR]) for (uint32_t r = t; r <= nrec; r += SEG_NT) ehbuf[g * (SEG_RMAX + 1) + r] = T.ohdr[r];
        // (the lengths block of a segment of equal read lengths: one thread's few dependent steps, beside the copies)
        if (len_same && t == SEG_NT - 1) seg_len_frame_block(S.arena + S.reg[S_LEN], S.raw[S_LEN], slots + (size_t)S.slot0[S_LEN] * SEG_PAGE, csize + S.chunk0[S_LEN]);
    }
    __syncthreads(); // the record tables are dead: their LDS becomes the entropy coder's
    SEG_STAMP(8);
    // ---- P5: entropy stage from LDS: qualities, plus lines, N positions, lengths
    const int order[4] = {S_QUAL, S_PLUS, S_NPOS, S_LEN};
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        const int s = order[q];
        const uint32_t M = S.raw[s];
        if (!M) continue;
        uint8_t *slot0 = slots + (size_t)S.slot0[s] * SEG_PAGE;
        uint32_t *cs0 = csize + S.chunk0[s];
        if (s == S_LEN && len_same) continue; // (written beside the copies above)
        if ((s == S_PLUS && !p_or) || (s == S_NPOS && np_tot == 2 * nrec)) {
            // a part of zero bytes (bare plus lines; no N anywhere): an RLE block per 16 KiB, as the coder would find out after a
            // histogram and a dozen barriers (fqz_entropy_dev.h: same_mask)
            const uint32_t nch = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
            if (t < nch) {
                const uint32_t mk = M - t * FQZ_CHUNK < FQZ_CHUNK ? M - t * FQZ_CHUNK : FQZ_CHUNK;
                const uint32_t bh = (t + 1 == nch ? 1u : 0u) | (1u << 1) | (mk << 3);
                *(uint32_t *)(slot0 + (size_t)t * FQZ_SLOT) = bh & 0xFFFFFFu; // (the repeated byte: 0)
                cs0[t] = 4;
            }
            continue;
        }
        // (diagnostic runs: the coder's own phase stamps of the quality part go behind the segments' stamps)
        seg_encode_part(S.u.ent, S.arena + S.reg[s], sarena + S.a_off[s], M, slot0, cs0, s == S_QUAL ? &S.qhist[0][0] : nullptr,
                        (stamps && s == S_QUAL) ? stamps + ((size_t)gridDim.x + g) * 16 : nullptr);
        __syncthreads();
        SEG_STAMP(9 + q);
    }
    SEG_STAMP(13);
#undef SEG_STAMP
}

__global__ __launch_bounds__(SEG_NT, 4) void k_seg_encode(const uint8_t *__restrict__ text, uint32_t n_text, EncInfo *info, SegInfo *seg, BlockPlan *plans, uint32_t rpb,
                                                      uint8_t *__restrict__ sarena, uint8_t *__restrict__ slots, uint32_t *__restrict__ csize, uint32_t *__restrict__ ehbuf,
                                                      SegHdrJob *__restrict__ jobs, uint32_t hcap, uint4 *__restrict__ hmap, uint4 *__restrict__ xmap, unsigned long long *stamps)
{
    __shared__ SegLds S;
    const uint32_t g = blockIdx.x;
    if (g >= info->n_segs || info->status) return;
    seg_workgroup<0>(S, text, n_text, info, seg, plans, rpb, sarena, slots, csize, ehbuf, jobs, hcap, hmap, xmap, g, stamps);
}

// encoder.DetectEncoding (quality.go:22-49) over the first block: the segments of block 0
__global__ __launch_bounds__(SEG_NT) void k_seg_detect(const uint8_t *__restrict__ text, uint32_t n_text, EncInfo *info, SegInfo *seg, BlockPlan *plans, uint32_t rpb)
{
    __shared__ SegLds S;
    const uint32_t g = blockIdx.x;
    if (info->status || !info->n_blocks || g >= plans[0].n_seg) return;
    seg_workgroup<1>(S, text, n_text, info, seg, plans, rpb, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, g);
}

// the headers model (fqz_hdrlz.h) over the headers chunks the segment workgroups announced
__global__ __launch_bounds__(256) void k_hdr_model_seg(const EncInfo *info, const SegHdrJob *jobs, uint32_t hcap, const uint32_t *ehbuf, const uint8_t *sarena, uint2 *hseq,
                                                       uint8_t *hlit, HdrSide *side, uint16_t *hhist)
{
    __shared__ __attribute__((aligned(16))) HdrModelLds S;
    const uint32_t o = blockIdx.x;
    if (o >= info->n_hchunks || o >= hcap || info->status) return;
    const SegHdrJob j = jobs[o];
    if (!j.mk) { if (threadIdx.x == 0) { HdrSide z = {0, 0, 0, 0}; side[o] = z; } return; } // a segment without records
    hdr_model_chunk(S, sarena + j.a_off, ehbuf + j.e_off, 0u, j.nrec, j.c0, j.mk, hseq + (size_t)o * HDR_MAX_SEQ, hlit + (size_t)o * FQZ_CHUNK, &side[o], hhist + (size_t)o * 256);
}

// the entropy stage over the literals of a segment's headers part (k_entropy_hdr with the segment path's numbering: chunk 0 of
// the part has the segment's number as its ordinal in the side buffers, further chunks the ordinals hmap[g].w ...)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_entropy_hdr_seg(const EncInfo *info, const uint4 *hmap, const SegInfo *seg, const uint8_t *sarena,
                                                         uint8_t *slots, uint32_t *csize, uint32_t hcap, const uint8_t *hlit, const HdrSide *side, const uint16_t *hhist)
{
    __shared__ __attribute__((aligned(16))) EntropyLds S;
    __shared__ HdrGroup H;
    const uint32_t g = blockIdx.x;
    if (g >= info->n_segs || info->status) return;
    const uint4 gd = hmap[g];
    const uint32_t M = gd.z & 0xFFFFFFu, t = threadIdx.x;
    if (!M) return;
    const uint8_t *src = sarena + gd.y;
    if (t < (M + FQZ_CHUNK - 1) / FQZ_CHUNK) {
        const uint32_t mk = M - t * FQZ_CHUNK < FQZ_CHUNK ? M - t * FQZ_CHUNK : FQZ_CHUNK, o = t ? gd.w + t - 1 : g;
        HdrSide sd = {0, mk, 0, 0};
        if (o < hcap) sd = side[o];
        H.nseq[t] = sd.nseq; H.n_lit[t] = sd.nseq ? sd.n_lit : mk;
        H.lit[t] = sd.nseq ? hlit + (size_t)o * FQZ_CHUNK : src + (size_t)t * FQZ_CHUNK;
        H.hist[t] = hhist + (size_t)(o < hcap ? o : 0) * 256;
    }
    __syncthreads();
    entropy_encode_group<true>(S, src, M, 0u, slots + (size_t)seg[g].slot0[S_HDR] * SEG_PAGE, &csize[seg[g].chunk0[S_HDR]], 0, nullptr, &H);
}

// k_hdr_patch with the segment path's numbering: the Sequences_Section of a headers chunk goes behind its literals
__global__ __launch_bounds__(64) void k_hdr_patch_seg(const EncInfo *info, const SegHdrJob *jobs, uint32_t hcap, const HdrSide *side, const uint8_t *hsec, uint8_t *slots, uint32_t *csize)
{
    const uint32_t o = blockIdx.x, lane = threadIdx.x;
    if (o >= info->n_hchunks || o >= hcap || info->status) return;
    const HdrSide sd = side[o];
    if (!sd.nseq) return;
    const SegHdrJob j = jobs[o];
    uint8_t *slot = slots + j.slot_off;
    const uint32_t bh = slot[0] | ((uint32_t)slot[1] << 8) | ((uint32_t)slot[2] << 16);
    if (((bh >> 1) & 3) != 2) return; // the chunk became a Raw block
    const uint32_t prov = csize[j.cs], ssz = sd.sec_len;
    const uint8_t *sec = hsec + (size_t)o * HDR_SEQ_CAP;
    for (uint32_t i = lane; i < ssz; i += 64) slot[prov + i] = sec[i];
    if (lane == 0) {
        const uint32_t fin = prov + ssz, nb = (bh & 7u) | ((fin - 3u) << 3);
        slot[0] = (uint8_t)nb; slot[1] = (uint8_t)(nb >> 8); slot[2] = (uint8_t)(nb >> 16);
        csize[j.cs] = fin;
    }
}

// ---------------------------------------------------------------------------------------------
// framing (FQZ-S1): frame sizes, payload and block offsets, the blocks in their final place
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t seg_idx_len(uint32_t n_seg) { return 24u + 8u * n_seg; }
__device__ __forceinline__ uint32_t seg_fh(uint32_t M) { return M < 256u ? 6u : 7u; }

// a workgroup per block: the frame lengths of its segments (from the compressed sizes of their zstd blocks), their places inside
// the payloads, the payload lengths.  csize is read as the coders left it (not scanned).
__global__ __launch_bounds__(256) void k_seg_sizes(EncInfo *info, BlockPlan *plans, SegInfo *seg, const uint32_t *__restrict__ csize)
{
    __shared__ uint32_t sh[16];
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (info->status || b >= info->n_blocks) return;
    BlockPlan *p = &plans[b];
    const uint32_t s0 = p->seg_base, ns = p->n_seg;
    uint32_t run[FQZ_NS] = {0, 0, 0, 0, 0, 0}, rawt[FQZ_NS] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i0 = 0; i0 < ns; i0 += 256) {
        const uint32_t i = i0 + t;
        uint32_t fl[FQZ_NS] = {0, 0, 0, 0, 0, 0}, rw[FQZ_NS] = {0, 0, 0, 0, 0, 0};
        SegInfo *sg = &seg[s0 + (i < ns ? i : 0)];
        if (i < ns) {
            for (int s = 0; s < FQZ_NS; s++) {
                const uint32_t M = sg->raw[s];
                rw[s] = M;
                if (!M) continue;
                const uint32_t nch = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
                uint32_t body = 0;
                if (s == S_SEQ) body = M + 3 * nch;
                else for (uint32_t k = 0; k < nch; k++) body += csize[sg->chunk0[s] + k];
                fl[s] = seg_fh(M) + body + 4;
            }
        }
        uint32_t ex[4], tot[4], ex2[4], tot2[4];
        const uint32_t va[4] = {fl[0], fl[1], fl[2], fl[3]}, vb[4] = {fl[4], fl[5], rw[0] + rw[1], rw[2] + rw[3] + rw[4] + rw[5]};
        seg_scan4<4>(va, sh, ex, tot);
        seg_scan4<4>(vb, sh, ex2, tot2);
        if (i < ns) {
            sg->foff[0] = run[0] + ex[0]; sg->foff[1] = run[1] + ex[1]; sg->foff[2] = run[2] + ex[2]; sg->foff[3] = run[3] + ex[3];
            sg->foff[4] = run[4] + ex2[0]; sg->foff[5] = run[5] + ex2[1];
            for (int s = 0; s < FQZ_NS; s++) sg->flen[s] = fl[s];
        }
        run[0] += tot[0]; run[1] += tot[1]; run[2] += tot[2]; run[3] += tot[3]; run[4] += tot2[0]; run[5] += tot2[1];
        // raw totals per stream: summed per thread below (six columns do not fit two scans)
        for (int s = 0; s < FQZ_NS; s++) rawt[s] += rw[s];
        __syncthreads();
    }
    // raw totals: wave sums, then across the four waves
    for (int s = 0; s < FQZ_NS; s++) {
        const uint32_t ws = wave_sum(rawt[s]);
        __syncthreads();
        if ((t & 63) == 0) sh[t >> 6] = ws;
        __syncthreads();
        rawt[s] = sh[0] + sh[1] + sh[2] + sh[3];
    }
    if (t == 0) {
        uint32_t orig = 0;
        for (int s = 0; s < FQZ_NS; s++) {
            p->len[s] = rawt[s];
            p->frame_len[s] = rawt[s] ? seg_idx_len(ns) + run[s] : 0u; // an empty stream is an empty payload
        }
        orig = rawt[S_QUAL];
        p->orig_seq = orig;
        for (int s = 0; s < FQZ_NS; s++) if (rawt[s]) atomicAdd(&info->stream_raw[s], (unsigned long long)rawt[s]);
    }
}

// one 256-thread workgroup: block sizes -> offsets, block headers (container.go:97-109), the index frame headers of the payloads
__global__ __launch_bounds__(256) void k_seg_layout(EncInfo *info, BlockPlan *plans, uint8_t *out, size_t out_cap)
{
    __shared__ uint32_t sh[4];
    if (blockIdx.x) return;
    const uint32_t t = threadIdx.x;
    if (t == 0 && info->error_key != ~0ull && info->status == 0) {
        info->status = -(int32_t)(info->error_key & 31);
        info->error_record = (uint32_t)(info->error_key >> 8);
    }
    __syncthreads();
    const uint32_t n_blocks = (info->status || info->seg_fallback) ? 0u : info->n_blocks;
    unsigned long long carry = 0;
    unsigned long long comp[FQZ_NS] = {0, 0, 0, 0, 0, 0};
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t b = base + t;
        BlockPlan *p = b < n_blocks ? &plans[b] : nullptr;
        uint32_t size = 0;
        if (p) { size = 36; for (int s = 0; s < FQZ_NS; s++) size += p->frame_len[s]; }
        uint32_t tot;
        const unsigned long long start = carry + block_excl_scan_256(size, sh, &tot);
        carry += tot;
        if (p) {
            unsigned long long pos = start + 36;
            for (int s = 0; s < FQZ_NS; s++) { p->frame_off[s] = (uint32_t)pos; pos += p->frame_len[s]; comp[s] += p->frame_len[s]; }
            p->out_off = (uint32_t)start;
            p->out_len = size;
        }
        __syncthreads();
    }
    for (int s = 0; s < FQZ_NS; s++) if (comp[s]) atomicAdd(&info->stream_comp[s], comp[s]);
    const unsigned long long total = carry;
    if (t == 0) {
        info->out_len = total;
        if ((total > out_cap || total > 0xFFFFFFF0ull) && !info->status) info->status = FQZ_E_DST_SMALL;
    }
    if (total > out_cap || total > 0xFFFFFFF0ull || info->status) return;
    for (uint32_t b = t; b < n_blocks; b += 256) {
        BlockPlan *p = &plans[b];
        uint8_t *h = out + p->out_off;
        put_le32(h + 0, p->nrec);
        put_le32(h + 4, p->frame_len[S_SEQ]);
        put_le32(h + 8, p->frame_len[S_QUAL]);
        put_le32(h + 12, p->frame_len[S_HDR]);
        put_le32(h + 16, p->frame_len[S_PLUS]);
        put_le32(h + 20, p->frame_len[S_NPOS]);
        put_le32(h + 24, p->frame_len[S_LEN]);
        put_le32(h + 28, p->orig_seq);
        put_le32(h + 32, p->orig_seq);
        for (int s = 0; s < FQZ_NS; s++) {
            if (!p->frame_len[s]) continue;
            uint8_t *f = out + p->frame_off[s]; // the index: a zstd skippable frame (k_seg_compact writes the entries)
            put_le32(f, 0x184D2A50u);
            put_le32(f + 4, seg_idx_len(p->n_seg) - 8);
            f[8] = 'F'; f[9] = 'Q'; f[10] = 'Z'; f[11] = 'I';
            f[12] = 2; f[13] = (uint8_t)s; f[14] = 0; f[15] = 0;
            put_le32(f + 16, p->len[s]);
            put_le32(f + 20, p->n_seg);
        }
    }
}

// n bytes src -> dst (any alignment of dst; src readable in 16-byte units from a 16-aligned base + any offset): the 256 threads
__device__ __forceinline__ void seg_copy(uint8_t *dst, const uint8_t *src, uint32_t n)
{
    const uint32_t t = threadIdx.x;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    const uint32_t rows = (n - head) >> 4;
    for (uint32_t j = t; j < rows; j += 256) *(uint4 *)(dst + head + 16 * j) = load_u128_unaligned(src + head + 16 * j);
    const uint32_t tail0 = head + 16 * rows;
    if (t < n - tail0) dst[tail0 + t] = src[tail0 + t];
}

// a workgroup per segment: its frames -> their places: frame header, the zstd blocks from their slots (packed bases: a Raw block
// header in front of each 16 KiB of the part's bytes), the content checksum; and the segment's entry in each payload's index
__global__ __launch_bounds__(256) void k_seg_compact(const EncInfo *info, const BlockPlan *plans, const SegInfo *seg, const uint8_t *__restrict__ slots, const uint32_t *__restrict__ csize,
                                                     const uint32_t *__restrict__ xsum, uint8_t *__restrict__ out)
{
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    if (info->status || info->seg_fallback || g >= info->n_segs) return;
    const SegInfo *sg = &seg[g];
    const uint32_t nrec = seg[g + 1].rec0 - sg->rec0;
    // the block: by the first record (an empty segment lies in the block of the record that follows it, or - behind a block's last
    // record - in the block before: decided by its number)
    uint32_t lo = 0, hi = info->n_blocks;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (plans[mid].seg_base <= g) lo = mid; else hi = mid; }
    const BlockPlan *p = &plans[lo];
    const uint32_t si = g - p->seg_base;
    for (int s = 0; s < FQZ_NS; s++) {
        if (!p->frame_len[s]) continue;
        uint8_t *const pay = out + p->frame_off[s];
        const uint32_t M = sg->raw[s], fl = sg->flen[s];
        if (t == 0) {
            uint8_t *e = pay + 24 + 8 * si;
            e[0] = (uint8_t)fl; e[1] = (uint8_t)(fl >> 8); e[2] = (uint8_t)(fl >> 16);
            e[3] = (uint8_t)M; e[4] = (uint8_t)(M >> 8); e[5] = (uint8_t)(M >> 16);
            e[6] = (uint8_t)nrec; e[7] = (uint8_t)(nrec >> 8);
        }
        if (!M) continue;
        uint8_t *f = pay + seg_idx_len(p->n_seg) + sg->foff[s];
        const uint32_t fh = seg_fh(M), nch = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
        if (t == 0) {
            f[0] = 0x28; f[1] = 0xB5; f[2] = 0x2F; f[3] = 0xFD;
            if (M < 256u) { f[4] = 0x24; f[5] = (uint8_t)M; }
            else { f[4] = 0x64; f[5] = (uint8_t)(M - 256u); f[6] = (uint8_t)((M - 256u) >> 8); }
            put_le32(f + fl - 4, xsum[sg->chunk0[s]]);
        }
        uint8_t *dst = f + fh;
        for (uint32_t k = 0; k < nch; k++) {
            if (s == S_SEQ) {
                const uint32_t mk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK;
                const uint32_t bh = (k + 1 == nch ? 1u : 0u) | (0u << 1) | (mk << 3);
                if (t == 0) { dst[0] = (uint8_t)bh; dst[1] = (uint8_t)(bh >> 8); dst[2] = (uint8_t)(bh >> 16); }
                seg_copy(dst + 3, slots + (size_t)sg->slot0[s] * SEG_PAGE + (size_t)k * FQZ_CHUNK, mk);
                dst += 3 + mk;
            } else {
                const uint32_t n = csize[sg->chunk0[s] + k];
                seg_copy(dst, slots + (size_t)sg->slot0[s] * SEG_PAGE + (size_t)k * FQZ_SLOT, n);
                dst += n;
            }
        }
    }
}
