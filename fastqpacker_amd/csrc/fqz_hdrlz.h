// fqz_hdrlz.h — header-stream modelling of the FQZ-H2 profile on the device (byte-identical to oracle/fqz_entropy.c:
// hdr_chunk_model / hdr_write_sequences).
//
// The headers stream is [u16 H][H bytes] per record (compress.go:514-515); consecutive Illumina headers share a long
// prefix and a long suffix, which an order-0 coder cannot see (the reference's zstd level 1 does: EncodeAll on the headers
// stream, compress.go:525).  Inside one 16 KiB chunk (= one zstd block; matches never leave the block, so blocks stay
// independent and are modelled, coded and decoded in parallel) every record that lies inside the chunk together with its
// predecessor p yields
//   head_i : the common prefix with p (offset len_p); when the u16 length prefixes differ it starts behind them, at byte 2
//   tail_i : the common suffix with p of what the head left (offset len_i)
// tail_i and head_{i+1} share their offset (len_i) and are merged when they touch.  Candidates of at least HDR_MIN_MATCH
// bytes become zstd sequences (literal length, match length, offset) coded with the PREDEFINED FSE tables (RFC 8878
// 3.1.1.3.2.2: no table description to build); the bytes between them are the block's literals, Huffman-coded with the
// group's table as before.
#pragma once
#include "fqz_device.h"

#define HDR_MIN_MATCH 6u
#define HDR_MAX_SEQ 2048u            // sequences per chunk; a chunk with more than HDR_MAX_SEQ / 2 records inside is coded without matches
#define HDR_SEQ_CAP (FQZ_CHUNK + 64u) // bytes of a Sequences_Section kept (one that long never beats the Raw block)
#define HDR_OVERFLOW 0xFFFFFFFFu

struct HdrSide { uint32_t nseq, n_lit, sec_len, pad; }; // per headers chunk (ordinal): sequences, literal bytes, Sequences_Section bytes

// ---- predefined distributions (RFC 8878 3.1.1.3.2.2.1-3) and their compression tables (FSE_buildCTable), at compile time
struct HdrCt { uint16_t state[64]; int32_t dnb[53]; int32_t dfs[53]; };
constexpr int hdr_highbit(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }
template <int NSYM>
constexpr HdrCt hdr_make_ct(const short (&norm)[NSYM], int log)
{
    HdrCt ct{};
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    int cumul[NSYM + 1] = {};
    uint8_t tsym[64] = {};
    int high = size - 1;
    for (int u = 1; u <= NSYM; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tsym[high--] = (uint8_t)(u - 1); }
        else cumul[u] = cumul[u - 1] + norm[u - 1];
    }
    int pos = 0;
    for (int sy = 0; sy < NSYM; sy++)
        for (int k = 0; k < norm[sy]; k++) {
            tsym[pos] = (uint8_t)sy;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    for (int u = 0; u < size; u++) { const int sy = tsym[u]; ct.state[cumul[sy]++] = (uint16_t)(size + u); }
    int total = 0;
    for (int sy = 0; sy < NSYM; sy++) {
        const int n = norm[sy];
        if (n == 0) { ct.dnb[sy] = ((log + 1) << 16) - (1 << log); ct.dfs[sy] = 0; }
        else if (n == -1 || n == 1) { ct.dnb[sy] = (log << 16) - (1 << log); ct.dfs[sy] = total - 1; total++; }
        else {
            const int maxbits = log - hdr_highbit((uint32_t)(n - 1));
            ct.dnb[sy] = (maxbits << 16) - (n << maxbits);
            ct.dfs[sy] = total - n;
            total += n;
        }
    }
    return ct;
}
constexpr short HDR_LL_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
constexpr short HDR_ML_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
constexpr short HDR_OF_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__constant__ const uint32_t c_hll_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                              48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__constant__ const uint8_t c_hll_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ const uint32_t c_hml_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                              30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                              16387, 32771, 65539};
__constant__ const uint8_t c_hml_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                             0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ const uint8_t c_hll_code[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                                             22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
__constant__ const uint8_t c_hml_code[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
                                              32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
                                              40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
                                              42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};

// ---------------------------------------------------------------------------------------------
// k_hdr_model: one workgroup per headers chunk -> its sequences (hseq: ll | ml << 16, offset) and its literals (hlit)
// ---------------------------------------------------------------------------------------------
#define HDR_RUN_MAX 64u  // literal runs up to this long are copied by their own lane
#define HDR_STAGE 4096u  // literal bytes staged in LDS before they leave in 16-byte pieces
#define HDR_TPAD 16u // bytes in front of the chunk in LDS: the backward compares read 8 bytes at a time
struct HdrModelLds {
    uint8_t text[HDR_TPAD + FQZ_CHUNK + 16];
    uint16_t e[HDR_MAX_SEQ / 2 + 2];                             // record starts, chunk-relative (e[n_in] = end of the last record)
    uint16_t hs[HDR_MAX_SEQ / 2 + 2], hl[HDR_MAX_SEQ / 2 + 2], tl[HDR_MAX_SEQ / 2 + 2]; // head start / length, tail length
    uint32_t sh[8];
    uint32_t carry[2];      // sequences << 16 | matched bytes; end of the last match so far
    uint32_t q_n;           // long literal runs (> HDR_RUN_MAX bytes) of the strip: all threads copy them
    uint32_t q[FQZ_CHUNK / HDR_RUN_MAX + 4][3]; // {source (chunk-relative), destination (literal offset), bytes}; a chunk cannot hold more
    __attribute__((aligned(16))) uint8_t stage[HDR_STAGE + 16];
    uint32_t hist[256];     // byte histogram of the chunk's literals (the entropy stage builds its table from these)
};
__device__ __forceinline__ unsigned long long hdr_ld64(const uint8_t *p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }

// exclusive scans over the 256-thread workgroup: a = running sum, m = running maximum; totals in *ta, *tm.  sh: 8 words
__device__ __forceinline__ void hdr_scan2(uint32_t a, uint32_t m, uint32_t *sh, uint32_t *ex_a, uint32_t *ex_m, uint32_t *ta, uint32_t *tm)
{
    const uint32_t l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t ia = wave_incl_scan(a);
    uint32_t im = m;
#define HDR_DPP_MAX(v, ctrl, rows) do { const uint32_t x_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xF, false); v = x_ > v ? x_ : v; } while (0)
    HDR_DPP_MAX(im, 0x111, 0xF); HDR_DPP_MAX(im, 0x112, 0xF); HDR_DPP_MAX(im, 0x114, 0xF); HDR_DPP_MAX(im, 0x118, 0xF); // row_shr 1, 2, 4, 8
    HDR_DPP_MAX(im, 0x142, 0xA); HDR_DPP_MAX(im, 0x143, 0xC);                                                           // row_bcast 15, 31
#undef HDR_DPP_MAX
    uint32_t ea = ia - a, em = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)im, 0x138, 0xF, 0xF, false); // wave_shr:1 (lane 0 gets 0)
    __syncthreads(); // (the previous use of sh)
    if (l == 63) { sh[w] = ia; sh[4 + w] = im; }
    __syncthreads();
    uint32_t ba = 0, bm = 0, sa = 0, sm = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t xa = sh[k], xm = sh[4 + k];
        if (k < w) { ba += xa; bm = xm > bm ? xm : bm; }
        sa += xa; sm = xm > sm ? xm : sm;
    }
    *ex_a = ba + ea; *ex_m = bm > em ? bm : em; *ta = sa; *tm = sm;
}

__device__ void hdr_model_chunk(HdrModelLds &S, const uint8_t *__restrict__ stream, const uint32_t *__restrict__ Eh, uint32_t rec0, uint32_t nrec,
                                uint32_t c0, uint32_t mk, uint2 *__restrict__ hseq, uint8_t *__restrict__ hlit, HdrSide *side, uint16_t *__restrict__ hhist)
{
    const uint32_t t = threadIdx.x;
    S.hist[t] = 0;
    // the whole chunk is its own literal run (no sequences): histogram of the text in LDS; ends with the counts in hhist
    auto hist_all = [&]() {
        for (uint32_t i = t * 4; i < mk; i += 256 * 4) {
            const uint32_t w = *(const uint32_t *)&S.text[HDR_TPAD + i], n = mk - i < 4 ? mk - i : 4;
            for (uint32_t q = 0; q < n; q++) atomicAdd(&S.hist[(w >> (8 * q)) & 0xFF], 1u);
        }
        __syncthreads();
        hhist[t] = (uint16_t)S.hist[t];
    };
    const uint32_t base_e = Eh[rec0], c1 = c0 + mk;
    uint8_t *const text = S.text + HDR_TPAD;
    for (uint32_t i = t * 16; i < mk; i += 256 * 16) *(uint4 *)&text[i] = load_u128_unaligned(stream + c0 + i); // (in flight during the search)
    // ---- the records that lie wholly inside [c0, c1): first = #records that start before c0, last1 = #records that end at or
    //      before c1; two-level search, 256 probes a level (the record table is a sorted array in global memory)
    const uint32_t seg = (nrec + 255) / 256;
    uint32_t first, last1;
    {
        const uint32_t i1 = t * seg;
        const uint32_t v = i1 < nrec ? Eh[rec0 + i1] - base_e : 0xFFFFFFFFu, v2 = i1 < nrec ? Eh[rec0 + i1 + 1] - base_e : 0xFFFFFFFFu;
        const uint32_t j = (uint32_t)__syncthreads_count(v < c0), j2 = (uint32_t)__syncthreads_count(v2 <= c1); // segments that begin with a hit
        uint32_t hit = 0, hit2 = 0;
        for (uint32_t u = t; u < seg; u += 256) { // inside the last such segment
            if (j) { const uint32_t r = (j - 1) * seg + u; hit += r < nrec && Eh[rec0 + r] - base_e < c0; }
            if (j2) { const uint32_t r = (j2 - 1) * seg + u; hit2 += r < nrec && Eh[rec0 + r + 1] - base_e <= c1; }
        }
        uint32_t h1 = 0, h2 = 0;
        for (uint32_t q = 0; q < 32; q++) { // (seg <= 2^32 / 256: a count per thread of at most seg / 256 + 1; summed bit by bit)
            h1 += (uint32_t)__syncthreads_count((hit >> q) & 1u) << q;
            h2 += (uint32_t)__syncthreads_count((hit2 >> q) & 1u) << q;
            if (!__syncthreads_or((hit >> (q + 1)) | (hit2 >> (q + 1)))) break;
        }
        first = j ? (j - 1) * seg + h1 : 0;
        last1 = j2 ? (j2 - 1) * seg + h2 : 0;
    }
    uint32_t n_in = last1 > first ? last1 - first : 0;
    if (2 * n_in > HDR_MAX_SEQ) n_in = 0;
    if (n_in < 2) {
        if (t == 0) { side->nseq = 0; side->n_lit = mk; side->sec_len = 0; side->pad = 0; }
        __syncthreads(); // (the text is in LDS)
        hist_all();
        return;
    }
    for (uint32_t k = t; k <= n_in; k += 256) S.e[k] = (uint16_t)(Eh[rec0 + first + k] - base_e - c0);
    if (t == 0) { S.carry[0] = S.carry[1] = 0; S.q_n = 0; }
    if (t < 4) ((uint32_t *)S.text)[t] = 0;
    __syncthreads();
    // ---- head and tail of every record against its predecessor (a lane per record, 8 bytes per compare)
    for (uint32_t k = t; k < n_in; k += 256) {
        uint32_t h_start = S.e[k], h_len = 0, tail = 0;
        if (k) {
            const uint32_t s = S.e[k], e = S.e[k + 1], len = e - s, ps = S.e[k - 1], plen = s - ps;
            const uint32_t lim = len < plen ? len : plen;
            auto prefix = [&](uint32_t from) { // common prefix of the two records from byte `from` on, at most lim - from
                uint32_t n = 0;
                const uint32_t room = lim - from;
                while (n < room) {
                    const unsigned long long x = hdr_ld64(text + s + from + n) ^ hdr_ld64(text + ps + from + n);
                    if (x) { n += (uint32_t)__builtin_ctzll(x) >> 3; break; }
                    n += 8;
                }
                return n < room ? n : room;
            };
            h_len = prefix(0);
            if (h_len < HDR_MIN_MATCH) { // behind the length prefixes
                h_start = s + 2;
                h_len = lim > 2 ? prefix(2) : 0;
                if (h_len < HDR_MIN_MATCH) { h_start = s; h_len = 0; }
            }
            const uint32_t used = h_len ? (h_start + h_len) - s : 0;
            const uint32_t tlim = len - used < plen ? len - used : plen;
            while (tail < tlim) {
                const unsigned long long x = hdr_ld64(text + e - tail - 8) ^ hdr_ld64(text + s - tail - 8);
                if (x) { tail += (uint32_t)__builtin_clzll(x) >> 3; break; }
                tail += 8;
            }
            tail = tail < tlim ? tail : tlim;
        }
        S.hs[k] = (uint16_t)h_start; S.hl[k] = (uint16_t)h_len; S.tl[k] = (uint16_t)tail;
    }
    __syncthreads();
    // ---- candidates in stream order: A_k = tail of record k-1 (+ the head of record k when they touch), B_k = head of record k alone.
    //      The literals (the bytes between the matches) are compacted through a staging window in LDS and leave in whole
    //      16-byte pieces: a store per literal byte and lane is what the vector memory pipeline is slowest at
    uint32_t a0 = 0; // literal offset of stage[0] (a multiple of 16); stage[0 .. lit_end - a0) is filled and not yet flushed
    for (uint32_t k0 = 1; k0 <= n_in; k0 += 256) {
        const uint32_t k = k0 + t;
        uint32_t posA = 0, lenA = 0, posB = 0, lenB = 0, off = 0;
        if (k <= n_in) {
            const uint32_t tail_prev = k >= 2 ? S.tl[k - 1] : 0u;
            off = (uint32_t)S.e[k] - S.e[k - 1];
            const uint32_t s_k = S.e[k], hl = k < n_in ? S.hl[k] : 0u, hs = k < n_in ? S.hs[k] : 0u;
            const bool merged = tail_prev && hl && hs == s_k;
            if (tail_prev) { posA = s_k - tail_prev; lenA = tail_prev + (merged ? hl : 0u); }
            if (hl && !merged) { posB = hs; lenB = hl; }
            if (lenA < HDR_MIN_MATCH) lenA = 0;
            if (lenB < HDR_MIN_MATCH) lenB = 0;
        }
        const uint32_t cnt = (lenA ? 1u : 0u) + (lenB ? 1u : 0u);
        const uint32_t my_end = lenB ? posB + lenB : (lenA ? posA + lenA : 0u);
        uint32_t ex, ex_e, tot, tot_e;
        hdr_scan2((cnt << 16) | (lenA + lenB), my_end, S.sh, &ex, &ex_e, &tot, &tot_e);
        const uint32_t c0v = S.carry[0], c1v = S.carry[1];
        ex += c0v;
        ex_e = ex_e > c1v ? ex_e : c1v;
        const uint32_t ex_c = ex >> 16, ex_m = ex & 0xFFFFu;
        // this thread's literal runs: text[src, src + ll) -> literal offset dst
        uint32_t srcA = 0, dstA = 0, llA = 0, srcB = 0, dstB = 0, llB = 0;
        if (lenA) { srcA = ex_e; llA = posA - ex_e; dstA = ex_e - ex_m; hseq[ex_c] = make_uint2(llA | (lenA << 16), off); }
        if (lenB) {
            srcB = lenA ? posA + lenA : ex_e; llB = posB - srcB; dstB = srcB - (ex_m + lenA);
            hseq[ex_c + (lenA ? 1u : 0u)] = make_uint2(llB | (lenB << 16), off);
        }
        if (llA > HDR_RUN_MAX) { const uint32_t qi = atomicAdd(&S.q_n, 1u); S.q[qi][0] = srcA; S.q[qi][1] = dstA; S.q[qi][2] = llA; llA = 0; }
        if (llB > HDR_RUN_MAX) { const uint32_t qi = atomicAdd(&S.q_n, 1u); S.q[qi][0] = srcB; S.q[qi][1] = dstB; S.q[qi][2] = llB; llB = 0; }
        const uint32_t new_end = tot_e > c1v ? tot_e : c1v;
        const uint32_t lit_end = new_end - ((c0v + tot) & 0xFFFFu); // literals in front of the last match so far
        __syncthreads();
        if (t == 0) { S.carry[0] = c0v + tot; S.carry[1] = new_end; }
        const uint32_t qn = S.q_n;
        for (;;) { // windows of HDR_STAGE literal bytes (one, unless the strip holds long runs)
            const uint32_t w_end = a0 + HDR_STAGE;
            auto put = [&](uint32_t src, uint32_t dst, uint32_t ll) {
                const uint32_t a = dst > a0 ? dst : a0, b = dst + ll < w_end ? dst + ll : w_end;
                for (uint32_t j = a; j < b; j++) { const uint8_t v = text[src + (j - dst)]; S.stage[j - a0] = v; atomicAdd(&S.hist[v], 1u); }
            };
            if (llA) put(srcA, dstA, llA);
            if (llB) put(srcB, dstB, llB);
            for (uint32_t qi = 0; qi < qn; qi++) { // long runs: all threads copy
                const uint32_t src = S.q[qi][0], dst = S.q[qi][1], ll = S.q[qi][2];
                const uint32_t a = dst > a0 ? dst : a0, b = dst + ll < w_end ? dst + ll : w_end;
                for (uint32_t j = a + t; j < b; j += 256) { const uint8_t v = text[src + (j - dst)]; S.stage[j - a0] = v; atomicAdd(&S.hist[v], 1u); }
            }
            __syncthreads();
            const uint32_t have_end = lit_end < w_end ? lit_end : w_end, flush_end = have_end & ~15u;
            for (uint32_t j = a0 + t * 16; j < flush_end; j += 256 * 16) *(uint4 *)(hlit + j) = *(const uint4 *)&S.stage[j - a0];
            uint32_t rem = 0;
            if (t < have_end - flush_end) rem = S.stage[flush_end - a0 + t]; // (< 16 bytes wait for the next pieces)
            __syncthreads();
            if (t < have_end - flush_end) S.stage[t] = (uint8_t)rem;
            a0 = flush_end;
            if (lit_end <= w_end) break;
        }
        __syncthreads();
        if (t == 0) S.q_n = 0;
    }
    __syncthreads();
    // ---- what is left in the stage, then the literals behind the last match
    const uint32_t nseq = S.carry[0] >> 16, matched = S.carry[0] & 0xFFFFu, last_end = S.carry[1];
    if (nseq) {
        const uint32_t lit_end = last_end - matched;
        if (t < lit_end - a0) hlit[a0 + t] = S.stage[t];
        for (uint32_t j = last_end + t; j < mk; j += 256) { const uint8_t v = text[j]; hlit[j - matched] = v; atomicAdd(&S.hist[v], 1u); }
        __syncthreads();
        hhist[t] = (uint16_t)S.hist[t];
    } else hist_all(); // (records inside, but no candidate long enough)
    if (t == 0) { side->nseq = nseq; side->n_lit = mk - matched; side->sec_len = 0; side->pad = 0; }
}

// The lengths stream (u32 per read, compress.go:501) through the same machinery: a chunk whose values are all equal - fixed-length
// reads - is 4-periodic, which an order-0 coder cannot see; it becomes its first value as literals + ONE sequence {4 literals,
// match of mk - 4 bytes at offset 4} (oracle fqzo_entropy_encode_stream_v, stream 5).  Any other chunk: no sequences, its
// histogram for the entropy stage.  sh: 256 words of LDS.
__device__ void len_model_chunk(uint32_t *sh, const uint8_t *__restrict__ chunk, uint32_t mk, uint2 *__restrict__ hseq, uint8_t *__restrict__ hlit, HdrSide *side,
                                uint16_t *__restrict__ hhist)
{
    const uint32_t t = threadIdx.x;
    const uint32_t *w = (const uint32_t *)chunk; // (16-byte aligned: the stream starts aligned and chunks are 16 KiB)
    const uint32_t nw = mk >> 2, w0 = nw ? w[0] : 0u;
    uint32_t diff = 0;
    for (uint32_t i = t; i < nw; i += 256) diff |= w[i] ^ w0;
    sh[t] = 0;
    const bool periodic = !__syncthreads_or(diff != 0) && mk >= 20 && (mk & 3) == 0;
    if (periodic) {
        if (t == 0) {
            hseq[0] = make_uint2(4u | ((mk - 4) << 16), 4u);
            for (int q = 0; q < 4; q++) { const uint32_t b = (w0 >> (8 * q)) & 0xFF; hlit[q] = (uint8_t)b; sh[b]++; }
            side->nseq = 1; side->n_lit = 4; side->sec_len = 0; side->pad = 0;
        }
    } else {
        for (uint32_t i = t; i < mk; i += 256) atomicAdd(&sh[chunk[i]], 1u);
        if (t == 0) { side->nseq = 0; side->n_lit = mk; side->sec_len = 0; side->pad = 0; }
    }
    __syncthreads();
    hhist[t] = (uint16_t)sh[t];
}

// ---------------------------------------------------------------------------------------------
// Sequences_Section of a chunk (count, modes byte = Predefined x 3, backward bitstream in ZSTD_encodeSequences order).
// The only serial part is the three FSE state chains (literal length, match length, offset codes, each walked from the
// last sequence to the first); what a state emits at a step does not depend on the other two.  So:
//   hdr_seq_chains   a LANE per chain, 16 chunks per wave: per sequence the bits each chain emits (hst: value | nbits)
//   hdr_seq_pack     a wave per chunk: every lane assembles the bits of one sequence (state bits + extra bits), a wave scan
//                    places them, LDS atomics merge them
// ---------------------------------------------------------------------------------------------
#define HDR_CB 24u // sequences per trip of hdr_seq_chains
// FSE_encodeSymbol WITHOUT a memory access on the state chain.  A symbol with normalised count c owns c entries of the
// state table (c <= 4 in the predefined distributions), and the step picks one of them:
//   nb = (state + deltaNbBits[sym]) >> 16,  k = (state >> nb) - c,  next state = entry k of the symbol
// so what the chain needs per sequence - the symbol's (up to) four next states packed in one word, its deltaNbBits and c -
// is looked up beforehand, for a whole trip of sequences side by side, and the walk itself is seven register operations a
// step.  (The chains run beside the entropy coder, which saturates the LDS pipeline: a dependent LDS read a step took four
// times as long there as on an idle chip.)  States count from 0: the real state is that + the table size.
struct HdrTrans { uint32_t cand[3][56]; int32_t dnb[3][56]; uint8_t cnt[3][56], init[3][56]; };
template <int NSYM>
constexpr void hdr_make_trans(const short (&norm)[NSYM], int log, uint32_t *cand, int32_t *dnb, uint8_t *cnt, uint8_t *init)
{
    const HdrCt ct = hdr_make_ct(norm, log);
    const int size = 1 << log;
    int cumul = 0;
    for (int sy = 0; sy < NSYM; sy++) {
        const int c = norm[sy] == -1 ? 1 : norm[sy];
        uint32_t packed = 0;
        for (int k = 0; k < c && k < 4; k++) packed |= (uint32_t)(ct.state[cumul + k] - size) << (8 * k);
        cand[sy] = packed;
        cnt[sy] = (uint8_t)c;
        dnb[sy] = ct.dnb[sy];
        const uint32_t nb0 = (uint32_t)(ct.dnb[sy] + (1 << 15)) >> 16;
        init[sy] = (uint8_t)(ct.state[(((nb0 << 16) - (uint32_t)ct.dnb[sy]) >> nb0) + (uint32_t)ct.dfs[sy]] - size);
        cumul += c;
    }
}
constexpr HdrTrans hdr_make_all_trans()
{
    HdrTrans t{};
    hdr_make_trans(HDR_LL_NORM, 6, t.cand[0], t.dnb[0], t.cnt[0], t.init[0]);
    hdr_make_trans(HDR_ML_NORM, 6, t.cand[1], t.dnb[1], t.cnt[1], t.init[1]);
    hdr_make_trans(HDR_OF_NORM, 5, t.cand[2], t.dnb[2], t.cnt[2], t.init[2]);
    return t;
}
__constant__ const HdrTrans c_hdr_trans = hdr_make_all_trans();
// what a chain needs for one sequence in ONE 64-bit LDS read, indexed by the value itself when it is small (literal length
// < 64, match length - 3 < 128: nearly always) and by 64 / 128 + code otherwise; offsets: by code.
// x = the symbol's candidate next states, y = deltaNbBits (20 bits) | count << 20 | code << 23
struct HdrChainLds {
    uint2 ll[64 + 36], ml[128 + 53], of[32];
};
__device__ __forceinline__ uint32_t hdr_ofv(const uint2 cur, const uint2 prev, bool has_prev)
{
    // Offset_Value: 1 = "the offset of the previous sequence" (when this sequence has literals), else offset + 3; the first
    // sequence of a block is always explicit: no block depends on the offset history its predecessors leave behind
    return (has_prev && cur.y == prev.y && (cur.x & 0xFFFFu) > 0) ? 1u : cur.y + 3u;
}
__device__ __forceinline__ void hdr_chain_tables(HdrChainLds &T)
{
    const uint32_t lane = threadIdx.x;
    auto entry = [](int chain, uint32_t code) { return make_uint2(c_hdr_trans.cand[chain][code], (uint32_t)c_hdr_trans.dnb[chain][code] | ((uint32_t)c_hdr_trans.cnt[chain][code] << 20) | (code << 23)); };
    T.ll[lane] = entry(0, c_hll_code[lane]);
    if (lane < 36) T.ll[64 + lane] = entry(0, lane);
    T.ml[lane] = entry(1, c_hml_code[lane]);
    T.ml[64 + lane] = entry(1, c_hml_code[64 + lane]);
    if (lane < 53) T.ml[128 + lane] = entry(1, lane);
    if (lane < 29) T.of[lane] = entry(2, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <int K>
__device__ __forceinline__ uint32_t hdr_quad(uint32_t v) // the value of lane K of this lane's quad (DPP: no LDS round trip)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}
// lane = 4 * (chunk of the wave) + chain (0 literal lengths, 1 match lengths, 2 offsets; 3 idles).
// hst[i] = ll bits | ml bits << 9 | of bits << 18, each value | nbits << 6; side->pad = the three final states.
// A trip takes HDR_CB sequences: their loads are in flight together, their codes and table rows are looked up side by side,
// and only the state walk itself (one dependent LDS read a step) is serial.
__device__ void hdr_seq_chains(HdrChainLds &T, const uint2 *__restrict__ hseq, uint32_t nseq, uint32_t *__restrict__ hst, HdrSide *side, uint32_t c, bool on)
{
    uint32_t st = 0;
    const uint32_t cc = c < 3 ? c : 2u; // (the idle lane walks along with the offsets chain: its results are dropped)
    const uint32_t tsize = cc == 2 ? 32u : 64u;
    // trips over aligned groups of HDR_CB sequences, from the last group down: 16-byte loads (two sequences each), all in flight
    for (int base = on ? ((int)nseq - 1) / (int)HDR_CB * (int)HDR_CB : -1; base >= 0; base -= (int)HDR_CB) {
        uint4 pr[HDR_CB / 2];
#pragma unroll
        for (int k = 0; k < (int)HDR_CB / 2; k++) pr[k] = ((const uint4 *)(hseq + base))[k]; // (past nseq: inside the side buffers, not used)
        const uint2 prev0 = base > 0 ? hseq[base - 1] : make_uint2(0, 0);
        uint2 ent[HDR_CB]; // of the symbol of this lane's chain for sequence base + j
#pragma unroll
        for (int j = 0; j < (int)HDR_CB; j++) {
            const bool live = base + j < (int)nseq;
            uint2 cur = (j & 1) ? make_uint2(pr[j >> 1].z, pr[j >> 1].w) : make_uint2(pr[j >> 1].x, pr[j >> 1].y);
            if (!live) cur = make_uint2(3u << 16, 0);
            const uint2 prv = j ? ((j & 1) ? make_uint2(pr[j >> 1].x, pr[j >> 1].y) : make_uint2(pr[(j - 1) >> 1].z, pr[(j - 1) >> 1].w)) : prev0;
            if (cc == 0) { const uint32_t ll = cur.x & 0xFFFFu; ent[j] = T.ll[ll < 64 ? ll : 64u + (uint32_t)highbit32_d(ll) + 19u]; }
            else if (cc == 1) { const uint32_t mlb = (cur.x >> 16) - 3; ent[j] = T.ml[mlb < 128 ? mlb : 128u + (uint32_t)highbit32_d(mlb) + 36u]; }
            else ent[j] = T.of[(uint32_t)highbit32_d(hdr_ofv(cur, prv, base + j > 0)) & 31u];
        }
        uint32_t outv[HDR_CB];
#pragma unroll
        for (int j = (int)HDR_CB - 1; j >= 0; j--) { // the walk, last sequence of the group first: registers only
            const bool live = base + j < (int)nseq, first = base + j == (int)nseq - 1;
            const uint32_t y = ent[j].y;
            if (first) st = c_hdr_trans.init[cc][(y >> 23) & 63u]; // the last sequence opens the chain: no output
            const uint32_t full = st + tsize, nb = (full + (y & 0xFFFFFu)) >> 16, k = ((full >> nb) - ((y >> 20) & 7u)) & 3u;
            const uint32_t nxt = (ent[j].x >> (8 * k)) & 0xFFu;
            outv[j] = first ? 0u : ((st & ((1u << nb) - 1)) | (nb << 6));
            st = (live && !first) ? nxt : st;
        }
#pragma unroll
        for (int j = 0; j < (int)HDR_CB; j++) {
            const uint32_t v = hdr_quad<0>(outv[j]) | (hdr_quad<1>(outv[j]) << 9) | (hdr_quad<2>(outv[j]) << 18);
            if (c == 0 && base + j < (int)nseq) hst[base + j] = v;
        }
    }
    const uint32_t fin = hdr_quad<0>(st) | (hdr_quad<1>(st) << 8) | (hdr_quad<2>(st) << 16);
    if (on && c == 0) side->pad = fin;
}

#define HDR_PW 160u // dwords of the packing window: a batch of 64 sequences is at most 64 x 66 bits = 132 dwords
struct HdrPackLds { uint32_t w[HDR_PW]; uint8_t ll_code[64], ml_code[128]; };
__device__ __forceinline__ void hdr_or_bits(uint32_t *w, uint32_t bitpos, unsigned long long v, uint32_t n) // bitpos: relative to the window
{
    if (!n) return;
    const uint32_t i = bitpos >> 5, sh = bitpos & 31;
    atomicOr(&w[i], (uint32_t)(v << sh));
    const unsigned long long hi = sh ? v >> (32 - sh) : v >> 16 >> 16;
    if (sh + n > 32) atomicOr(&w[i + 1], (uint32_t)hi);
    if (sh + n > 64) atomicOr(&w[i + 2], (uint32_t)(hi >> 32));
}
// one wave per chunk.  The section is assembled a batch of 64 sequences at a time in a small LDS window; the dwords a batch
// completes go to dst at once (dst is dword aligned), the one it leaves open stays for the next batch.
__device__ void hdr_seq_pack(HdrPackLds &S, const uint2 *__restrict__ hseq, uint32_t nseq, const uint32_t *__restrict__ hst, uint8_t *__restrict__ dst, HdrSide *side)
{
    const uint32_t lane = threadIdx.x;
    S.ll_code[lane] = c_hll_code[lane];
    S.ml_code[lane] = c_hml_code[lane];
    S.ml_code[64 + lane] = c_hml_code[64 + lane];
    for (uint32_t i = lane; i < HDR_PW; i += 64) S.w[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // Number_of_Sequences (1 or 2 bytes: nseq < 0x7F00) and the modes byte
    const uint32_t hb = nseq < 128 ? 2u : 3u;
    if (lane == 0) S.w[0] = nseq < 128 ? nseq : (((nseq >> 8) + 128) | ((nseq & 255) << 8));
    uint32_t P = 8 * hb; // bits of the section so far
    uint32_t w0 = 0;     // dword of the section that S.w[0] holds
    uint32_t *dst32 = (uint32_t *)dst;
    auto drain = [&]() { // whole dwords below P leave; the open one moves to the front
        const uint32_t full = (P >> 5) - w0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t keep[3];
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) { const uint32_t i = lane + 64 * r; keep[r] = i < HDR_PW ? S.w[i] : 0u; }
        const uint32_t open = S.w[full < HDR_PW ? full : 0];
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) { const uint32_t i = lane + 64 * r; if (i < full && w0 + i < HDR_SEQ_CAP / 4) dst32[w0 + i] = keep[r]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < HDR_PW; i += 64) S.w[i] = (i == 0) ? open : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        w0 += full;
    };
    for (int top = (int)nseq - 1; top >= 0; top -= 64) {
        const int i = top - (int)lane;
        uint32_t n1 = 0, n2 = 0;
        unsigned long long v1 = 0, v2 = 0;
        if (i >= 0) {
            const uint2 cur = hseq[i];
            const uint2 prev = i > 0 ? hseq[i - 1] : make_uint2(0, 0);
            const uint32_t ll = cur.x & 0xFFFFu, ml = cur.x >> 16, ofv = hdr_ofv(cur, prev, i > 0);
            const uint32_t lc = ll < 64 ? S.ll_code[ll] : (uint32_t)highbit32_d(ll) + 19;
            const uint32_t mlb = ml - 3, mc = mlb < 128 ? S.ml_code[mlb] : (uint32_t)highbit32_d(mlb) + 36;
            const uint32_t oc = (uint32_t)highbit32_d(ofv);
            if ((uint32_t)i + 1 < nseq) { // state bits: offsets, match lengths, literal lengths
                const uint32_t h = hst[i];
                const uint32_t bl = h & 63, nl = (h >> 6) & 7, bm = (h >> 9) & 63, nm = (h >> 15) & 7, bo = (h >> 18) & 63, no = (h >> 24) & 7;
                v1 = (unsigned long long)bo | ((unsigned long long)bm << no) | ((unsigned long long)bl << (no + nm));
                n1 = no + nm + nl;
            }
            const uint32_t lb = c_hll_bits[lc], mb = c_hml_bits[mc]; // extra bits: literal length, match length, offset
            v2 = (unsigned long long)(ll - c_hll_base[lc]) | ((unsigned long long)(ml - c_hml_base[mc]) << lb) | ((unsigned long long)(ofv - (1u << oc)) << (lb + mb));
            n2 = lb + mb + oc;
        }
        const uint32_t incl = wave_incl_scan(n1 + n2);
        const uint32_t at = P - 32 * w0 + incl - (n1 + n2);
        hdr_or_bits(S.w, at, v1, n1);
        hdr_or_bits(S.w, at + n1, v2, n2);
        P += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        drain();
    }
    if (lane == 0) { // FSE_flushCState: match lengths, offsets, literal lengths; then the end mark
        const uint32_t fin = side->pad;
        const unsigned long long tail = (unsigned long long)((fin >> 8) & 63) | ((unsigned long long)((fin >> 16) & 31) << 6) | ((unsigned long long)(fin & 63) << 11) | (1ull << 17);
        hdr_or_bits(S.w, P - 32 * w0, tail, 18);
    }
    P += 18;
    const uint32_t bytes = (P + 7) >> 3;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (bytes >= FQZ_CHUNK) { if (lane == 0) side->sec_len = HDR_OVERFLOW; return; } // (cannot happen for a block judged compressible: HDR_SSZ_BOUND)
    if (lane < 2 && w0 + lane < HDR_SEQ_CAP / 4 && 4 * (w0 + lane) < bytes) dst32[w0 + lane] = S.w[lane];
    if (lane == 0) side->sec_len = bytes;
}
