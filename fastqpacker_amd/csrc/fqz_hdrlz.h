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
__constant__ const HdrCt c_hdr_ll = hdr_make_ct(HDR_LL_NORM, 6);
__constant__ const HdrCt c_hdr_ml = hdr_make_ct(HDR_ML_NORM, 6);
__constant__ const HdrCt c_hdr_of = hdr_make_ct(HDR_OF_NORM, 5);
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
struct HdrModelLds {
    uint8_t text[FQZ_CHUNK + 16];
    uint16_t s[HDR_MAX_SEQ / 2 + 2], len[HDR_MAX_SEQ / 2 + 2];   // record starts (chunk-relative) and lengths
    uint16_t hs[HDR_MAX_SEQ / 2 + 2], hl[HDR_MAX_SEQ / 2 + 2], tl[HDR_MAX_SEQ / 2 + 2]; // head start / length, tail length
    uint32_t sh[12];
    uint32_t carry[3];      // sequences, matched bytes, end of the last match so far
    uint32_t q_n;           // long literal runs waiting for a cooperative copy
    uint32_t q[64][3];      // {source (chunk-relative), destination (literal offset), bytes}
    uint32_t first, n_in;
};

// exclusive scan of one value per thread over the 256-thread workgroup with op = + or max; sh: 4 words; *total = reduction
template <bool MAX>
__device__ __forceinline__ uint32_t hdr_block_scan(uint32_t v, uint32_t *sh, uint32_t *total)
{
    const uint32_t l = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, WAVE);
        if (l >= (uint32_t)d) incl = MAX ? (t > incl ? t : incl) : incl + t;
    }
    uint32_t excl = __shfl_up(incl, 1, WAVE);
    if (l == 0) excl = 0;
    __syncthreads();
    if (l == 63) sh[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t x = sh[k];
        if (k < w) base = MAX ? (x > base ? x : base) : base + x;
        tot = MAX ? (x > tot ? x : tot) : tot + x;
    }
    *total = tot;
    return MAX ? (base > excl ? base : excl) : base + excl;
}

__device__ void hdr_model_chunk(HdrModelLds &S, const uint8_t *__restrict__ stream, const uint32_t *__restrict__ Eh, uint32_t rec0, uint32_t nrec,
                                uint32_t c0, uint32_t mk, uint2 *__restrict__ hseq, uint8_t *__restrict__ hlit, HdrSide *side)
{
    const uint32_t t = threadIdx.x;
    const uint32_t base_e = Eh[rec0], c1 = c0 + mk;
    if (t == 0) {
        // records that lie wholly inside [c0, c1): the first one that starts at or behind c0 ... the last one that ends at or before c1
        uint32_t lo = 0, hi = nrec;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (Eh[rec0 + mid] - base_e < c0) lo = mid + 1; else hi = mid; }
        const uint32_t first = lo;
        uint32_t lo2 = first, hi2 = nrec; // first record index whose END lies behind c1
        while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2) >> 1; if (Eh[rec0 + mid + 1] - base_e <= c1) lo2 = mid + 1; else hi2 = mid; }
        uint32_t n_in = lo2 - first;
        if (2 * n_in > HDR_MAX_SEQ) n_in = 0;
        S.first = first; S.n_in = n_in;
        S.carry[0] = S.carry[1] = S.carry[2] = 0;
        S.q_n = 0;
    }
    for (uint32_t i = t * 16; i < mk; i += 256 * 16) *(uint4 *)&S.text[i] = load_u128_unaligned(stream + c0 + i); // (the arena is padded: whole rows)
    __syncthreads();
    const uint32_t first = S.first, n_in = S.n_in;
    if (n_in < 2) { if (t == 0) { side->nseq = 0; side->n_lit = mk; side->sec_len = 0; } return; }
    // ---- head and tail of every record against its predecessor (a lane per record)
    for (uint32_t k = t; k < n_in; k += 256) {
        const uint32_t s = Eh[rec0 + first + k] - base_e - c0, e = Eh[rec0 + first + k + 1] - base_e - c0, len = e - s;
        uint32_t h_start = s, h_len = 0, tail = 0;
        if (k) {
            const uint32_t ps = Eh[rec0 + first + k - 1] - base_e - c0, plen = s - ps;
            const uint32_t lim = len < plen ? len : plen;
            while (h_len < lim && S.text[s + h_len] == S.text[ps + h_len]) h_len++;
            if (h_len < HDR_MIN_MATCH) { // behind the length prefixes
                h_start = s + 2; h_len = 0;
                while (2 + h_len < lim && S.text[s + 2 + h_len] == S.text[ps + 2 + h_len]) h_len++;
                if (h_len < HDR_MIN_MATCH) { h_start = s; h_len = 0; }
            }
            const uint32_t used = h_len ? (h_start + h_len) - s : 0;
            const uint32_t tlim = len - used < plen ? len - used : plen;
            while (tail < tlim && S.text[e - 1 - tail] == S.text[s - 1 - tail]) tail++;
        }
        S.s[k] = (uint16_t)s; S.len[k] = (uint16_t)len; S.hs[k] = (uint16_t)h_start; S.hl[k] = (uint16_t)h_len; S.tl[k] = (uint16_t)tail;
    }
    __syncthreads();
    // ---- candidates in stream order: A_k = tail of record k-1 (+ the head of record k when they touch), B_k = head of record k alone
    for (uint32_t k0 = 1; k0 <= n_in; k0 += 256) {
        const uint32_t k = k0 + t;
        uint32_t posA = 0, lenA = 0, posB = 0, lenB = 0, off = 0;
        if (k <= n_in) {
            const uint32_t tail_prev = k >= 2 ? S.tl[k - 1] : 0u;
            off = S.len[k - 1];
            const uint32_t s_k = k < n_in ? S.s[k] : 0u, hl = k < n_in ? S.hl[k] : 0u, hs = k < n_in ? S.hs[k] : 0u;
            const bool merged = tail_prev && hl && hs == s_k;
            if (tail_prev) { posA = (uint32_t)S.s[k - 1] + S.len[k - 1] - tail_prev; lenA = tail_prev + (merged ? hl : 0u); }
            if (hl && !merged) { posB = hs; lenB = hl; }
            if (lenA < HDR_MIN_MATCH) lenA = 0;
            if (lenB < HDR_MIN_MATCH) lenB = 0;
        }
        const uint32_t cnt = (lenA ? 1u : 0u) + (lenB ? 1u : 0u);
        const uint32_t my_end = lenB ? posB + lenB : (lenA ? posA + lenA : 0u);
        uint32_t tot_c, tot_m, tot_e;
        const uint32_t ex_c = S.carry[0] + hdr_block_scan<false>(cnt, S.sh, &tot_c);
        const uint32_t ex_m = S.carry[1] + hdr_block_scan<false>(lenA + lenB, S.sh + 4, &tot_m);
        uint32_t ex_e = hdr_block_scan<true>(my_end, S.sh + 8, &tot_e);
        ex_e = ex_e > S.carry[2] ? ex_e : S.carry[2];
        auto emit = [&](uint32_t idx, uint32_t prev_end, uint32_t matched_before, uint32_t pos, uint32_t ml) {
            const uint32_t ll = pos - prev_end, lit_off = prev_end - matched_before;
            hseq[idx] = make_uint2(ll | (ml << 16), off);
            if (ll <= 32) { for (uint32_t j = 0; j < ll; j++) hlit[lit_off + j] = S.text[prev_end + j]; }
            else {
                const uint32_t qi = atomicAdd(&S.q_n, 1u);
                if (qi < 64) { S.q[qi][0] = prev_end; S.q[qi][1] = lit_off; S.q[qi][2] = ll; }
                else for (uint32_t j = 0; j < ll; j++) hlit[lit_off + j] = S.text[prev_end + j];
            }
        };
        if (lenA) emit(ex_c, ex_e, ex_m, posA, lenA);
        if (lenB) emit(ex_c + (lenA ? 1u : 0u), lenA ? posA + lenA : ex_e, ex_m + lenA, posB, lenB);
        __syncthreads();
        if (t == 0) { S.carry[0] += tot_c; S.carry[1] += tot_m; S.carry[2] = tot_e > S.carry[2] ? tot_e : S.carry[2]; }
        // long literal runs of this strip: all threads copy
        const uint32_t qn = S.q_n < 64 ? S.q_n : 64;
        for (uint32_t qi = 0; qi < qn; qi++)
            for (uint32_t j = t; j < S.q[qi][2]; j += 256) hlit[S.q[qi][1] + j] = S.text[S.q[qi][0] + j];
        __syncthreads();
        if (t == 0) S.q_n = 0;
        __syncthreads();
    }
    // ---- the literals behind the last match
    const uint32_t nseq = S.carry[0], matched = S.carry[1], last_end = S.carry[2];
    if (nseq) for (uint32_t j = last_end + t; j < mk; j += 256) hlit[j - matched] = S.text[j];
    if (t == 0) { side->nseq = nseq; side->n_lit = mk - matched; side->sec_len = 0; }
}

// ---------------------------------------------------------------------------------------------
// k_hdr_seq: Sequences_Section of a chunk (count, modes byte = Predefined x 3, backward bitstream in ZSTD_encodeSequences
// order).  The FSE state chain is serial, so a LANE codes a chunk and a wave 64 chunks; every lane reads its sequences
// eight at a time.
// ---------------------------------------------------------------------------------------------
struct HdrBits { uint8_t *p; unsigned long long acc; uint32_t nb, n; };
__device__ __forceinline__ void hb_add(HdrBits &b, uint32_t v, uint32_t n)
{
    b.acc |= (unsigned long long)v << b.nb;
    b.nb += n;
    if (b.nb >= 32) {
        if (b.n + 4 <= HDR_SEQ_CAP) store_u32_unaligned(b.p + b.n, (uint32_t)b.acc);
        b.n += 4;
        b.acc >>= 32;
        b.nb -= 32;
    }
}
__device__ __forceinline__ uint32_t hdr_ofv(const uint2 cur, const uint2 prev, bool has_prev)
{
    // Offset_Value: 1 = "the offset of the previous sequence" (when this sequence has literals), else offset + 3; the first
    // sequence of a block is always explicit: no block depends on the offset history its predecessors leave behind
    return (has_prev && cur.y == prev.y && (cur.x & 0xFFFFu) > 0) ? 1u : cur.y + 3u;
}
__device__ __forceinline__ uint32_t hdr_fse_init(const HdrCt &ct, uint32_t sy)
{
    const uint32_t nb = (uint32_t)(ct.dnb[sy] + (1 << 15)) >> 16;
    const uint32_t value = (nb << 16) - (uint32_t)ct.dnb[sy];
    return ct.state[(value >> nb) + (uint32_t)ct.dfs[sy]];
}
__device__ __forceinline__ uint32_t hdr_fse_enc(const HdrCt &ct, HdrBits &bw, uint32_t st, uint32_t sy)
{
    const uint32_t nb = (st + (uint32_t)ct.dnb[sy]) >> 16;
    hb_add(bw, st & ((1u << nb) - 1), nb);
    return ct.state[(st >> nb) + (uint32_t)ct.dfs[sy]];
}

__device__ void hdr_encode_sequences(const uint2 *__restrict__ hseq, uint32_t nseq, uint8_t *__restrict__ dst, HdrSide *side)
{
    HdrBits bw;
    bw.p = dst; bw.acc = 0; bw.nb = 0; bw.n = 0;
    // Number_of_Sequences (1 or 2 bytes: nseq < 0x7F00) and the modes byte
    if (nseq < 128) hb_add(bw, nseq, 8);
    else { hb_add(bw, (nseq >> 8) + 128, 8); hb_add(bw, nseq & 255, 8); }
    hb_add(bw, 0, 8);
    uint32_t st_ll = 0, st_of = 0, st_ml = 0;
    bool first = true;
    for (int hi = (int)nseq - 1; hi >= 0; hi -= 8) { // sequences hi, hi-1, ... hi-7 and the one in front of them (for the repeat-offset test)
        const int lo = hi - 7 > 0 ? hi - 7 : 0;
        uint2 buf[9];
#pragma unroll
        for (int j = 0; j < 9; j++) { const int idx = hi - j; buf[j] = idx >= 0 ? hseq[idx] : make_uint2(0, 0); }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = hi - j;
            if (idx < lo) break;
            const uint2 cur = buf[j];
            const uint32_t ll = cur.x & 0xFFFFu, ml = cur.x >> 16;
            const uint32_t ofv = hdr_ofv(cur, buf[j + 1], idx > 0);
            const uint32_t lc = ll < 64 ? c_hll_code[ll] : (uint32_t)highbit32_d(ll) + 19;
            const uint32_t mlb = ml - 3, mc = mlb < 128 ? c_hml_code[mlb] : (uint32_t)highbit32_d(mlb) + 36;
            const uint32_t oc = (uint32_t)highbit32_d(ofv);
            if (first) {
                st_ml = hdr_fse_init(c_hdr_ml, mc); st_of = hdr_fse_init(c_hdr_of, oc); st_ll = hdr_fse_init(c_hdr_ll, lc);
                first = false;
            } else {
                st_of = hdr_fse_enc(c_hdr_of, bw, st_of, oc);
                st_ml = hdr_fse_enc(c_hdr_ml, bw, st_ml, mc);
                st_ll = hdr_fse_enc(c_hdr_ll, bw, st_ll, lc);
            }
            hb_add(bw, ll - c_hll_base[lc], c_hll_bits[lc]);
            hb_add(bw, ml - c_hml_base[mc], c_hml_bits[mc]);
            hb_add(bw, ofv - (1u << oc), oc);
        }
    }
    hb_add(bw, st_ml & 63, 6); // FSE_flushCState: match lengths, offsets, literal lengths
    hb_add(bw, st_of & 31, 5);
    hb_add(bw, st_ll & 63, 6);
    hb_add(bw, 1, 1);          // end mark
    // the bits that have not made a whole dword yet
    uint32_t tail_bytes = (bw.nb + 7) >> 3;
    for (uint32_t j = 0; j < tail_bytes; j++) { if (bw.n + j < HDR_SEQ_CAP) dst[bw.n + j] = (uint8_t)(bw.acc >> (8 * j)); }
    bw.n += tail_bytes;
    side->sec_len = bw.n >= FQZ_CHUNK ? HDR_OVERFLOW : bw.n; // a section that long cannot beat the Raw block
}
