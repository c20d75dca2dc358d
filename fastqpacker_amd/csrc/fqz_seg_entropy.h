// fqz_seg_entropy.h — the entropy stage of the segment path (fqz_seg.h): one part of a segment (<= 4 chunks of 16 KiB, lying in
// LDS) -> its zstd blocks, by a workgroup of 512 threads.
//
// The same deterministic construction as entropy_encode_group<false> (fqz_entropy_dev.h; oracle encode_group_chunks): one
// Huffman table for the part, a block per 16 KiB chunk, Raw / RLE by the FQZ-H2 tests.  What differs is how the work lies on
// the chip: a segment workgroup is alone with its latencies (two workgroups a CU, nothing else to switch to), so
//  * the table (histogram -> sorted symbols -> two-queue merge -> depths -> weights -> tree description -> canonical codes) is
//    built once by threads 0..255 - it is one dependent chain, the other 256 threads wait at its barriers -
//  * and the chunks are then coded two at a time, one by each half of the workgroup,
//  * in place: once a half has its chunk's symbols in registers, the chunk's bytes in LDS are dead and its 16 KiB become the
//    staging area of the block (a Compressed block is smaller than its chunk by construction, else the chunk is stored Raw),
//    so the part's own LDS is the only LDS the coding takes.
// The caller has copied the part to the stream arena before (content checksum, fqz_seg.h P4).
#pragma once
#include "fqz_entropy_dev.h"

#define SEGE_NT 512u

// part: the M bytes in LDS, 16-aligned, writable, followed by at least 16 bytes of slack; chunk k -> slot0 + k * FQZ_SLOT, csize0[k].
// xh != nullptr: the byte histogram of every chunk, [chunk][256] (made while the part was written).  All 512 threads call this.
// gcopy: the part's copy in global memory (written by this workgroup before a barrier): where Raw blocks are copied from.
__device__ __forceinline__ void seg_encode_part(EntropyLds &S, uint8_t *part, const uint8_t *gcopy, const uint32_t M, uint8_t *slot0, uint32_t *csize0, const uint32_t *xh,
                                                unsigned long long *stamps = nullptr)
{
#define SEGE_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
    const uint32_t tt = threadIdx.x, team = tt >> 8, t = tt & 255u, wave = t >> 6, lane = t & 63;
    const bool t0 = team == 0;
    const uint32_t nchunk = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
    const uint32_t m = M;
    // ---- phase 1: histogram of the part; per chunk, whether all its bytes are equal (RLE block)
    uint32_t same_mask = 0;
    if (xh) {
        uint32_t c = 0;
        for (uint32_t k = 0; k < nchunk; k++) {
            const uint32_t mk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK, h = t0 ? xh[256 * k + t] : 0u;
            c += h;
            const unsigned long long full = __ballot(t0 && h == mk);
            __syncthreads();
            if (t0 && lane == 0) S.misc[8 + wave] = full ? 1u : 0u;
            __syncthreads();
            if (S.misc[8] | S.misc[9] | S.misc[10] | S.misc[11]) same_mask |= 1u << k;
        }
        if (t0) S.ctab[t] = c;
        __syncthreads();
    } else {
        // all 512 threads, 16 bytes at a time; every wave first peels off its dominant byte (the first byte it sees): skewed data
        // would serialise the LDS atomics on one bin
        if (t0) { S.ctab[t] = 0; if (t < 4) S.misc[24 + t] = 0; }
        __syncthreads();
        const uint32_t nu = (M + 15) >> 4;
        uint32_t differs = 0; // bit c: some byte of chunk c seen by this thread differs from the chunk's first byte
        for (uint32_t u0 = 0; u0 < nu; u0 += SEGE_NT) {
            const uint32_t u = u0 + tt;
            const bool on = u < nu;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (on) v = *(const uint4 *)(part + 16 * u);
            const uint32_t have = on ? (M - 16 * u < 16 ? M - 16 * u : 16) : 0u;
            const uint32_t ck = (16 * u) >> 14; // (a 16-byte unit never straddles a chunk: chunks are multiples of 16)
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            const uint32_t cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v.x & 0xFF));
            const uint32_t cand4 = cand * 0x01010101u;
            const uint32_t b0 = on ? (uint32_t)part[(size_t)ck * FQZ_CHUNK] * 0x01010101u : 0u;
            uint32_t n_cand = 0, df = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t valid = have >= 4u * d + 4 ? 0x80808080u : (have > 4u * d ? (0x80808080u >> (8 * (4 * d + 4 - have))) : 0u);
                const uint32_t eq = zero_bytes(w4[d] ^ cand4) & valid;
                df |= ~zero_bytes(w4[d] ^ b0) & valid;
                n_cand += __popc(eq);
                uint32_t other = valid & ~eq;
                while (other) {
                    const int bit = __ffs(other) - 1;
                    other &= other - 1;
                    atomicAdd(&S.ctab[(w4[d] >> (bit - 7)) & 0xFF], 1u);
                }
            }
            if (df) differs |= 1u << ck;
            const uint32_t incl = wave_incl_scan(n_cand);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (lane == 0 && tot) atomicAdd(&S.ctab[cand], tot);
        }
        if (differs) atomicOr(&S.misc[24], differs);
        __syncthreads();
        same_mask = ~S.misc[24] & ((1u << nchunk) - 1u);
        __syncthreads();
    }
    SEGE_STAMP(1);
    uint32_t *const keys = lds_keys(S), *const sorted = lds_sorted(S);
    // ---- classify (threads 0..255 from here to the codes)
    if (t0) {
        const uint32_t c = S.ctab[t];
        keys[t] = c ? ((c << 8) | t) : 0u;
        const unsigned long long act = __ballot(c != 0);
        const uint32_t sqi = wave_incl_scan(c * c);
        if (lane == 63) S.misc[16 + wave] = sqi;
        if (lane == 0) S.misc[8 + wave] = (uint32_t)__popcll(act);
    }
    __syncthreads();
    const uint32_t n_active = S.misc[8] + S.misc[9] + S.misc[10] + S.misc[11];
    uint32_t mode = 2;
    {
        const unsigned long long sq = (unsigned long long)S.misc[16] + S.misc[17] + S.misc[18] + S.misc[19];
        if (n_active == 1) mode = 1;
        else if (m < 64) mode = 0;
        else if (sq * 230ull <= (unsigned long long)m * m) mode = 0;
    }
    __syncthreads();
    HufScratch *sc = (HufScratch *)S.out;
    uint32_t tree_size = 0, max_bits = 0;
    SEGE_STAMP(2);
    if (mode == 2) {
        // ---- sort the active symbols by (count, symbol)
        {
            const uint32_t my = t0 ? keys[t] : 0u;
            const unsigned long long bm = __ballot(my != 0);
            if (t0 && lane == 0) S.misc[16 + wave] = (uint32_t)__popcll(bm);
            __syncthreads();
            if (t0) {
                uint32_t base = 0;
                for (uint32_t w2 = 0; w2 < wave; w2++) base += S.misc[16 + w2];
                if (my) sorted[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1))] = my;
            }
            __syncthreads();
            uint32_t mine = (t0 && t < n_active) ? sorted[t] : 0, rank = 0;
            if (t0) for (uint32_t j = 0; j < n_active; j++) rank += sorted[j] < mine;
            __syncthreads();
            if (t0 && t < n_active) sorted[256 - n_active + rank] = mine;
        }
        __syncthreads();
        SEGE_STAMP(3);
        const uint32_t n = n_active;
        const uint32_t *key = sorted + (256 - n);
        // ---- two-queue Huffman merge, leaf preferred on ties (see entropy_encode_group)
        if (n <= 128) {
            if (t0 && wave == 0) {
                const uint32_t INF = 0xFFFFFFFFu;
                const int nn = __builtin_amdgcn_readfirstlane((int)n);
                const uint32_t lw0 = lane < n ? key[lane] >> 8 : INF, lw1 = lane + 64 < n ? key[lane + 64] >> 8 : INF;
                uint32_t iw0 = INF, iw1 = INF;
#define HQ_FETCH(v0, v1, idx, lim) ((idx) < (lim) ? (uint32_t)(((idx) & 64) ? rl((int)(v1), (idx) & 63) : rl((int)(v0), (idx) & 63)) : INF)
                int li = 0, ih = 0, it = 0;
                uint32_t cl = HQ_FETCH(lw0, lw1, 0, nn), ci = INF;
                for (int k = 0; k + 1 < nn; k++) {
                    const bool la = cl <= ci;
                    const int a = la ? li : nn + ih;
                    const uint32_t ca = la ? cl : ci;
                    li += la ? 1 : 0;
                    ih += la ? 0 : 1;
                    cl = HQ_FETCH(lw0, lw1, li, nn);
                    ci = HQ_FETCH(iw0, iw1, ih, it);
                    const bool lb = cl <= ci;
                    const int b = lb ? li : nn + ih;
                    const uint32_t cb2 = lb ? cl : ci;
                    li += lb ? 1 : 0;
                    ih += lb ? 0 : 1;
                    cl = HQ_FETCH(lw0, lw1, li, nn);
                    ci = HQ_FETCH(iw0, iw1, ih, it);
                    const uint32_t sum = ca + cb2;
                    iw0 = ((int)lane == (it & 63) && it < 64) ? sum : iw0;
                    iw1 = ((int)lane == (it & 63) && it >= 64) ? sum : iw1;
                    if (ih == it) ci = sum;
                    if (lane == 0) { sc->parent[a] = (uint16_t)(nn + it); sc->parent[b] = (uint16_t)(nn + it); }
                    it++;
                }
#undef HQ_FETCH
            }
        } else {
            if (t0 && t < n) sc->cnt[t] = key[t] >> 8;
            __syncthreads();
            if (tt == 0) {
                uint32_t li = 0, ih = n, it = n;
                uint32_t cl = sc->cnt[0], ci = 0;
                for (uint32_t k = 0; k + 1 < n; k++) {
                    uint32_t a, b, ca, cb2;
                    if (li < n && (ih >= it || cl <= ci)) { a = li++; ca = cl; cl = li < n ? sc->cnt[li] : 0; }
                    else { a = ih++; ca = ci; ci = ih < it ? sc->cnt[ih] : 0; }
                    if (li < n && (ih >= it || cl <= ci)) { b = li++; cb2 = cl; cl = li < n ? sc->cnt[li] : 0; }
                    else { b = ih++; cb2 = ci; ci = ih < it ? sc->cnt[ih] : 0; }
                    sc->cnt[it] = ca + cb2;
                    if (ih == it) ci = ca + cb2;
                    sc->parent[a] = (uint16_t)it;
                    sc->parent[b] = (uint16_t)it;
                    it++;
                }
            }
        }
        __syncthreads();
        // ---- leaf depths
        if (t0) {
            uint32_t d = 0;
            if (t < n) {
                uint32_t v = t;
                const uint32_t root = 2 * n - 2;
                while (v != root) { v = sc->parent[v]; d++; }
                sc->l[t] = (uint8_t)d;
            }
            const uint32_t any = d; // wave maximum: bit by bit with ballots (depths are small)
            uint32_t mx = 0;
            for (int b = 7; b >= 0; b--) { const unsigned long long hit = __ballot(((any >> b) & 1u) && (any >> (b + 1)) == (mx >> (b + 1))); if (hit) mx |= 1u << b; }
            if (lane == 0) S.misc[16 + wave] = mx;
        }
        __syncthreads();
        uint32_t maxd = max(max(S.misc[16], S.misc[17]), max(S.misc[18], S.misc[19]));
        if (maxd > FQZ_HUF_MAX_BITS) {
            if (tt == 0) {
                int K = 0;
                for (uint32_t i = 0; i < n; i++) {
                    if (sc->l[i] > FQZ_HUF_MAX_BITS) sc->l[i] = FQZ_HUF_MAX_BITS;
                    K += 1 << (FQZ_HUF_MAX_BITS - sc->l[i]);
                }
                while (K > (1 << FQZ_HUF_MAX_BITS)) {
                    int best = -1;
                    for (uint32_t i = 0; i < n; i++)
                        if (sc->l[i] < FQZ_HUF_MAX_BITS && (best < 0 || sc->l[i] > sc->l[best])) best = (int)i;
                    sc->l[best]++;
                    K -= 1 << (FQZ_HUF_MAX_BITS - sc->l[best]);
                }
                int slack = (1 << FQZ_HUF_MAX_BITS) - K;
                while (slack > 0) {
                    for (int i = (int)n - 1; i >= 0 && slack > 0; i--)
                        while (sc->l[i] > 1 && (1 << (FQZ_HUF_MAX_BITS - sc->l[i])) <= slack) {
                            slack -= 1 << (FQZ_HUF_MAX_BITS - sc->l[i]);
                            sc->l[i]--;
                        }
                }
                uint32_t md = 0;
                for (uint32_t i = 0; i < n; i++) md = sc->l[i] > md ? sc->l[i] : md;
                S.misc[20] = md;
            }
            __syncthreads();
            maxd = S.misc[20];
        }
        max_bits = maxd;
        SEGE_STAMP(4);
        // ---- lengths back to symbol order, weights, highest symbol
        if (t0) S.nbits[t] = 0;
        __syncthreads();
        if (t0 && t < n) S.nbits[key[t] & 0xFF] = sc->l[t];
        __syncthreads();
        const uint32_t nbv = t0 ? S.nbits[t] : 0u;
        {
            const unsigned long long bm = __ballot(nbv != 0);
            if (t0 && lane == 0) S.misc[16 + wave] = bm ? wave * 64 + (63 - (uint32_t)__clzll((long long)bm)) : 0;
        }
        __syncthreads();
        const uint32_t nw = max(max(S.misc[16], S.misc[17]), max(S.misc[18], S.misc[19]));
        if (t0) S.w[t] = (t < nw && nbv) ? (uint8_t)(maxd + 1 - nbv) : 0;
        __syncthreads();
        // ---- Huffman_Tree_Description
        if (nw <= 128) {
            if (tt == 0) sc->tree[0] = (uint8_t)(128 + (nw - 1));
            if (t0 && 2 * t < nw) sc->tree[1 + t] = (uint8_t)((S.w[2 * t] << 4) + S.w[2 * t + 1]);
            tree_size = (nw + 1) / 2 + 1;
        } else {
            // (fse_weights_wg is a 256-thread routine: the second half only meets its barriers)
            uint32_t h = 0;
            if (t0) h = fse_weights_wg(S, sc, (int)nw, nullptr);
            else { for (int q = 0; q < FSE_WEIGHTS_BARRIERS; q++) __syncthreads(); }
            if (tt == 0) {
                uint32_t ts = 0;
                if (h > 1 && h < 128) { sc->tree[0] = (uint8_t)h; ts = h + 1; }
                S.misc[6] = ts;
            }
            __syncthreads();
            tree_size = S.misc[6];
            if (!tree_size) mode = 0;
        }
    }
    SEGE_STAMP(5);
    if (mode == 2) {
        // ---- canonical codes (RFC 8878 4.2.1.3)
        const uint32_t nb = t0 ? S.nbits[t] : 0u;
        uint32_t my_rank = 0;
        for (uint32_t len = 1; len <= max_bits; len++) {
            const unsigned long long bm = __ballot(nb == len);
            if (nb == len) my_rank = (uint32_t)__popcll(bm & ((1ull << lane) - 1));
            if (t0 && lane == 0) S.cc[32 + wave * 16 + len] = (uint32_t)__popcll(bm);
        }
        __syncthreads();
        if (tt == 0) {
            uint32_t minv = 0;
            for (uint32_t len = max_bits; len > 0; len--) {
                const uint32_t cnt = S.cc[32 + len] + S.cc[48 + len] + S.cc[64 + len] + S.cc[80 + len];
                S.cc[len] = minv;
                minv = (minv + cnt) >> 1;
            }
        }
        __syncthreads();
        if (t0) {
            uint32_t code = 0;
            if (nb) {
                uint32_t before = 0;
                for (uint32_t w2 = 0; w2 < wave; w2++) before += S.cc[32 + w2 * 16 + nb];
                code = S.cc[nb] + before + my_rank;
            }
            S.ctab[t] = code | (nb << 16);
        }
    }
    // the tree description stays in a register (of both halves: either may code chunk 0)
    const uint8_t tree_byte = (mode == 2 && t < tree_size) ? sc->tree[t] : 0;
    if (tt == 0) S.misc[28] = 0; // the tree has not been sent yet
    __syncthreads();
    SEGE_STAMP(6);
    // ---- phase 2: a zstd block per chunk, two chunks at a time: half `team` takes chunks team, team + 2.
    //      S.misc[32 + 16 * team ..]: that half's scratch (stream bits / stream starts, mode, total, tree offset)
    uint32_t *const hm = S.misc + 32 + 16 * team;
#pragma clang loop unroll(disable)
    for (uint32_t k0 = 0; k0 < nchunk; k0 += 2) {
        const uint32_t k = k0 + team;
        const bool have_chunk = k < nchunk;
        const uint32_t mk = have_chunk ? (M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK) : 0u;
        uint8_t *cpart = part + (size_t)k * FQZ_CHUNK;
        uint8_t *slot = slot0 + (size_t)k * FQZ_SLOT;
        const uint32_t lastblk = (k + 1 == nchunk) ? 1u : 0u;
        const bool rle = have_chunk && (same_mask & (1u << k));
        if (rle && t == 0) { // RLE block: 3-byte header + the byte
            const uint32_t bh = lastblk | (1u << 1) | (mk << 3);
            *(uint32_t *)slot = (bh & 0xFFFFFF) | ((uint32_t)cpart[0] << 24);
            csize0[k] = 4;
        }
        const bool code_it = have_chunk && !rle && mode == 2;
        // the symbols of this lane: wave w of the half encodes stream w, lane l the `per` consecutive symbols [l * per, (l + 1) * per)
        uint32_t sym[16], cnt = 0;
        const uint32_t nstreams = mk >= 256 ? 4u : 1u;
        if (code_it) {
            if (mk == FQZ_CHUNK) {
                const uint4 *p = (const uint4 *)(cpart + 4096 * wave + 64 * lane);
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                sym[0] = a.x; sym[1] = a.y; sym[2] = a.z; sym[3] = a.w; sym[4] = b.x; sym[5] = b.y; sym[6] = b.z; sym[7] = b.w;
                sym[8] = c.x; sym[9] = c.y; sym[10] = c.z; sym[11] = c.w; sym[12] = d.x; sym[13] = d.y; sym[14] = d.z; sym[15] = d.w;
                cnt = 64;
            } else {
                const uint32_t sg = nstreams == 4 ? (mk + 3) / 4 : mk;
                const uint32_t seg_base = wave * sg;
                uint32_t seg_len = 0;
                if (wave < nstreams) seg_len = (wave == nstreams - 1) ? mk - seg_base : sg;
                const uint32_t per = ((seg_len + 63) / 64 + 3) & ~3u;
                uint32_t sym_a = lane * per, sym_b = sym_a + per;
                if (sym_a > seg_len) sym_a = seg_len;
                if (sym_b > seg_len) sym_b = seg_len;
                cnt = sym_b - sym_a;
                const uint8_t *mine = cpart + seg_base + sym_a;
#pragma unroll
                for (int d = 0; d < 16; d++) sym[d] = 4u * d < cnt ? lds_load_u32_unaligned(mine + 4 * d) : 0u;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 16; d++) sym[d] = 0;
        }
        // ---- pass 1: bits per lane (the code table is read-only by now)
        uint32_t my_bits = 0;
        if (code_it) {
#pragma unroll
            for (int d = 0; d < 16; d++) {
                if (4u * d + 4 <= cnt) {
                    my_bits += (S.ctab[sym[d] & 0xFF] >> 16) + (S.ctab[(sym[d] >> 8) & 0xFF] >> 16) + (S.ctab[(sym[d] >> 16) & 0xFF] >> 16) + (S.ctab[sym[d] >> 24] >> 16);
                } else if (4u * d < cnt) {
                    for (uint32_t z = 0; z < cnt - 4u * d; z++) my_bits += S.ctab[(sym[d] >> (8 * z)) & 0xFF] >> 16;
                }
            }
        }
        const uint32_t incl = wave_incl_scan(my_bits);
        const uint32_t tot_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t bit_off = tot_bits - incl;
        if (lane == 0) hm[wave] = tot_bits;
        __syncthreads(); // every lane of both halves has its symbols in registers: the chunks' bytes in LDS are dead from here
        // ---- the chunk's own bytes become the staging area of its block's CONTENT (smaller than the chunk, or the chunk is stored
        //      Raw; the 3-byte block header goes straight to the slot): cleared, then the literals header by one lane
        uint32_t *stage = (uint32_t *)cpart;
        if (code_it) for (uint32_t i = t; i < (mk + 3) / 4; i += 256) stage[i] = 0;
        __syncthreads();
        // tree placement: the first Compressed block of the part carries the tree.  Which chunk that is depends on the chunks
        // before it (RLE chunks and chunks that turn Raw carry none), so the halves settle it in chunk order: chunk k0 (half 0)
        // decides first, then chunk k0 + 1 (half 1) knowing the outcome.
        for (uint32_t turn = 0; turn < 2; turn++) {
            if (team == turn && code_it && t == 0) {
                const uint32_t tree_sent = S.misc[28];
                const uint32_t tsz2 = tree_sent ? 0u : tree_size;
                uint32_t ssz[4] = {0, 0, 0, 0}, total_streams = 0;
                for (uint32_t q = 0; q < nstreams; q++) { ssz[q] = (hm[q] >> 3) + 1; total_streams += ssz[q]; }
                const uint32_t lit_csize = tsz2 + (nstreams == 4 ? 6 : 0) + total_streams;
                const uint32_t lh = mk < 1024 ? 3 : (mk < 16384 ? 4 : 5);
                const uint32_t content = lh + lit_csize + 1;
                if (content >= mk) hm[5] = 0;
                else {
                    hm[5] = 2;
                    uint8_t *o = (uint8_t *)stage;
                    const uint32_t bh = lastblk | (2u << 1) | (content << 3);
                    const uint32_t lt = tree_sent ? 3u : 2u;
                    slot[0] = (uint8_t)bh; slot[1] = (uint8_t)(bh >> 8); slot[2] = (uint8_t)(bh >> 16);
                    if (lh == 3) {
                        const uint32_t v = lt | ((nstreams == 4 ? 1u : 0u) << 2) | (mk << 4) | (lit_csize << 14);
                        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
                    } else if (lh == 4) {
                        const uint32_t v = lt | (2u << 2) | (mk << 4) | (lit_csize << 18);
                        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24);
                    } else {
                        const uint32_t v = lt | (3u << 2) | (mk << 4) | (lit_csize << 22);
                        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24);
                        o[4] = (uint8_t)(lit_csize >> 10);
                    }
                    uint32_t pos = lh + tsz2; // (positions inside the content)
                    if (nstreams == 4) {
                        for (int q = 0; q < 3; q++) { o[pos + 2 * q] = (uint8_t)ssz[q]; o[pos + 2 * q + 1] = (uint8_t)(ssz[q] >> 8); }
                        pos += 6;
                    }
                    for (uint32_t q = 0; q < 4; q++) { hm[8 + q] = pos; pos += ssz[q]; }
                    o[pos] = 0; // Number_of_Sequences = 0
                    hm[6] = pos + 1; // == content
                    hm[7] = lh;
                    hm[4] = tsz2;
                    S.misc[28] = 1; // the tree has been sent
                }
            }
            __syncthreads();
        }
        const uint32_t cmode = code_it ? hm[5] : 0u;
        if (code_it && cmode == 2) {
            uint8_t *o = (uint8_t *)stage;
            if (t < hm[4]) o[hm[7] + t] = tree_byte;
        }
        __syncthreads();
        if (code_it && cmode == 2 && wave < nstreams) {
            // ---- pass 2: symbols last-to-first, LSB-first bit packing (HUF_compress1X order)
            uint32_t P0 = 8 * hm[8 + wave] + bit_off;
            uint32_t word = P0 >> 5;
            uint32_t fill = P0 & 31;
            unsigned long long acc = 0;
#define PUT2(s1, s2) do { const uint32_t e1 = S.ctab[(s1)], e2 = S.ctab[(s2)];                    \
                          acc |= (unsigned long long)(e1 & 0xFFFF) << fill; fill += e1 >> 16;  \
                          acc |= (unsigned long long)(e2 & 0xFFFF) << fill; fill += e2 >> 16;  \
                          const uint32_t adv = fill >> 5;                                      \
                          if (adv) atomicOr(&stage[word], (uint32_t)acc);                      \
                          acc >>= (adv << 5); word += adv; fill &= 31; } while (0)
#define PUT1(s1) do { const uint32_t e1 = S.ctab[(s1)];                                           \
                      acc |= (unsigned long long)(e1 & 0xFFFF) << fill; fill += e1 >> 16;      \
                      const uint32_t adv = fill >> 5;                                          \
                      if (adv) atomicOr(&stage[word], (uint32_t)acc);                          \
                      acc >>= (adv << 5); word += adv; fill &= 31; } while (0)
#pragma unroll
            for (int d = 15; d >= 0; d--) {
                if (4u * d + 4 <= cnt) {
                    PUT2(sym[d] >> 24, (sym[d] >> 16) & 0xFF);
                    PUT2((sym[d] >> 8) & 0xFF, sym[d] & 0xFF);
                } else if (4u * d < cnt) {
                    for (int z = (int)(cnt - 4u * d) - 1; z >= 0; z--) PUT1((sym[d] >> (8 * z)) & 0xFF);
                }
            }
#undef PUT2
#undef PUT1
            if (lane == 0) { acc |= 1ull << fill; fill += 1; }
            if (fill) atomicOr(&stage[word], (uint32_t)acc);
            if (fill > 32) atomicOr(&stage[word + 1], (uint32_t)(acc >> 32));
        }
        __syncthreads();
        if (code_it && cmode == 2) {
            const uint32_t content = hm[6];
            for (uint32_t i = t; i < (content + 3) / 4; i += 256) store_u32_unaligned(slot + 3 + 4 * i, stage[i]); // (behind the block header)
            if (t == 0) csize0[k] = 3 + content;
        } else if (have_chunk && !rle) {
            // Raw block: 3-byte header + the mk bytes, from the part's copy in global memory (the LDS copy may have been cleared for
            // a block that did not pay)
            const uint32_t bh = lastblk | (0u << 1) | (mk << 3);
            if (t == 0) { slot[0] = (uint8_t)bh; slot[1] = (uint8_t)(bh >> 8); slot[2] = (uint8_t)(bh >> 16); csize0[k] = 3 + mk; }
            const uint8_t *csrc = gcopy + (size_t)k * FQZ_CHUNK;
            for (uint32_t off = t * 16; off < mk; off += 256 * 16) {
                if (off + 16 <= mk) store_u128_unaligned(slot + 3 + off, *(const uint4 *)(csrc + off));
                else
                    for (uint32_t q = off; q < mk; q++) slot[3 + q] = csrc[q];
            }
        }
        __syncthreads(); // the halves' scratch is reused by the next pair of chunks
    }
    SEGE_STAMP(9);
#undef SEGE_STAMP
}
