// fqz_rans.h — FQZ-R1: the quality stream of a version-3 container in interleaved rANS blocks ("container v3
// with interleaved rANS, behind a flag"; the reference reserves the version byte, container.go:19-25, and its decoder
// rejects what it does not know, compress.go:571-573).  Format: oracle/fqz_entropy.c "FQZ-R1" and DESIGN.md §4.
//
// A group (up to four 16 KiB chunks, one frame, one frequency table) is coded by ONE wave: sixteen lanes per chunk, a lane
// per rANS state.  Byte i of a chunk belongs to lane (i >> 4) & 15, so a lane reads (encoder) or writes (decoder) whole
// 16-byte units and the sixteen lanes of a chunk cover 256 consecutive bytes a round.  A step is a chain of a dozen
// dependent integer operations per lane (no memory on the chain of the encoder, two LDS reads on the decoder's), 1024 steps
// a chunk: the kernels are latency-bound chains laid across many waves (a 1 GB batch has ~6 500 groups), like the FSE chains
// of the headers model.
#pragma once
#include "fqz_device.h"
#include "fqz_internal.h"

#define RANS_L 65536u

__device__ __forceinline__ void rans_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------
struct RansEncLds {
    uint32_t hist[256];
    uint4 tab[256];      // x: floor(2^32 / f) (2^32 - 1 for f = 1), y: f, z: cum, w: 4096 - f
    uint8_t ser[1 + 3 * 256 + 3]; // the table as it travels: nsym - 1 | symbols | frequencies | pad to an even size
    __attribute__((aligned(16))) uint8_t stage[4][1024]; // a chunk's words on their way out: a ring over the word area, flushed 256 bytes at a time
};

// Writes the blocks of the group's chunks into their slots (slot0 + k * FQZ_SLOT) and their sizes into csize0[k], like
// entropy_encode_group.  One wave.  Decisions (oracle encode_group_rans): a chunk of one repeated byte is an RLE block and
// stays out of the histogram; short or near-flat groups are Raw; the first chunk that is not RLE carries the table, and if
// coding does not shrink it the whole group is Raw; any other chunk that coding does not shrink is a Raw block on its own.
__device__ void rans_encode_group(RansEncLds &S, const uint8_t *src, const uint32_t M, uint8_t *slot0, uint32_t *csize0, const int dbg = 0)
{
    const uint32_t lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const uint32_t nchunk = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
    // ---- histogram of the chunks that will be coded.  Quality deltas are mostly zero: zeros are counted in registers
    //      (SWAR), everything else goes to LDS atomics
#pragma unroll
    for (int p = 0; p < 4; p++) S.hist[lane + 64 * p] = 0;
    __syncthreads();
    uint32_t same_mask = 0, Mc = 0;
#pragma clang loop unroll(disable)
    for (uint32_t k = 0; k < nchunk; k++) {
        const uint32_t mk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK;
        const uint8_t *csrc = src + (size_t)k * FQZ_CHUNK;
        const uint32_t b0 = (uint32_t)csrc[0] * 0x01010101u;
        uint32_t n0 = 0, differs = 0;
        // (four loads in flight a lane: a wave walks its 64 KiB alone, and a load at a time is a memory round trip at a time)
        for (uint32_t off0 = lane * 16; off0 < mk; off0 += 4 * 64 * 16) {
            uint4 vv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t off = off0 + (uint32_t)r * 64u * 16u;
                vv[r] = off < mk ? *(const uint4 *)(csrc + off) : make_uint4(0u, 0u, 0u, 0u); // (16-byte aligned; the arena is padded behind the last stream)
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t off = off0 + (uint32_t)r * 64u * 16u;
                const uint32_t have = off < mk ? (mk - off < 16u ? mk - off : 16u) : 0u;
                const uint32_t w[4] = {vv[r].x, vv[r].y, vv[r].z, vv[r].w};
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t valid = have >= 4u * d + 4 ? 0x80808080u : (have > 4u * d ? (0x80808080u >> (8 * (4 * d + 4 - have))) : 0u);
                    const uint32_t eq = zero_bytes(w[d]) & valid;
                    differs |= ~zero_bytes(w[d] ^ b0) & valid;
                    n0 += __popc(eq);
                    uint32_t other = valid & ~eq; // 0x80 per byte that needs an atomic
                    while (other) {
                        const int bit = __ffs(other) - 1; // 7, 15, 23 or 31
                        other &= other - 1;
                        atomicAdd(&S.hist[(w[d] >> (bit - 7)) & 0xFF], 1u);
                    }
                }
            }
        }
        n0 = wave_sum(n0);
        if (__ballot(differs != 0) == 0ull) { // one repeated byte: RLE block, its counts leave the histogram again
            same_mask |= 1u << k;
            if ((b0 & 0xFF) && lane == 0) atomicSub(&S.hist[b0 & 0xFF], mk);
        } else {
            if (lane == 0 && n0) atomicAdd(&S.hist[0], n0);
            Mc += mk;
        }
    }
    __syncthreads();
    // ---- frequencies: round(count * 4096 / Mc), at least 1, the rounding error goes to the most frequent symbol
    uint32_t c[4], f[4], n_active = 0, sq = 0, best_key = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        c[p] = S.hist[lane + 64 * p];
        n_active += (uint32_t)__popcll(__ballot(c[p] != 0));
        sq += c[p] * c[p]; // (with two or more symbols the sum of squares is below 2^32)
        const uint32_t key = c[p] ? ((c[p] << 8) | (255u - (lane + 64 * p))) : 0u;
        best_key = key > best_key ? key : best_key;
    }
    sq = wave_sum(sq);
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(best_key, d, WAVE); best_key = o > best_key ? o : best_key; }
    bool coded = n_active >= 2 && Mc >= 64 && !((unsigned long long)sq * 230ull <= (unsigned long long)Mc * Mc);
    const uint32_t Md = Mc ? Mc : 1u;
    uint32_t fsum = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        f[p] = c[p] ? (c[p] * 8192u + Md) / (2u * Md) : 0u;
        if (c[p] && !f[p]) f[p] = 1;
        fsum += f[p];
    }
    fsum = wave_sum(fsum);
    {
        const uint32_t best_sym = 255u - (best_key & 0xFF);
        bool bad = false;
#pragma unroll
        for (int p = 0; p < 4; p++)
            if (coded && lane + 64 * p == best_sym) {
                const int fixed = (int)f[p] + 4096 - (int)fsum;
                if (fixed < 1) bad = true;
                else f[p] = (uint32_t)fixed;
            }
        if (__ballot(bad) != 0ull) coded = false;
    }
    if (dbg == 1) coded = false; // (timing experiments: histogram only)
    uint32_t tlen = 0;
    if (coded) {
        uint32_t cum_carry = 0, n_carry = 0;
        if (lane == 0) S.ser[0] = (uint8_t)(n_active - 1);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t incl = wave_incl_scan(f[p]);
            const uint32_t cum = cum_carry + incl - f[p];
            cum_carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const unsigned long long bm = __ballot(f[p] != 0);
            const uint32_t rank = n_carry + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
            n_carry += (uint32_t)__popcll(bm);
            uint4 e = make_uint4(0u, 0u, 0u, 0u);
            if (f[p]) {
                e.x = f[p] == 1 ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / f[p]);
                e.y = f[p]; e.z = cum; e.w = 4096u - f[p];
                S.ser[1 + rank] = (uint8_t)(lane + 64 * p);
                S.ser[1 + n_active + 2 * rank] = (uint8_t)f[p];
                S.ser[2 + n_active + 2 * rank] = (uint8_t)(f[p] >> 8);
            }
            S.tab[lane + 64 * p] = e;
        }
        tlen = 1 + 3 * n_active;
        if (!(n_active & 1)) { if (lane == 0) S.ser[tlen] = 0; tlen++; }
    }
    __syncthreads();
    // ---- the chunks, side by side: quad q codes chunk q, lane j of the quad is coder j.  Steps from the last to the first;
    //      within a step the coders that renormalise append their words in descending lane order
    const uint32_t carrier = (uint32_t)__ffs(~same_mask & ((1u << nchunk) - 1u)) - 1u; // first chunk that is not RLE (0xFFFFFFFF: none)
    const uint32_t mk = q < nchunk ? (M - q * FQZ_CHUNK < FQZ_CHUNK ? M - q * FQZ_CHUNK : FQZ_CHUNK) : 0u;
    const bool mine = coded && q < nchunk && !((same_mask >> q) & 1u);
    const uint8_t *csrc = src + (size_t)q * FQZ_CHUNK;
    uint8_t *slot = slot0 + (size_t)q * FQZ_SLOT;
    const uint32_t tl = (mine && q == carrier) ? tlen : 0u;
    const uint32_t woff = 6u + tl; // block header 3, m 2, tflag 1, table
    uint32_t x = RANS_L, wc = 0, flushed = 0;
    uint8_t *stg = S.stage[q];
    if (coded && dbg != 2) { // (wave-uniform; dbg 2: tables, no coding)
        const uint32_t U = (M < FQZ_CHUNK ? M + 255u : FQZ_CHUNK + 255u) / 256u; // chunk 0 is the longest: every quad runs its rounds, idle beyond its own
        auto fetch = [&](int u) -> uint4 {
            const uint32_t off = (uint32_t)u * 256u + j * 16u;
            return (u >= 0 && mine && off < mk) ? *(const uint4 *)(csrc + off) : make_uint4(0u, 0u, 0u, 0u);
        };
        uint4 v0 = fetch((int)U - 1), v1 = fetch((int)U - 2), v2 = fetch((int)U - 3);
#pragma clang loop unroll(disable)
        for (int u = (int)U - 1; u >= 0; u--) {
            const uint4 v3 = fetch(u - 3); // three units ahead: the loads are off the chain
            const uint32_t off = (uint32_t)u * 256u + j * 16u;
            const uint32_t have = (mine && off < mk) ? (mk - off < 16u ? mk - off : 16u) : 0u;
            const uint32_t w[4] = {v0.x, v0.y, v0.z, v0.w};
            // the common case - four chunks, each lane a whole unit - runs without the per-lane conditions.  The words go to a ring
            // in LDS and leave 256 bytes at a time: a 2-byte store to global memory at every step made the kernel wait for the
            // vector-memory pipeline (6.7 M store instructions a batch: 0.8 ms)
            const bool fast = __ballot(have == 16u) == ~0ull;
            if (fast) {
#pragma unroll
                for (int b = 15; b >= 0; b--) {
                    const uint4 e = S.tab[(w[b >> 2] >> (8 * (b & 3))) & 0xFFu];
                    const bool need = x >= (e.y << 20);
                    const uint32_t seg = (uint32_t)(__ballot(need) >> (16 * q)) & 0xFFFFu;
                    if (need) *(uint16_t *)(stg + ((2u * (wc + (uint32_t)__popc(seg >> (j + 1)))) & 1023u)) = (uint16_t)x;
                    x = need ? x >> 16 : x;
                    wc += (uint32_t)__popc(seg);
                    uint32_t q0 = __umulhi(x, e.x);
                    q0 += (x - __umul24(q0, e.y)) >= e.y ? 1u : 0u; // the reciprocal rounds down: the quotient is short by at most one
                    x = __umul24(q0, e.w) + (x + e.z);              // (q << 12) + (x - q f) + cum
                }
            } else {
#pragma unroll
                for (int b = 15; b >= 0; b--) {
                    const uint4 e = S.tab[(w[b >> 2] >> (8 * (b & 3))) & 0xFFu];
                    const bool act = (uint32_t)b < have;
                    const bool need = act && x >= (e.y << 20);
                    const uint32_t seg = (uint32_t)(__ballot(need) >> (16 * q)) & 0xFFFFu;
                    if (need) {
                        *(uint16_t *)(stg + ((2u * (wc + (uint32_t)__popc(seg >> (j + 1)))) & 1023u)) = (uint16_t)x;
                        x >>= 16;
                    }
                    wc += (uint32_t)__popc(seg);
                    if (act) {
                        uint32_t q0 = __umulhi(x, e.x);
                        q0 += (x - __umul24(q0, e.y)) >= e.y ? 1u : 0u;
                        x = __umul24(q0, e.w) + (x + e.z);
                    }
                }
            }
            rans_lds_order();
            for (;;) { // (a unit adds at most 512 bytes to less than 256 pending ones)
                const bool go = mine && 2u * wc - flushed >= 256u;
                if (__ballot(go) == 0ull) break;
                if (go) {
                    const uint32_t o = woff + flushed + 16u * j;
                    if (o + 16u <= FQZ_SLOT) store_u128_unaligned(slot + o, *(const uint4 *)(stg + ((flushed + 16u * j) & 1023u))); // (a block that would run past this is not kept)
                    flushed += 256u;
                }
            }
            v0 = v1; v1 = v2; v2 = v3;
        }
    }
    // ---- verdicts
    const uint32_t content = 3u + tl + 2u * wc + 64u;
    const bool fits = mine && content < mk;
    const unsigned long long fitm = __ballot(fits);
    const bool carrier_ok = coded && carrier < nchunk && ((fitm >> (16 * carrier)) & 1ull);
    const bool keep = fits && carrier_ok;
    if (keep) {
        const uint32_t pend = 2u * wc - flushed, n16 = pend >> 4, rest = (pend & 15u) >> 1; // what is still in the ring (< 256 bytes)
        if (j < n16) store_u128_unaligned(slot + woff + flushed + 16u * j, *(const uint4 *)(stg + ((flushed + 16u * j) & 1023u)));
        if (j < rest) *(uint16_t *)(slot + woff + flushed + 16u * n16 + 2u * j) = *(const uint16_t *)(stg + ((flushed + 16u * n16 + 2u * j) & 1023u));
        store_u32_unaligned(slot + woff + 2u * wc + 4u * j, x);
        if (j == 0) {
            const uint32_t bh = ((q + 1 == nchunk) ? 1u : 0u) | (3u << 1) | (content << 3);
            slot[0] = (uint8_t)bh; slot[1] = (uint8_t)(bh >> 8); slot[2] = (uint8_t)(bh >> 16);
            slot[3] = (uint8_t)mk; slot[4] = (uint8_t)(mk >> 8); slot[5] = tl ? 1 : 0;
            csize0[q] = 3u + content;
        }
        for (uint32_t i = j; i < tl; i += 16) slot[6 + i] = S.ser[i];
    }
    const unsigned long long keepm = __ballot(keep);
    if (dbg) return; // (timing experiments: garbage out)
    {
        unsigned long long all = 0;
        for (uint32_t k = 0; k < nchunk; k++) all |= 1ull << (16 * k);
        if ((keepm & all) == all) return; // every chunk is an rANS block: done
    }
    // the words of a chunk that is stored raw after all are overwritten below by other lanes of this wave: its stores have to
    // be through first (workgroup scope: a wait, not the L2 write-back of a device-scope fence - that cost 0.2 ms a batch)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma clang loop unroll(disable)
    for (uint32_t k = 0; k < nchunk; k++) {
        if ((keepm >> (16 * k)) & 1ull) continue;
        const uint32_t mkk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK;
        const uint8_t *ks = src + (size_t)k * FQZ_CHUNK;
        uint8_t *kslot = slot0 + (size_t)k * FQZ_SLOT;
        const uint32_t lastblk = k + 1 == nchunk ? 1u : 0u;
        if ((same_mask >> k) & 1u) { // RLE block: 3-byte header + the byte
            if (lane == 0) {
                const uint32_t bh = lastblk | (1u << 1) | (mkk << 3);
                *(uint32_t *)kslot = (bh & 0xFFFFFFu) | ((uint32_t)ks[0] << 24);
                csize0[k] = 4;
            }
            continue;
        }
        const uint32_t bh = lastblk | (0u << 1) | (mkk << 3); // Raw block
        if (lane == 0) { kslot[0] = (uint8_t)bh; kslot[1] = (uint8_t)(bh >> 8); kslot[2] = (uint8_t)(bh >> 16); csize0[k] = 3 + mkk; }
        for (uint32_t off = lane * 16; off < mkk; off += 64 * 16) {
            if (off + 16 <= mkk) store_u128_unaligned(kslot + 3 + off, *(const uint4 *)(ks + off));
            else
                for (uint32_t i = off; i < mkk; i++) kslot[3 + i] = ks[i];
        }
    }
}

#ifdef FQZ_RANS_DECODER // (fqz_decode.hip: needs its DecChunk)
// ---------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------
#define RANS_WIN 1024u // ring of a chunk's word area in LDS (the coders of a chunk take at most 256 bytes in 8 steps)
struct RansDecLds {
    uint8_t slot[4096];   // 12-bit slot -> symbol
    uint32_t ft[256];     // symbol -> f | cum << 16
    uint8_t esym[256];    // table entry -> symbol
    uint8_t win[4][RANS_WIN];
};

// One wave per group that has rANS blocks.  gd.x: id of the group's first chunk, gd.y: chunks in the group | index of the
// chunk that carries the table << 8 (k_dec_index found them and checked the block headers against the index).
__device__ void rans_decode_group(RansDecLds &S, const uint8_t *in, const DecChunk *chunks, const uint2 gd, uint8_t *arena, bool *failed)
{
    const uint32_t lane = threadIdx.x & 63u, q = lane >> 4, j = lane & 15u;
    const uint32_t nchunk = gd.y & 0xFFu, carrier = (gd.y >> 8) & 0xFFu;
    bool bad = nchunk < 1 || nchunk > 4 || carrier >= nchunk;
    // ---- the table (content of the carrier: m u16 | tflag | nsym - 1 | symbols | frequencies | pad)
    const DecChunk cc = chunks[gd.x + (bad ? 0u : carrier)];
    const uint8_t *tp = in + cc.src_off + 3;
    uint32_t nsym = 0, tlen = 0;
    if (cc.csize < 3 + 64 + 8) bad = true;
    else {
        nsym = (uint32_t)tp[0] + 1u;
        tlen = 1 + 3 * nsym + ((nsym & 1) ? 0u : 1u);
        if (nsym < 2 || 3 + tlen + 64 > cc.csize) bad = true;
    }
    if (bad) { *failed = true; return; }
#pragma unroll
    for (int p = 0; p < 4; p++) { S.ft[lane + 64 * p] = 0; ((uint4 *)S.slot)[lane + 64 * p] = make_uint4(0u, 0u, 0u, 0u); }
    __syncthreads();
    {
        uint32_t cum_carry = 0, prev_sym_carry = 0xFFFFFFFFu;
        for (uint32_t e0 = 0; e0 < nsym; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool on = e < nsym;
            const uint32_t sym = on ? tp[1 + e] : 0u;
            const uint32_t fr = on ? (uint32_t)tp[1 + nsym + 2 * e] | ((uint32_t)tp[2 + nsym + 2 * e] << 8) : 0u;
            uint32_t prev = (uint32_t)__shfl_up((int)sym, 1, WAVE);
            if (lane == 0) prev = prev_sym_carry;
            if (on && (fr < 1 || fr > 4095 || (prev != 0xFFFFFFFFu && sym <= prev))) bad = true;
            const uint32_t incl = wave_incl_scan(fr);
            const uint32_t cum = cum_carry + incl - fr;
            if (on && cum + fr <= 4096) {
                S.ft[sym] = fr | (cum << 16);
                S.esym[e] = (uint8_t)sym;
                S.slot[cum] = (uint8_t)e; // start marker: the entry index (entry 0 starts at slot 0, where the marker is the fill value)
            } else if (on) bad = true;
            cum_carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            prev_sym_carry = (uint32_t)__builtin_amdgcn_readlane((int)sym, __builtin_amdgcn_readfirstlane((int)((nsym - e0 < 64 ? nsym - e0 : 64) - 1)));
        }
        if (cum_carry != 4096) bad = true;
        if (!(nsym & 1) && tp[tlen - 1] != 0) bad = true;
    }
    if (__ballot(bad) != 0ull) { *failed = true; return; }
    __syncthreads();
    { // slot -> symbol: a running maximum over the start markers (entries start at ascending slots), 64 slots a lane
        uint32_t m[16], run = 0;
#pragma unroll
        for (int d = 0; d < 16; d++) {
            const uint32_t w = ((const uint32_t *)S.slot)[lane * 16 + d];
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { const uint32_t v = (w >> (8 * b)) & 0xFF; run = v > run ? v : run; o |= run << (8 * b); }
            m[d] = o;
        }
        uint32_t incl = run; // inclusive maximum over the lanes below
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, WAVE); if (lane >= (uint32_t)d) incl = o > incl ? o : incl; }
        uint32_t before = (uint32_t)__shfl_up((int)incl, 1, WAVE);
        if (lane == 0) before = 0;
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 16; d++) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { uint32_t v = (m[d] >> (8 * b)) & 0xFF; v = v > before ? v : before; o |= (uint32_t)S.esym[v] << (8 * b); }
            ((uint32_t *)S.slot)[lane * 16 + d] = o;
        }
    }
    __syncthreads();
    // ---- the chunks: quad q decodes chunk q
    DecChunk c;
    c.btype = 0;
    if (q < nchunk) c = chunks[gd.x + q];
    const bool mine = q < nchunk && c.btype == 4;
    const uint8_t *src = in + c.src_off;
    uint32_t m = 0, p = 0, states = 0;
    if (mine) {
        m = rd16(src);
        const uint32_t tflag = src[2];
        p = 3 + ((tflag & 1) ? tlen : 0u);
        if (m != c.regen || m < 1 || (tflag & ~1u) || ((tflag & 1) != (q == carrier ? 1u : 0u)) || c.csize < p + 64 || ((c.csize - 64 - p) & 1)) bad = true;
        states = c.csize - 64;
    }
    uint32_t x = RANS_L;
    if (mine && !bad) {
        x = load_u32_unaligned(src + states + 4 * j);
        if (x < RANS_L) bad = true;
    }
    if (__ballot(bad) != 0ull) { *failed = true; return; }
    // Offsets into the input, shifted by the parity of the word area's start so that every word sits at an even offset (one
    // 16-bit LDS read a word): the word area is [wbase, wp0) of inq, the ring holds [lo, wp) at least
    const uint32_t par = mine ? (c.src_off + p) & 1u : 0u;
    const uint8_t *inq = in + par;
    const uint32_t wbase = mine ? c.src_off + p - par : 0u;
    uint32_t wp = mine ? c.src_off + states - par : 0u, lo;
    uint8_t *win = S.win[q];
    {
        const uint32_t target = wp > 512u ? (wp - 512u) & ~15u : 0u;
        const uint32_t top = (wp + 15u) & ~15u;
        if (mine)
            for (uint32_t a = target + 16 * j; a < top; a += 256) *(uint4 *)(win + (a & (RANS_WIN - 1))) = load_u128_unaligned(inq + a);
        lo = target;
    }
    rans_lds_order();
    uint32_t U = 0;
    {
        uint32_t mm = mine ? m : 0u;
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(mm, d, WAVE); mm = o > mm ? o : mm; }
        U = (mm + 255u) / 256u;
    }
    uint8_t *dst = arena + c.dst_off;
    const uint32_t below = (1u << j) - 1u; // the coders of my chunk that take their words before me
#pragma clang loop unroll(disable)
    for (uint32_t u = 0; u < U; u++) {
        const uint32_t off = u * 256u + j * 16u;
        const uint32_t have = (mine && off < m) ? (m - off < 16u ? m - off : 16u) : 0u;
        const bool fast = __ballot(have == 16u) == ~0ull; // four chunks, every lane a whole unit: no per-lane conditions
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // refill: what the next eight steps may consume beyond the ring's low mark, loaded now, written behind the steps
            const uint32_t target = wp > 512u ? (wp - 512u) & ~15u : 0u;
            uint4 fill = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t fa = lo - 16u * (j + 1);
            const bool do_fill = mine && lo >= 16u * (j + 1) && fa >= target;
            if (do_fill) fill = load_u128_unaligned(inq + fa);
            if (fast) {
#pragma unroll
                for (int b8 = 0; b8 < 8; b8++) {
                    const int b = h * 8 + b8;
                    const uint32_t sl = x & 4095u;
                    const uint32_t s = S.slot[sl];
                    const uint32_t e = S.ft[s];
                    x = __umul24(e & 0xFFFFu, x >> 12) + (sl - (e >> 16));
                    w[b >> 2] |= s << (8 * (b & 3));
                    const bool need = x < RANS_L;
                    const uint32_t seg = (uint32_t)(__ballot(need) >> (16 * q)) & 0xFFFFu;
                    if (need) x = (x << 16) | *(const uint16_t *)(win + ((wp - 2u * ((uint32_t)__popc(seg & below) + 1u)) & (RANS_WIN - 1)));
                    wp -= 2u * (uint32_t)__popc(seg);
                }
            } else {
#pragma unroll
                for (int b8 = 0; b8 < 8; b8++) {
                    const int b = h * 8 + b8;
                    const uint32_t sl = x & 4095u;
                    const uint32_t s = S.slot[sl];
                    const uint32_t e = S.ft[s];
                    const bool act = (uint32_t)b < have;
                    const uint32_t xn = __umul24(e & 0xFFFFu, x >> 12) + (sl - (e >> 16));
                    const bool need = act && xn < RANS_L;
                    const uint32_t seg = (uint32_t)(__ballot(need) >> (16 * q)) & 0xFFFFu;
                    if (act) { x = xn; w[b >> 2] |= s << (8 * (b & 3)); }
                    if (need) x = (x << 16) | *(const uint16_t *)(win + ((wp - 2u * ((uint32_t)__popc(seg & below) + 1u)) & (RANS_WIN - 1)));
                    wp -= 2u * (uint32_t)__popc(seg);
                }
            }
            if (mine && wp < wbase) { bad = true; wp = wbase; } // more words taken than the block holds: garbage, but in bounds
            if (do_fill) *(uint4 *)(win + (fa & (RANS_WIN - 1))) = fill;
            if (mine && lo > target) lo = target;
            rans_lds_order();
        }
        if (have == 16) store_u128_unaligned(dst + off, make_uint4(w[0], w[1], w[2], w[3]));
        else
            for (uint32_t i = 0; i < have; i++) dst[off + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
    if (mine && (wp != wbase || x != RANS_L)) bad = true;
    if (__ballot(bad) != 0ull) *failed = true;
}
#endif
