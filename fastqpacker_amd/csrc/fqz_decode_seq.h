// fqz_decode_seq.h — fast path for the zstd blocks of our headers streams that carry sequences (FQZ-H2, fqz_hdrlz.h).
//
// Such a block is self-contained by construction: all three symbol tables are the PREDEFINED ones (modes byte 0), the
// first sequence states its offset explicitly, the only repeat code used is "the offset of the previous sequence of this
// block" and no match reaches in front of the block.  So the blocks of a payload are decoded independently of each other:
//   k_dec_huf / k_dec_entropy   the literals section, into a scratch area (as for any other block)
//   k_dec_seq_fse               the sequence bit stream: a LANE per block (the three FSE state chains are serial; 64 blocks
//                               per wave hide each other's table lookups) -> (literal length, match length, offset) triples
//   k_dec_seq_exec              a WAVE per block: literal runs are placed with a wave scan, the matches (each may copy what
//                               its predecessor produced: header i copies from header i - 1) run in order inside LDS
// Anything outside that profile (another mode, another repeat code, an offset past the block start, more than
// DSEQ_MAX sequences) sends the batch to the general path (k_dec_lz), which implements all of RFC 8878 3.1.1.3.2.
// Included by fqz_decode.hip only.
#pragma once

// (DSEQ_MAX, DSEQ_STRIDE, DSEQ_INVALID: fqz_decode.hip, next to DecChunk)

struct SeqDt { uint32_t ll[64], ml[64], of[32]; };       // sym | nb << 8 | base << 16
template <int NSYM>
constexpr void seq_make_dt(const short (&norm)[NSYM], int log, uint32_t *dt)
{
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint8_t sym[64] = {};
    uint16_t next[64] = {};
    int high = size - 1;
    for (int s = 0; s < NSYM; s++) {
        if (norm[s] == -1) { sym[high--] = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    int pos = 0;
    for (int s = 0; s < NSYM; s++)
        for (int i = 0; i < norm[s]; i++) {
            sym[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    for (int u = 0; u < size; u++) {
        const int s = sym[u];
        const uint32_t ns = next[s]++;
        int hb = 0;
        for (uint32_t v = ns; v >>= 1;) hb++;
        const uint32_t nb = (uint32_t)(log - hb);
        dt[u] = (uint32_t)s | (nb << 8) | ((((ns << nb) - (uint32_t)size) & 0xFFFFu) << 16);
    }
}
constexpr short DSEQ_LL_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
constexpr short DSEQ_ML_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
constexpr short DSEQ_OF_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
constexpr SeqDt seq_make_all()
{
    SeqDt d{};
    seq_make_dt(DSEQ_LL_NORM, 6, d.ll);
    seq_make_dt(DSEQ_ML_NORM, 6, d.ml);
    seq_make_dt(DSEQ_OF_NORM, 5, d.of);
    return d;
}
__constant__ const SeqDt c_seq_dt = seq_make_all();

struct SeqFseLds {
    SeqDt dt;
    uint32_t ll_base[36], ml_base[53];
    uint8_t ll_bits[36], ml_bits[56];
};

__device__ __forceinline__ uint32_t bbp_take(BackBitsP &b, uint32_t nb) // nb <= 32 bits from the top; call bbp_refill first
{
    const uint32_t v = nb ? (uint32_t)(b.buf >> (64 - nb)) : 0u;
    b.buf = nb ? b.buf << nb : b.buf;
    b.avail -= (int)nb;
    return v;
}

__global__ __launch_bounds__(64) void k_dec_seq_fse(const uint8_t *in, DecInfo *info, const DecChunk *chunks, uint8_t *arena)
{
    __shared__ SeqFseLds T;
    const uint32_t lane = threadIdx.x, id = blockIdx.x * 64 + lane;
    if (info->status) return;
    DecChunk c;
    c.seq_len = 0;
    if (id < info->n_chunks) c = chunks[id];
    if (!__ballot(c.seq_len != 0)) return;
    {
        const uint32_t *src = (const uint32_t *)&c_seq_dt;
        uint32_t *dst = (uint32_t *)&T.dt;
        for (uint32_t i = lane; i < sizeof(SeqDt) / 4; i += 64) dst[i] = src[i];
        if (lane < 36) { T.ll_base[lane] = c_ll_base[lane]; T.ll_bits[lane] = c_ll_bits[lane]; }
        if (lane < 53) { T.ml_base[lane] = c_ml_base[lane]; T.ml_bits[lane] = c_ml_bits[lane]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (!c.seq_len) return;
    uint8_t *scr = arena + c.dst_off;
    uint2 *out = (uint2 *)(scr + FQZ_CHUNK);
    uint32_t *hdr = (uint32_t *)(scr + 2 * FQZ_CHUNK);
    const uint8_t *sq = in + c.src_off + c.csize - c.seq_len;
    const uint32_t sn = c.seq_len;
    int verdict = 0; // 0 fine, 1 = not our profile (general path), 2 = corrupt
    uint32_t nseq = sq[0], shdr = 1;
    if (nseq >= 128) {
        if (nseq == 255 || sn < 2) verdict = 1;
        else { nseq = ((nseq - 128) << 8) + sq[1]; shdr = 2; }
    }
    if (!verdict && (nseq == 0 || nseq > DSEQ_MAX || sn < shdr + 2 || sq[shdr] != 0)) verdict = 1;
    BackBitsP br;
    int bits_left = 0;
    if (!verdict) {
        bits_left = bbp_init(br, sq + shdr + 1, sn - shdr - 1);
        if (bits_left < 0) verdict = 2;
    }
    if (!verdict) {
        bbp_refill(br);
        uint32_t st_ll = bbp_take(br, 6), st_of = bbp_take(br, 5), st_ml = bbp_take(br, 6);
        bits_left -= 17;
        uint32_t o = 0, lit_used = 0, prev_off = 0;
        for (uint32_t i = 0; i < nseq; i++) {
            const uint32_t e_ll = T.dt.ll[st_ll], e_of = T.dt.of[st_of], e_ml = T.dt.ml[st_ml];
            const uint32_t oc = e_of & 0xFF, mc = e_ml & 0xFF, lc = e_ll & 0xFF;
            if (oc > 24) { verdict = 1; break; }
            bbp_refill(br);
            const uint32_t of_val = (1u << oc) + bbp_take(br, oc);
            bbp_refill(br);
            const uint32_t mb = T.ml_bits[mc], lb = T.ll_bits[lc];
            const uint32_t ml = T.ml_base[mc] + bbp_take(br, mb);
            const uint32_t ll = T.ll_base[lc] + bbp_take(br, lb);
            bits_left -= (int)(oc + mb + lb);
            if (i + 1 < nseq) {
                bbp_refill(br);
                const uint32_t nl = (e_ll >> 8) & 0xFF, nm = (e_ml >> 8) & 0xFF, no = (e_of >> 8) & 0xFF;
                st_ll = (e_ll >> 16) + bbp_take(br, nl);
                st_ml = (e_ml >> 16) + bbp_take(br, nm);
                st_of = (e_of >> 16) + bbp_take(br, no);
                bits_left -= (int)(nl + nm + no);
            }
            if (bits_left < 0) { verdict = 2; break; }
            uint32_t offset;
            if (of_val > 3) offset = of_val - 3;
            else if (of_val == 1 && ll > 0 && i > 0) offset = prev_off; // "the offset of the previous sequence" (of this block)
            else { verdict = 1; break; }
            prev_off = offset;
            if (ll > c.regen - lit_used || ll > c.out_len - o || ml > c.out_len - o - ll) { verdict = 2; break; }
            if (offset > o + ll) { verdict = 1; break; } // reaches in front of the block: not ours
            out[i] = make_uint2(ll | (ml << 16), offset);
            lit_used += ll;
            o += ll + ml;
        }
        if (!verdict && (bits_left != 0 || o + (c.regen - lit_used) != c.out_len)) verdict = 2;
    }
    *hdr = verdict ? DSEQ_INVALID : nseq;
    // (a corrupt block is left to the general path as well: it gives the authoritative verdict)
    if (verdict) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
}

struct SeqExecLds {
    uint8_t out[FQZ_CHUNK + 16];
    uint8_t lit[FQZ_CHUNK + 16];
};

__global__ __launch_bounds__(64) void k_dec_seq_exec(DecInfo *info, const DecChunk *chunks, uint8_t *arena)
{
    __shared__ __attribute__((aligned(16))) SeqExecLds S;
    const uint32_t lane = threadIdx.x, id = blockIdx.x;
    if (id >= info->n_chunks || info->status) return;
    const DecChunk c = chunks[id];
    if (!c.seq_len) return;
    const uint8_t *scr = arena + c.dst_off;
    const uint2 *seqs = (const uint2 *)(scr + FQZ_CHUNK);
    const uint32_t nseq = *(const uint32_t *)(scr + 2 * FQZ_CHUNK);
    if (nseq == DSEQ_INVALID || nseq > DSEQ_MAX) return;
    const uint32_t n_lit = c.regen, n_out = c.out_len;
    for (uint32_t i = lane * 16; i < n_lit; i += 64 * 16) *(uint4 *)&S.lit[i] = *(const uint4 *)(scr + i); // (whole uint4: the scratch is padded)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint32_t o = 0, lp = 0;
    for (uint32_t base = 0; base < nseq; base += 64) {
        const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
        uint2 q = make_uint2(0, 0);
        if (lane < cnt) q = seqs[base + lane];
        const uint32_t ll = q.x & 0xFFFFu, ml = q.x >> 16;
        const uint32_t in_o = wave_incl_scan(ll + ml), in_l = wave_incl_scan(ll);
        const uint32_t op = o + in_o - (ll + ml), lq = lp + in_l - ll; // this sequence's literals go to out[op, op + ll), its match behind them
        for (uint32_t k = 0; k < ll; k++) S.out[op + k] = S.lit[lq + k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t mdst = op + ll;
        for (uint32_t j = 0; j < cnt; j++) { // matches in order: a match may read what the one before it wrote
            const uint32_t mlj = (uint32_t)__builtin_amdgcn_readlane((int)ml, (int)j), dj = (uint32_t)__builtin_amdgcn_readlane((int)mdst, (int)j);
            const uint32_t fj = (uint32_t)__builtin_amdgcn_readlane((int)q.y, (int)j);
            const uint8_t *ms = S.out + dj - fj;
            if (fj >= mlj) { for (uint32_t k = lane; k < mlj; k += 64) S.out[dj + k] = ms[k]; }
            else { for (uint32_t k = lane; k < mlj; k += 64) S.out[dj + k] = ms[k % fj]; } // overlapping: a pattern fill
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        o += (uint32_t)__builtin_amdgcn_readlane((int)in_o, 63);
        lp += (uint32_t)__builtin_amdgcn_readlane((int)in_l, 63);
    }
    for (uint32_t k = lane; k < n_lit - lp; k += 64) S.out[o + k] = S.lit[lp + k]; // the literals behind the last match
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint8_t *dst = arena + c.out_off; // 16-byte aligned (a chunk of a 16-aligned stream)
    const uint32_t full = n_out & ~15u;
    for (uint32_t i = lane * 16; i < full; i += 64 * 16) *(uint4 *)(dst + i) = *(const uint4 *)&S.out[i];
    for (uint32_t i = full + lane; i < n_out; i += 64) dst[i] = S.out[i];
}
