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

// Per decoding state, packed for one 64-bit LDS read: x = symbol | nbits << 8 | next-state base << 16 (c_seq_dt),
// y = base of the value the symbol stands for | its extra bits << 24 (offsets: y = extra bits, the value is 1 << bits + extra)
#define SEQ_WIN 128u          // bytes of a lane's bit stream staged in LDS at a time
#define SEQ_WIN_STRIDE 33u    // dwords between the windows of neighbouring lanes (odd: no systematic bank conflicts)
struct SeqFseLds {
    uint2 ll[64], ml[64], of[32];
    uint32_t win[64 * SEQ_WIN_STRIDE + 4];
};

// backward bit reader over a lane's window: the stream bytes [wbase, wbase + SEQ_WIN) sit in LDS; bytes are consumed from
// the end of the stream towards its start, four at a time
struct SeqBits {
    unsigned long long buf; // next bits at the MSB end
    int avail;              // valid bits in buf
    int byte_pos;           // stream bytes [0, byte_pos) not merged yet
    int wbase;              // stream offset of the window's first byte
};
__device__ __forceinline__ void seq_window_load(uint32_t *win, const uint8_t *p, int byte_pos, int *wbase)
{
    int wb = byte_pos - (int)SEQ_WIN;
    wb = wb < 0 ? 0 : wb;
    *wbase = wb;
    // (a piece may read up to 15 bytes behind byte_pos: the rest of the stream, or - behind the section - the frame's checksum
    //  and the payloads that follow the headers payload in every block)
#pragma unroll
    for (uint32_t q = 0; q < SEQ_WIN / 16; q++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (wb + (int)(16 * q) < byte_pos) v = load_u128_unaligned(p + wb + 16 * q);
        win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ void seq_refill(SeqBits &b, const uint32_t *win)
{
    if (b.avail <= 32 && b.byte_pos > 0) {
        const int take = b.byte_pos < 4 ? b.byte_pos : 4;
        const int at = b.byte_pos - 4 - b.wbase;                 // window offset of the dword that ends at byte_pos (>= -3)
        const int a4 = at >> 2;                                   // (arithmetic shift: -1 for at < 0)
        const uint32_t lo = a4 >= 0 ? win[a4] : 0u, hi = win[a4 + 1];
        const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)at & 3u);
        const uint32_t keep = take == 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (8 * take)); // a partial first word: its low bytes precede the stream
        b.buf |= (unsigned long long)(w & keep) << (32 - b.avail);
        b.avail += 8 * take;
        b.byte_pos -= take;
    }
}
__device__ __forceinline__ uint32_t seq_take(SeqBits &b, uint32_t nb)
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
        const uint32_t el = c_seq_dt.ll[lane], em = c_seq_dt.ml[lane];
        T.ll[lane] = make_uint2(el, c_ll_base[el & 0xFF] | ((uint32_t)c_ll_bits[el & 0xFF] << 24));
        T.ml[lane] = make_uint2(em, c_ml_base[em & 0xFF] | ((uint32_t)c_ml_bits[em & 0xFF] << 24));
        if (lane < 32) { const uint32_t eo = c_seq_dt.of[lane]; T.of[lane] = make_uint2(eo, eo & 0xFF); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const bool on = c.seq_len != 0;
    uint8_t *scr = arena + (on ? c.dst_off : 0u);
    uint2 *out = (uint2 *)(scr + FQZ_CHUNK);
    const uint8_t *sq = in + (on ? c.src_off + c.csize - c.seq_len : 0u);
    const uint32_t sn = c.seq_len;
    int verdict = 0; // 0 fine, 1 = not our profile (general path), 2 = corrupt
    uint32_t nseq = 0, shdr = 1;
    if (on) {
        nseq = sq[0];
        if (nseq >= 128) {
            if (nseq == 255 || sn < 2) verdict = 1;
            else { nseq = ((nseq - 128) << 8) + sq[1]; shdr = 2; }
        }
        if (!verdict && (nseq == 0 || nseq > DSEQ_MAX || sn < shdr + 2 || sq[shdr] != 0)) verdict = 1;
        if (!verdict && sq[sn - 1] == 0) verdict = 2; // no end mark
    }
    if (!on || verdict) nseq = 0;
    const uint8_t *bp = sq + shdr + 1;            // the bit stream
    const int bn = on && !verdict ? (int)(sn - shdr - 1) : 0;
    uint32_t *win = T.win + lane * SEQ_WIN_STRIDE;
    SeqBits br;
    br.buf = 0; br.avail = 0; br.byte_pos = bn; br.wbase = 0;
    int bits_left = 0;
    if (nseq) {
        seq_window_load(win, bp, br.byte_pos, &br.wbase);
        seq_refill(br, win);
        seq_refill(br, win);
        const int pad = 8 - highbit32_d(bp[bn - 1]); // end mark and the zero bits above it
        br.buf <<= pad;
        br.avail -= pad;
        bits_left = bn * 8 - pad;
    }
    uint32_t st_ll = 0, st_of = 0, st_ml = 0;
    if (nseq) {
        seq_refill(br, win);
        st_ll = seq_take(br, 6); st_of = seq_take(br, 5); st_ml = seq_take(br, 6);
        bits_left -= 17;
    }
    uint32_t o = 0, lit_used = 0, prev_off = 0;
    for (uint32_t i = 0; i < nseq; i++) {
        if (br.byte_pos - br.wbase < 12 && br.wbase > 0) seq_window_load(win, bp, br.byte_pos, &br.wbase); // (three refills may follow before the next check)
        const uint2 e_ll = T.ll[st_ll & 63], e_of = T.of[st_of & 31], e_ml = T.ml[st_ml & 63];
        const uint32_t oc = e_of.y;
        if (oc > 24) { verdict = 1; break; }
        seq_refill(br, win);
        const uint32_t of_val = (1u << oc) + seq_take(br, oc);
        seq_refill(br, win);
        const uint32_t mb = e_ml.y >> 24, lb = e_ll.y >> 24;
        const uint32_t ml = (e_ml.y & 0xFFFFFFu) + seq_take(br, mb);
        const uint32_t ll = (e_ll.y & 0xFFFFFFu) + seq_take(br, lb);
        bits_left -= (int)(oc + mb + lb);
        if (i + 1 < nseq) {
            seq_refill(br, win);
            const uint32_t nl = (e_ll.x >> 8) & 0xFF, nm = (e_ml.x >> 8) & 0xFF, no = (e_of.x >> 8) & 0xFF;
            st_ll = (e_ll.x >> 16) + seq_take(br, nl);
            st_ml = (e_ml.x >> 16) + seq_take(br, nm);
            st_of = (e_of.x >> 16) + seq_take(br, no);
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
    if (!on) return;
    if (!verdict && (bits_left != 0 || o + (c.regen - lit_used) != c.out_len)) verdict = 2;
    *(uint32_t *)(scr + 2 * FQZ_CHUNK) = verdict ? DSEQ_INVALID : nseq;
    // (a corrupt block is left to the general path as well: it gives the authoritative verdict)
    if (verdict) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
}

#define SEQ_LWIN 2048u // literal bytes of a batch of sequences staged at a time
struct SeqExecLds {
    uint8_t out[FQZ_CHUNK + 16];
    uint8_t lw[SEQ_LWIN + 32];
};
// LDS operations of one wave execute in issue order: a match may read what the match before it wrote without waiting for
// it, as long as the compiler keeps the order
#define SEQ_LDS_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

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
    uint32_t o = 0, lp = 0;
    // literal bytes [from, from + n) of the block -> S.lw (n <= SEQ_LWIN); `from` is arbitrary, the loads are 16-byte pieces
    auto stage = [&](uint32_t from, uint32_t n) {
        const uint32_t a0 = from & ~15u;
        for (uint32_t i = lane * 16; i < n + (from - a0); i += 64 * 16) *(uint4 *)&S.lw[i] = *(const uint4 *)(scr + a0 + i); // (the scratch is padded)
        return from - a0; // S.lw[that + k] = literal from + k
    };
    for (uint32_t base = 0; base < nseq; base += 64) {
        const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
        uint2 q = make_uint2(0, 0);
        if (lane < cnt) q = seqs[base + lane];
        const uint32_t ll = q.x & 0xFFFFu, ml = q.x >> 16;
        const uint32_t in_o = wave_incl_scan(ll + ml), in_l = wave_incl_scan(ll);
        const uint32_t op = o + in_o - (ll + ml), lq = in_l - ll;   // literals go to out[op, op + ll); lq: offset in the batch's literals
        const uint32_t l_tot = (uint32_t)__builtin_amdgcn_readlane((int)in_l, 63);
        for (uint32_t w0 = 0; w0 < l_tot; w0 += SEQ_LWIN) {          // the batch's literals, a window at a time
            const uint32_t wn = l_tot - w0 < SEQ_LWIN ? l_tot - w0 : SEQ_LWIN;
            SEQ_LDS_ORDER();
            const uint32_t sh = stage(lp + w0, wn);
            SEQ_LDS_ORDER();
            const uint32_t a = lq > w0 ? lq : w0, b = lq + ll < w0 + wn ? lq + ll : w0 + wn; // this lane's part of the window
            for (uint32_t k = a; k < b; k++) S.out[op + (k - lq)] = S.lw[sh + (k - w0)];
        }
        SEQ_LDS_ORDER();
        const uint32_t mdst = op + ll;
        for (uint32_t j = 0; j < cnt; j++) { // matches in order: a match may read what the one before it wrote
            const uint32_t mlj = (uint32_t)__builtin_amdgcn_readlane((int)ml, (int)j), dj = (uint32_t)__builtin_amdgcn_readlane((int)mdst, (int)j);
            const uint32_t fj = (uint32_t)__builtin_amdgcn_readlane((int)q.y, (int)j);
            const uint8_t *ms = S.out + dj - fj;
            if (fj >= mlj) { for (uint32_t k = lane; k < mlj; k += 64) S.out[dj + k] = ms[k]; }
            else { for (uint32_t k = lane; k < mlj; k += 64) S.out[dj + k] = ms[k % fj]; } // overlapping: a pattern fill
            SEQ_LDS_ORDER();
        }
        o += (uint32_t)__builtin_amdgcn_readlane((int)in_o, 63);
        lp += l_tot;
    }
    for (uint32_t w0 = lp; w0 < n_lit; w0 += SEQ_LWIN) { // the literals behind the last match
        const uint32_t wn = n_lit - w0 < SEQ_LWIN ? n_lit - w0 : SEQ_LWIN;
        SEQ_LDS_ORDER();
        const uint32_t sh = stage(w0, wn);
        SEQ_LDS_ORDER();
        for (uint32_t k = lane; k < wn; k += 64) S.out[o + (w0 - lp) + k] = S.lw[sh + k];
    }
    SEQ_LDS_ORDER();
    uint8_t *dst = arena + c.out_off; // 16-byte aligned (a chunk of a 16-aligned stream)
    const uint32_t full = n_out & ~15u;
    for (uint32_t i = lane * 16; i < full; i += 64 * 16) *(uint4 *)(dst + i) = *(const uint4 *)&S.out[i];
    for (uint32_t i = full + lane; i < n_out; i += 64) dst[i] = S.out[i];
}
