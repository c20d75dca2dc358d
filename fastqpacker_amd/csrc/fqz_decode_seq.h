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

// k_dec_seq_fse: FOUR lanes per block (16 blocks per wave): lane 0 follows the literal-length state, lane 1 the match-length
// state, lane 2 the offset state (lane 3 idles), all with the same instructions: a step is one table read (per decoding
// state, packed for one 64-bit LDS read: x = symbol | nbits << 8 | next-state base << 16 (c_seq_dt), y = base of the value the
// symbol stands for | its extra bits << 24), an exchange of the six bit counts inside the quad (DPP), and two bit-field reads
// from the block's stream window in LDS.  The bit position is the only thing the three chains share.
#define SEQ_WIN 128u          // bytes of a block's bit stream staged in LDS at a time (a multiple of 16)
#define SEQ_WIN_STRIDE 33u    // dwords between the windows of neighbouring blocks
struct SeqFseLds {
    uint2 tab[3][64];
    uint32_t win[16 * SEQ_WIN_STRIDE + 4];
};
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) // the value of lane K of this lane's quad
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}
// bits [lo, lo + n) of the stream (n <= 24) from the window that starts at stream byte wb (a multiple of 4); branch-free
__device__ __forceinline__ uint32_t seq_bits(const uint32_t *win, int wb, int lo, uint32_t n)
{
    const uint32_t rel = (uint32_t)(lo - 8 * wb), d = (rel >> 5) & 31u;
    const uint32_t v = __builtin_amdgcn_alignbit(win[d + 1], win[d], rel & 31u);
    return v & ((1u << n) - 1u);
}

__global__ __launch_bounds__(64) void k_dec_seq_fse(const uint8_t *in, DecInfo *info, const DecChunk *chunks, uint8_t *arena, uint32_t first)
{
    __shared__ SeqFseLds T;
    __builtin_amdgcn_s_setprio(3); // a serial chain of short steps beside kernels that fill every issue slot: its waves go first
    const uint32_t lane = threadIdx.x, g = lane >> 2, c = lane & 3, id = first + blockIdx.x * 16 + g; // (chunks from `first` on: not the qualities)
    if (info->status) return;
    DecChunk ch;
    ch.seq_len = 0;
    if (id < info->n_chunks) ch = chunks[id];
    if (!__ballot(ch.seq_len != 0)) return;
    {
        const uint32_t el = c_seq_dt.ll[lane], em = c_seq_dt.ml[lane];
        T.tab[0][lane] = make_uint2(el, c_ll_base[el & 0xFF] | ((uint32_t)c_ll_bits[el & 0xFF] << 24));
        T.tab[1][lane] = make_uint2(em, c_ml_base[em & 0xFF] | ((uint32_t)c_ml_bits[em & 0xFF] << 24));
        const uint32_t eo = c_seq_dt.of[lane & 31];
        T.tab[2][lane] = make_uint2(eo, (eo & 0xFF) << 24); // offsets: the code is the number of extra bits, the base 1 << code
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (!ch.seq_len) return; // (whole quads leave)
    uint8_t *scr = arena + ch.dst_off;
    uint2 *out = (uint2 *)(scr + FQZ_CHUNK);
    const uint8_t *sq = in + ch.src_off + ch.csize - ch.seq_len;
    const uint32_t sn = ch.seq_len;
    int verdict = 0; // 0 fine, 1 = not our profile (general path), 2 = corrupt
    uint32_t nseq = sq[0], shdr = 1;
    if (nseq >= 128) {
        if (nseq == 255 || sn < 2) verdict = 1;
        else { nseq = ((nseq - 128) << 8) + sq[1]; shdr = 2; }
    }
    if (!verdict && (nseq == 0 || nseq > DSEQ_MAX || sn < shdr + 2 || sq[shdr] != 0)) verdict = 1;
    if (!verdict && sq[sn - 1] == 0) verdict = 2; // no end mark
    const uint8_t *bp = sq + shdr + 1;             // the bit stream: bn bytes, read from its last bit down
    const int bn = verdict ? 0 : (int)(sn - shdr - 1);
    uint32_t *win = T.win + g * SEQ_WIN_STRIDE;
    int p = 0, wb = 0;                              // bits [0, p) are still to be read; the window holds the bytes [wb, wb + SEQ_WIN)
    auto window = [&]() {
        int top = (p + 7) >> 3;
        wb = top > (int)SEQ_WIN ? (top - (int)SEQ_WIN + 3) & ~3 : 0;
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            const uint32_t piece = c + 4 * q;
            uint4 v = make_uint4(0, 0, 0, 0);
            const int at = wb + (int)(16 * piece);
            if (at + 16 <= bn) v = load_u128_unaligned(bp + at);
            else if (at < bn) { // the last piece of the stream: byte by byte, nothing behind the section is touched
                uint32_t w[4] = {0, 0, 0, 0};
                for (int k = 0; at + k < bn; k++) w[k >> 2] |= (uint32_t)bp[at + k] << (8 * (k & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            win[4 * piece] = v.x; win[4 * piece + 1] = v.y; win[4 * piece + 2] = v.z; win[4 * piece + 3] = v.w;
        }
        if (c == 0) win[32] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (read by the other lanes of the quad: LDS operations of a wave execute in order)
        __builtin_amdgcn_wave_barrier();
    };
    uint32_t st = 0;
    if (!verdict) {
        p = bn * 8 - (8 - highbit32_d(bp[bn - 1])); // without the end mark and the zero bits above it
        if (p < 17) verdict = 2;
    }
    if (!verdict) {
        window();
        // initial states: literal lengths (6 bits), offsets (5), match lengths (6)
        st = seq_bits(win, wb, c == 0 ? p - 6 : (c == 2 ? p - 11 : p - 17), c == 2 ? 5u : 6u);
        p -= 17;
    }
    uint32_t o = 0, lit_used = 0, prev_off = 0;
    const uint32_t cc = c == 3 ? 2u : c; // (the idle lane walks along with the offsets chain: its results are dropped)
    const uint32_t m_lt2 = cc < 2 ? ~0u : 0u, m_lt1 = cc < 1 ? ~0u : 0u, m_ge1 = cc >= 1 ? ~0u : 0u, m_ge2 = cc >= 2 ? ~0u : 0u;
    uint32_t bad_profile = 0, bad_data = 0;
    uint2 held = make_uint2(0, 0);
    for (uint32_t i = 0; i < (verdict ? 0u : nseq); i++) { // one rarely-taken branch a step; everything else is straight-line
        // the window is refilled by all sixteen blocks of the wave together, when the first of them runs short: a refill is a global
        // round trip the whole wave waits for (1-2 us), and with every block refilling on its own schedule - each every ~35
        // sequences - some block did in every other step (0.28 ms for chains that take 0.06)
        if (__ballot(p - 80 < 8 * wb && wb > 0) != 0ull) { if (wb > 0) window(); }
        const uint2 e = T.tab[cc][st & 63];
        const bool last = i + 1 == nseq;
        const uint32_t xb = e.y >> 24, sb = last ? 0u : (e.x >> 8) & 0xFF;
        const uint32_t x0 = quad_bcast<0>(xb), x1 = quad_bcast<1>(xb), x2 = quad_bcast<2>(xb);
        const uint32_t s0 = quad_bcast<0>(sb), s1 = quad_bcast<1>(sb), s2 = quad_bcast<2>(sb);
        const uint32_t xs = x0 + x1 + x2, total = xs + s0 + s1 + s2;
        bad_profile |= x2 > 24;
        bad_data |= (int)total > p;
        // stream order (from the top): offset extra bits, match-length extra bits, literal-length extra bits, then the state
        // bits of literal lengths, match lengths, offsets
        const uint32_t endA = x2 + (x1 & m_lt2) + (x0 & m_lt1);
        const uint32_t endB = xs + s0 + (s1 & m_ge1) + (s2 & m_ge2);
        const uint32_t A = seq_bits(win, wb, p - (int)endA, xb), B = seq_bits(win, wb, p - (int)endB, sb);
        const uint32_t val = (cc == 2 ? 1u << (xb & 31u) : e.y & 0xFFFFFFu) + A;
        st = last ? st : (e.x >> 16) + B;
        p -= (int)total;
        const uint32_t ll = quad_bcast<0>(val), ml = quad_bcast<1>(val), of_val = quad_bcast<2>(val);
        // an explicit offset, or "the offset of the previous sequence" (of this block); anything else is not our profile
        const bool rep = of_val <= 3;
        bad_profile |= rep & !((of_val == 1) & (ll > 0) & (i > 0));
        const uint32_t offset = rep ? prev_off : of_val - 3;
        prev_off = offset;
        bad_data |= (ll > ch.regen - lit_used) | (ll > ch.out_len - o) | (ml > ch.out_len - o - ll);
        bad_profile |= offset > o + ll; // reaches in front of the block: not ours
        // two triples a store (16 bytes): half the store instructions in a kernel that runs beside the Huffman decode
        const uint2 cur = make_uint2(ll | (ml << 16), offset);
        if (c == 0 && (i & 1)) *(uint4 *)&out[i - 1] = make_uint4(held.x, held.y, cur.x, cur.y);
        if (c == 0 && last && !(i & 1)) out[i] = cur;
        held = cur;
        lit_used += ll;
        o += ll + ml;
        if (bad_profile | bad_data) break;
    }
    if (!verdict && bad_data) verdict = 2;
    if (!verdict && bad_profile) verdict = 1;
    if (!verdict && (p != 0 || o + (ch.regen - lit_used) != ch.out_len)) verdict = 2;
    if (c == 0) {
        *(uint32_t *)(scr + 2 * FQZ_CHUNK) = verdict ? DSEQ_INVALID : nseq;
        // (a corrupt block is left to the general path as well: it gives the authoritative verdict)
        if (verdict) dec_fail(info, FQZ_DEC_RETRY_GENERAL);
    }
}

// k_dec_seq_exec: a wave per block.  A match almost always copies from the record before it, so the wave keeps only the last
// SEQ_RING bytes of its output in LDS (a ring, flushed to the stream in 16-byte pieces as it advances) and a small window of
// the literals: ~6 KiB a wave instead of the block's 16 KiB + literals, which is what decides how many blocks a CU works on at
// once - the sequences of a block are a serial chain (a match may read what the match before it wrote), every step of it one
// LDS round trip.  Sequences are taken in batches of up to 64 whose output fits half the ring: a wave scan places all their
// literal runs at once, then the matches run in order.  Sequences too large for that (a run or a match longer than a
// quarter of the ring, an offset beyond half of it) go one by one through a slower path that is still exact.
#define SEQ_RING 4096u
#define SEQ_RMASK (SEQ_RING - 1u)
#define SEQ_LWIN 1024u // literal bytes staged at a time
struct SeqExecLds {
    uint8_t ring[SEQ_RING];
    uint8_t lw[SEQ_LWIN + 32];
};
// LDS operations of one wave execute in issue order: a match may read what the match before it wrote without waiting for
// it, as long as the compiler keeps the order
#define SEQ_LDS_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__global__ __launch_bounds__(64) void k_dec_seq_exec(DecInfo *info, const DecChunk *chunks, uint8_t *arena, uint32_t first)
{
    __shared__ __attribute__((aligned(16))) SeqExecLds S;
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x, id = first + blockIdx.x;
    if (id >= info->n_chunks || info->status) return;
    const DecChunk c = chunks[id];
    if (!c.seq_len) return;
    const uint8_t *scr = arena + c.dst_off;
    const uint2 *seqs = (const uint2 *)(scr + FQZ_CHUNK);
    const uint32_t nseq = *(const uint32_t *)(scr + 2 * FQZ_CHUNK);
    if (nseq == DSEQ_INVALID || nseq > DSEQ_MAX) return;
    const uint32_t n_lit = c.regen, n_out = c.out_len;
    uint8_t *dst = arena + c.out_off; // 16-byte aligned (a chunk of a 16-aligned stream)
    uint32_t o = 0, lp = 0;           // output bytes produced, literals consumed
    uint32_t flushed = 0;             // (a multiple of 16) output bytes [0, flushed) are in the stream, [flushed, o) only in the ring
    auto flush = [&]() {              // whole 16-byte pieces below o leave the ring
        const uint32_t to = o & ~15u;
        SEQ_LDS_ORDER();
        for (uint32_t i = flushed + lane * 16; i < to; i += 64 * 16) *(uint4 *)(dst + i) = *(const uint4 *)&S.ring[i & SEQ_RMASK];
        flushed = to;
    };
    // the literal window: S.lw[ws_sh + k] = literal ws + k for k < ws_n
    uint32_t ws = 0, ws_n = 0, ws_sh = 0;
    auto stage = [&](uint32_t from) { // literals [from, from + SEQ_LWIN) of the block (as far as they exist), 16-byte pieces
        const uint32_t a0 = from & ~15u, n = n_lit - from < SEQ_LWIN ? n_lit - from : SEQ_LWIN;
        SEQ_LDS_ORDER();
        for (uint32_t i = lane * 16; i < n + (from - a0); i += 64 * 16) *(uint4 *)&S.lw[i] = *(const uint4 *)(scr + a0 + i); // (the scratch is padded)
        SEQ_LDS_ORDER();
        ws = from; ws_n = n; ws_sh = from - a0;
    };
    // literals [lf, lf + n) -> output [at, at + n), all lanes (n may be large: window by window, the ring flushed as it fills)
    auto run_all = [&](uint32_t lf, uint32_t at, uint32_t n) {
        for (uint32_t done = 0; done < n;) {
            if (lf + done < ws || lf + done >= ws + ws_n) stage(lf + done);
            uint32_t m = ws + ws_n - (lf + done);
            m = m < n - done ? m : n - done;
            m = m < SEQ_RING / 4 ? m : SEQ_RING / 4;
            o = at + done;
            flush();
            for (uint32_t k = lane; k < m; k += 64) S.ring[(at + done + k) & SEQ_RMASK] = S.lw[ws_sh + (lf + done - ws) + k];
            done += m;
        }
        o = at + n;
    };
    for (uint32_t base = 0; base < nseq;) {
        uint2 q = make_uint2(0, 0);
        if (base + lane < nseq) q = seqs[base + lane];
        const uint32_t ll = q.x & 0xFFFFu, ml = q.x >> 16, off = q.y;
        const uint32_t in_o = wave_incl_scan(ll + ml), in_l = wave_incl_scan(ll);
        // the batch: the leading sequences whose output fits half the ring, none of them "large"
        const bool large = ll > SEQ_RING / 4 || ml > SEQ_RING / 4 || off > SEQ_RING / 2;
        const unsigned long long fits = __ballot(base + lane < nseq && !large && in_o <= SEQ_RING / 2 - 16);
        const uint32_t cnt = ~fits ? (uint32_t)__builtin_ctzll(~fits) : 64u; // the run of ones from lane 0 up
        if (cnt == 0) { // one large sequence, by all lanes
            const uint32_t ll0 = (uint32_t)__builtin_amdgcn_readlane((int)ll, 0), ml0 = (uint32_t)__builtin_amdgcn_readlane((int)ml, 0);
            const uint32_t off0 = (uint32_t)__builtin_amdgcn_readlane((int)off, 0);
            run_all(lp, o, ll0);
            lp += ll0;
            const uint32_t d0 = o;
            // the match in pieces: a piece never reads what it writes itself (so the lanes of a piece are independent)
            const uint32_t piece_max = off0 >= SEQ_RING / 4 ? SEQ_RING / 4 : (off0 >= 64 ? off0 : off0 * (64 / off0));
            for (uint32_t done = 0; done < ml0;) {
                const uint32_t m = ml0 - done < piece_max ? ml0 - done : piece_max;
                o = d0 + done;
                flush();
                if (off0 > SEQ_RING / 2) { // the source may have left the ring: it is in the stream (flushed above: at least SEQ_RING / 2 - 16 bytes back)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __builtin_amdgcn_s_waitcnt(0);
                    for (uint32_t k = lane; k < m; k += 64)
                        S.ring[(d0 + done + k) & SEQ_RMASK] = __hip_atomic_load(dst + d0 + done + k - off0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    SEQ_LDS_ORDER();
                    for (uint32_t k = lane; k < m; k += 64) S.ring[(d0 + done + k) & SEQ_RMASK] = S.ring[(d0 + done - off0 + (k % off0)) & SEQ_RMASK];
                }
                SEQ_LDS_ORDER();
                done += m;
            }
            o = d0 + ml0;
            base += 1;
            continue;
        }
        flush(); // (what the last batch produced; the ring has room for this one: half of it, and every source is at most half of it back)
        const uint32_t l_tot = (uint32_t)__builtin_amdgcn_readlane((int)in_l, (int)cnt - 1), o_tot = (uint32_t)__builtin_amdgcn_readlane((int)in_o, (int)cnt - 1);
        const uint32_t op = o + in_o - (ll + ml), lq = lp + in_l - ll; // literals [lq, lq + ll) go to output [op, op + ll), the match behind them
        const uint32_t my_ll = lane < cnt ? ll : 0u;
        for (uint32_t w0 = lp; w0 < lp + l_tot;) { // the batch's literals through the window
            if (w0 < ws || w0 >= ws + ws_n || (lp + l_tot > ws + ws_n && w0 == lp && ws + ws_n < n_lit)) stage(w0);
            const uint32_t w1 = ws + ws_n < lp + l_tot ? ws + ws_n : lp + l_tot;
            const uint32_t a = lq > w0 ? lq : w0, b = lq + my_ll < w1 ? lq + my_ll : w1; // this lane's part of [w0, w1)
            for (uint32_t k = a; k < b; k++) S.ring[(op + (k - lq)) & SEQ_RMASK] = S.lw[ws_sh + (k - ws)];
            w0 = w1;
        }
        SEQ_LDS_ORDER();
        const uint32_t mdst = op + ll;
        for (uint32_t j = 0; j < cnt; j++) { // matches in order: a match may read what the one before it wrote
            const uint32_t mlj = (uint32_t)__builtin_amdgcn_readlane((int)ml, (int)j), dj = (uint32_t)__builtin_amdgcn_readlane((int)mdst, (int)j);
            const uint32_t fj = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)j);
            if (mlj <= 64 && fj >= mlj) { if (lane < mlj) S.ring[(dj + lane) & SEQ_RMASK] = S.ring[(dj - fj + lane) & SEQ_RMASK]; } // (nearly all of them)
            else if (fj >= mlj) { for (uint32_t k = lane; k < mlj; k += 64) S.ring[(dj + k) & SEQ_RMASK] = S.ring[(dj - fj + k) & SEQ_RMASK]; }
            else { for (uint32_t k = lane; k < mlj; k += 64) S.ring[(dj + k) & SEQ_RMASK] = S.ring[(dj - fj + (k % fj)) & SEQ_RMASK]; } // overlapping: a pattern fill
            SEQ_LDS_ORDER();
        }
        o += o_tot;
        lp += l_tot;
        base += cnt;
    }
    run_all(lp, o, n_lit - lp); // the literals behind the last match
    flush();
    SEQ_LDS_ORDER();
    for (uint32_t i = flushed + lane; i < n_out; i += 64) dst[i] = S.ring[i & SEQ_RMASK];
}
