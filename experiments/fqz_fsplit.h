// fqz_fsplit.h — the fused, single-pass front end of the encode pipeline (gfx950 / MI355X).
//
// One kernel reads the FASTQ text ONCE and leaves the pre-entropy streams in HBM.  It replaces, fused,
//   fqparser.readLine / nextInto          internal/fqparser/parser.go:136-243   (line index, record grammar, CR stripping)
//   the record loop of compressBlockWithBuffers  internal/compress/compress.go:474-520
//     encoder.AppendPackedBases           internal/encoder/sequence.go:139-184
//     encoder.NormalizeQuality+DeltaEncode internal/encoder/quality.go:53-103
//   encoder.DetectEncoding                internal/encoder/quality.go:22-49     (mode 1: first block only)
//
// A workgroup takes one tile of text after the other (tickets).  It stages the tile in LDS, finds its newlines
// (registers -> a list in LDS), learns how many newlines precede the tile from a decoupled look-back over the per-tile
// counts, and thereby which lines are header lines (line index mod 4).  It OWNS the records whose header line starts
// inside the tile; the up to four newlines a straddling last record needs beyond the tile are found by a short forward
// scan.  The sizes of the owned records in the seq / qual / headers / plus streams are summed and a second look-back
// (four columns) turns them into absolute stream offsets; the records are then cut into 16-byte pieces and split
// piece-centric (a wave per <= 64 records, every lane one piece) reading the text from LDS.  The line index, the record
// table and the offset table never touch HBM, and between the tile load and the stream stores the only global round
// trips of a tile are the two look-backs (measured: every dependent global access costs 2-3 us under load, which is
// what a version that re-read the text through L2 spent its time on).
//
// What still goes to HBM per record: the N-position byte count (k_npos_write places the rare N lists once their total
// is scanned), the text offset of the sequence line and the read length (12 B / record).
#pragma once
#include "fqz_device.h"

#ifndef FS_NT
#define FS_NT 256u                 // threads per workgroup
#endif
#ifndef FS_TILE_BYTES
#define FS_TILE_BYTES 32768u       // text bytes per tile (4096 for texts with very short lines)
#endif
#ifndef FS_WAVES_MIN
#define FS_WAVES_MIN 6
#endif
#ifndef FS_LDS_TEXT
#define FS_LDS_TEXT 0 // 1: stage the tile's text in LDS for the split (fewer, larger workgroups per CU); 0: the split re-reads its pieces through L2
#endif
#ifndef FS_POLL_SLEEP
#define FS_POLL_SLEEP 1 // s_sleep units (64 cycles) between two polls of a look-back
#endif
#ifndef FS_ABLATE
#define FS_ABLATE 0 // timing experiments only: 1 leave after the first look-back, 2 after the second, 3 skip the piece loops
#endif
#define FS_NW (FS_NT / 64u)
#define FS_ST_AGG (1ull << 62)
#define FS_ST_PREFIX (2ull << 62)
#define FS_ST_VALUE ((1ull << 62) - 1)
#define FS_HUGE (1ull << 40)       // "beyond anything": published by tiles that leave early in detect mode
#define FS_NONE 0xFFFFFFFFu
#ifndef FS_STRIP
#define FS_STRIP 256u              // records per strip (one lane per record)
#endif
#define FS_STAMP(k) do { if (a.stamps && t == 0) a.stamps[(size_t)tile * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

struct FsRegions { uint8_t *p[FQZ_NS]; }; // base of every pre-entropy stream region (S_NPOS: the nPos arena)

struct FsBStart { // what the tile that owns record b * rpb knows about the start of block b
    uint32_t text;   // text offset of the record's header line (FS_NONE: not seen)
    uint32_t e[4];   // bytes of the seq / qual / headers / plus streams before the record (whole batch)
    uint32_t pad[3];
};

struct FsArgs {
    const uint8_t *text;
    uint32_t n, n_tiles;
    EncInfo *info;
    unsigned long long *st1; // [n_tiles]     look-back state of the newline counts
    unsigned long long *st2; // [n_tiles][4]  look-back state of the stream sizes
    uint32_t *ticket;
    FsRegions reg;
    uint32_t *Enpos;   // [rec_cap + 1] 2 + 2 * (N count) per record
    uint32_t *rec_seq; // [rec_cap]     text offset of the sequence line
    uint32_t *rec_L;   // [rec_cap]     read length
    FsBStart *bstart;  // [block_cap + 1]
    uint32_t rec_cap, block_cap, rpb, final_batch, mode; // mode 0 = encode, 1 = DetectEncoding over block 0 only
    unsigned long long *stamps; // diagnostic (FQZ_DBG_FS_STAMPS): 8 x s_memtime per tile
};

// one full wave: exclusive prefix of `tot` over the tiles before `tile`; publishes this tile's inclusive prefix
__device__ __forceinline__ unsigned long long fs_lookback(unsigned long long *state, uint32_t stride, uint32_t tile, uint32_t lane, unsigned long long tot)
{
    unsigned long long excl = 0;
    if (FS_ABLATE == 4) return 0; // (timing experiment: no hand-offs at all)
    if (tile > 0) {
        if (lane == 0) __hip_atomic_store(&state[(size_t)tile * stride], FS_ST_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int look = (int)tile - 1;
        for (;;) {
            const int idx = look - (int)lane;
            const unsigned long long sv = idx >= 0 ? __hip_atomic_load(&state[(size_t)idx * stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FS_ST_PREFIX;
            const uint32_t flag = (uint32_t)(sv >> 62);
            const unsigned long long pmask = __ballot(flag == 2), zmask = __ballot(flag == 0);
            const int fp = pmask ? __ffsll((long long)pmask) - 1 : 64;          // nearest predecessor with a full prefix
            const unsigned long long need = fp >= 63 ? ~0ull : ((2ull << fp) - 1); // lanes 0..fp must have published
            if (zmask & need) { __builtin_amdgcn_s_sleep(FS_POLL_SLEEP); continue; }
            unsigned long long part = (int)lane <= fp ? (sv & FS_ST_VALUE) : 0ull;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, WAVE);
            excl += part;
            if (pmask) break;
            look -= 64;
        }
    }
    if (lane == 0) __hip_atomic_store(&state[(size_t)tile * stride], FS_ST_PREFIX | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

template <uint32_t TILE>
struct FsLds {
    static constexpr uint32_t NLCAP = TILE == 4096u ? 4096u : TILE / 8u; // newline list capacity (entries)
    static constexpr uint32_t R = TILE / (16u * FS_NT) ? TILE / (16u * FS_NT) : 1u; // rounds of FS_NT x 16 bytes
    uint8_t text[FS_LDS_TEXT ? TILE + 32 : 16];  // the tile (bytes at or beyond the end of the text read as 0)
    uint16_t nl[NLCAP + 8];   // tile-local positions of the tile's newlines, in text order
    uint32_t ext[4];          // absolute positions of the first newlines behind the tile
    uint32_t n_ext;
    uint32_t wt[R][FS_NW];    // newline counts per (round, wave)
    uint32_t G_lo, G_hi;      // newlines before the tile
    uint32_t csum[4][FS_NW];  // per column, per wave
    uint32_t excl[4];         // stream bytes before the tile
    uint32_t next_tile;
    // record table of the strip in flight
    uint32_t s_hdr[FS_STRIP], s_seq[FS_STRIP], s_plus[FS_STRIP], s_qual[FS_STRIP];
    uint32_t L[FS_STRIP], H[FS_STRIP], P[FS_STRIP];
    uint32_t d_seq[FS_STRIP], d_qual[FS_STRIP], d_hdr[FS_STRIP], d_plus[FS_STRIP];
    uint32_t nn[FS_STRIP];    // "complete" while the table is filled; then the N bases (first 65536 positions) counted by the pieces
};

// 16 bytes at tile offset `off` (off + 16 <= TILE) from the staged text: aligned dwords + a funnel shift per dword
// (unaligned 64/128-bit LDS accesses are replayed; 32-bit ones are not)
__device__ __forceinline__ void fs_lds_piece(const uint8_t *tile_text, uint32_t off, uint32_t x[4])
{
    const uint32_t *p = (const uint32_t *)(tile_text + (off & ~3u));
    const uint32_t sh = off & 3u;
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
    x[0] = __builtin_amdgcn_alignbyte(w1, w0, sh);
    x[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
    x[2] = __builtin_amdgcn_alignbyte(w3, w2, sh);
    x[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
}

// TILE: text bytes per workgroup pass, FS_TILE_BYTES (default) or 4096 (any text: a 4 KiB tile cannot hold more newlines than the list)
template <uint32_t TILE>
__global__ __launch_bounds__(FS_NT) __attribute__((amdgpu_waves_per_eu(FS_WAVES_MIN, 8))) void k_fsplit(const FsArgs a)
{
    constexpr uint32_t R = FsLds<TILE>::R;
    constexpr uint32_t NLCAP = FsLds<TILE>::NLCAP;
    static_assert(TILE <= 65536u && R <= 8u && (TILE % (16u * FS_NT) == 0 || TILE < 16u * FS_NT), "tile shape");
    __shared__ __attribute__((aligned(16))) FsLds<TILE> S;
    const uint8_t *__restrict__ text = a.text;
    const uint32_t n = a.n, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    EncInfo *info = a.info;
    if (t == 0) S.next_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    uint32_t tile = S.next_tile;
    while (tile < a.n_tiles) {
        __syncthreads(); // (everyone has read next_tile; the tables of the previous tile are free)
        const uint32_t tbase = tile * TILE;
        FS_STAMP(0);
        bool live = true; // false: this pass has nothing (more) to do for its tile
        if (a.mode == 1 && tile > __hip_atomic_load(&info->detect_done_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            // DetectEncoding looks at block 0 only, and an earlier tile already starts beyond it: nothing to do, but the
            // tiles behind this one must not wait for it
            if (t == 0) __hip_atomic_store(&a.st1[tile], FS_ST_PREFIX | FS_HUGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            live = false;
        }
        uint32_t cnt_tile = 0;
        if (live) {
            // ---- phase A: stage the tile, find its newlines.  Text order = (round, thread, byte).  Per round a lane keeps a
            //      16-bit mask (bit b: byte b of its 16 bytes is '\n'), two rounds per register.
            uint32_t m2[(R + 1) / 2];
            uint32_t cpk[(R + 1) / 2]; // newline counts of the two rounds, 2 x 16 bits
            uint4 v[R];
#pragma unroll
            for (uint32_t r = 0; r < R; r++) {
                v[r] = make_uint4(0, 0, 0, 0);
                if (r * FS_NT * 16u + 16u * t < TILE) v[r] = load_text16(text, tbase + r * FS_NT * 16u + 16u * t, n);
            }
#pragma unroll
            for (uint32_t r = 0; r < R; r++) {
                if (FS_LDS_TEXT && r * FS_NT * 16u + 16u * t < TILE) *(uint4 *)&S.text[FS_LDS_TEXT ? r * FS_NT * 16u + 16u * t : 0] = v[r];
                const uint32_t w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
                uint32_t m16 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t z = zero_bytes(w[k] ^ 0x0A0A0A0Au);
                    m16 |= ((((z >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * k);
                }
                const uint32_t cr = __popc(m16);
                if (r & 1) { m2[r / 2] |= m16 << 16; cpk[r / 2] |= cr << 16; }
                else { m2[r / 2] = m16; cpk[r / 2] = cr; }
            }
            if (FS_LDS_TEXT && t < 2) *(uint4 *)&S.text[FS_LDS_TEXT ? TILE + 16 * t : 0] = make_uint4(0, 0, 0, 0);
            uint32_t expk[(R + 1) / 2]; // newlines of the round in earlier lanes of the wave, 2 x 16 bits (a wave's round holds <= 1024)
#pragma unroll
            for (uint32_t h = 0; h < (R + 1) / 2; h++) {
                const uint32_t incl = wave_incl_scan(cpk[h]);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                expk[h] = incl - cpk[h];
                if (lane == 0) { S.wt[2 * h][wave] = tot & 0xFFFF; if (2 * h + 1 < R) S.wt[2 * h + 1 < R ? 2 * h + 1 : 0][wave] = tot >> 16; }
            }
            __syncthreads();
            {
                // exclusive prefix of the (round, wave) counts in text order: every wave scans the R x NW table itself (one LDS read
                // per lane, one wave scan) and picks its entries with v_readlane; the table is wave-uniform data
                static_assert(R * FS_NW <= 64, "the (round, wave) table must fit one wave");
                const uint32_t mine = lane < R * FS_NW ? (&S.wt[0][0])[lane] : 0u;
                const uint32_t incl = wave_incl_scan(mine);
                cnt_tile = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t excl = incl - mine;
                const uint32_t uw = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
#pragma unroll
                for (uint32_t r = 0; r < R; r++) {
                    uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)(r * FS_NW + uw)) + ((expk[r / 2] >> (16 * (r & 1))) & 0xFFFF);
                    uint32_t mk = (m2[r / 2] >> (16 * (r & 1))) & 0xFFFF;
                    while (mk) {
                        const uint32_t b = (uint32_t)__ffs(mk) - 1;
                        mk &= mk - 1;
                        if (idx < NLCAP) S.nl[idx] = (uint16_t)(r * FS_NT * 16u + 16u * t + b);
                        idx++;
                    }
                }
            }
            FS_STAMP(1);
            // ---- phase B: wave 0 resolves the number of newlines before the tile; wave 1 finds the (up to) four newlines
            //      behind the tile that the last owned record may need
            if (wave == 0) {
                const unsigned long long G = fs_lookback(a.st1, 1, tile, lane, cnt_tile);
                if (lane == 0) { S.G_lo = (uint32_t)G; S.G_hi = (uint32_t)(G >> 32); }
            } else if (wave == 1) {
                uint32_t found = 0;
                for (unsigned long long pos = (unsigned long long)tbase + TILE; pos < n && found < 4; pos += 1024) {
                    const uint32_t off = (uint32_t)pos + 16u * lane;
                    const uint4 vv = load_text16(text, off, n);
                    const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
                    uint32_t mk = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t z = zero_bytes(w[k] ^ 0x0A0A0A0Au);
                        mk |= ((((z >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * k);
                    }
                    const uint32_t cc = __popc(mk);
                    const uint32_t incl = wave_incl_scan(cc);
                    uint32_t idx = found + incl - cc;
                    while (mk) {
                        const uint32_t b = (uint32_t)__ffs(mk) - 1;
                        mk &= mk - 1;
                        if (idx < 4) S.ext[idx] = off + b;
                        idx++;
                    }
                    found += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                if (lane == 0) S.n_ext = found < 4 ? found : 4u;
            }
            __syncthreads();
            FS_STAMP(2);
        }
        const unsigned long long G64 = live ? (((unsigned long long)S.G_hi << 32) | S.G_lo) : FS_HUGE;
        if (live && tile + 1 == a.n_tiles && t == 0 && a.mode == 0) info->n_lines = (uint32_t)(G64 + cnt_tile);
        const bool overflow = cnt_tile > NLCAP;
        if (live && overflow && t == 0) atomicOr(&info->index_overflow, 1u);
        if (G64 >= FS_HUGE) live = false; // (detect mode: behind a tile that left early)
        const uint32_t G = (uint32_t)G64;
        const uint32_t c_nl = overflow ? 0u : cnt_tile, n_avail = c_nl + (overflow || !live ? 0u : S.n_ext);
        // owned header lines: line 4r starts behind newline 4r - 1, which lies in this tile (record 0: tile 0)
        const uint32_t r_lo = tile == 0 ? 0u : G / 4u + 1u;
        const uint32_t r_hi1 = overflow ? r_lo : (G + c_nl) / 4u + 1u; // one past the last owned header line
        const uint32_t n_owned = live && r_hi1 > r_lo ? r_hi1 - r_lo : 0u;
        if (live && a.mode == 1 && r_lo >= a.rpb) {
            if (t == 0) atomicMin(&info->detect_done_tile, tile);
            live = false;
        }
        if (FS_ABLATE == 1) live = false;
        // absolute position of newline j of the tile's list (j < n_avail), list entries behind the tile included
        auto NL = [&](uint32_t j) -> uint32_t { return j < c_nl ? tbase + S.nl[j] : S.ext[j - c_nl]; };
        // text byte at absolute position pos >= tbase: from the staged tile when it lies inside
        auto TB = [&](uint32_t pos) -> uint32_t { return FS_LDS_TEXT && pos - tbase < TILE ? (uint32_t)S.text[FS_LDS_TEXT ? pos - tbase : 0] : (uint32_t)text[pos]; };
        struct RecF { uint32_t s0, s1, s2, s3, L, H, P, complete; };
        // fields of owned record i (header line 4 (r_lo + i)); check: report its format errors (parser.go:136-183)
        auto rec_fields = [&](uint32_t i, bool check) -> RecF {
            RecF f;
            const uint32_t r = r_lo + i;
            const int jb = (int)(4u * r - 1u - G); // list index of the newline in front of the header line (-1: start of the text)
            f.s0 = r == 0 ? 0u : NL((uint32_t)jb) + 1u;
            f.complete = (uint32_t)(jb + 4) < n_avail ? 1u : 0u;
            f.s1 = f.s2 = f.s3 = f.L = f.H = f.P = 0;
            if (f.complete) {
                const uint32_t q1 = NL((uint32_t)(jb + 1)), q2 = NL((uint32_t)(jb + 2)), q3 = NL((uint32_t)(jb + 3)), q4 = NL((uint32_t)(jb + 4));
                f.s1 = q1 + 1; f.s2 = q2 + 1; f.s3 = q3 + 1;
                // length without '\n' and without one trailing '\r' (parser.go:213-215)
                uint32_t l0 = q1 - f.s0, l1 = q2 - f.s1, l2 = q3 - f.s2, l3 = q4 - f.s3;
                // the six bytes the grammar looks at, fetched unconditionally so that the loads are in flight together (an
                // empty line reads its own newline, which is not a '\r')
                const uint32_t c0 = TB(q1 - (l0 ? 1 : 0)), c1 = TB(q2 - (l1 ? 1 : 0)), c2 = TB(q3 - (l2 ? 1 : 0)), c3 = TB(q4 - (l3 ? 1 : 0));
                const uint32_t b0 = TB(f.s0), b2 = TB(f.s2);
                l0 -= c0 == '\r'; l1 -= c1 == '\r'; l2 -= c2 == '\r'; l3 -= c3 == '\r';
                if (l0 == 0 || b0 != '@') { if (check) report_error(info, r, 0, FQZ_E_HDR_AT); l0 = 1; }
                if (l2 == 0 || b2 != '+') { if (check) report_error(info, r, 1, FQZ_E_SEP_PLUS); l2 = 1; }
                if (check && l1 != l3) report_error(info, r, 2, FQZ_E_LEN_MISMATCH);
                uint32_t H = l0 - 1, P = l2 - 1;
                if (H > 65535u || P > 65535u) { if (check) report_error(info, r, 3, FQZ_E_FIELD_WRAP); H &= 0xFFFF; P &= 0xFFFF; }
                f.L = l1; f.H = H; f.P = P;
            } else if (check && a.final_batch) {
                // the trailing partial record of the input: the lines that exist are still validated before EOF is hit
                // (parser.go:138-165), then it is dropped (parser.go:196-199)
                const uint32_t have = n_avail - (uint32_t)(jb + 1); // complete lines of the record: 0..3
                if (have >= 1 && TB(f.s0) != '@') report_error(info, r, 0, FQZ_E_HDR_AT);
                if (have >= 3) { const uint32_t s2 = NL((uint32_t)(jb + 2)) + 1; if (TB(s2) != '+') report_error(info, r, 1, FQZ_E_SEP_PLUS); }
            }
            return f;
        };
        if (live && a.mode == 1) {
            // ---- DetectEncoding (quality.go:22-49): minimum quality byte over the records of block 0
            uint32_t mn = 255;
            for (uint32_t i = wave; i < n_owned; i += FS_NW) {
                if (r_lo + i >= a.rpb) break;
                const RecF f = rec_fields(i, false);
                if (!f.complete) break;
                for (uint32_t k = lane; k < f.L; k += WAVE) { const uint32_t b = TB(f.s3 + k); mn = b < mn ? b : mn; }
            }
            mn = wave_min(mn);
            if (lane == 0 && mn < 255) atomicMin(&info->min_qual, mn);
            live = false;
        }
        if (live) {
            // ---- phase C: stream sizes of the owned records -> tile sums -> look-back -> stream offsets of the tile.
            //      The fields of the first strip's records go straight into the record table.
            uint32_t tsum[4] = {0, 0, 0, 0};
            for (uint32_t i = t; i < n_owned; i += FS_NT) {
                const RecF f = rec_fields(i, true);
                if (f.complete) { tsum[0] += (f.L + 3) >> 2; tsum[1] += f.L; tsum[2] += 2 + f.H; tsum[3] += 2 + f.P; }
                if (i < FS_STRIP) {
                    S.s_hdr[i] = f.s0 + 1; S.s_seq[i] = f.s1; S.s_plus[i] = f.s2 + 1; S.s_qual[i] = f.s3;
                    S.L[i] = f.L; S.H[i] = f.H; S.P[i] = f.P; S.nn[i] = f.complete;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t ws = wave_sum(tsum[q]); if (lane == 0) S.csum[q][wave] = ws; }
            __syncthreads();
            FS_STAMP(3);
            if (wave < 4) { // wave w resolves column w
                const uint32_t q = wave;
                const uint32_t tot = wave_sum(lane < FS_NW ? S.csum[q][lane] : 0u);
                const unsigned long long ex = fs_lookback(a.st2 + q, 4, tile, lane, tot);
                if (lane == 0) {
                    S.excl[q] = (uint32_t)ex; // stream offsets inside a batch fit 32 bits (k_fplan checks the totals)
                    if (tile + 1 == a.n_tiles) info->tot[q] = ex + tot;
                }
            }
            __syncthreads();
            FS_STAMP(4);
        }
        if (FS_ABLATE == 2) live = false;
        if (live) {
            // ---- phase D: strip by strip: stream offsets (one lane per record) -> piece-centric split (a wave per <= 64 records)
            const uint32_t qoff = info->qual_off;
            uint32_t carry[4] = {S.excl[0], S.excl[1], S.excl[2], S.excl[3]};
            for (uint32_t i0 = 0; i0 < n_owned; i0 += FS_STRIP) {
                const uint32_t i = i0 + t, r = r_lo + i;
                const bool mine = t < FS_STRIP && i < n_owned;
                uint32_t s0m1 = 0, cpl = 0, fL = 0, fH = 0, fP = 0; // header line start + 1 ... of this lane's record
                if (i0) { // (the first strip's fields were filled in by phase C)
                    __syncthreads(); // the previous strip's table is free
                    if (mine) {
                        const RecF f = rec_fields(i, false);
                        S.s_hdr[t] = f.s0 + 1; S.s_seq[t] = f.s1; S.s_plus[t] = f.s2 + 1; S.s_qual[t] = f.s3;
                        S.L[t] = f.L; S.H[t] = f.H; S.P[t] = f.P; S.nn[t] = f.complete;
                    }
                }
                if (mine) { s0m1 = S.s_hdr[t]; cpl = S.nn[t]; fL = S.L[t]; fH = S.H[t]; fP = S.P[t]; }
                uint32_t v[4] = {0, 0, 0, 0};
                if (cpl) { v[0] = (fL + 3) >> 2; v[1] = fL; v[2] = 2 + fH; v[3] = 2 + fP; }
                uint32_t inc[4];
                __syncthreads(); // (csum is free; every lane has read its nn)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    inc[q] = wave_incl_scan(v[q]);
                    if (lane == 63) S.csum[q][wave] = inc[q];
                }
                __syncthreads();
                uint32_t off[4], stot[4];
                {
                    // per column: bytes of the waves before this one and of the whole strip, from one scan of the 4 x NW table
                    static_assert(4 * FS_NW <= 64, "the (column, wave) table must fit one wave");
                    const uint32_t mine2 = lane < 4 * FS_NW ? (&S.csum[0][0])[lane] : 0u;
                    const uint32_t incl2 = wave_incl_scan(mine2), excl2 = incl2 - mine2;
                    const uint32_t uw = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t seg0 = q ? (uint32_t)__builtin_amdgcn_readlane((int)incl2, q * (int)FS_NW - 1) : 0u;
                        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)excl2, (int)(q * FS_NW + uw)) - seg0;
                        stot[q] = (uint32_t)__builtin_amdgcn_readlane((int)incl2, q * (int)FS_NW + (int)FS_NW - 1) - seg0;
                        off[q] = carry[q] + before + inc[q] - v[q];
                    }
                }
                const uint32_t n_strip = n_owned - i0 < FS_STRIP ? n_owned - i0 : FS_STRIP;
                if (mine) {
                    S.d_seq[t] = off[0]; S.d_qual[t] = off[1]; S.d_hdr[t] = off[2]; S.d_plus[t] = off[3];
                    S.nn[t] = 0;
                    if (r % a.rpb == 0 && r / a.rpb <= a.block_cap) { // the start of a block (also when the record itself is incomplete)
                        FsBStart b;
                        b.text = s0m1 - 1; b.e[0] = off[0]; b.e[1] = off[1]; b.e[2] = off[2]; b.e[3] = off[3];
                        b.pad[0] = b.pad[1] = b.pad[2] = 0;
                        a.bstart[r / a.rpb] = b;
                    }
                    if (cpl && r < a.rec_cap) {
                        a.rec_seq[r] = S.s_seq[t];
                        a.rec_L[r] = fL;
                        *(uint32_t *)(a.reg.p[S_LEN] + 4ull * r) = fL; // lengths stream: u32 L (compress.go:501)
                        // record prefixes: u16 H, u16 P (compress.go:514-519)
                        uint8_t *dh = a.reg.p[S_HDR] + off[2], *dp = a.reg.p[S_PLUS] + off[3];
                        dh[0] = (uint8_t)fH; dh[1] = (uint8_t)(fH >> 8);
                        dp[0] = (uint8_t)fP; dp[1] = (uint8_t)(fP >> 8);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) carry[q] += stot[q];
                __syncthreads();
                if (i0 == 0) FS_STAMP(5);
                // ---- the strip's records, shared out evenly over the waves in runs of <= 64
                const uint32_t per = (n_strip + FS_NW - 1) / FS_NW; // <= 64
                const uint32_t w_lo = wave * per, w_n = w_lo < n_strip ? (n_strip - w_lo < per ? n_strip - w_lo : per) : 0u;
                if (w_n && FS_ABLATE != 3) {
                    const uint32_t me = w_lo + lane; // table row of this lane's record
                    const bool have = lane < w_n && r_lo + i0 + me < a.rec_cap;
                    uint8_t *const o_seq = a.reg.p[S_SEQ], *const o_qual = a.reg.p[S_QUAL], *const o_hdr = a.reg.p[S_HDR], *const o_plus = a.reg.p[S_PLUS];
                    // 16 text bytes at absolute offset src: from the staged tile, or (the last record may leave the tile) from memory
                    auto piece = [&](uint32_t src, uint32_t x[4]) {
                        if (FS_LDS_TEXT && src - tbase + 16 <= TILE) fs_lds_piece(S.text, src - tbase, x);
                        else load_piece(text, src, n, x);
                    };
                    // ---- bases and qualities share the piece map (both lines of a record have L bytes)
                    // bases: 16 bases -> 4 packed bytes (sequence.go:139-184); N counts per record (compress.go:477-488)
                    // quality: q'[0] = q[0]-off, q'[j] = q[j]-q[j-1], restarting per record (quality.go:53-103)
                    // (the record fields are read from the table phase by phase: fewer registers live at once)
                    const uint32_t L = have ? S.L[me] : 0u, s_seq = have ? S.s_seq[me] : 0u, s_qual = have ? S.s_qual[me] : 0u;
                    const uint32_t d_seq = have ? S.d_seq[me] : 0u, d_qual = have ? S.d_qual[me] : 0u;
                    const uint32_t pq = (L + 15) >> 4, iq = wave_incl_scan(pq);
                    const PieceMap pm_iq = piece_map_make(pq, iq);
                    const uint32_t Tq = (uint32_t)__builtin_amdgcn_readlane((int)iq, 63);
                    for (uint32_t pb = 0; pb < Tq; pb += WAVE) {
                        const uint32_t p = pb + lane;
                        const bool on = p < Tq;
                        uint32_t ri, k;
                        piece_locate(pm_iq, iq, pq, on ? p : 0, &ri, &k);
                        const uint32_t Li = (uint32_t)__shfl((int)L, (int)ri, WAVE);
                        const uint32_t src = (uint32_t)__shfl((int)s_seq, (int)ri, WAVE), srcq = (uint32_t)__shfl((int)s_qual, (int)ri, WAVE);
                        const uint32_t dst = (uint32_t)__shfl((int)d_seq, (int)ri, WAVE), dstq = (uint32_t)__shfl((int)d_qual, (int)ri, WAVE);
                        const uint32_t have_b = Li - 16 * k < 16 ? Li - 16 * k : 16;
                        uint32_t x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
                        if (on) { piece(src + 16 * k, x); piece(srcq + 16 * k, y); }
                        // the byte before a quality piece is the last byte of the previous lane's piece (same read, k - 1); only lane 0
                        // has to fetch it from the text
                        const uint32_t left = (uint32_t)__shfl_up((int)(y[3] >> 24), 1, WAVE);
                        if (on) {
                            uint32_t out = 0, nn = 0, beyond = 0;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                uint32_t v4 = x[q], in_read = 0x80808080u;
                                if (have_b < 4u * q + 4) { // bytes past the read pack as 0
                                    const uint32_t hv = have_b > 4u * q ? have_b - 4u * q : 0;
                                    v4 = hv ? v4 & ((1u << (8 * hv)) - 1) : 0;
                                    in_read = hv ? in_read >> (8 * (4 - hv)) : 0;
                                }
                                const uint32_t vmask = acgt_mask(v4);
                                const uint32_t invalid = ~vmask & in_read;
                                out |= pack4(v4, vmask) << (8 * q);
                                if (invalid) {
                                    const uint32_t b0 = 16 * k + 4 * q;
                                    if (b0 + 3 < FQZ_MAX_SEQUENCE_LENGTH) nn += __popc(invalid);
                                    else
                                        for (uint32_t z = 0; z < 4; z++)
                                            if (invalid & (0x80u << (8 * z))) { if (b0 + z < FQZ_MAX_SEQUENCE_LENGTH) nn++; else beyond = 1; }
                                }
                            }
                            uint32_t prev = k ? (lane ? left : TB(srcq + 16 * k - 1)) : qoff;
#pragma unroll
                            for (int q = 0; q < 4; q++) { const uint32_t yq = y[q]; y[q] = sub_bytes(yq, (yq << 8) | (prev & 0xFF)); prev = yq >> 24; }
                            const uint32_t nb = (have_b + 3) >> 2;
                            uint8_t *o = o_seq + dst + 4 * k;
                            if (nb == 4) store_u32_unaligned(o, out);
                            else { // 1..3 packed bytes at the end of a read
                                if (nb & 2) { uint16_t v2 = (uint16_t)out; __builtin_memcpy(o, &v2, 2); }
                                if (nb & 1) o[nb & 2] = (uint8_t)(out >> (8 * (nb & 2)));
                            }
                            if (beyond) report_error(info, r_lo + i0 + w_lo + ri, 4, FQZ_E_LONG_N);
                            if (nn) atomicAdd(&S.nn[w_lo + ri], nn);
                            store_piece(o_qual + dstq + 16 * k, y, have_b);
                        }
                    }
                    // ---- header and plus payloads (without '@' / '+'), after their u16 length
                    const uint32_t H = have ? S.H[me] : 0u, s_hdr = have ? S.s_hdr[me] : 0u, d_hdr = have ? S.d_hdr[me] : 0u;
                    const uint32_t ph = (H + 15) >> 4, ih = wave_incl_scan(ph);
                    const PieceMap pm_ih = piece_map_make(ph, ih);
                    const uint32_t Th = (uint32_t)__builtin_amdgcn_readlane((int)ih, 63);
                    for (uint32_t pb = 0; pb < Th; pb += WAVE) {
                        const uint32_t p = pb + lane;
                        const bool on = p < Th;
                        uint32_t ri, k;
                        piece_locate(pm_ih, ih, ph, on ? p : 0, &ri, &k);
                        const uint32_t Hi = (uint32_t)__shfl((int)H, (int)ri, WAVE), src = (uint32_t)__shfl((int)s_hdr, (int)ri, WAVE);
                        const uint32_t dst = (uint32_t)__shfl((int)d_hdr, (int)ri, WAVE);
                        if (on) {
                            uint32_t x[4];
                            piece(src + 16 * k, x);
                            store_piece(o_hdr + dst + 2 + 16 * k, x, Hi - 16 * k < 16 ? Hi - 16 * k : 16);
                        }
                    }
                    const uint32_t P = have ? S.P[me] : 0u, s_plus = have ? S.s_plus[me] : 0u, d_plus = have ? S.d_plus[me] : 0u;
                    const uint32_t pp = (P + 15) >> 4, ip = wave_incl_scan(pp);
                    const uint32_t Tp = (uint32_t)__builtin_amdgcn_readlane((int)ip, 63);
                    const PieceMap pm_ip = piece_map_make(pp, ip);
                    for (uint32_t pb = 0; pb < Tp; pb += WAVE) {
                        const uint32_t p = pb + lane;
                        const bool on = p < Tp;
                        uint32_t ri, k;
                        piece_locate(pm_ip, ip, pp, on ? p : 0, &ri, &k);
                        const uint32_t Pi = (uint32_t)__shfl((int)P, (int)ri, WAVE), src = (uint32_t)__shfl((int)s_plus, (int)ri, WAVE);
                        const uint32_t dst = (uint32_t)__shfl((int)d_plus, (int)ri, WAVE);
                        if (on) {
                            uint32_t x[4];
                            piece(src + 16 * k, x);
                            store_piece(o_plus + dst + 2 + 16 * k, x, Pi - 16 * k < 16 ? Pi - 16 * k : 16);
                        }
                    }
                }
                __syncthreads();
                if (i0 == 0) FS_STAMP(6);
                // bytes of the record in the nPos stream: u16 count + u16 per N position (compress.go:495-498)
                if (mine && cpl && r < a.rec_cap) a.Enpos[r] = 2 + 2 * S.nn[t];
            }
        }
        // ---- next tile
        // (the ticket is taken only now: a tile whose ticket is held while its holder still works on another one keeps every
        //  tile behind it waiting in its look-backs)
        if (t == 0) S.next_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        tile = S.next_tile;
    }
}
