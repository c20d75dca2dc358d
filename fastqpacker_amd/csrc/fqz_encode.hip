// fqz_encode.hip — device-resident encode pipeline (gfx950 / MI355X).
//
// Replaces, for a batch of blocks at once, the reference's
//   fqparser.nextInto/readLine            internal/fqparser/parser.go:136-243   (k_fsplit, fqz_fsplit.h)
//   encoder.DetectEncoding                internal/encoder/quality.go:22-49     (k_fsplit, mode 1)
//   compressBlockWithBuffers record loop  internal/compress/compress.go:474-520 (k_fsplit, k_npos_write)
//     encoder.AppendPackedBases           internal/encoder/sequence.go:139-184
//     encoder.NormalizeQuality+DeltaEncode internal/encoder/quality.go:53-103
//   6 x zstd.Encoder.EncodeAll            compress.go:523-528                   (k_entropy: Huffman-literal zstd blocks)
//   BlockHeader.Write + payload concat    container.go:97-109, compress.go:532-552 (k_layout, k_compact)
//
// All integer/byte work, HBM-bound: no MFMA.  One launch sequence handles every
// block of the batch; no host round trip until the result struct is read back.
#include "fqz_ctx.h"
#include "fqz_device.h"
#include "fqz_entropy_dev.h"

#include <stdlib.h>
#include <string.h>

// ===========================================================================
// generic in-place exclusive scan of a u32 array (n may live on the device); data[n] receives the total.
// One pass: a workgroup owns 4096 elements (a wave 1024 consecutive ones in 16 coalesced rows, kept in registers),
// the sum of the tiles before it comes from a decoupled look-back over state[] (flag << 62 | value: 1 = tile sum,
// 2 = inclusive prefix).  Tiles are handed out by an atomic ticket, so every predecessor of a running tile is
// running or done and the wait always ends.  state[] (tiles + 1 words, the last one is the ticket) must be zero.
// ===========================================================================
#define SCAN_ITEMS 16
#define SCAN_TILE (256 * SCAN_ITEMS)
#define SC_AGG (1ull << 62)
#define SC_PREFIX (2ull << 62)
#define SC_VALUE ((1ull << 62) - 1)

__device__ __forceinline__ uint32_t scan_n(const uint32_t *n_ptr, uint32_t n_add) { return (n_ptr ? *n_ptr : 0u) + n_add; }

__global__ __launch_bounds__(256) void k_scan(uint32_t *__restrict__ data, const uint32_t *n_ptr, uint32_t n_add, unsigned long long *state, uint32_t n_tiles_cap)
{
    __shared__ uint32_t s_tile, shw[4], s_excl;
    const uint32_t n = scan_n(n_ptr, n_add);
    if ((unsigned long long)blockIdx.x * SCAN_TILE > n) return; // exactly the tiles that hold an index <= n take a ticket
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if (t == 0) s_tile = atomicAdd((uint32_t *)(state + n_tiles_cap), 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t rw = tile * SCAN_TILE + wave * (64 * SCAN_ITEMS);
    uint32_t v[SCAN_ITEMS], tsum = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) {
        const uint32_t i = rw + 64 * j + lane;
        v[j] = i < n ? data[i] : 0u;
        tsum += v[j];
    }
    const uint32_t ws = wave_sum(tsum);
    if (lane == 0) shw[wave] = ws;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { if (k < wave) wbase += shw[k]; tot += shw[k]; }
    if (wave == 0) {
        unsigned long long excl = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&state[tile], SC_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int look = (int)tile - 1;
            for (;;) {
                const int idx = look - (int)lane;
                const unsigned long long sv = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SC_PREFIX;
                const uint32_t flag = (uint32_t)(sv >> 62);
                const unsigned long long pmask = __ballot(flag == 2), zmask = __ballot(flag == 0);
                const int fp = pmask ? __ffsll((long long)pmask) - 1 : 64;          // nearest predecessor with a full prefix
                const unsigned long long need = fp >= 63 ? ~0ull : ((2ull << fp) - 1); // lanes 0..fp must have published
                if (zmask & need) { __builtin_amdgcn_s_sleep(2); continue; }
                unsigned long long part = (int)lane <= fp ? (sv & SC_VALUE) : 0ull;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, WAVE);
                excl += part;
                if (pmask) break;
                look -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&state[tile], SC_PREFIX | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = (uint32_t)excl;
        }
    }
    __syncthreads();
    uint32_t carry = s_excl + wbase;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) {
        const uint32_t i = rw + 64 * j + lane;
        const uint32_t inc = wave_incl_scan(v[j]);
        if (i <= n) data[i] = carry + inc - v[j]; // [n] = total
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
}

// data: n_cap + 1 u32, scanned in place; data[n] receives the total
// state: tiles + 1 zeroed 64-bit words (the batch encoder zeroes all its look-back states in k_init); nullptr: own buffer + memset
static int launch_scan(fqz_ctx *ctx, const char *label, hipStream_t st, uint32_t *data, const uint32_t *n_ptr, uint32_t n_add, uint32_t n_cap,
                       unsigned long long *state = nullptr)
{
    const uint32_t tiles = n_cap / SCAN_TILE + 1;
    if (!state) {
        DevBuf &sb = ctx->enc.scan_state;
        int rc = sb.ensure(8ull * ((size_t)tiles + 1));
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(sb.p, 0, 8ull * ((size_t)tiles + 1), st));
        state = sb.as<unsigned long long>();
    }
    ctx->prof.begin(label, st);
    hipLaunchKernelGGL(k_scan, dim3(tiles), dim3(256), 0, st, data, n_ptr, n_add, state, tiles);
    ctx->prof.end(st);
    return FQZ_OK;
}

// ===========================================================================
// batch state
// ===========================================================================
__global__ __launch_bounds__(256) void k_init(EncInfo *info, int qual_encoding, unsigned long long *zstate, uint32_t zwords)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < zwords; i += gridDim.x * 256) zstate[i] = 0; // look-back states and tickets
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        EncInfo z;
        memset(&z, 0, sizeof z);
        z.error_key = ~0ull;
        z.min_qual = 255;
        z.qual_off = qual_encoding == FQZ_ENCODING_PHRED64 ? 64 : 33;
        z.detect_done_tile = 0xFFFFFFFFu;
        *info = z;
    }
}

// 16 text bytes at offset off (off and text 16-byte aligned); bytes at or beyond n read as 0.  An aligned 16-byte block
// that holds a valid byte never leaves that byte's page, so the block is always loaded whole and masked in registers.
__device__ __forceinline__ uint4 load_text16(const uint8_t *text, uint32_t off, uint32_t n)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (off < n) {
        v = *(const uint4 *)(text + off);
        if (off + 16 > n) {
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            mask_tail16(w, n - off);
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    return v;
}

#include "fqz_fsplit.h"

__global__ void k_finish_detect(EncInfo *info)
{
    if (threadIdx.x || blockIdx.x) return;
    uint32_t mn = info->min_qual;
    // quality.go:36-48: <59 anywhere -> Phred33; none -> Phred33; min>=64 -> Phred64; 59..63 -> Phred33
    info->qual_off = (mn != 255 && mn >= 64) ? 64 : 33;
}

// ===========================================================================
// batch plan: which records are encoded, where every block's streams lie (one 256-thread workgroup)
// ===========================================================================
struct FsCaps { unsigned long long cap[4]; }; // capacities of the seq / qual / headers / plus regions

__global__ __launch_bounds__(256) void k_fplan(EncInfo *info, const FsBStart *bstart, BlockPlan *plans, uint32_t n_bytes, uint32_t rpb, uint32_t final_batch,
                                               uint32_t rec_cap, uint32_t block_cap, FsCaps caps, uint32_t main_cap)
{
    __shared__ uint32_t sh[8], s_stop;
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        const uint32_t n_lines = info->n_lines;
        const uint32_t total = n_lines / 4;                                // whole records present
        uint32_t n_rec = final_batch ? total : (total / rpb) * rpb;       // a batch that is not the last one ends on a block boundary
        uint32_t n_blocks = (n_rec + rpb - 1) / rpb;
        info->n_rec_total = total;
        int32_t status = 0;
        if (info->index_overflow || total > rec_cap || n_blocks > block_cap) status = FQZ_E_TOO_LARGE;
        for (int q = 0; q < 4; q++) if (info->tot[q] > caps.cap[q] || info->tot[q] > 0xFFFFFFF0ull) status = FQZ_E_TOO_LARGE;
        if (!status && info->error_key != ~0ull) {
            // records behind the last whole block of a non-final batch are parsed again with the next batch: not their error yet
            const uint32_t rec = (uint32_t)(info->error_key >> 8);
            if (final_batch || rec < n_rec) { status = -(int32_t)(info->error_key & 31); info->error_record = rec; }
        }
        info->error_key = ~0ull; // later stages report through the same key (k_npos_write)
        if (status) { n_rec = 0; n_blocks = 0; }
        info->status = status;
        info->n_rec = n_rec;
        info->n_blocks = n_blocks;
        info->consumed = final_batch ? n_bytes : (n_rec ? bstart[n_rec / rpb].text : 0u);
        s_stop = status != 0;
    }
    __syncthreads();
    if (s_stop) return;
    const uint32_t n_rec = info->n_rec, n_blocks = info->n_blocks;
    const int order[5] = {S_SEQ, S_QUAL, S_HDR, S_PLUS, S_LEN};
    uint32_t carry_ch = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256) {
        const uint32_t b = b0 + t;
        uint32_t len[5] = {0, 0, 0, 0, 0}, st[4] = {0, 0, 0, 0}, ch = 0, r0 = 0, r1 = 0;
        if (b < n_blocks) {
            r0 = b * rpb;
            r1 = r0 + rpb < n_rec ? r0 + rpb : n_rec;
            const FsBStart bs = bstart[b];
            // the next block's first record exists (its header line at least) unless this block is the short last one
            const bool whole = r1 == r0 + rpb;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                st[q] = bs.e[q];
                const uint32_t en = whole ? bstart[b + 1].e[q] : (uint32_t)info->tot[q];
                len[q] = en - st[q];
                ch += (len[q] + FQZ_CHUNK - 1) / FQZ_CHUNK;
            }
            len[4] = 4 * (r1 - r0);
            ch += (len[4] + FQZ_CHUNK - 1) / FQZ_CHUNK;
        }
        uint32_t tot_c;
        uint32_t ex_c = carry_ch + block_excl_scan_256(ch, sh + 4, &tot_c);
        if (b < n_blocks) {
            BlockPlan *p = &plans[b];
            p->rec0 = r0;
            p->nrec = r1 - r0;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int s = order[q];
                p->len[s] = len[q];
                p->a_off[s] = q < 4 ? st[q] : 4 * r0; // offset inside the stream's region
                p->chunk_base[s] = ex_c;               // main chunk ids: block by block, in this stream order
                ex_c += (len[q] + FQZ_CHUNK - 1) / FQZ_CHUNK;
                atomicAdd(&info->stream_raw[s], (unsigned long long)len[q]);
            }
            p->orig_seq = len[1];
        }
        carry_ch += tot_c;
    }
    if (t == 0) {
        if (carry_ch > main_cap) { info->status = FQZ_E_TOO_LARGE; info->n_blocks = 0; info->n_rec = 0; return; }
        info->n_main = carry_ch;
        info->n_chunks = carry_ch;
    }
}

// nPos stream of every block + its chunks once the N counts are scanned (one 256-thread workgroup)
__global__ __launch_bounds__(256) void k_plan2(EncInfo *info, const uint32_t *Enpos, BlockPlan *plans, uint32_t chunk_cap)
{
    __shared__ uint32_t sh[8];
    const uint32_t t = threadIdx.x;
    const uint32_t n_blocks = info->n_blocks;
    uint32_t carry_ch = info->n_main;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256) {
        const uint32_t b = b0 + t;
        uint32_t len = 0, start = 0;
        if (b < n_blocks) {
            const uint32_t r0 = plans[b].rec0, r1 = r0 + plans[b].nrec;
            start = Enpos[r0];
            len = Enpos[r1] - start;
        }
        uint32_t tot_c;
        const uint32_t ex_c = carry_ch + block_excl_scan_256((len + FQZ_CHUNK - 1) / FQZ_CHUNK, sh + 4, &tot_c);
        if (b < n_blocks) {
            BlockPlan *p = &plans[b];
            p->len[S_NPOS] = len;       // (the region holds the worst case: 2 bytes per record + 2 per base)
            p->a_off[S_NPOS] = start;
            p->chunk_base[S_NPOS] = ex_c; // nPos chunk ids follow all main chunks
            atomicAdd(&info->stream_raw[S_NPOS], (unsigned long long)len);
        }
        carry_ch += tot_c;
    }
    if (t == 0) {
        if (carry_ch > chunk_cap) { info->status = FQZ_E_TOO_LARGE; info->n_blocks = 0; info->n_main = 0; carry_ch = 0; }
        info->n_chunks = carry_ch;
    }
}

#define RL(v, i) __builtin_amdgcn_readlane((int)(v), (i))
// N positions: u16 count + ascending u16 positions per record (compress.go:477-488, 495-498).  Enpos has been scanned:
// Enpos[r] = offset of record r in the nPos region.
__global__ __launch_bounds__(256) void k_npos_write(const uint8_t *text, uint32_t n_text, EncInfo *info, const uint32_t *Enpos, const uint32_t *rec_seq,
                                                    const uint32_t *rec_L, uint8_t *npos_arena)
{
    const uint32_t n_rec = info->n_rec;
    if (info->status) return;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6, lane = lane_id();
    const uint32_t n_groups = (n_rec + 63) >> 6;
    for (uint32_t g = wave; g < n_groups; g += nwaves) {
        const uint32_t r = g * 64 + lane;
        uint32_t s_seq = 0, L = 0, NN = 0, d_npos = 0;
        if (r < n_rec) {
            d_npos = Enpos[r];
            NN = (Enpos[r + 1] - d_npos - 2) >> 1;
            uint8_t *dn = npos_arena + d_npos;
            if (NN > 65535u) { report_error(info, r, 5, FQZ_E_FIELD_WRAP); NN = 0; }
            dn[0] = (uint8_t)NN; dn[1] = (uint8_t)(NN >> 8);
            if (NN) { s_seq = rec_seq[r]; L = rec_L[r]; }
        }
        // Piece-centric like the split: the reads that have N are cut into 16-base pieces, a lane finds the non-ACGT bases of
        // one piece and writes their positions after those of the earlier pieces of the same read (wave scan of the
        // per-piece counts minus its value at the read's first piece; a read that continues from the previous round takes
        // the carried count).
        const uint32_t limit = L < FQZ_MAX_SEQUENCE_LENGTH ? L : FQZ_MAX_SEQUENCE_LENGTH; // positions >= 65536 are not recorded
        const uint32_t pn = NN ? (limit + 15) >> 4 : 0;
        const uint32_t in_ = wave_incl_scan(pn);
        const uint32_t Tn = (uint32_t)RL(in_, 63);
        if (!Tn) continue;
        const PieceMap pm = piece_map_make(pn, in_);
        uint32_t carry = 0;
        for (uint32_t base = 0; base < Tn; base += WAVE) {
            const uint32_t p = base + lane;
            const bool on = p < Tn;
            uint32_t i, k;
            piece_locate(pm, in_, pn, on ? p : 0, &i, &k);
            const uint32_t lim_i = (uint32_t)__shfl((int)limit, (int)i, WAVE), src = (uint32_t)__shfl((int)s_seq, (int)i, WAVE);
            const uint32_t dst = (uint32_t)__shfl((int)d_npos, (int)i, WAVE);
            uint32_t bad16 = 0; // bit b: base 16 k + b is not one of ACGTacgt
            if (on) {
                uint32_t x[4];
                load_piece(text, (size_t)src + 16 * k, n_text, x);
                const uint32_t have = lim_i - 16 * k < 16 ? lim_i - 16 * k : 16;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t inv = ~acgt_mask(x[q]) & 0x80808080u;
                    bad16 |= ((((inv >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * q);
                }
                bad16 &= have >= 16 ? 0xFFFFu : ((1u << have) - 1);
            }
            const uint32_t cntp = __popc(bad16);
            const uint32_t incl = wave_incl_scan(cntp), excl = incl - cntp;
            const uint32_t head_excl = (uint32_t)__shfl((int)excl, (int)(lane >= k ? lane - k : 0), WAVE);
            uint32_t rank = lane >= k ? excl - head_excl : carry + excl;
            if (on) {
                uint8_t *o = npos_arena + dst + 2;
                uint32_t m2 = bad16;
                while (m2) {
                    const uint32_t bpos = 16 * k + (uint32_t)(__ffs(m2) - 1);
                    m2 &= m2 - 1;
                    o[2 * rank] = (uint8_t)bpos;
                    o[2 * rank + 1] = (uint8_t)(bpos >> 8);
                    rank++;
                }
            }
            const uint32_t before_l = lane >= k ? excl - head_excl : carry + excl;
            carry = (uint32_t)RL(before_l + cntp, 63); // only read by lanes whose read started before the next round
        }
    }
}


// ===========================================================================
// K5/K6 entropy stage: one workgroup per 16 KiB chunk -> one zstd block
// (replaces zstd.Encoder.EncodeAll, compress.go:523-528; format: RFC 8878)
// The construction is the deterministic "FQZ-H1" profile specified in DESIGN.md
// and restated on the CPU in oracle/fqz_entropy.c; outputs are byte-identical.
// ===========================================================================
// chunk id -> (block, stream, chunk index inside the stream).  Main chunks are numbered block by block in the
// order seq, qual, headers, plus, lengths; the nPos chunks of all blocks follow (their count is only known
// after the sequence stage has counted the N bases).
__device__ __forceinline__ void locate_chunk(const EncInfo *info, const BlockPlan *plans, uint32_t chunk, uint32_t *bidx, int *stream, uint32_t *cidx)
{
    const uint32_t nb = info->n_blocks;
    const bool is_npos = chunk >= info->n_main;
    const int key = is_npos ? S_NPOS : S_SEQ;
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (plans[mid].chunk_base[key] <= chunk) lo = mid; else hi = mid; }
    const BlockPlan *p = &plans[lo];
    int s = S_NPOS;
    if (!is_npos) {
        const int order[5] = {S_SEQ, S_QUAL, S_HDR, S_PLUS, S_LEN};
        s = S_SEQ;
        for (int q = 0; q < 5; q++) {
            int k = order[q];
            uint32_t nch = (p->len[k] + FQZ_CHUNK - 1) / FQZ_CHUNK;
            if (nch && chunk >= p->chunk_base[k] && chunk < p->chunk_base[k] + nch) s = k;
        }
    }
    *bidx = lo;
    *stream = s;
    *cidx = chunk - p->chunk_base[s];
}

// groups of up to FQZ_GROUP consecutive chunks of one stream share a Huffman table: one thread per chunk finds the
// group leaders and appends a descriptor {first chunk id, arena offset, bytes | last << 24 | stream << 28, 0}
// cinfo[chunk] = block | stream << 24, for k_compact (which would otherwise repeat the search, one dependent load after another)
__global__ __launch_bounds__(256) void k_group_map(EncInfo *info, const BlockPlan *plans, uint4 *gmap, uint32_t group_cap, uint32_t *cinfo)
{
    const uint32_t chunk = blockIdx.x * 256 + threadIdx.x;
    if (chunk >= info->n_chunks) return;
    uint32_t b, c;
    int s;
    locate_chunk(info, plans, chunk, &b, &s, &c);
    cinfo[chunk] = b | ((uint32_t)s << 24);
    if (c % FQZ_GROUP) return;
    const BlockPlan *p = &plans[b];
    const uint32_t off = c * FQZ_CHUNK;
    const uint32_t M = p->len[s] - off < FQZ_GROUP * FQZ_CHUNK ? p->len[s] - off : FQZ_GROUP * FQZ_CHUNK;
    const uint32_t last = off + M == p->len[s];
    const uint32_t g = atomicAdd(&info->n_groups, 1u); // any order: groups are independent
    if (g < group_cap) gmap[g] = make_uint4(chunk, p->a_off[s] + off, M | (last << 24) | ((uint32_t)s << 28), 0u);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_entropy(const EncInfo *info, const uint4 *gmap, const FsRegions reg,
                                                 uint8_t *slots, uint32_t *csize, int dbg_stop, unsigned long long *stamps)
{
    __shared__ __attribute__((aligned(16))) EntropyLds S;
    if (blockIdx.x >= info->n_groups) return;
    const uint4 gd = gmap[blockIdx.x];
    const uint32_t chunk = gd.x, M = gd.z & 0xFFFFFFu, last = (gd.z >> 24) & 1u, s = gd.z >> 28;
    if (stamps) { stamps += (size_t)chunk * 16; if (threadIdx.x == 0) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[15] = (unsigned long long)s; } }
    const uint8_t *src = reg.p[s] + gd.y; // any alignment
    entropy_encode_group(S, src, M, last, slots + (size_t)chunk * FQZ_SLOT, &csize[chunk], dbg_stop, stamps);
}

// ===========================================================================
// K7 framing (container.go:97-109, compress.go:532-552) + compaction
// ===========================================================================
__device__ __forceinline__ void put_le32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

// csize has been scanned in place (exclusive prefix, total at [n_chunks])
// one 256-thread workgroup, one thread per block: block size = 36 + its six frames, offsets by a workgroup scan
__global__ __launch_bounds__(256) void k_layout(EncInfo *info, BlockPlan *plans, const uint32_t *cpre, uint8_t *out, size_t out_cap)
{
    __shared__ uint32_t sh[4];
    if (blockIdx.x) return;
    const uint32_t t = threadIdx.x;
    if (t == 0 && info->error_key != ~0ull && info->status == 0) { // errors raised after the plan kernels (nPos count overflow)
        info->status = -(int32_t)(info->error_key & 31);
        info->error_record = (uint32_t)(info->error_key >> 8);
    }
    __syncthreads();
    const uint32_t n_blocks = info->status ? 0u : info->n_blocks;
    unsigned long long carry = 0; // bytes of the blocks of earlier strips
    unsigned long long comp[FQZ_NS] = {0, 0, 0, 0, 0, 0};
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t b = base + t;
        BlockPlan *p = b < n_blocks ? &plans[b] : nullptr;
        uint32_t flen[FQZ_NS] = {0, 0, 0, 0, 0, 0}, size = 0;
        if (p) {
            size = 36;
            for (int s = 0; s < FQZ_NS; s++) {
                uint32_t nch = (p->len[s] + FQZ_CHUNK - 1) / FQZ_CHUNK;
                flen[s] = nch ? 10 + (cpre[p->chunk_base[s] + nch] - cpre[p->chunk_base[s]]) : 0;
                size += flen[s];
            }
        }
        uint32_t tot;
        const unsigned long long start = carry + block_excl_scan_256(size, sh, &tot);
        carry += tot;
        if (p) {
            unsigned long long pos = start + 36;
            for (int s = 0; s < FQZ_NS; s++) {
                p->frame_off[s] = (uint32_t)pos;
                p->frame_len[s] = flen[s];
                pos += flen[s];
                comp[s] += flen[s];
            }
            p->out_off = (uint32_t)start;
            p->out_len = size;
        }
        __syncthreads();
    }
    for (int s = 0; s < FQZ_NS; s++) if (comp[s]) atomicAdd(&info->stream_comp[s], comp[s]);
    const unsigned long long total = carry; // identical in every thread
    if (t == 0) {
        info->out_len = total;
        if ((total > out_cap || total > 0xFFFFFFF0ull) && !info->status) info->status = FQZ_E_DST_SMALL;
    }
    if (total > out_cap || total > 0xFFFFFFF0ull || info->status) return;
    for (uint32_t b = t; b < n_blocks; b += 256) {
        BlockPlan *p = &plans[b];
        uint8_t *h = out + p->out_off;
        // BlockHeader v2: NumRecords, Seq, Qual, Header, Plus, NPositions, SeqLengths, OriginalSeq, OriginalQual
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
            uint8_t *f = out + p->frame_off[s];
            f[0] = 0x28; f[1] = 0xB5; f[2] = 0x2F; f[3] = 0xFD; // zstd magic
            f[4] = 0x80;                                        // FCS 4 bytes, windowed, no checksum
            f[5] = 0x38;                                        // window 128 KiB
            put_le32(f + 6, p->len[s]);
        }
    }
}

// One workgroup per chunk: slot (16-byte aligned) -> its place in the frame (any alignment).  The destination comes from
// three dependent loads (cinfo -> plan -> scanned sizes), uniform across the workgroup, so they go through the scalar unit;
// the body is copied in 16-byte rows aligned to the destination.
__global__ __launch_bounds__(256) void k_compact(const EncInfo *info, const BlockPlan *plans, const uint8_t *slots, const uint32_t *cpre,
                                                 const uint32_t *cinfo, uint8_t *out)
{
    const uint32_t chunk = blockIdx.x;
    if (info->status || chunk >= info->n_chunks) return;
    const uint32_t t = threadIdx.x;
    const uint32_t ci = cinfo[chunk], c0 = cpre[chunk], c1 = cpre[chunk + 1];
    const BlockPlan *p = &plans[ci & 0xFFFFFFu];
    const uint32_t s = ci >> 24;
    uint8_t *dst = out + p->frame_off[s] + 10 + (c0 - cpre[p->chunk_base[s]]);
    const uint32_t n = c1 - c0;
    const uint8_t *src = slots + (size_t)chunk * FQZ_SLOT;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    const uint32_t rows = (n - head) >> 4;
    for (uint32_t j = t; j < rows; j += 256) *(uint4 *)(dst + head + 16 * j) = load_u128_unaligned(src + head + 16 * j);
    const uint32_t tail0 = head + 16 * rows;
    if (t < n - tail0) dst[tail0 + t] = src[tail0 + t];
}

// ===========================================================================
// host side
// ===========================================================================
static int fqz_dbg_stop()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("FQZ_DBG_STOP"); v = e ? atoi(e) : 0; }
    return v;
}
// FQZ_DBG_STAMPS=1: per-chunk s_memtime stamps (16 x u64 per chunk) for phase timing; dumped by fqz_debug_stamps
static unsigned long long *fqz_dbg_stamps(EncState &e)
{
    static int on = -1;
    if (on < 0) { const char *v = getenv("FQZ_DBG_STAMPS"); on = v ? atoi(v) : 0; }
    if (!on) return nullptr;
    if (e.stamps.ensure((size_t)e.chunk_cap * 16 * 8)) return nullptr;
    (void)hipMemset(e.stamps.p, 0, (size_t)e.chunk_cap * 16 * 8);
    return e.stamps.as<unsigned long long>();
}
int fqz_enc_get_fs_stamps(fqz_ctx *ctx, unsigned long long *out, size_t max_tiles, size_t *n_tiles)
{
    EncState &e = ctx->enc;
    if (!e.fs_stamps.p || e.in_flight) return FQZ_E_ARG;
    size_t n = e.n_tiles < max_tiles ? e.n_tiles : max_tiles;
    HIP_TRY(hipMemcpy(out, e.fs_stamps.p, n * 64, hipMemcpyDeviceToHost));
    *n_tiles = n;
    return FQZ_OK;
}
int fqz_enc_get_stamps(fqz_ctx *ctx, unsigned long long *out, size_t max_chunks, size_t *n_chunks)
{
    EncState &e = ctx->enc;
    const EncInfo *hi = e.h_info.as<EncInfo>();
    if (!hi || !e.stamps.p) return FQZ_E_ARG;
    size_t n = hi->n_chunks < max_chunks ? hi->n_chunks : max_chunks;
    HIP_TRY(hipMemcpy(out, e.stamps.p, n * 16 * 8, hipMemcpyDeviceToHost));
    *n_chunks = n;
    return FQZ_OK;
}
static inline uint32_t grid_for_waves(uint32_t n_items)
{
    // one wave per item, 4 waves per workgroup, capped: the kernels grid-stride
    uint32_t g = (n_items + 3) / 4;
    if (g > 8192) g = 8192;
    return g ? g : 1;
}

int fqz_enc_launch(fqz_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint32_t rpb, int qual_encoding, uint32_t flags, uint8_t *d_out,
                   size_t out_cap, hipStream_t st)
{
    EncState &e = ctx->enc;
    if (e.in_flight) return FQZ_E_ARG;
    if (!rpb) rpb = FQZ_DEFAULT_BLOCK_SIZE;
    if (n_bytes >= 0x7FFFFFFFull) return FQZ_E_TOO_LARGE;
    if (((uintptr_t)d_text & 15) || ((uintptr_t)d_out & 15)) return FQZ_E_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = (uint32_t)n_bytes;
    const uint32_t final_batch = (flags & FQZ_BATCH_FINAL) ? 1u : 0u;

    // ---- capacities (grown lazily; the record capacity assumes >= 32 bytes per record and is retried on overflow)
    const bool small_tiles = e.small_tiles_left > 0; // (the text of the last batches had lines too short for the 32 KiB tiles' newline lists)
    if (small_tiles) e.small_tiles_left--;
    e.small_tiles_now = small_tiles;
    const uint32_t tile_bytes = small_tiles ? 4096u : FS_TILE_BYTES;
    e.n_tiles = (n + tile_bytes - 1) / tile_bytes;
    uint32_t rec_cap = n / 32 + 1024;
    if (e.rec_cap > rec_cap && e.n_bytes == n_bytes) rec_cap = e.rec_cap; // keep a grown capacity on retry
    e.rec_cap = rec_cap;
    e.block_cap = e.rec_cap / rpb + 2;
    FsCaps caps;
    caps.cap[0] = (size_t)n / 8 + e.rec_cap + 4096;   // seq:  sum ceil(L/4)
    caps.cap[1] = (size_t)n / 2 + 4096;               // qual: sum L
    caps.cap[2] = (size_t)n + 4096;                   // headers: sum (2 + H) <= text
    caps.cap[3] = (size_t)n + 4096;                   // plus
    const size_t len_cap = 4ull * e.rec_cap + 4096;
    e.npos_cap = (size_t)n + 2ull * e.rec_cap + 4096;
    const size_t main_bytes = (size_t)n + 4ull * e.rec_cap + 4096;
    const size_t main_cap = main_bytes / FQZ_CHUNK + 5ull * e.block_cap + 8;       // seq / qual / headers / plus / lengths chunks
    const size_t npos_chunk_cap = e.npos_cap / FQZ_CHUNK + 1ull * e.block_cap + 8; // nPos chunks
    const size_t chunk_cap = main_cap + npos_chunk_cap;
    if (chunk_cap > 0x7FFFFFFFull) return FQZ_E_TOO_LARGE;
    e.chunk_cap = (uint32_t)chunk_cap;

    int rc;
    if ((rc = e.info.ensure(sizeof(EncInfo)))) return rc;
    if ((rc = e.Enpos.ensure(4ull * (e.rec_cap + 8)))) return rc;
    if ((rc = e.rec_seq.ensure(4ull * (e.rec_cap + 8)))) return rc;
    if ((rc = e.rec_L.ensure(4ull * (e.rec_cap + 8)))) return rc;
    if ((rc = e.bstart.ensure(sizeof(FsBStart) * ((size_t)e.block_cap + 2)))) return rc;
    if ((rc = e.plans.ensure(sizeof(BlockPlan) * (size_t)e.block_cap))) return rc;
    if ((rc = e.reg[S_SEQ].ensure(caps.cap[0] + 64))) return rc;
    if ((rc = e.reg[S_QUAL].ensure(caps.cap[1] + 64))) return rc;
    if ((rc = e.reg[S_HDR].ensure(caps.cap[2] + 64))) return rc;
    if ((rc = e.reg[S_PLUS].ensure(caps.cap[3] + 64))) return rc;
    if ((rc = e.reg[S_LEN].ensure(len_cap + 64))) return rc;
    if ((rc = e.reg[S_NPOS].ensure(e.npos_cap + 64))) return rc;
    if ((rc = e.slots.ensure((size_t)e.chunk_cap * FQZ_SLOT))) return rc;
    if ((rc = e.csize.ensure(4ull * (2ull * e.chunk_cap + 4)))) return rc; // compressed sizes (scanned in place) | cinfo
    if ((rc = e.h_info.ensure(sizeof(EncInfo)))) return rc;
    if ((rc = e.h_plans.ensure(sizeof(BlockPlan) * (size_t)e.block_cap))) return rc;

    e.streams_valid = true;
    e.d_text = d_text; e.n_bytes = n_bytes; e.rpb = rpb; e.flags = flags; e.d_out = d_out; e.out_cap = out_cap; e.stream = st;

    EncInfo *info = e.info.as<EncInfo>();
    uint32_t *Enpos = e.Enpos.as<uint32_t>();
    uint32_t *csize = e.csize.as<uint32_t>();
    BlockPlan *plans = e.plans.as<BlockPlan>();
    uint8_t *slots = e.slots.as<uint8_t>();
    FsRegions reg;
    for (int s = 0; s < FQZ_NS; s++) reg.p[s] = e.reg[s].as<uint8_t>();

    // one zeroed buffer for every look-back state of the batch:
    // [newline counts | stream sizes x 4 | newline counts of the detect pass | nPos scan | chunk scan | tickets]
    const uint32_t zt_npos = e.rec_cap / SCAN_TILE + 2, zt_chunks = e.chunk_cap / SCAN_TILE + 2;
    const uint32_t zwords = 6 * e.n_tiles + zt_npos + zt_chunks + 2;
    if ((rc = e.zstate.ensure(8ull * zwords))) return rc;
    unsigned long long *st1 = e.zstate.as<unsigned long long>(), *st2 = st1 + e.n_tiles, *st1d = st2 + 4ull * e.n_tiles;
    unsigned long long *z_npos = st1d + e.n_tiles, *z_chunks = z_npos + zt_npos, *tickets = z_chunks + zt_chunks;
    hipLaunchKernelGGL(k_init, dim3((zwords + 255) / 256 < 64 ? (zwords + 255) / 256 : 64), dim3(256), 0, st, info, qual_encoding, st1, zwords);
    HIP_TRY(hipMemsetAsync(e.bstart.p, 0xFF, sizeof(FsBStart) * ((size_t)e.block_cap + 2), st));
    FsArgs fa;
    fa.text = d_text; fa.n = n; fa.n_tiles = e.n_tiles; fa.info = info; fa.st1 = st1; fa.st2 = st2; fa.ticket = (uint32_t *)tickets;
    fa.reg = reg; fa.Enpos = Enpos; fa.rec_seq = e.rec_seq.as<uint32_t>(); fa.rec_L = e.rec_L.as<uint32_t>(); fa.bstart = e.bstart.as<FsBStart>();
    fa.rec_cap = e.rec_cap; fa.block_cap = e.block_cap; fa.rpb = rpb; fa.final_batch = final_batch; fa.mode = 0;
    {
        static int on = -1;
        if (on < 0) { const char *v = getenv("FQZ_DBG_FS_STAMPS"); on = v ? atoi(v) : 0; }
        fa.stamps = nullptr;
        if (on && !e.fs_stamps.ensure((size_t)e.n_tiles * 64 + 64)) {
            HIP_TRY(hipMemsetAsync(e.fs_stamps.p, 0, (size_t)e.n_tiles * 64, st));
            fa.stamps = e.fs_stamps.as<unsigned long long>();
        }
    }
    const uint32_t fs_grid = e.n_tiles < 2048u ? e.n_tiles : 2048u; // persistent workgroups: every one takes tile after tile by ticket
    if (e.n_tiles) {
        if (qual_encoding == FQZ_DETECT_ENCODING) { // encoder.DetectEncoding over block 0 (compress.go:146-154): a pass of its own, the offset is needed by the split
            FsArgs fd = fa;
            fd.mode = 1; fd.st1 = st1d; fd.ticket = (uint32_t *)(tickets + 1);
            if (small_tiles) PROF(ctx, st, "k_detect", hipLaunchKernelGGL(k_fsplit<4096u>, dim3(fs_grid), dim3(FS_NT), 0, st, fd));
            else PROF(ctx, st, "k_detect", hipLaunchKernelGGL(k_fsplit<FS_TILE_BYTES>, dim3(fs_grid), dim3(FS_NT), 0, st, fd));
            hipLaunchKernelGGL(k_finish_detect, dim3(1), dim3(64), 0, st, info);
        }
        if (small_tiles) PROF(ctx, st, "k_fsplit", hipLaunchKernelGGL(k_fsplit<4096u>, dim3(fs_grid), dim3(FS_NT), 0, st, fa));
        else PROF(ctx, st, "k_fsplit", hipLaunchKernelGGL(k_fsplit<FS_TILE_BYTES>, dim3(fs_grid), dim3(FS_NT), 0, st, fa));
    }
    PROF(ctx, st, "k_fplan", hipLaunchKernelGGL(k_fplan, dim3(1), dim3(256), 0, st, info, e.bstart.as<FsBStart>(), plans, n, rpb, final_batch, e.rec_cap, e.block_cap,
                                               caps, (uint32_t)main_cap));
    if ((rc = launch_scan(ctx, "scan_npos", st, Enpos, &info->n_rec, 0, e.rec_cap, z_npos))) return rc;
    PROF(ctx, st, "k_plan2", hipLaunchKernelGGL(k_plan2, dim3(1), dim3(256), 0, st, info, Enpos, plans, e.chunk_cap));
    PROF(ctx, st, "k_npos_write", hipLaunchKernelGGL(k_npos_write, dim3(grid_for_waves((e.rec_cap + 63) / 64)), dim3(256), 0, st, d_text, n, info, Enpos,
                                                    e.rec_seq.as<uint32_t>(), e.rec_L.as<uint32_t>(), reg.p[S_NPOS]));
    const uint32_t group_cap = e.chunk_cap / FQZ_GROUP + FQZ_NS * e.block_cap + 8;
    if ((rc = e.gmap.ensure(16ull * group_cap))) return rc;
    PROF(ctx, st, "k_group_map", hipLaunchKernelGGL(k_group_map, dim3((e.chunk_cap + 255) / 256), dim3(256), 0, st, info, plans, e.gmap.as<uint4>(), group_cap, csize + e.chunk_cap + 2));
    PROF(ctx, st, "k_entropy", hipLaunchKernelGGL(k_entropy, dim3(group_cap), dim3(256), 0, st, info, e.gmap.as<uint4>(), reg, slots, csize, fqz_dbg_stop(), fqz_dbg_stamps(e)));
    if ((rc = launch_scan(ctx, "scan_chunks", st, csize, &info->n_chunks, 0, e.chunk_cap, z_chunks))) return rc;
    PROF(ctx, st, "k_layout", hipLaunchKernelGGL(k_layout, dim3(1), dim3(256), 0, st, info, plans, csize, d_out, out_cap));
    PROF(ctx, st, "k_compact", hipLaunchKernelGGL(k_compact, dim3(e.chunk_cap), dim3(256), 0, st, info, plans, slots, csize, csize + e.chunk_cap + 2, d_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e.h_info.p, info, sizeof(EncInfo), hipMemcpyDeviceToHost, st));
    e.in_flight = true;
    return FQZ_OK;
}

int fqz_enc_finish(fqz_ctx *ctx, fqz_batch_result *res, uint64_t *block_off, uint64_t *block_len, size_t max_blocks)
{
    EncState &e = ctx->enc;
    if (!e.in_flight) return FQZ_E_ARG;
    e.in_flight = false;
    HIP_TRY(hipStreamSynchronize(e.stream));
    const EncInfo *hi = e.h_info.as<EncInfo>();
    if (res) {
        memset(res, 0, sizeof *res);
        res->n_records = hi->n_rec;
        res->n_blocks = hi->n_blocks;
        res->consumed = hi->consumed;
        res->out_len = hi->status ? 0 : hi->out_len;
        res->status = hi->status;
        res->error_record = hi->error_record;
        res->qual_encoding = hi->qual_off == 64 ? FQZ_ENCODING_PHRED64 : FQZ_ENCODING_PHRED33;
        res->n_chunks = hi->n_chunks;
        for (int s = 0; s < FQZ_NS; s++) { res->stream_raw[s] = hi->stream_raw[s]; res->stream_comp[s] = hi->stream_comp[s]; }
    }
    if (hi->status == FQZ_E_TOO_LARGE && hi->index_overflow && !e.small_tiles_now) {
        // a 32 KiB tile with more newlines than its list in LDS holds: this batch is redone, and the next ones are done, with
        // 4 KiB tiles (such inputs come in runs; after 16 launches the large tiles get another try)
        e.small_tiles_left = 16;
        return FQZ_E_TOO_LARGE;
    }
    if (hi->status == FQZ_E_TOO_LARGE && hi->n_rec_total > e.rec_cap) {
        // more records than the optimistic capacity: remember the exact need so that a relaunch fits
        e.rec_cap = hi->n_rec_total + 16;
        return FQZ_E_TOO_LARGE;
    }
    if (hi->status) return hi->status;
    if ((block_off || block_len) && hi->n_blocks) {
        if (hi->n_blocks > max_blocks) return FQZ_E_DST_SMALL;
        HIP_TRY(hipMemcpyAsync(e.h_plans.p, e.plans.p, sizeof(BlockPlan) * (size_t)hi->n_blocks, hipMemcpyDeviceToHost, e.stream));
        HIP_TRY(hipStreamSynchronize(e.stream));
        const BlockPlan *hp = e.h_plans.as<BlockPlan>();
        for (uint32_t b = 0; b < hi->n_blocks; b++) {
            if (block_off) block_off[b] = hp[b].out_off;
            if (block_len) block_len[b] = hp[b].out_len;
        }
    }
    return FQZ_OK;
}

int fqz_enc_get_streams(fqz_ctx *ctx, uint32_t block, uint8_t *streams[6], size_t stream_len[6])
{
    EncState &e = ctx->enc;
    const EncInfo *hi = e.h_info.as<EncInfo>();
    if (!hi || e.in_flight || block >= hi->n_blocks || !e.streams_valid) return FQZ_E_ARG;
    BlockPlan p;
    HIP_TRY(hipMemcpy(&p, e.plans.as<BlockPlan>() + block, sizeof p, hipMemcpyDeviceToHost));
    for (int s = 0; s < FQZ_NS; s++) {
        size_t cap = stream_len[s];
        stream_len[s] = p.len[s];
        if (!streams || !streams[s]) continue;
        if (cap < p.len[s]) return FQZ_E_DST_SMALL;
        const uint8_t *base = e.reg[s].as<uint8_t>() + p.a_off[s];
        if (p.len[s]) HIP_TRY(hipMemcpy(streams[s], base, p.len[s], hipMemcpyDeviceToHost));
    }
    return FQZ_OK;
}

// ---------------------------------------------------------------------------
// entropy stage alone: one stream -> one zstd frame (unit-level parity with the oracle)
// ---------------------------------------------------------------------------
__global__ void k_single_plan(EncInfo *info, BlockPlan *plans, uint32_t n)
{
    if (threadIdx.x || blockIdx.x) return;
    BlockPlan p;
    memset(&p, 0, sizeof p);
    p.nrec = 1;
    p.len[S_SEQ] = n;
    uint32_t chunks = (n + FQZ_CHUNK - 1) / FQZ_CHUNK;
    for (int s = 1; s < FQZ_NS; s++) p.chunk_base[s] = chunks;
    plans[0] = p;
    info->n_blocks = 1;
    info->n_chunks = chunks;
    info->n_main = chunks;
}

__global__ void k_single_layout(EncInfo *info, BlockPlan *plans, const uint32_t *cpre, uint8_t *out, size_t out_cap)
{
    if (threadIdx.x || blockIdx.x) return;
    BlockPlan *p = &plans[0];
    uint32_t flen = 10 + cpre[info->n_chunks];
    p->frame_off[S_SEQ] = 0;
    p->frame_len[S_SEQ] = flen;
    info->out_len = flen;
    if (flen > out_cap) { info->status = FQZ_E_DST_SMALL; return; }
    out[0] = 0x28; out[1] = 0xB5; out[2] = 0x2F; out[3] = 0xFD; out[4] = 0x80; out[5] = 0x38;
    put_le32(out + 6, p->len[S_SEQ]);
}

int fqz_enc_entropy_only(fqz_ctx *ctx, const uint8_t *d_src, size_t n, uint8_t *d_dst, size_t cap, size_t *out_len, hipStream_t st)
{
    EncState &e = ctx->enc;
    if (e.in_flight || n >= 0x7FFFFFFFull) return FQZ_E_ARG;
    uint32_t chunks = (uint32_t)((n + FQZ_CHUNK - 1) / FQZ_CHUNK);
    int rc;
    if ((rc = e.info.ensure(sizeof(EncInfo)))) return rc;
    if ((rc = e.plans.ensure(sizeof(BlockPlan) * 2))) return rc;
    if ((rc = e.slots.ensure((size_t)(chunks + 1) * FQZ_SLOT))) return rc;
    if ((rc = e.csize.ensure(4ull * (2ull * chunks + 4)))) return rc;
    if ((rc = e.h_info.ensure(sizeof(EncInfo)))) return rc;
    EncInfo *info = e.info.as<EncInfo>();
    BlockPlan *plans = e.plans.as<BlockPlan>();
    uint32_t *csize = e.csize.as<uint32_t>();
    hipLaunchKernelGGL(k_init, dim3(1), dim3(256), 0, st, info, 0, (unsigned long long *)nullptr, 0u);
    hipLaunchKernelGGL(k_single_plan, dim3(1), dim3(64), 0, st, info, plans, (uint32_t)n);
    const uint32_t group_cap = chunks / FQZ_GROUP + 8;
    if ((rc = e.gmap.ensure(16ull * group_cap))) return rc;
    hipLaunchKernelGGL(k_group_map, dim3((chunks + 255) / 256), dim3(256), 0, st, info, plans, e.gmap.as<uint4>(), group_cap, csize + chunks + 2);
    FsRegions reg;
    for (int s = 0; s < FQZ_NS; s++) reg.p[s] = const_cast<uint8_t *>(d_src);
    PROF(ctx, st, "k_entropy", hipLaunchKernelGGL(k_entropy, dim3(group_cap), dim3(256), 0, st, info, e.gmap.as<uint4>(), reg, e.slots.as<uint8_t>(), csize, 0, (unsigned long long *)nullptr));
    if ((rc = launch_scan(ctx, "scan_chunks", st, csize, &info->n_chunks, 0, chunks))) return rc;
    hipLaunchKernelGGL(k_single_layout, dim3(1), dim3(64), 0, st, info, plans, csize, d_dst, cap);
    PROF(ctx, st, "k_compact", hipLaunchKernelGGL(k_compact, dim3(chunks), dim3(256), 0, st, info, plans, e.slots.as<uint8_t>(), csize, csize + chunks + 2, d_dst));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e.h_info.p, info, sizeof(EncInfo), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const EncInfo *hi = e.h_info.as<EncInfo>();
    if (hi->status) return hi->status;
    *out_len = hi->out_len;
    return FQZ_OK;
}
