// fqz_entropy_dev.h — device side of the entropy stage: one 256-thread workgroup turns a <= 16 KiB
// chunk of a pre-entropy stream into one zstd block.
//
// Replaces zstd.Encoder.EncodeAll (internal/compress/compress.go:523-528) with the deterministic
// "FQZ-H2" construction of DESIGN.md §4; byte-identical to oracle/fqz_entropy.c.
//
// The kernel is latency bound (serial chains in the table build), so its throughput is set by how many
// workgroups a CU holds, i.e. by the LDS footprint.  The chunk therefore lives in REGISTERS: it is staged
// once through the (not yet used) output buffer to hand every lane the 64 consecutive symbols it will
// encode, then the same LDS becomes the table-build scratch and finally the output block.  18.6 KiB per
// workgroup -> 8 workgroups per CU.
//
// Work split inside the workgroup:
//   all 256 lanes : histogram, rank sort, leaf depths, weights, canonical codes, bit packing
//   wave 0, scalar: the two-queue Huffman merge (n_active-1 dependent steps; weights in VGPR lanes, v_readlane)
//   all four waves: the FSE compression of the weights (speculative chains, see fse_weights_wg)
#pragma once
#include "fqz_device.h"

#define OUT_WORDS ((FQZ_CHUNK + 64) / 4)

struct HufScratch {          // aliases the (not yet used) output staging buffer
    uint32_t cnt[512];       // node weights: leaves 0..n-1 (sorted), internal n..2n-2
    uint16_t parent[512];
    uint8_t l[256];          // code lengths in sorted order
    uint8_t tree[272];       // Huffman_Tree_Description
    uint32_t fse_bits[64];   // FSE bitstream of the weights (dword aligned)
    uint32_t nc_bits[8];     // FSE NCount header bytes
    uint32_t start[2][4];    // true start state of every chunk of the two state chains
    uint32_t fin[2];         // final states (flushed at the end of the stream)
    uint32_t rec[256];       // bits | nbits << 16 emitted at each weight position
    uint8_t endmap[2][4][64];// end state of a chunk for every possible start state
};

#define KEYS_WORD_OFF 1472   // sort keys / sorted keys (2 x 256 words) sit behind HufScratch
#define TRACE_WORD_OFF 2048
static_assert(sizeof(HufScratch) <= KEYS_WORD_OFF * 4, "HufScratch overlaps the sort keys");
static_assert(KEYS_WORD_OFF + 512 <= TRACE_WORD_OFF, "sort keys overlap the FSE trace");

struct EntropyLds {
    uint32_t out[OUT_WORDS];             // chunk staging during the load; HufScratch + sort keys + FSE trace during the
                                         // table build; then the zstd block
    uint32_t ctab[256];                  // byte histogram during the load, then code | nbits << 16
    uint32_t cc[128];                    // canonical-code scratch
    uint32_t misc[64];                   // (32..63: the two halves of a 512-thread segment workgroup, fqz_seg_entropy.h)
    uint8_t nbits[256];
    uint8_t w[256];
#ifdef FQZ_LDS_PAD
    uint32_t pad[FQZ_LDS_PAD]; // occupancy experiments only
#endif
};
__device__ __forceinline__ uint32_t *lds_keys(EntropyLds &S) { return S.out + KEYS_WORD_OFF; }
__device__ __forceinline__ uint32_t *lds_sorted(EntropyLds &S) { return S.out + KEYS_WORD_OFF + 256; }
// FSE state trace [chain 2][chunk 4][step 32][start state 32] (8 KiB)
__device__ __forceinline__ uint8_t *lds_trace(EntropyLds &S) { return (uint8_t *)(S.out + TRACE_WORD_OFF); }

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane)); }

// ---------------------------------------------------------------------------------------------
// FSE compression of the Huffman weights (only for alphabets with more than 128 weights).
// FQZ-H2 does not fit a distribution to the weights of every chunk: it picks one of two fixed normalised
// distributions over the weight values 0..11 (table log 5) by the share of zero weights.  The NCount header and
// the compression tables of both are compile-time constants; what is left per chunk is the encoding itself.
// Mirrors oracle fse_compress_weights().
// ---------------------------------------------------------------------------------------------
#define FSE_W_LOG 5
#define FSE_W_SIZE 32
struct FseFixed {
    uint16_t trans[FSE_W_SIZE * 12]; // [state - 32][weight] = next state | bits to emit << 8  (FSE_encodeSymbol)
    uint8_t init_state[12];          // FSE_initCState2: first state for a given last weight (no output)
    uint8_t nc[8];                   // FSE NCount header
    uint32_t nc_len;
};
constexpr int fse_highbit(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }
constexpr FseFixed make_fse_fixed(int n0, int n1, int n2, int n3, int n4, int n5, int n6, int n7, int n8, int n9, int n10, int n11)
{
    FseFixed f{};
    const int norm[12] = {n0, n1, n2, n3, n4, n5, n6, n7, n8, n9, n10, n11};
    const int table_log = FSE_W_LOG, table_size = FSE_W_SIZE, alphabet = 12;
    { // FSE_writeNCount (all counts are >= 1 here, so there are no zero runs)
        uint32_t bits = 0, outp = 0;
        int bc = 0, remaining = table_size + 1, threshold = table_size, nb = table_log + 1, sym = 0;
        bits += (uint32_t)(table_log - 5) << bc; bc += 4;
        while (sym < alphabet && remaining > 1) {
            int c = norm[sym++];
            int max = (2 * threshold - 1) - remaining;
            remaining -= c;
            c++;
            if (c >= threshold) c += max;
            bits += (uint32_t)c << bc;
            bc += nb;
            bc -= (c < max) ? 1 : 0;
            while (remaining < threshold) { nb--; threshold >>= 1; }
            if (bc > 16) { f.nc[outp++] = (uint8_t)bits; f.nc[outp++] = (uint8_t)(bits >> 8); bits >>= 16; bc -= 16; }
        }
        if (bc > 0) f.nc[outp++] = (uint8_t)bits;
        if (bc > 8) f.nc[outp++] = (uint8_t)(bits >> 8);
        f.nc_len = outp;
    }
    { // FSE_buildCTable
        int cumul[13] = {};
        uint8_t table_symbol[FSE_W_SIZE] = {};
        for (int s = 1; s <= alphabet; s++) cumul[s] = cumul[s - 1] + norm[s - 1];
        const int step = (table_size >> 1) + (table_size >> 3) + 3, mask = table_size - 1;
        int pos = 0;
        for (int s = 0; s < alphabet; s++)
            for (int i = 0; i < norm[s]; i++) { table_symbol[pos] = (uint8_t)s; pos = (pos + step) & mask; }
        uint16_t state_table[FSE_W_SIZE] = {};
        int dnb[12] = {}, dfs[12] = {};
        for (int u = 0; u < table_size; u++) { int sy = table_symbol[u]; state_table[cumul[sy]++] = (uint16_t)(table_size + u); }
        int total = 0;
        for (int s = 0; s < alphabet; s++) {
            if (norm[s] == 1) { dnb[s] = (table_log << 16) - table_size; dfs[s] = total - 1; total++; }
            else {
                const int max_bits_out = table_log - fse_highbit((uint32_t)(norm[s] - 1));
                dnb[s] = (max_bits_out << 16) - (norm[s] << max_bits_out);
                dfs[s] = total - norm[s];
                total += norm[s];
            }
        }
        for (int s = 0; s < alphabet; s++) {
            for (int st = table_size; st < 2 * table_size; st++) {
                const uint32_t nb = ((uint32_t)st + (uint32_t)dnb[s]) >> 16;
                f.trans[(st - table_size) * 12 + s] = (uint16_t)(state_table[(st >> nb) + dfs[s]] | (nb << 8));
            }
            const uint32_t nb_out = (uint32_t)(dnb[s] + (1 << 15)) >> 16;
            const uint32_t value = (nb_out << 16) - (uint32_t)dnb[s];
            f.init_state[s] = (uint8_t)state_table[(value >> nb_out) + dfs[s]];
        }
    }
    return f;
}
__constant__ const FseFixed c_fse_fixed[2] = {
    make_fse_fixed(21, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1), // sparse alphabets: at least half of the weights are 0
    make_fse_fixed(1, 1, 2, 3, 5, 6, 5, 3, 2, 2, 1, 1),  // dense alphabets
};

// The whole workgroup calls this.  The two interleaved state chains (even / odd weight positions, walked from the
// last weight to the first) are inherently sequential, so they are cut into 4 chunks each and every chunk is
// simulated for EVERY possible start state in parallel (2 chains x 4 chunks x 32 states = 256 lanes); the true
// start states are then chained through the end-state maps, every weight position looks up the state it is
// encoded from in the recorded trace, and the bit stream is assembled with a wave scan.
// Returns the compressed size (valid in wave 0): 0 = nothing to code.
#define FSE_WEIGHTS_BARRIERS 5 // workgroup barriers inside (n > 2): what threads beyond the first 256 of a larger workgroup must match
__device__ __forceinline__ uint32_t fse_weights_wg(EntropyLds &S, HufScratch *sc, int n_in, unsigned long long *stamps = nullptr)
{
#define FSE_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    const uint32_t t = threadIdx.x, wave = t >> 6;
    const int lane = (int)(t & 63);
    if (n <= 2) return 0;
    // S.misc[24] NCount bytes
    if (wave == 0) {
        int zeros = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int idx = lane + 64 * j; zeros += (int)__popcll(__ballot(idx < n && S.w[idx] == 0)); }
        const int prof = 2 * zeros >= n ? 0 : 1;
        const FseFixed &F = c_fse_fixed[prof];
        if (lane < 8) ((uint8_t *)sc->nc_bits)[lane] = F.nc[lane];
        if (lane == 0) { S.misc[24] = F.nc_len; S.misc[23] = (uint32_t)prof; }
        // FSE_initCState2 for the last two positions: no output
        if (lane < 2) { const int i = n - 1 - lane; sc->start[i & 1][0] = F.init_state[S.w[i]]; }
    }
    __syncthreads();
    // the transition table of the chosen distribution, into the (by now dead) Huffman node array
    uint16_t *trans = (uint16_t *)sc->cnt;
    if (t < FSE_W_SIZE * 12 / 2) ((uint32_t *)trans)[t] = ((const uint32_t *)c_fse_fixed[S.misc[23]].trans)[t];
    __syncthreads();
    FSE_STAMP(12);
    const int table_log = FSE_W_LOG, table_size = FSE_W_SIZE;
    const int top = n - 3;                      // highest position that emits bits
    const int nq = 128 / table_size;            // chunks per chain: 4
    const int CH = 128 / nq;                    // steps per chunk
    // chain c holds the positions of parity c, walked downwards: step k <-> position top_c - 2k.
    // trace[c][q][k][u] = state after step k of chunk q when the chunk is entered in state u
    uint8_t *trace = lds_trace(S);
    {
        const int c = (int)(t >> 7), q = (int)((t & 127) / (uint32_t)table_size), u = (int)(t & (uint32_t)(table_size - 1));
        const int top_c = ((top & 1) == c) ? top : top - 1;
        uint32_t st = (uint32_t)u; // state - 32
        int p = top_c - 2 * q * CH;
        uint8_t *tr = trace + ((c * 4 + q) * 32) * 32 + u;
        uint32_t sy = p >= 0 ? S.w[p] : 0u; // the weight of the next step is fetched one step ahead: one dependent LDS read per step
        for (int k = 0; k < CH && p >= 0; k++, p -= 2) {
            const uint32_t sy_next = p >= 2 ? S.w[p - 2] : 0u;
            st = (trans[st * 12 + sy] & 0xFFu) - (uint32_t)table_size;
            tr[k * 32] = (uint8_t)(st + (uint32_t)table_size);
            sy = sy_next;
        }
        st += (uint32_t)table_size;
        sc->endmap[c][q][u] = (uint8_t)(st - (uint32_t)table_size);
    }
    __syncthreads();
    FSE_STAMP(13);
    if (t < 2) { // chain the true start states through the end-state maps
        uint32_t st = sc->start[t][0];
        for (int q = 0; q < nq; q++) {
            sc->start[t][q] = st;
            st = (uint32_t)table_size + sc->endmap[t][q][st - (uint32_t)table_size];
        }
        sc->fin[t] = st;
    }
    sc->rec[t] = 0;
    __syncthreads();
    { // every weight position looks up the state it is encoded from and derives its output bits
        const int p = (int)t;
        if (p <= top) {
            const int c = p & 1, top_c = ((top & 1) == c) ? top : top - 1;
            const int kg = (top_c - p) >> 1, q = kg / CH, k = kg - q * CH;
            const uint32_t s0 = sc->start[c][q];
            const uint32_t st = k ? (uint32_t)trace[(((c * 4 + q) * 32) + (k - 1)) * 32 + (int)(s0 - 32u)] : s0;
            const int sy = S.w[p];
            const uint32_t nb = (uint32_t)trans[(st - 32u) * 12 + (uint32_t)sy] >> 8;
            sc->rec[p] = (st & ((1u << nb) - 1)) | (nb << 16);
        }
    }
    __syncthreads();
    FSE_STAMP(14);
    // ---- parallel assembly by wave 0: position p is emitted after every position > p (p <= top), so its bit
    // offset is the number of bits of all higher positions
    uint32_t result = 0;
    if (wave == 0) {
        sc->fse_bits[lane] = 0;
        wave_lds_sync();
        uint32_t above = 0; // bits of the registers j' > j
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            uint32_t r = sc->rec[64 * j + lane];
            uint32_t nbj = r >> 16, bits = r & 0xFFFF;
            uint32_t incl = wave_incl_scan(nbj);
            uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t off = above + (tot - incl);
            if (nbj) {
                uint32_t w = off >> 5, sh = off & 31;
                atomicOr(&sc->fse_bits[w], bits << sh);
                if (sh + nbj > 32) atomicOr(&sc->fse_bits[w + 1], bits >> (32 - sh));
            }
            above += tot;
        }
        // tail: flush state2 (odd positions), then state1 (even), then the end mark
        uint32_t total_bits = above;
        if (lane == 0) {
            unsigned long long tail = (unsigned long long)(sc->fin[1] & (uint32_t)(table_size - 1)) |
                                      ((unsigned long long)(sc->fin[0] & (uint32_t)(table_size - 1)) << table_log) | (1ull << (2 * table_log));
            uint32_t w = total_bits >> 5, sh = total_bits & 31;
            atomicOr(&sc->fse_bits[w], (uint32_t)(tail << sh));
            if (sh + 2 * table_log + 1 > 32) atomicOr(&sc->fse_bits[w + 1], (uint32_t)((tail << sh) >> 32));
        }
        total_bits += 2 * (uint32_t)table_log + 1;
        const uint32_t fse_bytes = (total_bits + 7) >> 3;
        // ---- NCount bytes, then the FSE stream, into the tree description
        const uint32_t nc_bytes = S.misc[24];
        wave_lds_sync();
        uint8_t *dst = sc->tree + 1;
        const uint8_t *nb8 = (const uint8_t *)sc->nc_bits, *fb8 = (const uint8_t *)sc->fse_bits;
        if ((uint32_t)lane < nc_bytes) dst[lane] = nb8[lane];
        for (uint32_t i = (uint32_t)lane; i < fse_bytes; i += 64) dst[nc_bytes + i] = fb8[i];
        result = nc_bytes + fse_bytes;
    }
    return result; // meaningful in wave 0
}

// ---------------------------------------------------------------------------------------------
// One GROUP of up to FQZ_GROUP consecutive 16 KiB chunks of a stream -> one zstd block per chunk, all sharing ONE
// Huffman table built from the histogram of the whole group (the first Compressed block carries the tree, the
// others are treeless).  src is 16-byte aligned global memory holding the M group bytes; chunk k goes to
// slot0 + k * FQZ_SLOT and its size to csize0[k].  All 256 threads call this; they return together.
//
// The kernel is bound by the LDS pipeline (histogram atomics, code-table lookups, bit packing, the table build), so:
//  * phase 1 loads every chunk once for the histogram, phase 2 loads it again (L2-hot) to encode it; the table
//    build in between — a quarter of a chunk's LDS traffic — is paid once per 64 KiB;
//  * a full chunk does not touch LDS on its way in: lane l of wave w owns the 64 consecutive bytes at 4096 w + 64 l
//    and loads them straight from global memory (4 x 128 bit; the four loads of a wave cover the same 32 cache
//    lines, the vector L1 merges them).  Partial chunks (the last one of a stream) have odd stream lengths and go
//    through a staged copy in the not-yet-used output buffer.
//
// Symbol ownership inside a chunk (fixed by the format: 4 Huffman streams of ceil(m/4) bytes when m >= 256, else 1):
// wave w encodes stream w, lane l the `per` consecutive symbols [l*per, (l+1)*per) of it, per <= 64, kept in sym[16].
// ---------------------------------------------------------------------------------------------
// dbg_stop > 0 (FQZ_DBG_STOP, timing experiments only): leave after that phase with dummy 4-byte blocks
// stamps != nullptr (FQZ_DBG_STAMPS, diagnostic runs only): lane 0 records s_memtime at every phase boundary
#define DBG_STOP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime(); \
                         if (dbg_stop == (k)) { if (threadIdx.x < nchunk) csize0[threadIdx.x] = 4; return; } } while (0)
__device__ __forceinline__ uint32_t lds_load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

struct ChunkSyms { // the symbols of this lane for the chunk in flight
    uint32_t sym[16];
    uint32_t cnt, nstreams;
};

// every lane gets its symbols of chunk [src, src + m) (see "symbol ownership"); ends with a barrier
__device__ __forceinline__ void load_chunk_syms(EntropyLds &S, const uint8_t *src, const uint32_t m, ChunkSyms &C)
{
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    C.nstreams = m >= 256 ? 4 : 1;
    if (m == FQZ_CHUNK) {
        const uint4 *p = (const uint4 *)(src + 4096 * wave + 64 * lane);
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        C.sym[0] = a.x; C.sym[1] = a.y; C.sym[2] = a.z; C.sym[3] = a.w; C.sym[4] = b.x; C.sym[5] = b.y; C.sym[6] = b.z; C.sym[7] = b.w;
        C.sym[8] = c.x; C.sym[9] = c.y; C.sym[10] = c.z; C.sym[11] = c.w; C.sym[12] = d.x; C.sym[13] = d.y; C.sym[14] = d.z; C.sym[15] = d.w;
        C.cnt = 64;
        __syncthreads();
        return;
    }
    const uint32_t seg = C.nstreams == 4 ? (m + 3) / 4 : m;
    const uint32_t seg_base = wave * seg;
    uint32_t seg_len = 0;
    if (wave < C.nstreams) seg_len = (wave == C.nstreams - 1) ? m - seg_base : seg;
    const uint32_t per = ((seg_len + 63) / 64 + 3) & ~3u; // symbols per lane, a multiple of 4, <= 64
    uint32_t sym_a = lane * per, sym_b = sym_a + per;
    if (sym_a > seg_len) sym_a = seg_len;
    if (sym_b > seg_len) sym_b = seg_len;
    C.cnt = sym_b - sym_a;
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { // all four loads in flight before the first use; bytes at or beyond m read as 0
        const uint32_t off = (t + 256 * q) * 16;
        const uint32_t have = off < m ? (m - off < 16 ? m - off : 16) : 0;
        v[q] = make_uint4(0, 0, 0, 0);
        if (have == 16) v[q] = *(const uint4 *)(src + off);
        else if (have) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < have; k++) w[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
            v[q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    if (t < OUT_WORDS - FQZ_CHUNK / 4) S.out[FQZ_CHUNK / 4 + t] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) *(uint4 *)&S.out[(t + 256 * q) * 4] = v[q];
    __syncthreads();
    const uint8_t *mine = (const uint8_t *)S.out + seg_base + sym_a;
#pragma unroll
    for (int d = 0; d < 16; d++) C.sym[d] = 4u * d < C.cnt ? lds_load_u32_unaligned(mine + 4 * d) : 0u;
    __syncthreads(); // the staged copy may be overwritten
}

// Every group is a zstd frame of its own (FQZ-H2): its last chunk carries the Last_Block bit.  force_raw: the 2-bit packed
// bases are Raw blocks by definition - no histogram, no table, no bit counting.
//
// HDR (headers stream, fqz_hdrlz.h): chunk k has been modelled into H->nseq[k] sequences and H->n_lit[k] literals at
// H->lit[k]; the table is built over the literals, the block carries the Huffman-coded literals followed by the
// Sequences_Section.  That section comes out of a serial chain (the FSE states) that runs beside this kernel: the block is
// written here WITHOUT it (its size counts with the bound HDR_SSZ_BOUND when the block is judged against the Raw block - the
// FQZ-H2 rule, oracle encode_group_chunks) and k_hdr_patch appends it and completes the block header.  A chunk without
// sequences has lit = the chunk itself.
struct HdrGroup {
    const uint8_t *lit[FQZ_GROUP];
    const uint16_t *hist[FQZ_GROUP]; // byte histogram of the chunk's literals (k_hdr_model counted them while it compacted them)
    uint32_t n_lit[FQZ_GROUP], nseq[FQZ_GROUP];
};
#define HDR_SSZ_BOUND(n) (((n) < 128u ? 2u : 3u) + ((n) * 66u + 18u + 7u) / 8u)

// xh (the segment path, fqz_seg.h): the byte histogram of every chunk of the group, [chunk][256] counts, made while the part was
// written - phase 1 then has nothing to read
template <bool HDR>
__device__ __forceinline__ void entropy_encode_group(EntropyLds &S, const uint8_t *src, const uint32_t M, const uint32_t force_raw, uint8_t *slot0, uint32_t *csize0,
                                     const int dbg_stop = 0, unsigned long long *stamps = nullptr, const HdrGroup *H = nullptr, const uint32_t *xh = nullptr)
{
    const uint32_t last = 1;
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const uint32_t nchunk = (M + FQZ_CHUNK - 1) / FQZ_CHUNK;
    uint32_t m = M; // the table-build code below speaks of "m": the byte count the histogram covers
    if (HDR) { m = 0; for (uint32_t k = 0; k < nchunk; k++) m += H->n_lit[k]; }
    // ---- phase 1: histogram of the whole group; per chunk, whether all its bytes are equal (RLE block).
    //      Skewed data (quality deltas are ~90 % zeros) would serialise LDS atomics on one bin, so every wave first peels
    //      off its dominant byte: the candidate is the first byte the wave sees, matches are counted with SWAR compares
    //      in registers and added once per wave.
    S.ctab[t] = 0;
    __syncthreads(); // (the first chunk's counts must not arrive before every bin is cleared: full chunks are loaded without a barrier)
    uint32_t same_mask = 0; // bit k: chunk k is one repeated byte
    if (HDR) { // the histograms come with the literals: no pass over them here
        uint32_t c = 0;
        for (uint32_t k = 0; k < nchunk; k++) {
            const uint32_t h = H->hist[k][t];
            c += h;
            if (__syncthreads_count(h != 0) == 1 && !H->nseq[k]) same_mask |= 1u << k;
        }
        S.ctab[t] = c;
    }
    if (!HDR && xh) { // a chunk whose histogram has one bin with all its bytes is one repeated byte
        uint32_t c = 0;
        for (uint32_t k = 0; k < nchunk; k++) {
            const uint32_t mk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK, h = xh[256 * k + t];
            c += h;
            const unsigned long long full = __ballot(h == mk);
            __syncthreads();
            if ((t & 63) == 0) S.misc[8 + (t >> 6)] = full ? 1u : 0u;
            __syncthreads();
            if (S.misc[8] | S.misc[9] | S.misc[10] | S.misc[11]) same_mask |= 1u << k;
        }
        S.ctab[t] = c;
        __syncthreads();
    }
#pragma clang loop unroll(disable)
    for (uint32_t k = 0; k < ((force_raw || HDR || xh) ? 0u : nchunk); k++) {
        const uint32_t mk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK;
        const uint8_t *csrc = HDR ? H->lit[k] : src + (size_t)k * FQZ_CHUNK;
        ChunkSyms C;
        if (!HDR && mk == FQZ_CHUNK) {
            // a full chunk: the histogram does not care which lane holds which bytes, so the loads are the coalesced kind (a wave
            // reads 1 KiB in a row) and need no barrier; the layout the coder wants (64 consecutive bytes a lane) is loaded in phase 2
            const uint4 *p4 = (const uint4 *)csrc;
            const uint4 a = p4[t], b = p4[t + 256], c = p4[t + 512], d = p4[t + 768];
            C.sym[0] = a.x; C.sym[1] = a.y; C.sym[2] = a.z; C.sym[3] = a.w; C.sym[4] = b.x; C.sym[5] = b.y; C.sym[6] = b.z; C.sym[7] = b.w;
            C.sym[8] = c.x; C.sym[9] = c.y; C.sym[10] = c.z; C.sym[11] = c.w; C.sym[12] = d.x; C.sym[13] = d.y; C.sym[14] = d.z; C.sym[15] = d.w;
            C.cnt = 64; C.nstreams = 4;
        } else load_chunk_syms(S, csrc, HDR ? H->n_lit[k] : mk, C);
        const uint32_t b0 = (uint32_t)csrc[0] * 0x01010101u;
        const uint32_t cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)(C.sym[0] & 0xFF)); // wave-uniform candidate byte
        const uint32_t cand4 = cand * 0x01010101u;
        uint32_t n_cand = 0, differs = 0;
        const bool full_chunk = !HDR && mk == FQZ_CHUNK; // (every lane holds 64 bytes: no validity masks to work out - a fifth of the pass's instructions)
        if (full_chunk) {
#pragma unroll
            for (int d = 0; d < 16; d++) {
                const uint32_t eq = zero_bytes(C.sym[d] ^ cand4);
                differs |= ~zero_bytes(C.sym[d] ^ b0) & 0x80808080u;
                n_cand += __popc(eq);
                uint32_t other = 0x80808080u & ~eq; // 0x80 per byte that still needs an atomic
                while (other) {
                    int bit = __ffs(other) - 1; // 7, 15, 23 or 31
                    other &= other - 1;
                    atomicAdd(&S.ctab[(C.sym[d] >> (bit - 7)) & 0xFF], 1u);
                }
            }
        } else {
#pragma unroll
        for (int d = 0; d < 16; d++) {
            const uint32_t valid = C.cnt >= 4u * d + 4 ? 0x80808080u : (C.cnt > 4u * d ? (0x80808080u >> (8 * (4 * d + 4 - C.cnt))) : 0u);
            const uint32_t eq = zero_bytes(C.sym[d] ^ cand4) & valid;
            differs |= ~zero_bytes(C.sym[d] ^ b0) & valid;
            n_cand += __popc(eq);
            uint32_t other = valid & ~eq; // 0x80 per byte that still needs an atomic
            while (other) {
                int bit = __ffs(other) - 1; // 7, 15, 23 or 31
                other &= other - 1;
                atomicAdd(&S.ctab[(C.sym[d] >> (bit - 7)) & 0xFF], 1u);
            }
        }
        }
        n_cand = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(n_cand), 63); // (DPP: no trip through the LDS crossbar)
        if (lane == 0 && n_cand) atomicAdd(&S.ctab[cand], n_cand);
        if (!__syncthreads_or(differs != 0) && !(HDR && H->nseq[k])) same_mask |= 1u << k;
    }
    DBG_STOP(1);
    uint32_t *const keys = lds_keys(S), *const sorted = lds_sorted(S);
    // S.misc: 4 n_active, 5 mode (0 raw, 1 rle, 2 huffman), 6 tree size, 7 max bits, 8..11 per-wave scratch / stream bits,
    //         12 total bytes, 13 max count, 14 tree offset, 16..19 per-wave scratch, 20 max depth, 21 nw
    // ---- merge histograms, classify -------------------------------------------------
    {
        uint32_t c = S.ctab[t];
        keys[t] = c ? ((c << 8) | t) : 0u;
        unsigned long long act = __ballot(c != 0);
        uint32_t sq = wave_sum(c * c); // c <= 16384, the sum of squares <= 2^28
        if (lane == 0) { S.misc[8 + wave] = (uint32_t)__popcll(act); S.misc[16 + wave] = sq; }
    }
    __syncthreads();
    const uint32_t n_active = S.misc[8] + S.misc[9] + S.misc[10] + S.misc[11];
    uint32_t mode = 2;
    if (force_raw) mode = 0;
    else {
        unsigned long long sq = (unsigned long long)S.misc[16] + S.misc[17] + S.misc[18] + S.misc[19];
        if (n_active == 1) mode = 1;                                          // RLE block
        else if (m < 64) mode = 0;                                            // raw
        else if (sq * 230ull <= (unsigned long long)m * m) mode = 0;          // collision entropy >= log2(230) = 7.85 bits: nothing to gain
    }
    __syncthreads(); // misc[8..19] are reused below

    HufScratch *sc = (HufScratch *)S.out;
    uint32_t tree_size = 0, max_bits = 0;
    DBG_STOP(2);
    if (mode == 2) {
        // ---- sort the active symbols by (count, symbol): compact the non-zero keys, then every active key
        //      counts the smaller ones (keys are distinct)
        {
            uint32_t my = keys[t];
            unsigned long long bm = __ballot(my != 0);
            if (lane == 0) S.misc[16 + wave] = (uint32_t)__popcll(bm);
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t w2 = 0; w2 < wave; w2++) base += S.misc[16 + w2];
            if (my) sorted[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1))] = my; // compacted, unsorted
            __syncthreads();
            uint32_t mine = t < n_active ? sorted[t] : 0, rank = 0;
            for (uint32_t j = 0; j < n_active; j++) rank += sorted[j] < mine;
            __syncthreads();
            if (t < n_active) sorted[256 - n_active + rank] = mine;
        }
        __syncthreads();
        DBG_STOP(3);
        const uint32_t n = n_active;
        const uint32_t *key = sorted + (256 - n);
        // ---- two-queue Huffman merge, leaf preferred on ties: n - 1 dependent steps.  Up to 128 active symbols
        //      (always, for FASTQ streams) wave 0 runs it on the scalar unit: the sorted leaf weights and the internal
        //      node weights sit in VGPR lanes (node i in lane i % 64 of register i / 64) and are fetched with
        //      v_readlane into SGPRs, all queue state is wave-uniform; only the parent links go to LDS.
        if (n <= 128) {
            if (wave == 0) {
                const uint32_t INF = 0xFFFFFFFFu;
                const int nn = __builtin_amdgcn_readfirstlane((int)n);
                const uint32_t lw0 = lane < n ? key[lane] >> 8 : INF, lw1 = lane + 64 < n ? key[lane + 64] >> 8 : INF;
                uint32_t iw0 = INF, iw1 = INF;
#define HQ_FETCH(v0, v1, idx, lim) ((idx) < (lim) ? (uint32_t)(((idx) & 64) ? rl((int)(v1), (idx) & 63) : rl((int)(v0), (idx) & 63)) : INF)
                int li = 0, ih = 0, it = 0; // leaves taken, internal nodes taken / made (internal node j has id nn + j)
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
                    iw0 = ((int)lane == (it & 63) && it < 64) ? sum : iw0; // v_cndmask: no branch
                    iw1 = ((int)lane == (it & 63) && it >= 64) ? sum : iw1;
                    if (ih == it) ci = sum; // the new node is the head of an empty internal queue
                    if (lane == 0) { sc->parent[a] = (uint16_t)(nn + it); sc->parent[b] = (uint16_t)(nn + it); }
                    it++;
                }
#undef HQ_FETCH
            }
        } else {
            if (t < n) sc->cnt[t] = key[t] >> 8;
            __syncthreads();
            if (t == 0) {
                uint32_t li = 0, ih = n, it = n;
                uint32_t cl = sc->cnt[0], ci = 0; // head weights of the leaf and internal queues
                for (uint32_t k = 0; k + 1 < n; k++) {
                    uint32_t a, b, ca, cb2;
                    if (li < n && (ih >= it || cl <= ci)) { a = li++; ca = cl; cl = li < n ? sc->cnt[li] : 0; }
                    else { a = ih++; ca = ci; ci = ih < it ? sc->cnt[ih] : 0; }
                    if (li < n && (ih >= it || cl <= ci)) { b = li++; cb2 = cl; cl = li < n ? sc->cnt[li] : 0; }
                    else { b = ih++; cb2 = ci; ci = ih < it ? sc->cnt[ih] : 0; }
                    sc->cnt[it] = ca + cb2;
                    if (ih == it) ci = ca + cb2; // the new node is the head of an empty internal queue
                    sc->parent[a] = (uint16_t)it;
                    sc->parent[b] = (uint16_t)it;
                    it++;
                }
            }
        }
        __syncthreads();
        // ---- leaf depths: every leaf walks to the root
        {
            uint32_t d = 0;
            if (t < n) {
                uint32_t v = t, root = 2 * n - 2;
                while (v != root) { v = sc->parent[v]; d++; }
                sc->l[t] = (uint8_t)d;
            }
            uint32_t mx = d;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) { uint32_t o = __shfl_xor(mx, s, WAVE); mx = o > mx ? o : mx; }
            if (lane == 0) S.misc[16 + wave] = mx;
        }
        __syncthreads();
        uint32_t maxd = max(max(S.misc[16], S.misc[17]), max(S.misc[18], S.misc[19]));
        if (maxd > FQZ_HUF_MAX_BITS) {
            // ---- length limiting (rare): clamp, repair the Kraft sum in units of 2^-11 (one lane)
            if (t == 0) {
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
        DBG_STOP(4);
        // ---- lengths back to symbol order, weights, highest symbol
        S.nbits[t] = 0;
        __syncthreads();
        if (t < n) S.nbits[key[t] & 0xFF] = sc->l[t];
        __syncthreads();
        uint32_t nb = S.nbits[t];
        DBG_STOP(10);
        {
            unsigned long long bm = __ballot(nb != 0);
            if (lane == 0) S.misc[16 + wave] = bm ? wave * 64 + (63 - (uint32_t)__clzll((long long)bm)) : 0;
        }
        __syncthreads();
        const uint32_t nw = max(max(S.misc[16], S.misc[17]), max(S.misc[18], S.misc[19])); // weights for symbols 0..nw-1
        S.w[t] = (t < nw && nb) ? (uint8_t)(maxd + 1 - nb) : 0;
        __syncthreads();
        DBG_STOP(11);
        // ---- Huffman_Tree_Description: direct 4-bit weights when they fit, FSE-compressed otherwise
        if (nw <= 128) {
            if (t == 0) sc->tree[0] = (uint8_t)(128 + (nw - 1));
            if (2 * t < nw) sc->tree[1 + t] = (uint8_t)((S.w[2 * t] << 4) + S.w[2 * t + 1]);
            tree_size = (nw + 1) / 2 + 1;
        } else {
            uint32_t h = fse_weights_wg(S, sc, (int)nw, stamps); // all four waves; the size is valid in wave 0
            if (t == 0) {
                uint32_t ts = 0;
                if (h > 1 && h < 128) { sc->tree[0] = (uint8_t)h; ts = h + 1; } // header byte < 128 = size of the FSE-compressed weights
                S.misc[6] = ts;
            }
            __syncthreads();
            tree_size = S.misc[6];
            if (!tree_size) mode = 0; // not representable: raw block
        }
    }
    DBG_STOP(5);
    if (mode == 2) {
        // ---- canonical codes (RFC 8878 4.2.1.3): from the longest length up, symbol order inside a length
        {
            uint32_t nb = S.nbits[t];
            uint32_t my_rank = 0;
            for (uint32_t len = 1; len <= max_bits; len++) {
                unsigned long long bm = __ballot(nb == len);
                if (nb == len) my_rank = (uint32_t)__popcll(bm & ((1ull << lane) - 1));
                if (lane == 0) S.cc[32 + wave * 16 + len] = (uint32_t)__popcll(bm);
            }
            __syncthreads();
            if (t == 0) {
                uint32_t minv = 0;
                for (uint32_t len = max_bits; len > 0; len--) {
                    uint32_t cnt = S.cc[32 + len] + S.cc[48 + len] + S.cc[64 + len] + S.cc[80 + len];
                    S.cc[len] = minv;
                    minv = (minv + cnt) >> 1;
                }
            }
            __syncthreads();
            uint32_t code = 0;
            if (nb) {
                uint32_t before = 0;
                for (uint32_t w2 = 0; w2 < wave; w2++) before += S.cc[32 + w2 * 16 + nb];
                code = S.cc[nb] + before + my_rank;
            }
            S.ctab[t] = code | (nb << 16);
        }
    }
    // the tree description stays in a register: the output buffer is reused by every chunk below
    const uint8_t tree_byte = (mode == 2 && t < tree_size) ? sc->tree[t] : 0;
    __syncthreads();
    DBG_STOP(6);
    // ---- phase 2: one zstd block per chunk
    uint32_t tree_sent = 0;
#pragma clang loop unroll(disable)
    for (uint32_t k = 0; k < nchunk; k++) {
        const uint32_t mk = M - k * FQZ_CHUNK < FQZ_CHUNK ? M - k * FQZ_CHUNK : FQZ_CHUNK;
        const uint8_t *csrc = src + (size_t)k * FQZ_CHUNK;
        uint8_t *slot = slot0 + (size_t)k * FQZ_SLOT;
        const uint32_t lastblk = (last && k + 1 == nchunk) ? 1u : 0u;
        // the block's entry in the FQZI index (k_compact copies it from here): the bit positions at which a decoder may enter each
        // of the four Huffman streams (at the symbols lanes 16, 32 and 48 start with); zeros for any other kind of block
        uint16_t *const ent = (uint16_t *)(slot + FQZ_SLOT_ENT);
        if (same_mask & (1u << k)) { // RLE block: 3-byte header + the byte
            if (t < FQZ_ENT) ent[t] = 0;
            if (t == 0) {
                const uint32_t bh = lastblk | (1u << 1) | (mk << 3);
                *(uint32_t *)slot = (bh & 0xFFFFFF) | ((uint32_t)csrc[0] << 24);
                csize0[k] = 4;
            }
            continue;
        }
        uint32_t cmode = mode; // 2 = Huffman with the group table, anything else = raw
        const uint32_t nseq = HDR ? H->nseq[k] : 0u;
        const uint32_t ml = HDR ? H->n_lit[k] : mk;       // literals of the block (== the chunk when there are no sequences)
        const uint32_t sec_sz = nseq ? HDR_SSZ_BOUND(nseq) : 1u; // Sequences_Section bytes as judged (Number_of_Sequences = 0: one byte)
        if (cmode == 2) {
            ChunkSyms C;
            load_chunk_syms(S, HDR ? H->lit[k] : csrc, ml, C);
            for (uint32_t i = t; i < OUT_WORDS; i += 256) S.out[i] = 0;
            // ---- pass 1: bits per lane, per stream (wave w encodes stream w)
            uint32_t my_bits = 0;
            if (ml == FQZ_CHUNK) { // (a full chunk - uniform for the workgroup: every lane holds 64 symbols, no per-dword tests)
#pragma unroll
                for (int d = 0; d < 16; d++)
                    my_bits += (S.ctab[C.sym[d] & 0xFF] >> 16) + (S.ctab[(C.sym[d] >> 8) & 0xFF] >> 16) + (S.ctab[(C.sym[d] >> 16) & 0xFF] >> 16) + (S.ctab[C.sym[d] >> 24] >> 16);
            } else {
#pragma unroll
            for (int d = 0; d < 16; d++) {
                if (4u * d + 4 <= C.cnt) {
                    my_bits += (S.ctab[C.sym[d] & 0xFF] >> 16) + (S.ctab[(C.sym[d] >> 8) & 0xFF] >> 16) + (S.ctab[(C.sym[d] >> 16) & 0xFF] >> 16) +
                               (S.ctab[C.sym[d] >> 24] >> 16);
                } else if (4u * d < C.cnt) {
                    for (uint32_t z = 0; z < C.cnt - 4u * d; z++) my_bits += S.ctab[(C.sym[d] >> (8 * z)) & 0xFF] >> 16;
                }
            }
            }
            const uint32_t incl = wave_incl_scan(my_bits);
            const uint32_t tot_bits = __shfl(incl, 63, WAVE);
            const uint32_t bit_off = tot_bits - incl; // bits of all higher lanes = symbols written before mine
            const uint32_t my_ent = bit_off + my_bits; // bits of my symbols and all behind them: where a decoder enters the stream at my first symbol
            if (lane == 0) S.misc[8 + wave] = tot_bits;
            __syncthreads();
            // ---- sizes, raw fallback, headers (one lane; before any atomicOr touches those words)
            const uint32_t tsz = tree_sent ? 0u : tree_size;
            if (t == 0) {
                const uint32_t nstreams = C.nstreams;
                uint32_t ssz[4] = {0, 0, 0, 0}, total_streams = 0;
                for (uint32_t q = 0; q < nstreams; q++) { ssz[q] = (S.misc[8 + q] >> 3) + 1; total_streams += ssz[q]; }
                const uint32_t lit_csize = tsz + (nstreams == 4 ? 6 : 0) + total_streams;
                const uint32_t lh = ml < 1024 ? 3 : (ml < 16384 ? 4 : 5);
                const uint32_t content = lh + lit_csize + sec_sz;
                if (content >= mk) S.misc[5] = 0;
                else {
                    S.misc[5] = 2;
                    uint8_t *o = (uint8_t *)S.out;
                    const uint32_t bh = lastblk | (2u << 1) | (content << 3);
                    const uint32_t lt = tree_sent ? 3u : 2u; // treeless once the group's table has been sent
                    o[0] = (uint8_t)bh; o[1] = (uint8_t)(bh >> 8); o[2] = (uint8_t)(bh >> 16);
                    if (lh == 3) {
                        uint32_t v = lt | ((nstreams == 4 ? 1u : 0u) << 2) | (ml << 4) | (lit_csize << 14);
                        o[3] = (uint8_t)v; o[4] = (uint8_t)(v >> 8); o[5] = (uint8_t)(v >> 16);
                    } else if (lh == 4) {
                        uint32_t v = lt | (2u << 2) | (ml << 4) | (lit_csize << 18);
                        o[3] = (uint8_t)v; o[4] = (uint8_t)(v >> 8); o[5] = (uint8_t)(v >> 16); o[6] = (uint8_t)(v >> 24);
                    } else {
                        uint32_t v = lt | (3u << 2) | (ml << 4) | (lit_csize << 22);
                        o[3] = (uint8_t)v; o[4] = (uint8_t)(v >> 8); o[5] = (uint8_t)(v >> 16); o[6] = (uint8_t)(v >> 24);
                        o[7] = (uint8_t)(lit_csize >> 10);
                    }
                    uint32_t pos = 3 + lh + tsz;
                    if (nstreams == 4) {
                        for (int q = 0; q < 3; q++) { o[pos + 2 * q] = (uint8_t)ssz[q]; o[pos + 2 * q + 1] = (uint8_t)(ssz[q] >> 8); }
                        pos += 6;
                    }
                    for (uint32_t q = 0; q < 4; q++) { S.misc[8 + q] = pos; pos += ssz[q]; } // stream start bytes
                    if (!nseq) o[pos] = 0;                                                     // Number_of_Sequences = 0
                    S.misc[12] = pos + (nseq ? 0u : 1u);                                       // (sequences: appended by k_hdr_patch)
                    S.misc[14] = 3 + lh; // tree offset
                }
            }
            __syncthreads();
            cmode = S.misc[5];
            if (cmode == 2) {
                uint8_t *o = (uint8_t *)S.out;
                if (C.nstreams == 4) { if (lane && !(lane & 15)) ent[3 * wave + (lane >> 4) - 1] = (uint16_t)my_ent; }
                else if (t < FQZ_ENT) ent[t] = 0;
                if (t < tsz) o[S.misc[14] + t] = tree_byte;
                __syncthreads();
                // ---- pass 2: symbols last-to-first, LSB-first bit packing (HUF_compress1X order)
                if (wave < C.nstreams) {
                    uint32_t P0 = 8 * S.misc[8 + wave] + bit_off;
                    uint32_t word = P0 >> 5;
                    uint32_t fill = P0 & 31;
                    unsigned long long acc = 0;
                    // two symbols per step (<= 22 new bits on top of < 32 pending fit the 64-bit accumulator)
#define PUT2(s1, s2) do { uint32_t e1 = S.ctab[(s1)], e2 = S.ctab[(s2)];                         \
                          acc |= (unsigned long long)(e1 & 0xFFFF) << fill; fill += e1 >> 16;  \
                          acc |= (unsigned long long)(e2 & 0xFFFF) << fill; fill += e2 >> 16;  \
                          uint32_t adv = fill >> 5; /* 0 or 1 whole words completed */         \
                          if (adv) atomicOr(&S.out[word], (uint32_t)acc); /* (its first bits may be a neighbour's) */ \
                          acc >>= (adv << 5); word += adv; fill &= 31; } while (0)
#define PUT1(s1) do { uint32_t e1 = S.ctab[(s1)];                                                \
                      acc |= (unsigned long long)(e1 & 0xFFFF) << fill; fill += e1 >> 16;      \
                      uint32_t adv = fill >> 5;                                                \
                      if (adv) atomicOr(&S.out[word], (uint32_t)acc);                          \
                      acc >>= (adv << 5); word += adv; fill &= 31; } while (0)
#pragma unroll
                    for (int d = 15; d >= 0; d--) {
                        if (4u * d + 4 <= C.cnt) {
                            PUT2(C.sym[d] >> 24, (C.sym[d] >> 16) & 0xFF);
                            PUT2((C.sym[d] >> 8) & 0xFF, C.sym[d] & 0xFF);
                        } else if (4u * d < C.cnt) {
                            for (int z = (int)(C.cnt - 4u * d) - 1; z >= 0; z--) PUT1((C.sym[d] >> (8 * z)) & 0xFF);
                        }
                    }
#undef PUT2
#undef PUT1
                    if (lane == 0) { acc |= 1ull << fill; fill += 1; } // end mark above the first symbol's code
                    if (fill) atomicOr(&S.out[word], (uint32_t)acc);
                    if (fill > 32) atomicOr(&S.out[word + 1], (uint32_t)(acc >> 32));
                }
                __syncthreads();
                const uint32_t total = S.misc[12];
                uint32_t *slot32 = (uint32_t *)slot;
                for (uint32_t i = t; i < (total + 3) / 4; i += 256) slot32[i] = S.out[i];
                if (t == 0) csize0[k] = total;
                tree_sent = 1;
                __syncthreads(); // S.out and S.misc are reused by the next chunk
                continue;
            }
            __syncthreads();
        }
        if (HDR && mode != 2 && nseq) { // no table for this group: the literals of a block with sequences travel raw
            const uint32_t lh = ml < 32 ? 1u : (ml < 4096 ? 2u : 3u), content = lh + ml + sec_sz;
            if (content < mk) {
                if (t < FQZ_ENT) ent[t] = 0;
                if (t == 0) {
                    const uint32_t bh = lastblk | (2u << 1) | (content << 3);
                    slot[0] = (uint8_t)bh; slot[1] = (uint8_t)(bh >> 8); slot[2] = (uint8_t)(bh >> 16);
                    const uint32_t v = lh == 1 ? ml << 3 : (((lh == 2 ? 1u : 3u) << 2) | (ml << 4)); // Raw_Literals_Block, size format by lh
                    for (uint32_t q = 0; q < lh; q++) slot[3 + q] = (uint8_t)(v >> (8 * q));
                    csize0[k] = 3 + lh + ml; // (the section follows: k_hdr_patch)
                }
                const uint8_t *lit = H->lit[k];
                for (uint32_t i = t; i < ml; i += 256) slot[3 + lh + i] = lit[i];
                continue;
            }
        }
        // raw block: 3-byte header + the mk bytes, copied from global memory (L2-hot) with 128-bit accesses
        {
            const uint32_t bh = lastblk | (0u << 1) | (mk << 3);
            if (t < FQZ_ENT) ent[t] = 0;
            if (t == 0) { slot[0] = (uint8_t)bh; slot[1] = (uint8_t)(bh >> 8); slot[2] = (uint8_t)(bh >> 16); }
            for (uint32_t off = t * 16; off < mk; off += 256 * 16) {
                if (off + 16 <= mk) store_u128_unaligned(slot + 3 + off, *(const uint4 *)(csrc + off));
                else
                    for (uint32_t q = off; q < mk; q++) slot[3 + q] = csrc[q];
            }
            if (t == 0) csize0[k] = 3 + mk;
        }
    }
    DBG_STOP(9);
}
