/*
 * fqz_entropy.c — oracle for the entropy stage (the "FQZ-H2" profile of DESIGN.md section 4; FQZ-R1 and FQZ-S1 further down).
 *
 * TEST INFRASTRUCTURE ONLY (see fqz_oracle.h).
 *
 * The reference pushes each of the six pre-entropy streams through
 * zstd.Encoder.EncodeAll at SpeedFastest (/root/reference
 * internal/compress/compress.go:115-118, 523-528) and reads them back with
 * zstd.Decoder.DecodeAll (compress.go:785-814).  That library
 * (github.com/klauspost/compress v1.19.1, go.mod:8) is not on disk; its
 * compressed bytes are pinned by no reference test ("parity unpinned").  What
 * IS a contract is the wire format: each payload is a sequence of zstd frames
 * (RFC 8878) that DecodeAll joins.  This file restates the published format
 * for the subset we emit (fqzo_entropy_encode_stream_v has the full layout):
 *
 *   payload = [skippable index frame 'FQZI'] + one frame per GROUP of four
 *             16 KiB chunks: magic | FHD (Single_Segment, Content_Checksum,
 *             Frame_Content_Size) | one block per chunk | XXH64 low 32 bits
 *   block   = RLE block        when all bytes are equal,
 *             Raw block        when the group is < 64 bytes, near-flat
 *                              (sum of squared counts * 230 <= M^2), the packed
 *                              bases, or Huffman does not shrink the chunk,
 *             Compressed block = Huffman-coded literals (1 stream if m < 256,
 *             else 4 streams; one table per group, later blocks treeless) +
 *             "0 sequences" - or, for modelled headers / lengths chunks,
 *             sequences on the predefined FSE tables.
 *   empty stream -> 0 bytes (klauspost EncodeAll without WithZeroFrames).
 *
 * The construction below is fully deterministic; the HIP encoder
 * (fastqpacker_amd/csrc) implements the same steps and must match it
 * byte-for-byte (tests/test_gpu_encode.py, test_gpu_fuzz.py, test_gpu_fullsize.py).
 */
#include "fqz_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define HUF_MAX_BITS 11
#define FSE_W_MAXLOG 6

static inline int highbit32(uint32_t v) { return 31 - __builtin_clz(v); }
static inline void put32le(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* ===================================================================== */
/* Huffman code lengths                                                   */
/* ===================================================================== */

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

int fqzo_huf_code_lengths(const uint32_t count[256], uint8_t nbits[256])
{
    uint32_t key[256];
    int n = 0;
    memset(nbits, 0, 256);
    for (int s = 0; s < 256; s++)
        if (count[s]) key[n++] = (count[s] << 8) | (uint32_t)s; /* count <= 2^17 */
    if (n < 2) return 0;
    qsort(key, (size_t)n, sizeof key[0], cmp_u32); /* ascending (count, symbol): total order */

    /* two-queue Huffman merge; leaf preferred on ties */
    uint32_t cnt[512];
    uint16_t parent[512];
    for (int i = 0; i < n; i++) cnt[i] = key[i] >> 8;
    int li = 0, ih = n, it = n; /* leaf head, internal head, internal tail */
    for (int k = 0; k < n - 1; k++) {
        int a, b;
        if (li < n && (ih >= it || cnt[li] <= cnt[ih])) a = li++; else a = ih++;
        if (li < n && (ih >= it || cnt[li] <= cnt[ih])) b = li++; else b = ih++;
        cnt[it] = cnt[a] + cnt[b];
        parent[a] = parent[b] = (uint16_t)it;
        it++;
    }
    uint8_t depth[512];
    int root = 2 * n - 2;
    depth[root] = 0;
    for (int v = root - 1; v >= 0; v--) depth[v] = (uint8_t)(depth[parent[v]] + 1);

    uint8_t l[256];
    int maxd = 0;
    for (int i = 0; i < n; i++) { l[i] = depth[i]; if (l[i] > maxd) maxd = l[i]; }

    if (maxd > HUF_MAX_BITS) {
        /* length limiting: clamp, then repair the Kraft sum (unit 2^-11) */
        int32_t K = 0;
        for (int i = 0; i < n; i++) {
            if (l[i] > HUF_MAX_BITS) l[i] = HUF_MAX_BITS;
            K += 1 << (HUF_MAX_BITS - l[i]);
        }
        while (K > (1 << HUF_MAX_BITS)) {
            int best = -1;
            for (int i = 0; i < n; i++)
                if (l[i] < HUF_MAX_BITS && (best < 0 || l[i] > l[best])) best = i;
            l[best]++;
            K -= 1 << (HUF_MAX_BITS - l[best]);
        }
        int32_t slack = (1 << HUF_MAX_BITS) - K;
        while (slack > 0) {
            for (int i = n - 1; i >= 0 && slack > 0; i--)
                while (l[i] > 1 && (1 << (HUF_MAX_BITS - l[i])) <= slack) {
                    slack -= 1 << (HUF_MAX_BITS - l[i]);
                    l[i]--;
                }
        }
        maxd = 0;
        for (int i = 0; i < n; i++) if (l[i] > maxd) maxd = l[i];
    }
    for (int i = 0; i < n; i++) nbits[key[i] & 0xFF] = l[i];
    return maxd;
}

void fqzo_huf_codes(const uint8_t nbits[256], int max_bits, uint16_t code[256])
{
    /* RFC 8878 4.2.1.3: codes handed out from the longest length upward, in
     * symbol order within a length. */
    uint32_t nb_per_rank[HUF_MAX_BITS + 2] = {0}, val_per_rank[HUF_MAX_BITS + 2] = {0};
    for (int s = 0; s < 256; s++) nb_per_rank[nbits[s]]++;
    uint32_t min = 0;
    for (int n = max_bits; n > 0; n--) {
        val_per_rank[n] = min;
        min += nb_per_rank[n];
        min >>= 1;
    }
    for (int s = 0; s < 256; s++) code[s] = nbits[s] ? (uint16_t)val_per_rank[nbits[s]]++ : 0;
}

/* ===================================================================== */
/* FSE for the Huffman weights (RFC 8878 4.1 / 4.2.1.2)                   */
/* ===================================================================== */

typedef struct {
    uint8_t *p;
    uint64_t acc;
    int nb;
} bitw;
static inline void bw_add(bitw *b, uint32_t v, int n)
{
    b->acc |= (uint64_t)v << b->nb;
    b->nb += n;
    while (b->nb >= 8) { *b->p++ = (uint8_t)b->acc; b->acc >>= 8; b->nb -= 8; }
}
static inline uint8_t *bw_close(bitw *b) /* end mark 1 + pad */
{
    bw_add(b, 1, 1);
    if (b->nb) { *b->p++ = (uint8_t)b->acc; b->acc = 0; b->nb = 0; }
    return b->p;
}

/* The profile does not fit a distribution to the weights of every group: it picks one of two fixed normalised
 * distributions over the weight values 0..11 (table log 5), by the share of zero weights (absent symbols).  The
 * compressed description grows by ~15 bytes per chunk against a fitted table, and the encoder saves the serial
 * count / normalise / NCount / table-build steps — 15 % of the GPU entropy kernel.  (A zstd encoder is free to
 * choose any normalised counts; the decoder reads them from the NCount header as usual.) */
#define FSE_W_LOG 5
static const int FSE_W_NORM[2][12] = {
    {21, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},  /* sparse alphabets: at least half of the weights are 0 */
    {1, 1, 2, 3, 5, 6, 5, 3, 2, 2, 1, 1},   /* dense alphabets */
};

/* returns compressed size, 0 = nothing to code (caller falls back) */
static size_t fse_compress_weights(const uint8_t *w, int n, uint8_t *dst)
{
    if (n <= 2) return 0;
    int zeros = 0;
    for (int i = 0; i < n; i++) zeros += (w[i] == 0);
    const int *fixed = FSE_W_NORM[2 * zeros >= n ? 0 : 1];
    const int maxw = 11, table_log = FSE_W_LOG, table_size = 1 << FSE_W_LOG;
    int norm[13] = {0};
    for (int s = 0; s <= maxw; s++) norm[s] = fixed[s];

    /* FSE_writeNCount */
    uint8_t *op = dst;
    {
        uint32_t bits = 0;
        int bc = 0;
        int remaining = table_size + 1, threshold = table_size, nb = table_log + 1;
        int sym = 0, alphabet = maxw + 1, prev0 = 0;
        bits += (uint32_t)(table_log - 5) << bc; bc += 4;
        while (sym < alphabet && remaining > 1) {
            if (prev0) {
                int start = sym;
                while (sym < alphabet && !norm[sym]) sym++;
                if (sym == alphabet) break;
                while (sym >= start + 3) { start += 3; bits += 3u << bc; bc += 2; }
                bits += (uint32_t)(sym - start) << bc; bc += 2;
                if (bc > 16) { *op++ = (uint8_t)bits; *op++ = (uint8_t)(bits >> 8); bits >>= 16; bc -= 16; }
            }
            {
                int c = norm[sym++];
                int max = (2 * threshold - 1) - remaining;
                remaining -= c;
                c++;
                if (c >= threshold) c += max;
                bits += (uint32_t)c << bc;
                bc += nb;
                bc -= (c < max);
                prev0 = (c == 1);
                while (remaining < threshold) { nb--; threshold >>= 1; }
            }
            if (bc > 16) { *op++ = (uint8_t)bits; *op++ = (uint8_t)(bits >> 8); bits >>= 16; bc -= 16; }
        }
        if (bc > 0) *op++ = (uint8_t)bits;
        if (bc > 8) *op++ = (uint8_t)(bits >> 8);
    }

    /* FSE_buildCTable */
    uint16_t state_table[64];
    uint8_t table_symbol[64];
    int32_t delta_nb[13], delta_find[13];
    {
        int cumul[14];
        cumul[0] = 0;
        for (int s = 1; s <= maxw + 1; s++) cumul[s] = cumul[s - 1] + norm[s - 1];
        int step = (table_size >> 1) + (table_size >> 3) + 3, mask = table_size - 1, pos = 0;
        for (int s = 0; s <= maxw; s++)
            for (int i = 0; i < norm[s]; i++) { table_symbol[pos] = (uint8_t)s; pos = (pos + step) & mask; }
        for (int u = 0; u < table_size; u++) { int s = table_symbol[u]; state_table[cumul[s]++] = (uint16_t)(table_size + u); }
        int total = 0;
        for (int s = 0; s <= maxw; s++) {
            if (norm[s] == 0) { delta_nb[s] = ((table_log + 1) << 16) - table_size; delta_find[s] = 0; }
            else if (norm[s] == 1) { delta_nb[s] = (table_log << 16) - table_size; delta_find[s] = total - 1; total++; }
            else {
                int max_bits_out = table_log - highbit32((uint32_t)(norm[s] - 1));
                int min_state_plus = norm[s] << max_bits_out;
                delta_nb[s] = (max_bits_out << 16) - min_state_plus;
                delta_find[s] = total - norm[s];
                total += norm[s];
            }
        }
    }

    /* encode backwards with two states: state1 = even positions, state2 = odd */
    bitw bw = { op, 0, 0 };
    uint32_t st[2];
    int inited[2] = {0, 0};
    for (int i = n - 1; i >= 0; i--) {
        int which = i & 1, s = w[i];
        if (!inited[which]) { /* FSE_initCState2 */
            uint32_t nb_out = (uint32_t)(delta_nb[s] + (1 << 15)) >> 16;
            uint32_t value = (nb_out << 16) - (uint32_t)delta_nb[s];
            st[which] = state_table[(value >> nb_out) + delta_find[s]];
            inited[which] = 1;
        } else {
            uint32_t nb_out = (st[which] + (uint32_t)delta_nb[s]) >> 16;
            bw_add(&bw, st[which] & ((1u << nb_out) - 1), (int)nb_out);
            st[which] = state_table[(st[which] >> nb_out) + delta_find[s]];
        }
    }
    bw_add(&bw, st[1] & (uint32_t)(table_size - 1), table_log); /* flush state2 then state1 */
    bw_add(&bw, st[0] & (uint32_t)(table_size - 1), table_log);
    op = bw_close(&bw);
    return (size_t)(op - dst);
}

size_t fqzo_huf_write_tree(const uint8_t nbits[256], int max_bits, uint8_t *dst)
{
    int max_sym = 255;
    while (max_sym > 0 && !nbits[max_sym]) max_sym--;
    int nw = max_sym; /* weights for symbols 0..max_sym-1; the last is implied */
    uint8_t w[256];
    for (int s = 0; s < nw; s++) w[s] = nbits[s] ? (uint8_t)(max_bits + 1 - nbits[s]) : 0;
    if (nw <= 128) { /* direct 4-bit weights whenever they fit: no serial FSE pass on the GPU for ASCII-range alphabets */
        dst[0] = (uint8_t)(128 + (nw - 1));
        w[nw] = 0;
        for (int i = 0; i < nw; i += 2) dst[i / 2 + 1] = (uint8_t)((w[i] << 4) + w[i + 1]);
        return (size_t)((nw + 1) / 2) + 1;
    }
    uint8_t tmp[300];
    size_t h = fse_compress_weights(w, nw, tmp);
    if (h > 1 && h < 128) { /* header byte < 128 = size of the FSE-compressed weights */
        dst[0] = (uint8_t)h;
        memcpy(dst + 1, tmp, h);
        return h + 1;
    }
    return 0; /* more than 128 weights and FSE does not help: the chunk is stored raw */
}

/* ===================================================================== */
/* one zstd block per chunk                                               */
/* ===================================================================== */

static inline void put_block_header(uint8_t *dst, int last, int type, uint32_t size)
{
    uint32_t v = (uint32_t)(last & 1) | ((uint32_t)type << 1) | (size << 3);
    dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
}

static size_t raw_block(const uint8_t *src, size_t m, int last, uint8_t *dst)
{
    put_block_header(dst, last, 0, (uint32_t)m);
    memcpy(dst + 3, src, m);
    return 3 + m;
}

/* HUF 1X stream: symbols written last-to-first, then the end mark */
static size_t huf_stream(const uint8_t *src, size_t n, const uint16_t *code, const uint8_t *nbits, uint8_t *dst)
{
    bitw bw = { dst, 0, 0 };
    for (size_t i = n; i-- > 0;) bw_add(&bw, code[src[i]], nbits[src[i]]);
    return (size_t)(bw_close(&bw) - dst);
}

/* One GROUP of up to FQZO_GROUP consecutive 16 KiB chunks = up to 64 KiB of a stream.  The profile builds ONE Huffman
 * table per group, from the histogram of the whole group: the first Compressed block of the group carries the tree
 * description, the others are "treeless" (Literals_Block_Type 3, RFC 8878 3.1.1.3.1.1: reuse the previous table).
 * On the GPU the table build (sort, Huffman merge, FSE weight coding) is a quarter of the LDS traffic of a chunk; a
 * group pays it once for four blocks, and the decoder builds four times fewer tables. */
/* ===================================================================== */
/* header-stream modelling: matches with the previous record as zstd       */
/* sequences (RFC 8878 3.1.1.3.2) with the predefined FSE tables            */
/* ===================================================================== */
/* The headers stream is [u16 H][H bytes] per record (compress.go:514-515) and consecutive Illumina headers share a long
 * prefix and a long suffix.  Inside one 16 KiB chunk (= one zstd block; matches never leave the block, so blocks stay
 * independent) every record that lies wholly inside the chunk, and whose predecessor does too, is compared with its
 * predecessor: a = common prefix, b = common suffix of what is left.  The suffix of record i (aligned at the ends: offset
 * = length of record i) and the prefix of record i + 1 (aligned at the starts: offset = length of record i again) are
 * adjacent and share their offset, so every record boundary gives at most ONE match, of b_i + a_{i+1} bytes at offset
 * len_i; it is emitted when it is at least HDR_MIN_MATCH bytes long.  Everything else is literals (Huffman-coded with
 * the group's table as before).  Offsets are always coded explicitly (Offset_Value = offset + 3), never as repeat
 * offsets, so that no block depends on the offset history its predecessors leave behind; all three symbol types use
 * Predefined_Mode, so there is no table description to build. */
#define HDR_MIN_MATCH 6
#define HDR_MAX_SEQ 2048 /* per chunk; a chunk with more records inside is coded without matches */
typedef struct { uint32_t ll, ml, off; } hseq;

static const short LL_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const short ML_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const short OF_NORM[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                     48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                     30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                     16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* FSE compression table of a normalised distribution (counts may be -1 = "less than one"): FSE_buildCTable */
typedef struct { uint16_t state[512]; int32_t dnb[64], dfs[64]; int log; } fse_ct;
static void fse_build_ct(fse_ct *ct, const short *norm, int nsym, int log)
{
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    int cumul[65];
    uint8_t tsym[512];
    int high = size - 1;
    cumul[0] = 0;
    for (int u = 1; u <= nsym; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tsym[high--] = (uint8_t)(u - 1); }
        else cumul[u] = cumul[u - 1] + norm[u - 1];
    }
    int pos = 0;
    for (int sy = 0; sy < nsym; sy++)
        for (int k = 0; k < norm[sy]; k++) {
            tsym[pos] = (uint8_t)sy;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    for (int u = 0; u < size; u++) { const int sy = tsym[u]; ct->state[cumul[sy]++] = (uint16_t)(size + u); }
    int total = 0;
    for (int sy = 0; sy < nsym; sy++) {
        const int n = norm[sy];
        if (n == 0) { ct->dnb[sy] = ((log + 1) << 16) - (1 << log); ct->dfs[sy] = 0; }
        else if (n == -1 || n == 1) { ct->dnb[sy] = (log << 16) - (1 << log); ct->dfs[sy] = total - 1; total++; }
        else {
            const int maxbits = log - highbit32((uint32_t)(n - 1));
            ct->dnb[sy] = (maxbits << 16) - (n << maxbits);
            ct->dfs[sy] = total - n;
            total += n;
        }
    }
    ct->log = log;
}
static inline uint32_t fse_init_state(const fse_ct *ct, int sy) /* FSE_initCState2 */
{
    const uint32_t nb = (uint32_t)(ct->dnb[sy] + (1 << 15)) >> 16;
    const uint32_t value = (nb << 16) - (uint32_t)ct->dnb[sy];
    return ct->state[(value >> nb) + (uint32_t)ct->dfs[sy]];
}
static inline uint32_t fse_encode(const fse_ct *ct, bitw *bw, uint32_t st, int sy) /* FSE_encodeSymbol */
{
    const uint32_t nb = (st + (uint32_t)ct->dnb[sy]) >> 16;
    bw_add(bw, st & ((1u << nb) - 1), (int)nb);
    return ct->state[(st >> nb) + (uint32_t)ct->dfs[sy]];
}
static inline int ll_code(uint32_t ll)
{
    static const uint8_t T[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                                  22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
    return ll < 64 ? T[ll] : highbit32(ll) + 19;
}
static inline int ml_code(uint32_t mlb) /* mlb = match length - 3 */
{
    static const uint8_t T[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
                                   32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
                                   40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
                                   42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
    return mlb < 128 ? T[mlb] : highbit32(mlb) + 36;
}

/* Sequences_Section of n > 0 sequences: count, modes byte (all Predefined), bitstream (ZSTD_encodeSequences order) */
static fse_ct LL, ML, OF; /* compression tables of the predefined distributions */
static void hdr_ct_init(void) { fse_build_ct(&LL, LL_NORM, 36, 6); fse_build_ct(&ML, ML_NORM, 53, 6); fse_build_ct(&OF, OF_NORM, 29, 5); }
static size_t hdr_write_sequences(const hseq *sq, uint32_t n, uint8_t *dst)
{
    static pthread_once_t once = PTHREAD_ONCE_INIT; /* (the pipeline's workers call this concurrently) */
    pthread_once(&once, hdr_ct_init);
    uint8_t *op = dst;
    if (n < 128) *op++ = (uint8_t)n;
    else { *op++ = (uint8_t)((n >> 8) + 128); *op++ = (uint8_t)n; } /* n < 0x7F00 */
    *op++ = 0; /* Symbol_Compression_Modes: Predefined_Mode x 3 */
    bitw bw = { op, 0, 0 };
    /* Offset_Value: 1 = "the offset of the previous sequence" (legal when this sequence has literals, RFC 8878 3.1.1.5) for a
     * sequence whose offset repeats the previous one of THIS block; offset + 3 otherwise.  The first sequence of a block is
     * always explicit, so no block depends on the offset history its predecessors leave behind. */
#define HDR_OFV(k) (((k) > 0 && sq[k].off == sq[(k) - 1].off && sq[k].ll > 0) ? 1u : sq[k].off + 3u)
    uint32_t i = n - 1;
    uint32_t ofv = HDR_OFV(i);
    int lc = ll_code(sq[i].ll), mc = ml_code(sq[i].ml - 3), oc = highbit32(ofv);
    uint32_t st_ml = fse_init_state(&ML, mc), st_of = fse_init_state(&OF, oc), st_ll = fse_init_state(&LL, lc);
    bw_add(&bw, sq[i].ll - LL_BASE[lc], LL_BITS[lc]);
    bw_add(&bw, sq[i].ml - ML_BASE[mc], ML_BITS[mc]);
    bw_add(&bw, ofv - (1u << oc), oc);
    while (i-- > 0) {
        ofv = HDR_OFV(i);
        lc = ll_code(sq[i].ll); mc = ml_code(sq[i].ml - 3); oc = highbit32(ofv);
        st_of = fse_encode(&OF, &bw, st_of, oc);
        st_ml = fse_encode(&ML, &bw, st_ml, mc);
        st_ll = fse_encode(&LL, &bw, st_ll, lc);
        bw_add(&bw, sq[i].ll - LL_BASE[lc], LL_BITS[lc]);
        bw_add(&bw, sq[i].ml - ML_BASE[mc], ML_BITS[mc]);
        bw_add(&bw, ofv - (1u << oc), oc);
    }
#undef HDR_OFV
    bw_add(&bw, st_ml & 63, 6); /* FSE_flushCState: ML, OF, LL */
    bw_add(&bw, st_of & 31, 5);
    bw_add(&bw, st_ll & 63, 6);
    return (size_t)(bw_close(&bw) - dst);
}

/* sequences of chunk [c0, c0 + mk) of a headers stream whose records start at rs[0 .. nr) (rs[nr] = end of the stream);
 * lit receives the literals (<= mk bytes).  Returns the number of sequences (0: no matches, lit = the chunk).
 * Per record i that lies inside the chunk together with its predecessor p:
 *   head_i : the common prefix with p (aligned at the starts: offset len_p); when the u16 length prefixes differ (headers
 *            of different length) the comparison starts behind them, at byte 2
 *   tail_i : the common suffix with p of what the head left (aligned at the ends: offset len_i)
 * tail_i and head_{i+1} share their offset (len_i) and are merged when they touch.  Candidates shorter than HDR_MIN_MATCH
 * stay literals. */
static uint32_t hdr_chunk_model(const uint8_t *stream, const uint32_t *rs, uint32_t nr, uint32_t c0, uint32_t mk, hseq *sq, uint8_t *lit, uint32_t *n_lit)
{
    const uint32_t c1 = c0 + mk;
    uint32_t lo = 0, hi = nr; /* first record that starts at or behind c0 */
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rs[mid] < c0) lo = mid + 1; else hi = mid; }
    const uint32_t first = lo;
    uint32_t n_in = 0;
    while (first + n_in < nr && rs[first + n_in + 1] <= c1) n_in++;
    if (2 * n_in > HDR_MAX_SEQ) n_in = 0;
    uint32_t nseq = 0, prev_end = c0, nl = 0;
    /* pending candidate (a tail waiting for the next record's head) */
    uint32_t pend_pos = 0, pend_len = 0, pend_off = 0;
#define HDR_EMIT(pos_, len_, off_) do { if ((len_) >= HDR_MIN_MATCH) { \
        sq[nseq].ll = (pos_) - prev_end; sq[nseq].ml = (len_); sq[nseq].off = (off_); \
        memcpy(lit + nl, stream + prev_end, (pos_) - prev_end); nl += (pos_) - prev_end; prev_end = (pos_) + (len_); nseq++; } } while (0)
    for (uint32_t k = 1; k < n_in; k++) { /* k = 0 has no predecessor inside the chunk */
        const uint32_t i = first + k, s = rs[i], e = rs[i + 1], len = e - s, ps = rs[i - 1], plen = s - ps;
        const uint32_t lim = len < plen ? len : plen;
        uint32_t h_start = s, h_len = 0;
        while (h_len < lim && stream[s + h_len] == stream[ps + h_len]) h_len++;
        if (h_len < HDR_MIN_MATCH) { /* behind the length prefixes */
            h_start = s + 2; h_len = 0;
            while (2 + h_len < lim && stream[s + 2 + h_len] == stream[ps + 2 + h_len]) h_len++;
            if (h_len < HDR_MIN_MATCH) { h_start = s; h_len = 0; }
        }
        /* the head joins the pending tail of the predecessor when they touch (same offset: len_p) */
        if (pend_len && h_len && h_start == s && pend_pos + pend_len == s) { pend_len += h_len; HDR_EMIT(pend_pos, pend_len, pend_off); }
        else {
            if (pend_len) HDR_EMIT(pend_pos, pend_len, pend_off);
            if (h_len) HDR_EMIT(h_start, h_len, plen);
        }
        pend_len = 0;
        /* tail: common suffix of what is left */
        const uint32_t used = h_len ? (h_start + h_len) - s : 0;
        uint32_t tl = 0, tlim = len - used < plen ? len - used : plen;
        while (tl < tlim && stream[e - 1 - tl] == stream[s - 1 - tl]) tl++;
        if (tl) { pend_pos = e - tl; pend_len = tl; pend_off = len; }
    }
    if (pend_len) HDR_EMIT(pend_pos, pend_len, pend_off);
#undef HDR_EMIT
    memcpy(lit + nl, stream + prev_end, c1 - prev_end);
    nl += c1 - prev_end;
    *n_lit = nl;
    return nseq;
}

/* ===================================================================== */
/* one group (= one zstd frame): a Huffman table over its literals, one    */
/* zstd block per chunk                                                    */
/* ===================================================================== */
typedef struct {
    const uint8_t *raw; uint32_t mk;       /* the chunk as it lies in the stream */
    const uint8_t *lit; uint32_t n_lit;    /* its literals (== raw, mk when there are no sequences) */
    const hseq *sq; uint32_t nseq;
} gchunk;

#define HDR_SSZ_BOUND(n) (((n) < 128 ? 2u : 3u) + ((n) * 66u + 18u + 7u) / 8u) /* count + modes byte + bit stream: <= 66 bits a sequence, 17 flush bits, the end mark */
/* ent (NULL: not wanted): per chunk the twelve ENTRY POINTS of its four Huffman streams (FQZI index, flags bit 1), zeros for
 * any block that is not Huffman-coded in four streams.  A stream of L symbols is written last symbol first, so a decoder reads
 * symbol 0 first, from the top of the stream down - one lane, L dependent steps.  With per = (ceil(L / 64) + 3) & ~3 (the symbols
 * a lane of the GPU encoder owns) the stream is cut at the symbols e_j = min(16 j per, L), j = 1..3, and the entry point E_j is
 * the number of bits the symbols [e_j, L) take = the bit position (LSB-first from the stream's first byte) right above the code
 * of symbol e_j: a decoder lane starts there and decodes the symbols [e_j, e_j+1) without waiting for the lane in front of it. */
#define FQZO_ENT 12
static size_t encode_group_chunks(const gchunk *ch, int nch, int force_raw, uint8_t *dst, uint16_t (*ent)[FQZO_ENT])
{
    if (ent) memset(ent, 0, sizeof(uint16_t) * FQZO_ENT * (size_t)nch);
    uint32_t count[256] = {0};
    size_t M = 0;
    for (int k = 0; k < nch; k++) { for (uint32_t i = 0; i < ch[k].n_lit; i++) count[ch[k].lit[i]]++; M += ch[k].n_lit; }
    int huff = 1, n_active = 0;
    for (int sy = 0; sy < 256; sy++) n_active += count[sy] != 0;
    if (force_raw) huff = 0;                   /* 2-bit packed bases: Raw blocks by definition (FQZ-H2), no histogram at all */
    else if (n_active <= 1) huff = 0;          /* one symbol: every chunk is an RLE block (or, with sequences, has raw literals) */
    else if (M < 64) huff = 0;
    else {   /* near-flat histogram: collision entropy -log2(sum p^2) >= log2(230) = 7.85 bits bounds the Shannon entropy
              * from below, so a Huffman table could save < 2 %: store raw (this is what 2-bit packed bases look like) */
        uint64_t sq = 0;
        for (int sy = 0; sy < 256; sy++) sq += (uint64_t)count[sy] * count[sy];
        if (sq * 230 <= (uint64_t)M * M) huff = 0;
    }
    uint8_t nbits[256];
    uint16_t code[256];
    uint8_t tree[260];
    size_t tree_size = 0;
    if (huff) {
        int max_bits = fqzo_huf_code_lengths(count, nbits);
        tree_size = fqzo_huf_write_tree(nbits, max_bits, tree);
        if (!tree_size) huff = 0;
        else fqzo_huf_codes(nbits, max_bits, code);
    }
    int tree_sent = 0;
    uint8_t *out = dst;
    uint8_t *seqsec = (uint8_t *)malloc(16 + 12 * (size_t)HDR_MAX_SEQ);
    for (int k = 0; k < nch; k++) {
        const size_t m = ch[k].n_lit, mk = ch[k].mk;
        const uint8_t *c = ch[k].lit;
        const int lastblk = k + 1 == nch;
        const uint32_t nseq = ch[k].nseq;
        int same = !force_raw && nseq == 0;
        for (size_t i = 1; same && i < m; i++) if (c[i] != c[0]) same = 0;
        if (same) { /* RLE block */
            put_block_header(out, lastblk, 1, (uint32_t)mk);
            out[3] = c[0];
            out += 4;
            continue;
        }
        const size_t ssz = nseq ? hdr_write_sequences(ch[k].sq, nseq, seqsec) : 1; /* Number_of_Sequences = 0: one byte */
        size_t content = 0, lh = 0, lit_csize = 0, ssize[4] = {0, 0, 0, 0};
        int nstreams = 1;
        size_t seg = m;
        const size_t tsz = tree_sent ? 0 : tree_size;
        if (huff) {
            nstreams = m >= 256 ? 4 : 1;
            seg = nstreams == 4 ? (m + 3) / 4 : m;
            size_t streams_total = 0;
            for (int q = 0; q < nstreams; q++) {
                size_t a = (size_t)q * seg, b = (q == nstreams - 1) ? m : a + seg;
                uint64_t bits = 0;
                for (size_t i = a; i < b; i++) bits += nbits[c[i]];
                ssize[q] = (size_t)(bits >> 3) + 1;
                streams_total += ssize[q];
            }
            lit_csize = tsz + (nstreams == 4 ? 6 : 0) + streams_total;
            lh = m < 1024 ? 3 : (m < 16384 ? 4 : 5);
            content = lh + lit_csize + ssz;
        } else if (nseq) { /* no table for this group: the literals of a block with sequences travel raw */
            lh = m < 32 ? 1 : (m < 4096 ? 2 : 3);
            content = lh + m + ssz;
        }
        /* Raw block when coding does not pay.  With sequences the test uses an upper bound of the section size
         * (66 bits a sequence) instead of the section itself: a parallel encoder then places and codes the literals of a
         * block without waiting for the serial FSE state chains of its sequences (FQZ-H2 rule; costs nothing unless a
         * block is within a few hundred bytes of not compressing at all) */
        const size_t judged = nseq ? content - ssz + HDR_SSZ_BOUND(nseq) : content;
        if ((!huff && !nseq) || judged >= mk) { out += raw_block(ch[k].raw, mk, lastblk, out); continue; }
        put_block_header(out, lastblk, 2, (uint32_t)content);
        uint8_t *op = out + 3;
        if (!huff) { /* Raw_Literals_Block */
            if (lh == 1) op[0] = (uint8_t)(m << 3);
            else if (lh == 2) { const uint32_t v = (1u << 2) | ((uint32_t)m << 4); op[0] = (uint8_t)v; op[1] = (uint8_t)(v >> 8); }
            else { const uint32_t v = (3u << 2) | ((uint32_t)m << 4); op[0] = (uint8_t)v; op[1] = (uint8_t)(v >> 8); op[2] = (uint8_t)(v >> 16); }
            op += lh;
            memcpy(op, c, m);
            op += m;
        } else {
            const uint32_t lt = tree_sent ? 3u : 2u; /* treeless once the group's table has been sent */
            if (lh == 3) {
                uint32_t v = lt | ((nstreams == 4 ? 1u : 0u) << 2) | ((uint32_t)m << 4) | ((uint32_t)lit_csize << 14);
                op[0] = (uint8_t)v; op[1] = (uint8_t)(v >> 8); op[2] = (uint8_t)(v >> 16);
            } else if (lh == 4) {
                uint32_t v = lt | (2u << 2) | ((uint32_t)m << 4) | ((uint32_t)lit_csize << 18);
                op[0] = (uint8_t)v; op[1] = (uint8_t)(v >> 8); op[2] = (uint8_t)(v >> 16); op[3] = (uint8_t)(v >> 24);
            } else {
                uint32_t v = lt | (3u << 2) | ((uint32_t)m << 4) | ((uint32_t)lit_csize << 22);
                op[0] = (uint8_t)v; op[1] = (uint8_t)(v >> 8); op[2] = (uint8_t)(v >> 16); op[3] = (uint8_t)(v >> 24);
                op[4] = (uint8_t)(lit_csize >> 10);
            }
            op += lh;
            memcpy(op, tree, tsz);
            op += tsz;
            tree_sent = 1;
            if (nstreams == 4) {
                for (int q = 0; q < 3; q++) { op[2 * q] = (uint8_t)ssize[q]; op[2 * q + 1] = (uint8_t)(ssize[q] >> 8); }
                op += 6;
            }
            for (int q = 0; q < nstreams; q++) {
                size_t a = (size_t)q * seg, b = (q == nstreams - 1) ? m : a + seg;
                (void)huf_stream(c + a, b - a, code, nbits, op);
                op += ssize[q];
                if (ent && nstreams == 4) {
                    const size_t L = b - a, per = ((L + 63) / 64 + 3) & ~(size_t)3;
                    for (int j = 1; j <= 3; j++) {
                        const size_t e = 16 * (size_t)j * per < L ? 16 * (size_t)j * per : L;
                        uint32_t bits = 0;
                        for (size_t i = a + e; i < b; i++) bits += nbits[c[i]];
                        ent[k][3 * q + j - 1] = (uint16_t)bits; /* <= 4096 symbols x 11 bits */
                    }
                }
            }
        }
        if (nseq) { memcpy(op, seqsec, ssz); op += ssz; }
        else *op++ = 0; /* Number_of_Sequences = 0 */
        out = op;
    }
    free(seqsec);
    return (size_t)(out - dst);
}

/* a group of plain chunks: M consecutive bytes cut every FQZO_CHUNK */
static size_t encode_group_ex(const uint8_t *src, size_t M, int last, int force_raw, uint8_t *dst, uint16_t (*ent)[FQZO_ENT])
{
    (void)last; /* every group is a frame of its own: its last chunk carries the Last_Block bit */
    gchunk ch[FQZO_GROUP];
    int nch = 0;
    for (size_t off = 0; off < M; off += FQZO_CHUNK) {
        const uint32_t m = (uint32_t)(M - off < FQZO_CHUNK ? M - off : FQZO_CHUNK);
        ch[nch].raw = ch[nch].lit = src + off; ch[nch].mk = ch[nch].n_lit = m; ch[nch].sq = NULL; ch[nch].nseq = 0;
        nch++;
    }
    return encode_group_chunks(ch, nch, force_raw, dst, ent);
}
size_t fqzo_encode_group(const uint8_t *src, size_t M, int last, uint8_t *dst) { return encode_group_ex(src, M, last, 0, dst, NULL); }

/* a single chunk = a group of one */
size_t fqzo_encode_chunk(const uint8_t *src, size_t m, int last, uint8_t *dst) { return fqzo_encode_group(src, m, last, dst); }


/* ===================================================================== */
/* FQZ-R1: interleaved rANS blocks (container version 3)   */
/* ===================================================================== */
/* A version-3 file is a version-2 file (same headers, same six payloads, same FQZ-H2 framing: index frame, one frame per
 * group, content checksums) in which the blocks of the QUALITY payload may have Block_Type 3 - reserved in RFC 8878, so
 * these payloads are no longer zstd and the stock decoder rejects the file by its version (compress.go:571-573):
 *
 *   Block_Header (3 bytes, type 3, Block_Size = what follows)
 *   u16le m          bytes the block regenerates (1..16384)
 *   u8    tflag      1: a frequency table follows - in the first type-3 block of a frame and only there - else 0
 *   [ u8 nsym-1 | nsym symbols, ascending | nsym x u16le frequency (1..4095, sum 4096) | a zero byte if nsym is even ]
 *   16-bit words     little endian, in the order the encoder emitted them
 *   16 x u32le       the final states of the 16 coders (lane 0 first)
 *
 * Sixteen order-0 rANS coders (state in [2^16, 2^32), 12-bit frequencies, 16-bit renormalisation) share the block: byte i
 * belongs to coder (i >> 4) & 15 - sixteen consecutive bytes a coder, so a decoder lane stores 16-byte units - and is its
 * ((i >> 8) << 4 | (i & 15))-th symbol ("step").  The encoder runs the steps from the last to the first, within a step
 * the coders from 15 down to 0, every coder starting at 2^16, and appends a word whenever a coder renormalises; the
 * decoder starts from the final states, runs steps and coders upwards, takes its words from the end of the word area
 * backwards, and must arrive at the start of the word area with every state back at 2^16 (which is checked).
 * Frequencies: round(count * 4096 / M) over the group's histogram, at least 1, the rounding error goes to the most
 * frequent symbol (the smallest one on ties). */
#define RANS_L 65536u
typedef struct { uint16_t freq[256], cum[256]; uint8_t ser[1 + 3 * 256 + 1]; size_t ser_len; } rans_tab;

static int rans_normalise(const uint32_t count[256], uint32_t M, rans_tab *t)
{
    int best = -1, n_active = 0;
    uint32_t sum = 0;
    for (int sy = 0; sy < 256; sy++) {
        t->freq[sy] = 0;
        if (!count[sy]) continue;
        uint32_t f = (count[sy] * 8192u + M) / (2u * M);
        if (!f) f = 1;
        t->freq[sy] = (uint16_t)f;
        sum += f;
        n_active++;
        if (best < 0 || count[sy] > count[best]) best = sy;
    }
    if (n_active < 2) return 0;
    const int fixed = (int)t->freq[best] + 4096 - (int)sum;
    if (fixed < 1) return 0; /* (cannot happen for histograms that pass the flatness test; the group is then stored raw) */
    t->freq[best] = (uint16_t)fixed;
    uint32_t c = 0;
    uint8_t *o = t->ser;
    *o++ = (uint8_t)(n_active - 1);
    for (int sy = 0; sy < 256; sy++) if (t->freq[sy]) *o++ = (uint8_t)sy;
    for (int sy = 0; sy < 256; sy++) {
        t->cum[sy] = (uint16_t)c;
        c += t->freq[sy];
        if (t->freq[sy]) { *o++ = (uint8_t)t->freq[sy]; *o++ = (uint8_t)(t->freq[sy] >> 8); }
    }
    if (!(n_active & 1)) *o++ = 0;
    t->ser_len = (size_t)(o - t->ser);
    return n_active;
}

/* content of a type-3 block (without the 3-byte block header); dst holds 6 + 770 + 2 m + 64 bytes */
static size_t rans_block_content(const uint8_t *c, uint32_t m, const rans_tab *t, int with_table, uint8_t *dst)
{
    uint8_t *op = dst;
    *op++ = (uint8_t)m; *op++ = (uint8_t)(m >> 8); *op++ = with_table ? 1 : 0;
    if (with_table) { memcpy(op, t->ser, t->ser_len); op += t->ser_len; }
    uint32_t x[16];
    for (int j = 0; j < 16; j++) x[j] = RANS_L;
    const uint32_t steps = 16 * ((m + 255) / 256);
    for (uint32_t st = steps; st-- > 0;)
        for (int j = 15; j >= 0; j--) {
            const uint32_t i = ((st >> 4) << 8) | ((uint32_t)j << 4) | (st & 15);
            if (i >= m) continue;
            const uint32_t f = t->freq[c[i]];
            if (x[j] >= (f << 20)) { *op++ = (uint8_t)x[j]; *op++ = (uint8_t)(x[j] >> 8); x[j] >>= 16; }
            x[j] = ((x[j] / f) << 12) + (x[j] % f) + t->cum[c[i]];
        }
    for (int j = 0; j < 16; j++) { put32le(op, x[j]); op += 4; }
    return (size_t)(op - dst);
}

/* a group of the quality stream of a version-3 file: same decisions as encode_group_chunks (RLE block for a chunk of one
 * byte, Raw blocks for short or near-flat groups, Raw block for a chunk that coding does not shrink), rANS in the place of
 * the Huffman-coded literals */
static size_t encode_group_rans(const uint8_t *src, size_t M, uint8_t *dst)
{
    /* the histogram covers the chunks that are coded: a chunk of one repeated byte becomes an RLE block and stays out of it */
    uint32_t count[256] = {0}, Mc = 0;
    int same_k[FQZO_GROUP] = {0};
    for (size_t off = 0, k = 0; off < M; off += FQZO_CHUNK, k++) {
        const uint32_t mk = (uint32_t)(M - off < FQZO_CHUNK ? M - off : FQZO_CHUNK);
        int same = 1;
        for (uint32_t i = 1; same && i < mk; i++) if (src[off + i] != src[off]) same = 0;
        same_k[k] = same;
        if (same) continue;
        for (uint32_t i = 0; i < mk; i++) count[src[off + i]]++;
        Mc += mk;
    }
    int coded = 1, n_active = 0;
    for (int sy = 0; sy < 256; sy++) n_active += count[sy] != 0;
    if (n_active <= 1 || Mc < 64) coded = 0;
    else {
        uint64_t sq = 0;
        for (int sy = 0; sy < 256; sy++) sq += (uint64_t)count[sy] * count[sy];
        if (sq * 230 <= (uint64_t)Mc * Mc) coded = 0;
    }
    rans_tab tab;
    if (coded && !rans_normalise(count, Mc, &tab)) coded = 0;
    /* the first chunk that is not RLE carries the table; if coding does not shrink it, the group is not coded at all */
    uint8_t *tmp = (uint8_t *)malloc(6 + sizeof tab.ser + 2 * (size_t)FQZO_CHUNK + 64);
    int carrier = -1;
    for (size_t off = 0, k = 0; off < M && carrier < 0; off += FQZO_CHUNK, k++) if (!same_k[k]) carrier = (int)k;
    if (coded) {
        const size_t off = (size_t)carrier * FQZO_CHUNK;
        const uint32_t mk = (uint32_t)(M - off < FQZO_CHUNK ? M - off : FQZO_CHUNK);
        if (rans_block_content(src + off, mk, &tab, 1, tmp) >= mk) coded = 0;
    }
    uint8_t *out = dst;
    for (size_t off = 0; off < M; off += FQZO_CHUNK) {
        const uint32_t mk = (uint32_t)(M - off < FQZO_CHUNK ? M - off : FQZO_CHUNK);
        const uint8_t *c = src + off;
        const int lastblk = off + mk == M, k = (int)(off / FQZO_CHUNK);
        if (same_k[k]) { put_block_header(out, lastblk, 1, mk); out[3] = c[0]; out += 4; continue; }
        if (coded) {
            const size_t content = rans_block_content(c, mk, &tab, k == carrier, tmp);
            if (content < mk) { /* (always, for the carrier) */
                put_block_header(out, lastblk, 3, (uint32_t)content);
                memcpy(out + 3, tmp, content);
                out += 3 + content;
                continue;
            }
        }
        out += raw_block(c, mk, lastblk, out);
    }
    free(tmp);
    return (size_t)(out - dst);
}

/* ===================================================================== */
/* frame                                                                  */
/* ===================================================================== */

/* XXH64 (Y. Collet's public algorithm): zstd's Content_Checksum is its low 32 bits over the frame's content, seed 0
 * (RFC 8878 3.1.1).  The reference keeps this checksum on purpose (PERFORMANCE.md E033, README.md:87). */
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static inline uint64_t xrotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t xread64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t xread32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) { return xrotl(acc + in * XP2, 31) * XP1; }
static inline uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * XP1 + XP4; }
uint64_t fqzo_xxh64(const uint8_t *p, size_t len, uint64_t seed)
{
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t *lim = end - 32;
        do {
            v1 = xround(v1, xread64(p)); v2 = xround(v2, xread64(p + 8)); v3 = xround(v3, xread64(p + 16)); v4 = xround(v4, xread64(p + 24));
            p += 32;
        } while (p <= lim);
        h = xrotl(v1, 1) + xrotl(v2, 7) + xrotl(v3, 12) + xrotl(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else h = seed + XP5;
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xround(0, xread64(p)); h = xrotl(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)xread32(p) * XP1; h = xrotl(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p++) * XP5; h = xrotl(h, 11) * XP1; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* "FQZ-H2" payload of one pre-entropy stream of n bytes (n == 0: a 0-byte payload):
 *   [index]   a zstd SKIPPABLE frame (RFC 8878 3.1.2, magic 0x184D2A50; every zstd decoder, the reference's DecodeAll
 *             included, skips it): 'FQZI', version 1, the stream id, the pre-entropy length, the number of zstd blocks and
 *             the size of each (3 bytes, block header included) - what a parallel decoder needs to find every block
 *             without walking the chain of block headers;
 *   [frames]  one zstd frame per GROUP of FQZO_GROUP chunks (<= 64 KiB of the stream): Single_Segment, Frame_Content_Size,
 *             Content_Checksum; one zstd block per 16 KiB chunk (one Huffman table per group, the later
 *             blocks treeless).  The 2-bit packed bases (stream 0) are Raw blocks by definition.
 * Frames are independent, so encoder and decoder work on every group in parallel and the content checksum (an
 * inherently serial hash) runs over 64 KiB at a time, one hash per four lanes. */
#define FQZO_IDX_HDR 24 /* magic 4 + size 4 + 'FQZI' 4 + version, stream, flags 4 + raw length 4 + block count 4 */

size_t fqzo_entropy_bound(size_t n)
{
    if (!n) return 0;
    const size_t chunks = (n + FQZO_CHUNK - 1) / FQZO_CHUNK, groups = (chunks + FQZO_GROUP - 1) / FQZO_GROUP;
    return FQZO_IDX_HDR + (3 + 2 * FQZO_ENT) * chunks + (4 + 4 * (n / 128 + 1)) /* record samples: a record is >= 2 bytes */ + groups * (7 + 4) + n + 3 * chunks;
}

size_t fqzo_entropy_encode_stream(const uint8_t *src, size_t n, int stream, uint8_t *dst) { return fqzo_entropy_encode_stream_v(src, n, stream, 2, dst); }

size_t fqzo_entropy_encode_stream_v(const uint8_t *src, size_t n, int stream, int version, uint8_t *dst)
{
    if (!n) return 0;
    const size_t chunks = (n + FQZO_CHUNK - 1) / FQZO_CHUNK;
    /* headers / plus / nPos streams are chains of length-prefixed records ([u16 k][k units]; compress.go:507-517): their
     * record starts (a walk over the prefixes); anything that is not a clean chain of records has none */
    uint32_t *rs = NULL, nr = 0;
    if (stream == 2 || stream == 3 || stream == 4) {
        const size_t unit = stream == 4 ? 2 : 1;
        rs = (uint32_t *)malloc(sizeof(uint32_t) * (n / 2 + 2));
        size_t pos = 0;
        while (pos + 2 <= n) { rs[nr++] = (uint32_t)pos; pos += 2 + unit * (size_t)(src[pos] | (src[pos + 1] << 8)); }
        if (pos != n) { free(rs); rs = NULL; nr = 0; } else rs[nr] = (uint32_t)n;
    }
    /* record samples: the stream offset of every 64th record (64, 128, ...), so that a parallel decoder finds the records
     * without walking the chain from the start (it checks them against the walk it does between two samples).  Not for
     * streams whose records are all bare prefixes (offset = 2 x record) */
    const uint32_t ns = (rs && nr > 64 && n != 2 * (size_t)nr) ? (nr - 1) / 64 : 0;
    /* entry points of the Huffman streams (encode_group_chunks): every stream but the packed bases (Raw blocks by definition) and
     * the rANS-coded qualities of a version-3 file */
    const int has_ent = stream != 0 && !(version == 3 && stream == 1);
    const size_t ent_at = FQZO_IDX_HDR + 3 * chunks + (ns ? 4 + 4 * (size_t)ns : 0);
    const size_t idx_len = ent_at + (has_ent ? 2 * FQZO_ENT * chunks : 0);
    uint8_t *idx = dst, *op = dst + idx_len;
    put32le(idx, 0x184D2A50u);
    put32le(idx + 4, (uint32_t)(idx_len - 8));
    idx[8] = 'F'; idx[9] = 'Q'; idx[10] = 'Z'; idx[11] = 'I';
    idx[12] = 1; idx[13] = (uint8_t)stream; idx[14] = (uint8_t)((ns ? 1 : 0) | (has_ent ? 2 : 0)); idx[15] = 0; /* flags: bit 0 = record samples, bit 1 = entry points */
    put32le(idx + 16, (uint32_t)n);
    put32le(idx + 20, (uint32_t)chunks);
    uint8_t *ent = idx + FQZO_IDX_HDR;
    if (ns) {
        uint8_t *sp = idx + FQZO_IDX_HDR + 3 * chunks;
        put32le(sp, nr);
        for (uint32_t k = 1; k <= ns; k++) put32le(sp + 4 * k, rs[64 * k]);
    }
    const size_t G = (size_t)FQZO_GROUP * FQZO_CHUNK;
    if (stream != 2) { free(rs); rs = NULL; } /* (only the headers are modelled against their records) */
    hseq *sqbuf = rs ? (hseq *)malloc(sizeof(hseq) * FQZO_GROUP * HDR_MAX_SEQ) : NULL;
    uint8_t *litbuf = rs ? (uint8_t *)malloc(G) : NULL;
    hseq lseq[FQZO_GROUP]; /* lengths stream: a chunk of equal u32 values = its first value + one match at offset 4 */
    uint16_t gent[FQZO_GROUP][FQZO_ENT];
    uint8_t *entp = idx + ent_at;
    for (size_t off = 0; off < n; off += G) {
        const size_t M = n - off < G ? n - off : G;
        op[0] = 0x28; op[1] = 0xB5; op[2] = 0x2F; op[3] = 0xFD;
        if (M < 256) { op[4] = 0x24; op[5] = (uint8_t)M; op += 6; }                       /* single segment, checksum, FCS 1 byte */
        else { op[4] = 0x64; op[5] = (uint8_t)(M - 256); op[6] = (uint8_t)((M - 256) >> 8); op += 7; } /* FCS 2 bytes, value - 256 */
        size_t body;
        if (rs) {
            gchunk ch[FQZO_GROUP];
            int nch = 0;
            for (size_t co = 0; co < M; co += FQZO_CHUNK, nch++) {
                const uint32_t mk = (uint32_t)(M - co < FQZO_CHUNK ? M - co : FQZO_CHUNK);
                uint32_t nl = 0;
                ch[nch].raw = src + off + co; ch[nch].mk = mk;
                ch[nch].sq = sqbuf + (size_t)nch * HDR_MAX_SEQ;
                ch[nch].lit = litbuf + co;
                ch[nch].nseq = hdr_chunk_model(src, rs, nr, (uint32_t)(off + co), mk, sqbuf + (size_t)nch * HDR_MAX_SEQ, litbuf + co, &nl);
                ch[nch].n_lit = nl;
            }
            body = encode_group_chunks(ch, nch, 0, op, has_ent ? gent : NULL);
        } else if (version == 3 && stream == 1) body = encode_group_rans(src + off, M, op); /* FQZ-R1 */
        else if (stream == 5) {
            /* u32 read lengths (compress.go:501): fixed-length reads make the stream 4-periodic, which an order-0 coder cannot see
             * (1 bit a byte at best) and zstd level 1 codes as one match.  Per 16 KiB chunk (a zstd block of its own, as for the
             * headers): all values equal and at least 20 bytes -> literals = the first value, one sequence {4 literals, match of
             * mk - 4 bytes at offset 4} on the predefined tables; any other chunk is plain. */
            gchunk ch[FQZO_GROUP];
            int nch = 0;
            for (size_t co = 0; co < M; co += FQZO_CHUNK, nch++) {
                const uint32_t mk = (uint32_t)(M - co < FQZO_CHUNK ? M - co : FQZO_CHUNK);
                const uint8_t *c = src + off + co;
                ch[nch].raw = ch[nch].lit = c; ch[nch].mk = ch[nch].n_lit = mk; ch[nch].sq = NULL; ch[nch].nseq = 0;
                if (mk >= 20 && (mk & 3) == 0) {
                    int same = 1;
                    for (uint32_t i = 4; i < mk && same; i++) same = c[i] == c[i - 4];
                    if (same) { lseq[nch].ll = 4; lseq[nch].ml = mk - 4; lseq[nch].off = 4; ch[nch].sq = &lseq[nch]; ch[nch].nseq = 1; ch[nch].n_lit = 4; }
                }
            }
            body = encode_group_chunks(ch, nch, 0, op, has_ent ? gent : NULL);
        } else body = encode_group_ex(src + off, M, 1, stream == 0, op, has_ent ? gent : NULL);
        /* the index lists the size of every zstd block of the group */
        for (size_t q = 0; q < body;) {
            const uint32_t bh = op[q] | ((uint32_t)op[q + 1] << 8) | ((uint32_t)op[q + 2] << 16);
            const uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
            const uint32_t sz = 3 + (type == 1 ? 1u : bs);
            ent[0] = (uint8_t)sz; ent[1] = (uint8_t)(sz >> 8); ent[2] = (uint8_t)(sz >> 16);
            ent += 3;
            q += sz;
        }
        if (has_ent) {
            const size_t gch = (M + FQZO_CHUNK - 1) / FQZO_CHUNK;
            for (size_t k = 0; k < gch; k++)
                for (int j = 0; j < FQZO_ENT; j++) { *entp++ = (uint8_t)gent[k][j]; *entp++ = (uint8_t)(gent[k][j] >> 8); }
        }
        op += body;
        put32le(op, (uint32_t)fqzo_xxh64(src + off, M, 0));
        op += 4;
    }
    free(rs); free(sqbuf); free(litbuf);
    return (size_t)(op - dst);
}

/* ===================================================================== */
/* FQZ-S1: SEGMENT FRAMING of a block's payloads (container version 2)    */
/* ===================================================================== */
/* A block's text is cut into segments of FQZO_SEG_TEXT bytes, counted from the first byte of the block's first record; a
 * record belongs to the segment its first byte ('@') lies in.  Every segment contributes ONE zstd frame to each of the six
 * payloads - the part of that stream its records produce - so that one GPU workgroup turns a segment's text into its six
 * frames without the pre-entropy streams ever leaving the chip.  A payload is
 *   [index]  a zstd skippable frame: magic 0x184D2A50, size, 'FQZI', version 2, stream id, u16 flags (0), pre-entropy bytes of the
 *            whole payload u32, number of segments u32, then per segment { u24 bytes of its frame (0: none), u24 pre-entropy bytes,
 *            u16 records }: a decoder places every frame, every record range and every output range with three scans
 *   [frames] per segment with content: magic, FHD 0x24 / 0x64 (Single_Segment, Content_Checksum, 1- or 2-byte content size), one
 *            zstd block per 16 KiB of content with ONE Huffman table for the frame (FQZ-H2 rules: encode_group_chunks), checksum.
 * Stock decoders read it as what it is: a skippable frame and a run of ordinary frames (DecodeAll semantics, compress.go:785-814).
 * Per stream: packed bases Raw; headers modelled per 16 KiB chunk against the previous record (hdr_chunk_model); a chunk of the
 * lengths stream whose u32 values are all equal is the first value as literals + ONE sequence (literal length 4, match length
 * mk - 4, offset 4); everything else Huffman / RLE / Raw by the FQZ-H2 tests.
 * A block qualifies only if every segment holds at most FQZO_SEG_RMAX records and its six stream parts, each rounded up to 16
 * bytes, fit FQZO_SEG_ARENA bytes (what the workgroup keeps in LDS); any other block is written with the FQZ-H2 framing above. */
size_t fqzo_seg_index_len(uint32_t n_seg) { return 24 + 8 * (size_t)n_seg; }

void fqzo_seg_index_write(uint8_t *idx, int stream, uint32_t raw_total, uint32_t n_seg)
{
    put32le(idx, 0x184D2A50u);
    put32le(idx + 4, (uint32_t)(fqzo_seg_index_len(n_seg) - 8));
    idx[8] = 'F'; idx[9] = 'Q'; idx[10] = 'Z'; idx[11] = 'I';
    idx[12] = 2; idx[13] = (uint8_t)stream; idx[14] = 0; idx[15] = 0;
    put32le(idx + 16, raw_total);
    put32le(idx + 20, n_seg);
}

void fqzo_seg_index_entry(uint8_t *idx, uint32_t seg, uint32_t comp, uint32_t raw, uint32_t nrec)
{
    uint8_t *e = idx + 24 + 8 * (size_t)seg;
    e[0] = (uint8_t)comp; e[1] = (uint8_t)(comp >> 8); e[2] = (uint8_t)(comp >> 16);
    e[3] = (uint8_t)raw; e[4] = (uint8_t)(raw >> 8); e[5] = (uint8_t)(raw >> 16);
    e[6] = (uint8_t)nrec; e[7] = (uint8_t)(nrec >> 8);
}

/* the frame of one segment's part [src, src + n) of stream `stream`; n <= FQZO_GROUP * FQZO_CHUNK.  Returns its size (0 for n == 0) */
size_t fqzo_seg_frame(const uint8_t *src, size_t n, int stream, uint8_t *dst)
{
    if (!n) return 0;
    uint8_t *op = dst;
    op[0] = 0x28; op[1] = 0xB5; op[2] = 0x2F; op[3] = 0xFD;
    if (n < 256) { op[4] = 0x24; op[5] = (uint8_t)n; op += 6; }
    else { op[4] = 0x64; op[5] = (uint8_t)(n - 256); op[6] = (uint8_t)((n - 256) >> 8); op += 7; }
    gchunk ch[FQZO_GROUP];
    int nch = 0;
    uint32_t *rs = NULL, nr = 0;
    hseq *sqbuf = NULL;
    uint8_t *litbuf = NULL;
    if (stream == 2) { /* headers: record starts from the length prefixes (the part is a clean chain of whole records) */
        rs = (uint32_t *)malloc(sizeof(uint32_t) * (n / 2 + 2));
        size_t pos = 0;
        while (pos + 2 <= n) { rs[nr++] = (uint32_t)pos; pos += 2 + (size_t)(src[pos] | (src[pos + 1] << 8)); }
        if (pos != n) { free(rs); rs = NULL; nr = 0; } else rs[nr] = (uint32_t)n;
        if (rs) { sqbuf = (hseq *)malloc(sizeof(hseq) * FQZO_GROUP * HDR_MAX_SEQ); litbuf = (uint8_t *)malloc((size_t)FQZO_GROUP * FQZO_CHUNK); }
    }
    hseq lseq[FQZO_GROUP];
    for (size_t co = 0; co < n; co += FQZO_CHUNK, nch++) {
        const uint32_t mk = (uint32_t)(n - co < FQZO_CHUNK ? n - co : FQZO_CHUNK);
        ch[nch].raw = ch[nch].lit = src + co; ch[nch].mk = ch[nch].n_lit = mk; ch[nch].sq = NULL; ch[nch].nseq = 0;
        if (rs) {
            uint32_t nl = 0;
            ch[nch].sq = sqbuf + (size_t)nch * HDR_MAX_SEQ;
            ch[nch].lit = litbuf + co;
            ch[nch].nseq = hdr_chunk_model(src, rs, nr, (uint32_t)co, mk, sqbuf + (size_t)nch * HDR_MAX_SEQ, litbuf + co, &nl);
            ch[nch].n_lit = nl;
        } else if (stream == 5 && mk >= 20 && (mk & 3) == 0) { /* lengths: all u32 of the chunk equal -> 4 literals + one match at offset 4
                                                                 * (from 20 bytes on: below, the block would be judged worse than Raw) */
            int same = 1;
            for (uint32_t i = 4; i < mk && same; i++) same = src[co + i] == src[co + i - 4];
            if (same) {
                lseq[nch].ll = 4; lseq[nch].ml = mk - 4; lseq[nch].off = 4;
                ch[nch].sq = &lseq[nch]; ch[nch].nseq = 1; ch[nch].n_lit = 4;
            }
        }
    }
    op += encode_group_chunks(ch, nch, stream == 0, op, NULL);
    put32le(op, (uint32_t)fqzo_xxh64(src, n, 0));
    op += 4;
    free(rs); free(sqbuf); free(litbuf);
    return (size_t)(op - dst);
}

size_t fqzo_seg_frame_bound(size_t n) { return n ? n + 7 + 4 + 3 * ((n + FQZO_CHUNK - 1) / FQZO_CHUNK) + 16 : 0; }

/* a stream of no particular kind: Huffman-coded like the quality stream */
size_t fqzo_entropy_encode(const uint8_t *src, size_t n, uint8_t *dst) { return fqzo_entropy_encode_stream(src, n, 1, dst); }

/* ===================================================================== */
/* decoder (subset: no sequences)                                         */
/* ===================================================================== */

/* backward bit reader over [p, p+n) */
typedef struct {
    const uint8_t *p;
    size_t n;
    long pos; /* number of unread bits */
} bitr;
static int br_init(bitr *b, const uint8_t *p, size_t n)
{
    if (!n || !p[n - 1]) return -1;
    b->p = p;
    b->n = n;
    b->pos = (long)(n - 1) * 8 + highbit32(p[n - 1]);
    return 0;
}
/* peek nb (<= 16) bits just below pos (zero-padded below bit 0) */
static inline uint32_t br_peek(const bitr *b, int nb)
{
    long lo = b->pos - nb;
    if (lo >= 0 && (size_t)(lo >> 3) + 4 <= b->n) {
        uint32_t x;
        memcpy(&x, b->p + (lo >> 3), 4); /* little-endian host */
        return (x >> (lo & 7)) & ((1u << nb) - 1);
    }
    uint32_t v = 0;
    for (int i = nb - 1; i >= 0; i--) {
        long bit = lo + i;
        uint32_t x = bit >= 0 ? (uint32_t)(b->p[bit >> 3] >> (bit & 7)) & 1u : 0u;
        v = (v << 1) | x;
    }
    return v;
}
static inline uint32_t br_read(bitr *b, int nb)
{
    uint32_t v = br_peek(b, nb);
    b->pos -= nb;
    return v;
}

/* FSE_readNCount + decode of the weight stream; returns number of weights or -1 */
static int fse_decode_weights(const uint8_t *src, size_t n, uint8_t *w, int cap)
{
    if (n < 2) return -1;
    /* read NCount with a forward bit reader */
    uint64_t acc = 0;
    size_t avail_bits = n * 8, bitpos = 0;
#define NC_PEEK(nb) ((uint32_t)({ uint64_t _v = 0; for (int _i = 0; _i < 4; _i++) { size_t _b = (bitpos >> 3) + (size_t)_i; if (_b < n) _v |= (uint64_t)src[_b] << (8 * _i); } (_v >> (bitpos & 7)) & ((1u << (nb)) - 1); }))
    (void)acc;
    int table_log = (int)NC_PEEK(4) + 5;
    bitpos += 4;
    if (table_log > FSE_W_MAXLOG) return -1;
    int table_size = 1 << table_log;
    int remaining = table_size + 1, threshold = table_size, nb = table_log + 1;
    int norm[256], sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= 255) {
        if (prev0) {
            int n0 = sym;
            for (;;) {
                uint32_t r = NC_PEEK(2);
                bitpos += 2;
                n0 += (int)r;
                if (r != 3) break;
            }
            if (n0 > 255) return -1;
            while (sym < n0) norm[sym++] = 0;
        }
        int max = (2 * threshold - 1) - remaining;
        int count;
        uint32_t lowv = NC_PEEK(nb - 1);
        if ((int)lowv < max) { count = (int)lowv; bitpos += (size_t)(nb - 1); }
        else {
            count = (int)NC_PEEK(nb);
            if (count >= threshold) count -= max;
            bitpos += (size_t)nb;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = count;
        prev0 = !count;
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (bitpos > avail_bits) return -1;
    }
#undef NC_PEEK
    if (remaining != 1) return -1;
    int max_sv = sym - 1;
    size_t hdr = (bitpos + 7) >> 3;
    if (hdr >= n) return -1;

    /* FSE_buildDTable */
    uint8_t dsym[64], dnb[64];
    uint16_t dnew[64];
    {
        uint16_t next[256];
        int high = table_size - 1;
        for (int s = 0; s <= max_sv; s++) {
            if (norm[s] == -1) { dsym[high--] = (uint8_t)s; next[s] = 1; }
            else next[s] = (uint16_t)norm[s];
        }
        int step = (table_size >> 1) + (table_size >> 3) + 3, mask = table_size - 1, pos = 0;
        for (int s = 0; s <= max_sv; s++)
            for (int i = 0; i < norm[s]; i++) {
                dsym[pos] = (uint8_t)s;
                pos = (pos + step) & mask;
                while (pos > high) pos = (pos + step) & mask;
            }
        if (pos != 0) return -1;
        for (int u = 0; u < table_size; u++) {
            int s = dsym[u];
            uint32_t ns = next[s]++;
            dnb[u] = (uint8_t)(table_log - highbit32(ns));
            dnew[u] = (uint16_t)((ns << dnb[u]) - (uint32_t)table_size);
        }
    }
    bitr br;
    if (br_init(&br, src + hdr, n - hdr) < 0) return -1;
    if (br.pos < 2 * table_log) return -1;
    uint32_t s1 = br_read(&br, table_log), s2 = br_read(&br, table_log);
    int out = 0;
    for (;;) {
        /* mirrors FSE_decompress_usingDTable_generic's tail handling */
        if (out >= cap) return -1;
        w[out++] = dsym[s1];
        if (br.pos < dnb[s1]) { if (out >= cap) return -1; w[out++] = dsym[s2]; break; }
        s1 = dnew[s1] + br_read(&br, dnb[s1]);
        if (out >= cap) return -1;
        w[out++] = dsym[s2];
        if (br.pos < dnb[s2]) { if (out >= cap) return -1; w[out++] = dsym[s1]; break; }
        s2 = dnew[s2] + br_read(&br, dnb[s2]);
    }
    return out;
}

typedef struct { uint8_t sym, nb; } hdec;

/* Huffman_Tree_Description -> decode table; returns bytes consumed or -1 */
static long huf_read_table(const uint8_t *src, size_t n, hdec *dt, int *table_log_out)
{
    if (!n) return -1;
    uint8_t w[256];
    int nw;
    size_t used;
    uint8_t hb = src[0];
    if (hb >= 128) {
        nw = hb - 127;
        used = 1 + (size_t)(nw + 1) / 2;
        if (used > n) return -1;
        for (int i = 0; i < nw; i += 2) {
            w[i] = src[1 + i / 2] >> 4;
            if (i + 1 < nw) w[i + 1] = src[1 + i / 2] & 15;
        }
    } else {
        used = 1 + (size_t)hb;
        if (used > n) return -1;
        nw = fse_decode_weights(src + 1, hb, w, 255);
        if (nw < 0) return -1;
    }
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        if (w[i] > 12) return -1;
        total += (1u << w[i]) >> 1;
    }
    if (!total) return -1;
    int table_log = highbit32(total) + 1;
    if (table_log > 12) return -1;
    uint32_t rest = (1u << table_log) - total;
    if (rest & (rest - 1)) return -1; /* must be a power of two */
    w[nw] = (uint8_t)(highbit32(rest) + 1);
    nw++;
    uint32_t rank_start[14] = {0}, rank_cnt[14] = {0};
    for (int i = 0; i < nw; i++) rank_cnt[w[i]]++;
    if (rank_cnt[1] < 2 || (rank_cnt[1] & 1)) return -1;
    uint32_t next = 0;
    for (int r = 1; r <= table_log; r++) { rank_start[r] = next; next += rank_cnt[r] << (r - 1); }
    for (int s = 0; s < nw; s++) {
        if (!w[s]) continue;
        uint32_t len = (1u << w[s]) >> 1;
        for (uint32_t u = 0; u < len; u++) { dt[rank_start[w[s]] + u].sym = (uint8_t)s; dt[rank_start[w[s]] + u].nb = (uint8_t)(table_log + 1 - w[s]); }
        rank_start[w[s]] += len;
    }
    *table_log_out = table_log;
    return (long)used;
}

static int huf_decode_stream(const uint8_t *src, size_t n, const hdec *dt, int table_log, uint8_t *dst, size_t count)
{
    bitr br;
    if (br_init(&br, src, n) < 0) return -1;
    for (size_t i = 0; i < count; i++) {
        uint32_t v = br_peek(&br, table_log);
        dst[i] = dt[v].sym;
        br.pos -= dt[v].nb;
        if (br.pos < 0) return -1;
    }
    return br.pos == 0 ? 0 : -1;
}

/* literals section of a Compressed block; block must regenerate exactly the literals */
/* the Huffman table of the previous Compressed literals of the frame, for treeless blocks */
typedef struct { hdec dt[4096]; int table_log; int valid; } huf_state;

/* Sequences_Section with Predefined_Mode tables (what the header modelling above writes; other modes are left to
 * libzstd by the callers): dst[0 .. n_lit) holds the literals; the block is rebuilt in place from the end of a copy. */
typedef struct { uint8_t sym, nb; uint16_t base; } fse_de;
static void fse_build_dt(fse_de *dt, const short *norm, int nsym, int log)
{
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint16_t next[64];
    int high = size - 1;
    for (int sy = 0; sy < nsym; sy++) {
        if (norm[sy] == -1) { dt[high--].sym = (uint8_t)sy; next[sy] = 1; }
        else next[sy] = (uint16_t)norm[sy];
    }
    int pos = 0;
    for (int sy = 0; sy < nsym; sy++)
        for (int k = 0; k < norm[sy]; k++) {
            dt[pos].sym = (uint8_t)sy;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    for (int u = 0; u < size; u++) {
        const uint32_t ns = next[dt[u].sym]++;
        dt[u].nb = (uint8_t)(log - highbit32(ns));
        dt[u].base = (uint16_t)((ns << dt[u].nb) - (uint32_t)size);
    }
}
static fse_de LLD[64], MLD[64], OFD[32]; /* decoding tables of the predefined distributions */
static void hdr_dt_init(void) { fse_build_dt(LLD, LL_NORM, 36, 6); fse_build_dt(MLD, ML_NORM, 53, 6); fse_build_dt(OFD, OF_NORM, 29, 5); }
static long exec_sequences(const uint8_t *sec, size_t n, uint8_t *dst, size_t n_lit, size_t cap)
{
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, hdr_dt_init);
    size_t p = 0;
    uint32_t nseq = sec[p++];
    if (nseq >= 128) { if (p >= n) return FQZO_E_ENTROPY; if (nseq == 255) return FQZO_E_ENTROPY; nseq = ((nseq - 128) << 8) + sec[p++]; }
    if (p >= n || sec[p++] != 0) return FQZO_E_ENTROPY; /* Predefined_Mode for all three */
    bitr br;
    if (br_init(&br, sec + p, n - p) < 0) return FQZO_E_ENTROPY;
    uint8_t *lit = (uint8_t *)malloc(n_lit ? n_lit : 1);
    if (!lit) return FQZO_E_ENTROPY;
    memcpy(lit, dst, n_lit);
    uint32_t st_ll = br_read(&br, 6), st_of = br_read(&br, 5), st_ml = br_read(&br, 6);
    uint32_t rep[3] = {1, 4, 8};
    size_t out = 0, lpos = 0;
    long err = 0;
    for (uint32_t i = 0; i < nseq; i++) {
        const int oc = OFD[st_of].sym, mc = MLD[st_ml].sym, lc = LLD[st_ll].sym;
        const uint32_t of_val = (1u << oc) + br_read(&br, oc);
        const uint32_t ml = ML_BASE[mc] + br_read(&br, ML_BITS[mc]);
        const uint32_t ll = LL_BASE[lc] + br_read(&br, LL_BITS[lc]);
        if (i + 1 < nseq) {
            st_ll = LLD[st_ll].base + br_read(&br, LLD[st_ll].nb);
            st_ml = MLD[st_ml].base + br_read(&br, MLD[st_ml].nb);
            st_of = OFD[st_of].base + br_read(&br, OFD[st_of].nb);
        }
        /* FQZ-H2 blocks are self-contained: an explicit offset, or "the offset of the previous sequence of this block"
         * (Offset_Value 1 with literals).  Anything else needs the offset history of earlier blocks: not this subset */
        uint32_t offset;
        if (of_val > 3) offset = of_val - 3;
        else if (of_val == 1 && ll > 0 && i > 0) offset = rep[0];
        else { err = FQZO_E_ENTROPY; break; }
        rep[0] = offset;
        if (ll > n_lit - lpos || out + ll + ml > cap || offset == 0 || offset > out + ll) { err = FQZO_E_ENTROPY; break; }
        memcpy(dst + out, lit + lpos, ll);
        out += ll; lpos += ll;
        for (uint32_t k = 0; k < ml; k++) dst[out + k] = dst[out + k - offset];
        out += ml;
    }
    if (!err) {
        if (out + (n_lit - lpos) > cap) err = FQZO_E_DST_SMALL;
        else { memcpy(dst + out, lit + lpos, n_lit - lpos); out += n_lit - lpos; }
    }
    free(lit);
    return err ? err : (long)out;
}

static long decode_compressed_block(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, huf_state *hs)
{
    if (n < 2) return FQZO_E_ENTROPY;
    int type = src[0] & 3, fmt = (src[0] >> 2) & 3;
    size_t lh, regen, csize = 0;
    int nstreams = 1;
    if (type == 0 || type == 1) { /* Raw / RLE literals */
        if (fmt == 0 || fmt == 2) { lh = 1; regen = src[0] >> 3; }
        else if (fmt == 1) { lh = 2; regen = (src[0] >> 4) | ((size_t)src[1] << 4); }
        else { if (n < 3) return FQZO_E_ENTROPY; lh = 3; regen = (src[0] >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); }
        if (regen > cap) return FQZO_E_DST_SMALL;
        if (type == 0) {
            if (lh + regen + 1 > n) return FQZO_E_ENTROPY;
            memcpy(dst, src + lh, regen);
            csize = regen;
        } else {
            if (lh + 1 + 1 > n) return FQZO_E_ENTROPY;
            memset(dst, src[lh], regen);
            csize = 1;
        }
    } else { /* Compressed (2) or treeless (3): same size fields */
        if (fmt <= 1) {
            if (n < 3) return FQZO_E_ENTROPY;
            uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16);
            lh = 3; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; nstreams = fmt ? 4 : 1;
        } else if (fmt == 2) {
            if (n < 4) return FQZO_E_ENTROPY;
            uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
            lh = 4; regen = (v >> 4) & 0x3FFF; csize = v >> 18; nstreams = 4;
        } else {
            if (n < 5) return FQZO_E_ENTROPY;
            uint64_t v = src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24) | ((uint64_t)src[4] << 32);
            lh = 5; regen = (size_t)((v >> 4) & 0x3FFFF); csize = (size_t)(v >> 22); nstreams = 4;
        }
        if (regen > cap) return FQZO_E_DST_SMALL;
        if (lh + csize + 1 > n) return FQZO_E_ENTROPY;
        const uint8_t *ip = src + lh;
        long used = 0;
        if (type == 2) {
            used = huf_read_table(ip, csize, hs->dt, &hs->table_log);
            if (used < 0) return FQZO_E_ENTROPY;
            hs->valid = 1;
        } else if (!hs->valid) return FQZO_E_ENTROPY; /* treeless without a previous table */
        const hdec *dt = hs->dt;
        const int table_log = hs->table_log;
        ip += used;
        size_t rem = csize - (size_t)used;
        if (nstreams == 1) {
            if (huf_decode_stream(ip, rem, dt, table_log, dst, regen) < 0) return FQZO_E_ENTROPY;
        } else {
            if (rem < 10) return FQZO_E_ENTROPY;
            size_t s1 = ip[0] | ((size_t)ip[1] << 8), s2 = ip[2] | ((size_t)ip[3] << 8), s3 = ip[4] | ((size_t)ip[5] << 8);
            if (6 + s1 + s2 + s3 >= rem) return FQZO_E_ENTROPY;
            size_t s4 = rem - 6 - s1 - s2 - s3;
            size_t seg = (regen + 3) / 4;
            if (3 * seg > regen) return FQZO_E_ENTROPY;
            const uint8_t *p = ip + 6;
            if (huf_decode_stream(p, s1, dt, table_log, dst, seg) < 0) return FQZO_E_ENTROPY;
            if (huf_decode_stream(p + s1, s2, dt, table_log, dst + seg, seg) < 0) return FQZO_E_ENTROPY;
            if (huf_decode_stream(p + s1 + s2, s3, dt, table_log, dst + 2 * seg, seg) < 0) return FQZO_E_ENTROPY;
            if (huf_decode_stream(p + s1 + s2 + s3, s4, dt, table_log, dst + 3 * seg, regen - 3 * seg) < 0) return FQZO_E_ENTROPY;
        }
    }
    /* sequences section */
    if (lh + csize + 1 > n) return FQZO_E_ENTROPY;
    if (src[lh + csize] == 0) return lh + csize + 1 == n ? (long)regen : FQZO_E_ENTROPY; /* 0 sequences: the literals are the block */
    return exec_sequences(src + lh + csize, n - lh - csize, dst, regen, cap);
}

static int parse_frame_header(const uint8_t *src, size_t n, size_t *hdr_size, long *fcs, int *has_checksum)
{
    if (n < 6) return -1;
    if (!(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD)) return -1;
    uint8_t fhd = src[4];
    int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
    if (fhd & 0x08) return -1; /* reserved bit */
    *has_checksum = (fhd >> 2) & 1;
    size_t p = 5;
    if (!single) p += 1;
    static const int dict_sz[4] = {0, 1, 2, 4};
    p += (size_t)dict_sz[dict];
    int fcs_sz = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (p + (size_t)fcs_sz > n) return -1;
    *fcs = -1;
    if (fcs_sz) {
        uint64_t v = 0;
        for (int i = 0; i < fcs_sz; i++) v |= (uint64_t)src[p + (size_t)i] << (8 * i);
        if (fcs_sz == 2) v += 256;
        *fcs = (long)v;
    }
    *hdr_size = p + (size_t)fcs_sz;
    return 0;
}

static int is_skippable(const uint8_t *p, size_t n)
{
    return n >= 8 && (p[0] & 0xF0) == 0x50 && p[1] == 0x2A && p[2] == 0x4D && p[3] == 0x18;
}

/* pre-entropy bytes of a payload: the sum of the Frame_Content_Size fields of its frames (every frame must state it) */
long fqzo_entropy_content_size(const uint8_t *src, size_t n)
{
    size_t ip = 0;
    uint64_t total = 0;
    while (ip < n) {
        if (is_skippable(src + ip, n - ip)) {
            const uint32_t sz = src[ip + 4] | ((uint32_t)src[ip + 5] << 8) | ((uint32_t)src[ip + 6] << 16) | ((uint32_t)src[ip + 7] << 24);
            if ((uint64_t)sz + 8 > n - ip) return FQZO_E_ENTROPY;
            ip += 8 + (size_t)sz;
            continue;
        }
        size_t h; long fcs; int ck;
        if (parse_frame_header(src + ip, n - ip, &h, &fcs, &ck) < 0 || fcs < 0) return FQZO_E_ENTROPY;
        ip += h;
        for (;;) {
            if (ip + 3 > n) return FQZO_E_ENTROPY;
            const uint32_t bh = src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
            const uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
            const size_t adv = 3 + (type == 1 ? 1u : bs);
            if (adv > n - ip) return FQZO_E_ENTROPY; /* (type 3: a FQZ-R1 block, sized like a Compressed block) */
            ip += adv;
            if (bh & 1) break;
        }
        if (ck) { if (ip + 4 > n) return FQZO_E_ENTROPY; ip += 4; }
        total += (uint64_t)fcs;
        if (total > 0x7FFFFFFFull) return FQZO_E_ENTROPY;
    }
    return (long)total;
}

/* FQZ-R1 block (see the encoder): returns the bytes regenerated or a negative error */
typedef struct { int valid; uint16_t freq[256], cum[256]; uint8_t slot[4096]; } rans_dtab;
static long rans_decode_block(const uint8_t *src, size_t bs, uint8_t *dst, size_t cap, rans_dtab *t)
{
    if (bs < 3 + 64) return FQZO_E_ENTROPY;
    const uint32_t m = src[0] | ((uint32_t)src[1] << 8);
    if (m < 1 || m > FQZO_CHUNK || (src[2] & ~1u)) return FQZO_E_ENTROPY;
    if (m > cap) return FQZO_E_DST_SMALL;
    size_t p = 3;
    if (src[2] & 1) {
        if (t->valid) return FQZO_E_ENTROPY; /* one table a frame */
        const uint32_t nsym = (uint32_t)src[p] + 1;
        const size_t tl = 1 + 3 * (size_t)nsym + ((nsym & 1) ? 0 : 1);
        if (nsym < 2 || p + tl + 64 > bs) return FQZO_E_ENTROPY;
        const uint8_t *sy = src + p + 1, *fr = sy + nsym;
        uint32_t c = 0;
        memset(t->freq, 0, sizeof t->freq);
        for (uint32_t k = 0; k < nsym; k++) {
            const uint32_t f = fr[2 * k] | ((uint32_t)fr[2 * k + 1] << 8);
            if ((k && sy[k] <= sy[k - 1]) || f < 1 || f > 4095 || c + f > 4096) return FQZO_E_ENTROPY;
            t->freq[sy[k]] = (uint16_t)f; t->cum[sy[k]] = (uint16_t)c;
            memset(t->slot + c, sy[k], f);
            c += f;
        }
        if (c != 4096 || (!(nsym & 1) && src[p + tl - 1] != 0)) return FQZO_E_ENTROPY;
        t->valid = 1;
        p += tl;
    } else if (!t->valid) return FQZO_E_ENTROPY;
    const size_t states = bs - 64;
    if ((states - p) & 1) return FQZO_E_ENTROPY;
    uint32_t x[16];
    for (int j = 0; j < 16; j++) {
        x[j] = src[states + 4 * j] | ((uint32_t)src[states + 4 * j + 1] << 8) | ((uint32_t)src[states + 4 * j + 2] << 16) | ((uint32_t)src[states + 4 * j + 3] << 24);
        if (x[j] < RANS_L) return FQZO_E_ENTROPY;
    }
    size_t wp = states;
    const uint32_t steps = 16 * ((m + 255) / 256);
    for (uint32_t st = 0; st < steps; st++)
        for (int j = 0; j < 16; j++) {
            const uint32_t i = ((st >> 4) << 8) | ((uint32_t)j << 4) | (st & 15);
            if (i >= m) continue;
            const uint32_t sl = x[j] & 4095u, sy = t->slot[sl];
            dst[i] = (uint8_t)sy;
            x[j] = t->freq[sy] * (x[j] >> 12) + sl - t->cum[sy];
            if (x[j] < RANS_L) {
                if (wp < p + 2) return FQZO_E_ENTROPY;
                wp -= 2;
                x[j] = (x[j] << 16) | src[wp] | ((uint32_t)src[wp + 1] << 8);
            }
        }
    if (wp != p) return FQZO_E_ENTROPY;
    for (int j = 0; j < 16; j++) if (x[j] != RANS_L) return FQZO_E_ENTROPY;
    return (long)m;
}

long fqzo_entropy_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
    size_t out = 0;
    size_t ip = 0;
    while (ip < n) { /* concatenated frames are legal; DecodeAll decodes them all and skips skippable frames */
        if (is_skippable(src + ip, n - ip)) {
            const uint32_t sz = src[ip + 4] | ((uint32_t)src[ip + 5] << 8) | ((uint32_t)src[ip + 6] << 16) | ((uint32_t)src[ip + 7] << 24);
            if ((uint64_t)sz + 8 > n - ip) return FQZO_E_ENTROPY;
            ip += 8 + (size_t)sz;
            continue;
        }
        size_t h; long fcs; int ck;
        if (parse_frame_header(src + ip, n - ip, &h, &fcs, &ck) < 0) return FQZO_E_ENTROPY;
        ip += h;
        size_t frame_start = out;
        huf_state *hs = (huf_state *)calloc(1, sizeof(huf_state));
        if (!hs) return FQZO_E_ENTROPY;
        rans_dtab *rt = NULL;
        long err = 0;
        for (;;) {
            if (ip + 3 > n) { err = FQZO_E_ENTROPY; break; }
            uint32_t bh = src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
            ip += 3;
            int last = bh & 1, type = (bh >> 1) & 3;
            size_t bs = bh >> 3;
            if (type == 0) {
                if (ip + bs > n) { err = FQZO_E_ENTROPY; break; }
                if (out + bs > cap) { err = FQZO_E_DST_SMALL; break; }
                memcpy(dst + out, src + ip, bs);
                ip += bs; out += bs;
            } else if (type == 1) {
                if (ip + 1 > n) { err = FQZO_E_ENTROPY; break; }
                if (out + bs > cap) { err = FQZO_E_DST_SMALL; break; }
                memset(dst + out, src[ip], bs);
                ip += 1; out += bs;
            } else if (type == 2) {
                if (ip + bs > n || bs > 128 * 1024) { err = FQZO_E_ENTROPY; break; }
                long r = decode_compressed_block(src + ip, bs, dst + out, cap - out, hs);
                if (r < 0) { err = r; break; }
                ip += bs; out += (size_t)r;
            } else { /* type 3: FQZ-R1 (version-3 files) */
                if (ip + bs > n) { err = FQZO_E_ENTROPY; break; }
                if (!rt && !(rt = (rans_dtab *)calloc(1, sizeof(rans_dtab)))) { err = FQZO_E_ENTROPY; break; }
                long r = rans_decode_block(src + ip, bs, dst + out, cap - out, rt);
                if (r < 0) { err = r; break; }
                ip += bs; out += (size_t)r;
            }
            if (last) break;
        }
        free(hs);
        free(rt);
        if (err) return err;
        if (ck) { /* Content_Checksum: low 32 bits of XXH64 over the frame's content; a mismatch is an error */
            if (ip + 4 > n) return FQZO_E_ENTROPY;
            const uint32_t want = src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
            if ((uint32_t)fqzo_xxh64(dst + frame_start, out - frame_start, 0) != want) return FQZO_E_ENTROPY;
            ip += 4;
        }
        if (fcs >= 0 && (size_t)fcs != out - frame_start) return FQZO_E_ENTROPY;
    }
    return (long)out;
}
