/*
 * fqz_oracle.c — CPU restatement of fqpack's per-block encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see fqz_oracle.h).  Each function cites the
 * reference lines (relative to /root/reference) it restates.
 */
#define _GNU_SOURCE
#include "fqz_oracle.h"

#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ===================================================================== */
/* internal/encoder/sequence.go                                           */
/* ===================================================================== */

/* sequence.go:15-51: baseLookup (A/a=0 C/c=1 G/g=2 T/t=3, everything else 0),
 * isNBase (1 for every byte outside ACGTacgt). */
static uint8_t base_lookup[256], is_n_base[256];
static pthread_once_t tables_once = PTHREAD_ONCE_INIT;
static void init_tables(void)
{
    memset(base_lookup, 0, sizeof base_lookup);
    base_lookup['C'] = base_lookup['c'] = 1;
    base_lookup['G'] = base_lookup['g'] = 2;
    base_lookup['T'] = base_lookup['t'] = 3;
    memset(is_n_base, 1, sizeof is_n_base);
    for (const char *p = "ACGTacgt"; *p; p++) is_n_base[(uint8_t)*p] = 0;
}

size_t fqzo_pack_bases(const uint8_t *seq, size_t n, uint8_t *packed, uint16_t *npos)
{
    pthread_once(&tables_once, init_tables);
    /* sequence.go:151-170: 4 bases per byte, base j in bits 2*(j%4) */
    size_t full = n >> 2;
    for (size_t i = 0; i < full; i++) {
        const uint8_t *b = seq + (i << 2);
        packed[i] = (uint8_t)(base_lookup[b[0]] | (base_lookup[b[1]] << 2) | (base_lookup[b[2]] << 4) | (base_lookup[b[3]] << 6));
    }
    size_t rem = n & 3;
    if (rem) {
        uint8_t v = 0;
        for (size_t j = 0; j < rem; j++) v |= (uint8_t)(base_lookup[seq[(full << 2) + j]] << (j << 1));
        packed[full] = v;
    }
    /* sequence.go:172-181: N scan bounded by MaxSequenceLength */
    size_t limit = n > FQZO_MAX_SEQUENCE_LENGTH ? FQZO_MAX_SEQUENCE_LENGTH : n, k = 0;
    for (size_t i = 0; i < limit; i++)
        if (is_n_base[seq[i]]) npos[k++] = (uint16_t)i;
    return k;
}

int fqzo_unpack_bases(const uint8_t *packed, const uint16_t *npos, size_t n_npos, size_t seq_len, uint8_t *seq)
{
    static const uint8_t bases[4] = {'A', 'C', 'G', 'T'};
    /* sequence.go:199-216 */
    for (size_t i = 0; i < seq_len; i++) seq[i] = bases[(packed[i >> 2] >> ((i & 3) << 1)) & 3];
    /* sequence.go:218-220 */
    for (size_t k = 0; k < n_npos; k++) {
        if (npos[k] >= seq_len) return -1;
        seq[npos[k]] = 'N';
    }
    return 0;
}

/* ===================================================================== */
/* internal/encoder/quality.go                                            */
/* ===================================================================== */

int fqzo_detect_encoding(const uint8_t *const *quals, const size_t *lens, size_t n)
{
    /* quality.go:22-49 */
    uint8_t min = 255;
    for (size_t r = 0; r < n; r++)
        for (size_t i = 0; i < lens[r]; i++) {
            uint8_t b = quals[r][i];
            if (b < min) min = b;
            if (b < 59) return FQZO_PHRED33;
        }
    if (min == 255) return FQZO_PHRED33;
    if (min >= 64) return FQZO_PHRED64;
    return FQZO_PHRED33;
}

static inline uint8_t phred_offset(int enc) { return enc == FQZO_PHRED64 ? 64 : 33; } /* quality.go:5-6 */

void fqzo_normalize_quality(uint8_t *q, size_t n, int enc)
{
    uint8_t off = phred_offset(enc);
    for (size_t i = 0; i < n; i++) q[i] = (uint8_t)(q[i] - off); /* quality.go:59-61 */
}
void fqzo_denormalize_quality(uint8_t *q, size_t n, int enc)
{
    uint8_t off = phred_offset(enc);
    for (size_t i = 0; i < n; i++) q[i] = (uint8_t)(q[i] + off); /* quality.go:72-74 */
}
void fqzo_delta_encode(uint8_t *q, size_t n)
{
    if (n <= 1) return;                                                  /* quality.go:82-84 */
    for (size_t i = n - 1; i > 0; i--) q[i] = (uint8_t)(q[i] - q[i - 1]); /* quality.go:86-102 (backwards, in place) */
}
void fqzo_delta_decode(uint8_t *q, size_t n)
{
    if (n <= 1) return; /* quality.go:108-110 */
    uint8_t acc = q[0];
    for (size_t i = 1; i < n; i++) { acc = (uint8_t)(acc + q[i]); q[i] = acc; } /* quality.go:113-117 */
}

/* ===================================================================== */
/* internal/fqformat/container.go                                         */
/* ===================================================================== */

static inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline uint32_t get32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static inline uint32_t get16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }

void fqzo_write_file_header(const fqzo_file_header *h, uint8_t out[10])
{
    out[0] = 'F'; out[1] = 'Q'; out[2] = 'Z'; out[3] = 0; /* container.go:11 */
    out[4] = h->version;                                     /* container.go:40 */
    put32(out + 5, h->block_size);
    out[9] = h->flags;
}

int fqzo_read_file_header(const uint8_t *in, size_t n, fqzo_file_header *h)
{
    if (n < 4) return FQZO_E_SHORT;
    if (!(in[0] == 'F' && in[1] == 'Q' && in[2] == 'Z' && in[3] == 0)) return FQZO_E_MAGIC; /* container.go:53-55 */
    if (n < 10) return FQZO_E_SHORT;
    h->version = in[4];
    h->block_size = get32(in + 5);
    h->flags = in[9];
    return FQZO_OK;
}

int fqzo_write_block_header(const fqzo_block_header *b, uint8_t version, uint8_t *out)
{
    if (version == 1) { /* container.go:85-96 */
        put32(out + 0, b->num_records); put32(out + 4, b->seq_size); put32(out + 8, b->qual_size);
        put32(out + 12, b->header_size); put32(out + 16, b->npos_size); put32(out + 20, b->lengths_size);
        put32(out + 24, b->original_seq_size); put32(out + 28, b->original_qual_size);
        return 32;
    }
    if (version == 2 || version == 3) { /* container.go:97-109 (version 3 keeps the 36-byte header) */
        put32(out + 0, b->num_records); put32(out + 4, b->seq_size); put32(out + 8, b->qual_size);
        put32(out + 12, b->header_size); put32(out + 16, b->plus_size); put32(out + 20, b->npos_size);
        put32(out + 24, b->lengths_size); put32(out + 28, b->original_seq_size); put32(out + 32, b->original_qual_size);
        return 36;
    }
    return FQZO_E_BLOCK_VERSION;
}

int fqzo_read_block_header(const uint8_t *in, size_t n, uint8_t version, fqzo_block_header *b)
{
    memset(b, 0, sizeof *b);
    if (version == 1) { /* container.go:118-132: PlusDataSize stays 0 */
        if (n < 32) return FQZO_E_SHORT;
        b->num_records = get32(in); b->seq_size = get32(in + 4); b->qual_size = get32(in + 8);
        b->header_size = get32(in + 12); b->npos_size = get32(in + 16); b->lengths_size = get32(in + 20);
        b->original_seq_size = get32(in + 24); b->original_qual_size = get32(in + 28);
        return 32;
    }
    if (version == 2 || version == 3) { /* container.go:133-148 */
        if (n < 36) return FQZO_E_SHORT;
        b->num_records = get32(in); b->seq_size = get32(in + 4); b->qual_size = get32(in + 8);
        b->header_size = get32(in + 12); b->plus_size = get32(in + 16); b->npos_size = get32(in + 20);
        b->lengths_size = get32(in + 24); b->original_seq_size = get32(in + 28); b->original_qual_size = get32(in + 32);
        return 36;
    }
    return FQZO_E_BLOCK_VERSION;
}

const char *fqzo_strerror(int code)
{
    switch (code) {
    case FQZO_OK: return "ok";
    case FQZO_E_SHORT: return "unexpected EOF";
    case FQZO_E_MAGIC: return "invalid magic bytes: not an FQZ file";
    case FQZO_E_BLOCK_VERSION: return "unsupported block header version";
    case FQZO_E_FILE_VERSION: return "unsupported file version";
    case FQZO_E_HDR_AT: return "invalid FASTQ: header line must start with @";
    case FQZO_E_SEP_PLUS: return "invalid FASTQ: separator line must start with +";
    case FQZO_E_LEN_MISMATCH: return "invalid FASTQ: sequence and quality lengths must match";
    case FQZO_E_LONG_N: return "sequence has ambiguous bases beyond position 65536; N-position tracking is limited to 65536 bp";
    case FQZO_E_TRUNC_HEADER: return "truncated header data";
    case FQZO_E_TRUNC_PLUS: return "truncated plus-line payload data";
    case FQZO_E_TRUNC_SEQ: return "truncated sequence data";
    case FQZO_E_TRUNC_QUAL: return "truncated quality data";
    case FQZO_E_TRUNC_LEN: return "truncated length data";
    case FQZO_E_TRUNC_NPOS: return "truncated N position data";
    case FQZO_E_ENTROPY: return "decompressing stream: invalid or unsupported zstd frame";
    case FQZO_E_READ_DATA: return "reading compressed data: unexpected EOF";
    case FQZO_E_NOMEM: return "out of memory";
    case FQZO_E_DST_SMALL: return "destination buffer too small";
    case FQZO_E_FIELD_WRAP: return "header, plus-line payload or N count exceeds 65535 (u16 field would wrap)";
    case FQZO_E_NPOS_RANGE: return "N position beyond read length";
    default: return "unknown error";
    }
}

/* ===================================================================== */
/* internal/fqparser/parser.go                                            */
/* ===================================================================== */

/* parser.go:209-243 readLine: returns 0 and the line (without '\n', and
 * without one trailing '\r'), or -1 for io.EOF (no '\n' left: any partial
 * tail is discarded, parser.go:210-220 — ReadSlice returns the data with
 * io.EOF and readLine drops it). */
static int read_line(const uint8_t *text, size_t n, size_t *pos, uint32_t *off, uint32_t *len)
{
    size_t p = *pos;
    if (p >= n) return -1;
    const uint8_t *nl = memchr(text + p, '\n', n - p);
    if (!nl) { *pos = n; return -1; }
    size_t e = (size_t)(nl - text);
    size_t l = e - p;
    if (l > 0 && text[e - 1] == '\r') l--;
    *off = (uint32_t)p;
    *len = (uint32_t)l;
    *pos = e + 1;
    return 0;
}

long fqzo_parse_batch(const uint8_t *text, size_t n, size_t *pos, fqzo_record *recs, size_t max_records, int *eof)
{
    *eof = 0;
    size_t i = 0;
    for (; i < max_records; i++) {
        fqzo_record r;
        uint32_t off, len;
        /* parser.go:136-183 nextInto */
        if (read_line(text, n, pos, &off, &len) < 0) goto hit_eof;
        if (len == 0 || text[off] != '@') return FQZO_E_HDR_AT; /* parser.go:142-144 */
        r.hdr_off = off + 1; r.hdr_len = len - 1;
        if (read_line(text, n, pos, &off, &len) < 0) goto hit_eof;
        r.seq_off = off; r.seq_len = len;
        if (read_line(text, n, pos, &off, &len) < 0) goto hit_eof;
        if (len == 0 || text[off] != '+') return FQZO_E_SEP_PLUS; /* parser.go:163-165 */
        r.plus_off = off + 1; r.plus_len = len - 1;
        if (read_line(text, n, pos, &off, &len) < 0) goto hit_eof;
        r.qual_off = off; r.qual_len = len;
        if (r.seq_len != r.qual_len) return FQZO_E_LEN_MISMATCH; /* parser.go:179-181 */
        recs[i] = r;
    }
    return (long)i;
hit_eof:
    /* parser.go:196-199: EOF after >=1 record is swallowed by ReadBatch; the
     * caller sees it on the next call (i == 0). */
    *eof = 1;
    *pos = n;
    return (long)i;
}

/* ===================================================================== */
/* compress.go:471-520 — six pre-entropy streams                          */
/* ===================================================================== */

void fqzo_streams_free(fqzo_streams *s)
{
    for (int k = 0; k < FQZO_NSTREAMS; k++) { free(s->data[k]); s->data[k] = NULL; s->len[k] = 0; }
}

static int is_acgt(uint8_t b)
{
    return b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'a' || b == 'c' || b == 'g' || b == 't';
}

int fqzo_split_block(const uint8_t *text, const fqzo_record *recs, size_t n_rec, int enc, fqzo_streams *s)
{
    pthread_once(&tables_once, init_tables);
    memset(s, 0, sizeof *s);
    /* pass 1: sizes */
    size_t sz[FQZO_NSTREAMS] = {0};
    for (size_t i = 0; i < n_rec; i++) {
        const fqzo_record *r = &recs[i];
        const uint8_t *seq = text + r->seq_off;
        /* compress.go:477-488: long-read guard */
        if (r->seq_len > FQZO_MAX_SEQUENCE_LENGTH)
            for (size_t j = FQZO_MAX_SEQUENCE_LENGTH; j < r->seq_len; j++)
                if (!is_acgt(seq[j])) return FQZO_E_LONG_N;
        size_t limit = r->seq_len > FQZO_MAX_SEQUENCE_LENGTH ? FQZO_MAX_SEQUENCE_LENGTH : r->seq_len, nn = 0;
        for (size_t j = 0; j < limit; j++) nn += is_n_base[seq[j]];
        /* App. B-6: the reference lets these u16 fields wrap silently; we reject (documented deviation) */
        if (nn > 65535 || r->hdr_len > 65535 || r->plus_len > 65535) return FQZO_E_FIELD_WRAP;
        sz[FQZO_S_SEQ] += (r->seq_len + 3) >> 2;
        sz[FQZO_S_QUAL] += r->qual_len;
        sz[FQZO_S_HEADERS] += 2 + r->hdr_len;
        sz[FQZO_S_PLUS] += 2 + r->plus_len;
        sz[FQZO_S_NPOS] += 2 + 2 * nn;
        sz[FQZO_S_LENGTHS] += 4;
    }
    for (int k = 0; k < FQZO_NSTREAMS; k++) {
        s->data[k] = malloc(sz[k] ? sz[k] : 1);
        if (!s->data[k]) { fqzo_streams_free(s); return FQZO_E_NOMEM; }
    }
    uint16_t *npos = malloc(sizeof(uint16_t) * FQZO_MAX_SEQUENCE_LENGTH);
    if (!npos) { fqzo_streams_free(s); return FQZO_E_NOMEM; }
    /* pass 2: the per-record loop, compress.go:474-520 */
    size_t o[FQZO_NSTREAMS] = {0};
    for (size_t i = 0; i < n_rec; i++) {
        const fqzo_record *r = &recs[i];
        /* compress.go:491-492 */
        size_t nn = fqzo_pack_bases(text + r->seq_off, r->seq_len, s->data[FQZO_S_SEQ] + o[FQZO_S_SEQ], npos);
        o[FQZO_S_SEQ] += (r->seq_len + 3) >> 2;
        /* compress.go:495-498 */
        uint8_t *np = s->data[FQZO_S_NPOS] + o[FQZO_S_NPOS];
        put16(np, (uint32_t)nn);
        for (size_t k = 0; k < nn; k++) put16(np + 2 + 2 * k, npos[k]);
        o[FQZO_S_NPOS] += 2 + 2 * nn;
        /* compress.go:501 */
        put32(s->data[FQZO_S_LENGTHS] + o[FQZO_S_LENGTHS], r->seq_len);
        o[FQZO_S_LENGTHS] += 4;
        s->original_seq_size += r->seq_len; /* compress.go:503 */
        /* compress.go:506-511 */
        uint8_t *q = s->data[FQZO_S_QUAL] + o[FQZO_S_QUAL];
        memcpy(q, text + r->qual_off, r->qual_len);
        fqzo_normalize_quality(q, r->qual_len, enc);
        fqzo_delta_encode(q, r->qual_len);
        o[FQZO_S_QUAL] += r->qual_len;
        s->original_qual_size += r->qual_len;
        /* compress.go:514-515 */
        uint8_t *h = s->data[FQZO_S_HEADERS] + o[FQZO_S_HEADERS];
        put16(h, r->hdr_len);
        memcpy(h + 2, text + r->hdr_off, r->hdr_len);
        o[FQZO_S_HEADERS] += 2 + r->hdr_len;
        /* compress.go:518-519 */
        uint8_t *pl = s->data[FQZO_S_PLUS] + o[FQZO_S_PLUS];
        put16(pl, r->plus_len);
        memcpy(pl + 2, text + r->plus_off, r->plus_len);
        o[FQZO_S_PLUS] += 2 + r->plus_len;
    }
    free(npos);
    for (int k = 0; k < FQZO_NSTREAMS; k++) s->len[k] = o[k];
    return FQZO_OK;
}

/* ===================================================================== */
/* compress.go:944-1078 — blockReader.writeRecord                         */
/* ===================================================================== */

static long join_impl(const uint8_t *const data[FQZO_NSTREAMS], const size_t len[FQZO_NSTREAMS], uint32_t num_records,
                      int enc, uint8_t *out, size_t cap)
{
    size_t so = 0, qo = 0, ho = 0, po = 0, no = 0, lo = 0, w = 0;
    const uint8_t *seqd = data[FQZO_S_SEQ], *quald = data[FQZO_S_QUAL], *hdrd = data[FQZO_S_HEADERS];
    const uint8_t *plusd = data[FQZO_S_PLUS], *nposd = data[FQZO_S_NPOS], *lend = data[FQZO_S_LENGTHS];
    size_t plus_len = plusd ? len[FQZO_S_PLUS] : 0;
    uint8_t off = phred_offset(enc);
    for (uint32_t r = 0; r < num_records; r++) {
        /* readSeqLength compress.go:1046-1053 */
        if (lo + 4 > len[FQZO_S_LENGTHS]) return FQZO_E_TRUNC_LEN;
        size_t L = get32(lend + lo);
        lo += 4;
        /* readNPositions compress.go:1055-1078 */
        if (no + 2 > len[FQZO_S_NPOS]) return FQZO_E_TRUNC_NPOS;
        size_t nn = get16(nposd + no);
        no += 2;
        if (no + 2 * nn > len[FQZO_S_NPOS]) return FQZO_E_TRUNC_NPOS;
        const uint8_t *np = nposd + no;
        no += 2 * nn;
        /* appendHeader compress.go:977-992 */
        if (ho + 2 > len[FQZO_S_HEADERS]) return FQZO_E_TRUNC_HEADER;
        size_t H = get16(hdrd + ho);
        ho += 2;
        if (ho + H > len[FQZO_S_HEADERS]) return FQZO_E_TRUNC_HEADER;
        if (out) {
            if (w + H + 2 > cap) return FQZO_E_DST_SMALL;
            out[w] = '@'; memcpy(out + w + 1, hdrd + ho, H); out[w + 1 + H] = '\n';
        }
        w += H + 2;
        ho += H;
        /* appendSequence compress.go:1017-1029 */
        size_t pl = (L + 3) / 4;
        if (so + pl > len[FQZO_S_SEQ]) return FQZO_E_TRUNC_SEQ;
        if (out) {
            if (w + L + 1 > cap) return FQZO_E_DST_SMALL;
            static const uint8_t bases[4] = {'A', 'C', 'G', 'T'};
            const uint8_t *pk = seqd + so;
            for (size_t i = 0; i < L; i++) out[w + i] = bases[(pk[i >> 2] >> ((i & 3) << 1)) & 3];
            for (size_t k = 0; k < nn; k++) {
                size_t p = get16(np + 2 * k);
                if (p >= L) return FQZO_E_NPOS_RANGE;
                out[w + p] = 'N';
            }
            out[w + L] = '\n';
        } else {
            for (size_t k = 0; k < nn; k++) if (get16(np + 2 * k) >= L) return FQZO_E_NPOS_RANGE;
        }
        w += L + 1;
        so += pl;
        /* appendPlusLine compress.go:994-1015 */
        if (plus_len == 0) {
            if (out) { if (w + 2 > cap) return FQZO_E_DST_SMALL; out[w] = '+'; out[w + 1] = '\n'; }
            w += 2;
        } else {
            if (po + 2 > plus_len) return FQZO_E_TRUNC_PLUS;
            size_t P = get16(plusd + po);
            po += 2;
            if (po + P > plus_len) return FQZO_E_TRUNC_PLUS;
            if (out) {
                if (w + P + 2 > cap) return FQZO_E_DST_SMALL;
                out[w] = '+'; memcpy(out + w + 1, plusd + po, P); out[w + 1 + P] = '\n';
            }
            w += P + 2;
            po += P;
        }
        /* appendQuality compress.go:1031-1044: DeltaDecode then DenormalizeQuality */
        if (qo + L > len[FQZO_S_QUAL]) return FQZO_E_TRUNC_QUAL;
        if (out) {
            if (w + L + 1 > cap) return FQZO_E_DST_SMALL;
            uint8_t acc = 0;
            for (size_t i = 0; i < L; i++) { acc = (uint8_t)(acc + quald[qo + i]); out[w + i] = (uint8_t)(acc + off); }
            out[w + L] = '\n';
        }
        w += L + 1;
        qo += L;
    }
    return (long)w;
}

long fqzo_join_block(const uint8_t *const data[FQZO_NSTREAMS], const size_t len[FQZO_NSTREAMS], uint32_t num_records,
                     int enc, uint8_t *out, size_t cap)
{
    return join_impl(data, len, num_records, enc, out, cap);
}
long fqzo_join_block_size(const uint8_t *const data[FQZO_NSTREAMS], const size_t len[FQZO_NSTREAMS], uint32_t num_records)
{
    return join_impl(data, len, num_records, 0, NULL, 0);
}

/* ===================================================================== */
/* optional system libzstd (CPU-baseline leg only)                        */
/* ===================================================================== */

typedef size_t (*zstd_compress_fn)(void *, size_t, const void *, size_t, int);
typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef size_t (*zstd_bound_fn)(size_t);
typedef unsigned (*zstd_iserr_fn)(size_t);
typedef unsigned (*zstd_ver_fn)(void);
typedef unsigned long long (*zstd_fcs_fn)(const void *, size_t);
static struct {
    int tried;
    void *h;
    zstd_compress_fn compress;
    zstd_decompress_fn decompress;
    zstd_bound_fn bound;
    zstd_iserr_fn is_error;
    zstd_ver_fn version;
    zstd_fcs_fn fcs;
} zs;
static pthread_once_t zs_once = PTHREAD_ONCE_INIT;
static void zs_load(void)
{
    zs.tried = 1;
    zs.h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!zs.h) return;
    zs.compress = (zstd_compress_fn)dlsym(zs.h, "ZSTD_compress");
    zs.decompress = (zstd_decompress_fn)dlsym(zs.h, "ZSTD_decompress");
    zs.bound = (zstd_bound_fn)dlsym(zs.h, "ZSTD_compressBound");
    zs.is_error = (zstd_iserr_fn)dlsym(zs.h, "ZSTD_isError");
    zs.version = (zstd_ver_fn)dlsym(zs.h, "ZSTD_versionNumber");
    zs.fcs = (zstd_fcs_fn)dlsym(zs.h, "ZSTD_getFrameContentSize");
    if (!zs.compress || !zs.decompress || !zs.bound || !zs.is_error || !zs.version || !zs.fcs) { dlclose(zs.h); zs.h = NULL; }
}
unsigned fqzo_libzstd_version(void)
{
    pthread_once(&zs_once, zs_load);
    return zs.h ? zs.version() : 0;
}

/* ===================================================================== */
/* compress.go:125-192, 194-238, 240-278 — Compress                       */
/* ===================================================================== */

size_t fqzo_compress_bound(size_t n_bytes)
{
    /* worst case: every pre-entropy byte stored raw + framing; pre-entropy <= input + 4 B/record, record >= 4 B */
    return 10 + 2 * n_bytes + (n_bytes / 1024 + 64) * 64 + 4096;
}

typedef struct {
    const uint8_t *text;
    const fqzo_record *recs;
    size_t n_rec;
    int enc, entropy;
    int framing;    /* fqzo_options.framing */
    size_t text_end; /* first byte behind the block's last record */
    uint8_t *out; /* malloc'd block bytes */
    size_t out_len;
    int err;
} enc_job;

/* FQZ-S1 (fqz_entropy.c): the block as segments, if it qualifies.  Returns 1 and the block in j->out, 0 if the block does not
 * qualify (the caller writes it with the FQZ-H2 framing), or a negative error.  `whole` = the streams of the whole block. */
static int encode_block_segments(enc_job *j, const fqzo_streams *whole)
{
    const size_t B0 = j->recs[0].hdr_off - 1;
    const uint32_t n_seg = (uint32_t)((j->text_end - B0 + FQZO_SEG_TEXT - 1) / FQZO_SEG_TEXT);
    uint32_t *first = (uint32_t *)calloc((size_t)n_seg + 1, sizeof(uint32_t)); /* first record of every segment */
    if (!first) return FQZO_E_NOMEM;
    {
        uint32_t seg = 0;
        for (size_t r = 0; r < j->n_rec; r++) {
            const uint32_t sr = (uint32_t)((j->recs[r].hdr_off - 1 - B0) / FQZO_SEG_TEXT);
            while (seg < sr) first[++seg] = (uint32_t)r;
        }
        while (seg < n_seg) first[++seg] = (uint32_t)j->n_rec;
    }
    fqzo_streams *ss = (fqzo_streams *)calloc(n_seg ? n_seg : 1, sizeof(fqzo_streams));
    if (!ss) { free(first); return FQZO_E_NOMEM; }
    int ok = 1, err = 0;
    for (uint32_t s = 0; s < n_seg && ok && !err; s++) {
        const uint32_t cnt = first[s + 1] - first[s];
        if (cnt > FQZO_SEG_RMAX) { ok = 0; break; }
        if (!cnt) continue;
        err = fqzo_split_block(j->text, j->recs + first[s], cnt, j->enc, &ss[s]);
        if (err) break;
        size_t need = 0;
        for (int k = 0; k < FQZO_NSTREAMS; k++) need += (ss[s].len[k] + 15) & ~(size_t)15;
        if (need > FQZO_SEG_ARENA) ok = 0;
    }
    uint8_t *out = NULL;
    size_t w = 36;
    if (ok && !err) {
        size_t cap = 36;
        for (int k = 0; k < FQZO_NSTREAMS; k++) {
            cap += fqzo_seg_index_len(n_seg);
            for (uint32_t s = 0; s < n_seg; s++) cap += fqzo_seg_frame_bound(ss[s].len[k]);
        }
        out = (uint8_t *)malloc(cap);
        if (!out) err = FQZO_E_NOMEM;
    }
    if (ok && !err) {
        uint32_t comp[FQZO_NSTREAMS];
        for (int k = 0; k < FQZO_NSTREAMS; k++) { /* compress.go:523-528: order seq, qual, headers, plus, nPos, lengths */
            comp[k] = 0;
            if (!whole->len[k]) continue; /* an empty stream is an empty payload (EncodeAll without zero frames) */
            uint8_t *idx = out + w;
            fqzo_seg_index_write(idx, k, (uint32_t)whole->len[k], n_seg);
            size_t p = w + fqzo_seg_index_len(n_seg);
            for (uint32_t s = 0; s < n_seg; s++) {
                const size_t fl = fqzo_seg_frame(ss[s].data[k], ss[s].len[k], k, out + p);
                fqzo_seg_index_entry(idx, s, (uint32_t)fl, (uint32_t)ss[s].len[k], first[s + 1] - first[s]);
                p += fl;
            }
            comp[k] = (uint32_t)(p - w);
            w = p;
        }
        fqzo_block_header bh = { (uint32_t)j->n_rec, comp[0], comp[1], comp[2], comp[3], comp[4], comp[5],
                                 whole->original_seq_size, whole->original_qual_size };
        fqzo_write_block_header(&bh, 2, out);
        j->out = out;
        j->out_len = w;
    }
    for (uint32_t s = 0; s < n_seg; s++) fqzo_streams_free(&ss[s]);
    free(ss); free(first);
    if (err) { free(out); return err; }
    return ok ? 1 : 0;
}

/* compress.go:471-555 compressBlockWithBuffers */
static void encode_block_job(enc_job *j)
{
    fqzo_streams s;
    j->out = NULL; j->out_len = 0;
    j->err = fqzo_split_block(j->text, j->recs, j->n_rec, j->enc, &s);
    if (j->err) return;
    if (j->entropy == 0 && j->framing == 1) {
        const int r = encode_block_segments(j, &s);
        if (r) { fqzo_streams_free(&s); if (r < 0) j->err = r; return; }
    }
    size_t cap = 36;
    for (int k = 0; k < FQZO_NSTREAMS; k++)
        cap += j->entropy == 1 ? zs.bound(s.len[k]) : fqzo_entropy_bound(s.len[k]);
    uint8_t *out = malloc(cap);
    if (!out) { fqzo_streams_free(&s); j->err = FQZO_E_NOMEM; return; }
    size_t comp[FQZO_NSTREAMS], w = 36;
    /* compress.go:523-528: order seq, qual, headers, plus, nPos, lengths */
    for (int k = 0; k < FQZO_NSTREAMS; k++) {
        if (j->entropy == 1) {
            comp[k] = s.len[k] ? zs.compress(out + w, cap - w, s.data[k], s.len[k], 1) : 0;
            if (zs.is_error(comp[k])) { free(out); fqzo_streams_free(&s); j->err = FQZO_E_ENTROPY; return; }
        } else {
            comp[k] = fqzo_entropy_encode_stream_v(s.data[k], s.len[k], k, j->entropy == 2 ? 3 : 2, out + w);
        }
        w += comp[k];
    }
    fqzo_block_header bh = { /* compress.go:532-542 */
        (uint32_t)j->n_rec, (uint32_t)comp[0], (uint32_t)comp[1], (uint32_t)comp[2], (uint32_t)comp[3],
        (uint32_t)comp[4], (uint32_t)comp[5], s.original_seq_size, s.original_qual_size };
    fqzo_write_block_header(&bh, 2, out);
    fqzo_streams_free(&s);
    j->out = out;
    j->out_len = w;
}

typedef struct {
    enc_job *jobs;
    size_t n_jobs;
    size_t next;
    pthread_mutex_t mu;
} job_pool;

static void *enc_worker(void *arg)
{
    job_pool *p = arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        size_t i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->n_jobs) return NULL;
        encode_block_job(&p->jobs[i]);
    }
}

static int resolve_workers(int w)
{
    if (w > 0) return w;
    long n = sysconf(_SC_NPROCESSORS_ONLN); /* compress.go:132-134 runtime.NumCPU() */
    return n > 0 ? (int)n : 1;
}

long fqzo_compress(const uint8_t *fastq, size_t n, uint8_t *out, size_t cap, const fqzo_options *opt)
{
    fqzo_options o = {0, 0, 0, 0, 0, 0, 0};
    if (opt) o = *opt;
    if (o.block_index && o.entropy != 2) return FQZO_E_FILE_VERSION;
    if (!o.block_size) o.block_size = 100000;       /* compress.go:126-131 */
    if (!o.batch_records) o.batch_records = 100000; /* compress.go:48-52: batches are always 100 000 records (App. B-4) */
    int workers = resolve_workers(o.workers);
    if (o.entropy == 1 && !fqzo_libzstd_version()) return FQZO_E_ENTROPY;

    /* parse everything (the reference parses on one goroutine, compress.go:254-257) */
    size_t max_recs = n / 4 + 1, n_rec = 0, pos = 0;
    fqzo_record *recs = malloc(sizeof(fqzo_record) * max_recs);
    if (!recs) return FQZO_E_NOMEM;
    for (;;) {
        int eof;
        size_t want = o.batch_records;
        if (want > max_recs - n_rec) want = max_recs - n_rec;
        if (!want) break;
        long got = fqzo_parse_batch(fastq, n, &pos, recs + n_rec, want, &eof);
        if (got < 0) { free(recs); return got; } /* "parsing FASTQ: ..." compress.go:143 */
        n_rec += (size_t)got;
        if (eof || got == 0) break;
    }

    /* compress.go:146-154 + quality.go:22-49: encoding from the first batch only */
    int enc = FQZO_PHRED33;
    {
        size_t first = n_rec < o.batch_records ? n_rec : o.batch_records;
        uint8_t min = 255;
        int below59 = 0;
        for (size_t r = 0; r < first && !below59; r++)
            for (uint32_t i = 0; i < recs[r].qual_len; i++) {
                uint8_t b = fastq[recs[r].qual_off + i];
                if (b < min) min = b;
                if (b < 59) { below59 = 1; break; }
            }
        if (!below59 && min != 255 && min >= 64) enc = FQZO_PHRED64;
    }
    if (o.force_encoding == 1) enc = FQZO_PHRED33;
    if (o.force_encoding == 2) enc = FQZO_PHRED64;

    /* compress.go:157-168 */
    if (cap < 10) { free(recs); return FQZO_E_DST_SMALL; }
    fqzo_file_header fh = { (uint8_t)(o.entropy == 2 ? 3 : 2), o.block_size, (uint8_t)(enc == FQZO_PHRED64 ? FQZO_FLAG_PHRED64 : 0) };
    fqzo_write_file_header(&fh, out);
    size_t w = 10;

    size_t n_jobs = (n_rec + o.batch_records - 1) / o.batch_records;
    if (!n_jobs) { /* App. B-7: empty input -> header only (and an empty block table when one is asked for) */
        free(recs);
        if (o.block_index) {
            if (cap < w + 24) return FQZO_E_DST_SMALL;
            memset(out + w, 0xFF, 4); memcpy(out + w + 4, "FQZX", 4); memset(out + w + 8, 0, 4);
            memset(out + w + 12, 0, 8); out[w + 12] = (uint8_t)w; memcpy(out + w + 20, "FQZX", 4);
            w += 24;
        }
        return (long)w;
    }
    enc_job *jobs = calloc(n_jobs, sizeof *jobs);
    if (!jobs) { free(recs); return FQZO_E_NOMEM; }
    for (size_t b = 0; b < n_jobs; b++) {
        jobs[b].text = fastq;
        jobs[b].recs = recs + b * o.batch_records;
        jobs[b].n_rec = (b + 1 == n_jobs) ? n_rec - b * o.batch_records : o.batch_records;
        jobs[b].enc = enc;
        jobs[b].entropy = o.entropy;
        jobs[b].framing = o.framing;
        if (b + 1 < n_jobs) jobs[b].text_end = jobs[b].recs[jobs[b].n_rec].hdr_off - 1;
        else { /* behind the newline of the last record's quality line (a '\r' in front of it was stripped by the parser) */
            const fqzo_record *lr = &jobs[b].recs[jobs[b].n_rec - 1];
            size_t e = (size_t)lr->qual_off + lr->qual_len;
            if (e < n && fastq[e] == '\r') e++;
            jobs[b].text_end = e + 1;
        }
    }
    job_pool pool = { jobs, n_jobs, 0, PTHREAD_MUTEX_INITIALIZER };
    if ((size_t)workers > n_jobs) workers = (int)n_jobs;
    if (workers <= 1) {
        enc_worker(&pool);
    } else {
        pthread_t th[256];
        if (workers > 256) workers = 256;
        for (int t = 0; t < workers; t++) pthread_create(&th[t], NULL, enc_worker, &pool);
        for (int t = 0; t < workers; t++) pthread_join(th[t], NULL);
    }
    /* ordered collector, compress.go:365-403 */
    long ret = 0;
    for (size_t b = 0; b < n_jobs; b++) {
        if (!ret && jobs[b].err) ret = jobs[b].err;
        if (!ret) {
            if (w + jobs[b].out_len > cap) ret = FQZO_E_DST_SMALL;
            else { memcpy(out + w, jobs[b].out, jobs[b].out_len); w += jobs[b].out_len; }
        }
        free(jobs[b].out);
    }
    if (!ret && o.block_index) { /* block table: mark | 'FQZX' | n | n x { u64 offset, u32 records } | u64 offset of the table | 'FQZX' */
        const size_t need = 12 + 12 * n_jobs + 12;
        if (w + need > cap) ret = FQZO_E_DST_SMALL;
        else {
            uint8_t *t = out + w;
            const uint64_t at = w;
            uint64_t off = 10;
            memset(t, 0xFF, 4); memcpy(t + 4, "FQZX", 4);
            for (int i = 0; i < 4; i++) t[8 + i] = (uint8_t)(n_jobs >> (8 * i));
            t += 12;
            for (size_t b = 0; b < n_jobs; b++, t += 12) {
                for (int i = 0; i < 8; i++) t[i] = (uint8_t)(off >> (8 * i));
                for (int i = 0; i < 4; i++) t[8 + i] = (uint8_t)((uint32_t)jobs[b].n_rec >> (8 * i));
                off += jobs[b].out_len;
            }
            for (int i = 0; i < 8; i++) t[i] = (uint8_t)(at >> (8 * i));
            memcpy(t + 8, "FQZX", 4);
            w += need;
        }
    }
    free(jobs);
    free(recs);
    return ret ? ret : (long)w;
}

/* ===================================================================== */
/* compress.go:558-604, 721-837 — Decompress                              */
/* ===================================================================== */

typedef struct {
    const uint8_t *payload[FQZO_NSTREAMS];
    size_t psize[FQZO_NSTREAMS];
    uint32_t num_records;
    uint8_t version;
    int enc;
    uint8_t *out; /* destination slice (phase 2) */
    size_t cap;
    long result;  /* bytes or error */
    int phase;    /* 1: entropy-decode + size; 2: join into out, free */
    uint8_t *data[FQZO_NSTREAMS];
    size_t len[FQZO_NSTREAMS];
} dec_job;

static long decode_any(const uint8_t *src, size_t n, uint8_t **dst, size_t *len)
{
    *dst = NULL; *len = 0;
    if (!n) return 0;
    long fcs = fqzo_entropy_content_size(src, n);
    if (fcs == FQZO_E_ENTROPY) return FQZO_E_ENTROPY;
    size_t cap = fcs >= 0 ? (size_t)fcs : 0;
    if (fcs < 0) { /* no content size: fall back to libzstd's answer or grow */
        cap = n * 64 + 65536;
    }
    uint8_t *buf = malloc(cap ? cap : 1);
    if (!buf) return FQZO_E_NOMEM;
    long r = fqzo_entropy_decode(src, n, buf, cap);
    if (r == FQZO_E_ENTROPY && fqzo_libzstd_version()) {
        /* a frame with LZ sequences (e.g. written by the stock encoder): our
         * oracle decoder covers only the Huffman-literal subset, so let the
         * independent libzstd decode it when present */
        size_t z = zs.decompress(buf, cap, src, n);
        r = zs.is_error(z) ? FQZO_E_ENTROPY : (long)z;
    }
    if (r < 0) { free(buf); return r; }
    *dst = buf; *len = (size_t)r;
    return r;
}

/* compress.go:780-837 decompressJobToPooledBuffer */
static void decode_block_job(dec_job *j)
{
    long r = 0;
    if (j->phase == 1) {
        /* compress.go:783-814: plus first (v2), then seq, qual, headers, nPos, lengths */
        static const int order[FQZO_NSTREAMS] = {FQZO_S_PLUS, FQZO_S_SEQ, FQZO_S_QUAL, FQZO_S_HEADERS, FQZO_S_NPOS, FQZO_S_LENGTHS};
        for (int q = 0; q < FQZO_NSTREAMS && r >= 0; q++) {
            int k = order[q];
            if (k == FQZO_S_PLUS && j->version < 2) continue;
            r = decode_any(j->payload[k], j->psize[k], &j->data[k], &j->len[k]);
        }
    }
    if (r >= 0) {
        const uint8_t *cd[FQZO_NSTREAMS];
        size_t len[FQZO_NSTREAMS];
        for (int k = 0; k < FQZO_NSTREAMS; k++) { cd[k] = j->data[k]; len[k] = j->len[k]; }
        if (j->version < 2) { cd[FQZO_S_PLUS] = NULL; len[FQZO_S_PLUS] = 0; }
        r = j->phase == 2 ? fqzo_join_block(cd, len, j->num_records, j->enc, j->out, j->cap)
                          : fqzo_join_block_size(cd, len, j->num_records);
    }
    j->result = r;
}

typedef struct {
    dec_job *jobs;
    size_t n_jobs, next;
    pthread_mutex_t mu;
} dec_pool;
static void *dec_worker(void *arg)
{
    dec_pool *p = arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        size_t i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->n_jobs) return NULL;
        decode_block_job(&p->jobs[i]);
    }
}
static void run_dec(dec_job *jobs, size_t n_jobs, int workers)
{
    dec_pool pool = { jobs, n_jobs, 0, PTHREAD_MUTEX_INITIALIZER };
    if ((size_t)workers > n_jobs) workers = (int)n_jobs;
    if (workers <= 1) { dec_worker(&pool); return; }
    pthread_t th[256];
    if (workers > 256) workers = 256;
    for (int t = 0; t < workers; t++) pthread_create(&th[t], NULL, dec_worker, &pool);
    for (int t = 0; t < workers; t++) pthread_join(th[t], NULL);
}

long fqzo_decompress(const uint8_t *fqz, size_t n, uint8_t *out, size_t cap, int workers)
{
    workers = resolve_workers(workers);
    fqzo_file_header fh;
    int e = fqzo_read_file_header(fqz, n, &fh); /* compress.go:567-570 */
    if (e) return e;
    if (fh.version != 1 && fh.version != 2 && fh.version != 3) return FQZO_E_FILE_VERSION; /* compress.go:571-573; 3 = FQZ-R1 (ours) */
    int enc = (fh.flags & FQZO_FLAG_PHRED64) ? FQZO_PHRED64 : FQZO_PHRED33; /* compress.go:576-579 */

    /* readNextDecompressJob / readCompressedStreams, compress.go:721-758 */
    size_t pos = 10, n_jobs = 0, cap_jobs = 16;
    dec_job *jobs = malloc(cap_jobs * sizeof *jobs);
    if (!jobs) return FQZO_E_NOMEM;
    while (pos < n) { /* clean EOF only at a block-header boundary, compress.go:614-617 */
        if (fh.version == 3 && n - pos >= 8 && !memcmp(fqz + pos, "\xFF\xFF\xFF\xFF" "FQZX", 8)) break; /* FQZ-R1's block table: the chain ends here */
        fqzo_block_header bh;
        int hs = fqzo_read_block_header(fqz + pos, n - pos, fh.version, &bh);
        if (hs < 0) { free(jobs); return hs; } /* "reading block header: unexpected EOF" */
        pos += (size_t)hs;
        if (n_jobs == cap_jobs) {
            cap_jobs *= 2;
            dec_job *nj = realloc(jobs, cap_jobs * sizeof *jobs);
            if (!nj) { free(jobs); return FQZO_E_NOMEM; }
            jobs = nj;
        }
        dec_job *j = &jobs[n_jobs++];
        memset(j, 0, sizeof *j);
        /* order on the wire: seq, qual, headers, [plus], nPos, lengths (compress.go:738-751) */
        uint32_t sizes[FQZO_NSTREAMS] = {bh.seq_size, bh.qual_size, bh.header_size, bh.plus_size, bh.npos_size, bh.lengths_size};
        for (int k = 0; k < FQZO_NSTREAMS; k++) {
            if (k == FQZO_S_PLUS && fh.version < 2) continue;
            if ((size_t)sizes[k] > n - pos) { free(jobs); return FQZO_E_READ_DATA; }
            j->payload[k] = fqz + pos;
            j->psize[k] = sizes[k];
            pos += sizes[k];
        }
        j->num_records = bh.num_records;
        j->version = fh.version;
        j->enc = enc;
    }
    if (!n_jobs) { free(jobs); return 0; } /* compress.go:591-593 */

    /* phase 1: entropy decode + sizes (so blocks can be written at their final offsets in parallel) */
    for (size_t b = 0; b < n_jobs; b++) jobs[b].phase = 1;
    run_dec(jobs, n_jobs, workers);
    size_t total = 0;
    long ret = 0;
    for (size_t b = 0; b < n_jobs; b++) {
        if (jobs[b].result < 0) { ret = jobs[b].result; break; }
        total += (size_t)jobs[b].result;
    }
    if (!ret && out) {
        if (total > cap) ret = FQZO_E_DST_SMALL;
        else {
            size_t off = 0;
            for (size_t b = 0; b < n_jobs; b++) {
                size_t sz = (size_t)jobs[b].result;
                jobs[b].out = out + off;
                jobs[b].cap = sz;
                jobs[b].phase = 2;
                off += sz;
            }
            run_dec(jobs, n_jobs, workers);
            for (size_t b = 0; b < n_jobs; b++) if (jobs[b].result < 0) { ret = jobs[b].result; break; }
        }
    }
    for (size_t b = 0; b < n_jobs; b++)
        for (int k = 0; k < FQZO_NSTREAMS; k++) free(jobs[b].data[k]);
    free(jobs);
    return ret ? ret : (long)total;
}
