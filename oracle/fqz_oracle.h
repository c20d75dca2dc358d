/*
 * fqz_oracle.h — CPU restatement ("oracle") of the fqpack per-block hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fastqpacker_amd/,
 * include/, the C-ABI library) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as a checker.
 *
 * Parity status: the reference is Go and cannot be built here (no Go
 * toolchain).  This restatement is pinned by the known-answer tables the
 * reference's own tests hold (tests/test_oracle_kat.py).
 * The entropy stage (klauspost/compress zstd v1.19.1, go.mod:8, not vendored)
 * is NOT reproduced byte-for-byte: "parity unpinned" at that boundary — no
 * reference test pins compressed bytes.  What is pinned there: every payload
 * is a valid zstd frame (RFC 8878) that decodes to the exact pre-entropy
 * stream; tests prove that with the system libzstd as an independent decoder.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.
 */
#ifndef FQZ_ORACLE_H
#define FQZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- internal/encoder/sequence.go ------------------------------------- */
#define FQZO_MAX_SEQUENCE_LENGTH 65536u /* sequence.go:11 */

/* sequence.go:139-184 AppendPackedBases.  Writes (n+3)/4 bytes to packed,
 * appends positions (< 65536) of bytes not in ACGTacgt to npos.  Returns the
 * number of N positions written. npos must hold min(n,65536) entries. */
size_t fqzo_pack_bases(const uint8_t *seq, size_t n, uint8_t *packed, uint16_t *npos);

/* sequence.go:188-223 AppendUnpackBases. Positions >= seq_len are an error
 * (Go would panic on the slice index): returns -1, else 0. */
int fqzo_unpack_bases(const uint8_t *packed, const uint16_t *npos, size_t n_npos,
                      size_t seq_len, uint8_t *seq);

/* ---- internal/encoder/quality.go -------------------------------------- */
#define FQZO_PHRED33 0 /* quality.go:14 */
#define FQZO_PHRED64 1 /* quality.go:15 */

/* quality.go:22-49 DetectEncoding over a list of quality strings. */
int fqzo_detect_encoding(const uint8_t *const *quals, const size_t *lens, size_t n);
/* quality.go:53-62 / 66-75 */
void fqzo_normalize_quality(uint8_t *q, size_t n, int enc);
void fqzo_denormalize_quality(uint8_t *q, size_t n, int enc);
/* quality.go:81-103 / 107-118 */
void fqzo_delta_encode(uint8_t *q, size_t n);
void fqzo_delta_decode(uint8_t *q, size_t n);

/* ---- internal/fqformat/container.go ----------------------------------- */
#define FQZO_FLAG_PAIRED_END 0x01 /* container.go:15 */
#define FQZO_FLAG_PHRED64 0x02    /* container.go:16 */

typedef struct {
    uint8_t version;
    uint32_t block_size;
    uint8_t flags;
} fqzo_file_header;

typedef struct { /* container.go:70-80 */
    uint32_t num_records, seq_size, qual_size, header_size, plus_size;
    uint32_t npos_size, lengths_size, original_seq_size, original_qual_size;
} fqzo_block_header;

/* container.go:35-45: writes 10 bytes. */
void fqzo_write_file_header(const fqzo_file_header *h, uint8_t out[10]);
/* container.go:48-67: 0 ok; FQZO_E_SHORT if n<10; FQZO_E_MAGIC on bad magic. */
int fqzo_read_file_header(const uint8_t *in, size_t n, fqzo_file_header *h);
/* container.go:83-113: returns bytes written (32 or 36) or FQZO_E_BLOCK_VERSION. */
int fqzo_write_block_header(const fqzo_block_header *b, uint8_t version, uint8_t *out);
/* container.go:116-152: returns bytes consumed, FQZO_E_SHORT, or FQZO_E_BLOCK_VERSION. */
int fqzo_read_block_header(const uint8_t *in, size_t n, uint8_t version, fqzo_block_header *b);

/* ---- error codes (mirror the reference's error strings) ---------------- */
enum {
    FQZO_OK = 0,
    FQZO_E_SHORT = -1,          /* io.ErrUnexpectedEOF / io.EOF while reading a header */
    FQZO_E_MAGIC = -2,          /* "invalid magic bytes: not an FQZ file" container.go:54 */
    FQZO_E_BLOCK_VERSION = -3,  /* "unsupported block header version" container.go:111,150 */
    FQZO_E_FILE_VERSION = -4,   /* "unsupported file version: %d" compress.go:572 */
    FQZO_E_HDR_AT = -5,         /* "invalid FASTQ: header line must start with @" parser.go:143 */
    FQZO_E_SEP_PLUS = -6,       /* "invalid FASTQ: separator line must start with +" parser.go:164 */
    FQZO_E_LEN_MISMATCH = -7,   /* "invalid FASTQ: sequence and quality lengths must match" parser.go:180 */
    FQZO_E_LONG_N = -8,         /* "... ambiguous bases beyond position 65536 ..." compress.go:484 */
    FQZO_E_TRUNC_HEADER = -9,   /* "truncated header data" compress.go:979,985 */
    FQZO_E_TRUNC_PLUS = -10,    /* "truncated plus-line payload data" compress.go:1002,1007 */
    FQZO_E_TRUNC_SEQ = -11,     /* "truncated sequence data" compress.go:1020 */
    FQZO_E_TRUNC_QUAL = -12,    /* "truncated quality data" compress.go:1033 */
    FQZO_E_TRUNC_LEN = -13,     /* "truncated length data" compress.go:1048 */
    FQZO_E_TRUNC_NPOS = -14,    /* "truncated N position data" compress.go:1057,1072 */
    FQZO_E_ENTROPY = -15,       /* "decompressing <stream>: ..." compress.go:787-813 */
    FQZO_E_READ_DATA = -16,     /* "reading compressed data: ..." compress.go:732 */
    FQZO_E_NOMEM = -17,
    FQZO_E_DST_SMALL = -18,
    FQZO_E_FIELD_WRAP = -19,    /* header/plus > 65535 B or > 65535 N: u16 wrap in the reference (App. B-6); rejected here */
    FQZO_E_NPOS_RANGE = -20,    /* N position >= read length (Go would panic) */
};
const char *fqzo_strerror(int code);

/* ---- internal/fqparser/parser.go -------------------------------------- */
typedef struct {
    uint32_t hdr_off, hdr_len;   /* without leading '@' */
    uint32_t seq_off, seq_len;
    uint32_t plus_off, plus_len; /* without leading '+' */
    uint32_t qual_off, qual_len;
} fqzo_record;

/* parser.go:136-243 (nextInto + readLine) over a memory buffer.  Parses up to
 * max_records records starting at *pos; advances *pos past what was consumed.
 * Returns number of records parsed (>=0) or a negative error.  *eof is set
 * when the input was exhausted (io.EOF surfaced).  Restates: CR stripping
 * (parser.go:213-215), a final line without '\n' is discarded with EOF
 * (parser.go:210-220), EOF inside a record after >=1 records is swallowed
 * (parser.go:196-199), EOF with zero records is returned as EOF. */
long fqzo_parse_batch(const uint8_t *text, size_t n, size_t *pos, fqzo_record *recs,
                      size_t max_records, int *eof);

/* ---- internal/compress/compress.go: six pre-entropy streams ----------- */
enum { FQZO_S_SEQ = 0, FQZO_S_QUAL, FQZO_S_HEADERS, FQZO_S_PLUS, FQZO_S_NPOS, FQZO_S_LENGTHS, FQZO_NSTREAMS };

typedef struct {
    uint8_t *data[FQZO_NSTREAMS];
    size_t len[FQZO_NSTREAMS];
    uint32_t original_seq_size, original_qual_size;
} fqzo_streams;

/* compress.go:471-520 (the per-record loop of compressBlockWithBuffers).
 * Allocates s->data[*]; free with fqzo_streams_free. */
int fqzo_split_block(const uint8_t *text, const fqzo_record *recs, size_t n_rec, int enc, fqzo_streams *s);
void fqzo_streams_free(fqzo_streams *s);

/* compress.go:944-1078 (blockReader.writeRecord and helpers).  plus==NULL or
 * plus_len==0 means "bare +" (compress.go:995-999).  Returns bytes written to
 * out (>=0) or a negative error. */
long fqzo_join_block(const uint8_t *const data[FQZO_NSTREAMS], const size_t len[FQZO_NSTREAMS],
                     uint32_t num_records, int enc, uint8_t *out, size_t cap);
/* Exact output size of fqzo_join_block (or negative error), without writing. */
long fqzo_join_block_size(const uint8_t *const data[FQZO_NSTREAMS], const size_t len[FQZO_NSTREAMS],
                          uint32_t num_records);

/* ---- entropy stage ("FQZ-H2" profile: zstd frames) --
 * Replaces zstd.Encoder.EncodeAll (compress.go:523-528).  Deterministic:
 * the HIP encoder must produce identical bytes (spec in DESIGN.md §Entropy).
 */
#define FQZO_CHUNK 16384u
#define FQZO_GROUP 4u   /* chunks that share one Huffman table */
size_t fqzo_entropy_bound(size_t n);
/* Returns frame size; n==0 -> 0 bytes (klauspost EncodeAll without zero frames). */
size_t fqzo_entropy_encode(const uint8_t *src, size_t n, uint8_t *dst);
/* the payload of stream `stream` (0 seq .. 5 lengths) of a block: the 2-bit packed bases are Raw by definition */
size_t fqzo_entropy_encode_stream(const uint8_t *src, size_t n, int stream, uint8_t *dst);
/* version 3 (FQZ-R1): the quality stream (stream 1) is coded with interleaved rANS blocks; version 2 = the above */
size_t fqzo_entropy_encode_stream_v(const uint8_t *src, size_t n, int stream, int version, uint8_t *dst);
/* FQZ-S1, the segment framing of container version 2 (fqz_entropy.c): limits a block must meet, and the pieces of a payload */
#define FQZO_SEG_TEXT 65536u   /* text bytes per segment, from the first byte of the block's first record */
#define FQZO_SEG_RMAX 384u     /* records per segment */
#define FQZO_SEG_ARENA 51200u  /* bytes of the six stream parts of a segment, each rounded up to 16 */
size_t fqzo_seg_index_len(uint32_t n_seg);
void fqzo_seg_index_write(uint8_t *idx, int stream, uint32_t raw_total, uint32_t n_seg);
void fqzo_seg_index_entry(uint8_t *idx, uint32_t seg, uint32_t comp, uint32_t raw, uint32_t nrec);
size_t fqzo_seg_frame(const uint8_t *src, size_t n, int stream, uint8_t *dst);
size_t fqzo_seg_frame_bound(size_t n);
/* XXH64 (zstd content checksum = its low 32 bits, seed 0) */
uint64_t fqzo_xxh64(const uint8_t *p, size_t len, uint64_t seed);
/* Decoder for any zstd frame made only of Raw / RLE / Compressed blocks whose
 * Compressed blocks carry zero sequences (Huffman, raw or RLE literals).
 * Replaces zstd.Decoder.DecodeAll (compress.go:785-814) for our own output.
 * Returns decoded size or negative error. */
long fqzo_entropy_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
/* Frame_Content_Size when present, else -1; 0 for empty input. */
long fqzo_entropy_content_size(const uint8_t *src, size_t n);

/* Building blocks of the entropy stage, exposed for unit tests and so that the
 * HIP kernels can be checked step by step. */
/* Length-limited (<=11) Huffman code lengths for a 256-bin histogram.
 * Returns max code length (0 when fewer than 2 symbols are present). */
int fqzo_huf_code_lengths(const uint32_t count[256], uint8_t nbits[256]);
/* Canonical zstd code values from lengths (RFC 8878 4.2.1.3). */
void fqzo_huf_codes(const uint8_t nbits[256], int max_bits, uint16_t code[256]);
/* Huffman_Tree_Description bytes; returns size, 0 if not representable. */
size_t fqzo_huf_write_tree(const uint8_t nbits[256], int max_bits, uint8_t *dst);
/* One zstd block for chunk [src, src+m): returns bytes written. */
size_t fqzo_encode_chunk(const uint8_t *src, size_t m, int last, uint8_t *dst);
size_t fqzo_encode_group(const uint8_t *src, size_t M, int last, uint8_t *dst); /* up to FQZO_GROUP chunks sharing one Huffman table */

/* ---- whole-file pipeline: compress.Compress / compress.Decompress ------ */
typedef struct {
    uint32_t block_size;    /* Options.BlockSize (compress.go:75); 0 -> 100000 */
    int workers;            /* Options.Workers (compress.go:76); 0 -> all cores */
    uint32_t batch_records; /* records per block; 0 -> 100000 (batchPool, compress.go:48-52) */
    int entropy;            /* 0 = FQZ-H2 frames (Huffman literals); 1 = system libzstd level 1 via dlopen (CPU-baseline leg);
                             * 2 = container version 3 (FQZ-R1: FQZ-H2 with rANS-coded qualities) */
    int force_encoding;     /* 0 = DetectEncoding on the first batch (compress.go:146-154); 1 = Phred+33, 2 = Phred+64: a shard of a
                             * file whose first batch another process saw (multi-GPU sharding: rank 0 detects and broadcasts) */
    int block_index;        /* entropy == 2 only: 1 = append the block table behind the last block (include/fqz.h: FQZ-R1's optional
                             * on-disk block index) */
    int framing;            /* entropy == 0 only: 0 = FQZ-H2 group framing (the default of the HIP encoder), 1 = FQZ-S1 segment framing for
                             * every block that qualifies (experimental: FQZ_BATCH_SEG / FQZ_ENC_SEG=1 there) */
} fqzo_options;

size_t fqzo_compress_bound(size_t n_bytes);
/* compress.go:125-192.  Returns compressed size or negative error. */
long fqzo_compress(const uint8_t *fastq, size_t n, uint8_t *out, size_t cap, const fqzo_options *opt);
/* compress.go:558-604.  Returns FASTQ size or negative error. If out==NULL
 * only the output size is computed. */
long fqzo_decompress(const uint8_t *fqz, size_t n, uint8_t *out, size_t cap, int workers);

/* libzstd (dlopen) availability for the CPU-baseline leg: version number or 0. */
unsigned fqzo_libzstd_version(void);

#ifdef __cplusplus
}
#endif
#endif
