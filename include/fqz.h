/*
 * fqz.h — C ABI of libfqzhip: the MI355X-native per-block FASTQ codec that
 * replaces fqpack's block hot path.
 *
 * Drop-in boundary (reference paths are relative to vertti/fastqpacker):
 *   one call per block where internal/compress/compress.go calls
 *   compressBlockWithBuffers (compress.go:471-555, callers :204,:228,:294) and
 *   decompressJobToPooledBuffer / decompressBlockToWriter (compress.go:780-837,
 *   :870-890, callers :622,:684,:767), plus primitive mirrors of
 *   internal/encoder (sequence.go, quality.go) and internal/fqformat
 *   (container.go).  All entry points are extern "C", take plain pointers and
 *   sizes, and return 0 on success or a negative fqz_status.  The cgo binding
 *   a maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: an fqz_ctx is single-threaded, like the per-worker zstd
 * encoder/decoder it replaces (compress.go:281, :671); different contexts are
 * independent.  The library never retains caller pointers past return.
 *
 * Wire format: container framing is the reference's (container.go:11-152,
 * SURVEY.md App. A).  Each of the six stream payloads is a sequence of standard
 * zstd frames (RFC 8878; the "FQZ-H2" profile of DESIGN.md section 4: a
 * skippable index frame, then frames of Raw / RLE / Compressed blocks with
 * Huffman-coded literals, sequences on the predefined FSE tables for the
 * headers stream, and content checksums), so the stock `fqpack -d`
 * (zstd.Decoder.DecodeAll, compress.go:785-814) reads our output.
 */
#ifndef FQZ_H
#define FQZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes <-> the reference's error strings (fqz_strerror). */
typedef enum {
    FQZ_OK = 0,
    FQZ_E_SHORT = -1,          /* unexpected EOF in a file/block header */
    FQZ_E_MAGIC = -2,          /* "invalid magic bytes: not an FQZ file"                 container.go:54 */
    FQZ_E_BLOCK_VERSION = -3,  /* "unsupported block header version"                     container.go:111,150 */
    FQZ_E_FILE_VERSION = -4,   /* "unsupported file version: %d"                         compress.go:572 */
    FQZ_E_HDR_AT = -5,         /* "invalid FASTQ: header line must start with @"         parser.go:143 */
    FQZ_E_SEP_PLUS = -6,       /* "invalid FASTQ: separator line must start with +"      parser.go:164 */
    FQZ_E_LEN_MISMATCH = -7,   /* "invalid FASTQ: sequence and quality lengths must match" parser.go:180 */
    FQZ_E_LONG_N = -8,         /* "... ambiguous bases beyond position 65536 ..."        compress.go:484 */
    FQZ_E_TRUNC_HEADER = -9,   /* "truncated header data"                                compress.go:979 */
    FQZ_E_TRUNC_PLUS = -10,    /* "truncated plus-line payload data"                     compress.go:1002 */
    FQZ_E_TRUNC_SEQ = -11,     /* "truncated sequence data"                              compress.go:1020 */
    FQZ_E_TRUNC_QUAL = -12,    /* "truncated quality data"                               compress.go:1033 */
    FQZ_E_TRUNC_LEN = -13,     /* "truncated length data"                                compress.go:1048 */
    FQZ_E_TRUNC_NPOS = -14,    /* "truncated N position data"                            compress.go:1057 */
    FQZ_E_ENTROPY = -15,       /* "decompressing <stream>: ..." (bad / unsupported zstd frame) compress.go:787-813 */
    FQZ_E_READ_DATA = -16,     /* "reading compressed data: unexpected EOF"              compress.go:732 */
    FQZ_E_NOMEM = -17,
    FQZ_E_DST_SMALL = -18,
    FQZ_E_FIELD_WRAP = -19,    /* header / plus payload / N count > 65535: the reference wraps the u16 silently; we refuse */
    FQZ_E_NPOS_RANGE = -20,    /* N position >= read length (the Go code would panic) */
    FQZ_E_HIP = -30,           /* HIP runtime error (fqz_last_hip_error) */
    FQZ_E_NO_DEVICE = -31,     /* no HIP device: the product path never falls back to CPU */
    FQZ_E_ARG = -32,
    FQZ_E_TOO_LARGE = -33,     /* batch >= 2 GiB of FASTQ text (device offsets are 32 bits); slice the input */
    FQZ_E_IO = -34,
    FQZ_E_CHECKSUM = -35      /* a zstd frame's Content_Checksum does not match its decoded content */
} fqz_status;

const char *fqz_strerror(int status);
const char *fqz_last_hip_error(void);
/* "libfqzhip <version> gfx950" */
const char *fqz_version(void);

/* ---- constants of the reference API ------------------------------------ */
#define FQZ_DEFAULT_BLOCK_SIZE 100000u /* compress.DefaultBlockSize          compress.go:71 */
#define FQZ_MAX_SEQUENCE_LENGTH 65536u /* encoder.MaxSequenceLength          sequence.go:11 */
#define FQZ_PHRED33_OFFSET 33          /* encoder.Phred33Offset              quality.go:5 */
#define FQZ_PHRED64_OFFSET 64          /* encoder.Phred64Offset              quality.go:6 */
#define FQZ_ENCODING_PHRED33 0         /* encoder.EncodingPhred33            quality.go:14 */
#define FQZ_ENCODING_PHRED64 1         /* encoder.EncodingPhred64            quality.go:15 */
#define FQZ_FLAG_PAIRED_END 0x01       /* fqformat.FlagPairedEnd             container.go:15 */
#define FQZ_FLAG_PHRED64 0x02          /* fqformat.FlagPhred64               container.go:16 */
#define FQZ_VERSION1 1                 /* fqformat.Version1                  container.go:21 */
#define FQZ_VERSION2 2                 /* fqformat.Version2 (= CurrentVersion) container.go:22-24 */
/* Version 3 ("FQZ-R1"; not a reference format: the stock decoder rejects it by its version byte,
 * compress.go:571-573): a version-2 file whose quality payloads carry interleaved-rANS blocks (zstd's reserved block type
 * 3) in the place of Huffman-coded literals - order-0 entropy instead of >= 1 bit a symbol.  Written on request only
 * (fqz_options.container_version = 3 / FQZ_BATCH_V3); read by every decode entry point. */
#define FQZ_VERSION3 3
#define FQZ_FILE_HEADER_SIZE 10
#define FQZ_ENTROPY_CHUNK 16384u       /* bytes of a pre-entropy stream per zstd block */

/* ---- context ------------------------------------------------------------ */
typedef struct fqz_ctx fqz_ctx;
/* Binds to HIP device `device`; owns a stream and all device workspaces. */
int fqz_ctx_create(int device, fqz_ctx **out);
void fqz_ctx_destroy(fqz_ctx *ctx);
int fqz_device_count(void);

/* ---- fqformat: container framing (host side, byte-exact) ---------------- */
typedef struct { /* fqformat.FileHeader container.go:28-32 */
    uint8_t version;
    uint32_t block_size;
    uint8_t flags;
} fqz_file_header;
typedef struct { /* fqformat.BlockHeader container.go:70-80 */
    uint32_t num_records, seq_size, qual_size, header_size, plus_size;
    uint32_t npos_size, lengths_size, original_seq_size, original_qual_size;
} fqz_block_header;
void fqz_write_file_header(const fqz_file_header *h, uint8_t out[10]);                     /* FileHeader.Write   container.go:35 */
int fqz_read_file_header(const uint8_t *in, size_t n, fqz_file_header *h);                 /* ReadFileHeader     container.go:48 */
int fqz_write_block_header(const fqz_block_header *b, uint8_t version, uint8_t *out);      /* BlockHeader.Write  container.go:83; returns 32/36 */
int fqz_read_block_header(const uint8_t *in, size_t n, uint8_t version, fqz_block_header *b); /* ReadBlockHeader container.go:116; returns 32/36 */

/* ---- the per-block hot path (host buffers in, host buffers out) ---------- */
/* Upper bound of fqz_encode_block / fqz_encode_batch output for n_bytes of FASTQ. */
size_t fqz_encode_bound(size_t n_bytes);
/* The same for blocks of records_per_block records (0 = FQZ_DEFAULT_BLOCK_SIZE): small blocks pay a block header and six
 * payload framings (index, frame header, checksum) each. */
size_t fqz_encode_bound_blocks(size_t n_bytes, uint32_t records_per_block);

/* Replaces compressBlockWithBuffers (compress.go:471-555).  `fastq` holds whole
 * 4-line records (what fqparser.ReadBatch handed to the block codec); the GPU
 * indexes the lines itself (parser.go:136-243 semantics: '@'/'+' checks, CR
 * stripping, seq/qual length check).  Writes the 36-byte v2 block header
 * followed by the six payloads (seq, qual, headers, plus, nPos, lengths).
 * qual_encoding: FQZ_ENCODING_PHRED33 / _PHRED64. */
int fqz_encode_block(fqz_ctx *ctx, const uint8_t *fastq, size_t n_bytes, int qual_encoding,
                     uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *n_records);

/* Replaces decompressJobToPooledBuffer (compress.go:780-837): `block` is the
 * block header (32 B for version 1, 36 B for version 2) followed by its
 * payloads; writes FASTQ text. */
int fqz_decode_block(fqz_ctx *ctx, const uint8_t *block, size_t n, uint8_t version, int qual_encoding,
                     uint8_t *out, size_t out_cap, size_t *out_len);
/* FASTQ bytes fqz_decode_block will produce for this block (needs the GPU: it
 * entropy-decodes the lengths/headers/plus streams). */
int fqz_decode_block_size(fqz_ctx *ctx, const uint8_t *block, size_t n, uint8_t version, size_t *out_len);

/* ---- device-resident batches: the measured path -------------------------- */
typedef struct {
    uint32_t n_records;      /* records encoded (whole blocks unless `final`) */
    uint32_t n_blocks;
    uint64_t consumed;       /* FASTQ bytes consumed (ends on a record boundary) */
    uint64_t out_len;        /* bytes written to d_out */
    int32_t  status;         /* fqz_status of the device pipeline */
    uint32_t error_record;   /* record index for parser errors */
    int32_t  qual_encoding;  /* encoding used (detected if requested) */
    uint32_t n_chunks;       /* entropy chunks (zstd blocks) produced */
    uint64_t stream_raw[6];  /* pre-entropy bytes per stream, whole batch */
    uint64_t stream_comp[6]; /* payload bytes per stream, whole batch */
} fqz_batch_result;

#define FQZ_DETECT_ENCODING (-1)  /* run encoder.DetectEncoding (quality.go:22) over the first block on the GPU */
#define FQZ_BATCH_V3 2u           /* write FQZ_VERSION3 blocks: the quality stream in rANS blocks (the caller writes version 3 into the file header) */
#define FQZ_BATCH_SEG 4u          /* experimental: FQZ-S1 segment framing (one zstd frame per 64 KiB of a block's text and stream) for
                                   * every block that qualifies; any zstd decoder reads it, ours at the general path's speed */
#define FQZ_BATCH_HALVES 8u       /* experimental: a batch of 192 MiB or more runs as two halves in flight on two child contexts (same bytes out;
                                   * slower than one piece today - the host cannot queue two halves' launches fast enough; experiments/README.md) */
#define FQZ_BATCH_FINAL 1u        /* last batch of the input: a short last block is emitted, an unterminated tail is dropped (parser.go:210-220) */

/* Encodes records_per_block-record blocks from FASTQ text already in HBM.
 * d_fastq must start at a record boundary; with !FINAL only whole blocks are
 * encoded and `consumed` tells the caller where the next batch starts.
 * d_out receives the blocks back to back (no file header); block_off[b] /
 * block_len[b] (host arrays, may be NULL, capacity max_blocks) locate them.
 * `stream` is a hipStream_t (NULL = the context's own stream).  Synchronous
 * on return: the result struct is valid. n_bytes must be < 2 GiB (fqz_compress slices larger inputs into batches). */
int fqz_encode_batch_dev(fqz_ctx *ctx, const uint8_t *d_fastq, size_t n_bytes, uint32_t records_per_block,
                         int qual_encoding, uint32_t flags, uint8_t *d_out, size_t out_cap,
                         fqz_batch_result *res, uint64_t *block_off, uint64_t *block_len, size_t max_blocks,
                         void *stream);

/* Asynchronous halves of the above for benchmarking / overlap: launch enqueues
 * every kernel on `stream` without a host sync; finish waits and fills `res`.
 * finish returns FQZ_E_TOO_LARGE when the text has more (or much shorter)
 * lines than the workspace was sized for; it has then resized the context,
 * and the same launch + finish succeeds when repeated (at most twice;
 * fqz_encode_batch_dev does this by itself). */
int fqz_encode_batch_launch(fqz_ctx *ctx, const uint8_t *d_fastq, size_t n_bytes, uint32_t records_per_block,
                            int qual_encoding, uint32_t flags, uint8_t *d_out, size_t out_cap, void *stream);
int fqz_encode_batch_finish(fqz_ctx *ctx, fqz_batch_result *res, uint64_t *block_off, uint64_t *block_len,
                            size_t max_blocks);

/* Decodes `n_blocks` consecutive blocks (headers + payloads, as in a .fqz file
 * after the 10-byte file header) resident in HBM into FASTQ text in d_out. */
int fqz_decode_batch_dev(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding,
                         uint8_t *d_out, size_t out_cap, fqz_batch_result *res, void *stream);
/* The same with a hint: block_off[i] (host array) = offset of block i's header in d_blocks, as fqz_encode_batch_dev reports
 * them and as a reader that cut the batch out of a file knows them (the reference reads header and payloads block by block,
 * compress.go:721-758).  Saves the walk along the chain of block headers on the device (one dependent memory round trip per
 * block).  A hint, not trusted: offsets that do not add up are ignored and the chain is walked. */
int fqz_decode_batch_dev_hint(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding,
                              uint8_t *d_out, size_t out_cap, fqz_batch_result *res, const uint64_t *block_off, size_t n_blocks,
                              void *stream);
int fqz_decode_batch_launch(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding,
                            uint8_t *d_out, size_t out_cap, void *stream);
int fqz_decode_batch_finish(fqz_ctx *ctx, fqz_batch_result *res);

/* Test hook: the six pre-entropy streams of the last
 * fqz_encode_* call's block `block`, copied to host memory.  stream_len[k] is
 * in/out (capacity in, size out); streams[k] may be NULL to query sizes. */
int fqz_debug_get_streams(fqz_ctx *ctx, uint32_t block, uint8_t *streams[6], size_t stream_len[6]);

/* Diagnostic hook (FQZ_DBG_STAMPS=1 in the environment): per-chunk s_memtime stamps of the entropy kernel's phases. */
int fqz_debug_get_stamps(fqz_ctx *ctx, unsigned long long *out, size_t max_chunks, size_t *n_chunks);

/* ---- internal/encoder primitive mirrors (GPU-executed, host buffers) ----- */
/* encoder.PackBases / AppendPackedBases (sequence.go:58,139): packed gets
 * (n+3)/4 bytes, npos the positions (< 65536) of non-ACGTacgt bytes. */
int fqz_pack_bases(fqz_ctx *ctx, const uint8_t *seq, size_t n, uint8_t *packed, uint16_t *npos, size_t *n_npos);
/* encoder.UnpackBases / AppendUnpackBases (sequence.go:103,188). */
int fqz_unpack_bases(fqz_ctx *ctx, const uint8_t *packed, const uint16_t *npos, size_t n_npos, size_t seq_len, uint8_t *seq);
/* encoder.DetectEncoding (quality.go:22) over n strings laid out back to back. */
int fqz_detect_encoding(fqz_ctx *ctx, const uint8_t *quals, const uint64_t *offsets /* n+1 */, size_t n, int *encoding);
/* encoder.NormalizeQuality / DenormalizeQuality (quality.go:53,66), in place. */
int fqz_normalize_quality(fqz_ctx *ctx, uint8_t *qual, size_t n, int encoding);
int fqz_denormalize_quality(fqz_ctx *ctx, uint8_t *qual, size_t n, int encoding);
/* encoder.DeltaEncode / DeltaDecode (quality.go:81,107), in place, one read. */
int fqz_delta_encode(fqz_ctx *ctx, uint8_t *qual, size_t n);
int fqz_delta_decode(fqz_ctx *ctx, uint8_t *qual, size_t n);

/* ---- entropy stage alone (replaces zstd EncodeAll / DecodeAll) ----------- */
size_t fqz_entropy_bound(size_t n);
int fqz_entropy_encode(fqz_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len);  /* compress.go:523-528 */
int fqz_entropy_decode(fqz_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len);  /* compress.go:785-814 */

/* ---- compress.Compress / compress.Decompress (host pipeline over the GPU) -- */
typedef struct {          /* compress.Options compress.go:74-77 */
    uint32_t block_size;  /* BlockSize: written to the file header only; 0 -> 100000 */
    int32_t  workers;     /* Workers: kept for API parity; the GPU pipeline sizes itself. 0 -> default */
    uint32_t container_version; /* 0 or 2: CurrentVersion (container.go:24); 3: FQZ_VERSION3 */
    uint32_t block_index;       /* version 3 only: 1 = append the block table (below) behind the last block */
} fqz_options;

/* Block table of a version-3 file ("optional on-disk block index"; not a reference format).  Behind the last
 * block:  u32 0xFFFFFFFF (where a block header's NumRecords would stand: never a valid count) | 'FQZX' | u32 n_blocks |
 * n_blocks x { u64 offset of the block header from the start of the file, u32 NumRecords } | u64 offset of the table | 'FQZX'.
 * Readers of the block chain stop at it; a reader with random access finds it from the end of the file.
 * fqz_read_block_table: host only, no device work.  off / n_records may be NULL (only *n_blocks is wanted); FQZ_E_DST_SMALL
 * if cap is too small; FQZ_E_ARG if the file carries no table. */
#define FQZ_BLOCK_TABLE_MARK 0xFFFFFFFFu
int fqz_read_block_table(const uint8_t *fqz, size_t n, uint64_t *off, uint32_t *n_records, size_t cap, size_t *n_blocks);
typedef struct {          /* compress.DecompressOptions compress.go:80-82 */
    int32_t workers;
} fqz_decompress_options;

/* compress.Compress (compress.go:125) on memory buffers. opts may be NULL. */
int fqz_compress(fqz_ctx *ctx, const uint8_t *fastq, size_t n, uint8_t *out, size_t out_cap, size_t *out_len,
                 const fqz_options *opts);
/* compress.Decompress (compress.go:558). out==NULL: only *out_len is computed. */
int fqz_decompress(fqz_ctx *ctx, const uint8_t *fqz, size_t n, uint8_t *out, size_t out_cap, size_t *out_len,
                   const fqz_decompress_options *opts);
/* The reference's own shape: compress.Compress(io.Reader, io.Writer, *Options) compress.go:125 and
 * compress.Decompress(io.Reader, io.Writer, *DecompressOptions) :558.  read returns the bytes it put into dst
 * (0 = end of input, < 0 = error), write returns 0 on success.  Streaming: three slices are in flight (H2D of the next,
 * kernels of this one, D2H of the previous one); memory use does not grow with the input. */
typedef long (*fqz_read_fn)(void *user, uint8_t *dst, size_t cap);
typedef int (*fqz_write_fn)(void *user, const uint8_t *src, size_t n);
int fqz_compress_stream(fqz_ctx *ctx, fqz_read_fn read, void *read_user, fqz_write_fn write, void *write_user, const fqz_options *opts);
int fqz_decompress_stream(fqz_ctx *ctx, fqz_read_fn read, void *read_user, fqz_write_fn write, void *write_user,
                          const fqz_decompress_options *opts);
/* compress.Decompress into a buffer the library allocates (one decode, no sizing pass); release it with fqz_buffer_free. */
int fqz_decompress_alloc(fqz_ctx *ctx, const uint8_t *fqz, size_t n, uint8_t **out, size_t *out_len, const fqz_decompress_options *opts);
void fqz_buffer_free(uint8_t *p);
/* compress.Compress over several devices: the reference's worker pool + ordered collector (compress.go:240-278, 365-403)
 * as one host thread and one context per entry of devices[] (an entry may repeat).  Every device encodes a contiguous
 * range of whole 100 000-record blocks; the output is byte-identical to fqz_compress on one device. */
int fqz_compress_multi(const int *devices, int n_devices, const uint8_t *fastq, size_t n, uint8_t *out, size_t out_cap, size_t *out_len,
                       const fqz_options *opts);
/* compress.Decompress over several devices (readNextDecompressJob + worker pool + ordered writer, compress.go:630-758): the
 * block headers are walked on the host, every device decodes a contiguous range of whole blocks, the texts are concatenated
 * in order.  The library allocates *out (release it with fqz_buffer_free); identical to fqz_decompress_alloc. */
int fqz_decompress_multi(const int *devices, int n_devices, const uint8_t *fqz, size_t n, uint8_t **out, size_t *out_len,
                         const fqz_decompress_options *opts);
/* File-to-file forms used by the fqpack CLI driver (cmd/fqpack/main.go:190-203); stream through the same pipeline. */
int fqz_compress_file(fqz_ctx *ctx, const char *in_path, const char *out_path, const fqz_options *opts);
int fqz_decompress_file(fqz_ctx *ctx, const char *in_path, const char *out_path, const fqz_decompress_options *opts);

/* ---- per-kernel timing (HIP events on the launch stream) ------------------- */
/* When enabled, every kernel launch of the encode/decode pipelines is bracketed by
 * hipEventRecord on the stream it is launched on; fqz_profile_read accumulates the
 * elapsed times per kernel name since the last reset. */
int fqz_profile_enable(fqz_ctx *ctx, int on); /* 0 off, 1 every kernel, 2 only the dominant encode kernel (k_entropy), 3 only k_rans (version 3) */
int fqz_profile_reset(fqz_ctx *ctx);
/* names: '\n'-separated kernel names (NUL-terminated); ms[i] total milliseconds, calls[i] launches. */
int fqz_profile_read(fqz_ctx *ctx, char *names, size_t names_cap, double *ms, uint32_t *calls, size_t max_entries, size_t *n_entries);

/* ---- synthetic workload (bench / tests) ------ */
typedef struct {
    uint64_t seed;
    uint64_t first_record;   /* global index of the first record (shards differ) */
    uint32_t min_len, max_len;
    uint32_t n_permille;     /* per-mille of N bases (runs, mean length 3) */
    uint32_t phred;          /* 33 or 64 */
    uint32_t quality_profile;/* 0 = 4-level binned Markov, 1 = 41-level random walk */
} fqz_synth_params;
/* Writes whole records until fewer than one worst-case record fits in cap;
 * returns bytes written through *out_len and the record count. Host code. */
int fqz_synth_fastq(const fqz_synth_params *p, uint64_t n_records, uint8_t *out, size_t cap, size_t *out_len,
                    uint64_t *n_written);

#ifdef __cplusplus
}
#endif
#endif /* FQZ_H */
