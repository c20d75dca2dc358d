// fqz_internal.h — shared declarations of libfqzhip (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "fqz.h"

#define FQZ_CHUNK 16384u               // pre-entropy bytes per zstd block (== FQZ_ENTROPY_CHUNK)
#define FQZ_SLOT (FQZ_CHUNK + 64u)     // per-chunk staging slot in HBM (raw worst case 3+16384, padded for aligned reads)
#define FQZ_ENT 12u                    // entry points per zstd block in the FQZI index: 3 per Huffman stream (fqz_entropy_dev.h, k_dec_huf)
#define FQZ_SLOT_ENT (FQZ_CHUNK + 32u) // ... and where the encoder leaves them in the chunk's slot (behind the largest block, 3 + 16384 bytes)
#define FQZ_GROUP 4u                   // chunks that share one Huffman table (one workgroup of k_entropy)
#define FQZ_TILE 4096u                 // text bytes per line-index workgroup
#define FQZ_NS 6
enum { S_SEQ = 0, S_QUAL = 1, S_HDR = 2, S_PLUS = 3, S_NPOS = 4, S_LEN = 5 };

#define FQZ_HUF_MAX_BITS 11

// Device-resident scalar state of one encode batch.
struct EncInfo {
    uint32_t n_lines;       // '\n' count of the batch text
    uint32_t n_rec;         // records encoded
    uint32_t n_rec_total;   // whole records present
    uint32_t n_blocks;
    uint32_t consumed;      // text bytes consumed
    int32_t  status;        // fqz_status
    uint32_t error_record;
    uint32_t min_qual;      // min quality byte of block 0 (DetectEncoding)
    uint32_t qual_off;      // 33 or 64
    uint32_t n_chunks;      // all chunks: ids [0, n_main) = seq/qual/headers/plus/lengths, [n_main, n_chunks) = nPos
    uint32_t n_main;
    uint32_t n_groups;      // chunk groups (k_group_map)
    uint32_t arena_used;    // bytes of the main arena in use
    uint32_t npos_used;
    uint32_t index_overflow; // a 4 KiB tile holds more lines than its tile-local slot (k_line_local)
    uint32_t n_xgroups;     // all chunk groups = zstd frames, the Raw ones of the packed bases included (k_xxh)
    uint32_t n_hchunks;     // chunks of the headers streams (modelled before the entropy stage, fqz_hdrlz.h)
    uint32_t n_hgroups;     // their groups
    uint32_t n_rgroups;     // groups of the quality streams when they are coded with rANS (container version 3, fqz_rans.h)
    uint32_t n_segs;        // FQZ-S1 path (fqz_seg.h): segments of the batch
    uint32_t seg_fallback;  // some block does not qualify for the segment framing: the host encodes the batch's blocks the FQZ-H2 way where flagged
    uint32_t eh_used;       // entries of the headers record-offset table in use
    uint32_t pages_used;    // SEG_PAGE units of the slot pool handed out
    unsigned long long sarena_used; // bytes of the stream arena handed out
    unsigned long long error_key; // (record << 8 | check order << 4 | code index), min wins
    unsigned long long out_len;
    unsigned long long stream_raw[FQZ_NS];
    unsigned long long stream_comp[FQZ_NS];
};

// Per-block plan: where each pre-entropy stream of block b lives and its chunks.
struct BlockPlan {
    uint32_t rec0, nrec;
    uint32_t a_off[FQZ_NS];      // byte offset in the arena (S_NPOS: in the npos arena), 16-aligned
    uint32_t len[FQZ_NS];        // pre-entropy bytes
    uint32_t chunk_base[FQZ_NS]; // first chunk id
    uint32_t frame_off[FQZ_NS];  // offset of the payload in d_out (filled by layout)
    uint32_t frame_len[FQZ_NS];
    uint32_t out_off, out_len;   // block header + payloads in d_out
    uint32_t orig_seq;           // sum of read lengths
    uint32_t seg_base, n_seg;    // FQZ-S1: the block's segments
    uint32_t fallback;           // FQZ-S1: a segment of the block does not qualify
};

// Decode side -------------------------------------------------------------
struct DecInfo {
    int32_t  status;
    uint32_t n_blocks;
    uint32_t n_rec;
    uint32_t n_chunks;
    unsigned long long out_len;
    unsigned long long stream_raw[FQZ_NS];
    unsigned long long stream_comp[FQZ_NS];
    uint32_t n_rgroups;     // groups with rANS blocks (version-3 files; appended by k_dec_index)
    uint32_t n_groups;      // groups of 64 records (k_dec_sizes sums them, the scans run over the groups, k_dec_assemble scans inside one)
};

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return fqz_set_hip_error(_e, #expr); \
    } while (0)

int fqz_set_hip_error(hipError_t e, const char *what);

static inline size_t fqz_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
