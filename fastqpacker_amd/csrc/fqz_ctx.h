// fqz_ctx.h — the context object behind the opaque fqz_ctx of include/fqz.h.
#pragma once
#include <string.h>
#include "fqz_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return FQZ_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = fqz_align_up(bytes + bytes / 8, 1 << 20);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fqz_set_hip_error(e, "hipMalloc(workspace)");
        cap = want;
        return FQZ_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return (T *)p; }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return FQZ_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = fqz_align_up(bytes + bytes / 8, 1 << 16);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return fqz_set_hip_error(e, "hipHostMalloc");
        cap = want;
        return FQZ_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return (T *)p; }
};

// extras of a launch that is one half of a batch encoded as two halves in flight (fqz_api.hip: encode_batch_halves)
struct EncLaunchExtra {
    uint32_t skip = 0;                 // the text proper starts at d_text[skip] (< 16)
    int phase = 0;                     // 0 whole launch, 1 front end only, 2 the rest (enc_launch_groups)
    bool want_front = false;           // copy the counters to h_front behind the front end (line index, record table, plan) and record ev_front
    bool want_layout_event = false;    // record ev_layout behind k_layout
    const EncInfo *prev_info = nullptr; // device: the launch whose blocks lie in front of this one's in d_out
    hipEvent_t prev_layout = nullptr;  // ... and the event behind its k_layout
};

struct EncState {
    EncLaunchExtra extra;
    PinnedBuf h_front;
    hipEvent_t ev_front = nullptr, ev_layout = nullptr;
    // inputs of the batch in flight
    const uint8_t *d_text = nullptr;
    size_t n_bytes = 0;
    uint32_t rpb = 0;
    uint32_t flags = 0;
    uint8_t *d_out = nullptr;
    size_t out_cap = 0;
    hipStream_t stream = nullptr;
    bool in_flight = false;
    hipStream_t side = nullptr;            // the headers chain (model, sequences, entropy) runs here beside the entropy coder of the other streams
    hipStream_t side2 = nullptr;           // the content checksums
    hipStream_t side3 = nullptr;           // the headers' Sequences_Sections (serial FSE state chains)
    hipEvent_t ev_join2 = nullptr, ev_join3 = nullptr, ev_npos = nullptr, ev_gmap = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool two_pass_now = false;   // the launch in flight used the two-pass index
    uint32_t two_pass_left = 0;  // > 0 after a batch overflowed a tile-local line slot: this many launches count newlines first, then index
    // capacities
    uint32_t n_tiles = 0, line_cap = 0, rec_cap = 0, block_cap = 0, chunk_cap = 0;
    size_t arena_cap = 0, npos_cap = 0;
    // device workspaces
    DevBuf info;      // EncInfo
    DevBuf tile_cnt;  // u32[n_tiles+1]
    DevBuf ls;        // u32[line_cap+1] line starts
    DevBuf lf;        // u8[line_cap+1] line flags: '\r' before the newline | first-byte class << 1
    DevBuf E;         // u32[5][rec_cap+1]: seq, qual, hdr, plus, npos sizes -> exclusive offsets
    DevBuf zstate;    // look-back states and tickets of the batch's scans (zeroed by k_init)
    DevBuf scan_state; // look-back states of k_scan + its ticket
    DevBuf gmap;      // chunk-group descriptors (k_group_map)
    DevBuf hside;     // headers model: sequences | literals | Sequences_Sections | HdrSide | chunk list (fqz_hdrlz.h)
    uint32_t plans_pre = 0;           // block plans already copied to the host with the counters (fqz_enc_launch)
    uint32_t hcap = 0, hcap_need = 0; // headers chunks the side buffers hold / the last batch needed
    size_t hcap_need_bytes = 0;        // the size of the batch that reported hcap_need (a relaunch of it takes the exact need)
    double hcap_per_mb = 0;           // headers chunks per MiB of text of the last batch that overflowed the optimistic size
    DevBuf xmap;      // descriptors of every group (frame) for the content checksums | xsum[chunk_cap]
    DevBuf plans;     // BlockPlan[block_cap]
    DevBuf arena;     // seq/qual/hdr/plus/len pre-entropy streams
    DevBuf npos;      // nPos pre-entropy streams
    DevBuf slots;     // chunk_cap * FQZ_SLOT
    DevBuf csize;     // u32[chunk_cap+1] -> exclusive prefix
    DevBuf stamps;    // diagnostic s_memtime stamps (FQZ_DBG_STAMPS)
    bool streams_valid = false; // an encode has run: fqz_debug_get_streams can read its streams
    // FQZ-S1 path (fqz_seg.h)
    bool path_seg = false;      // the launch in flight took the segment path
    bool groups_once = false;   // the next launch takes the group path whatever the default (fqz_enc_finish: blocks that do not qualify; test hooks)
    bool no_mixed = false;      // inside enc_mixed
    int qual_encoding = 0;
    DevBuf segmeta;   // u32 bstart[block_cap + 2] | seg_base[block_cap + 2]
    DevBuf seg;       // SegInfo[seg_cap + 2]
    DevBuf hslots;    // headers blocks (FQZ_SLOT each) | SegHdrJob[hcap] | hcsize[hcap] | hord[hcap]
    PinnedBuf h_info; // EncInfo
    PinnedBuf h_plans;
};

struct DecState {
    const uint8_t *d_in = nullptr;
    size_t n_bytes = 0;
    uint8_t *d_out = nullptr;
    size_t out_cap = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;            // bases + qualities are entropy-decoded here while the record walks run on `stream`
    hipStream_t side2 = nullptr;           // the sequence bit streams of the headers blocks (needs the input only)
    hipEvent_t ev_join2 = nullptr, ev_x = nullptr, ev_joinx = nullptr, ev_huf = nullptr, ev_seq = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool in_flight = false;
    DevBuf info, blocks, chunks, frames, streams, rec, partials, tables, lz_scratch;
    PinnedBuf h_info, h_blocks;
    uint32_t n_blocks = 0;
    // arguments of the launch in flight (a decode that guessed the frame layout wrong is relaunched from finish)
    uint8_t version = 0;
    int qual_encoding = 0;
    uint8_t *user_out = nullptr;
    size_t user_cap = 0;
    bool general = false;
    bool skip_assemble = false;            // only the decoded size is wanted (fqz_decode_block_size, fqz_decompress with out == NULL)
    const uint64_t *hint_off = nullptr;    // fqz_decode_batch_dev_hint: where the caller says the block headers are (host array, this call only)
    size_t hint_n = 0;
    DevBuf hint;
    PinnedBuf h_hint;
};

struct ProfEntry { const char *name; hipEvent_t a, b; };
struct ProfTotal { std::string name; double ms; uint32_t calls; };
struct Prof {
    bool on = false;
    bool dominant_only = false; // bracket only the dominant kernel (two events per batch instead of ~40: nothing to perturb the timed region)
    const char *dominant = "k_entropy"; // (k_rans when the timed encode writes container version 3)
    bool armed = false;
    std::vector<ProfEntry> pending;
    std::vector<hipEvent_t> pool;
    std::vector<ProfTotal> totals;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(const char *name, hipStream_t st)
    {
        armed = on && (!dominant_only || !strcmp(name, dominant));
        if (!armed) return;
        ProfEntry e{name, get(), get()};
        (void)hipEventRecord(e.a, st);
        pending.push_back(e);
    }
    void end(hipStream_t st)
    {
        if (!armed || pending.empty()) return;
        (void)hipEventRecord(pending.back().b, st);
        armed = false;
    }
    void collect() // call after the stream has been synchronised
    {
        // FQZ_DBG_TIMELINE=1 (diagnostic): start and duration of every bracketed launch of the batch, from the events themselves -
        // the order of things as they run without a profiler's dispatch overhead (tools/timeline.py needs a rocprofv3 trace)
        static const bool tl = getenv("FQZ_DBG_TIMELINE") && atoi(getenv("FQZ_DBG_TIMELINE"));
        if (tl && !pending.empty()) {
            fprintf(stderr, "[fqz timeline] %zu launches\n", pending.size());
            for (ProfEntry &e : pending) {
                float t0 = 0, d = 0;
                (void)hipEventElapsedTime(&t0, pending.front().a, e.a);
                (void)hipEventElapsedTime(&d, e.a, e.b);
                fprintf(stderr, "[fqz timeline] %9.1f us  +%8.1f us  %s\n", t0 * 1e3, d * 1e3, e.name);
            }
        }
        for (ProfEntry &e : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
                bool found = false;
                for (ProfTotal &t : totals) if (t.name == e.name) { t.ms += ms; t.calls++; found = true; break; }
                if (!found) totals.push_back(ProfTotal{e.name, ms, 1});
            }
            pool.push_back(e.a);
            pool.push_back(e.b);
        }
        pending.clear();
    }
};
// brackets one launch: PROF(ctx, st, "name", hipLaunchKernelGGL(...));
#define PROF(ctx, st, name, launch) do { (ctx)->prof.begin(name, st); launch; (ctx)->prof.end(st); } while (0)

struct fqz_ctx {
    Prof prof;
    int device = 0;
    hipStream_t stream = nullptr;
    EncState enc;
    DecState dec;
    // staging for the host-buffer entry points
    DevBuf d_in, d_out;
    PinnedBuf h_stage;
    DevBuf sl_new[3];             // slices in flight of the streaming pipeline: device side,
    PinnedBuf sl_hin[3], sl_hout[3]; // pinned staging for callback sources / sinks
    std::vector<fqz_ctx *> lanes; // child contexts of the streaming pipeline (fqz_stream.hip): own stream and workspaces each
    fqz_ctx *half[2] = {nullptr, nullptr}; // child contexts of fqz_encode_batch_dev's two halves in flight (fqz_api.hip)
    hipEvent_t ev_half = nullptr;
    bool half_off = false;        // a batch of this context had to be redone in one piece: later ones go there directly
};

// fqz_encode.hip
int fqz_enc_launch(fqz_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint32_t rpb, int qual_encoding, uint32_t flags,
                   uint8_t *d_out, size_t out_cap, hipStream_t stream);
int fqz_enc_launch_ex(fqz_ctx *ctx, const uint8_t *d_text, size_t n_bytes, uint32_t rpb, int qual_encoding, uint32_t flags,
                      uint8_t *d_out, size_t out_cap, hipStream_t stream, const EncLaunchExtra *x);
int fqz_enc_finish(fqz_ctx *ctx, fqz_batch_result *res, uint64_t *block_off, uint64_t *block_len, size_t max_blocks);
int fqz_enc_get_streams(fqz_ctx *ctx, uint32_t block, uint8_t *streams[6], size_t stream_len[6]);
int fqz_enc_get_stamps(fqz_ctx *ctx, unsigned long long *out, size_t max_chunks, size_t *n_chunks);
// fqz_decode.hip
int fqz_dec_launch(fqz_ctx *ctx, const uint8_t *d_blocks, size_t n_bytes, uint8_t version, int qual_encoding, uint8_t *d_out,
                   size_t out_cap, hipStream_t stream);
int fqz_dec_finish(fqz_ctx *ctx, fqz_batch_result *res);
