// fqz_decode_lz.h — decoder for zstd frames that carry LZ sequences (RFC 8878 3.1.1.3.2), i.e. the payloads the
// stock fqpack writes with klauspost/compress (internal/compress/compress.go:523-528) or any other zstd encoder.
// Our own encoder never emits sequences, so this is the interoperability path of SURVEY §8 row f-3: it lets
// fqz_decompress read .fqz files that were not written by this library.
//
// One wave per payload (one frame, content size required).  Blocks of a frame depend on each other (matches reach
// back into earlier blocks, repeat offsets and entropy tables carry over), so the wave walks them in order:
//   literals   -> a per-wave scratch buffer (raw / RLE / Huffman with 1 or 4 streams, lane k decodes stream k;
//                 "treeless" blocks reuse the previous table)
//   sequences  -> every lane decodes the FSE bit stream redundantly with wave-uniform values (scalar unit), then all
//                 64 lanes copy the literals and the match of that sequence (overlapping matches as a pattern fill)
// Included by fqz_decode.hip only.
#pragma once

struct FseDEntry { uint8_t sym, nb; uint16_t base; };

struct LzLds {
    uint16_t huf[4096];        // Huffman decode table (sym | nbits << 8), kept for treeless blocks
    FseDEntry ll[512], of[256], ml[512];
    uint8_t w[260];            // Huffman weights
    uint8_t scratch[320];      // FSE tables of the weight decoder
    short norm[64];
    int huf_log, ll_log, of_log, ml_log, have_huf, have_ll, have_of, have_ml;
};

__constant__ const short c_ll_default[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ const short c_ml_default[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                             1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ const short c_of_default[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__constant__ const uint32_t c_ll_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                             48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__constant__ const uint8_t c_ll_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ const uint32_t c_ml_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                             30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                             16387, 32771, 65539};
__constant__ const uint8_t c_ml_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                            0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

// FSE decoding table from normalised counts (-1 = "less than one"); lane 0 only.  Returns 0 or -1.
__device__ int lz_build_dtable(const short *norm, int max_sym, int table_log, FseDEntry *dt)
{
    const int table_size = 1 << table_log;
    uint16_t next[64];
    int high = table_size - 1;
    for (int s = 0; s <= max_sym; s++) {
        if (norm[s] == -1) { dt[high--].sym = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    const int step = (table_size >> 1) + (table_size >> 3) + 3, mask = table_size - 1;
    int pos = 0;
    for (int s = 0; s <= max_sym; s++)
        for (int i = 0; i < norm[s]; i++) {
            dt[pos].sym = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    if (pos != 0) return -1;
    for (int u = 0; u < table_size; u++) {
        const int s = dt[u].sym;
        const uint32_t ns = next[s]++;
        const int nb = table_log - highbit32_d(ns);
        dt[u].nb = (uint8_t)nb;
        dt[u].base = (uint16_t)((ns << nb) - (uint32_t)table_size);
    }
    return 0;
}

// FSE NCount header (RFC 8878 4.1.1); lane 0 only.  Returns bytes used or -1.
__device__ int lz_read_ncount(const uint8_t *src, uint32_t n, int max_sym_allowed, int max_log, short *norm, int *max_sym_out, int *log_out)
{
    if (n < 1) return -1;
    uint32_t bitpos = 0;
    auto peek = [&](int nb) -> uint32_t {
        unsigned long long v = 0;
        for (int i = 0; i < 5; i++) { uint32_t q = (bitpos >> 3) + i; if (q < n) v |= (unsigned long long)src[q] << (8 * i); }
        return (uint32_t)(v >> (bitpos & 7)) & ((1u << nb) - 1);
    };
    const int table_log = (int)peek(4) + 5;
    bitpos += 4;
    if (table_log > max_log) return -1;
    const int table_size = 1 << table_log;
    int remaining = table_size + 1, threshold = table_size, nb = table_log + 1, sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= max_sym_allowed) {
        if (prev0) {
            int n0 = sym;
            for (;;) { uint32_t r = peek(2); bitpos += 2; n0 += (int)r; if (r != 3) break; }
            if (n0 > max_sym_allowed + 1) return -1;
            while (sym < n0) norm[sym++] = 0;
            if (sym > max_sym_allowed) break;
        }
        const int max = (2 * threshold - 1) - remaining;
        int count;
        const uint32_t lowv = peek(nb - 1);
        if ((int)lowv < max) { count = (int)lowv; bitpos += nb - 1; }
        else { count = (int)peek(nb); if (count >= threshold) count -= max; bitpos += nb; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (short)count;
        prev0 = !count;
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (bitpos > 8 * n) return -1;
    }
    if (remaining != 1 || sym < 1) return -1;
    *max_sym_out = sym - 1;
    *log_out = table_log;
    return (int)((bitpos + 7) >> 3);
}

// one of the three sequence tables, according to its compression mode; every lane calls it, lane 0 builds.
// Returns bytes consumed from src or -1.  (mode: 0 predefined, 1 RLE, 2 FSE, 3 repeat)
__device__ int lz_seq_table(LzLds &L, int which /*0 LL, 1 OF, 2 ML*/, int mode, const uint8_t *src, uint32_t n)
{
    FseDEntry *dt = which == 0 ? L.ll : (which == 1 ? L.of : L.ml);
    int *logp = which == 0 ? &L.ll_log : (which == 1 ? &L.of_log : &L.ml_log);
    int *havep = which == 0 ? &L.have_ll : (which == 1 ? &L.have_of : &L.have_ml);
    const int max_sym = which == 0 ? 35 : (which == 1 ? 31 : 52), max_log = which == 1 ? 8 : 9;
    int used = 0, rc = 0;
    if (threadIdx.x == 0) {
        if (mode == 0) {
            const short *d = which == 0 ? c_ll_default : (which == 1 ? c_of_default : c_ml_default);
            const int ns = which == 0 ? 36 : (which == 1 ? 29 : 53), lg = which == 1 ? 5 : 6;
            for (int i = 0; i < ns; i++) L.norm[i] = d[i];
            rc = lz_build_dtable(L.norm, ns - 1, lg, dt);
            *logp = lg;
            *havep = 1;
        } else if (mode == 1) {
            if (n < 1 || src[0] > max_sym) rc = -1;
            else { dt[0].sym = src[0]; dt[0].nb = 0; dt[0].base = 0; *logp = 0; *havep = 1; used = 1; }
        } else if (mode == 2) {
            int ms, lg;
            used = lz_read_ncount(src, n, max_sym, max_log, L.norm, &ms, &lg);
            if (used < 0) rc = -1;
            else { rc = lz_build_dtable(L.norm, ms, lg, dt); *logp = lg; *havep = 1; }
        } else if (!*havep) rc = -1;
        L.norm[60] = (short)rc;
        L.norm[61] = (short)used;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    rc = L.norm[60];
    used = L.norm[61];
    __builtin_amdgcn_wave_barrier();
    return rc < 0 ? -1 : used;
}

#define LZU(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

// Decodes the frame that starts at `in` (n bytes are there: the frame and whatever follows it) into dst[0, cap); a frame
// that states its content size must produce exactly that.  lit = per-wave scratch (128 KiB).  Every lane of the
// (single-wave) workgroup calls this with the same arguments.  Returns 0, -1 (malformed) or -2 (checksum); *used = the
// bytes of the frame, *made = its content bytes.
__device__ int lz_decode_frame(LzLds &L, const uint8_t *in, uint32_t n, uint8_t *dst, uint32_t cap, uint8_t *lit, uint32_t *used, uint32_t *made)
{
    const uint32_t lane = threadIdx.x;
    uint32_t hdr;
    long long fcs;
    int ck;
    *used = 0; *made = 0;
    if (LZU(frame_header(in, n, &hdr, &fcs, &ck) < 0)) return -1;
    if (fcs >= 0 && (unsigned long long)fcs > cap) return -1;
    const uint32_t raw = fcs >= 0 ? LZU((uint32_t)fcs) : cap;
    uint32_t pos = LZU(hdr), out = 0;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    if (lane == 0) { L.have_huf = L.have_ll = L.have_of = L.have_ml = 0; }
    for (;;) {
        if (n - pos < 3) return -1;
        const uint32_t bh = LZU(in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16));
        const uint32_t last = bh & 1, type = (bh >> 1) & 3, bs = bh >> 3;
        pos += 3;
        if (type == 3 || bs > 128 * 1024) return -1;
        if (type == 0) { // raw block
            if (bs > n - pos || bs > raw - out) return -1;
            for (uint32_t i = lane; i < bs; i += 64) dst[out + i] = in[pos + i];
            out += bs;
            pos += bs;
        } else if (type == 1) { // RLE block
            if (n - pos < 1 || bs > raw - out) return -1;
            const uint8_t v = in[pos];
            for (uint32_t i = lane; i < bs; i += 64) dst[out + i] = v;
            out += bs;
            pos += 1;
        } else {
            if (bs > n - pos || bs < 2) return -1;
            const uint8_t *b = in + pos;
            // ---- literals section
            const uint32_t lh0 = LZU(b[0]);
            const uint32_t lt = lh0 & 3, fmt = (lh0 >> 2) & 3;
            uint32_t lregen, lcsize = 0, lhdr, nstreams = 1;
            if (lt <= 1) {
                if (fmt == 0 || fmt == 2) { lregen = lh0 >> 3; lhdr = 1; }
                else if (fmt == 1) { if (bs < 2) return -1; lregen = (lh0 >> 4) | (LZU(b[1]) << 4); lhdr = 2; }
                else { if (bs < 3) return -1; lregen = (lh0 >> 4) | (LZU(b[1]) << 4) | (LZU(b[2]) << 12); lhdr = 3; }
                lcsize = lt == 0 ? lregen : 1;
            } else {
                if (bs < 5) return -1;
                const unsigned long long h5 = (unsigned long long)LZU(b[0]) | ((unsigned long long)LZU(b[1]) << 8) | ((unsigned long long)LZU(b[2]) << 16) |
                                              ((unsigned long long)LZU(b[3]) << 24) | ((unsigned long long)LZU(b[4]) << 32);
                if (fmt <= 1) { lregen = (uint32_t)(h5 >> 4) & 0x3FF; lcsize = (uint32_t)(h5 >> 14) & 0x3FF; lhdr = 3; nstreams = fmt == 0 ? 1 : 4; }
                else if (fmt == 2) { lregen = (uint32_t)(h5 >> 4) & 0x3FFF; lcsize = (uint32_t)(h5 >> 18) & 0x3FFF; lhdr = 4; nstreams = 4; }
                else { lregen = (uint32_t)(h5 >> 4) & 0x3FFFF; lcsize = (uint32_t)(h5 >> 22) & 0x3FFFF; lhdr = 5; nstreams = 4; }
            }
            if (lregen > 128 * 1024 || lhdr + lcsize > bs) return -1;
            const uint8_t *lp = b + lhdr;
            const uint8_t *litp = lit; // where the literals of this block can be read from
            if (lt == 0) litp = lp;    // raw literals: read them in place
            else if (lt == 1) { const uint8_t v = lp[0]; for (uint32_t i = lane; i < lregen; i += 64) lit[i] = v; }
            else {
                uint32_t tree_used = 0;
                if (lt == 2) {
                    int used = 0;
                    if (lane == 0) { int lg = 0; used = huf_read_table_dev(lp, lcsize, L.huf, &lg, L.w, L.scratch); L.huf_log = lg; L.have_huf = used >= 0; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    used = (int)LZU(used);
                    if (used < 0) return -1;
                    tree_used = (uint32_t)used;
                } else if (!LZU(L.have_huf)) return -1;
                __builtin_amdgcn_wave_barrier();
                const int hlog = (int)LZU(L.huf_log);
                const uint8_t *sp = lp + tree_used;
                const uint32_t sbytes = lcsize - tree_used;
                int bad = 0;
                if (nstreams == 1) {
                    if (lane == 0) bad = huf_decode_stream_dev(sp, sbytes, L.huf, hlog, lit, lregen) < 0;
                } else {
                    if (sbytes < 6) return -1;
                    const uint32_t s1 = LZU(sp[0] | ((uint32_t)sp[1] << 8)), s2 = LZU(sp[2] | ((uint32_t)sp[3] << 8)), s3 = LZU(sp[4] | ((uint32_t)sp[5] << 8));
                    if (6 + s1 + s2 + s3 > sbytes) return -1;
                    const uint32_t s4 = sbytes - 6 - s1 - s2 - s3, seg = (lregen + 3) / 4;
                    if (3 * seg > lregen) return -1;
                    if (lane < 4) {
                        const uint32_t soff = 6 + (lane > 0 ? s1 : 0) + (lane > 1 ? s2 : 0) + (lane > 2 ? s3 : 0);
                        const uint32_t slen = lane == 0 ? s1 : (lane == 1 ? s2 : (lane == 2 ? s3 : s4));
                        const uint32_t cnt = lane < 3 ? seg : lregen - 3 * seg;
                        bad = huf_decode_stream_dev(sp + soff, slen, L.huf, hlog, lit + lane * seg, cnt) < 0;
                    }
                }
                if (__ballot(bad)) return -1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // the literals are read by other lanes below
            }
            // ---- sequences section
            const uint8_t *sq = b + lhdr + lcsize;
            uint32_t sn = bs - lhdr - lcsize;
            if (sn < 1) return -1;
            uint32_t nseq = LZU(sq[0]), shdr = 1;
            if (nseq >= 128) {
                if (nseq < 255) { if (sn < 2) return -1; nseq = ((nseq - 128) << 8) + LZU(sq[1]); shdr = 2; }
                else { if (sn < 3) return -1; nseq = LZU(sq[1]) + (LZU(sq[2]) << 8) + 0x7F00; shdr = 3; }
            }
            uint32_t lpos = 0; // literals consumed
            if (nseq) {
                if (sn < shdr + 1) return -1;
                const uint32_t modes = LZU(sq[shdr]);
                if (modes & 3) return -1;
                uint32_t tp = shdr + 1;
                for (int which = 0; which < 3; which++) {
                    const int mode = (int)((modes >> (6 - 2 * which)) & 3);
                    const int used = lz_seq_table(L, which, mode, sq + tp, sn - tp);
                    if (used < 0) return -1;
                    tp += (uint32_t)used;
                }
                if (tp >= sn) return -1;
                const int ll_log = (int)LZU(L.ll_log), of_log = (int)LZU(L.of_log), ml_log = (int)LZU(L.ml_log);
                BackBits br;
                if (LZU(bb_init(br, sq + tp, sn - tp) < 0)) return -1;
                uint32_t st_ll = bb_read(br, ll_log), st_of = bb_read(br, of_log), st_ml = bb_read(br, ml_log);
                for (uint32_t i = 0; i < nseq; i++) {
                    const FseDEntry e_ll = L.ll[LZU(st_ll)], e_of = L.of[LZU(st_of)], e_ml = L.ml[LZU(st_ml)];
                    const uint32_t of_code = LZU(e_of.sym), ml_code = LZU(e_ml.sym), ll_code = LZU(e_ll.sym);
                    if (of_code > 31 || ml_code > 52 || ll_code > 35) return -1;
                    const uint32_t of_val = (1u << of_code) + LZU(bb_read(br, (int)of_code));
                    const uint32_t ml = c_ml_base[ml_code] + LZU(bb_read(br, c_ml_bits[ml_code]));
                    const uint32_t ll = c_ll_base[ll_code] + LZU(bb_read(br, c_ll_bits[ll_code]));
                    if (i + 1 < nseq) {
                        st_ll = LZU(e_ll.base) + LZU(bb_read(br, (int)LZU(e_ll.nb)));
                        st_ml = LZU(e_ml.base) + LZU(bb_read(br, (int)LZU(e_ml.nb)));
                        st_of = LZU(e_of.base) + LZU(bb_read(br, (int)LZU(e_of.nb)));
                    }
                    if (br.remaining < 0) return -1;
                    // repeat offsets (RFC 8878 3.1.1.5)
                    uint32_t offset;
                    if (of_val > 3) { offset = of_val - 3; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                    else {
                        const uint32_t idx = of_val + (ll == 0 ? 1u : 0u);
                        if (idx == 1) offset = rep0;
                        else if (idx == 2) { offset = rep1; rep1 = rep0; rep0 = offset; }
                        else if (idx == 3) { offset = rep2; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                        else { offset = rep0 - 1; if (!offset) return -1; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                    }
                    if (ll > lregen - lpos || ll > raw - out || ml > raw - out - ll || offset > out + ll) return -1;
                    for (uint32_t k = lane; k < ll; k += 64) dst[out + k] = litp[lpos + k];
                    lpos += ll;
                    out += ll;
                    // the match may overlap its own output: byte k comes from out - offset + (k mod offset), all of which exist
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    const uint8_t *ms = dst + out - offset;
                    if (offset >= ml) { for (uint32_t k = lane; k < ml; k += 64) dst[out + k] = ms[k]; }
                    else { for (uint32_t k = lane; k < ml; k += 64) dst[out + k] = ms[k % offset]; }
                    out += ml;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                }
            } else if (sn != shdr) return -1;
            // the literals after the last sequence
            const uint32_t tail = lregen - lpos;
            if (tail > raw - out) return -1;
            for (uint32_t k = lane; k < tail; k += 64) dst[out + k] = litp[lpos + k];
            out += tail;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            pos += bs;
        }
        if (last) break;
    }
    if (ck) { // Content_Checksum: low 32 bits of XXH64 over the frame's content (four lanes hash; a 15 MB stock frame is a long serial chain)
        if (n - pos < 4 || (fcs >= 0 && out != raw)) return -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const unsigned long long h = xxh64_quad(dst, lane < 4 ? out : 0u, lane);
        const uint32_t want = in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)h) != want) return -2;
        pos += 4;
    }
    *used = pos;
    *made = out;
    return (fcs < 0 || out == raw) ? 0 : -1;
}
