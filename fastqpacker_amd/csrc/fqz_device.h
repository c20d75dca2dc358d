// fqz_device.h — device-side helpers shared by the encode and decode kernels (gfx950, wave64).
#pragma once
#include "fqz_internal.h"

#define WAVE 64

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// 0x80 in every byte of y that is zero (exact per byte, no cross-byte carries)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t y)
{
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}
__device__ __forceinline__ uint32_t count_newlines(uint32_t x) { return __popc(zero_bytes(x ^ 0x0A0A0A0Au)); }

// bytewise x - y (mod 256 per byte)
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t y)
{
    const uint32_t H = 0x80808080u;
    return ((x | H) - (y & ~H)) ^ ((x ^ ~y) & H);
}
// bytewise x + y
__device__ __forceinline__ uint32_t add_bytes(uint32_t x, uint32_t y)
{
    const uint32_t H = 0x80808080u;
    return ((x & ~H) + (y & ~H)) ^ ((x ^ y) & H);
}

// 0x80 in every byte of x that is one of ACGTacgt.  u = x|0x20 folds case; the
// low 3 bits of a/c/g/t (1,3,7,4) index an 8-entry byte table through v_perm.
__device__ __forceinline__ uint32_t acgt_mask(uint32_t x)
{
    uint32_t u = x | 0x20202020u;
    uint32_t sel = u & 0x07070707u;
    // table bytes: [0]=0 [1]='a' [2]=0 [3]='c' | [4]='t' [5]=0 [6]=0 [7]='g'
    uint32_t e = __builtin_amdgcn_perm(0x67000074u, 0x63006100u, sel);
    return zero_bytes(e ^ u);
}
// 2-bit codes (A0 C1 G2 T3) of 4 ASCII bases packed into one byte; non-ACGT -> 0
__device__ __forceinline__ uint32_t pack4(uint32_t x, uint32_t valid80)
{
    uint32_t c = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    uint32_t keep = (valid80 >> 7) | (valid80 >> 6);
    c &= keep;
    uint32_t p = c | (c >> 6);
    p |= p >> 12;
    return p & 0xFFu;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ uint4 load_u128_unaligned(const uint8_t *p)
{
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void store_u128_unaligned(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }

// inclusive wave scan with DPP only (row shifts inside the rows of 16, then the two row broadcasts of gfx9): no trip through
// the LDS crossbar that __shfl_up (ds_bpermute) takes - which also means it is not slowed down by kernels that saturate LDS
#define FQZ_DPP_ADD(v, ctrl, rows) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v = FQZ_DPP_ADD(v, 0x111, 0xF); // row_shr:1
    v = FQZ_DPP_ADD(v, 0x112, 0xF); // row_shr:2
    v = FQZ_DPP_ADD(v, 0x114, 0xF); // row_shr:4
    v = FQZ_DPP_ADD(v, 0x118, 0xF); // row_shr:8
    v = FQZ_DPP_ADD(v, 0x142, 0xA); // row_bcast:15 -> rows 1 and 3
    v = FQZ_DPP_ADD(v, 0x143, 0xC); // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { uint32_t t = __shfl_xor(v, d, WAVE); v = t < v ? t : v; }
    return v;
}

// Exclusive scan of one value per thread over a 256-thread workgroup.
// sh must hold 4 uint32_t.  Returns the exclusive prefix; *total the sum.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *sh, uint32_t *total)
{
    uint32_t incl = wave_incl_scan(v);
    uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 63) sh[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        uint32_t s = sh[k];
        if (k < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

// ---- piece-centric kernels (k_split, k_dec_assemble): a lane handles one 16-byte piece of one record's line
// smallest i with incl[i] > p, for p < incl[63]; every lane of the wave must call it
__device__ __forceinline__ uint32_t piece_owner(uint32_t incl, uint32_t p)
{
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = 32; step; step >>= 1) {
        uint32_t v = (uint32_t)__shfl((int)incl, (int)(lo + step - 1), WAVE);
        if (v <= p) lo += step;
    }
    return lo & 63;
}

// the first nb (<= 16) bytes of w[] to dst (any alignment): one 128-bit store, or for a tail at most four stores of
// 8 / 4 / 2 / 1 bytes — never a byte loop (a divergent 15-iteration loop per tail piece cost more than the whole rest)
// piece p -> (record i of the wave's 64, piece k inside it).  incl = wave inclusive scan of the per-record piece
// counts cnt.  When every record that has pieces has the same count (fixed-length reads: the common case) the map is a
// division by a wave-uniform constant and needs no cross-lane traffic; otherwise a binary search over incl
// (ds_bpermute).  Every lane of the wave must call these.
struct PieceMap { uint32_t uni; float inv; };
__device__ __forceinline__ PieceMap piece_map_make(uint32_t cnt, uint32_t incl)
{
    PieceMap m;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
    // uniform <=> lanes 0..n-1 all have c0 pieces and the rest none, i.e. incl is min(lane + 1, n) * c0
    const bool odd = c0 == 0 || (cnt != c0 && cnt != 0) || (cnt == 0 && incl != total) || total > (1u << 22);
    m.uni = __ballot(odd) ? 0u : c0;
    m.inv = m.uni ? 1.0f / (float)m.uni : 0.0f;
    return m;
}
__device__ __forceinline__ void piece_locate(const PieceMap &m, uint32_t incl, uint32_t cnt, uint32_t p, uint32_t *i, uint32_t *k)
{
    if (m.uni) {
        uint32_t q = (uint32_t)(((float)p + 0.5f) * m.inv); // p < 2^22; off by one at most, fixed below
        if (q * m.uni > p) q--;
        if ((q + 1) * m.uni <= p) q++;
        q = q > 63 ? 63 : q;
        *i = q;
        *k = p - q * m.uni;
    } else {
        const uint32_t o = piece_owner(incl, p);
        *i = o;
        *k = p - (uint32_t)__shfl((int)(incl - cnt), (int)o, WAVE);
    }
}

__device__ __forceinline__ void store_piece(uint8_t *dst, const uint32_t w[4], uint32_t nb)
{
    if (nb >= 16) { store_u128_unaligned(dst, make_uint4(w[0], w[1], w[2], w[3])); return; }
    unsigned long long cur = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    if (nb & 8) {
        __builtin_memcpy(dst, &cur, 8);
        dst += 8;
        cur = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
    }
    if (nb & 4) { uint32_t v = (uint32_t)cur; __builtin_memcpy(dst, &v, 4); dst += 4; cur >>= 32; }
    if (nb & 2) { uint16_t v = (uint16_t)cur; __builtin_memcpy(dst, &v, 2); dst += 2; cur >>= 16; }
    if (nb & 1) *dst = (uint8_t)cur;
}

// 16 text bytes at text[off..off+16); bytes at or beyond n_text read as 0 (only the very last lines of the text get there)
__device__ __forceinline__ void load_piece(const uint8_t *text, size_t off, size_t n_text, uint32_t w[4])
{
    if (off + 16 <= n_text) {
        uint4 v = load_u128_unaligned(text + off);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (uint32_t b = 0; b < 16 && off + b < n_text; b++) w[b >> 2] |= (uint32_t)text[off + b] << (8 * (b & 3));
    }
}

__device__ __forceinline__ int highbit32_d(uint32_t v) { return 31 - __clz(v); }

__device__ __forceinline__ void report_error(EncInfo *info, uint32_t rec, uint32_t order, int code)
{
    unsigned long long key = ((unsigned long long)rec << 8) | ((unsigned long long)order << 5) | (unsigned long long)(-code);
    atomicMin(&info->error_key, key);
}
