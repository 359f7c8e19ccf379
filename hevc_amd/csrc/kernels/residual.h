// hevc_amd/csrc/kernels/residual.h — K3: forward transform, quantisation, scaling, inverse transform and
// reconstruction of all transform units of one CTU, executed by the CTU's workgroup on its LDS image.
//
// Index space: 1536 samples per CTU = 1024 luma (32x32) + 256 Cb (16x16) + 256 Cr.  The TU a sample belongs to is
// looked up in `tu_log2[16]` (CU size per 8x8 luma tile, 0 = no TU there); chroma TUs are the co-located half-size
// blocks.  Arithmetic: H.265 8.6.2-8.6.4 (scaling, transformation) for the decoder side; the conventional
// two-stage integer forward transform and dead-zone quantiser for the encoder side — identical, term for term, to
// oracle/hevc_oracle.c (orc_fwd_transform, orc_quant, orc_dequant, orc_inv_transform).
// Roofline note: 4 matrix passes of <=32 MACs per sample; the working set never leaves LDS (12 KiB per CTU).
#pragma once
#include "common.h"

namespace mihevc {

struct ResidualShared {
    int16_t mat[32][32];       // LDS copy of the transform matrix, 16-bit so rows feed v_dot2_i32_i16 as dword pairs
    int16_t mat_t[32][32];     // its transpose (inverse transform reads columns)
    int16_t res[1536];         // residual in, reconstructed residual out
    int16_t tmp[1536];         // stage intermediates (forward stage 1 output is stored TRANSPOSED inside its TU)
    int16_t coef[1536];        // transform coefficients, then scaled coefficients (stored TRANSPOSED inside the TU)
    int16_t lvl[1536];         // quantised levels (TU-local raster at CTU coordinates)
    uint32_t desc[1536];       // per sample: TU geometry, written once per TU map by residual_describe
    uint8_t tu_log2[16];       // per 8x8 luma tile: log2 of the TU (= CU) size, 0 = none
    uint8_t tu_intra[16];      // per tile: 1 = intra rounding
    unsigned cbf[3];           // bit t set: tile t's TU has a non-zero level in that plane (all tiles of a TU set the TU's first tile bit)
};

struct SampleLoc {
    int plane, x, y;           // coordinates inside the CTU's plane (luma 0..31, chroma 0..15)
    int log2n, tx0, ty0;       // TU geometry in the same coordinates; log2n == 0: not covered
    int tile0;                 // 8x8 tile index of the TU origin (for the cbf word)
    int intra;
    int stride, base;          // row stride and base offset of the plane inside the 1536-sample arrays
};

// A square luma region of the CTU (cx, cy, size 2^log2n) plus its two co-located chroma blocks: 1.5 n^2 samples.
// Phases that only concern one CU enumerate the region instead of all 1536 CTU samples.
struct Region {
    int cx, cy, log2n;
    DEV int count() const { return (1 << (2 * log2n)) + (1 << (2 * log2n - 1)); }
    DEV int index(int k) const      // k-th sample of the region -> index into the 1536-sample CTU arrays
    {
        const int n2 = 1 << (2 * log2n);
        if (k < n2) return (cy + (k >> log2n)) * 32 + cx + (k & ((1 << log2n) - 1));
        k -= n2;
        const int q = n2 >> 2, pl = k >= q, kk = pl ? k - q : k, l2 = log2n - 1;
        return 1024 + pl * 256 + ((cy >> 1) + (kk >> l2)) * 16 + (cx >> 1) + (kk & ((1 << l2) - 1));
    }
};
DEV Region whole_ctu() { return Region{0, 0, 5}; }

DEV SampleLoc locate(const ResidualShared &s, int idx)
{
    SampleLoc l;
    if (idx < 1024) { l.plane = 0; l.x = idx & 31; l.y = idx >> 5; l.stride = 32; l.base = 0; }
    else { int i = idx - 1024; l.plane = 1 + (i >> 8); i &= 255; l.x = i & 15; l.y = i >> 4; l.stride = 16; l.base = 1024 + (l.plane - 1) * 256; }
    int sh = l.plane ? 2 : 3;                       // samples per tile edge: 8 luma, 4 chroma
    int tile = (l.y >> sh) * 4 + (l.x >> sh);
    int lg = s.tu_log2[tile];
    l.intra = s.tu_intra[tile];
    if (!lg) { l.log2n = 0; l.tx0 = l.ty0 = l.tile0 = 0; return l; }
    int lgp = l.plane ? lg - 1 : lg;
    l.log2n = lgp;
    l.tx0 = l.x & ~((1 << lgp) - 1);
    l.ty0 = l.y & ~((1 << lgp) - 1);
    l.tile0 = (l.ty0 >> sh) * 4 + (l.tx0 >> sh);
    return l;
}
// packed form kept in LDS so the five transform phases do one read instead of re-deriving the geometry
DEV uint32_t pack_loc(const SampleLoc &l) { return (uint32_t)l.log2n | (uint32_t)l.tx0 << 3 | (uint32_t)l.ty0 << 8 | (uint32_t)l.tile0 << 13 | (uint32_t)l.intra << 17; }
DEV SampleLoc unpack_loc(uint32_t d, int idx)
{
    SampleLoc l;
    if (idx < 1024) { l.plane = 0; l.x = idx & 31; l.y = idx >> 5; l.stride = 32; l.base = 0; }
    else { int i = idx - 1024; l.plane = 1 + (i >> 8); i &= 255; l.x = i & 15; l.y = i >> 4; l.stride = 16; l.base = 1024 + (l.plane - 1) * 256; }
    l.log2n = (int)(d & 7); l.tx0 = (int)(d >> 3) & 31; l.ty0 = (int)(d >> 8) & 31; l.tile0 = (int)(d >> 13) & 15; l.intra = (int)(d >> 17) & 1;
    return l;
}

// sum_{i<n} a[i] * b[i] over contiguous, 4-byte aligned int16 runs (n even)
DEV int dot_i16(const int16_t *a, const int16_t *b, int n)
{
    int acc = 0;
    for (int i = 0; i < n; i += 2) acc = dot2_i16(load_u32_aligned(a + i), load_u32_aligned(b + i), acc);
    return acc;
}

// forward + quant + scaling + inverse for every TU of the region; s.desc must describe the region's samples
// (callers fill it while they form the residual).  qp / qp_c are syntax QPs.
template <class Ex> DEV void residual_pipeline(Ex &ex, ResidualShared &s, int qp, int qp_c, int bit_depth, Region rg)
{
    const int cnt = rg.count();
    ex.phase([&](int tid) {      // forward stage 1: rows; result stored transposed inside the TU
        for (int k = tid; k < cnt; k += NT) {
            const int idx = rg.index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) continue;
            int n = 1 << l.log2n, sh1 = l.log2n + bit_depth - 9, u = l.x - l.tx0, yl = l.y - l.ty0;
            int acc = dot_i16(s.mat[u << (5 - l.log2n)], s.res + l.base + l.y * l.stride + l.tx0, n);
            s.tmp[l.base + (l.ty0 + u) * l.stride + l.tx0 + yl] = (int16_t)(sh1 > 0 ? (acc + (1 << (sh1 - 1))) >> sh1 : acc);
        }
    });
    ex.phase([&](int tid) {      // forward stage 2: columns (contiguous thanks to the transposed intermediate)
        for (int k = tid; k < cnt; k += NT) {
            const int idx = rg.index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) continue;
            int n = 1 << l.log2n, sh2 = l.log2n + 6, u = l.x - l.tx0, v = l.y - l.ty0;
            int acc = dot_i16(s.mat[v << (5 - l.log2n)], s.tmp + l.base + (l.ty0 + u) * l.stride + l.tx0, n);
            acc = (acc + (1 << (sh2 - 1))) >> sh2;
            s.coef[idx] = (int16_t)clip3(-32768, 32767, acc);
        }
    });
    ex.phase([&](int tid) {      // quantisation + scaling (8.6.4.1, flat m = 16); scaled value stored transposed
        for (int k = tid; k < cnt; k += NT) {
            const int idx = rg.index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) { s.lvl[idx] = 0; continue; }
            int q = (l.plane ? qp_c : qp) + 6 * (bit_depth - 8);
            int qbits = 14 + q / 6 + (15 - bit_depth - l.log2n);
            long long add = (long long)(l.intra ? 171 : 85) << (qbits - 9);
            int c = s.coef[idx];
            long long a = ((long long)iabs(c) * g_tab.quant_scale[q % 6] + add) >> qbits;
            if (a > 32767) a = 32767;
            int lev = (int)(c < 0 ? -a : a);
            s.lvl[idx] = (int16_t)lev;
            if (lev) ex.atomic_or(&s.cbf[l.plane], 1u << l.tile0);
            int bd_shift = bit_depth + l.log2n - 5;
            long long scale = (long long)16 * g_tab.level_scale[q % 6] << (q / 6);
            long long d = (lev * scale + ((long long)1 << (bd_shift - 1))) >> bd_shift;
            s.tmp[l.base + (l.ty0 + (l.x - l.tx0)) * l.stride + l.tx0 + (l.y - l.ty0)] = (int16_t)(d < -32768 ? -32768 : d > 32767 ? 32767 : d);
        }
    });
    ex.phase([&](int tid) {      // inverse stage 1: columns, shift 7, clip to 16 bit (8.6.4.2)
        for (int k = tid; k < cnt; k += NT) {
            const int idx = rg.index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) continue;
            int n = 1 << l.log2n, xl = l.x - l.tx0, yy = l.y - l.ty0, step = 5 - l.log2n;
            // e[x][y] = sum_j transMatrix[j][y] * d[x][j]: column x of the scaled block is row x of the transposed store
            int acc = 0;
            const int16_t *dq = s.tmp + l.base + (l.ty0 + xl) * l.stride + l.tx0;
            for (int j = 0; j < n; j += 2) {
                uint32_t m2 = (uint32_t)(uint16_t)s.mat_t[yy][j << step] | (uint32_t)(uint16_t)s.mat_t[yy][(j + 1) << step] << 16;
                acc = dot2_i16(m2, load_u32_aligned(dq + j), acc);
            }
            s.coef[idx] = (int16_t)clip3(-32768, 32767, (acc + 64) >> 7);
        }
    });
    ex.phase([&](int tid) {      // inverse stage 2: rows, shift 20 - bitDepth
        for (int k = tid; k < cnt; k += NT) {
            const int idx = rg.index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) { s.res[idx] = 0; continue; }
            int n = 1 << l.log2n, xx = l.x - l.tx0, step = 5 - l.log2n, sh = 20 - bit_depth;
            int acc = 0;
            const int16_t *g = s.coef + l.base + l.y * l.stride + l.tx0;
            for (int j = 0; j < n; j += 2) {
                uint32_t m2 = (uint32_t)(uint16_t)s.mat_t[xx][j << step] | (uint32_t)(uint16_t)s.mat_t[xx][(j + 1) << step] << 16;
                acc = dot2_i16(m2, load_u32_aligned(g + j), acc);
            }
            s.res[idx] = (int16_t)((acc + (1 << (sh - 1))) >> sh);
        }
    });
}

template <class Ex> DEV void residual_init(Ex &ex, ResidualShared &s)
{
    ex.phase([&](int tid) {
        for (int i = tid; i < 1024; i += NT) { s.mat[i >> 5][i & 31] = g_tab.mat[i >> 5][i & 31]; s.mat_t[i & 31][i >> 5] = g_tab.mat[i >> 5][i & 31]; }
        if (tid < 16) { s.tu_log2[tid] = 0; s.tu_intra[tid] = 0; }
        if (tid < 3) s.cbf[tid] = 0;
    });
}

// coefficient-rate estimate of one 4x4 sub-block of levels, in 1/16 bit (oracle/hevc_oracle.c code_tu)
DEV int subblock_bits_q4(const int16_t *lv, int stride)
{
    int bits = 0, any = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int a = iabs(lv[y * stride + x]);
            if (!a) continue;
            any = 1;
            bits += a == 1 ? 40 : a == 2 ? 60 : 64 + 32 * ilog2u((unsigned)(a - 1));
        }
    return any ? bits + 24 : 0;
}

}  // namespace mihevc
