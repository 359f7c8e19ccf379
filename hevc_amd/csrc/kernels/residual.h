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

// pair tables: offsets of the 4-, 8-, 16-, 32-point sections (n/2 x n dwords each)
DEV int pair_off(int log2n) { return log2n == 2 ? 0 : log2n == 3 ? 8 : log2n == 4 ? 40 : 168; }

struct ResidualShared {
    // transform matrices as dword pairs for v_dot2_i32_i16, laid out so that the lanes of a wave (consecutive x)
    // read consecutive dwords: mp[off + p*n + u] = (M[u][2p], M[u][2p+1]),  mq[off + p*n + y] = (M[2p][y], M[2p+1][y])
    alignas(16) uint32_t mp[680];
    alignas(16) uint32_t mq[680];
    union {
    struct {
    alignas(16) int16_t res[1536];         // residual in, reconstructed residual out
    alignas(16) int16_t tmp[1536];         // stage intermediates, row-pair interleaved: element (r, c) at (r & ~1) * stride + 2c + (r & 1)
    alignas(16) int16_t coef[1536];        // transform coefficients / inverse stage-1 output (natural layout)
    alignas(16) int16_t lvl[1536];         // quantised levels (TU-local raster at CTU coordinates)
    uint32_t desc[1536];       // per sample: TU geometry, written by the caller while it forms the residual
    };
    alignas(16) uint32_t scratch[4608];    // the same 18 KB for a caller's own use while no residual is in flight (k_inter_ctu: fractional search)
    struct { alignas(16) long long d[192]; int b[192]; } cg;      // over `res`, which is idle between the forward and the inverse transform: per 2x4 block, its share of a coefficient group's distortion and bits
    };
    uint8_t tu_log2[16];       // per 8x8 luma tile: log2 of the TU (= CU) size, 0 = none
    uint8_t tu_intra[16];      // per tile: 1 = intra rounding
    unsigned cbf[3];           // bit t set: tile t's TU has a non-zero level in that plane (all tiles of a TU set the TU's first tile bit)
    int quant_scale[6], level_scale[6];   // LDS copies: no global-memory read on the per-sample path
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
    // k-th block of 2 rows x 4 columns (count() / 8 of them: luma first, then Cb, Cr) -> index of its top-left sample
    DEV int block_index(int k) const
    {
        const int nb = 1 << (2 * log2n - 3);             // luma blocks: (n / 2) x (n / 4)
        if (k < nb) { const int per = 1 << (log2n - 2); return (cy + 2 * (k >> (log2n - 2))) * 32 + cx + 4 * (k & (per - 1)); }
        k -= nb;
        const int q = nb >> 2, pl = k >= q, kk = pl ? k - q : k, l2 = log2n - 3;      // chroma plane: (n / 4) x (n / 8) blocks
        return 1024 + pl * 256 + ((cy >> 1) + 2 * (kk >> l2)) * 16 + (cx >> 1) + 4 * (kk & ((1 << l2) - 1));
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

// 16 / 8 bytes of an LDS image whose address is 16- / 8-byte aligned as one ds_read_b128 / ds_read_b64 (and the matching stores)
DEV void load_x4(const void *p, uint32_t (&v)[4]) { __builtin_memcpy(v, __builtin_assume_aligned(p, 16), 16); }
DEV void load_x2(const void *p, uint32_t (&v)[2]) { __builtin_memcpy(v, __builtin_assume_aligned(p, 8), 8); }
DEV void store_x4(void *p, const uint32_t (&v)[4]) { __builtin_memcpy(__builtin_assume_aligned(p, 16), v, 16); }
DEV void store_x2(void *p, const uint32_t (&v)[2]) { __builtin_memcpy(__builtin_assume_aligned(p, 8), v, 8); }

// forward + quant + scaling + inverse for every TU of the region; s.desc must describe the region's samples
// (callers fill it while they form the residual).  qp / qp_c are syntax QPs.
// One lane owns a BLOCK of 2 rows x 4 columns of outputs per stage (TUs are at least 4x4 and 4-aligned, so a block never straddles a
// TU): a matrix operand fetched once (ds_read_b128 / b64) feeds 8 v_dot2_i32_i16, the sample operands come as b128 rows or as the
// row-pair dwords of `tmp`, results leave as b64 / b128 stores.  The one-output-per-lane form spent two ds_read_b32 per dot2 and was
// LDS-issue bound (a third of k_inter_ctu's instructions, profiles/r02 phase table); a CTU is 192 blocks = one pass of the workgroup.
// cg_lam_q4 > 0 (inter CTUs): RD zero-out of 4x4 coefficient groups, oracle cg_zero_out — dropping a group adds D = sum r (2c - r) of squared
// coefficient error (c coefficient, r its reconstruction) and saves its levels' bits + one sub-block; drop <=> 16 D < ((cg_lam_q4 * bits) >> 4) << 2 (15 - bitDepth - log2n).
// A group is two vertically adjacent 2x4 blocks, i.e. two lanes: both leave their partial sums in LDS (`res` is free between the forward and the
// inverse transform) and the next phase lets each decide for its own half.
template <class Ex> DEV void residual_pipeline(Ex &ex, ResidualShared &s, int qp, int qp_c, int bit_depth, Region rg, int cg_lam_q4 = 0)
{
    const int nblk = rg.count() >> 3;
    long long *cg_d = s.cg.d;
    int *cg_b = s.cg.b;
    // row stages (forward 1, inverse 2): out(y, u) = sum_p M-pair(p, u) . in(y, 2p..2p+1); column stages (forward 2, inverse 1):
    // out(v, x) = sum_p M-pair(p, v) . in-row-pair(p, x), the row pairs of `tmp` being single dwords
    auto row_stage = [&](const uint32_t *mat, const int16_t *in, const SampleLoc &l, int (&acc)[2][4]) {
        const int n = 1 << l.log2n, u4 = l.x - l.tx0;
        const uint32_t *m = mat + pair_off(l.log2n) + u4;
        const int16_t *r0 = in + l.base + l.y * l.stride + l.tx0, *r1 = r0 + l.stride;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[j][i] = 0;
        if (n == 4) {
            uint32_t d0[2], d1[2];
            load_x2(r0, d0); load_x2(r1, d1);
#pragma unroll
            for (int p = 0; p < 2; p++) {
                uint32_t mm[4];
                load_x4(m + p * 4, mm);
#pragma unroll
                for (int i = 0; i < 4; i++) { acc[0][i] = dot2_i16(mm[i], d0[p], acc[0][i]); acc[1][i] = dot2_i16(mm[i], d1[p], acc[1][i]); }
            }
            return;
        }
        for (int p0 = 0; p0 < n / 2; p0 += 4) {
            uint32_t d0[4], d1[4];
            load_x4(r0 + 2 * p0, d0); load_x4(r1 + 2 * p0, d1);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t mm[4];
                load_x4(m + (p0 + j) * n, mm);
#pragma unroll
                for (int i = 0; i < 4; i++) { acc[0][i] = dot2_i16(mm[i], d0[j], acc[0][i]); acc[1][i] = dot2_i16(mm[i], d1[j], acc[1][i]); }
            }
        }
    };
    auto col_stage = [&](const uint32_t *mat, const SampleLoc &l, int (&acc)[2][4]) {
        const int n = 1 << l.log2n, v2 = l.y - l.ty0;
        const uint32_t *m = mat + pair_off(l.log2n) + v2;
        const int16_t *t = s.tmp + l.base + l.ty0 * l.stride + 2 * l.x;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[j][i] = 0;
#pragma unroll 4
        for (int p = 0; p < n / 2; p++) {
            uint32_t d[4], mm[2];
            load_x4(t + 2 * p * l.stride, d);
            load_x2(m + p * n, mm);
#pragma unroll
            for (int i = 0; i < 4; i++) { acc[0][i] = dot2_i16(mm[0], d[i], acc[0][i]); acc[1][i] = dot2_i16(mm[1], d[i], acc[1][i]); }
        }
    };
    // rows (y, y+1) x columns (x..x+3) into the row-pair interleaved `tmp`: four dwords
    auto store_pairs = [&](const SampleLoc &l, const int (&v)[2][4]) {
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = pack_lo16(v[0][i], v[1][i]);
        store_x4(s.tmp + l.base + l.y * l.stride + 2 * l.x, o);
    };
    auto store_rows = [&](int16_t *dst, const SampleLoc &l, const int (&v)[2][4]) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            uint32_t o[2] = {pack_lo16(v[j][0], v[j][1]), pack_lo16(v[j][2], v[j][3])};
            store_x2(dst + l.base + (l.y + j) * l.stride + l.x, o);
        }
    };
    ex.phase([&](int tid) {      // forward stage 1: rows
        for (int k = tid; k < nblk; k += NT) {
            const int idx = rg.block_index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) continue;
            const int sh1 = l.log2n + bit_depth - 9;
            int acc[2][4];
            row_stage(s.mp, s.res, l, acc);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[j][i] = sh1 > 0 ? (acc[j][i] + (1 << (sh1 - 1))) >> sh1 : acc[j][i];
            store_pairs(l, acc);
        }
    });
    ex.phase([&](int tid) {      // forward stage 2: columns
        for (int k = tid; k < nblk; k += NT) {
            const int idx = rg.block_index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n) continue;
            const int sh2 = l.log2n + 6;
            int acc[2][4];
            col_stage(s.mp, l, acc);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[j][i] = clip3(-32768, 32767, (acc[j][i] + (1 << (sh2 - 1))) >> sh2);
            store_rows(s.coef, l, acc);
        }
    });
    ex.phase([&](int tid) {      // quantisation + scaling (8.6.4.1, flat m = 16)
        for (int k = tid; k < nblk; k += NT) {
            const int idx = rg.block_index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            int lev[2][4], deq[2][4];
            if (!l.log2n) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) lev[j][i] = 0;
                store_rows(s.lvl, l, lev);
                continue;
            }
            const int q = (l.plane ? qp_c : qp) + 6 * (bit_depth - 8);
            const int qbits = 14 + q / 6 + (15 - bit_depth - l.log2n), bd_shift = bit_depth + l.log2n - 5;
            const long long add = (long long)(l.intra ? 171 : 85) << (qbits - 9);
            const long long scale = (long long)16 * s.level_scale[q % 6] << (q / 6);
            const int qs = s.quant_scale[q % 6];
            const bool cg = cg_lam_q4 > 0 && !l.intra;
            int any = 0, bits = 0;
            long long dsum = 0;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                uint32_t c2[2];
                load_x2(s.coef + l.base + (l.y + j) * l.stride + l.x, c2);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = (int)(int16_t)(c2[i >> 1] >> (16 * (i & 1)));
                    long long a = ((long long)iabs(c) * qs + add) >> qbits;
                    if (a > 32767) a = 32767;
                    const int lv = (int)(c < 0 ? -a : a);
                    lev[j][i] = lv;
                    any |= lv;
                    const long long d = (lv * scale + ((long long)1 << (bd_shift - 1))) >> bd_shift;
                    deq[j][i] = (int)(d < -32768 ? -32768 : d > 32767 ? 32767 : d);
                    if (cg && lv) { bits += rate_level((int)a); dsum += (long long)deq[j][i] * (2 * c - deq[j][i]); }
                }
            }
            store_rows(s.lvl, l, lev);
            store_pairs(l, deq);
            if (cg) { cg_d[k] = dsum; cg_b[k] = bits; }          // the TU's cbf bit waits for the group decision
            else if (any) ex.atomic_or(&s.cbf[l.plane], 1u << l.tile0);
        }
    });
    if (cg_lam_q4 > 0) ex.phase([&](int tid) {      // coefficient groups: keep or drop
        const int nbl = 1 << (2 * rg.log2n - 3), per_l = 1 << (rg.log2n - 2), per_c = per_l >> 1;
        for (int k = tid; k < nblk; k += NT) {
            const int idx = rg.block_index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n || l.intra || !cg_b[k]) continue;
            const int mate = k ^ (k < nbl ? per_l : per_c);          // the block above / below in the same 4x4 group
            const long long d = cg_d[k] + cg_d[mate];
            const int bits = cg_b[k] + cg_b[mate] + R_SB, tsh = 2 * (15 - bit_depth - l.log2n);
            if (16 * d < ((((long long)cg_lam_q4 * bits) >> 4) << tsh)) {
                const int zero[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                store_rows(s.lvl, l, zero);
                store_pairs(l, zero);
            } else ex.atomic_or(&s.cbf[l.plane], 1u << l.tile0);
        }
    });
    ex.phase([&](int tid) {      // inverse stage 1: columns, shift 7, clip to 16 bit (8.6.4.2)
        for (int k = tid; k < nblk; k += NT) {
            const int idx = rg.block_index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            if (!l.log2n || !((s.cbf[l.plane] >> l.tile0) & 1)) continue;     // a TU without levels reconstructs to zero: nothing to invert
            int acc[2][4];
            col_stage(s.mq, l, acc);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[j][i] = clip3(-32768, 32767, (acc[j][i] + 64) >> 7);
            store_rows(s.coef, l, acc);
        }
    });
    ex.phase([&](int tid) {      // inverse stage 2: rows, shift 20 - bitDepth
        for (int k = tid; k < nblk; k += NT) {
            const int idx = rg.block_index(k);
            SampleLoc l = unpack_loc(s.desc[idx], idx);
            int acc[2][4];
            if (!l.log2n || !((s.cbf[l.plane] >> l.tile0) & 1)) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[j][i] = 0;
                store_rows(s.res, l, acc);
                continue;
            }
            const int sh = 20 - bit_depth;
            row_stage(s.mq, s.coef, l, acc);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[j][i] = (int)(int16_t)((acc[j][i] + (1 << (sh - 1))) >> sh);
            store_rows(s.res, l, acc);
        }
    });
}

// the pair tables, built at compile time: every CTU program used to rebuild them from the byte matrix with a runtime division per entry
// (~400 VALU instructions per lane of every workgroup: 7 % of k_inter_ctu's instruction count, profiles/r02_a phase table)
struct PairTables { uint32_t mp[680], mq[680]; };
constexpr PairTables make_pair_tables()
{
    PairTables t{};
    const Tables m = make_tables();
    for (int i = 0; i < 680; i++) {
        const int lg = i < 8 ? 2 : i < 40 ? 3 : i < 168 ? 4 : 5, n = 1 << lg, off = lg == 2 ? 0 : lg == 3 ? 8 : lg == 4 ? 40 : 168, k = i - off, p = k / n, c = k % n, st = 5 - lg;
        t.mp[i] = (uint32_t)(uint16_t)(int16_t)m.mat[c << st][2 * p] | (uint32_t)(uint16_t)(int16_t)m.mat[c << st][2 * p + 1] << 16;
        t.mq[i] = (uint32_t)(uint16_t)(int16_t)m.mat[(2 * p) << st][c] | (uint32_t)(uint16_t)(int16_t)m.mat[(2 * p + 1) << st][c] << 16;
    }
    return t;
}
DEVCONST PairTables g_pair = make_pair_tables();

// one lane's share of the set-up, for a caller that folds it into a phase of its own (nothing else may touch the fields in that phase)
DEV void residual_init_lane(ResidualShared &s, int tid)
{
    for (int i = tid; i < 680; i += NT) { s.mp[i] = g_pair.mp[i]; s.mq[i] = g_pair.mq[i]; }
    if (tid < 16) { s.tu_log2[tid] = 0; s.tu_intra[tid] = 0; }
    if (tid < 6) { s.quant_scale[tid] = g_tab.quant_scale[tid]; s.level_scale[tid] = g_tab.level_scale[tid]; }
    if (tid < 3) s.cbf[tid] = 0;
}
template <class Ex> DEV void residual_init(Ex &ex, ResidualShared &s)
{
    ex.phase([&](int tid) { residual_init_lane(s, tid); });
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
            bits += rate_level(a);
        }
    return any ? bits + R_SB : 0;
}

}  // namespace mihevc
