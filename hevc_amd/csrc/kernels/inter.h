// hevc_amd/csrc/kernels/inter.h — K1 (integer full-search SAD + fractional SATD refinement) and the P-picture
// CTU program (quadtree decision, motion compensation, K3 residual, reconstruction).
//
// One 256-thread workgroup per 32x32 CTU; every CTU of a P picture is independent (the reference picture is the
// previous reconstructed picture), so the grid is all CTUs of all pictures in flight.  Source tile and reference
// search window are staged through LDS with row-contiguous loads; nothing else is read from HBM.
// Decisions are defined by oracle/hevc_oracle.c orc_analyze_inter_frame and reproduced exactly:
//   integer:  cost = (SAD << 4) + lambda_sad_q4 * (bits(4dx) + bits(4dy)), ties -> smaller raster position
//   tree:     bottom-up on the integer-vector SATD costs, J = (SATD << 4) + lambda * mvd bits + 4 lambda per CU, split adds 2 lambda,
//             whole wins ties
//   fraction: chosen CUs only: half-pel ring then quarter-pel ring, cost = (SATD << 4) + lambda * mvd bits, ties -> lower index
// Motion compensation is H.265 8.5.3.3.3 (luma 8-tap, chroma 4-tap, uni-prediction rounding).
#pragma once
#include "common.h"
#include "residual.h"

namespace mihevc {

template <typename T> struct InterArgs {
    Plane<const T> src[3];       // unpadded source planes (coded size)
    Plane<const T> ref[3];       // padded reference reconstruction
    Plane<T> rec[3];             // pre-deblock reconstruction out (may be padded planes; only [0,w)x[0,h) written)
    int w, h, ctus_w;
    CostParams prm;
    const int16_t *centers;      // per CTU (sx, sy) integer search centre, or nullptr
    int32_t *me;                 // per CTU 21 x (mvx, mvy, cost): written by me_search, read by inter_ctu
    mihevc_cu_rec *cu;           // (h/8) x (w/8)
    int16_t *coef[3];            // strides w, w/2, w/2
    unsigned long long *est;     // optional: picture-level rate estimate accumulator (1/16 bit), see DESIGN.md rate control
    int sparse_coef;             // 1: store levels only for TUs that have a non-zero one (the host coder never reads the others);
                                 //    lets `coef` point at pinned host memory so no 6 MB/picture D2H blit is needed
    IpInfo *ip;                  // per CTU, or nullptr: hand-over to the intra second pass (prm.intra_in_p)
    // B pictures (cfg.bframes): the anchor AFTER this picture in display order is list 1 (ref is list 0: the anchor before it); its search centres and
    // integer-search table.  ref1[0].p == nullptr: a P picture
    Plane<const T> ref1[3];
    const int16_t *centers1;
    int32_t *me1;
};
// the arguments of the integer search against list 1: the same block with the list-1 planes, centres and table in the list-0 places
template <typename T> DEV InterArgs<T> list1_view(const InterArgs<T> &a)
{
    InterArgs<T> b = a;
    for (int i = 0; i < 3; i++) b.ref[i] = a.ref1[i];
    b.centers = a.centers1; b.me = a.me1;
    return b;
}

// candidate 0 = centre, 1..8 = the ring (same order as oracle kFracOff)
DEVCONST int8_t kOff[9][2] = {{0, 0}, {-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
// which ring position the lanes of slot (lane >> 5) work on: a wave holds two slots, and the order gives wave 0 the two positions with no vertical fraction, wave 1 the two with no
// horizontal one (luma_half_diff's one-pass forms then run without divergence); positions keep their numbers, so nothing about the result changes
DEVCONST int8_t kRingSlot[8] = {4, 5, 2, 7, 1, 3, 6, 8};

// ------------------------------------------------------------------------------------------ search centres (pre-search)
// 1/4-size 8-bit pictures (rounded mean of every 4x4 luma block, reduced to 8 bits) of the source and of the reference; every
// CTU's 8x8 low-resolution block is searched over +-PRE_RANGE with clamped reads: cost = 4 * SAD + |dx| + |dy|, ties -> raster
// order (oracle: orc_pre_search).  One lane = four horizontal positions (v_qsad_pk_u16_u8), 8 quads x 29 rows = 232 lanes.
template <typename T> struct PreArgs {
    Plane<const T> src, ref;     // full-size luma planes (reference: padded plane, interior origin)
    uint8_t *lsrc, *lref;        // (w/4) x (h/4) each, row stride w/4
    int w, h, bit_depth;         // coded size
    int16_t *centers;            // out: 2 per CTU, integer luma samples
    unsigned *cost;              // out, optional: per CTU the smallest SAD of its 8x8 low-resolution block over the window (the B-picture probe sums them)
};
template <typename T> DEV void lowres_sample(const PreArgs<T> &a, int i)
{
    const int lw = a.w >> 2, n = lw * (a.h >> 2), sh = a.bit_depth - 8;
    if (i >= 2 * n) return;
    const Plane<const T> &p = i < n ? a.src : a.ref;
    const int k = i < n ? i : i - n, x = k % lw, y = k / lw;
    int s = 8 << sh;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const T *row = p.p + (ptrdiff_t)(4 * y + j) * p.stride + 4 * x;
        if (sizeof(T) == 1) {              // four samples in one dword, summed by one SAD against zero
            s = (int)sad_packed_u8(load_u32(row), 0u, (uint32_t)s);
        } else
            s += (int)row[0] + (int)row[1] + (int)row[2] + (int)row[3];
    }
    (i < n ? a.lsrc : a.lref)[k] = (uint8_t)(s >> (4 + sh));
}
constexpr int PRE_SPAN = 2 * PRE_RANGE + 1;      // 29 positions per axis
constexpr int PRE_WIN_W = 40;                    // columns a lane's 12-byte reads can touch: 8 quads x 4 + 8
constexpr int PRE_WIN_STRIDE = 96;               // bytes; 24 dwords = 8 mod 16 -> a wave's 8 quads x 8 rows hit 64 different banks
struct PreShared {
    uint8_t blk[64];
    uint8_t win[(8 + 2 * PRE_RANGE) * PRE_WIN_STRIDE + 16];
    unsigned long long best;
    unsigned sad0;               // SAD at zero displacement
};
template <typename T, class Ex> DEV void pre_search_program(Ex &ex, PreShared &s, const PreArgs<T> &a, int ctu)
{
    constexpr int R = PRE_RANGE, span = PRE_SPAN;
    const int lw = a.w >> 2, lh = a.h >> 2, wc = (lw + 7) >> 3, cx = ctu % wc, cy = ctu / wc;
    const int bw = lw - 8 * cx < 8 ? lw - 8 * cx : 8, bh = lh - 8 * cy < 8 ? lh - 8 * cy : 8;
    ex.phase([&](int tid) {
        if (tid < 64) { const int x = tid & 7, y = tid >> 3; s.blk[tid] = (x < bw && y < bh) ? a.lsrc[(8 * cy + y) * lw + 8 * cx + x] : (uint8_t)0; }
        // the window: every lane's loads are issued before its first LDS store (this loop was a 95k-cycle chain of dependent loads).  A window that
        // lies inside the low-resolution picture is fetched as dwords (two a lane); the clamped sample-by-sample form is for the CTUs at the picture's edge
        if (8 * cx - R >= 0 && 8 * cx - R + PRE_WIN_W <= lw && 8 * cy - R >= 0 && 8 * cy + 8 + R <= lh) {
            constexpr int DPR = PRE_WIN_W / 4, ND = (8 + 2 * PRE_RANGE) * DPR, IT = (ND + NT - 1) / NT;
            uint32_t v[IT];
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const int i = tid + k * NT;
                if (i < ND) v[k] = load_u32(a.lref + (size_t)(8 * cy - R + i / DPR) * lw + 8 * cx - R + 4 * (i % DPR));
            }
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const int i = tid + k * NT;
                if (i < ND) store_u32_aligned(s.win + (i / DPR) * PRE_WIN_STRIDE + 4 * (i % DPR), v[k]);
            }
        } else {
            constexpr int NW = (8 + 2 * PRE_RANGE) * PRE_WIN_W, IT = (NW + NT - 1) / NT;
            uint8_t v[IT];
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const int i = tid + k * NT;
                if (i < NW) v[k] = a.lref[clip3(0, lh - 1, 8 * cy + i / PRE_WIN_W - R) * lw + clip3(0, lw - 1, 8 * cx + i % PRE_WIN_W - R)];
            }
#pragma unroll
            for (int k = 0; k < IT; k++) {
                const int i = tid + k * NT;
                if (i < NW) s.win[(i / PRE_WIN_W) * PRE_WIN_STRIDE + i % PRE_WIN_W] = v[k];
            }
        }
        if (tid == 0) s.best = ~0ull;
    });
    ex.phase([&](int tid) {
        unsigned long long best = ~0ull;
        if (bw == 8 && bh == 8) {
            if (tid < 8 * span) {
                const int q = tid & 7, dyi = tid >> 3;
                uint64_t acc = 0;
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const uint8_t *w = s.win + (y + dyi) * PRE_WIN_STRIDE + 4 * q;
                    const uint32_t r0 = load_u32_aligned(w), r1 = load_u32_aligned(w + 4), r2 = load_u32_aligned(w + 8);
                    acc = qsad_u8(((uint64_t)r1 << 32) | r0, load_u32_aligned(s.blk + 8 * y), acc);
                    acc = qsad_u8(((uint64_t)r2 << 32) | r1, load_u32_aligned(s.blk + 8 * y + 4), acc);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int dxi = 4 * q + j;
                    if (dxi >= span) continue;
                    const unsigned sad = (unsigned)((acc >> (16 * j)) & 0xffff);
                    if (dxi == R && dyi == R) s.sad0 = sad;
                    const unsigned long long key = ((unsigned long long)(4 * sad + (unsigned)(iabs(dxi - R) + iabs(dyi - R))) << 12) | (unsigned)(dyi * span + dxi);
                    best = key < best ? key : best;
                }
            }
        } else {                 // partial block at the right / bottom picture edge: plain loops over the valid samples
            for (int p = tid; p < span * span; p += NT) {
                const int dxi = p % span, dyi = p / span;
                unsigned sad = 0;
                for (int y = 0; y < bh; y++)
                    for (int x = 0; x < bw; x++) sad += (unsigned)iabs((int)s.blk[y * 8 + x] - (int)s.win[(y + dyi) * PRE_WIN_STRIDE + x + dxi]);
                if (dxi == R && dyi == R) s.sad0 = sad;
                const unsigned long long key = ((unsigned long long)(4 * sad + (unsigned)(iabs(dxi - R) + iabs(dyi - R))) << 12) | (unsigned)p;
                best = key < best ? key : best;
            }
        }
        if (best != ~0ull) ex.atomic_min(&s.best, best);
    });
    ex.phase([&](int tid) {
        if (tid != 0) return;
        // the centre only moves when that halves the zero-displacement SAD (oracle: orc_pre_search)
        int p = (int)(s.best & 4095);
        const unsigned sad_best = (unsigned)(((s.best >> 12) - (unsigned)(iabs(p % span - R) + iabs(p / span - R))) >> 2);
        if (2 * sad_best >= s.sad0) p = R * span + R;
        a.centers[2 * ctu] = (int16_t)(4 * (p % span - R));
        a.centers[2 * ctu + 1] = (int16_t)(4 * (p / span - R));
        if (a.cost) a.cost[ctu] = sad_best;
    });
}

// ------------------------------------------------------------------------------------------ motion-constrained slices
// oracle mv_rows_ok / clamp_center_y: may a block of n luma rows at picture row y use vertical motion my (quarter samples) when the top /
// bottom edge of the picture is a slice boundary?  Its luma footprint (8-tap: rows -3 .. n + 4 around the integer part when my has a
// fraction) and chroma footprint (4-tap: -1 .. n/2 + 2 when my & 7) must stay inside.
DEV bool mv_rows_ok(int y, int n, int my, int h, int top, int bottom)
{
    const int ly0 = y + (my >> 2) - ((my & 3) ? 3 : 0), ly1 = y + n - 1 + (my >> 2) + ((my & 3) ? 4 : 0);
    const int cy0 = (y >> 1) + (my >> 3) - ((my & 7) ? 1 : 0), cy1 = (y >> 1) + (n >> 1) - 1 + (my >> 3) + ((my & 7) ? 2 : 0);
    if (top && (ly0 < 0 || cy0 < 0)) return false;
    if (bottom && (ly1 > h - 1 || cy1 > (h >> 1) - 1)) return false;
    return true;
}
DEV int clamp_center_y(int sy, int y0, int R, int h, int top, int bottom)
{
    const int ctu_h = h - y0 < CTU ? h - y0 : CTU;
    if (bottom && sy > h - (y0 + ctu_h) - R) sy = h - (y0 + ctu_h) - R;
    if (top && sy < R - y0) sy = R - y0;
    return sy;
}

// ------------------------------------------------------------------------------------------ integer search
// source tile and search window are held as the 8 most significant bits of every sample (8-bit input: the samples themselves), so
// Main10 searches with the packed quad-SAD too; SADs are scaled back by the dropped bits (oracle: sad_msb8)
template <typename T> struct MeShared {
    uint8_t src[32 * 32];
    unsigned long long best[21];
    uint8_t valid[21];
    unsigned nodeok[2 * MAX_RANGE + 1];   // bit n of word dy: node n lies inside the picture and (motion-constrained slices) vertical displacement index dy keeps its rows inside the slice
    // followed in LDS by the search window: uint8_t win[(32 + 2R) * wstride]
};
HDI int me_spanx(int R) { return ((2 * R + 1) + 3) & ~3; }   // horizontal positions, rounded up to whole quads
HDI int me_win_w(int R) { return 32 + me_spanx(R); }           // columns a quad's 8-byte windows can touch
// row stride in samples: >= width + 4, 4-sample aligned, and (in 4-sample units) = 8 mod 16 -- a wave's 8 quads x 8 rows then
// read 64 different LDS banks (17 units, the natural stride at R = 15, put rows 0 and 4 on the same banks: half of all window
// reads were bank-conflict cycles, profiles/r01_d_bench pmc_summary)
HDI int me_win_stride(int R)
{
    int u = (me_win_w(R) + 4 + 3) >> 2;
    while ((u & 15) != 8) u++;
    return u << 2;
}
HDI int me_win_elems(int R) { return (32 + 2 * R) * me_win_stride(R) + 16; }

// SADs of the four 8x8 blocks of one block row (8 CTU rows x 32 samples) at the 4 horizontal positions of a quad.
// src: CTU source rows (stride 32); ref: window at (row + dy, 4 * quad) — 4-byte aligned; out[block][position]
DEV void quad_block_row(const uint8_t *src, const uint8_t *ref, int ws, unsigned (&out)[4][4])
{
    uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll 2
    for (int r = 0; r < 8; r++) {
        uint32_t rr[9], cc[8];
#pragma unroll
        for (int d = 0; d < 9; d++) rr[d] = load_u32_aligned(ref + r * ws + 4 * d);
#pragma unroll
        for (int d = 0; d < 8; d++) cc[d] = load_u32_aligned(src + r * 32 + 4 * d);
#pragma unroll
        for (int d = 0; d < 8; d++) acc[d >> 1] = qsad_u8(((uint64_t)rr[d + 1] << 32) | rr[d], cc[d], acc[d >> 1]);
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) out[b][j] = (unsigned)((acc[b] >> (16 * j)) & 0xffff);
}
// window rows of a 16-bit plane reduced to their 8 most significant bits, four samples per LDS dword (same clamping as copy_window)
DEV void copy_window_msb(uint8_t *lds, int ls, const uint16_t *plane, int pstride, int ox, int oy, int w, int h, int lo_x, int hi_x, int lo_y, int hi_y, int sh, int tid)
{
    const int dpr = w / 4, col = tid & 31, rg = tid >> 5;
    for (int r = rg; r < h; r += NT / 32) {
        const int y = clip3(lo_y, hi_y, oy + r);
        const uint16_t *srow = plane + (ptrdiff_t)y * pstride;
        for (int d = col; d < dpr; d += 32) {
            const int x = clip3(lo_x, hi_x - 3, ox + d * 4);
            const uint32_t a = load_u32(srow + x), b = load_u32(srow + x + 2);
            const uint32_t v = ((a & 0xffffu) >> sh) | (((a >> 16) >> sh) << 8) | (((b & 0xffffu) >> sh) << 16) | (((b >> 16) >> sh) << 24);
            store_u32_aligned(lds + r * ls + d * 4, v);
        }
    }
}
DEV void copy_window_msb(uint8_t *lds, int ls, const uint8_t *plane, int pstride, int ox, int oy, int w, int h, int lo_x, int hi_x, int lo_y, int hi_y, int, int tid)
{
    copy_window<uint8_t>(lds, ls, plane, pstride, ox, oy, w, h, lo_x, hi_x, lo_y, hi_y, tid);
}

template <typename T, class Ex>
DEV void me_search_program(Ex &ex, MeShared<T> &s, uint8_t *win, const InterArgs<T> &a, int ctu)
{
    const int msb = sizeof(T) == 1 ? 0 : a.prm.bit_depth - 8;       // bits dropped from every sample for the search

    const int R = a.prm.me_range, spany = 2 * R + 1, spanx = me_spanx(R), quads = spanx >> 2, ws = me_win_stride(R), ww = me_win_w(R), wh = 32 + 2 * R;
    const int x0 = (ctu % a.ctus_w) * CTU, y0 = (ctu / a.ctus_w) * CTU;
    const int mct = a.prm.mc_top, mcb = a.prm.mc_bottom;
    const int sx = a.centers ? a.centers[2 * ctu] : 0, sy0 = a.centers ? a.centers[2 * ctu + 1] : 0;
    const int sy = (mct || mcb) ? clamp_center_y(sy0, y0, R, a.h, mct, mcb) : sy0;
    ex.phase([&](int tid) {
        if (sizeof(T) == 1 && x0 + CTU <= a.w && y0 + CTU <= a.h) {      // whole 8-bit CTU: one dword per lane
            const uint32_t v = load_u32(a.src[0].p + (size_t)(y0 + (tid >> 3)) * a.src[0].stride + x0 + 4 * (tid & 7));
            store_u32_aligned(s.src + 4 * tid, v);
        } else
        for (int i = tid; i < 1024; i += NT) {
            int x = x0 + (i & 31), y = y0 + (i >> 5);
            s.src[i] = (x < a.w && y < a.h) ? (uint8_t)(a.src[0].p[(size_t)y * a.src[0].stride + x] >> msb) : (uint8_t)0;
        }
        copy_window_msb(win, ws, a.ref[0].p, a.ref[0].stride, x0 + sx - R, y0 + sy - R, ww, wh, -PAD_Y, a.w + PAD_Y - 1, -PAD_Y, a.h + PAD_Y - 1, msb, tid);
        if (tid < 21) {
            int nx, ny, nl;
            node_geom(tid, nx, ny, nl);
            s.valid[tid] = x0 + nx + (1 << nl) <= a.w && y0 + ny + (1 << nl) <= a.h;
            s.best[tid] = ~0ull;
        }
        // per vertical displacement: the nodes that lie inside the picture and (a slice) whose rows the displacement keeps inside the slice
        for (int dyi = tid; dyi < spany; dyi += NT) {
            unsigned bits = 0;
            for (int n = 0; n < 21; n++) {
                int nx, ny, nl;
                node_geom(n, nx, ny, nl);
                const bool in_pic = x0 + nx + (1 << nl) <= a.w && y0 + ny + (1 << nl) <= a.h;
                if (in_pic && (!(mct || mcb) || mv_rows_ok(y0 + ny, 1 << nl, 4 * (sy + dyi - R), a.h, mct, mcb))) bits |= 1u << n;
            }
            s.nodeok[dyi] = bits;
        }
    });
    ex.phase([&](int tid) {
        // A lane's four candidates of a node (one quad of dx at one dy) are ranked in 32 bits, (cost << 2) | j -- at equal cost the lower j is the
        // earlier raster position, as the 64-bit (cost << 16) | position key ranks them -- and only the winner is widened and goes to the workgroup's
        // LDS minimum (filtered by a plain read first): no per-thread table of 21 running minima stays in registers across the unrolled SAD code,
        // and the 84 candidates of an item cost two VALU operations each instead of six with 64-bit compares
#ifdef MIHEVC_EXP_DPP_MIN
        for (int it0 = 0; it0 < quads * spany; it0 += NT) {             // (every lane of a wave takes part in the DPP reduction: lanes without an item redo item 0 and never win)
            const int item = it0 + tid < quads * spany ? it0 + tid : 0;
            const int q = item % quads, dyi = item / quads;
            const unsigned ok = it0 + tid < quads * spany ? s.nodeok[dyi] : 0u, pos0 = (unsigned)(dyi * spanx + 4 * q);
#else
        for (int item = tid; item < quads * spany; item += NT) {        // item = (quad of 4 dx, one dy)
            const int q = item % quads, dyi = item / quads;
            const unsigned ok = s.nodeok[dyi], pos0 = (unsigned)(dyi * spanx + 4 * q);
#endif
            const int by = mvd_bits(4 * (dyi - R)), ksh = 6 + msb;
            const uint8_t *srcp = s.src + opaque_zero();      // keep the 1 KiB source tile in LDS (hoisted into 256 VGPRs otherwise)
            unsigned bitsj[4], s32[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; j++) bitsj[j] = ((unsigned)(a.prm.lambda_sad_q4 * (mvd_bits(4 * (4 * q + j - R)) + by)) << 2) | (unsigned)j;
            auto consider = [&](int node, const unsigned (&sad)[4]) {
                unsigned k = (sad[0] << ksh) + bitsj[0];
#pragma unroll
                for (int j = 1; j < 4; j++) { const unsigned kj = (sad[j] << ksh) + bitsj[j]; k = kj < k ? kj : k; }
#ifdef MIHEVC_EXP_DPP_MIN
                // A/B of round 3: the wave's minimum by DPP (row_shr 1, 2, 4, 8, row_bcast 15 / 31: lane 63 ends with it), ONE LDS atomic per wave
                // and node instead of the filtered per-lane atomic below (build device.hip with -DMIHEVC_EXP_DPP_MIN; parity-green).  Measured on one box, two runs each
                // (profiles/r03/README.md): integer search 0.0386 -> 0.0400 ms per 1080p picture (+3.5 %): the 12 DPP moves + 6 64-bit compare / selects per node
                // are 21 x ~40 VALU operations a lane against ~4 for "read, compare, rarely an atomic", in a kernel that is VALU-bound.  The LDS form stays.
                unsigned long long key = ((ok >> node) & 1) ? (((unsigned long long)(k >> 2) << 16) | (pos0 + (k & 3))) : ~0ull;
                auto step = [&](auto ctrl, auto row_mask) {
                    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
                    const unsigned tl = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
                    const unsigned th = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
                    const unsigned long long t = ((unsigned long long)th << 32) | tl;
                    key = t < key ? t : key;
                };
                step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
                step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
                step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
                step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
                step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
                step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
                if ((tid & 63) == 63 && key < ex.peek(&s.best[node])) ex.atomic_min(&s.best[node], key);
#else
                if (!((ok >> node) & 1)) return;
                const unsigned long long key = ((unsigned long long)(k >> 2) << 16) | (pos0 + (k & 3));
                if (key < ex.peek(&s.best[node])) ex.atomic_min(&s.best[node], key);
#endif
            };
#pragma unroll
            for (int half = 0; half < 2; half++) {
                unsigned s16[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                for (int r2 = 0; r2 < 2; r2++) {
                    const int br = half * 2 + r2;
                    unsigned o[4][4];
                    quad_block_row(srcp + br * 8 * 32, win + (br * 8 + dyi) * ws + 4 * q, ws, o);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
#pragma unroll
                        for (int j = 0; j < 4; j++) s16[b >> 1][j] += o[b][j];
                        consider(5 + (half * 2 + (b >> 1)) * 4 + r2 * 2 + (b & 1), o[b]);
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) s32[j] += s16[h2][j];
                    consider(1 + half * 2 + h2, s16[h2]);
                }
            }
            consider(0, s32);
        }
    });
    ex.phase([&](int tid) {
        if (tid < 21) {
            int32_t *o = a.me + ((size_t)ctu * 21 + tid) * 3;
            if (s.valid[tid]) {
                int p = (int)(s.best[tid] & 0xffff);
                o[0] = 4 * (sx + p % spanx - R); o[1] = 4 * (sy + p / spanx - R); o[2] = (int32_t)(s.best[tid] >> 16);
            } else { o[0] = 0; o[1] = 0; o[2] = -1; }
        }
    });
}

// ------------------------------------------------------------------------------------------ P-picture CTU program
template <typename T> struct InterShared {
    ResidualShared rs;
    T src[1536];                 // Y 32x32, U 16x16, V 16x16
    T pred[1536];
    int mvx[21], mvy[21];
    unsigned cost[21];           // current best cost of each node (SATD << 4 + lambda * mvd bits)
    uint8_t valid[21];
    int satd[3][1][16];          // [level][0][tile]: tile SATDs at the integer vectors
    unsigned nsum[21];           // per node: sum of its tiles' integer-vector SATDs
    unsigned fsum[8][21];        // per ring candidate and chosen node: sum of its tiles' SATDs (LDS atomics from the tile lanes)
    unsigned long long rbest[21];     // per chosen node: min over the ring of (cost << 4 | position), position 0 = the centre
    unsigned j16[4];
    int use16[4], use32;
    uint8_t alias[3][16];        // level whose SATDs stand for (level, tile): a coarser node with the SAME vector as a finer one is not recomputed
    int tile_mvx[16], tile_mvy[16];
    uint8_t tile_node[16];
    uint8_t chosen[21];          // node is a CU of the decided quadtree
    unsigned est;                // CTU rate estimate, 1/16 bit
    unsigned ip_cost, ip_act, ip_tiles;   // intra second pass: chosen CUs' cost, source AC activity, tiles inside the picture
    unsigned long long ip_sse;
    unsigned tu_dc[3][16], tu_dz[3][16], tu_bits[3][16], tu_zero[3];   // RD zero-out: per TU (plane, first 8x8 tile) SSE coded / zeroed, level bits
    // followed in LDS by: T winY[(40 + 2R)^2 (stride padded)], T winU[(24 + R)^2], T winV[...]
};
// motion-compensation windows cover every vector the search can return: |mv| <= R + 3 (widened horizontal span)
HDI int mc_win_y(int R) { return (32 + 2 * (R + 3) + 8 + 3) & ~3; }
HDI int mc_win_y_stride(int R) { return mc_win_y(R) + 4; }
HDI int mc_win_c(int R) { return (16 + (R + 3) + 8 + 3) & ~3; }
HDI int mc_win_c_stride(int R) { return mc_win_c(R) + 4; }

// quarter-sample luma prediction of one 8x8 tile from the LDS window (8.5.3.3.3.1; the general 2-D form with the
// {0,0,0,64,0,0,0,0} tap set for a zero fraction is exact for every case).  p00 = window sample at the tile's
// integer position (element index i00 into the 4-byte aligned window image `win`).  When diff_src != nullptr returns the 8x8 Hadamard SATD of (src - pred), else writes pred.
template <typename T>
DEV int luma_tile(const T *win, int i00, int ws, int fx, int fy, int bit_depth, const T *diff_src, int src_stride, T *pred_out, int pred_stride)
{
    const int8_t *tx = g_tab.luma_tap[fx], *ty = g_tab.luma_tap[fy];
    const int shift1 = bit_depth - 8, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;
    int acc[8][8];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[j][i] = 0;
    for (int r = 0; r < 15; r++) {                 // intermediate row r corresponds to reference row r - 3
        int px[15];
        load_row15(win, i00 + (r - 3) * ws - 3, px);
        int hv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) v += tx[k] * px[i + k];
            hv[i] = v >> shift1;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int k = r - j;
            if (k >= 0 && k < 8) {
                int t = ty[k];
#pragma unroll
                for (int i = 0; i < 8; i++) acc[j][i] += t * hv[i];
            }
        }
    }
    const int off = 1 << (shift3 - 1);
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int v = clip3(0, maxv, ((acc[j][i] >> 6) + off) >> shift3);
            if (diff_src) acc[j][i] = (int)diff_src[j * src_stride + i] - v;
            else pred_out[j * pred_stride + i] = (T)v;
        }
    return diff_src ? hadamard8_satd(acc) : 0;
}

// 8-bit specialisation of luma_tile: the horizontal 8-tap filter of one row is 16 v_dot4_i32_i8 on byte-realigned
// dwords.  Samples are biased to signed bytes (p - 128); the taps of every fraction sum to 64, so the bias comes back
// as the constant 128 * 64.  Results are identical to the generic form (the fractional search was 72 % of
// k_inter_ctu, VALU bound on 8 multiply-adds per output: profiles/r01 ablation).
DEV int luma_tile(const uint8_t *win, int i00, int ws, int fx, int fy, int bit_depth, const uint8_t *diff_src, int src_stride, uint8_t *pred_out,
                  int pred_stride)
{
    const uint32_t tlo = load_u32(&g_tab.luma_tap[fx][0]), thi = load_u32(&g_tab.luma_tap[fx][4]);
    // vertical taps as four (even, odd) int16 pairs: the 8-tap column filter is 4 v_dot2_i32_i16 on row PAIRS of the
    // horizontally filtered samples (|hv| < 2^15 for every fraction) instead of 8 quarter-rate 32-bit multiplies
    uint32_t typ[4];
#pragma unroll
    for (int m = 0; m < 4; m++) typ[m] = pack_lo16(g_tab.luma_tap[fy][2 * m], g_tab.luma_tap[fy][2 * m + 1]);
    int acc[8][8], prev[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[j][i] = 0;
#pragma unroll
    for (int r = 0; r < 15; r++) {
        const int idx = i00 + (r - 3) * ws - 3, off = idx & 3;
        const uint8_t *p = win + (idx - off);
        uint32_t d[5], a[12];
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = load_u32_aligned(p + 4 * k);
        uint32_t q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = align_bytes(d[k + 1], d[k], off) ^ 0x80808080u;      // bytes 4k..4k+3 of the row, signed
#pragma unroll
        for (int k = 0; k < 3; k++) {
            a[4 * k] = q[k];
            a[4 * k + 1] = align_bytes(q[k + 1], q[k], 1);
            a[4 * k + 2] = align_bytes(q[k + 1], q[k], 2);
            a[4 * k + 3] = align_bytes(q[k + 1], q[k], 3);
        }
        int hv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) hv[i] = dot4_i8(a[i], tlo, dot4_i8(a[i + 4], thi, 128 * 64));
        if (r > 0) {
            // rows (r-1, r) feed output row j with taps (2m, 2m+1) where r - 1 - j = 2m
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t pr = pack_lo16(prev[i], hv[i]);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int j = r - 1 - 2 * m;
                    if (j >= 0 && j < 8) acc[j][i] = dot2_i16(pr, typ[m], acc[j][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) prev[i] = hv[i];
    }
    (void)bit_depth;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int v = clip3(0, 255, ((acc[j][i] >> 6) + 32) >> 6);
            if (diff_src) {
                // source rows as two aligned dwords (the tile is 8-aligned in the CTU image): 64 single-byte LDS loads issued up
                // front cost 64 VGPRs and pushed the kernel into scratch spills
                const uint32_t w = load_u32_aligned(diff_src + j * src_stride + (i & 4));
                acc[j][i] = (int)((w >> (8 * (i & 3))) & 255) - v;
            } else pred_out[j * pred_stride + i] = (uint8_t)v;
        }
    return diff_src ? hadamard8_satd(acc) : 0;
}

// 16-bit specialisation (Main10): samples are already int16 pairs in the window's dwords, so the horizontal 8-tap filter is
// 4 v_dot2_i32_i16 per output (odd outputs on dwords realigned by one sample), the column filter 4 v_dot2 on row pairs as in
// the 8-bit form.  Intermediates stay below 2^15 (8.5.3.3.3: 14-bit intermediate precision), results equal the generic form.
DEV int luma_tile(const uint16_t *win, int i00, int ws, int fx, int fy, int bit_depth, const uint16_t *diff_src, int src_stride, uint16_t *pred_out,
                  int pred_stride)
{
    uint32_t txp[4], typ[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        txp[m] = pack_lo16(g_tab.luma_tap[fx][2 * m], g_tab.luma_tap[fx][2 * m + 1]);
        typ[m] = pack_lo16(g_tab.luma_tap[fy][2 * m], g_tab.luma_tap[fy][2 * m + 1]);
    }
    const int shift1 = bit_depth - 8, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;
    int acc[8][8], prev[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[j][i] = 0;
#pragma unroll
    for (int r = 0; r < 15; r++) {
        const int idx = i00 + (r - 3) * ws - 3, off = idx & 1;
        const uint16_t *p = win + (idx - off);
        uint32_t d[9], e[8], o[7];
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = load_u32_aligned(p + 2 * k);
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = align_bytes(d[k + 1], d[k], 2 * off);     // samples (2k, 2k+1) of the row
#pragma unroll
        for (int k = 0; k < 7; k++) o[k] = align_bytes(e[k + 1], e[k], 2);           // samples (2k+1, 2k+2)
        int hv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int v = 0;
#pragma unroll
            for (int m = 0; m < 4; m++) v = dot2_i16((i & 1) ? o[(i >> 1) + m] : e[(i >> 1) + m], txp[m], v);
            hv[i] = v >> shift1;
        }
        if (r > 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t pr = pack_lo16(prev[i], hv[i]);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int j = r - 1 - 2 * m;
                    if (j >= 0 && j < 8) acc[j][i] = dot2_i16(pr, typ[m], acc[j][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) prev[i] = hv[i];
    }
    const int off3 = 1 << (shift3 - 1);
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int v = clip3(0, maxv, ((acc[j][i] >> 6) + off3) >> shift3);
            if (diff_src) acc[j][i] = (int)diff_src[j * src_stride + i] - v;
            else pred_out[j * pred_stride + i] = (uint16_t)v;
        }
    return diff_src ? hadamard8_satd(acc) : 0;
}

// ---- fractional search, two lanes per (tile, candidate): each lane owns 4 of the tile's 8 columns.  The whole-tile form (luma_tile with a
// source) kept 64 accumulators + filter state live and spilled at 4 workgroups per CU, and left half of the workgroup idle.
// Difference (source - quarter-sample prediction, 8.5.3.3.3.1) of the 8 rows x 4 columns whose first integer sample is window element i00.
DEV void luma_half_diff(const uint8_t *win, int i00, int ws, int fx, int fy, int, const uint8_t *src, int src_stride, int (&m)[8][4])
{
    const uint32_t tlo = load_u32(&g_tab.luma_tap[fx][0]), thi = load_u32(&g_tab.luma_tap[fx][4]);
    // A zero fraction makes that direction's 8-tap filter the identity tap {0, 0, 0, 64, 0, 0, 0, 0}, and the general form below then multiplies by 64 and shifts by 6 again:
    // the same values come out of one filter pass.  Worth a branch only when the whole wave takes it (inter_ctu_program orders the ring so that a wave's two candidates
    // share their zero fraction: half of the first ring's waves, and the second ring's wherever the vectors stayed on whole samples).
    if (wave_all(fy == 0)) {          // horizontal filter only: rows 0 .. 7 of the block, no vertical pass
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = i00 + j * ws - 3, off = idx & 3;
            const uint8_t *p = win + (idx - off);
            uint32_t d[4], q[3];
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = load_u32_aligned(p + 4 * k);
#pragma unroll
            for (int k = 0; k < 3; k++) q[k] = align_bytes(d[k + 1], d[k], off) ^ 0x80808080u;
            int hv[4];
            hv[0] = dot4_i8(q[0], tlo, dot4_i8(q[1], thi, 128 * 64));
#pragma unroll
            for (int i = 1; i < 4; i++) hv[i] = dot4_i8(align_bytes(q[1], q[0], i), tlo, dot4_i8(align_bytes(q[2], q[1], i), thi, 128 * 64));
            const uint32_t w = load_u32_aligned(src + j * src_stride);
#pragma unroll
            for (int i = 0; i < 4; i++) m[j][i] = (int)((w >> (8 * i)) & 255) - clip3(0, 255, (hv[i] + 32) >> 6);      // ((64 hv >> 6) + 32) >> 6
        }
        return;
    }
    if (wave_all(fx == 0)) {          // vertical filter only: the block's own 4 columns of rows -3 .. 11, no horizontal pass
        const uint32_t vlo = load_u32(&g_tab.luma_tap[fy][0]), vhi = load_u32(&g_tab.luma_tap[fy][4]);
        const int idx = i00 - 3 * ws, off = idx & 3;
        uint32_t rows[15];           // four samples of every row, as signed bytes
#pragma unroll
        for (int r = 0; r < 15; r++) {
            const uint8_t *p = win + (idx - off) + r * ws;
            rows[r] = align_bytes(load_u32_aligned(p + 4), load_u32_aligned(p), off) ^ 0x80808080u;
        }
        uint32_t col[4][4];          // [rows 4g .. 4g + 3][column]: the rows' bytes transposed (row 15 does not exist: zero)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t r4[4] = {rows[4 * g], rows[4 * g + 1], rows[4 * g + 2], g < 3 ? rows[4 * g + 3] : 0u};
            transpose_bytes4(r4, col[g]);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int g = j >> 2, sh = j & 3;
            const uint32_t w = load_u32_aligned(src + j * src_stride);
#pragma unroll
            for (int i = 0; i < 4; i++) {       // output row j of column i: rows j .. j + 7 against the taps
                const uint32_t lo = align_bytes(col[g + 1][i], col[g][i], sh), hi = align_bytes(col[g + 2][i], col[g + 1][i], sh);
                const int v = dot4_i8(lo, vlo, dot4_i8(hi, vhi, 128 * 64));      // sum of tap x sample: what (64 x sample, filtered) >> 6 is
                m[j][i] = (int)((w >> (8 * i)) & 255) - clip3(0, 255, (v + 32) >> 6);
            }
        }
        return;
    }
    uint32_t typ[4];
#pragma unroll
    for (int k = 0; k < 4; k++) typ[k] = pack_lo16(g_tab.luma_tap[fy][2 * k], g_tab.luma_tap[fy][2 * k + 1]);
    int acc[8][4], prev[4];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[j][i] = 0;
#pragma unroll
    for (int r = 0; r < 15; r++) {
        const int idx = i00 + (r - 3) * ws - 3, off = idx & 3;
        const uint8_t *p = win + (idx - off);
        uint32_t d[4], q[3];
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = load_u32_aligned(p + 4 * k);
#pragma unroll
        for (int k = 0; k < 3; k++) q[k] = align_bytes(d[k + 1], d[k], off) ^ 0x80808080u;      // bytes 4k..4k+3 of the row, signed
        int hv[4];
        hv[0] = dot4_i8(q[0], tlo, dot4_i8(q[1], thi, 128 * 64));
#pragma unroll
        for (int i = 1; i < 4; i++) hv[i] = dot4_i8(align_bytes(q[1], q[0], i), tlo, dot4_i8(align_bytes(q[2], q[1], i), thi, 128 * 64));
        if (r > 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t pr = pack_lo16(prev[i], hv[i]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = r - 1 - 2 * k;
                    if (j >= 0 && j < 8) acc[j][i] = dot2_i16(pr, typ[k], acc[j][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) prev[i] = hv[i];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = load_u32_aligned(src + j * src_stride);
#pragma unroll
        for (int i = 0; i < 4; i++) m[j][i] = (int)((w >> (8 * i)) & 255) - clip3(0, 255, ((acc[j][i] >> 6) + 32) >> 6);
    }
}
DEV void luma_half_diff(const uint16_t *win, int i00, int ws, int fx, int fy, int bit_depth, const uint16_t *src, int src_stride, int (&m)[8][4])
{
    uint32_t txp[4], typ[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        txp[k] = pack_lo16(g_tab.luma_tap[fx][2 * k], g_tab.luma_tap[fx][2 * k + 1]);
        typ[k] = pack_lo16(g_tab.luma_tap[fy][2 * k], g_tab.luma_tap[fy][2 * k + 1]);
    }
    const int shift1 = bit_depth - 8, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1, off3 = 1 << (shift3 - 1);
    // the horizontal 8-tap filter of intermediate row r (reference row r - 3): four 14-bit values
    auto row_h = [&](int r, int (&hv)[4]) {
        const int idx = i00 + (r - 3) * ws - 3, off = idx & 1;
        const uint16_t *p = win + (idx - off);
        uint32_t d[7], e[6], o[5];
#pragma unroll
        for (int k = 0; k < 7; k++) d[k] = load_u32_aligned(p + 2 * k);
#pragma unroll
        for (int k = 0; k < 6; k++) e[k] = align_bytes(d[k + 1], d[k], 2 * off);     // samples (2k, 2k+1) of the row
#pragma unroll
        for (int k = 0; k < 5; k++) o[k] = align_bytes(e[k + 1], e[k], 2);           // samples (2k+1, 2k+2)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v = dot2_i16((i & 1) ? o[(i >> 1) + k] : e[(i >> 1) + k], txp[k], v);
            hv[i] = v >> shift1;
        }
    };
    // a zero fraction makes that direction's filter the identity (see the 8-bit form): one pass gives the same values; taken when the whole wave has it
    if (wave_all(fy == 0)) {          // horizontal only: the block's own 8 rows; 64 hv >> 6 = hv
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int hv[4];
            row_h(j + 3, hv);
#pragma unroll
            for (int i = 0; i < 4; i++) m[j][i] = (int)src[j * src_stride + i] - clip3(0, maxv, (hv[i] + off3) >> shift3);
        }
        return;
    }
    const bool v_only = wave_all(fx == 0);      // vertical only: the horizontal pass of a whole-sample column is the sample x 64 >> shift1
    int acc[8][4], prev[4];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[j][i] = 0;
#pragma unroll
    for (int r = 0; r < 15; r++) {
        int hv[4];
        if (v_only) {
#pragma unroll
            for (int i = 0; i < 4; i++) hv[i] = (int)win[i00 + (r - 3) * ws + i] << (6 - shift1);
        } else row_h(r, hv);
        if (r > 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t pr = pack_lo16(prev[i], hv[i]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = r - 1 - 2 * k;
                    if (j >= 0 && j < 8) acc[j][i] = dot2_i16(pr, typ[k], acc[j][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) prev[i] = hv[i];
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) m[j][i] = (int)src[j * src_stride + i] - clip3(0, maxv, ((acc[j][i] >> 6) + off3) >> shift3);
}
// the half tile's share of the 8x8 Hadamard transform: all three vertical stages and the two horizontal stages inside its 4 columns;
// the last horizontal stage pairs column c of the two halves (inter_ctu_program)
DEV void hadamard_half(int (&m)[8][4])
{
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[i][x], q = m[i + st][x]; m[i][x] = p + q; m[i + st][x] = p - q; }
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const int p0 = m[y][0] + m[y][1], p1 = m[y][0] - m[y][1], p2 = m[y][2] + m[y][3], p3 = m[y][2] - m[y][3];
        m[y][0] = p0 + p2; m[y][1] = p1 + p3; m[y][2] = p0 - p2; m[y][3] = p1 - p3;
    }
}

// SATD of one 8x8 tile against the window at an INTEGER vector (both fractions zero: the prediction is the window itself)
template <typename T> DEV int luma_tile_int(const T *win, int i00, int ws, const T *src, int src_stride)
{
    int m[8][8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        int px[8];
        load_row8(win, i00 + j * ws, px);
#pragma unroll
        for (int i = 0; i < 8; i++) m[j][i] = (int)src[j * src_stride + i] - px[i];
    }
    return hadamard8_satd(m);
}

// four horizontally adjacent predicted luma samples (8.5.3.3.3.1) starting at window element i00: the final motion
// compensation of a CTU spread over all 256 lanes (16 tiles x 8 rows x 2 halves)
template <typename T>
DEV void luma_quad(const T *win, int i00, int ws, int fx, int fy, int bit_depth, T *out)
{
    const int8_t *tx = g_tab.luma_tap[fx], *ty = g_tab.luma_tap[fy];
    const int shift1 = bit_depth - 8, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int px[15];
        load_row15(win, i00 + (r - 3) * ws - 3, px);
        const int t = ty[r];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) v += tx[k] * px[i + k];
            acc[i] += t * (v >> shift1);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = (T)clip3(0, maxv, ((acc[i] >> 6) + (1 << (shift3 - 1))) >> shift3);
}

// 8-bit form of luma_quad: the horizontal 8-tap filter of a row's four outputs is 8 v_dot4_i32_i8 on byte-realigned dwords (samples biased to
// signed bytes, the taps of every fraction sum to 64), as in luma_half_diff; same results as the generic form
DEV void luma_quad(const uint8_t *win, int i00, int ws, int fx, int fy, int, uint8_t *out)
{
    const uint32_t tlo = load_u32(&g_tab.luma_tap[fx][0]), thi = load_u32(&g_tab.luma_tap[fx][4]);
    const int8_t *ty = g_tab.luma_tap[fy];
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int idx = i00 + (r - 3) * ws - 3, off = idx & 3;
        const uint8_t *p = win + (idx - off);
        uint32_t d[4], q[3];
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = load_u32_aligned(p + 4 * k);
#pragma unroll
        for (int k = 0; k < 3; k++) q[k] = align_bytes(d[k + 1], d[k], off) ^ 0x80808080u;
        const int t = ty[r];
        acc[0] += t * dot4_i8(q[0], tlo, dot4_i8(q[1], thi, 128 * 64));
#pragma unroll
        for (int i = 1; i < 4; i++) acc[i] += t * dot4_i8(align_bytes(q[1], q[0], i), tlo, dot4_i8(align_bytes(q[2], q[1], i), thi, 128 * 64));
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = (uint8_t)clip3(0, 255, ((acc[i] >> 6) + 32) >> 6);
}

template <typename T>
DEV int chroma_sample(const T *p00, int ws, int fx, int fy, int bit_depth)
{
    const int8_t *tx = g_tab.chroma_tap[fx], *ty = g_tab.chroma_tap[fy];
    const int shift1 = bit_depth - 8, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;
    int acc = 0;
    for (int r = 0; r < 4; r++) {
        const T *row = p00 + (r - 1) * ws - 1;
        int v = tx[0] * row[0] + tx[1] * row[1] + tx[2] * row[2] + tx[3] * row[3];
        acc += ty[r] * (v >> shift1);
    }
    return clip3(0, maxv, ((acc >> 6) + (1 << (shift3 - 1))) >> shift3);
}

// ---- B pictures: the 14-bit intermediate predictions of 8.5.3.3.3 (before the rounding of 8.5.3.3.4.2), four luma samples / one chroma sample
template <typename T> DEV void luma_quad14(const T *win, int i00, int ws, int fx, int fy, int bit_depth, int (&out)[4])
{
    const int8_t *tx = g_tab.luma_tap[fx], *ty = g_tab.luma_tap[fy];
    const int shift1 = bit_depth - 8;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int px[15];
        load_row15(win, i00 + (r - 3) * ws - 3, px);
        const int t = ty[r];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) v += tx[k] * px[i + k];
            acc[i] += t * (v >> shift1);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = acc[i] >> 6;
}
template <typename T> DEV int chroma_sample14(const T *p00, int ws, int fx, int fy, int bit_depth)
{
    const int8_t *tx = g_tab.chroma_tap[fx], *ty = g_tab.chroma_tap[fy];
    const int shift1 = bit_depth - 8;
    int acc = 0;
    for (int r = 0; r < 4; r++) {
        const T *row = p00 + (r - 1) * ws - 1;
        int v = tx[0] * row[0] + tx[1] * row[1] + tx[2] * row[2] + tx[3] * row[3];
        acc += ty[r] * (v >> shift1);
    }
    return acc >> 6;
}
// 8.5.3.3.4.2 default weighted sample prediction: one list, or the average of both
DEV int weighted_uni(int p, int bit_depth) { const int sh = 14 - bit_depth; return clip3(0, (1 << bit_depth) - 1, (p + (1 << (sh - 1))) >> sh); }
DEV int weighted_bi(int p0, int p1, int bit_depth) { const int sh = 15 - bit_depth; return clip3(0, (1 << bit_depth) - 1, (p0 + p1 + (1 << (sh - 1))) >> sh); }
// per-CTU state of the list-1 / bi-prediction part of a B picture's CTU program (behind the windows in LDS; P pictures do not carry it)
struct BiShared {
    int mx0[21], my0[21];        // refined list-0 vectors (InterShared::mvx / mvy go on to hold list 1)
    unsigned c0[21], cb[21];     // list-0 cost, bi-prediction SATD sum
    uint8_t mode[21];            // 0: list 0, 1: list 1, 2: both
    uint8_t tile_mode[16];
    int tile_mv1x[16], tile_mv1y[16];
    alignas(16) int16_t p0[1536];    // the 14-bit list-0 prediction of the CTU (Y, U, V as InterShared::pred), kept while the windows hold list 1
};

template <typename T, class Ex, bool BI = false>
DEV void inter_ctu_program(Ex &ex, InterShared<T> &s, T *win_y, T *win_u, T *win_v, const InterArgs<T> &a, int ctu, BiShared *bsh = nullptr)
{
    const int R0 = a.prm.me_range, R = R0 + 3, bd = a.prm.bit_depth, lam = a.prm.lambda_sad_q4;
    const int x0 = (ctu % a.ctus_w) * CTU, y0 = (ctu / a.ctus_w) * CTU;
    const int mct = a.prm.mc_top, mcb = a.prm.mc_bottom;
    const int sx = a.centers ? a.centers[2 * ctu] : 0, sy0 = a.centers ? a.centers[2 * ctu + 1] : 0;
    const int sy = (mct || mcb) ? clamp_center_y(sy0, y0, R0, a.h, mct, mcb) : sy0;
    const int wy = mc_win_y(R0), wys = mc_win_y_stride(R0), wc = mc_win_c(R0), wcs = mc_win_c_stride(R0);
    const int oy_x = x0 + sx - R - 4, oy_y = y0 + sy - R - 4;                                  // luma window origin
    const int oc_x = (x0 >> 1) + ((4 * sx - 4 * R - 3) >> 3) - 1, oc_y = (y0 >> 1) + ((4 * sy - 4 * R - 3) >> 3) - 1;

    ex.phase([&](int tid) {
        residual_init_lane(s.rs, tid);          // (transform tables and flags: nothing else in this phase touches them; a phase of their own was one more barrier)
        load_ctu_source<T>(s.src, a.src, x0, y0, a.w, a.h, tid);
        copy_window<T>(win_y, wys, a.ref[0].p, a.ref[0].stride, oy_x, oy_y, wy, wy, -PAD_Y, a.w + PAD_Y - 1, -PAD_Y, a.h + PAD_Y - 1, tid);
        copy_window<T>(win_u, wcs, a.ref[1].p, a.ref[1].stride, oc_x, oc_y, wc, wc, -PAD_C, (a.w >> 1) + PAD_C - 1, -PAD_C, (a.h >> 1) + PAD_C - 1, tid);
        copy_window<T>(win_v, wcs, a.ref[2].p, a.ref[2].stride, oc_x, oc_y, wc, wc, -PAD_C, (a.w >> 1) + PAD_C - 1, -PAD_C, (a.h >> 1) + PAD_C - 1, tid);
        if (tid < 21) {
            const int32_t *m = a.me + ((size_t)ctu * 21 + tid) * 3;
            s.mvx[tid] = m[0]; s.mvy[tid] = m[1]; s.valid[tid] = m[2] >= 0; s.cost[tid] = 0; s.nsum[tid] = 0;
        }
    });
    // SATD of every node at its INTEGER vector: the quadtree is decided on these (+ lambda * mvd bits), the fractional search then
    // runs for the chosen CUs only — 16 tiles x 8 ring positions = 128 lanes = two full waves per round, instead of one
    // pass per tree level (the search was 43 % of this kernel: profiles/r01)
    ex.phase([&](int tid) {
        for (int u = tid; u < 3 * 16; u += NT) {
            int level = u >> 4, t = u & 15;
            int txp = t & 3, typ = t >> 2, node = node_of_tile(level, txp, typ);
            if (!s.valid[node]) continue;
            // identical vectors give identical tile SATDs: let the finest level that shares the vector do the work
            int al = level;
            for (int fl = 2; fl > level; fl--) {
                int fn = node_of_tile(fl, txp, typ);
                if (s.valid[fn] && s.mvx[fn] == s.mvx[node] && s.mvy[fn] == s.mvy[node]) { al = fl; break; }
            }
            s.alias[level][t] = (uint8_t)al;
            if (al != level) continue;
            int px = x0 + txp * 8 + (s.mvx[node] >> 2) - oy_x, py = y0 + typ * 8 + (s.mvy[node] >> 2) - oy_y;
            s.satd[level][0][t] = luma_tile_int<T>(win_y, py * wys + px, wys, s.src + typ * 8 * 32 + txp * 8, 32);
        }
    });
    // every (level, tile) lane adds its tile's SATD (its own or the finer level's it aliases) to the node's sum
    ex.phase([&](int tid) {
        if (tid < 3 * 16) {
            const int level = tid >> 4, t = tid & 15, node = node_of_tile(level, t & 3, t >> 2);
            if (s.valid[node]) ex.atomic_add(&s.nsum[node], (unsigned)s.satd[s.alias[level][t]][0][t]);
        }
        if (tid >= 64 && tid < 64 + 8 * 21) s.fsum[(tid - 64) / 21][(tid - 64) % 21] = 0;
    });
    // quadtree decision, bottom up inside wave 0 (wave-local steps: 21 lanes cost the nodes, 4 lanes settle the 16x16 level, one lane
    // the 32x32 level, 16 lanes label their tiles); one thread walking the whole tree was 9k cycles of every CTU program
    ex.wave_step([&](int tid) {
        if (tid >= 21) return;
        s.chosen[tid] = 0;
        s.cost[tid] = s.valid[tid] ? (s.nsum[tid] << 4) + (unsigned)(lam * (mvd_bits(s.mvx[tid] - 4 * sx) + mvd_bits(s.mvy[tid] - 4 * sy))) : 0;
        if (tid == 0) { s.est = 0; s.ip_cost = 0; s.ip_act = 0; s.ip_tiles = 0; s.ip_sse = 0; }
    });
    ex.wave_step([&](int tid) {
        if (tid >= 4) return;
        const int q = tid;
        unsigned js = (unsigned)(lam * 2);
        for (int t = 0; t < 4; t++) if (s.valid[5 + 4 * q + t]) js += s.cost[5 + 4 * q + t] + (unsigned)(lam * 4);
        const unsigned jw = s.cost[1 + q] + (unsigned)(lam * 4);
        s.use16[q] = s.valid[1 + q] && jw <= js;
        s.j16[q] = s.use16[q] ? jw : js;
    });
    ex.wave_step([&](int tid) {
        if (tid != 0) return;
        const unsigned js32 = (unsigned)(lam * 2) + s.j16[0] + s.j16[1] + s.j16[2] + s.j16[3];
        s.use32 = s.valid[0] && s.cost[0] + (unsigned)(lam * 4) <= js32;
    });
    ex.phase([&](int tid) {
        if (tid >= 16) return;
        const int t = tid, txp = t & 3, typ = t >> 2, q = (typ >> 1) * 2 + (txp >> 1), n8 = node_of_tile(2, txp, typ);
        const int node = s.use32 ? 0 : s.use16[q] ? 1 + q : n8, inside = s.valid[n8];      // inside: the 8x8 tile itself lies inside the picture
        s.tile_node[t] = (uint8_t)node;
        if (inside) s.chosen[node] = 1;                   // several tiles may store the same 1
        s.rs.tu_log2[t] = inside ? (uint8_t)(node == 0 ? 5 : node < 5 ? 4 : 3) : 0;
        s.rs.tu_intra[t] = 0;
    });
    // fractional refinement of the chosen CUs: half-pel ring, then quarter-pel ring (the centre's cost is known).  csx / csy: the search centre the
    // vectors are priced against, wox / woy: origin of the luma window in LDS (a B picture refines list 1 with the same code after list 0)
    auto refine = [&](const int csx, const int csy, const int wox, const int woy) {
    for (int round = 0; round < 2; round++) {
        const int step = round == 0 ? 2 : 1;
        // 16 tiles x 8 ring positions x 2 column halves = the whole workgroup.  Wave-local steps: both lanes of a pair sit in one wave.
        ex.wave_step([&](int tid) {
            const int u = tid >> 1, half = tid & 1, k = kRingSlot[u >> 4], t = u & 15;
            if (!s.rs.tu_log2[t]) return;
            const int txp = t & 3, typ = t >> 2, node = s.tile_node[t];
            const int mx = s.mvx[node] + kOff[k][0] * step, my = s.mvy[node] + kOff[k][1] * step;
            const int px = x0 + txp * 8 + 4 * half + (mx >> 2) - wox, py = y0 + typ * 8 + (my >> 2) - woy;
            int m[8][4];
            luma_half_diff((const T *)win_y, py * wys + px, wys, mx & 3, my & 3, bd, (const T *)(s.src + typ * 8 * 32 + txp * 8 + 4 * half), 32, m);
            {
                // Differences are 9 bits at 8 bit, 11 at 10 bit; the five butterfly stages computed here grow them by 5: 255 x 32 and 1023 x 32 = 32736 both fit 16 bits, two values
                // a dword (v_pk_add / sub_i16).  Rows 2r and
                // 2r + 1 of a column share a dword: the butterflies between rows 2 and 4 apart and between columns 1 and 2 apart are whole-dword operations (4 stages x 16
                // packed operations; the one-value form took 160), the stage across the halves follows the exchange, and the stage INSIDE a dword is never computed:
                // it is the last one, and |a + b| + |a - b| = 2 max(|a|, |b|).
                uint32_t P[4][4];
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++)
#pragma unroll
                    for (int c = 0; c < 4; c++) P[r2][c] = perm_bytes((uint32_t)m[2 * r2 + 1][c], (uint32_t)m[2 * r2][c], 0x05040100u);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t a0 = pk_add16(P[0][c], P[1][c]), a1 = pk_sub16(P[0][c], P[1][c]), a2 = pk_add16(P[2][c], P[3][c]), a3 = pk_sub16(P[2][c], P[3][c]);
                    P[0][c] = pk_add16(a0, a2); P[1][c] = pk_add16(a1, a3); P[2][c] = pk_sub16(a0, a2); P[3][c] = pk_sub16(a1, a3);
                }
                uint32_t *x = s.rs.scratch + tid * 16;
                const int sw = tid >> 2;
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++) {
                    const uint32_t p0 = pk_add16(P[r2][0], P[r2][1]), p1 = pk_sub16(P[r2][0], P[r2][1]), p2 = pk_add16(P[r2][2], P[r2][3]), p3 = pk_sub16(P[r2][2], P[r2][3]);
                    const uint32_t o[4] = {pk_add16(p0, p2), pk_add16(p1, p3), pk_sub16(p0, p2), pk_sub16(p1, p3)};
                    store_x4(x + 4 * ((r2 + sw) & 3), o);        // chunk r2 = rows 2 r2, 2 r2 + 1 of the four columns; rotated as below
                }
            }
        });
        ex.wave_step([&](int tid) {      // the last butterfly stage across the halves: this lane takes rows 4 half .. 4 half + 3
            const int u = tid >> 1, half = tid & 1, t = u & 15;
            if (!s.rs.tu_log2[t]) return;
            const uint32_t *xa = s.rs.scratch + (tid & ~1) * 16, *xb = xa + 16;      // the pair's two blocks (both lanes of a pair share lane >> 2)
            const int sw = tid >> 2;
            unsigned sum = 0;
            {
                // packed form: the stage across the halves on whole dwords, then 2 max(|lo|, |hi|) per dword for the stage inside it + the absolute sum.  At 8 bit |values| <= 255 x 32
                // here, so eight maxima still fit the 16-bit lanes of an accumulator; at 10 bit (1023 x 32) every maximum is widened
                uint32_t acc[2] = {0, 0};
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    uint32_t va[4], vb[4];
                    const int at = 4 * ((2 * half + c + sw) & 3);
                    load_x4(xa + at, va); load_x4(xb + at, vb);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
#pragma unroll
                        for (int sgn = 0; sgn < 2; sgn++) {
                            const uint32_t v = sgn ? pk_sub16(va[i], vb[i]) : pk_add16(va[i], vb[i]);
                            const uint32_t av = pk_max_i16(v, pk_sub16(0u, v));                  // |lo|, |hi|
                            const uint32_t mx = pk_max_i16(av, (av >> 16) | (av << 16));          // both lanes: max(|lo|, |hi|)
                            if constexpr (sizeof(T) == 1) acc[c] = pk_add16(acc[c], mx); else sum += mx & 0xffffu;
                        }
                    }
                }
                if constexpr (sizeof(T) == 1) sum = (acc[0] & 0xffffu) + (acc[1] & 0xffffu);
                s.rs.scratch[4096 + tid] = 2u * sum;
            }
        });
        ex.phase([&](int tid) {
            const int u = tid >> 1, k = kRingSlot[u >> 4], t = u & 15;
            if (tid >= 1 && tid < 43 && (tid & 1)) s.rbest[tid >> 1] = (unsigned long long)s.cost[tid >> 1] << 4;      // the centre: candidate 0 of every node (odd lanes are free here)
            if ((tid & 1) || !s.rs.tu_log2[t]) return;
            const unsigned satd = (s.rs.scratch[4096 + tid] + s.rs.scratch[4096 + tid + 1] + 2) >> 2;
            ex.atomic_add(&s.fsum[k - 1][s.tile_node[t]], satd);      // the CU's candidate sum, one LDS atomic per tile
        });
        // one lane per (CU, ring position) prices its candidate and bids with an LDS minimum (cost, then position: the order a walk over the positions finds); 21 lanes
        // walking 8 positions each was a serial stretch of wave 0 with the other three waves idle at the barrier
        ex.phase([&](int tid) {
            const int node = tid >> 3, k = 1 + (tid & 7);
            if (node >= 21 || !s.valid[node] || !s.chosen[node]) return;
            const unsigned satd = s.fsum[k - 1][node];
            s.fsum[k - 1][node] = 0;                                      // ready for the next round
            const int mx = s.mvx[node] + kOff[k][0] * step, my = s.mvy[node] + kOff[k][1] * step;
            if (mct || mcb) {                    // a slice: candidates whose filter taps would reach across its edge are out
                int nx, ny, nl;
                node_geom(node, nx, ny, nl);
                if (!mv_rows_ok(y0 + ny, 1 << nl, my, a.h, mct, mcb)) return;
            }
            const unsigned c = (satd << 4) + (unsigned)(lam * (mvd_bits(mx - 4 * csx) + mvd_bits(my - 4 * csy)));
            ex.atomic_min(&s.rbest[node], ((unsigned long long)c << 4) | (unsigned)k);
        });
        ex.phase([&](int tid) {
            if (tid >= 21 || !s.valid[tid] || !s.chosen[tid]) return;
            const unsigned long long best = s.rbest[tid];
            const int k = (int)(best & 15);
            s.mvx[tid] += kOff[k][0] * step; s.mvy[tid] += kOff[k][1] * step;
            s.cost[tid] = (unsigned)(best >> 4);
        });
    }
    };
    refine(sx, sy, oy_x, oy_y);
    if constexpr (BI) {
        // ---- B picture (oracle: orc_analyze_b_frame).  The tree and the list-0 vectors stand; every CU of the tree now refines its list-1 vector
        // (from its own node's integer search against the anchor AFTER this picture), tries the bi-prediction of the two refined vectors and takes the
        // cheapest of SATD << 4 + lambda * (mvd bits + inter_pred_idc bins): list 0 (2), list 1 (2), both (1); ties in that order.
        BiShared &b = *bsh;
        const int sx1 = a.centers1 ? a.centers1[2 * ctu] : 0, sy1 = a.centers1 ? a.centers1[2 * ctu + 1] : 0;
        const int o1_x = x0 + sx1 - R - 4, o1_y = y0 + sy1 - R - 4;
        const int oc1_x = (x0 >> 1) + ((4 * sx1 - 4 * R - 3) >> 3) - 1, oc1_y = (y0 >> 1) + ((4 * sy1 - 4 * R - 3) >> 3) - 1;
        // the 14-bit list-0 prediction of every tile while the list-0 windows are still in LDS (rs.res cannot hold it: it shares its LDS with the
        // fractional search's scratch area)
        ex.phase([&](int tid) {
            if (tid < 21) { b.mx0[tid] = s.mvx[tid]; b.my0[tid] = s.mvy[tid]; b.c0[tid] = s.cost[tid]; b.cb[tid] = 0; }
            {
                const int t = tid >> 4, j = (tid >> 1) & 7, hx = (tid & 1) * 4, txp = t & 3, typ = t >> 2;
                if (s.rs.tu_log2[t]) {
                    const int node = s.tile_node[t], mx = s.mvx[node], my = s.mvy[node];
                    const int px = x0 + txp * 8 + hx + (mx >> 2) - oy_x, py = y0 + typ * 8 + j + (my >> 2) - oy_y;
                    int v[4];
                    luma_quad14<T>(win_y, py * wys + px, wys, mx & 3, my & 3, bd, v);
#pragma unroll
                    for (int i = 0; i < 4; i++) b.p0[(typ * 8 + j) * 32 + txp * 8 + hx + i] = (int16_t)v[i];
                }
            }
            for (int i = tid; i < 512; i += NT) {
                const int pl = i >> 8, k = i & 255, x = k & 15, y = k >> 4, t = (y >> 2) * 4 + (x >> 2);
                if (!s.rs.tu_log2[t]) continue;
                const int node = s.tile_node[t], mx = s.mvx[node], my = s.mvy[node];
                const int px = (x0 >> 1) + x + (mx >> 3) - oc_x, py = (y0 >> 1) + y + (my >> 3) - oc_y;
                b.p0[1024 + i] = (int16_t)chroma_sample14<T>((pl ? win_v : win_u) + py * wcs + px, wcs, mx & 7, my & 7, bd);
            }
        });
        // list-1 windows and integer vectors
        ex.phase([&](int tid) {
            copy_window<T>(win_y, wys, a.ref1[0].p, a.ref1[0].stride, o1_x, o1_y, wy, wy, -PAD_Y, a.w + PAD_Y - 1, -PAD_Y, a.h + PAD_Y - 1, tid);
            copy_window<T>(win_u, wcs, a.ref1[1].p, a.ref1[1].stride, oc1_x, oc1_y, wc, wc, -PAD_C, (a.w >> 1) + PAD_C - 1, -PAD_C, (a.h >> 1) + PAD_C - 1, tid);
            copy_window<T>(win_v, wcs, a.ref1[2].p, a.ref1[2].stride, oc1_x, oc1_y, wc, wc, -PAD_C, (a.w >> 1) + PAD_C - 1, -PAD_C, (a.h >> 1) + PAD_C - 1, tid);
            if (tid < 21) {
                const int32_t *m = a.me1 + ((size_t)ctu * 21 + tid) * 3;
                s.mvx[tid] = m[0]; s.mvy[tid] = m[1]; s.nsum[tid] = 0;
            }
        });
        ex.phase([&](int tid) {          // SATD of the tree's CUs at their list-1 integer vectors
            if (tid >= 16 || !s.rs.tu_log2[tid]) return;
            const int t = tid, txp = t & 3, typ = t >> 2, node = s.tile_node[t];
            const int px = x0 + txp * 8 + (s.mvx[node] >> 2) - o1_x, py = y0 + typ * 8 + (s.mvy[node] >> 2) - o1_y;
            ex.atomic_add(&s.nsum[node], (unsigned)luma_tile_int<T>(win_y, py * wys + px, wys, s.src + typ * 8 * 32 + txp * 8, 32));
        });
        ex.phase([&](int tid) {
            if (tid < 21 && s.valid[tid] && s.chosen[tid])
                s.cost[tid] = (s.nsum[tid] << 4) + (unsigned)(lam * (mvd_bits(s.mvx[tid] - 4 * sx1) + mvd_bits(s.mvy[tid] - 4 * sy1)));
        });
        refine(sx1, sy1, o1_x, o1_y);
        // bi-prediction of the two refined vectors: difference to the source per sample, then one lane per tile takes the 8x8 Hadamard sum
        ex.phase([&](int tid) {
            const int t = tid >> 4, j = (tid >> 1) & 7, hx = (tid & 1) * 4, txp = t & 3, typ = t >> 2;
            if (!s.rs.tu_log2[t]) return;
            const int node = s.tile_node[t], mx = s.mvx[node], my = s.mvy[node];
            const int px = x0 + txp * 8 + hx + (mx >> 2) - o1_x, py = y0 + typ * 8 + j + (my >> 2) - o1_y;
            int v[4];
            luma_quad14<T>(win_y, py * wys + px, wys, mx & 3, my & 3, bd, v);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int at = (typ * 8 + j) * 32 + txp * 8 + hx + i;
                s.rs.scratch[t * 64 + j * 8 + hx + i] = (uint32_t)((int)s.src[at] - weighted_bi((int)b.p0[at], v[i], bd));
            }
        });
        ex.phase([&](int tid) {
            if (tid >= 16 || !s.rs.tu_log2[tid]) return;
            int m[8][8];
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) m[j][i] = (int)s.rs.scratch[tid * 64 + j * 8 + i];
            ex.atomic_add(&b.cb[s.tile_node[tid]], (unsigned)hadamard8_satd(m));
        });
        ex.phase([&](int tid) {
            if (tid >= 21 || !s.valid[tid] || !s.chosen[tid]) return;
            const unsigned bits0 = (unsigned)(mvd_bits(b.mx0[tid] - 4 * sx) + mvd_bits(b.my0[tid] - 4 * sy)), bits1 = (unsigned)(mvd_bits(s.mvx[tid] - 4 * sx1) + mvd_bits(s.mvy[tid] - 4 * sy1));
            const unsigned cbi = (b.cb[tid] << 4) + (unsigned)lam * (bits0 + bits1);
            const unsigned long long k0 = (((unsigned long long)b.c0[tid] + (unsigned long long)(lam * 2)) << 2) | 0, k1 = (((unsigned long long)s.cost[tid] + (unsigned long long)(lam * 2)) << 2) | 1,
                                     k2 = (((unsigned long long)cbi + (unsigned long long)lam) << 2) | 2;
            const unsigned long long kb = k0 <= k1 ? (k0 <= k2 ? k0 : k2) : (k1 <= k2 ? k1 : k2);
            b.mode[tid] = (uint8_t)(kb & 3);
        });
        // final prediction of every CU by its mode: list 0 from the 14-bit samples kept in BiShared::p0, list 1 from the windows, or their average
        ex.phase([&](int tid) {
            if (tid < 16) {
                const int node = s.tile_node[tid], mode = b.mode[node];
                b.tile_mode[tid] = (uint8_t)mode;
                s.tile_mvx[tid] = mode != 1 ? b.mx0[node] : 0; s.tile_mvy[tid] = mode != 1 ? b.my0[node] : 0;
                b.tile_mv1x[tid] = mode != 0 ? s.mvx[node] : 0; b.tile_mv1y[tid] = mode != 0 ? s.mvy[node] : 0;
            }
            {
                const int t = tid >> 4, j = (tid >> 1) & 7, hx = (tid & 1) * 4, txp = t & 3, typ = t >> 2;
                if (s.rs.tu_log2[t]) {
                    const int node = s.tile_node[t], mode = b.mode[node], mx = s.mvx[node], my = s.mvy[node];
                    int v[4] = {0, 0, 0, 0};
                    if (mode != 0) {
                        const int px = x0 + txp * 8 + hx + (mx >> 2) - o1_x, py = y0 + typ * 8 + j + (my >> 2) - o1_y;
                        luma_quad14<T>(win_y, py * wys + px, wys, mx & 3, my & 3, bd, v);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int at = (typ * 8 + j) * 32 + txp * 8 + hx + i, p0 = (int)b.p0[at];
                        s.pred[at] = (T)(mode == 0 ? weighted_uni(p0, bd) : mode == 1 ? weighted_uni(v[i], bd) : weighted_bi(p0, v[i], bd));
                    }
                }
            }
            for (int i = tid; i < 512; i += NT) {
                const int pl = i >> 8, k = i & 255, x = k & 15, y = k >> 4, t = (y >> 2) * 4 + (x >> 2);
                if (!s.rs.tu_log2[t]) continue;
                const int node = s.tile_node[t], mode = b.mode[node], mx = s.mvx[node], my = s.mvy[node], p0 = (int)b.p0[1024 + i];
                int p1 = 0;
                if (mode != 0) {
                    const int px = (x0 >> 1) + x + (mx >> 3) - oc1_x, py = (y0 >> 1) + y + (my >> 3) - oc1_y;
                    p1 = chroma_sample14<T>((pl ? win_v : win_u) + py * wcs + px, wcs, mx & 7, my & 7, bd);
                }
                s.pred[1024 + i] = (T)(mode == 0 ? weighted_uni(p0, bd) : mode == 1 ? weighted_uni(p1, bd) : weighted_bi(p0, p1, bd));
            }
        });
    }
    // (the tiles' vectors for the CU records are written by the motion compensation phase below: a phase of its own for 16 lanes was one more barrier for every CTU)
    if (!BI && a.ip)
    ex.phase([&](int tid) {
        if (tid >= 64 && tid < 85 && s.valid[tid - 64] && s.chosen[tid - 64]) ex.atomic_add(&s.ip_cost, s.cost[tid - 64]);
        if (tid >= 128 && tid < 144 && s.rs.tu_log2[tid - 128]) ex.atomic_add(&s.ip_tiles, 1u);
    });
    // intra second-pass candidate (oracle: orc_analyze_inter_frame): the inter cost is above 4 per sample AND above the source's
    // own AC activity (8x8 Hadamard without the DC term).  The activity is only computed when the first test passes.
    if (!BI && a.ip && s.ip_cost >= ((4u * 64u * s.ip_tiles) << 4)) {
        ex.phase([&](int tid) {
            if (tid >= 16 || !s.rs.tu_log2[tid]) return;
            const T *sp = s.src + (tid >> 2) * 8 * 32 + (tid & 3) * 8;
            int m[8][8];
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) m[j][i] = (int)sp[j * 32 + i];
            ex.atomic_add(&s.ip_act, (unsigned)hadamard8_ac(m));
        });
    }
    // motion compensation of the chosen CUs: every lane predicts 4 luma samples and 2 chroma samples
    if constexpr (!BI)
    ex.phase([&](int tid) {
        if (tid < 16) { s.tile_mvx[tid] = s.mvx[s.tile_node[tid]]; s.tile_mvy[tid] = s.mvy[s.tile_node[tid]]; }      // read by the output phase, behind barriers
        {
            const int t = tid >> 4, j = (tid >> 1) & 7, hx = (tid & 1) * 4, txp = t & 3, typ = t >> 2;
            if (s.rs.tu_log2[t]) {
                int mx = s.mvx[s.tile_node[t]], my = s.mvy[s.tile_node[t]];
                int px = x0 + txp * 8 + hx + (mx >> 2) - oy_x, py = y0 + typ * 8 + j + (my >> 2) - oy_y;
                luma_quad(win_y, py * wys + px, wys, mx & 3, my & 3, bd, s.pred + (typ * 8 + j) * 32 + txp * 8 + hx);
            }
        }
        for (int i = tid; i < 512; i += NT) {
            int pl = i >> 8, k = i & 255, x = k & 15, y = k >> 4, t = (y >> 2) * 4 + (x >> 2);
            if (!s.rs.tu_log2[t]) continue;
            int mx = s.mvx[s.tile_node[t]], my = s.mvy[s.tile_node[t]];
            int px = (x0 >> 1) + x + (mx >> 3) - oc_x, py = (y0 >> 1) + y + (my >> 3) - oc_y;
            s.pred[1024 + i] = (T)chroma_sample<T>((pl ? win_v : win_u) + py * wcs + px, wcs, mx & 7, my & 7, bd);
        }
    });
    ex.phase([&](int tid) {
        for (int i = tid; i < 1536; i += NT) {
            s.rs.res[i] = (int16_t)((int)s.src[i] - (int)s.pred[i]);
            s.rs.desc[i] = pack_loc(locate(s.rs, i));
        }
        if (tid < 48) { s.tu_dc[tid >> 4][tid & 15] = 0; s.tu_dz[tid >> 4][tid & 15] = 0; s.tu_bits[tid >> 4][tid & 15] = 0; }
        if (tid < 3) s.tu_zero[tid] = 0;
    });
    residual_pipeline(ex, s.rs, a.prm.qp, a.prm.qp_c, bd, whole_ctu(), a.prm.rdo_cg > 0 ? (int)(((long long)a.prm.lambda_q4 * a.prm.rdo_cg) >> 1) : 0);
    // RD zero-out (oracle: code_tu_inter): a TU keeps its levels only if SSE_zero << 4 > (SSE_coded << 4) + (lambda * bits >> 4)
    if (a.prm.rdo_zero) {
        ex.phase([&](int tid) {
            const int maxv = (1 << bd) - 1;
            for (int i = 4 * tid; i < 1536; i += 4 * NT) {
                SampleLoc l = locate(s.rs, i);
                if (!l.log2n || !((s.rs.cbf[l.plane] >> l.tile0) & 1)) continue;
                unsigned dc = 0, dz = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int sv = (int)s.src[i + j], pv = (int)s.pred[i + j], d0 = sv - pv, d1 = sv - clip3(0, maxv, pv + s.rs.res[i + j]);
                    dz += (unsigned)(d0 * d0); dc += (unsigned)(d1 * d1);
                }
                if (dc) ex.atomic_add(&s.tu_dc[l.plane][l.tile0], dc);
                if (dz) ex.atomic_add(&s.tu_dz[l.plane][l.tile0], dz);
            }
            for (int sb = tid; sb < 96; sb += NT) {       // level bits per 4x4 sub-block, summed per TU
                int pl = sb < 64 ? 0 : 1 + ((sb - 64) >> 4), k = sb < 64 ? sb : (sb - 64) & 15;
                int per = pl ? 4 : 8, bx = (k & (per - 1)) * 4, by = (k >> (pl ? 2 : 3)) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0;
                SampleLoc l = locate(s.rs, base + by * stride + bx);
                if (!l.log2n || !((s.rs.cbf[l.plane] >> l.tile0) & 1)) continue;
                int b = subblock_bits_q4(s.rs.lvl + base + by * stride + bx, stride);
                if (b) ex.atomic_add(&s.tu_bits[pl][l.tile0], (unsigned)b);
            }
        });
        ex.phase([&](int tid) {
            if (tid >= 48) return;
            const int pl = tid >> 4, t = tid & 15;
            if (!((s.rs.cbf[pl] >> t) & 1) || !s.tu_bits[pl][t]) return;      // only a TU's first tile carries its cbf bit and sums
            const unsigned long long jz = (unsigned long long)s.tu_dz[pl][t] << 4;
            const unsigned long long jc = ((unsigned long long)s.tu_dc[pl][t] << 4) + (((unsigned long long)a.prm.lambda_q4 * (unsigned long long)(s.tu_bits[pl][t] + R_TU)) >> 4);
            if (jz <= jc) ex.atomic_or(&s.tu_zero[pl], 1u << t);
        });
        ex.phase([&](int tid) {
            for (int i = tid; i < 1536; i += NT) {
                SampleLoc l = locate(s.rs, i);
                if (l.log2n && ((s.tu_zero[l.plane] >> l.tile0) & 1)) { s.rs.lvl[i] = 0; s.rs.res[i] = 0; }
            }
            if (tid < 3) s.rs.cbf[tid] &= ~s.tu_zero[tid];       // (nothing in this phase reads cbf)
        });
    }
    // reconstruction + outputs
    ex.phase([&](int tid) {
        const int maxv = (1 << bd) - 1;
        for (int i = 4 * tid; i < 1536; i += 4 * NT) {      // four samples of one row per lane (a TU is at least 4 wide)
            SampleLoc l = locate(s.rs, i);
            if (!l.log2n) continue;
            int gx = (l.plane ? x0 >> 1 : x0) + l.x, gy = (l.plane ? y0 >> 1 : y0) + l.y, v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = clip3(0, maxv, (int)s.pred[i + j] + s.rs.res[i + j]);
            store4(a.rec[l.plane].p + (ptrdiff_t)gy * a.rec[l.plane].stride + gx, v[0], v[1], v[2], v[3]);
            if (a.ip) {
                unsigned sse = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { const int d = (int)s.src[i + j] - v[j]; sse += (unsigned)(d * d); }
                if (sse) ex.atomic_add(&s.ip_sse, (unsigned long long)sse);
            }
            if (!a.sparse_coef || ((s.rs.cbf[l.plane] >> l.tile0) & 1))
                store4(a.coef[l.plane] + (size_t)gy * (l.plane ? a.w >> 1 : a.w) + gx, s.rs.lvl[i], s.rs.lvl[i + 1], s.rs.lvl[i + 2], s.rs.lvl[i + 3]);
        }
        if (a.est || a.ip) {       // rate estimate: coefficient sub-block costs + a header per CU (oracle: inter estimate)
            unsigned e = 0;
            for (int sb = tid; sb < 96; sb += NT) {
                int pl = sb < 64 ? 0 : 1 + ((sb - 64) >> 4), k = sb < 64 ? sb : (sb - 64) & 15;
                int per = pl ? 4 : 8, bx = (k & (per - 1)) * 4, by = (k >> (pl ? 2 : 3)) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0, sh = pl ? 2 : 3;
                if (s.rs.tu_log2[(by >> sh) * 4 + (bx >> sh)]) e += (unsigned)subblock_bits_q4(s.rs.lvl + base + by * stride + bx, stride);
            }
            if (tid < 16 && s.rs.tu_log2[tid]) {
                int nx, ny, nl;
                node_geom(s.tile_node[tid], nx, ny, nl);
                if ((ny >> 3) * 4 + (nx >> 3) == tid)
                    e += R_INTER_CU + (unsigned)(R_TU * (int)(((s.rs.cbf[0] >> tid) & 1) + ((s.rs.cbf[1] >> tid) & 1) + ((s.rs.cbf[2] >> tid) & 1)));
            }
            if (e) ex.atomic_add(&s.est, e);
        }
        if (tid < 16 && s.rs.tu_log2[tid]) {
            int t = tid, txp = t & 3, typ = t >> 2, node = s.tile_node[t], nx, ny, nl;
            node_geom(node, nx, ny, nl);
            int t0 = (ny >> 3) * 4 + (nx >> 3);
            mihevc_cu_rec r;
            r.log2_size = (uint8_t)nl;
            r.flags = (uint8_t)(CU_INTER | ((s.rs.cbf[0] >> t0) & 1 ? CU_CBF_Y : 0) | ((s.rs.cbf[1] >> t0) & 1 ? CU_CBF_CB : 0) |
                                ((s.rs.cbf[2] >> t0) & 1 ? CU_CBF_CR : 0));
            r.chroma_mode = 1; r.qp = (uint8_t)a.prm.qp;
            r.intra_mode[0] = 1; r.intra_mode[1] = r.intra_mode[2] = r.intra_mode[3] = 0;
            r.mvx = (int16_t)s.tile_mvx[t]; r.mvy = (int16_t)s.tile_mvy[t];
            if constexpr (BI) {        // which lists the CU predicts from; the list-1 vector lives in the bytes inter CUs do not use (include/mihevc.h)
                const int mode = bsh->tile_mode[t], x1 = bsh->tile_mv1x[t], y1 = bsh->tile_mv1y[t];
                r.flags |= (uint8_t)((mode != 0 ? CU_L1 : 0) | (mode == 1 ? CU_NOL0 : 0));
                r.intra_mode[0] = (uint8_t)(x1 & 255); r.intra_mode[1] = (uint8_t)((x1 >> 8) & 255); r.intra_mode[2] = (uint8_t)(y1 & 255); r.intra_mode[3] = (uint8_t)((y1 >> 8) & 255);
            }
            r.cbf_y4 = 0; r.pad[0] = r.pad[1] = r.pad[2] = 0;
            a.cu[(size_t)((y0 >> 3) + typ) * (a.w >> 3) + (x0 >> 3) + txp] = r;
        }
    });
    if (a.est || a.ip) ex.phase([&](int tid) {
        if (tid != 0) return;
        if (a.est && s.est) ex.atomic_add_global(a.est, s.est);
        if (a.ip) {
            IpInfo o;
            o.jinter = (s.ip_sse << 4) + (((unsigned long long)a.prm.lambda_q4 * (unsigned long long)s.est) >> 4);
            o.est = s.est;
            o.cand = s.ip_cost > (s.ip_act << 4) && s.ip_cost >= ((4u * 64u * s.ip_tiles) << 4);
            a.ip[ctu] = o;
        }
    });
}

}  // namespace mihevc
