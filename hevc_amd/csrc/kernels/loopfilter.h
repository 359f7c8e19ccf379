// hevc_amd/csrc/kernels/loopfilter.h — K4: deblocking (H.265 8.7.2) and sample adaptive offset (8.7.3).
//
// These are the streaming stencils of the path — the HBM-shaped kernels.  Deblocking runs as
// two passes over the picture (all vertical edges, then all horizontal edges); within a pass every 8-sample edge is
// independent (filters read 4 and write 3 samples either side of edges 8 apart), so one thread owns one 4-sample
// edge segment and the picture is filtered in place.  Algorithmic traffic: one read + one write of the picture
// per pass.  SAO gathers per-CTU statistics with LDS atomics, decides offsets (oracle sao_eval, exactly), then a
// second kernel applies them while writing the final reference picture.
#pragma once
#include <type_traits>

#include "common.h"

namespace mihevc {

template <typename T> struct DeblockArgs {
    Plane<T> rec[3];
    int w, h;
    const mihevc_cu_rec *cu;
    int bit_depth, dir;          // dir 0: vertical edges, 1: horizontal edges
    int y_org;                   // rows in front of the picture's own first row that belong to the slice above (csrc/slice_group.h: its last kSeamRows rows, so that the
                                 // seam is an inner edge); the coding-block and chroma grids count from the picture's own row 0.  0: plain picture
};

DEV int edge_bs(const mihevc_cu_rec &p, const mihevc_cu_rec &q)      // 8.7.2.4 with CU = PU = TU
{
    if (!(p.flags & CU_INTER) || !(q.flags & CU_INTER)) return 2;
    if ((p.flags & CU_CBF_Y) || (q.flags & CU_CBF_Y)) return 1;
    // motion: every list holds ONE picture and the two lists' pictures differ, so "the same reference pictures and number of vectors" means "the same
    // lists", and the vectors to compare are those of the same list (oracle edge_bs)
    const int pu = p.flags & (CU_L1 | CU_NOL0), qu = q.flags & (CU_L1 | CU_NOL0);
    if (pu != qu) return 1;
    if (!(pu & CU_NOL0) && (iabs(p.mvx - q.mvx) >= 4 || iabs(p.mvy - q.mvy) >= 4)) return 1;
    if (pu & CU_L1) {
        const int px = (int16_t)(p.intra_mode[0] | (p.intra_mode[1] << 8)), py = (int16_t)(p.intra_mode[2] | (p.intra_mode[3] << 8));
        const int qx = (int16_t)(q.intra_mode[0] | (q.intra_mode[1] << 8)), qy = (int16_t)(q.intra_mode[2] | (q.intra_mode[3] << 8));
        if (iabs(px - qx) >= 4 || iabs(py - qy) >= 4) return 1;
    }
    return 0;
}

// One 4-sample luma segment of an edge between the blocks p (left / above) and q, and the co-located 2 chroma samples per plane when the edge is a chroma
// edge.  ly / cu / cv point at the segment's first sample on the q side; dir 0: a vertical edge (lines run down the rows), 1: a horizontal one.
// Works on any sample image: the picture in HBM (k_deblock) or a CTU's tile in LDS (the fused loop filter, sao_ctu_program).
template <typename T>
DEV void deblock_lines(T *ly, int ystride, T *cu, T *cv, int cstride, int dir, const mihevc_cu_rec &p, const mihevc_cu_rec &q, int bit_depth, bool chroma_edge)
{
    const int bs = edge_bs(p, q);
    if (!bs) return;
    const int maxv = (1 << bit_depth) - 1, sc = 1 << (bit_depth - 8);
    const int qpl = (p.qp + q.qp + 1) >> 1;
    const int beta = g_tab.beta[clip3(0, 51, qpl)] * sc, tc = g_tab.tc[clip3(0, 53, qpl + 2 * (bs - 1))] * sc;
    {
        const int s = dir == 0 ? 1 : ystride, t = dir == 0 ? ystride : 1;
        T *e = ly;
        int P[4][4], Q[4][4];      // [distance from edge][line]
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < 4; i++) { P[i][k] = e[-(i + 1) * (ptrdiff_t)s + k * (ptrdiff_t)t]; Q[i][k] = e[i * (ptrdiff_t)s + k * (ptrdiff_t)t]; }
        int dp0 = iabs(P[2][0] - 2 * P[1][0] + P[0][0]), dp3 = iabs(P[2][3] - 2 * P[1][3] + P[0][3]);
        int dq0 = iabs(Q[2][0] - 2 * Q[1][0] + Q[0][0]), dq3 = iabs(Q[2][3] - 2 * Q[1][3] + Q[0][3]);
        int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
        if (dpq0 + dpq3 < beta) {
            bool sam0 = 2 * dpq0 < (beta >> 2) && iabs(P[3][0] - P[0][0]) + iabs(Q[0][0] - Q[3][0]) < (beta >> 3) && iabs(P[0][0] - Q[0][0]) < ((5 * tc + 1) >> 1);
            bool sam3 = 2 * dpq3 < (beta >> 2) && iabs(P[3][3] - P[0][3]) + iabs(Q[0][3] - Q[3][3]) < (beta >> 3) && iabs(P[0][3] - Q[0][3]) < ((5 * tc + 1) >> 1);
            bool strong = sam0 && sam3, dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
            for (int k = 0; k < 4; k++) {
                int p0 = P[0][k], p1 = P[1][k], p2 = P[2][k], p3 = P[3][k], q0 = Q[0][k], q1 = Q[1][k], q2 = Q[2][k], q3 = Q[3][k];
                T *c = e + k * (ptrdiff_t)t;
                if (strong) {
                    c[-1 * (ptrdiff_t)s] = (T)clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                    c[-2 * (ptrdiff_t)s] = (T)clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
                    c[-3 * (ptrdiff_t)s] = (T)clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
                    c[0] = (T)clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                    c[s] = (T)clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
                    c[2 * (ptrdiff_t)s] = (T)clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
                } else {
                    int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                    if (iabs(delta) >= tc * 10) continue;
                    delta = clip3(-tc, tc, delta);
                    c[-1 * (ptrdiff_t)s] = (T)clip3(0, maxv, p0 + delta);
                    c[0] = (T)clip3(0, maxv, q0 - delta);
                    if (dep) c[-2 * (ptrdiff_t)s] = (T)clip3(0, maxv, p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
                    if (deq) c[s] = (T)clip3(0, maxv, q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
                }
            }
        }
    }
    if (bs == 2 && chroma_edge) {      // chroma edge on the 8-sample chroma grid (8.7.2.5.5)
        const int tcc = g_tab.tc[clip3(0, 53, chroma_qp_of(qpl) + 2)] * sc;
        for (int ci = 1; ci < 3; ci++) {
            const int s = dir == 0 ? 1 : cstride, t = dir == 0 ? cstride : 1;
            T *e = ci == 1 ? cu : cv;
            for (int k = 0; k < 2; k++) {
                T *c = e + k * (ptrdiff_t)t;
                int p0 = c[-1 * (ptrdiff_t)s], p1 = c[-2 * (ptrdiff_t)s], q0 = c[0], q1 = c[s];
                int delta = clip3(-tcc, tcc, (((q0 - p0) * 4 + p1 - q1 + 4) >> 3));
                c[-1 * (ptrdiff_t)s] = (T)clip3(0, maxv, p0 + delta);
                c[0] = (T)clip3(0, maxv, q0 - delta);
            }
        }
    }
}

// one thread = one segment of the picture in HBM: both passes of k_deblock walk all (8x8 block, half) pairs
template <typename T> DEV void deblock_segment(const DeblockArgs<T> &a, int seg_index)
{
    const int w8 = a.w >> 3, segs = w8 * (a.h >> 3) * 2;
    if (seg_index >= segs) return;
    const int blk = seg_index >> 1, seg = seg_index & 1, bx = blk % w8, by = blk / w8, x = bx * 8, y = by * 8, dir = a.dir;
    const mihevc_cu_rec q = a.cu[blk];
    const int mask = (1 << q.log2_size) - 1;
    const int yg = y - a.y_org;
    if (dir == 0 ? (x == 0 || (x & mask)) : (y == 0 || (yg & mask))) return;
    const mihevc_cu_rec p = a.cu[dir == 0 ? blk - 1 : blk - w8];
    const int ox = dir == 0 ? 0 : 4 * seg, oy = dir == 0 ? 4 * seg : 0;
    deblock_lines<T>(a.rec[0].p + (ptrdiff_t)(y + oy) * a.rec[0].stride + x + ox, a.rec[0].stride,
                     a.rec[1].p + (ptrdiff_t)((y + oy) >> 1) * a.rec[1].stride + ((x + ox) >> 1), a.rec[2].p + (ptrdiff_t)((y + oy) >> 1) * a.rec[2].stride + ((x + ox) >> 1),
                     a.rec[1].stride, dir, p, q, a.bit_depth, ((dir == 0 ? x : yg) & 15) == 0);
}

// ------------------------------------------------------------------------------------------ SAO
template <typename T> struct SaoArgs {
    Plane<const T> src[3];
    Plane<const T> dbk[3];       // deblocked picture
    Plane<T> out[3];             // final reconstruction (padded reference planes)
    int w, h, ctus_w;
    CostParams prm;
    mihevc_sao_ctu *sao;
    unsigned long long *sse;     // optional: 3 x u64 sum of squared error (source vs out), see k_frame_sse / k_sse_fold
    uint32_t *sse_ctu;           // optional: [n_ctu][3], every CTU program leaves the squared error of its own samples here (it holds source and output: no
                                 // second pass over the picture); k_sse_fold adds them up.  A CTB plane's sum is < 1024 x 1023^2 < 2^32
    const mihevc_cu_rec *cu;     // non-null: FUSED loop filter — `dbk` holds the PRE-deblock reconstruction and the CTU program deblocks its tile itself (both edge passes,
                                 // 8.7.2) before the SAO statistics: no deblocking pass over the picture, no in-place picture between the two filters.  One record per 8x8
                                 // block as DeblockArgs::cu; with halo_top / halo_bottom the record rows -1 / h / 8 and the sample rows -4 .. -1 / h .. h + 3 are the neighbours'
    int halo_top, halo_bottom;   // > 0: the picture is one slice (a band of CTU rows) of a picture whose other slices are coded elsewhere and whose filters run
                                 // ACROSS the seams: that many rows above row 0 / below row h - 1 hold the neighbour slices' samples (dbk: deblocked rows, at
                                 // least one; out: up to PAD rows of the final reconstruction, copied in before the border pad, which replicates the outermost
                                 // of them where the whole picture ends earlier); 0: a picture edge
};

// The CTU's tile in LDS: the CTB and a halo of LF_HALO_Y luma / LF_HALO_C chroma samples on every side.  SAO alone needs one deblocked sample around the CTB; the
// FUSED loop filter (SaoArgs::cu) loads the PRE-deblock reconstruction and filters the tile's own edges first: a deblocked sample of [-1, 32] depends on pre-deblock
// samples of [-4, 35] only (an edge reads 4 and writes 3 samples either side; the horizontal-edge pass reads vertical-edge-filtered samples of the same range), chroma
// [-1, 16] on [-2, 17].  Sample (x, y) of the CTB sits at tile index (y + HALO) * stride + x + 2 * HALO: the CTB's quads of four samples are 8-byte aligned.
constexpr int LF_HALO_Y = 4, LF_HALO_C = 2, SAO_TS_Y = 48, SAO_TS_C = 24, LF_ROWS_Y = 32 + 2 * LF_HALO_Y, LF_ROWS_C = 16 + 2 * LF_HALO_C;
template <typename T> struct SaoShared {
    int eo_n[3][4][5], eo_s[3][4][5], bo_n[3][32], bo_s[3][32];     // (these four first: zeroed as one int run)
    int8_t bo_off[3][32];
    long long bo_cost[3][32];
    int8_t eo_off[3][4][4];
    long long eo_cost[3][4];
    alignas(8) T tile_y[LF_ROWS_Y * SAO_TS_Y];
    alignas(8) T tile_c[2][LF_ROWS_C * SAO_TS_C];
    alignas(4) T src[1536];          // the source CTB (Y, U, V as in the CTU kernels)
    // 16 private copies of the statistics (copy = lane & 15, odd stride -> distinct banks): neighbouring samples mostly
    // fall in the same category/band, and 64 lanes hitting one LDS word serialise (SQ_LDS_BANK_CONFLICT, r01 profiles)
    // count and difference sum share one word, (1 << 20) + (d + bias) per sample: half the LDS atomics.  A copy sees at most
    // 64 samples of a plane, so the biased sum (< 64 * 2048) never reaches the count field.
    unsigned priv[3][16][53];        // per copy: eo[4][5] | bo[32], packed | the squared error of the copy's lanes (SaoArgs::sse_ctu)
    unsigned long long band_key[3];  // min over the 29 band positions of ((cost + bias) << 8 | position)
    mihevc_sao_ctu chosen;           // the CTU's parameters, for the apply phase
};

DEVCONST int8_t kEoDx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}};
DEVCONST int8_t kEoDy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
DEV int sgn3(int v) { return (v > 0) - (v < 0); }
// edgeIdx of 8.7.3.2 remapped to the SaoOffsetVal index; 0 when a neighbour lies outside the picture
template <typename T> DEV int eo_category(const Plane<const T> &p, int x, int y, int w, int h, int cls)
{
    int xa = x + kEoDx[cls][0], ya = y + kEoDy[cls][0], xb = x + kEoDx[cls][1], yb = y + kEoDy[cls][1];
    if (xa < 0 || xb < 0 || ya < 0 || yb < 0 || xa >= w || xb >= w || ya >= h || yb >= h) return 0;
    int c = p.p[(ptrdiff_t)y * p.stride + x];
    int e = 2 + sgn3(c - p.p[(ptrdiff_t)ya * p.stride + xa]) + sgn3(c - p.p[(ptrdiff_t)yb * p.stride + xb]);
    return e == 2 ? 0 : e < 2 ? e + 1 : e;
}

// offset minimising (n o^2 - 2 o s) * 16 + lambda * rate, walking from the rounded mean toward 0 (oracle sao_offset_rd).  A CTB plane has n <= 1024
// samples of |difference| < 1024 and |o| <= 31: every intermediate fits 32 bits ((1024 * 961 + 62 * 2^20) * 16 + lambda * 32 < 2^31), so the device
// form divides and multiplies in 32 bits (the 64-bit division alone was ~150 instructions on 108 lanes of every CTU)
DEV int sao_offset_rd(int n, int s, int sign_rule, int lam_q4, int band, int maxoff, long long &cost)
{
    if (n == 0) { cost += lam_q4; return 0; }
    int o = (int)((2u * (unsigned)iabs(s) + (unsigned)n) / (2u * (unsigned)n));
    if (s < 0) o = -o;
    if (sign_rule > 0 && o < 0) o = 0;
    if (sign_rule < 0 && o > 0) o = 0;
    o = clip3(-maxoff, maxoff, o);
    int best_o = 0, best = lam_q4;
    int step = o > 0 ? 1 : -1;
    for (int t = step; o != 0 && t != o + step; t += step) {
        int av = iabs(t), rate = (av < maxoff ? av + 1 : maxoff) + (band ? 1 : 0);
        int c = (n * t * t - 2 * t * s) * 16 + lam_q4 * rate;
        if (c < best) { best = c; best_o = t; }
    }
    cost += best;
    return best_o;
}

template <typename T, class Ex> DEV void sao_ctu_program(Ex &ex, SaoShared<T> &s, const SaoArgs<T> &a, int ctu)
{
    const int cx = ctu % a.ctus_w, cy = ctu / a.ctus_w, bd = a.prm.bit_depth, lam = a.prm.lambda_q4;
    const int maxoff = (1 << (imin(bd, 10) - 5)) - 1;
    const int x0 = cx * 32, y0 = cy * 32, wc = a.w >> 1, hc = a.h >> 1;
    // rows above row 0 / below row h - 1 exist when a neighbour slice lies there (the exchange of csrc/slice_group.h put them into the picture's margins)
    const int ylo = a.halo_top > 0 ? -LF_HALO_Y : 0, yup = a.halo_bottom > 0 ? LF_HALO_Y : 0;
    constexpr int QY = LF_ROWS_Y * 10, QC = LF_ROWS_C * 6, NQ = QY + 2 * QC;      // quads of four samples: luma columns -4 .. 35, chroma columns -4 .. 19
    ex.phase([&](int tid) {
        int *z = &s.eo_n[0][0][0];
        for (int i = tid; i < 3 * (20 + 20 + 32 + 32); i += NT) z[i] = 0;
        for (int i = tid; i < 3 * 16 * 53; i += NT) (&s.priv[0][0][0])[i] = 0;
        // the tile (deblocked, or still to be deblocked here) and the source CTB.  Whole quads of four samples, at most three a lane, and all of a lane's loads are
        // issued before its first LDS store.  Rows and columns outside the picture are read from clamped addresses: nothing ever looks at them.
        auto quad_at = [&](int it, int &pl, int &ty, int &q) {       // item -> plane, tile row, quad of the row (-1: the one left of the CTB)
            if (it < QY) { pl = 0; ty = it / 10; q = it % 10 - 1; return; }
            it -= QY;
            pl = 1 + it / QC; it %= QC; ty = it / 6; q = it % 6 - 1;
        };
        T qv[3][4] = {};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int it = tid + k * NT;
            if (it < NQ) {
                int pl, ty, q;
                quad_at(it, pl, ty, q);
                const int pw = pl ? wc : a.w, ph = pl ? hc : a.h, lo = pl ? ylo >> 1 : ylo, up = pl ? yup >> 1 : yup;
                const int gx = clip3(0, pw - 4, (pl ? x0 >> 1 : x0) + 4 * q), gy = clip3(lo, ph - 1 + up, (pl ? y0 >> 1 : y0) + ty - (pl ? LF_HALO_C : LF_HALO_Y));
                __builtin_memcpy(qv[k], a.dbk[pl].p + (ptrdiff_t)gy * a.dbk[pl].stride + gx, sizeof qv[k]);
            }
        }
        load_ctu_source<T>(s.src, a.src, x0, y0, a.w, a.h, tid);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int it = tid + k * NT;
            if (it < NQ) {
                int pl, ty, q;
                quad_at(it, pl, ty, q);
                T *d = pl ? &s.tile_c[pl - 1][ty * SAO_TS_C + 2 * LF_HALO_C + 4 * q] : &s.tile_y[ty * SAO_TS_Y + 2 * LF_HALO_Y + 4 * q];
                store4(d, (int)qv[k][0], (int)qv[k][1], (int)qv[k][2], (int)qv[k][3]);
            }
        }
    });
    if (a.cu) {
        // deblocking of the tile (8.7.2): every vertical edge segment that reaches columns [-4, 35] x rows [-4, 35], then every horizontal one.  A lane owns one
        // 4-sample segment of one edge: 5 edges (CTB-relative 0, 8, .. 32) x 10 segments along them (-4, 0, .. 32).  Edges 8 apart never touch the same samples.
        T *ty0 = s.tile_y + LF_HALO_Y * SAO_TS_Y + 2 * LF_HALO_Y, *tu0 = s.tile_c[0] + LF_HALO_C * SAO_TS_C + 2 * LF_HALO_C, *tv0 = s.tile_c[1] + LF_HALO_C * SAO_TS_C + 2 * LF_HALO_C;
        const int w8 = a.w >> 3;
        for (int dir = 0; dir < 2; dir++)
            ex.phase([&](int tid) {
                if (tid >= 50) return;
                const int e = 8 * (tid / 10), o = 4 * (tid % 10) - 4;              // position of the edge, position of the segment along it
                const int x = dir == 0 ? e : o, y = dir == 0 ? o : e, gx = x0 + x, gy = y0 + y;      // the segment's first q-side sample
                if (gx < 0 || gx >= a.w || gy < ylo || gy >= a.h + yup) return;
                if (dir == 0 ? gx == 0 : (gy == 0 ? a.halo_top <= 0 : gy >= a.h && a.halo_bottom <= 0)) return;      // a picture edge, not a block edge
                const ptrdiff_t blk = (ptrdiff_t)(gy >> 3) * w8 + (gx >> 3);            // (gy >> 3 = -1: the neighbour slice's record row)
                const mihevc_cu_rec q = a.cu[blk];
                if ((dir == 0 ? gx : gy) & ((1 << q.log2_size) - 1)) return;
                const mihevc_cu_rec p = a.cu[dir == 0 ? blk - 1 : blk - w8];
                deblock_lines<T>(ty0 + y * SAO_TS_Y + x, SAO_TS_Y, tu0 + (y >> 1) * SAO_TS_C + (x >> 1), tv0 + (y >> 1) * SAO_TS_C + (x >> 1), SAO_TS_C, dir, p, q, bd,
                                 ((dir == 0 ? gx : gy) & 15) == 0);
            });
    }
    ex.phase([&](int tid) {
        // a lane owns a strip of horizontally adjacent samples: four luma (row tid >> 3, quad tid & 7) and two chroma (plane tid >> 7): the three tile rows
        // around the strip are read once, and the horizontal differences are shared between neighbours
        auto strip = [&](auto count, int pl, int x, int y) {
            constexpr int N = decltype(count)::value;
            const int pw = pl ? (a.w >> 1) : a.w, ph = pl ? (a.h >> 1) : a.h, gx = (pl ? cx * 16 : cx * 32) + x, gy = (pl ? cy * 16 : cy * 32) + y;
            if (gx >= pw || gy >= ph) return;                      // plane widths are multiples of 4: a strip is inside or outside as a whole
            const T *tp = pl ? s.tile_c[pl - 1] : s.tile_y;
            const int ts = pl ? SAO_TS_C : SAO_TS_Y, ti = pl ? (y + LF_HALO_C) * ts + x + 2 * LF_HALO_C - 1 : (y + LF_HALO_Y) * ts + x + 2 * LF_HALO_Y - 1;      // tile sample left of the strip
            int up[N + 2], mid[N + 2], dn[N + 2], h[N + 1];
#pragma unroll
            for (int j = 0; j < N + 2; j++) { up[j] = tp[ti - ts + j]; mid[j] = tp[ti + j]; dn[j] = tp[ti + ts + j]; }
#pragma unroll
            for (int j = 0; j < N + 1; j++) h[j] = sgn3(mid[j + 1] - mid[j]);
            const T *sp = s.src + (pl ? 1024 + ((pl - 1) << 8) + y * 16 + x : y * 32 + x);
            unsigned *pv = s.priv[pl][tid & 15];
            const bool yin = (gy > 0 || a.halo_top > 0) && (gy < ph - 1 || a.halo_bottom > 0);
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int r = mid[i + 1], d = (int)sp[i] - r;
                const unsigned one = (1u << 20) + (unsigned)(d + (1 << bd));      // count 1, sum d + bias
                const bool xin = gx + i > 0 && gx + i < pw - 1;
                const int e0 = 2 + h[i] - h[i + 1], e1 = 2 + sgn3(r - up[i + 1]) + sgn3(r - dn[i + 1]);
                const int e2 = 2 + sgn3(r - up[i]) + sgn3(r - dn[i + 2]), e3 = 2 + sgn3(r - up[i + 2]) + sgn3(r - dn[i]);
                // edgeIdx of 8.7.3.2 remapped to the SaoOffsetVal index (0, 1, 2, 3, 4 -> 1, 2, 0, 3, 4); 0 when a neighbour lies outside the picture
                auto cat = [](int e, bool in) { return in ? (0x43021 >> (4 * e)) & 7 : 0; };
                ex.atomic_add(&pv[20 + (r >> (bd - 5))], one);
                ex.atomic_add(&pv[0 + cat(e0, xin)], one);
                ex.atomic_add(&pv[5 + cat(e1, yin)], one);
                ex.atomic_add(&pv[10 + cat(e2, xin && yin)], one);
                ex.atomic_add(&pv[15 + cat(e3, xin && yin)], one);
            }
        };
        strip(std::integral_constant<int, 4>{}, 0, (tid & 7) * 4, tid >> 3);
        strip(std::integral_constant<int, 2>{}, 1 + (tid >> 7), (tid & 7) * 2, (tid & 127) >> 3);
    });
    ex.phase([&](int tid) {          // fold the private copies
        for (int i = tid; i < 3 * 52; i += NT) {
            int pl = i / 52, e = i % 52, n = 0, sum = 0;
            for (int c = 0; c < 16; c++) { const unsigned v = s.priv[pl][c][e]; n += (int)(v >> 20); sum += (int)(v & 0xfffffu); }
            sum -= n << bd;                      // take the per-sample bias back out
            if (e < 20) { s.eo_n[pl][e / 5][e % 5] = n; s.eo_s[pl][e / 5][e % 5] = sum; }
            else { s.bo_n[pl][e - 20] = n; s.bo_s[pl][e - 20] = sum; }
        }
    });
    ex.phase([&](int tid) {
        if (tid < 96) {
            int pl = tid >> 5, b = tid & 31;
            long long c = 0;
            s.bo_off[pl][b] = (int8_t)sao_offset_rd(s.bo_n[pl][b], s.bo_s[pl][b], 0, lam, 1, maxoff, c);
            s.bo_cost[pl][b] = c;
        } else if (tid < 96 + 12) {
            int pl = (tid - 96) >> 2, c = (tid - 96) & 3;
            long long cost = (long long)lam * 4;
            for (int k = 1; k <= 4; k++) s.eo_off[pl][c][k - 1] = (int8_t)sao_offset_rd(s.eo_n[pl][c][k], s.eo_s[pl][c][k], k <= 2 ? 1 : -1, lam, 0, maxoff, cost);
            s.eo_cost[pl][c] = cost;
        } else if (tid >= 128 && tid < 131) s.band_key[tid - 128] = ~0ull;      // for the next phase's minimum (a phase of its own was one more barrier)
    });
    ex.phase([&](int tid) {          // best band position per plane: 87 lanes, ties -> lowest position
        if (tid < 87) {
            int pl = tid / 29, p = tid % 29;
            long long c = s.bo_cost[pl][p] + s.bo_cost[pl][p + 1] + s.bo_cost[pl][p + 2] + s.bo_cost[pl][p + 3];
            ex.atomic_min(&s.band_key[pl], ((unsigned long long)(c + (1ll << 50)) << 8) | (unsigned)p);
        }
    });
    ex.phase([&](int tid) {
        if (tid != 0) return;
        // candidates per plane in fixed order: 0 off, 1 band (best of 29 positions), 2..5 edge classes.  Every loop is unrolled so the
        // small tables stay in registers (they were 160 bytes of scratch per lane in round 1)
        long long cost[3][6];
        int band[3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            band[pl] = (int)(s.band_key[pl] & 255);
            long long bestb = (long long)(s.band_key[pl] >> 8) - (1ll << 50);
            cost[pl][0] = 0;
            cost[pl][1] = bestb + (long long)lam * 7;
#pragma unroll
            for (int c = 0; c < 4; c++) cost[pl][2 + c] = s.eo_cost[pl][c];
        }
        int bl = 0, bc = 0;
        long long best_l = cost[0][0], best_c = cost[1][0] + cost[2][0];
#pragma unroll
        for (int k = 1; k < 6; k++) {
            if (cost[0][k] < best_l) { best_l = cost[0][k]; bl = k; }
            if (cost[1][k] + cost[2][k] < best_c) { best_c = cost[1][k] + cost[2][k]; bc = k; }
        }
        mihevc_sao_ctu o = {};
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const int k = pl == 0 ? bl : bc, type = k == 0 ? 0 : k == 1 ? 1 : 2;
            if (pl < 2) { o.type[pl] = (uint8_t)type; o.eo_class[pl] = (uint8_t)(type == 2 ? k - 2 : 0); }
            o.band_pos[pl] = (uint8_t)(type == 1 ? band[pl] : 0);
#pragma unroll
            for (int i = 0; i < 4; i++) o.offset[pl][i] = type == 1 ? s.bo_off[pl][band[pl] + i] : type == 2 ? s.eo_off[pl][k == 0 ? 0 : k < 2 ? 0 : k - 2][i] : (int8_t)0;
        }
        a.sao[ctu] = o;
        s.chosen = o;
    });
    // apply (8.7.3) to the CTU's own samples, from the deblocked tile already in LDS (a CTU's samples depend on its own parameters and on deblocked
    // neighbour SAMPLES only): the separate pass over the picture re-read it from HBM behind one more launch boundary (18 + 6 us per step).  A lane
    // owns the strips of the statistics phase: four luma and two chroma samples.
    ex.phase([&](int tid) {
        const mihevc_sao_ctu &o = s.chosen;
        const int maxv = (1 << bd) - 1;
        auto strip = [&](auto count, int pl, int x, int y) {
            constexpr int N = decltype(count)::value;
            const int pw = pl ? (a.w >> 1) : a.w, ph = pl ? (a.h >> 1) : a.h, gx = (pl ? cx * 16 : cx * 32) + x, gy = (pl ? cy * 16 : cy * 32) + y;
            if (gx >= pw || gy >= ph) return;
            const T *tp = pl ? s.tile_c[pl - 1] : s.tile_y;
            const int ts = pl ? SAO_TS_C : SAO_TS_Y, ti = pl ? (y + LF_HALO_C) * ts + x + 2 * LF_HALO_C - 1 : (y + LF_HALO_Y) * ts + x + 2 * LF_HALO_Y - 1;
            const int type = o.type[pl ? 1 : 0], cls = o.eo_class[pl ? 1 : 0];
            int v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int r = tp[ti + 1 + i];
                int out = r;
                if (type == 2) {
                    const int xa = gx + i + kEoDx[cls][0], ya = gy + kEoDy[cls][0], xb = gx + i + kEoDx[cls][1], yb = gy + kEoDy[cls][1];
                    const int ylo = a.halo_top > 0 ? -1 : 0, yhi = a.halo_bottom > 0 ? ph : ph - 1;
                    if (!(xa < 0 || xb < 0 || ya < ylo || yb < ylo || xa >= pw || xb >= pw || ya > yhi || yb > yhi)) {
                        const int e = 2 + sgn3(r - tp[ti + 1 + i + kEoDy[cls][0] * ts + kEoDx[cls][0]]) + sgn3(r - tp[ti + 1 + i + kEoDy[cls][1] * ts + kEoDx[cls][1]]);
                        const int k = e == 2 ? 0 : e < 2 ? e + 1 : e;
                        if (k) out = clip3(0, maxv, r + o.offset[pl][k - 1]);
                    }
                } else if (type == 1) {
                    const int k = ((r >> (bd - 5)) - o.band_pos[pl]) & 31;
                    if (k < 4) out = clip3(0, maxv, r + o.offset[pl][k]);
                }
                v[i] = out;
            }
            T *dst = a.out[pl].p + (ptrdiff_t)gy * a.out[pl].stride + gx;
            if (N == 4) store4(dst, v[0], v[1], v[2], v[3]);
            else { dst[0] = (T)v[0]; dst[1] = (T)v[1]; }
            if (a.sse_ctu) {          // squared error of the strip into the lane's private copy (word 52 of a copy is the spare one, zero since phase 1)
                const T *sp = s.src + (pl ? 1024 + ((pl - 1) << 8) + y * 16 + x : y * 32 + x);
                unsigned e = 0;
#pragma unroll
                for (int i = 0; i < N; i++) { const int d = (int)sp[i] - v[i]; e += (unsigned)(d * d); }
                if (e) ex.atomic_add(&s.priv[pl][tid & 15][52], e);
            }
        };
        strip(std::integral_constant<int, 4>{}, 0, (tid & 7) * 4, tid >> 3);
        strip(std::integral_constant<int, 2>{}, 1 + (tid >> 7), (tid & 7) * 2, (tid & 127) >> 3);
    });
    if (a.sse_ctu) ex.phase([&](int tid) {
        if (tid >= 3) return;
        unsigned e = 0;
        for (int c = 0; c < 16; c++) e += s.priv[tid][c][52];
        a.sse_ctu[3 * ctu + tid] = e;
    });
}

// apply (8.7.3) to one sample value v at (gx, gy) of plane pl
template <typename T> DEV int sao_sample_value(const SaoArgs<T> &a, const mihevc_sao_ctu &o, int pl, int gx, int gy, int v)
{
    const int pw = pl ? a.w >> 1 : a.w, ph = pl ? a.h >> 1 : a.h, bd = a.prm.bit_depth, maxv = (1 << bd) - 1;
    const int type = o.type[pl ? 1 : 0];
    if (type == 2) {
        int k = eo_category<T>(a.dbk[pl], gx, gy, pw, ph, o.eo_class[pl ? 1 : 0]);
        if (k) v = clip3(0, maxv, v + o.offset[pl][k - 1]);
    } else if (type == 1) {
        int k = ((v >> (bd - 5)) - o.band_pos[pl]) & 31;
        if (k < 4) v = clip3(0, maxv, v + o.offset[pl][k]);
    }
    return v;
}
// one thread per sample of a plane-row segment; also usable with sao == nullptr (plain copy)
template <typename T> DEV void sao_apply_sample(const SaoArgs<T> &a, int pl, int gx, int gy)
{
    int v = a.dbk[pl].p[(ptrdiff_t)gy * a.dbk[pl].stride + gx];
    if (a.sao) v = sao_sample_value<T>(a, a.sao[(gy >> (pl ? 4 : 5)) * a.ctus_w + (gx >> (pl ? 4 : 5))], pl, gx, gy, v);
    a.out[pl].p[(ptrdiff_t)gy * a.out[pl].stride + gx] = (T)v;
}
// one thread per four samples of a row (gx a multiple of 4: the quad lies in one CTU; plane widths are multiples of 4): one load and one
// store per quad, the CTU's parameters fetched once, and CTUs without SAO are a plain copy
template <typename T> DEV void sao_apply_quad(const SaoArgs<T> &a, int pl, int gx, int gy)
{
    T q[4];
    __builtin_memcpy(q, a.dbk[pl].p + (ptrdiff_t)gy * a.dbk[pl].stride + gx, sizeof q);
    if (a.sao) {
        const mihevc_sao_ctu &o = a.sao[(gy >> (pl ? 4 : 5)) * a.ctus_w + (gx >> (pl ? 4 : 5))];
        if (o.type[pl ? 1 : 0]) {
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = (T)sao_sample_value<T>(a, o, pl, gx + j, gy, (int)q[j]);
        }
    }
    store4(a.out[pl].p + (ptrdiff_t)gy * a.out[pl].stride + gx, q[0], q[1], q[2], q[3]);
}

// border extension of a padded plane: thread per border sample
template <typename T> DEV void pad_sample(Plane<T> p, int w, int h, int pad, int idx)
{
    const int pw = w + 2 * pad, ph = h + 2 * pad;
    if (idx >= pw * ph) return;
    int x = idx % pw - pad, y = idx / pw - pad;
    if (x >= 0 && x < w && y >= 0 && y < h) return;
    p.p[(ptrdiff_t)y * p.stride + x] = p.p[(ptrdiff_t)clip3(0, h - 1, y) * p.stride + clip3(0, w - 1, x)];
}

// the same, indexed over the border samples only: pad rows above, pad rows below, then 2 x pad columns beside each picture row
HDI int pad_border_count(int w, int h, int pad) { return 2 * pad * (w + 2 * pad) + 2 * pad * h; }
template <typename T> DEV void pad_border_sample(Plane<T> p, int w, int h, int pad, int j)
{
    const int pw = w + 2 * pad, band = pad * pw;
    int idx;
    if (j < band) idx = j;
    else if (j < 2 * band) idx = (pad + h) * pw + (j - band);
    else {
        const int k = j - 2 * band, y = k / (2 * pad), c = k % (2 * pad);
        if (y >= h) return;
        idx = (pad + y) * pw + (c < pad ? c : w + c);
    }
    pad_sample<T>(p, w, h, pad, idx);
}

// the same border in quads of four samples (pad, plane width and row pitch are multiples of 4: a quad is border on one side of the picture or
// a copy of picture columns, never both): one aligned store per lane instead of four byte stores, a quarter of the index arithmetic
HDI int pad_border_quads(int w, int h, int pad) { return pad_border_count(w, h, pad) >> 2; }
// top / bottom: that many pad rows above / below hold neighbour slices' samples (columns [0, w)): only their left and right ends are filled in, and
// rows further out replicate the outermost of them (the whole picture ends there)
template <typename T> DEV void pad_border_quad(Plane<T> p, int w, int h, int pad, int j, int top = 0, int bottom = 0)
{
    const int qpr = (w + 2 * pad) >> 2, band = pad * qpr, side = pad >> 2;      // quads per padded row / per band above or below / per side of a picture row
    int x, y;
    if (j < 2 * band) {
        const int k = j < band ? j : j - band;
        y = (j < band ? -pad : h) + k / qpr; x = (k % qpr) * 4 - pad;
    } else {
        const int k = j - 2 * band, r = k / (2 * side), c = k % (2 * side);
        if (r >= h) return;
        y = r; x = c < side ? c * 4 - pad : w + (c - side) * 4;
    }
    const T *row = p.p + (ptrdiff_t)clip3(-top, h - 1 + bottom, y) * p.stride;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (int)row[clip3(0, w - 1, x + i)];
    store4(p.p + (ptrdiff_t)y * p.stride + x, v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------------------------------ rows between the slices of one picture
// one job = `rows` rows of `row_bytes` bytes (a multiple of 4, 4-byte aligned at both ends) from src to dst; src may live in another device's memory
// (peer access over xGMI): the halo exchange of csrc/slice_group.h is a handful of these per lane and step
struct RowCopy { const void *src; void *dst; int row_bytes, rows, src_pitch, dst_pitch; };
DEV void copy_rows_item(const RowCopy &j, int i, int n_items)
{
    const int per = j.row_bytes >> 2, total = per * j.rows;
    for (int k = i; k < total; k += n_items) {
        const int r = k / per, c = k % per;
        ((uint32_t *)((uint8_t *)j.dst + (ptrdiff_t)r * j.dst_pitch))[c] = ((const uint32_t *)((const uint8_t *)j.src + (ptrdiff_t)r * j.src_pitch))[c];
    }
}

}  // namespace mihevc
