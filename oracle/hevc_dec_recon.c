/* oracle/hevc_dec_recon.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * The decoder's OWN reconstruction arithmetic: scaling + inverse transforms (8.6.2 - 8.6.4), intra sample prediction (8.4.4.2.3 - 8.4.4.2.6),
 * deblocking (8.7.2) and SAO (8.7.3), written a second time from the clauses of ITU-T H.265 and sharing NOTHING with hevc_oracle.c — no
 * function, no table, no helper macro.  Until round 3 hevc_dec.c called the oracle's orc_dequant / orc_inv_transform / orc_intra_* /
 * orc_deblock_frame / orc_sao_apply_frame, so "the stream decodes to the encoder's reconstruction" could not see an error in those five;
 * now an error has to be made twice, in two differently organised implementations, to go unnoticed:
 *   - transforms here are the direct matrix product of 8.6.4.2 with a full 32x32 transMatrix built from the cosine quarter-wave symmetry
 *     (hevc_oracle.c: partial butterflies over a row table);
 *   - intra prediction here works on p[x][y] addressed as in the clauses, with ref[] built per 8.4.4.2.6 (hevc_oracle.c: one linear
 *     4N+1 array with mode flipping);
 *   - deblocking here walks edges per 8x8 grid position and 4-sample segment in clause order with a per-picture Bs map
 *     (hevc_oracle.c: per-CU-record passes);
 *   - SAO here classifies per sample from SaoTypeIdx / SaoEoClass tables with the picture / slice / tile clipping rules of 8.7.3.
 * Parity with libx265 stays UNPINNED (no third-party decoder exists on this pool); this is the strongest second opinion available.
 * tests/test_decoder_second_opinion.py runs both implementations on random inputs and on closed-form cases.
 */
#include <stdlib.h>
#include <string.h>

#include "hevc_dec_recon.h"

static inline int d2_clip(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static inline int d2_abs(int v) { return v < 0 ? -v : v; }
static inline int d2_sign(int v) { return (v > 0) - (v < 0); }

/* ------------------------------------------------------------------ Table 8-10: QpC as a function of qPi (ChromaArrayType == 1) */
int d2_chroma_qp(int qpi)
{
    static const signed char t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};      /* qPi = 30 .. 43 */
    if (qpi < 30) return qpi;
    if (qpi >= 44) return qpi - 6;
    return t[qpi - 30];
}

/* ------------------------------------------------------------------ 8.6.4.2: transMatrix.  Row k of the 32-point DCT is
 * c(k) cos((2n + 1) k pi / 64); the standard's integers for the quarter wave m = 0 .. 32 (angle m pi / 64) are listed once and every
 * coefficient is looked up through the symmetries cos(pi - a) = -cos(a), cos(2 pi - a) = cos(a).  (coefficients: equation 8-? "transMatrix",
 * columns 0 .. 15 and 16 .. 31.) */
static const signed char kQuarter[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                         61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
static int trans_coef(int k, int n)              /* transMatrix[k][n], 32-point */
{
    int m = ((2 * n + 1) * k) & 127;             /* angle in units of pi / 64, period 128 */
    if (m > 64) m = 128 - m;                     /* cos(2 pi - a) = cos(a) */
    return m <= 32 ? kQuarter[m] : -kQuarter[64 - m];
}
static const signed char kDst7[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};      /* 8.6.4.2, equation for nTbS = 4, trType = 1 */

/* 8.6.2 + 8.6.3 + 8.6.4 for one transform block without scaling lists (m = 16): residual samples r[y * n + x] */
static void residual_block(const int16_t *lvl, int *r, int log2n, int qp, int bit_depth, int dst4)
{
    const int n = 1 << log2n;
    static const int levelScale[6] = {40, 45, 51, 57, 64, 72};
    int d[32 * 32], e[32 * 32];
    /* 8.6.3: bdShift = BitDepth + Log2(nTbS) + 10 - 15 (extended_precision_processing off: CoeffMinY = -32768) */
    const int bdShift = bit_depth + log2n - 5;
    const int qpp = qp + 6 * (bit_depth - 8);    /* qP = Qp'Y = QpY + QpBdOffsetY (the caller hands QpY / QpC) */
    for (int i = 0; i < n * n; i++) {
        long long v = ((long long)lvl[i] * 16 * levelScale[qpp % 6] << (qpp / 6)) + (1LL << (bdShift - 1));
        d[i] = d2_clip(-32768, 32767, (int)(v >> bdShift));
    }
    /* 8.6.4.1: columns first (each column x: e[x][y] = sum_j transMatrix[j][y] * d[x][j]), intermediate clip after >> 7, then rows, then
     * bdShift = 20 - BitDepth */
    const int step = 32 >> log2n;                /* the nTbS-point matrix is rows 0, step, 2 step ... of the 32-point one, first nTbS columns */
    for (int x = 0; x < n; x++)
        for (int y = 0; y < n; y++) {
            long long s = 0;
            for (int j = 0; j < n; j++) s += (long long)(dst4 ? kDst7[j][y] : trans_coef(j * step, y)) * d[j * n + x];
            e[y * n + x] = d2_clip(-32768, 32767, (int)((s + 64) >> 7));
        }
    const int sh2 = 20 - bit_depth;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            long long s = 0;
            for (int j = 0; j < n; j++) s += (long long)(dst4 ? kDst7[j][x] : trans_coef(j * step, x)) * e[y * n + j];
            r[y * n + x] = (int)((s + (1LL << (sh2 - 1))) >> sh2);
        }
}

/* ... then 8.6.7 (picture construction): prediction + residual, clipped to the sample range */
void d2_residual_add(pix *dst, int stride, const int16_t *lvl, int log2n, int qp, int bit_depth, int dst4)
{
    const int n = 1 << log2n, maxv = (1 << bit_depth) - 1;
    int r[32 * 32];
    residual_block(lvl, r, log2n, qp, bit_depth, dst4);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) dst[(size_t)y * stride + x] = (pix)d2_clip(0, maxv, dst[(size_t)y * stride + x] + r[y * n + x]);
}

/* residual samples only (test hook) */
void d2_residual(const int16_t *lvl, int32_t *res, int log2n, int qp, int bit_depth, int dst4) { residual_block(lvl, res, log2n, qp, bit_depth, dst4); }

/* ------------------------------------------------------------------ 8.4.4.2: intra sample prediction
 * ref: 4 nTbS + 1 neighbouring samples AFTER the substitution process of 8.4.4.2.2, ordered p[-1][2 nTbS - 1] ... p[-1][0], p[-1][-1],
 * p[0][-1] ... p[2 nTbS - 1][-1] (the order in which 8.4.4.2.2 walks them).  PL(y) = p[-1][y], PT(x) = p[x][-1], y / x = -1 is the corner. */
void d2_intra_pred(const pix *ref, pix *dst, int stride, int log2n, int mode, int c_idx, int bit_depth, int strong_enabled)
{
    const int n = 1 << log2n, maxv = (1 << bit_depth) - 1;
    int left[65], top[65];                       /* index + 1: [0] is the corner p[-1][-1] */
    for (int y = -1; y < 2 * n; y++) left[y + 1] = ref[2 * n - 1 - y];
    for (int x = -1; x < 2 * n; x++) top[x + 1] = x < 0 ? ref[2 * n] : ref[2 * n + 1 + x];
#define PL(y) left[(y) + 1]
#define PT(x) top[(x) + 1]
    /* 8.4.4.2.3 filtering process of neighbouring samples */
    int filter = 0;
    if (c_idx == 0 && mode != 1 && n != 4) {
        const int dv = d2_abs(mode - 26), dh = d2_abs(mode - 10), minDistVerHor = dv < dh ? dv : dh;
        const int thres = n == 8 ? 7 : n == 16 ? 1 : 0;      /* intraHorVerDistThres[nTbS] */
        filter = minDistVerHor > thres;
    }
    if (filter) {
        int fl[65], ft[65];
        const int lim = 1 << (bit_depth - 5);
        const int bi = strong_enabled && n == 32 && d2_abs(PT(-1) + PT(2 * n - 1) - 2 * PT(n - 1)) < lim && d2_abs(PL(-1) + PL(2 * n - 1) - 2 * PL(n - 1)) < lim;
        if (bi) {
            fl[0] = ft[0] = PT(-1);
            for (int i = 0; i < 63; i++) {
                fl[i + 1] = ((63 - i) * PL(-1) + (i + 1) * PL(63) + 32) >> 6;
                ft[i + 1] = ((63 - i) * PT(-1) + (i + 1) * PT(63) + 32) >> 6;
            }
            fl[64] = PL(63); ft[64] = PT(63);
        } else {
            fl[0] = ft[0] = (PL(0) + 2 * PL(-1) + PT(0) + 2) >> 2;
            for (int i = 0; i < 2 * n - 1; i++) {
                fl[i + 1] = (PL(i + 1) + 2 * PL(i) + PL(i - 1) + 2) >> 2;
                ft[i + 1] = (PT(i - 1) + 2 * PT(i) + PT(i + 1) + 2) >> 2;
            }
            fl[2 * n] = PL(2 * n - 1); ft[2 * n] = PT(2 * n - 1);
        }
        memcpy(left, fl, sizeof(int) * (2 * n + 1));
        memcpy(top, ft, sizeof(int) * (2 * n + 1));
    }
    if (mode == 0) {                             /* 8.4.4.2.4 INTRA_PLANAR */
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                dst[(size_t)y * stride + x] = (pix)(((n - 1 - x) * PL(y) + (x + 1) * PT(n) + (n - 1 - y) * PT(x) + (y + 1) * PL(n) + n) >> (log2n + 1));
        return;
    }
    if (mode == 1) {                             /* 8.4.4.2.5 INTRA_DC */
        int sum = n;
        for (int i = 0; i < n; i++) sum += PT(i) + PL(i);
        const int dc = sum >> (log2n + 1);
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) dst[(size_t)y * stride + x] = (pix)dc;
        if (c_idx == 0 && n < 32) {
            dst[0] = (pix)((PL(0) + 2 * dc + PT(0) + 2) >> 2);
            for (int x = 1; x < n; x++) dst[x] = (pix)((PT(x) + 3 * dc + 2) >> 2);
            for (int y = 1; y < n; y++) dst[(size_t)y * stride] = (pix)((PL(y) + 3 * dc + 2) >> 2);
        }
        return;
    }
    /* 8.4.4.2.6 INTRA_ANGULAR2 .. 34: Table 8-4 (intraPredAngle) and Table 8-5 (invAngle) */
    static const signed char angle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0,
                                          2, 5, 9, 13, 17, 21, 26, 32};
    static const short inv[35] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910,
                                  -1638, -4096, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int a = angle[mode], vertical = mode >= 18;
    int rbuf[3 * 32 + 2], *r = rbuf + 32;        /* ref[-nTbS .. 2 nTbS] */
    /* main side: the row above for modes >= 18, the left column otherwise; the other one is the side reference projected by invAngle */
#define MAIN(i) (vertical ? PT(i) : PL(i))
#define SIDE(i) (vertical ? PL(i) : PT(i))
    for (int x = 0; x <= n; x++) r[x] = MAIN(-1 + x);
    if (a < 0) {
        const int last = (n * a) >> 5;
        if (last < -1)
            for (int x = last; x <= -1; x++) r[x] = SIDE(-1 + ((x * inv[mode] + 128) >> 8));
    } else {
        for (int x = n + 1; x <= 2 * n; x++) r[x] = MAIN(-1 + x);
    }
    for (int j = 0; j < n; j++) {                /* j runs along the prediction direction's minor axis: y for vertical modes, x for horizontal ones */
        const int idx = ((j + 1) * a) >> 5, fact = ((j + 1) * a) & 31;
        for (int i = 0; i < n; i++) {
            const int v = fact ? ((32 - fact) * r[i + idx + 1] + fact * r[i + idx + 2] + 16) >> 5 : r[i + idx + 1];
            if (vertical) dst[(size_t)j * stride + i] = (pix)v;
            else dst[(size_t)i * stride + j] = (pix)v;
        }
    }
    /* boundary smoothing of the pure vertical / horizontal modes (predModeIntra 26 / 10, luma, nTbS < 32; disableIntraBoundaryFilter is 0 in
     * version-1 profiles) */
    if (c_idx == 0 && n < 32) {
        if (mode == 26) for (int y = 0; y < n; y++) dst[(size_t)y * stride] = (pix)d2_clip(0, maxv, PT(0) + ((PL(y) - PL(-1)) >> 1));
        if (mode == 10) for (int x = 0; x < n; x++) dst[x] = (pix)d2_clip(0, maxv, PL(0) + ((PT(x) - PT(-1)) >> 1));
    }
#undef MAIN
#undef SIDE
#undef PL
#undef PT
}

/* ------------------------------------------------------------------ 8.7.2 deblocking filter process */
static const unsigned char kBetaPrime[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
                                             26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};      /* Table 8-12 */
static const unsigned char kTcPrime[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3,
                                           3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};               /* Table 8-12 */

typedef struct {
    const d2_picture_info *pi;
    pix *pl[3];
    int stride[3];
    unsigned char *bs;      /* [dir][(y >> 2) * (w >> 3) ... ] boundary strength per 4-sample segment of every 8x8-grid edge */
} dbk_ctx;

static const orc_cu_rec *rec_at(const d2_picture_info *pi, int x, int y) { return &pi->cu[(size_t)(y >> 3) * (pi->w >> 3) + (x >> 3)]; }
static int band_of(const d2_picture_info *pi, int y)
{
    const int row = y >> 5;
    int k = 0;
    while (k + 1 < pi->n_bands && pi->band_row0[k + 1] <= row) k++;
    return k;
}
static int tile_index(const d2_picture_info *pi, int x, int y)
{
    int i = 0, j = 0;
    const int cx = x >> 5, cy = y >> 5;
    while (i + 1 < pi->tile_cols && pi->col_bd[i + 1] <= cx) i++;
    while (j + 1 < pi->tile_rows && pi->row_bd[j + 1] <= cy) j++;
    return j * pi->tile_cols + i;
}

/* 8.7.2.3 (transform / prediction block edges; here every coding block is one prediction block, and one transform block unless it is an
 * intra NxN block whose 4x4 transform blocks meet off the 8x8 grid) + 8.7.2.4 (boundary filtering strength) for the edge whose Q sample is
 * (xq, yq) and whose P sample is the one left of / above it.  Returns 0 when there is no edge to filter. */
static int edge_bs(const d2_picture_info *pi, int dir, int xq, int yq)
{
    const int xp = dir ? xq : xq - 1, yp = dir ? yq - 1 : yq;
    if (xp < 0 || yp < 0) return 0;                                               /* picture boundary */
    const orc_cu_rec *q = rec_at(pi, xq, yq), *p = rec_at(pi, xp, yp);
    const int size = 1 << q->log2_size;
    if ((dir ? yq : xq) & (size - 1)) return 0;                                   /* inside a coding block: no transform or prediction edge on the 8x8 grid */
    /* slice / tile boundaries: the left / upper edge of a slice is filtered only if the CURRENT slice (the one holding q0) allows it
     * (slice_loop_filter_across_slices_enabled_flag, 7.4.7.1: "deblocking ... across the left and upper boundary of the current slice"); an edge
     * that coincides with the lower boundary of a slice that forbids it is left alone too (same clause, last sentence); tile boundaries follow
     * loop_filter_across_tiles_enabled_flag */
    const int bq = band_of(pi, yq), bp = band_of(pi, yp);
    if (bq != bp && !(pi->band_lf_across[bq] && pi->band_lf_across[bp])) return 0;
    if (!pi->lf_across_tiles && tile_index(pi, xq, yq) != tile_index(pi, xp, yp)) return 0;
    if (!(q->flags & ORC_F_INTER) || !(p->flags & ORC_F_INTER)) return 2;
    if ((q->flags & ORC_F_CBF_Y) || (p->flags & ORC_F_CBF_Y)) return 1;          /* transform block edge with a non-zero luma level on either side */
    /* prediction: one motion vector and one reference picture per block in P slices (B slices: see d2_motion_differs) */
    return d2_motion_differs(p, q, pi->poc_of_ref);
}

/* 8.7.2.4, motion part: different reference pictures or numbers of motion vectors, or a vector component differing by 4 or more in
 * quarter-sample units.  The determination is made on the PICTURES referenced, not on list indices. */
int d2_motion_differs(const orc_cu_rec *p, const orc_cu_rec *q, const int *poc_of_ref)
{
    /* the pictures each block predicts from (poc_of_ref[list]: RefPicList0[0], RefPicList1[0]; NULL in P slices: one list) and its vectors */
    int np = 0, nq = 0, rp[2], rq[2], vp[2][2], vq[2][2];
    for (int l = 0; l < 2; l++) {
        const int up = l ? (p->flags & ORC_F_L1) != 0 : !(p->flags & ORC_F_NOL0), uq = l ? (q->flags & ORC_F_L1) != 0 : !(q->flags & ORC_F_NOL0);
        if (up) { rp[np] = poc_of_ref ? poc_of_ref[l] : l; vp[np][0] = l ? orc_mv1x(p) : p->mvx; vp[np][1] = l ? orc_mv1y(p) : p->mvy; np++; }
        if (uq) { rq[nq] = poc_of_ref ? poc_of_ref[l] : l; vq[nq][0] = l ? orc_mv1x(q) : q->mvx; vq[nq][1] = l ? orc_mv1y(q) : q->mvy; nq++; }
    }
    if (np != nq) return 1;                                                      /* different number of motion vectors */
#define FAR(a, b) (d2_abs((a)[0] - (b)[0]) >= 4 || d2_abs((a)[1] - (b)[1]) >= 4)
    if (np == 1) return rp[0] != rq[0] || FAR(vp[0], vq[0]);
    /* two vectors each */
    if (!((rp[0] == rq[0] && rp[1] == rq[1]) || (rp[0] == rq[1] && rp[1] == rq[0]))) return 1;      /* different reference pictures */
    if (rp[0] != rp[1]) {                                                        /* two different pictures: compare the vectors that refer to the same one */
        const int swap = rp[0] != rq[0];
        return FAR(vp[0], vq[swap ? 1 : 0]) || FAR(vp[1], vq[swap ? 0 : 1]);
    }
    /* both vectors of both blocks refer to one picture: either pairing may match */
    return (FAR(vp[0], vq[0]) || FAR(vp[1], vq[1])) && (FAR(vp[0], vq[1]) || FAR(vp[1], vq[0]));
#undef FAR
}

/* 8.7.2.5.3 decisions + 8.7.2.5.7 luma sample filtering for one 4-sample segment.  s points at q0 of line 0; `across` is the step from p to q,
 * `along` the step from one line to the next. */
static void luma_segment(pix *s, ptrdiff_t across, ptrdiff_t along, int bs, int qp_l, int bit_depth, int beta_off2, int tc_off2)
{
    const int maxv = (1 << bit_depth) - 1;
    const int qb = d2_clip(0, 51, qp_l + (beta_off2 << 1)), beta = kBetaPrime[qb] * (1 << (bit_depth - 8));
    const int qt = d2_clip(0, 53, qp_l + 2 * (bs - 1) + (tc_off2 << 1)), tc = kTcPrime[qt] * (1 << (bit_depth - 8));
#define P(i, k) ((int)s[(ptrdiff_t)(k) * along - (ptrdiff_t)((i) + 1) * across])
#define Q(i, k) ((int)s[(ptrdiff_t)(k) * along + (ptrdiff_t)(i) * across])
    const int dp0 = d2_abs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = d2_abs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
    const int dq0 = d2_abs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = d2_abs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
    const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
    if (d >= beta) return;                                                        /* dE = 0 */
    /* 8.7.2.5.6 decision process for a luma sample, lines 0 and 3 */
    const int dsam0 = 2 * dpq0 < (beta >> 2) && d2_abs(P(3, 0) - P(0, 0)) + d2_abs(Q(0, 0) - Q(3, 0)) < (beta >> 3) && d2_abs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
    const int dsam3 = 2 * dpq3 < (beta >> 2) && d2_abs(P(3, 3) - P(0, 3)) + d2_abs(Q(0, 3) - Q(3, 3)) < (beta >> 3) && d2_abs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
    const int dE = (dsam0 && dsam3) ? 2 : 1;
    const int dEp = dp < ((beta + (beta >> 1)) >> 3), dEq = dq < ((beta + (beta >> 1)) >> 3);
    for (int k = 0; k < 4; k++) {
        const int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k), q0 = Q(0, k), q1 = Q(1, k), q2 = Q(2, k), q3 = Q(3, k);
        pix *row = s + (ptrdiff_t)k * along;
        if (dE == 2) {
            row[-1 * across] = (pix)d2_clip(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            row[-2 * across] = (pix)d2_clip(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
            row[-3 * across] = (pix)d2_clip(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            row[0] = (pix)d2_clip(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            row[1 * across] = (pix)d2_clip(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
            row[2 * across] = (pix)d2_clip(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
        } else {
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (d2_abs(delta) >= tc * 10) continue;
            delta = d2_clip(-tc, tc, delta);
            row[-1 * across] = (pix)d2_clip(0, maxv, p0 + delta);
            row[0] = (pix)d2_clip(0, maxv, q0 - delta);
            if (dEp) row[-2 * across] = (pix)d2_clip(0, maxv, p1 + d2_clip(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
            if (dEq) row[1 * across] = (pix)d2_clip(0, maxv, q1 + d2_clip(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
        }
    }
#undef P
#undef Q
}

/* 8.7.2.5.5 + 8.7.2.5.8: chroma, bS == 2 only, 4 chroma lines (= one 8-luma-sample... the caller hands segments of 2 chroma samples per 4 luma) */
static void chroma_segment(pix *s, ptrdiff_t across, ptrdiff_t along, int lines, int qp_avg, int c_off, int bit_depth, int tc_off2)
{
    const int maxv = (1 << bit_depth) - 1;
    const int qpc = d2_chroma_qp(qp_avg + c_off);
    const int qt = d2_clip(0, 53, qpc + 2 + (tc_off2 << 1)), tc = kTcPrime[qt] * (1 << (bit_depth - 8));
    for (int k = 0; k < lines; k++) {
        pix *row = s + (ptrdiff_t)k * along;
        const int p0 = row[-1 * across], p1 = row[-2 * across], q0 = row[0], q1 = row[1 * across];
        const int delta = d2_clip(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
        row[-1 * across] = (pix)d2_clip(0, maxv, p0 + delta);
        row[0] = (pix)d2_clip(0, maxv, q0 - delta);
    }
}

void d2_deblock_picture(const d2_picture_info *pi, pix *y, pix *u, pix *v, int stride, int cstride)
{
    const int w = pi->w, h = pi->h;
    /* 8.7.2: all vertical edges of the picture first (their output is the input of the horizontal pass), luma and chroma */
    for (int dir = 0; dir < 2; dir++) {
        /* boundary strengths are derived from syntax only, so they can be computed edge by edge while filtering */
        for (int ye = 0; ye < h; ye += dir ? 8 : 4)
            for (int xe = 0; xe < w; xe += dir ? 4 : 8) {
                const int bs = edge_bs(pi, dir, xe, ye);
                if (!bs) continue;
                const int xp = dir ? xe : xe - 1, yp = dir ? ye - 1 : ye;
                const int qp_l = (rec_at(pi, xe, ye)->qp + rec_at(pi, xp, yp)->qp + 1) >> 1;
                luma_segment(y + (size_t)ye * stride + xe, dir ? stride : 1, dir ? 1 : stride, bs, qp_l, pi->bit_depth, pi->beta_offset_div2, pi->tc_offset_div2);
                /* chroma edges lie on the 8x8 CHROMA sample grid (16 luma samples) and are filtered where bS == 2 (8.7.2.5.? edge filtering process,
                 * "ChromaArrayType != 0 and bS == 2 and (((xQ >> 3) << 3) == xQ in chroma units)") */
                if (bs == 2 && !((dir ? ye : xe) & 15)) {
                    const int xc = xe >> 1, yc = ye >> 1;
                    chroma_segment(u + (size_t)yc * cstride + xc, dir ? cstride : 1, dir ? 1 : cstride, 2, qp_l, pi->cb_qp_offset, pi->bit_depth, pi->tc_offset_div2);
                    chroma_segment(v + (size_t)yc * cstride + xc, dir ? cstride : 1, dir ? 1 : cstride, 2, qp_l, pi->cr_qp_offset, pi->bit_depth, pi->tc_offset_div2);
                }
            }
    }
}

/* ------------------------------------------------------------------ 8.7.3 sample adaptive offset */
static void sao_plane(const d2_picture_info *pi, const pix *in, int istride, pix *out, int ostride, int c_idx)
{
    const int sh = c_idx ? 1 : 0, pw = pi->w >> sh, ph = pi->h >> sh, ctb = 32 >> sh, wc = (pi->w + 31) >> 5;
    const int maxv = (1 << pi->bit_depth) - 1, band_shift = pi->bit_depth - 5;
    static const signed char hpos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vpos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};      /* Table 8-13 */
    for (int yy = 0; yy < ph; yy++)
        for (int xx = 0; xx < pw; xx++) {
            const int rx = xx / ctb, ry = yy / ctb;
            const orc_sao_ctu *s = &pi->sao[(size_t)ry * wc + rx];
            const int type = s->type[c_idx ? 1 : 0], cur = in[(size_t)yy * istride + xx];
            int val = cur;
            if (type == 1) {                      /* band offset: bandTable[(k + sao_band_position) & 31] = k + 1, k = 0 .. 3 */
                const int k = ((cur >> band_shift) - s->band_pos[c_idx]) & 31;
                if (k < 4) val = d2_clip(0, maxv, cur + s->offset[c_idx][k]);
            } else if (type == 2) {               /* edge offset */
                const int cls = s->eo_class[c_idx ? 1 : 0];
                int edge = 2, usable = 1;
                for (int k = 0; k < 2 && usable; k++) {
                    const int xn = xx + hpos[cls][k], yn = yy + vpos[cls][k];
                    if (xn < 0 || yn < 0 || xn >= pw || yn >= ph) { usable = 0; break; }       /* outside the picture */
                    /* a neighbour in another slice / tile counts only if filtering across that boundary is allowed (8.7.3.? "the sample at location
                     * (xSik', ySjk') belongs to a different slice and ..." / loop_filter_across_tiles_enabled_flag) */
                    const int bq = band_of(pi, yy << sh), bn = band_of(pi, yn << sh);
                    if (bq != bn && !(pi->band_lf_across[bq] && pi->band_lf_across[bn])) { usable = 0; break; }
                    if (!pi->lf_across_tiles && tile_index(pi, xx << sh, yy << sh) != tile_index(pi, xn << sh, yn << sh)) { usable = 0; break; }
                    edge += d2_sign(cur - (int)in[(size_t)yn * istride + xn]);
                }
                if (usable) {
                    if (edge == 0 || edge == 1 || edge == 2) edge = edge == 2 ? 0 : edge + 1;
                    if (edge) val = d2_clip(0, maxv, cur + s->offset[c_idx][edge - 1]);         /* SaoOffsetVal[edgeIdx], edgeIdx 1 .. 4 */
                }
            }
            out[(size_t)yy * ostride + xx] = (pix)val;
        }
}

void d2_sao_picture(const d2_picture_info *pi, const pix *y, const pix *u, const pix *v, int stride, int cstride, pix *oy, pix *ou, pix *ov, int ostride, int ocstride)
{
    sao_plane(pi, y, stride, oy, ostride, 0);
    sao_plane(pi, u, cstride, ou, ocstride, 1);
    sao_plane(pi, v, cstride, ov, ocstride, 2);
}

/* ------------------------------------------------------------------ test hooks (tests/test_decoder_second_opinion.py): plain-picture forms */
static void plain_info(d2_picture_info *pi, int w, int h, int bit_depth, const orc_cu_rec *cu, const orc_sao_ctu *sao, int *row0, unsigned char *across)
{
    memset(pi, 0, sizeof *pi);
    pi->w = w; pi->h = h; pi->bit_depth = bit_depth; pi->cu = cu; pi->sao = sao;
    pi->n_bands = 1; row0[0] = 0; row0[1] = (h + 31) >> 5; pi->band_row0 = row0; across[0] = 1; pi->band_lf_across = across;
    pi->tile_cols = pi->tile_rows = 1; pi->lf_across_tiles = 1;
}
void orc_dec2_deblock(pix *y, pix *u, pix *v, int stride, int cstride, int w, int h, const orc_cu_rec *cu, int bit_depth)
{
    d2_picture_info pi; int row0[2]; unsigned char across[1];
    plain_info(&pi, w, h, bit_depth, cu, NULL, row0, across);
    d2_deblock_picture(&pi, y, u, v, stride, cstride);
}
void orc_dec2_sao(const pix *y, const pix *u, const pix *v, int stride, int cstride, pix *oy, pix *ou, pix *ov, int w, int h, int bit_depth, const orc_sao_ctu *sao)
{
    d2_picture_info pi; int row0[2]; unsigned char across[1];
    plain_info(&pi, w, h, bit_depth, NULL, sao, row0, across);
    d2_sao_picture(&pi, y, u, v, stride, cstride, oy, ou, ov, stride, cstride);
}
void orc_dec2_residual(const int16_t *lvl, int32_t *res, int log2n, int qp, int bit_depth, int dst4) { d2_residual(lvl, res, log2n, qp, bit_depth, dst4); }
void orc_dec2_intra_pred(const pix *ref, pix *dst, int log2n, int mode, int c_idx, int bit_depth, int strong) { d2_intra_pred(ref, dst, 1 << log2n, log2n, mode, c_idx, bit_depth, strong); }
int orc_dec2_chroma_qp(int qpi) { return d2_chroma_qp(qpi); }
