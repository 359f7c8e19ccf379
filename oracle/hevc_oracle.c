/* oracle/hevc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT (see hevc_oracle.h for scope and pinning).
 *
 * Scalar restatement of the per-CTU encode loop the reference obtains from libx265
 * (reference call sites: core/transcoder.py:398-412 operating point, :463 `-c:v`, :506 spawn).
 * Normative parts cite their H.265 clause; encoder-side choices are defined here and mirrored by
 * hevc_amd/csrc/kernels/ (HIP).  PARITY UNPINNED vs libx265 (no golden vectors exist; see header).
 */
#include "hevc_oracle.h"
#include <stdlib.h>
#include <string.h>

#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : (v) > (hi) ? (hi) : (v))
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int ilog2u(unsigned v) { int r = 0; while (v >>= 1) r++; return r; }

/* ------------------------------------------------------------------------------------------------
 * Transform matrices — H.265 8.6.4.2 (transMatrix) generated from the 32 unique cosine magnitudes.
 * ------------------------------------------------------------------------------------------------ */
static const int16_t kCos64[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                   61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
static int16_t g_mat32[32][32];
static int g_mat_ready;
static const int16_t kDst4[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

static void build_matrix(void)
{
    if (g_mat_ready) return;
    for (int k = 0; k < 32; k++)
        for (int n = 0; n < 32; n++) {
            int t = (k * (2 * n + 1)) % 128; /* angle in units of pi/64 */
            if (t > 64) t = 128 - t;
            g_mat32[k][n] = (int16_t)(t > 32 ? -kCos64[64 - t] : kCos64[t]);
        }
    g_mat_ready = 1;
}
void orc_transform_matrix(int16_t *out)
{
    build_matrix();
    memcpy(out, g_mat32, sizeof g_mat32);
}
static inline int mat(int log2n, int dst, int k, int n)
{
    if (dst) return kDst4[k][n];
    return g_mat32[k << (5 - log2n)][n];
}

/* Forward transform (encoder side; the conventional two-stage integer form):
 * stage 1 horizontal, shift log2N + bitDepth - 9; stage 2 vertical, shift log2N + 6. */
void orc_fwd_transform(const int16_t *res, int rstride, int16_t *coef, int log2n, int dst, int bit_depth)
{
    build_matrix();
    int n = 1 << log2n, s1 = log2n + bit_depth - 9, s2 = log2n + 6;
    int32_t tmp[32 * 32];
    for (int y = 0; y < n; y++)
        for (int u = 0; u < n; u++) {
            int64_t acc = 0;
            for (int x = 0; x < n; x++) acc += mat(log2n, dst, u, x) * res[y * rstride + x];
            tmp[y * n + u] = (int32_t)(s1 > 0 ? (acc + (1 << (s1 - 1))) >> s1 : acc);
        }
    for (int v = 0; v < n; v++)
        for (int u = 0; u < n; u++) {
            int64_t acc = 0;
            for (int y = 0; y < n; y++) acc += (int64_t)mat(log2n, dst, v, y) * tmp[y * n + u];
            acc = (acc + (1 << (s2 - 1))) >> s2;
            coef[v * n + u] = (int16_t)CLIP3(-32768, 32767, acc);
        }
}

/* Inverse transform — H.265 8.6.4.2: columns first (shift 7, clip to 16 bit), then rows (shift 20 - bitDepth). */
void orc_inv_transform(const int16_t *coef, int16_t *res, int rstride, int log2n, int dst, int bit_depth)
{
    build_matrix();
    int n = 1 << log2n, s2 = 20 - bit_depth;
    int32_t g[32 * 32];
    for (int x = 0; x < n; x++)
        for (int y = 0; y < n; y++) {
            int64_t acc = 0;
            for (int j = 0; j < n; j++) acc += mat(log2n, dst, j, y) * coef[j * n + x];
            acc = (acc + 64) >> 7;
            g[y * n + x] = (int32_t)CLIP3(-32768, 32767, acc);
        }
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int64_t acc = 0;
            for (int j = 0; j < n; j++) acc += (int64_t)mat(log2n, dst, j, x) * g[y * n + j];
            res[y * rstride + x] = (int16_t)((acc + (1 << (s2 - 1))) >> s2);
        }
}

static const int kQuantScale[6] = {26214, 23302, 20560, 18396, 16384, 14564};
static const int kLevelScale[6] = {40, 45, 51, 57, 64, 72};

/* Quantisation (encoder side): dead-zone rounding 171/512 (intra) or 85/512 (inter). Returns #nonzero. */
int orc_quant(const int16_t *coef, int16_t *lvl, int log2n, int qp, int bit_depth, int intra)
{
    int q = qp + 6 * (bit_depth - 8);
    int qbits = 14 + q / 6 + (15 - bit_depth - log2n);
    int64_t add = (int64_t)(intra ? 171 : 85) << (qbits - 9);
    int scale = kQuantScale[q % 6], nnz = 0, n2 = 1 << (2 * log2n);
    for (int i = 0; i < n2; i++) {
        int c = coef[i];
        int64_t a = ((int64_t)iabs(c) * scale + add) >> qbits;
        if (a > 32767) a = 32767;
        lvl[i] = (int16_t)(c < 0 ? -a : a);
        nnz += a != 0;
    }
    return nnz;
}

/* Scaling (dequantisation) — H.265 8.6.4.1 with flat m = 16. */
void orc_dequant(const int16_t *lvl, int16_t *coef, int log2n, int qp, int bit_depth)
{
    int q = qp + 6 * (bit_depth - 8);
    int bd_shift = bit_depth + log2n - 5;
    int64_t scale = (int64_t)16 * kLevelScale[q % 6] << (q / 6);
    int n2 = 1 << (2 * log2n);
    for (int i = 0; i < n2; i++) {
        int64_t v = (lvl[i] * scale + ((int64_t)1 << (bd_shift - 1))) >> bd_shift;
        coef[i] = (int16_t)CLIP3(-32768, 32767, v);
    }
}

/* Chroma QP mapping, ChromaArrayType == 1 — H.265 Table 8-10 (offsets 0). Input/outputs are syntax QPs. */
int orc_chroma_qp(int qp_y)
{
    static const int8_t t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
    int qpi = CLIP3(-12, 57, qp_y);
    if (qpi < 30) return qpi;
    if (qpi > 43) return qpi - 6;
    return t[qpi - 30];
}

/* ------------------------------------------------------------------------------------------------
 * Intra prediction — H.265 8.4.4.2
 * ------------------------------------------------------------------------------------------------ */
static inline int zorder6(int bx, int by) /* 3 bits each -> interleaved */
{
    int z = 0;
    for (int i = 0; i < 3; i++) z |= ((bx >> i) & 1) << (2 * i) | ((by >> i) & 1) << (2 * i + 1);
    return z;
}
/* z-scan address of the 4x4 luma unit containing (x,y) — 6.4.1 / 6.5.2 at min-TB granularity */
static inline int zaddr(int x, int y, int pic_w)
{
    int wc = (pic_w + ORC_CTU - 1) >> ORC_CTU_LOG2;
    int ctu = (y >> ORC_CTU_LOG2) * wc + (x >> ORC_CTU_LOG2);
    return (ctu << 6) | zorder6((x & (ORC_CTU - 1)) >> 2, (y & (ORC_CTU - 1)) >> 2);
}

/* 8.4.4.2.2 reference sample availability + substitution.
 * ref[0] = p[-1][2N-1] ... ref[2N-1] = p[-1][0], ref[2N] = p[-1][-1], ref[2N+1+x] = p[x][-1].
 * (x0,y0), pic_w/h in units of the component's samples. avail_map is unused (z-order rule is applied). */
void orc_intra_build_ref(const pix *rec, int stride, int x0, int y0, int log2n, int pic_w, int pic_h,
                         const uint8_t *avail_map, int map_stride, int c_idx, int bit_depth, pix *ref)
{
    (void)avail_map; (void)map_stride;
    orc_intra_build_ref_tiles(rec, stride, x0, y0, log2n, pic_w, pic_h, c_idx, bit_depth, 1, 1, ref);
}

static int same_tile(int xa, int ya, int xb, int yb, int lw, int lh, int tc, int tr)   /* luma positions */
{
    if (tc <= 1 && tr <= 1) return 1;
    int wc = (lw + ORC_CTU - 1) >> ORC_CTU_LOG2, hc = (lh + ORC_CTU - 1) >> ORC_CTU_LOG2;
    if (tc < 1) tc = 1;
    if (tr < 1) tr = 1;
    return orc_tile_of(xa >> ORC_CTU_LOG2, tc, wc) == orc_tile_of(xb >> ORC_CTU_LOG2, tc, wc) &&
           orc_tile_of(ya >> ORC_CTU_LOG2, tr, hc) == orc_tile_of(yb >> ORC_CTU_LOG2, tr, hc);
}

void orc_intra_build_ref_tiles(const pix *rec, int stride, int x0, int y0, int log2n, int pic_w, int pic_h,
                               int c_idx, int bit_depth, int tile_cols, int tile_rows, pix *ref)
{
    int n = 1 << log2n, s = c_idx ? 1 : 0, total = 4 * n + 1;
    int lw = pic_w << s, lh = pic_h << s;
    int zc = zaddr(x0 << s, y0 << s, lw);
    uint8_t av[4 * 32 + 1];
    for (int i = 0; i < total; i++) {
        int xn, yn;
        if (i < 2 * n) { xn = x0 - 1; yn = y0 + 2 * n - 1 - i; }
        else if (i == 2 * n) { xn = x0 - 1; yn = y0 - 1; }
        else { xn = x0 + (i - 2 * n - 1); yn = y0 - 1; }
        int ok = xn >= 0 && yn >= 0 && xn < pic_w && yn < pic_h && zaddr(xn << s, yn << s, lw) < zc &&
                 same_tile(xn << s, yn << s, x0 << s, y0 << s, lw, lh, tile_cols, tile_rows);
        av[i] = (uint8_t)ok;
        ref[i] = ok ? rec[yn * stride + xn] : 0;
    }
    int first = -1;
    for (int i = 0; i < total; i++) if (av[i]) { first = i; break; }
    if (first < 0) {
        for (int i = 0; i < total; i++) ref[i] = (pix)(1 << (bit_depth - 1));
        return;
    }
    if (!av[0]) ref[0] = ref[first];
    for (int i = 1; i < total; i++) if (!av[i]) ref[i] = ref[i - 1];
}

/* 8.4.4.2.3 filtering of neighbouring samples. Writes filt[] (copy when the filter is off). */
void orc_intra_filter_ref(const pix *ref, pix *filt, int log2n, int mode, int c_idx, int bit_depth, int strong)
{
    int n = 1 << log2n, total = 4 * n + 1;
    int on = 0;
    if (c_idx == 0 && mode != 1 && n != 4) {
        int d1 = iabs(mode - 26), d2 = iabs(mode - 10);
        int dist = d1 < d2 ? d1 : d2;
        int thr = n == 8 ? 7 : n == 16 ? 1 : 0;
        on = dist > thr;
    }
    if (!on) { memcpy(filt, ref, total * sizeof(pix)); return; }
    if (strong && n == 32) {
        int thr = 1 << (bit_depth - 5);
        int c = ref[64], a = ref[0], b = ref[128];
        if (iabs(c + b - 2 * ref[96]) < thr && iabs(c + a - 2 * ref[32]) < thr) {
            filt[0] = ref[0]; filt[64] = ref[64]; filt[128] = ref[128];
            for (int i = 1; i < 64; i++) filt[i] = (pix)((i * c + (64 - i) * a + 32) >> 6);
            for (int x = 0; x < 63; x++) filt[65 + x] = (pix)(((63 - x) * c + (x + 1) * b + 32) >> 6);
            return;
        }
    }
    filt[0] = ref[0]; filt[total - 1] = ref[total - 1];
    for (int i = 1; i < total - 1; i++) filt[i] = (pix)((ref[i - 1] + 2 * ref[i] + ref[i + 1] + 2) >> 2);
}

static const int8_t kIntraAngle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                       -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
static const int16_t kInvAngle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};

/* 8.4.4.2.4 planar, .5 DC, .6 angular. `ref` is the (already filtered where applicable) 4N+1 array. */
void orc_intra_pred(const pix *ref, pix *dst, int dstride, int log2n, int mode, int c_idx, int bit_depth)
{
    int n = 1 << log2n, maxv = (1 << bit_depth) - 1;
    const pix *left = ref + 2 * n - 1; /* left[-y] = p[-1][y] */
    const pix *top = ref + 2 * n + 1;  /* top[x]  = p[x][-1] ; top[-1] = corner */
#define PL(y) left[-(y)]
#define PT(x) top[(x)]
    if (mode == 0) {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                dst[y * dstride + x] = (pix)(((n - 1 - x) * PL(y) + (x + 1) * PT(n) + (n - 1 - y) * PT(x) + (y + 1) * PL(n) + n) >> (log2n + 1));
        return;
    }
    if (mode == 1) {
        int sum = n;
        for (int i = 0; i < n; i++) sum += PT(i) + PL(i);
        int dc = sum >> (log2n + 1);
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) dst[y * dstride + x] = (pix)dc;
        if (c_idx == 0 && n < 32) {
            dst[0] = (pix)((PL(0) + 2 * dc + PT(0) + 2) >> 2);
            for (int x = 1; x < n; x++) dst[x] = (pix)((PT(x) + 3 * dc + 2) >> 2);
            for (int y = 1; y < n; y++) dst[y * dstride] = (pix)((PL(y) + 3 * dc + 2) >> 2);
        }
        return;
    }
    int angle = kIntraAngle[mode];
    int vertical = mode >= 18;
    pix buf[3 * 32 + 1];
    pix *r = buf + 32; /* r[-n .. 2n] */
    /* main reference: for vertical modes the top row, else the left column; side = the other one */
    for (int i = 0; i <= n; i++) r[i] = vertical ? top[i - 1] : left[-(i - 1)];
    if (angle < 0) {
        int last = (n * angle) >> 5;
        if (last < -1) {
            int inv = kInvAngle[mode - 11];
            for (int i = -1; i >= last; i--) {
                int k = -1 + ((i * inv + 128) >> 8);
                r[i] = vertical ? left[-k] : top[k];
            }
        }
    } else {
        for (int i = n + 1; i <= 2 * n; i++) r[i] = vertical ? top[i - 1] : left[-(i - 1)];
    }
    for (int a = 0; a < n; a++) {       /* a: along the prediction direction (y for vertical, x for horizontal) */
        int idx = ((a + 1) * angle) >> 5, f = ((a + 1) * angle) & 31;
        for (int b = 0; b < n; b++) {
            int v = f ? ((32 - f) * r[b + idx + 1] + f * r[b + idx + 2] + 16) >> 5 : r[b + idx + 1];
            if (vertical) dst[a * dstride + b] = (pix)v; else dst[b * dstride + a] = (pix)v;
        }
    }
    if (angle == 0 && c_idx == 0 && n < 32) {   /* modes 26 / 10 edge filter */
        int corner = top[-1];
        if (vertical)
            for (int y = 0; y < n; y++) { int v = PT(0) + ((PL(y) - corner) >> 1); dst[y * dstride] = (pix)CLIP3(0, maxv, v); }
        else
            for (int x = 0; x < n; x++) { int v = PL(0) + ((PT(x) - corner) >> 1); dst[x] = (pix)CLIP3(0, maxv, v); }
    }
#undef PL
#undef PT
}

/* ------------------------------------------------------------------------------------------------
 * Inter prediction sample interpolation — H.265 8.5.3.3.3 (+ default weighted prediction 8.5.3.3.4.2, uni-pred)
 * ------------------------------------------------------------------------------------------------ */
static const int8_t kLumaTap[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t kChromaTap[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

void orc_interp_luma(const pix *ref, int rstride, int x, int y, int mvx, int mvy, int w, int h, int bit_depth,
                     pix *dst, int dstride)
{
    int fx = mvx & 3, fy = mvy & 3, xi = x + (mvx >> 2), yi = y + (mvy >> 2);
    int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;
    int off = 1 << (shift3 - 1);
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const pix *p = ref + (yi + j) * rstride + xi + i;
            int v;
            if (!fx && !fy) v = p[0] << shift3;
            else if (!fy) { int a = 0; for (int k = 0; k < 8; k++) a += kLumaTap[fx][k] * p[k - 3]; v = a >> shift1; }
            else if (!fx) { int a = 0; for (int k = 0; k < 8; k++) a += kLumaTap[fy][k] * p[(k - 3) * rstride]; v = a >> shift1; }
            else {
                int a = 0;
                for (int r = 0; r < 8; r++) {
                    int t = 0;
                    for (int k = 0; k < 8; k++) t += kLumaTap[fx][k] * p[(r - 3) * rstride + k - 3];
                    a += kLumaTap[fy][r] * (t >> shift1);
                }
                v = a >> 6;
            }
            v = (v + off) >> shift3;
            dst[j * dstride + i] = (pix)CLIP3(0, maxv, v);
        }
}

void orc_interp_chroma(const pix *ref, int rstride, int xc, int yc, int mvx, int mvy, int wc, int hc, int bit_depth,
                       pix *dst, int dstride)
{
    int fx = mvx & 7, fy = mvy & 7, xi = xc + (mvx >> 3), yi = yc + (mvy >> 3);
    int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, shift3 = 14 - bit_depth, maxv = (1 << bit_depth) - 1;
    int off = 1 << (shift3 - 1);
    for (int j = 0; j < hc; j++)
        for (int i = 0; i < wc; i++) {
            const pix *p = ref + (yi + j) * rstride + xi + i;
            int v;
            if (!fx && !fy) v = p[0] << shift3;
            else if (!fy) { int a = 0; for (int k = 0; k < 4; k++) a += kChromaTap[fx][k] * p[k - 1]; v = a >> shift1; }
            else if (!fx) { int a = 0; for (int k = 0; k < 4; k++) a += kChromaTap[fy][k] * p[(k - 1) * rstride]; v = a >> shift1; }
            else {
                int a = 0;
                for (int r = 0; r < 4; r++) {
                    int t = 0;
                    for (int k = 0; k < 4; k++) t += kChromaTap[fx][k] * p[(r - 1) * rstride + k - 1];
                    a += kChromaTap[fy][r] * (t >> shift1);
                }
                v = a >> 6;
            }
            v = (v + off) >> shift3;
            dst[j * dstride + i] = (pix)CLIP3(0, maxv, v);
        }
}

int orc_sad(const pix *a, int as, const pix *b, int bs, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) s += iabs(a[y * as + x] - b[y * bs + x]);
    return s;
}

static int hadamard8(const pix *a, int as, const pix *b, int bs)
{
    int m[8][8];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) m[y][x] = a[y * as + x] - b[y * bs + x];
    for (int y = 0; y < 8; y++)
        for (int st = 1; st < 8; st <<= 1)
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[y][i], q = m[y][i + st]; m[y][i] = p + q; m[y][i + st] = p - q; }
    for (int x = 0; x < 8; x++)
        for (int st = 1; st < 8; st <<= 1)
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[i][x], q = m[i + st][x]; m[i][x] = p + q; m[i + st][x] = p - q; }
    int s = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) s += iabs(m[y][x]);
    return (s + 2) >> 2;
}
static int hadamard4(const pix *a, int as, const pix *b, int bs)
{
    int m[4][4];
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) m[y][x] = a[y * as + x] - b[y * bs + x];
    for (int y = 0; y < 4; y++)
        for (int st = 1; st < 4; st <<= 1)
            for (int i = 0; i < 4; i++)
                if (!(i & st)) { int p = m[y][i], q = m[y][i + st]; m[y][i] = p + q; m[y][i + st] = p - q; }
    for (int x = 0; x < 4; x++)
        for (int st = 1; st < 4; st <<= 1)
            for (int i = 0; i < 4; i++)
                if (!(i & st)) { int p = m[i][x], q = m[i + st][x]; m[i][x] = p + q; m[i + st][x] = p - q; }
    int s = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) s += iabs(m[y][x]);
    return (s + 1) >> 1;
}
/* SATD: sum over 8x8 Hadamard tiles ((sum|c|+2)>>2 each); 4x4 blocks use the 4x4 Hadamard ((sum+1)>>1). */
int orc_satd(const pix *a, int as, const pix *b, int bs, int w, int h)
{
    int s = 0;
    if (w >= 8 && h >= 8) {
        for (int y = 0; y < h; y += 8)
            for (int x = 0; x < w; x += 8) s += hadamard8(a + y * as + x, as, b + y * bs + x, bs);
    } else {
        for (int y = 0; y < h; y += 4)
            for (int x = 0; x < w; x += 4) s += hadamard4(a + y * as + x, as, b + y * bs + x, bs);
    }
    return s;
}

/* bins of one mvd component: greater0, greater1, EG1(|d|-2), sign (H.265 7.3.8.9 / 9.3.3) */
int orc_mvd_bits(int d)
{
    int a = iabs(d);
    if (a == 0) return 1;
    if (a == 1) return 3;
    return 3 + 2 * ilog2u((unsigned)a);
}

void orc_pad_plane(pix *p, int stride, int w, int h, int pad)
{
    for (int y = 0; y < h; y++) {
        pix *row = p + y * stride;
        for (int x = 1; x <= pad; x++) { row[-x] = row[0]; row[w - 1 + x] = row[w - 1]; }
    }
    for (int y = 1; y <= pad; y++) {
        memcpy(p - y * stride - pad, p - pad, (w + 2 * pad) * sizeof(pix));
        memcpy(p + (h - 1 + y) * stride - pad, p + (h - 1) * stride - pad, (w + 2 * pad) * sizeof(pix));
    }
}

#define ORC_R_LEVEL(a) ((a) == 1 ? 33 : (a) == 2 ? 50 : 53 + 27 * ilog2u((unsigned)((a) - 1)))
#define ORC_R_SB 143
#define ORC_R_TU 30
#define ORC_R_INTER_CU 80u
#define ORC_R_INTRA_CU 128u

/* ------------------------------------------------------------------------------------------------
 * Residual coding of one TU (shared by intra and inter): returns cbf, writes levels + reconstruction.
 * ------------------------------------------------------------------------------------------------ */
/* RD zero-out of 4x4 coefficient groups ("RDOQ-lite"), inter TUs only, cg_lam_q4 > 0: a group of levels is dropped when the
 * distortion it removes does not pay for its bits.  Everything is decided in the coefficient domain from what the quantiser already holds: with
 * c the transform coefficient and r its reconstruction (orc_dequant), dropping the group adds  D = sum r (2c - r)  (= sum c^2 - (c - r)^2) of
 * squared coefficient error, which is  D >> 2 (15 - bitDepth - log2n)  of squared sample error (the transform's gain), and saves the rate model's
 * bits = sum level bits + one sub-block (ORC_R_*, 1/16 bit).  Drop  <=>  16 D < ((cg_lam_q4 * bits) >> 4) << 2 (15 - bitDepth - log2n).
 * No context modelling and no last-position search: every group decides alone, which is what makes it one short phase on the GPU. */
static int cg_zero_out(const int16_t *coef, int16_t *lvl, int log2n, int qp, int bit_depth, int cg_lam_q4)
{
    int n = 1 << log2n, tsh = 2 * (15 - bit_depth - log2n), nnz = 0;
    int16_t deq[32 * 32];
    orc_dequant(lvl, deq, log2n, qp, bit_depth);
    for (int sy = 0; sy < n; sy += 4)
        for (int sx = 0; sx < n; sx += 4) {
            int64_t d = 0;
            int bits = 0, cnt = 0;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) {
                    int i = (sy + y) * n + sx + x, a = iabs(lvl[i]);
                    if (!a) continue;
                    cnt++;
                    bits += ORC_R_LEVEL(a);
                    d += (int64_t)deq[i] * (2 * (int64_t)coef[i] - deq[i]);
                }
            if (!cnt) continue;
            bits += ORC_R_SB;
            if (16 * d < ((((int64_t)cg_lam_q4 * bits) >> 4) << tsh)) {
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++) lvl[(sy + y) * n + sx + x] = 0;
            } else nnz += cnt;
        }
    return nnz;
}

static int code_tu(const pix *src, int sstride, const pix *pred, int pstride, pix *rec, int rstride,
                   int16_t *coef_out, int cstride, int log2n, int qp, int bit_depth, int intra, int dst,
                   int64_t *sse_out, int *bits_q4_out, int cg_lam_q4)
{
    int n = 1 << log2n, maxv = (1 << bit_depth) - 1;
    int16_t res[32 * 32], coef[32 * 32], lvl[32 * 32], rc[32 * 32];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) res[y * n + x] = (int16_t)(src[y * sstride + x] - pred[y * pstride + x]);
    orc_fwd_transform(res, n, coef, log2n, dst, bit_depth);
    int nnz = orc_quant(coef, lvl, log2n, qp, bit_depth, intra);
    if (nnz && cg_lam_q4 > 0 && !intra) nnz = cg_zero_out(coef, lvl, log2n, qp, bit_depth, cg_lam_q4);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) coef_out[y * cstride + x] = lvl[y * n + x];
    if (nnz) {
        orc_dequant(lvl, coef, log2n, qp, bit_depth);
        orc_inv_transform(coef, rc, n, log2n, dst, bit_depth);
    } else {
        memset(rc, 0, sizeof(int16_t) * n * n);
    }
    int64_t sse = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int v = pred[y * pstride + x] + rc[y * n + x];
            v = CLIP3(0, maxv, v);
            rec[y * rstride + x] = (pix)v;
            int d = src[y * sstride + x] - v;
            sse += d * d;
        }
    if (sse_out) *sse_out = sse;
    if (bits_q4_out) {
        int bits = 0;
        for (int sy = 0; sy < n; sy += 4)
            for (int sx = 0; sx < n; sx += 4) {
                int any = 0;
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++) {
                        int a = iabs(lvl[(sy + y) * n + sx + x]);
                        if (!a) continue;
                        any = 1;
                        bits += ORC_R_LEVEL(a);
                    }
                if (any) bits += ORC_R_SB;
            }
        if (bits) bits += ORC_R_TU;
        *bits_q4_out = bits;
    }
    return nnz != 0;
}

/* Inter TU with the RD zero-out (prm->rdo_zero): code it, then keep the levels only if they pay for themselves --
 * SSE_zero << 4 <= (SSE_coded << 4) + (lambda * bits >> 4) turns the TU into an all-zero one (reconstruction = prediction). */
static int code_tu_inter(const pix *src, int sstride, const pix *pred, int pstride, pix *rec, int rstride, int16_t *coef_out, int cstride,
                         int log2n, int qp, int bit_depth, const orc_params *prm)
{
    int64_t sse;
    int bits;
    int cbf = code_tu(src, sstride, pred, pstride, rec, rstride, coef_out, cstride, log2n, qp, bit_depth, 0, 0, &sse, &bits,
                      prm->rdo_cg > 0 ? (int)(((int64_t)prm->lambda_q4 * prm->rdo_cg) >> 1) : 0);
    if (!cbf || !prm->rdo_zero) return cbf;
    int n = 1 << log2n;
    int64_t sse0 = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) { int d = src[y * sstride + x] - pred[y * pstride + x]; sse0 += d * d; }
    if (((uint64_t)sse0 << 4) > ((uint64_t)sse << 4) + (((uint64_t)prm->lambda_q4 * (uint64_t)bits) >> 4)) return 1;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) { coef_out[y * cstride + x] = 0; rec[y * rstride + x] = pred[y * pstride + x]; }
    return 0;
}

/* Rate model, 1/16 bit units, fitted to this build's CABAC on I and P pictures, 8 and 10 bit, QP 22..34 (within 5 %; the first model
 * undercounted sparse P residuals by 45 % and overcounted inter CU headers 7x): a level costs ORC_R_LEVEL, every 4x4 sub-block with a level
 * ORC_R_SB (significance map + last position share), every TU with a level ORC_R_TU, an inter CU ORC_R_INTER_CU (mostly merge / skip),
 * an intra CU ORC_R_INTRA_CU. */
static int cu_cbf_count(const orc_cu_rec *r)
{
    int n = ((r->flags & ORC_F_CBF_CB) != 0) + ((r->flags & ORC_F_CBF_CR) != 0);
    if (r->flags & ORC_F_NXN) { for (int k = 0; k < 4; k++) n += (r->cbf_y4 >> k) & 1; }
    else n += (r->flags & ORC_F_CBF_Y) != 0;
    return n;
}

/* Picture-level rate estimate in 1/16 bit (drives the rate controller; mirrored by the kernels' `est` accumulators):
 * every 4x4 sub-block with a non-zero level costs 24 + sum f(|level|) (f as in code_tu), every CU a header of
 * 8 bits (intra) or 6 + mvd bits relative to the CTU's search centre (inter). */
static uint64_t estimate_bits(const orc_cu_rec *cu, const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v, int w, int h,
                              const int16_t *centers)
{
    uint64_t est = 0;
    int w8 = w >> 3, wc = (w + ORC_CTU - 1) / ORC_CTU;
    for (int by = 0; by < h >> 3; by++)
        for (int bx = 0; bx < w8; bx++) {
            const orc_cu_rec *r = &cu[by * w8 + bx];
            int mask = (1 << r->log2_size) - 1;
            if (((bx * 8) & mask) || ((by * 8) & mask)) continue;
            if (r->flags & ORC_F_INTER) {
                int ctu = (by * 8 / ORC_CTU) * wc + bx * 8 / ORC_CTU;
                int sx = centers ? centers[2 * ctu] : 0, sy = centers ? centers[2 * ctu + 1] : 0;
                (void)sx; (void)sy;
                est += ORC_R_INTER_CU;
            } else est += ORC_R_INTRA_CU;
            est += (unsigned)(ORC_R_TU * cu_cbf_count(r));
        }
    for (int pl = 0; pl < 3; pl++) {
        const int16_t *c = pl == 0 ? coef_y : pl == 1 ? coef_u : coef_v;
        int pw = pl ? w / 2 : w, ph = pl ? h / 2 : h;
        for (int y = 0; y < ph; y += 4)
            for (int x = 0; x < pw; x += 4) {
                int bits = 0, any = 0;
                for (int j = 0; j < 4; j++)
                    for (int i = 0; i < 4; i++) {
                        int a = iabs(c[(y + j) * pw + x + i]);
                        if (!a) continue;
                        any = 1;
                        bits += ORC_R_LEVEL(a);
                    }
                if (any) est += (unsigned)(bits + ORC_R_SB);
            }
    }
    return est;
}

/* the same estimate restricted to CTU (cx, cy) */
static uint64_t estimate_bits_ctu(const orc_cu_rec *cu, const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v, int w, int h,
                                  const int16_t *centers, int cx, int cy)
{
    uint64_t est = 0;
    int w8 = w >> 3, wc = (w + ORC_CTU - 1) / ORC_CTU, ctu = cy * wc + cx;
    int x0 = cx * ORC_CTU, y0 = cy * ORC_CTU, x1 = x0 + ORC_CTU < w ? x0 + ORC_CTU : w, y1 = y0 + ORC_CTU < h ? y0 + ORC_CTU : h;
    for (int by = y0 >> 3; by < y1 >> 3; by++)
        for (int bx = x0 >> 3; bx < x1 >> 3; bx++) {
            const orc_cu_rec *r = &cu[by * w8 + bx];
            int mask = (1 << r->log2_size) - 1;
            if (((bx * 8) & mask) || ((by * 8) & mask)) continue;
            if (r->flags & ORC_F_INTER) {
                int sx = centers ? centers[2 * ctu] : 0, sy = centers ? centers[2 * ctu + 1] : 0;
                (void)sx; (void)sy;
                est += ORC_R_INTER_CU;
            } else est += ORC_R_INTRA_CU;
            est += (unsigned)(ORC_R_TU * cu_cbf_count(r));
        }
    for (int pl = 0; pl < 3; pl++) {
        const int16_t *c = pl == 0 ? coef_y : pl == 1 ? coef_u : coef_v;
        int sh = pl ? 1 : 0, pw = w >> sh;
        for (int y = y0 >> sh; y < y1 >> sh; y += 4)
            for (int x = x0 >> sh; x < x1 >> sh; x += 4) {
                int bits = 0, any = 0;
                for (int j = 0; j < 4; j++)
                    for (int i = 0; i < 4; i++) {
                        int a = iabs(c[(y + j) * pw + x + i]);
                        if (!a) continue;
                        any = 1;
                        bits += ORC_R_LEVEL(a);
                    }
                if (any) est += (unsigned)(bits + ORC_R_SB);
            }
    }
    return est;
}

/* ================================================================================================
 * K1 + K3 : inter frame
 * ================================================================================================ */
static const int8_t kFracOff[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};

/* SAD of an 8x8 block in the integer search: on the 8 most significant bits of every sample, scaled back to the sample range.  At
 * 8 bits this is the plain SAD; Main10 then searches with the same packed quad-SAD instruction and half the LDS traffic. */
static uint32_t sad_msb8(const pix *a, int as, const pix *b, int bs, int sh)
{
    uint32_t s = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) s += (uint32_t)iabs((a[y * as + x] >> sh) - (b[y * bs + x] >> sh));
    return s << sh;
}

/* node n of the CTU quadtree: 0 = 32x32, 1..4 = 16x16 (z-order), 5..20 = 8x8 (z-order inside each 16x16) */
static void node_geom(int node, int *x, int *y, int *log2n)
{
    if (node == 0) { *x = 0; *y = 0; *log2n = 5; return; }
    if (node < 5) { int q = node - 1; *x = (q & 1) * 16; *y = (q >> 1) * 16; *log2n = 4; return; }
    int q = (node - 5) >> 2, s = (node - 5) & 3;
    *x = (q & 1) * 16 + (s & 1) * 8; *y = (q >> 1) * 16 + (s >> 1) * 8; *log2n = 3;
}

/* ---- search centres from the 1/4-size pictures ------------------------------------------------------------------
 * The integer search covers +-me_range around a per-CTU centre.  With pre_search the centre comes from a full search of
 * +-ORC_PRE_RANGE low-resolution samples (+-56 luma samples) of the CTU's 8x8 low-resolution block: cost = 4 * SAD + |dx| + |dy|
 * (a slight pull towards zero), ties -> first position in raster order; samples outside the low-resolution picture clamp. */
#define ORC_PRE_RANGE 14
void orc_lowres(const pix *src, int stride, int w, int h, int bit_depth, pix *dst)
{
    int lw = w >> 2, lh = h >> 2, sh = bit_depth - 8;
    for (int y = 0; y < lh; y++)
        for (int x = 0; x < lw; x++) {
            int s = 8 << sh;
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) s += src[(4 * y + j) * stride + 4 * x + i];
            dst[y * lw + x] = (pix)(s >> (4 + sh));
        }
}
void orc_pre_search(const pix *lsrc, const pix *lref, int lw, int lh, int16_t *centers) { orc_pre_search_cost(lsrc, lref, lw, lh, centers, NULL); }
/* the same, also returning per CTU the smallest SAD of its low-resolution block over the window (the session's B-picture probe sums them) */
void orc_pre_search_cost(const pix *lsrc, const pix *lref, int lw, int lh, int16_t *centers, uint32_t *costs)
{
    int wc = (lw + 7) >> 3, hc = (lh + 7) >> 3, R = ORC_PRE_RANGE, span = 2 * R + 1;
    for (int cy = 0; cy < hc; cy++)
        for (int cx = 0; cx < wc; cx++) {
            int bw = lw - 8 * cx < 8 ? lw - 8 * cx : 8, bh = lh - 8 * cy < 8 ? lh - 8 * cy : 8;
            uint64_t best = ~0ull;
            uint32_t sad0 = 0;
            for (int dy = -R; dy <= R; dy++)
                for (int dx = -R; dx <= R; dx++) {
                    uint32_t sad = 0;
                    for (int y = 0; y < bh; y++)
                        for (int x = 0; x < bw; x++) {
                            int rx = CLIP3(0, lw - 1, 8 * cx + x + dx), ry = CLIP3(0, lh - 1, 8 * cy + y + dy);
                            sad += (uint32_t)iabs(lsrc[(8 * cy + y) * lw + 8 * cx + x] - lref[ry * lw + rx]);
                        }
                    if (!dx && !dy) sad0 = sad;
                    uint64_t key = ((uint64_t)(4 * sad + (uint32_t)(iabs(dx) + iabs(dy))) << 12) | (uint32_t)((dy + R) * span + dx + R);
                    if (key < best) best = key;
                }
            /* the centre only moves when that halves the zero-displacement SAD: small true motion is inside the
             * integer search anyway, and centres that jitter with the noise cost motion-vector bits (-0.11 dB on the bench clip) */
            int p = (int)(best & 4095);
            uint32_t sad_best = (uint32_t)(((best >> 12) - (uint32_t)(iabs(p % span - R) + iabs(p / span - R))) >> 2);
            if (2 * sad_best >= sad0) p = R * span + R;
            centers[2 * (cy * wc + cx)] = (int16_t)(4 * (p % span - R));
            centers[2 * (cy * wc + cx) + 1] = (int16_t)(4 * (p / span - R));
            if (costs) costs[cy * wc + cx] = sad_best;
        }
}

/* One picture over several devices: the picture a session codes may be a SLICE whose upper / lower neighbour is reconstructed on another
 * device and never seen (prm->mc_top / mc_bottom).  Motion compensation must then stay inside the slice: a block of n luma rows at row y
 * with vertical motion my (quarter samples) reads luma rows y + (my >> 2) + [-3, n + 4] when my has a fraction (8-tap), else its own n rows,
 * and chroma rows y/2 + (my >> 3) + [-1, n/2 + 2] when my & 7 (4-tap), else its n/2 rows. */
static int mv_rows_ok(int y, int n, int my, int h, int top, int bottom)
{
    int ly0 = y + (my >> 2) - ((my & 3) ? 3 : 0), ly1 = y + n - 1 + (my >> 2) + ((my & 3) ? 4 : 0);
    int cy0 = (y >> 1) + (my >> 3) - ((my & 7) ? 1 : 0), cy1 = (y >> 1) + (n >> 1) - 1 + (my >> 3) + ((my & 7) ? 2 : 0);
    if (top && (ly0 < 0 || cy0 < 0)) return 0;
    if (bottom && (ly1 > h - 1 || cy1 > (h >> 1) - 1)) return 0;
    return 1;
}
/* the search centre of a CTU is pulled back so that the whole +-R window keeps the CTU's rows inside the slice: every node then has
 * admissible integer candidates */
static int clamp_center_y(int sy, int y0, int R, int h, int top, int bottom)
{
    int ctu_h = h - y0 < ORC_CTU ? h - y0 : ORC_CTU;
    if (bottom && sy > h - (y0 + ctu_h) - R) sy = h - (y0 + ctu_h) - R;
    if (top && sy < R - y0) sy = R - y0;
    return sy;
}

/* 8x8 Hadamard "activity" of a source tile: the SATD it would have against a flat prediction of its own mean (the DC term
 * dropped) -- the yardstick a CTU's inter cost is held against before the intra second pass looks at it */
static int hadamard8_ac(const pix *a, int as)
{
    int m[8][8];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) m[y][x] = a[y * as + x];
    for (int y = 0; y < 8; y++)
        for (int st = 1; st < 8; st <<= 1)
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[y][i], q = m[y][i + st]; m[y][i] = p + q; m[y][i + st] = p - q; }
    for (int x = 0; x < 8; x++)
        for (int st = 1; st < 8; st <<= 1)
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[i][x], q = m[i + st][x]; m[i][x] = p + q; m[i + st][x] = p - q; }
    int s = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) if (x || y) s += iabs(m[y][x]);
    return (s + 2) >> 2;
}
static void intra_in_p_pass(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride, int w, int h,
                            const orc_params *prm, pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                            orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, const uint8_t *cand, const uint64_t *jinter);

void orc_analyze_inter_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                             const pix *ref_y, const pix *ref_u, const pix *ref_v, int ref_stride, int ref_cstride,
                             int w, int h, const orc_params *prm, const int16_t *centers,
                             pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                             orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, int32_t *me_dump, uint64_t *est)
{
    /* search domain: dy in [-R, R]; dx in [-R, -R + spanx - 1] with spanx = 2R+1 rounded UP to a multiple of 4 (the
     * device evaluates four horizontal positions per v_qsad_pk_u16_u8, so the row is widened instead of masked) */
    const int R = prm->me_range, spany = 2 * R + 1, spanx = (spany + 3) & ~3, bd = prm->bit_depth, lam = prm->lambda_sad_q4;
    const int wc = (w + ORC_CTU - 1) / ORC_CTU, hc = (h + ORC_CTU - 1) / ORC_CTU, w8 = w >> 3;
    uint32_t *sad8 = (uint32_t *)malloc(sizeof(uint32_t) * 16 * spanx * spany);
    int16_t *own_centers = NULL;
    if (!centers && prm->pre_search) {
        pix *ls = (pix *)malloc(sizeof(pix) * (w >> 2) * (h >> 2)), *lr = (pix *)malloc(sizeof(pix) * (w >> 2) * (h >> 2));
        own_centers = (int16_t *)malloc(sizeof(int16_t) * 2 * wc * hc);
        orc_lowres(src_y, src_stride, w, h, bd, ls);
        orc_lowres(ref_y, ref_stride, w, h, bd, lr);
        orc_pre_search(ls, lr, w >> 2, h >> 2, own_centers);
        centers = own_centers;
        free(ls); free(lr);
    }
    uint8_t *ip_cand = (uint8_t *)calloc((size_t)wc * hc, 1);
    uint64_t *ip_jinter = (uint64_t *)calloc((size_t)wc * hc, sizeof(uint64_t));
    for (int cy = 0; cy < hc; cy++)
        for (int cx = 0; cx < wc; cx++) {
            int ctu = cy * wc + cx, x0 = cx * ORC_CTU, y0 = cy * ORC_CTU;
            int sx = centers ? centers[2 * ctu] : 0, sy = centers ? centers[2 * ctu + 1] : 0;
            const int mct = prm->mc_top, mcb = prm->mc_bottom;
            if (mct || mcb) sy = clamp_center_y(sy, y0, R, h, mct, mcb);
            int valid[21], nx[21], ny[21], nl[21];
            for (int nd = 0; nd < 21; nd++) {
                node_geom(nd, &nx[nd], &ny[nd], &nl[nd]);
                valid[nd] = x0 + nx[nd] + (1 << nl[nd]) <= w && y0 + ny[nd] + (1 << nl[nd]) <= h;
            }
            /* --- integer full search: SAD of every 8x8 block at every position --- */
            for (int b = 0; b < 16; b++) {
                int nd = 5 + b;
                if (!valid[nd]) continue;
                for (int dy = -R; dy <= R; dy++)
                    for (int dx = -R; dx < -R + spanx; dx++)
                        sad8[(b * spany + dy + R) * spanx + dx + R] = sad_msb8(
                            src_y + (y0 + ny[nd]) * src_stride + x0 + nx[nd], src_stride,
                            ref_y + (y0 + ny[nd] + sy + dy) * ref_stride + x0 + nx[nd] + sx + dx, ref_stride, bd - 8);
            }
            int mvx[21], mvy[21];
            uint32_t cost[21];
            for (int nd = 0; nd < 21; nd++) {
                mvx[nd] = mvy[nd] = 0; cost[nd] = 0;
                if (!valid[nd]) continue;
                uint64_t best = ~0ull;
                for (int dy = -R; dy <= R; dy++)
                    for (int dx = -R; dx < -R + spanx; dx++) {
                        uint32_t s = 0;
                        int p = (dy + R) * spanx + dx + R;
                        if ((mct || mcb) && !mv_rows_ok(y0 + ny[nd], 1 << nl[nd], 4 * (sy + dy), h, mct, mcb)) continue;
                        if (nd == 0) for (int b = 0; b < 16; b++) s += sad8[b * spanx * spany + p];
                        else if (nd < 5) for (int b = 0; b < 4; b++) s += sad8[((nd - 1) * 4 + b) * spanx * spany + p];
                        else s = sad8[(nd - 5) * spanx * spany + p];
                        uint32_t c = (s << 4) + (uint32_t)(lam * (orc_mvd_bits(4 * dx) + orc_mvd_bits(4 * dy)));
                        uint64_t key = ((uint64_t)c << 16) | (uint32_t)p;
                        if (key < best) best = key;
                    }
                int p = (int)(best & 0xffff);
                mvx[nd] = 4 * (sx + p % spanx - R); mvy[nd] = 4 * (sy + p / spanx - R);
                cost[nd] = (uint32_t)(best >> 16);
            }
            if (me_dump)
                for (int nd = 0; nd < 21; nd++) {
                    me_dump[(ctu * 21 + nd) * 3 + 0] = valid[nd] ? mvx[nd] : 0;
                    me_dump[(ctu * 21 + nd) * 3 + 1] = valid[nd] ? mvy[nd] : 0;
                    me_dump[(ctu * 21 + nd) * 3 + 2] = valid[nd] ? (int32_t)cost[nd] : -1;
                }
            /* --- quadtree decision, bottom-up, on the SATD of every node at its INTEGER vector (+ lambda * mvd bits).  The
             *     fractional search then runs for the chosen CUs only: a CTU is covered once (16 tiles x 8 ring positions per round
             *     on the device) instead of once per tree level --- */
            uint32_t J[21], cint[21];
            pix pred[32 * 32];
            for (int nd = 0; nd < 21; nd++) {
                J[nd] = cint[nd] = 0;
                if (!valid[nd]) continue;
                int n = 1 << nl[nd], bx = x0 + nx[nd], by = y0 + ny[nd];
                orc_interp_luma(ref_y, ref_stride, bx, by, mvx[nd], mvy[nd], n, n, bd, pred, n);
                cint[nd] = ((uint32_t)orc_satd(src_y + by * src_stride + bx, src_stride, pred, n, n, n) << 4) +
                           (uint32_t)(lam * (orc_mvd_bits(mvx[nd] - 4 * sx) + orc_mvd_bits(mvy[nd] - 4 * sy)));
                J[nd] = cint[nd] + (uint32_t)(lam * 4);
            }
            int use16[4], use32;
            uint32_t J16[4];
            for (int q = 0; q < 4; q++) {
                uint32_t js = (uint32_t)(lam * 2);
                for (int s = 0; s < 4; s++) if (valid[5 + 4 * q + s]) js += J[5 + 4 * q + s];
                use16[q] = valid[1 + q] && J[1 + q] <= js;
                J16[q] = use16[q] ? J[1 + q] : js;
            }
            {
                uint32_t js = (uint32_t)(lam * 2);
                for (int q = 0; q < 4; q++) js += J16[q];
                use32 = valid[0] && J[0] <= js;
            }
            /* --- fractional refinement of the chosen CUs with SATD: half-pel ring then quarter-pel ring --- */
            uint32_t cost_sum = 0;
            for (int nd = 0; nd < 21; nd++) {
                if (!valid[nd]) continue;
                int chosen;
                if (nd == 0) chosen = use32;
                else if (nd < 5) chosen = !use32 && use16[nd - 1];
                else chosen = !use32 && !use16[(nd - 5) >> 2];
                if (!chosen) continue;
                int n = 1 << nl[nd], bx = x0 + nx[nd], by = y0 + ny[nd];
                const pix *s = src_y + by * src_stride + bx;
                int cmx = mvx[nd], cmy = mvy[nd];
                uint32_t cbest = cint[nd];
                for (int step = 2; step >= 1; step--) {
                    uint64_t best = ((uint64_t)cbest << 4) | 0;
                    for (int k = 0; k < 8; k++) {
                        int tx = cmx + kFracOff[k][0] * step, ty = cmy + kFracOff[k][1] * step;
                        if ((mct || mcb) && !mv_rows_ok(by, n, ty, h, mct, mcb)) continue;
                        orc_interp_luma(ref_y, ref_stride, bx, by, tx, ty, n, n, bd, pred, n);
                        uint32_t c = ((uint32_t)orc_satd(s, src_stride, pred, n, n, n) << 4) +
                                     (uint32_t)(lam * (orc_mvd_bits(tx - 4 * sx) + orc_mvd_bits(ty - 4 * sy)));
                        uint64_t key = ((uint64_t)c << 4) | (uint32_t)(k + 1);
                        if (key < best) best = key;
                    }
                    int k = (int)(best & 15);
                    if (k) { cmx += kFracOff[k - 1][0] * step; cmy += kFracOff[k - 1][1] * step; }
                    cbest = (uint32_t)(best >> 4);
                }
                mvx[nd] = cmx; mvy[nd] = cmy;
                cost_sum += cbest;
            }
            /* --- residual coding of the chosen CUs --- */
            for (int nd = 0; nd < 21; nd++) {
                if (!valid[nd]) continue;
                int chosen;
                if (nd == 0) chosen = use32;
                else if (nd < 5) chosen = !use32 && use16[nd - 1];
                else chosen = !use32 && !use16[(nd - 5) >> 2];
                if (!chosen) continue;
                int n = 1 << nl[nd], bx = x0 + nx[nd], by = y0 + ny[nd];
                pix pc[16 * 16];
                int flags = ORC_F_INTER;
                orc_interp_luma(ref_y, ref_stride, bx, by, mvx[nd], mvy[nd], n, n, bd, pred, n);
                if (code_tu_inter(src_y + by * src_stride + bx, src_stride, pred, n, rec_y + by * rec_stride + bx, rec_stride,
                                  coef_y + by * w + bx, w, nl[nd], prm->qp, bd, prm)) flags |= ORC_F_CBF_Y;
                for (int c = 0; c < 2; c++) {
                    const pix *rp = c ? ref_v : ref_u, *sp = c ? src_v : src_u;
                    pix *dp = c ? rec_v : rec_u;
                    int16_t *cp = c ? coef_v : coef_u;
                    orc_interp_chroma(rp, ref_cstride, bx / 2, by / 2, mvx[nd], mvy[nd], n / 2, n / 2, bd, pc, n / 2);
                    if (code_tu_inter(sp + (by / 2) * src_cstride + bx / 2, src_cstride, pc, n / 2,
                                      dp + (by / 2) * rec_cstride + bx / 2, rec_cstride, cp + (by / 2) * (w / 2) + bx / 2, w / 2,
                                      nl[nd] - 1, prm->qp_c, bd, prm)) flags |= c ? ORC_F_CBF_CR : ORC_F_CBF_CB;
                }
                for (int yy = 0; yy < n; yy += 8)
                    for (int xx = 0; xx < n; xx += 8) {
                        orc_cu_rec *r = &cu[((by + yy) >> 3) * w8 + ((bx + xx) >> 3)];
                        memset(r, 0, sizeof *r);
                        r->log2_size = (uint8_t)nl[nd]; r->flags = (uint8_t)flags; r->qp = (uint8_t)prm->qp;
                        r->mvx = (int16_t)mvx[nd]; r->mvy = (int16_t)mvy[nd];
                        r->intra_mode[0] = 1; r->chroma_mode = 1;
                    }
            }
            if (prm->intra_in_p) {
                /* second-pass candidate: the inter cost (SATD << 4 + lambda * mvd bits of the chosen CUs) exceeds both 4 per sample
                 * and the source's own AC activity -- predicting every tile by its mean would have done better */
                uint32_t act = 0, tiles = 0;
                for (int t = 0; t < 16; t++) {
                    int tx = x0 + (t & 3) * 8, ty = y0 + (t >> 2) * 8;
                    if (tx + 8 > w || ty + 8 > h) continue;
                    act += (uint32_t)hadamard8_ac(src_y + ty * src_stride + tx, src_stride);
                    tiles++;
                }
                ip_cand[ctu] = cost_sum > (act << 4) && cost_sum >= ((4u * 64u * tiles) << 4);
                /* J of the inter version in the units of the intra decision: SSE << 4 + lambda * estimated bits */
                uint64_t sse = 0;
                for (int pl = 0; pl < 3; pl++) {
                    const pix *sp = pl == 0 ? src_y : pl == 1 ? src_u : src_v, *rp = pl == 0 ? rec_y : pl == 1 ? rec_u : rec_v;
                    int ss = pl ? src_cstride : src_stride, rs = pl ? rec_cstride : rec_stride, sh = pl ? 1 : 0;
                    int bx = x0 >> sh, by = y0 >> sh, bw = (w >> sh) - bx < (ORC_CTU >> sh) ? (w >> sh) - bx : (ORC_CTU >> sh);
                    int bh = (h >> sh) - by < (ORC_CTU >> sh) ? (h >> sh) - by : (ORC_CTU >> sh);
                    for (int y = 0; y < bh; y++)
                        for (int x = 0; x < bw; x++) { int d = sp[(by + y) * ss + bx + x] - rp[(by + y) * rs + bx + x]; sse += (uint64_t)(d * d); }
                }
                uint64_t bits = estimate_bits_ctu(cu, coef_y, coef_u, coef_v, w, h, centers, cx, cy);
                ip_jinter[ctu] = (sse << 4) + (((uint64_t)prm->lambda_q4 * bits) >> 4);
            }
        }
    if (prm->intra_in_p)
        intra_in_p_pass(src_y, src_u, src_v, src_stride, src_cstride, w, h, prm, rec_y, rec_u, rec_v, rec_stride, rec_cstride, cu, coef_y, coef_u, coef_v,
                        ip_cand, ip_jinter);
    free(ip_cand); free(ip_jinter);
    free(sad8);
    if (est) *est = estimate_bits(cu, coef_y, coef_u, coef_v, w, h, centers);
    free(own_centers);
}

/* ================================================================================================
 * K1 + K3 : B picture (two reference pictures, one per list)
 * ================================================================================================ */
/* 14-bit intermediate prediction samples of 8.5.3.3.3 (before the rounding of 8.5.3.3.4.2): luma at quarter-sample, chroma at eighth-sample vectors */
static void interp14(const pix *ref, int rstride, int x, int y, int mvx, int mvy, int w, int h, int bit_depth, int chroma, int16_t *dst, int dstride)
{
    const int taps = chroma ? 4 : 8, half = chroma ? 1 : 3, fmask = chroma ? 7 : 3, fsh = chroma ? 3 : 2;
    const int fx = mvx & fmask, fy = mvy & fmask, xi = x + (mvx >> fsh), yi = y + (mvy >> fsh);
    const int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, shift3 = 14 - bit_depth;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const pix *p = ref + (yi + j) * rstride + xi + i;
            int v;
#define TAP(f, k) (chroma ? kChromaTap[f][k] : kLumaTap[f][k])
            if (!fx && !fy) v = p[0] << shift3;
            else if (!fy) { int a = 0; for (int k = 0; k < taps; k++) a += TAP(fx, k) * p[k - half]; v = a >> shift1; }
            else if (!fx) { int a = 0; for (int k = 0; k < taps; k++) a += TAP(fy, k) * p[(k - half) * rstride]; v = a >> shift1; }
            else {
                int a = 0;
                for (int r = 0; r < taps; r++) {
                    int t = 0;
                    for (int k = 0; k < taps; k++) t += TAP(fx, k) * p[(r - half) * rstride + k - half];
                    a += TAP(fy, r) * (t >> shift1);
                }
                v = a >> 6;
            }
#undef TAP
            dst[j * dstride + i] = (int16_t)v;
        }
}
/* 8.5.3.3.4.2: default weighted sample prediction of a block from its list-0 and / or list-1 14-bit predictions (NULL = list not used) */
static void weighted_default(const int16_t *p0, const int16_t *p1, int n, int bit_depth, pix *dst)
{
    const int maxv = (1 << bit_depth) - 1, shift1 = 14 - bit_depth, shift2 = 15 - bit_depth;
    for (int i = 0; i < n; i++) {
        int v;
        if (p0 && p1) v = (p0[i] + p1[i] + (1 << (shift2 - 1))) >> shift2;
        else v = ((p0 ? p0[i] : p1[i]) + (1 << (shift1 - 1))) >> shift1;
        dst[i] = (pix)CLIP3(0, maxv, v);
    }
}

/* the integer full search of orc_analyze_inter_frame for one CTU against one reference (no slice constraint: B pictures are not coded as slices) */
static void ctu_integer_search(const pix *src_y, int src_stride, const pix *ref_y, int ref_stride, int x0, int y0, int sx, int sy, int R, int bd, int lam,
                               const int *valid, const int *nx, const int *ny, uint32_t *sad8, int mvx[21], int mvy[21], uint32_t cost[21])
{
    const int spany = 2 * R + 1, spanx = (spany + 3) & ~3;
    for (int b = 0; b < 16; b++) {
        int nd = 5 + b;
        if (!valid[nd]) continue;
        for (int dy = -R; dy <= R; dy++)
            for (int dx = -R; dx < -R + spanx; dx++)
                sad8[(b * spany + dy + R) * spanx + dx + R] = sad_msb8(src_y + (y0 + ny[nd]) * src_stride + x0 + nx[nd], src_stride,
                                                                         ref_y + (y0 + ny[nd] + sy + dy) * ref_stride + x0 + nx[nd] + sx + dx, ref_stride, bd - 8);
    }
    for (int nd = 0; nd < 21; nd++) {
        mvx[nd] = mvy[nd] = 0; cost[nd] = 0;
        if (!valid[nd]) continue;
        uint64_t best = ~0ull;
        for (int dy = -R; dy <= R; dy++)
            for (int dx = -R; dx < -R + spanx; dx++) {
                uint32_t s = 0;
                int p = (dy + R) * spanx + dx + R;
                if (nd == 0) for (int b = 0; b < 16; b++) s += sad8[b * spanx * spany + p];
                else if (nd < 5) for (int b = 0; b < 4; b++) s += sad8[((nd - 1) * 4 + b) * spanx * spany + p];
                else s = sad8[(nd - 5) * spanx * spany + p];
                uint32_t c = (s << 4) + (uint32_t)(lam * (orc_mvd_bits(4 * dx) + orc_mvd_bits(4 * dy)));
                uint64_t key = ((uint64_t)c << 16) | (uint32_t)p;
                if (key < best) best = key;
            }
        int p = (int)(best & 0xffff);
        mvx[nd] = 4 * (sx + p % spanx - R); mvy[nd] = 4 * (sy + p / spanx - R);
        cost[nd] = (uint32_t)(best >> 16);
    }
}
/* SATD << 4 + lambda * mvd bits of block (bx, by, n) at vector (mx, my) against ref, vectors priced against the search centre (sx, sy) */
static uint32_t block_cost(const pix *src_y, int src_stride, const pix *ref_y, int ref_stride, int bx, int by, int n, int mx, int my, int sx, int sy, int bd, int lam)
{
    pix pred[32 * 32];
    orc_interp_luma(ref_y, ref_stride, bx, by, mx, my, n, n, bd, pred, n);
    return ((uint32_t)orc_satd(src_y + by * src_stride + bx, src_stride, pred, n, n, n) << 4) + (uint32_t)(lam * (orc_mvd_bits(mx - 4 * sx) + orc_mvd_bits(my - 4 * sy)));
}
/* half- then quarter-sample ring around (*mx, *my) whose cost is *c: the fractional refinement of orc_analyze_inter_frame */
static void refine_fraction(const pix *src_y, int src_stride, const pix *ref_y, int ref_stride, int bx, int by, int n, int sx, int sy, int bd, int lam, int *mx, int *my, uint32_t *c)
{
    for (int step = 2; step >= 1; step--) {
        uint64_t best = ((uint64_t)*c << 4) | 0;
        for (int k = 0; k < 8; k++) {
            int tx = *mx + kFracOff[k][0] * step, ty = *my + kFracOff[k][1] * step;
            uint32_t cc = block_cost(src_y, src_stride, ref_y, ref_stride, bx, by, n, tx, ty, sx, sy, bd, lam);
            uint64_t key = ((uint64_t)cc << 4) | (uint32_t)(k + 1);
            if (key < best) best = key;
        }
        int k = (int)(best & 15);
        if (k) { *mx += kFracOff[k - 1][0] * step; *my += kFracOff[k - 1][1] * step; }
        *c = (uint32_t)(best >> 4);
    }
}

void orc_analyze_b_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                         const pix *ref0_y, const pix *ref0_u, const pix *ref0_v, const pix *ref1_y, const pix *ref1_u, const pix *ref1_v,
                         int ref_stride, int ref_cstride, int w, int h, const orc_params *prm, const int16_t *centers0, const int16_t *centers1,
                         pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                         orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, int32_t *me_dump0, int32_t *me_dump1, uint64_t *est)
{
    const int R = prm->me_range, spany = 2 * R + 1, spanx = (spany + 3) & ~3, bd = prm->bit_depth, lam = prm->lambda_sad_q4;
    const int wc = (w + ORC_CTU - 1) / ORC_CTU, hc = (h + ORC_CTU - 1) / ORC_CTU, w8 = w >> 3;
    uint32_t *sad8 = (uint32_t *)malloc(sizeof(uint32_t) * 16 * spanx * spany);
    const pix *refy[2] = {ref0_y, ref1_y}, *refu[2] = {ref0_u, ref1_u}, *refv[2] = {ref0_v, ref1_v};
    const int16_t *cen[2] = {centers0, centers1};
    int32_t *dump[2] = {me_dump0, me_dump1};
    for (int cy = 0; cy < hc; cy++)
        for (int cx = 0; cx < wc; cx++) {
            const int ctu = cy * wc + cx, x0 = cx * ORC_CTU, y0 = cy * ORC_CTU;
            int valid[21], nx[21], ny[21], nl[21];
            for (int nd = 0; nd < 21; nd++) {
                node_geom(nd, &nx[nd], &ny[nd], &nl[nd]);
                valid[nd] = x0 + nx[nd] + (1 << nl[nd]) <= w && y0 + ny[nd] + (1 << nl[nd]) <= h;
            }
            int mvx[2][21], mvy[2][21], sx[2], sy[2];
            uint32_t cost[2][21];
            for (int l = 0; l < 2; l++) {
                sx[l] = cen[l] ? cen[l][2 * ctu] : 0; sy[l] = cen[l] ? cen[l][2 * ctu + 1] : 0;
                ctu_integer_search(src_y, src_stride, refy[l], ref_stride, x0, y0, sx[l], sy[l], R, bd, lam, valid, nx, ny, sad8, mvx[l], mvy[l], cost[l]);
                if (dump[l])
                    for (int nd = 0; nd < 21; nd++) {
                        dump[l][(ctu * 21 + nd) * 3 + 0] = valid[nd] ? mvx[l][nd] : 0;
                        dump[l][(ctu * 21 + nd) * 3 + 1] = valid[nd] ? mvy[l][nd] : 0;
                        dump[l][(ctu * 21 + nd) * 3 + 2] = valid[nd] ? (int32_t)cost[l][nd] : -1;
                    }
            }
            /* quadtree on the list-0 search, as in a P picture */
            uint32_t J[21], cint[21];
            for (int nd = 0; nd < 21; nd++) {
                J[nd] = cint[nd] = 0;
                if (!valid[nd]) continue;
                cint[nd] = block_cost(src_y, src_stride, refy[0], ref_stride, x0 + nx[nd], y0 + ny[nd], 1 << nl[nd], mvx[0][nd], mvy[0][nd], sx[0], sy[0], bd, lam);
                J[nd] = cint[nd] + (uint32_t)(lam * 4);
            }
            int use16[4], use32;
            uint32_t J16[4];
            for (int q = 0; q < 4; q++) {
                uint32_t js = (uint32_t)(lam * 2);
                for (int s = 0; s < 4; s++) if (valid[5 + 4 * q + s]) js += J[5 + 4 * q + s];
                use16[q] = valid[1 + q] && J[1 + q] <= js;
                J16[q] = use16[q] ? J[1 + q] : js;
            }
            {
                uint32_t js = (uint32_t)(lam * 2);
                for (int q = 0; q < 4; q++) js += J16[q];
                use32 = valid[0] && J[0] <= js;
            }
            for (int nd = 0; nd < 21; nd++) {
                if (!valid[nd]) continue;
                int chosen;
                if (nd == 0) chosen = use32;
                else if (nd < 5) chosen = !use32 && use16[nd - 1];
                else chosen = !use32 && !use16[(nd - 5) >> 2];
                if (!chosen) continue;
                const int n = 1 << nl[nd], bx = x0 + nx[nd], by = y0 + ny[nd];
                /* both lists: fractional refinement around the integer vector of THIS node's own search */
                int mx[2], my[2];
                uint32_t c[2];
                for (int l = 0; l < 2; l++) {
                    mx[l] = mvx[l][nd]; my[l] = mvy[l][nd];
                    c[l] = l == 0 ? cint[nd] : block_cost(src_y, src_stride, refy[1], ref_stride, bx, by, n, mx[1], my[1], sx[1], sy[1], bd, lam);
                    refine_fraction(src_y, src_stride, refy[l], ref_stride, bx, by, n, sx[l], sy[l], bd, lam, &mx[l], &my[l], &c[l]);
                }
                /* bi-prediction of the two refined vectors */
                int16_t p14[2][32 * 32];
                pix pred[32 * 32];
                for (int l = 0; l < 2; l++) interp14(refy[l], ref_stride, bx, by, mx[l], my[l], n, n, bd, 0, p14[l], n);
                weighted_default(p14[0], p14[1], n * n, bd, pred);
                const uint32_t cb = ((uint32_t)orc_satd(src_y + by * src_stride + bx, src_stride, pred, n, n, n) << 4) +
                                    (uint32_t)(lam * (orc_mvd_bits(mx[0] - 4 * sx[0]) + orc_mvd_bits(my[0] - 4 * sy[0]) + orc_mvd_bits(mx[1] - 4 * sx[1]) + orc_mvd_bits(my[1] - 4 * sy[1])));
                const uint64_t k0 = (((uint64_t)c[0] + (uint64_t)(lam * 2)) << 2) | 0, k1 = (((uint64_t)c[1] + (uint64_t)(lam * 2)) << 2) | 1, k2 = (((uint64_t)cb + (uint64_t)lam) << 2) | 2;
                const uint64_t kb = k0 <= k1 ? (k0 <= k2 ? k0 : k2) : (k1 <= k2 ? k1 : k2);
                const int mode = (int)(kb & 3), use0 = mode != 1, use1 = mode != 0;
                /* final prediction + residual */
                int flags = ORC_F_INTER | (use1 ? ORC_F_L1 : 0) | (use0 ? 0 : ORC_F_NOL0);
                weighted_default(use0 ? p14[0] : NULL, use1 ? p14[1] : NULL, n * n, bd, pred);
                if (code_tu_inter(src_y + by * src_stride + bx, src_stride, pred, n, rec_y + by * rec_stride + bx, rec_stride, coef_y + by * w + bx, w, nl[nd], prm->qp, bd, prm))
                    flags |= ORC_F_CBF_Y;
                for (int ci = 0; ci < 2; ci++) {
                    const pix *sp = ci ? src_v : src_u;
                    pix *dp = ci ? rec_v : rec_u, pc[16 * 16];
                    int16_t *cp = ci ? coef_v : coef_u, c14[2][16 * 16];
                    for (int l = 0; l < 2; l++)
                        if (l ? use1 : use0) interp14(ci ? refv[l] : refu[l], ref_cstride, bx / 2, by / 2, mx[l], my[l], n / 2, n / 2, bd, 1, c14[l], n / 2);
                    weighted_default(use0 ? c14[0] : NULL, use1 ? c14[1] : NULL, (n / 2) * (n / 2), bd, pc);
                    if (code_tu_inter(sp + (by / 2) * src_cstride + bx / 2, src_cstride, pc, n / 2, dp + (by / 2) * rec_cstride + bx / 2, rec_cstride,
                                      cp + (by / 2) * (w / 2) + bx / 2, w / 2, nl[nd] - 1, prm->qp_c, bd, prm)) flags |= ci ? ORC_F_CBF_CR : ORC_F_CBF_CB;
                }
                for (int yy = 0; yy < n; yy += 8)
                    for (int xx = 0; xx < n; xx += 8) {
                        orc_cu_rec *r = &cu[((by + yy) >> 3) * w8 + ((bx + xx) >> 3)];
                        memset(r, 0, sizeof *r);
                        r->log2_size = (uint8_t)nl[nd]; r->flags = (uint8_t)flags; r->qp = (uint8_t)prm->qp; r->chroma_mode = 1;
                        if (use0) { r->mvx = (int16_t)mx[0]; r->mvy = (int16_t)my[0]; }
                        if (use1) orc_set_mv1(r, mx[1], my[1]);
                    }
            }
        }
    free(sad8);
    if (est) *est = estimate_bits(cu, coef_y, coef_u, coef_v, w, h, centers0);
}

/* ================================================================================================
 * K2 + K3 : intra frame
 * ================================================================================================ */
typedef struct {
    const pix *src[3]; int sstride[3];
    pix *rec[3]; int rstride[3];
    int16_t *coef[3];
    orc_cu_rec *cu;
    int w, h, w8;
    const orc_params *prm;
} intra_ctx;

/* 8.4.2: candModeList from the left (x-1,y) and above (x,y-1) CUs; above is DC outside the current CTU row */
/* (x, y) is the top-left luma sample of the PU (a CU, or one 4x4 PU of an NxN CU); a neighbouring NxN CU answers with the
 * mode of the 4x4 PU that holds the neighbouring sample */
static void mpm_list(const intra_ctx *c, int x, int y, int cand[3])
{
    int a = 1, b = 1;
    if (x > 0 && same_tile(x - 1, y, x, y, c->w, c->h, c->prm->tile_cols, c->prm->tile_rows)) {
        const orc_cu_rec *r = &c->cu[(y >> 3) * c->w8 + ((x - 1) >> 3)];
        if (!(r->flags & ORC_F_INTER)) a = r->intra_mode[(r->flags & ORC_F_NXN) ? ((y >> 2) & 1) * 2 + (((x - 1) >> 2) & 1) : 0];
    }
    if (y > 0 && ((y - 1) >> ORC_CTU_LOG2) == (y >> ORC_CTU_LOG2)) {
        const orc_cu_rec *r = &c->cu[((y - 1) >> 3) * c->w8 + (x >> 3)];
        if (!(r->flags & ORC_F_INTER)) b = r->intra_mode[(r->flags & ORC_F_NXN) ? (((y - 1) >> 2) & 1) * 2 + ((x >> 2) & 1) : 0];
    }
    if (a == b) {
        if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
        else { cand[0] = a; cand[1] = 2 + ((a + 29) & 31); cand[2] = 2 + ((a - 2 + 1) & 31); }
    } else {
        cand[0] = a; cand[1] = b;
        cand[2] = (a != 0 && b != 0) ? 0 : (a != 1 && b != 1) ? 1 : 26;
    }
}
static int intra_mode_bits(const int cand[3], int mode)
{
    if (mode == cand[0]) return 2;
    if (mode == cand[1] || mode == cand[2]) return 3;
    return 6;
}

/* ---- stage A of an intra CTU: the PLAN.  Every quadtree node is costed on the SOURCE picture (its neighbours stand in for the
 * reconstruction, with the real availability rules: picture, tile, z-order), so all CTUs of a picture are independent and the device plans
 * them in one launch; only stage B (intra_cu below: prediction from the real reconstruction, residual, reconstruction) runs as a CTU
 * wavefront, and only for the CUs the plan chose.  Round 1 searched modes and tree depth-first on the reconstruction: 21 CUs x ~16 barrier
 * phases inside the wavefront, 23 % of the device time for 4 of 300 pictures.
 *   luma mode of a node: min over 35 modes of (SATD << 4) + lambda_sad * bits, bits 2 / 3 / 6 against a candidate list built (8.4.2) from
 *     the planned modes of the SAME-LEVEL nodes left of and above it inside the CTU (DC outside the CTU); ties -> lowest mode
 *   chroma mode: DM / planar / 26 / 10 / DC (a candidate equal to the luma mode stands for 34) by SATD over Cb + Cr + lambda_sad * (1 | 3)
 *   node cost J = (SSE << 4) + lambda * estimated bits of coding Y, Cb, Cr with those modes (prediction still from the source neighbourhood)
 *   tree: bottom-up, split = lambda + children, whole = J + lambda, whole wins ties */
typedef struct { uint8_t chosen[21], mode[21], cmode[21]; } intra_plan;

static void intra_plan_ctu(const intra_ctx *c, int x0, int y0, intra_plan *pl)
{
    const orc_params *prm = c->prm;
    const int bd = prm->bit_depth, lam = prm->lambda_sad_q4;
    int valid[21], nx[21], ny[21], nl[21];
    uint64_t J[21];
    pix ref[129], filt[129], pred[32 * 32];
    memset(pl, 0, sizeof *pl);
    for (int nd = 0; nd < 21; nd++) {
        node_geom(nd, &nx[nd], &ny[nd], &nl[nd]);
        valid[nd] = x0 + nx[nd] + (1 << nl[nd]) <= c->w && y0 + ny[nd] + (1 << nl[nd]) <= c->h;
        J[nd] = 0;
    }
    for (int level = 2; level >= 0; level--) {
        int first = level == 2 ? 5 : level == 1 ? 1 : 0, count = level == 2 ? 16 : level == 1 ? 4 : 1;
        for (int nd = first; nd < first + count; nd++) {
            if (!valid[nd]) continue;
            const int log2n = nl[nd], n = 1 << log2n, x = x0 + nx[nd], y = y0 + ny[nd];
            int a = 1, b = 1, cand[3];
            for (int k = first; k < nd; k++) {
                if (!valid[k]) continue;
                if (nx[k] + n == nx[nd] && ny[k] == ny[nd]) a = pl->mode[k];
                if (ny[k] + n == ny[nd] && nx[k] == nx[nd]) b = pl->mode[k];
            }
            if (a == b) {
                if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
                else { cand[0] = a; cand[1] = 2 + ((a + 29) & 31); cand[2] = 2 + ((a - 2 + 1) & 31); }
            } else {
                cand[0] = a; cand[1] = b;
                cand[2] = (a != 0 && b != 0) ? 0 : (a != 1 && b != 1) ? 1 : 26;
            }
            orc_intra_build_ref_tiles(c->src[0], c->sstride[0], x, y, log2n, c->w, c->h, 0, bd, prm->tile_cols, prm->tile_rows, ref);
            const pix *s = c->src[0] + y * c->sstride[0] + x;
            uint64_t best = ~0ull;
            for (int mode = 0; mode < 35; mode++) {
                orc_intra_filter_ref(ref, filt, log2n, mode, 0, bd, 1);
                orc_intra_pred(filt, pred, n, log2n, mode, 0, bd);
                uint32_t cst = ((uint32_t)orc_satd(s, c->sstride[0], pred, n, n, n) << 4) + (uint32_t)(lam * intra_mode_bits(cand, mode));
                uint64_t key = ((uint64_t)cst << 6) | (uint32_t)mode;
                if (key < best) best = key;
            }
            const int mode = (int)(best & 63);
            uint32_t cost = (uint32_t)(best >> 6);
            int cmode = mode;
            if (prm->chroma_modes) {
                static const int base[4] = {0, 26, 10, 1};
                int xc = x >> 1, yc = y >> 1, l2 = log2n - 1, nc = n >> 1;
                pix refc[2][129];
                for (int ci = 1; ci < 3; ci++)
                    orc_intra_build_ref_tiles(c->src[ci], c->sstride[ci], xc, yc, l2, c->w >> 1, c->h >> 1, ci, bd, prm->tile_cols, prm->tile_rows, refc[ci - 1]);
                uint64_t bestc = ~0ull;
                for (int k = 0; k < 5; k++) {
                    int m = k == 0 ? mode : (base[k - 1] == mode ? 34 : base[k - 1]);
                    uint32_t satd = 0;
                    for (int ci = 1; ci < 3; ci++) {
                        orc_intra_pred(refc[ci - 1], pred, nc, l2, m, ci, bd);
                        satd += (uint32_t)orc_satd(c->src[ci] + yc * c->sstride[ci] + xc, c->sstride[ci], pred, nc, nc, nc);
                    }
                    uint64_t key = ((uint64_t)((satd << 4) + (uint32_t)(lam * (k == 0 ? 1 : 3))) << 3) | (uint32_t)k;
                    if (key < bestc) bestc = key;
                }
                int k = (int)(bestc & 7);
                if (k) cmode = base[k - 1] == mode ? 34 : base[k - 1];
                cost += (uint32_t)(bestc >> 3);
            }
            pl->mode[nd] = (uint8_t)mode; pl->cmode[nd] = (uint8_t)cmode;
            /* RD cost of the node with these modes, still on the source neighbourhood: residual coding of Y, Cb, Cr into scratch */
            {
                pix rec_tmp[32 * 32];
                int16_t coef_tmp[32 * 32];
                int64_t sse, sse_total = 0;
                int bits, bits_total = 16 * intra_mode_bits(cand, mode) + 16 + 24 + (cmode != mode ? 32 : 0);
                orc_intra_filter_ref(ref, filt, log2n, mode, 0, bd, 1);
                orc_intra_pred(filt, pred, n, log2n, mode, 0, bd);
                code_tu(s, c->sstride[0], pred, n, rec_tmp, n, coef_tmp, n, log2n, prm->qp, bd, 1, 0, &sse, &bits, 0);
                sse_total += sse; bits_total += bits;
                for (int ci = 1; ci < 3; ci++) {
                    int xc = x >> 1, yc = y >> 1, l2 = log2n - 1, nc = n >> 1;
                    orc_intra_build_ref_tiles(c->src[ci], c->sstride[ci], xc, yc, l2, c->w >> 1, c->h >> 1, ci, bd, prm->tile_cols, prm->tile_rows, ref);
                    orc_intra_pred(ref, pred, nc, l2, cmode, ci, bd);
                    code_tu(c->src[ci] + yc * c->sstride[ci] + xc, c->sstride[ci], pred, nc, rec_tmp, nc, coef_tmp, nc, l2, prm->qp_c, bd, 1, 0, &sse, &bits, 0);
                    sse_total += sse; bits_total += bits;
                }
                J[nd] = ((uint64_t)sse_total << 4) + (((uint64_t)prm->lambda_q4 * (uint64_t)bits_total) >> 4);
            }
            (void)cost;
        }
    }
    /* tree, bottom-up on the RD costs: split = lambda + children, whole = own + lambda, whole wins ties (the rule the depth-first search
     * on the reconstruction used); nodes outside the picture count 0, a node that does not fit is always split */
    const uint64_t lam_split = ((uint64_t)prm->lambda_q4 * 16) >> 4;
    int use16[4], use32;
    uint64_t J16[4], js32 = lam_split;
    for (int q = 0; q < 4; q++) {
        uint64_t js = lam_split;
        for (int k = 0; k < 4; k++) if (valid[5 + 4 * q + k]) js += J[5 + 4 * q + k];
        use16[q] = valid[1 + q] && J[1 + q] + lam_split <= js;
        J16[q] = use16[q] ? J[1 + q] + lam_split : js;
        int any = 0;
        for (int k = 0; k < 4; k++) any |= valid[5 + 4 * q + k];
        js32 += any ? J16[q] : 0;
    }
    use32 = valid[0] && J[0] + lam_split <= js32;
    for (int nd = 0; nd < 21; nd++) {
        if (!valid[nd]) continue;
        pl->chosen[nd] = (uint8_t)(nd == 0 ? use32 : nd < 5 ? (!use32 && use16[nd - 1]) : (!use32 && !use16[(nd - 5) >> 2]));
    }
}

/* ---- stage B: one planned 2Nx2N intra CU: prediction from the reconstruction, residual coding (Y, Cb, Cr), records; returns RD cost */
static uint64_t intra_cu(intra_ctx *c, int x, int y, int log2n, int mode, int cmode)
{
    const orc_params *prm = c->prm;
    int n = 1 << log2n, bd = prm->bit_depth;
    pix ref[129], filt[129], pred[32 * 32];
    int cand[3];
    mpm_list(c, x, y, cand);
    orc_intra_build_ref_tiles(c->rec[0], c->rstride[0], x, y, log2n, c->w, c->h, 0, bd, prm->tile_cols, prm->tile_rows, ref);
    const pix *s = c->src[0] + y * c->sstride[0] + x;
    int64_t sse, sse_total = 0;
    int bits, bits_total = 16 * intra_mode_bits(cand, mode) + 16 + 24;
    int flags = 0;
    orc_intra_filter_ref(ref, filt, log2n, mode, 0, bd, 1);
    orc_intra_pred(filt, pred, n, log2n, mode, 0, bd);
    if (code_tu(s, c->sstride[0], pred, n, c->rec[0] + y * c->rstride[0] + x, c->rstride[0],
                c->coef[0] + y * c->w + x, c->w, log2n, prm->qp, bd, 1, 0, &sse, &bits, 0)) flags |= ORC_F_CBF_Y;
    sse_total += sse; bits_total += bits;
    /* intra_chroma_pred_mode (7.4.9.6 / Table 8-2) other than DM: 2 more bits in the rate estimate */
    if (cmode != mode) bits_total += 32;
    for (int ci = 1; ci < 3; ci++) {
        int xc = x >> 1, yc = y >> 1, l2 = log2n - 1, nc = n >> 1;
        orc_intra_build_ref_tiles(c->rec[ci], c->rstride[ci], xc, yc, l2, c->w >> 1, c->h >> 1, ci, bd, prm->tile_cols, prm->tile_rows, ref);
        orc_intra_pred(ref, pred, nc, l2, cmode, ci, bd);
        if (code_tu(c->src[ci] + yc * c->sstride[ci] + xc, c->sstride[ci], pred, nc,
                    c->rec[ci] + yc * c->rstride[ci] + xc, c->rstride[ci], c->coef[ci] + yc * (c->w >> 1) + xc, c->w >> 1,
                    l2, prm->qp_c, bd, 1, 0, &sse, &bits, 0)) flags |= ci == 1 ? ORC_F_CBF_CB : ORC_F_CBF_CR;
        sse_total += sse; bits_total += bits;
    }
    for (int yy = 0; yy < n; yy += 8)
        for (int xx = 0; xx < n; xx += 8) {
            orc_cu_rec *r = &c->cu[((y + yy) >> 3) * c->w8 + ((x + xx) >> 3)];
            memset(r, 0, sizeof *r);
            r->log2_size = (uint8_t)log2n; r->flags = (uint8_t)flags; r->qp = (uint8_t)prm->qp;
            r->intra_mode[0] = r->intra_mode[1] = r->intra_mode[2] = r->intra_mode[3] = (uint8_t)mode;
            r->chroma_mode = (uint8_t)cmode;
        }
    return ((uint64_t)sse_total << 4) + (((uint64_t)prm->lambda_q4 * (uint64_t)bits_total) >> 4);
}

/* the same 8x8 CU as four 4x4 PUs (part_mode NxN): per PU in z-order a 35-mode SATD search on the reconstructed
 * neighbourhood (earlier PUs included), DST-VII luma TU; chroma is one 4x4 TU per plane predicted with PU 0's mode (DM).
 * Overwrites the CU's reconstruction, levels and record; returns the RD cost in the units of intra_cu. */
static uint64_t intra_cu_nxn(intra_ctx *c, int x, int y)
{
    const orc_params *prm = c->prm;
    int bd = prm->bit_depth;
    orc_cu_rec *r = &c->cu[(y >> 3) * c->w8 + (x >> 3)];
    memset(r, 0, sizeof *r);
    r->log2_size = 3; r->flags = ORC_F_NXN; r->qp = (uint8_t)prm->qp;
    r->intra_mode[0] = r->intra_mode[1] = r->intra_mode[2] = r->intra_mode[3] = 1;
    int64_t sse, sse_total = 0;
    int bits, bits_total = 16 + 24;
    pix ref[17], filt[17], pred[16];
    for (int k = 0; k < 4; k++) {
        int xp = x + (k & 1) * 4, yp = y + (k >> 1) * 4, cand[3];
        mpm_list(c, xp, yp, cand);
        orc_intra_build_ref_tiles(c->rec[0], c->rstride[0], xp, yp, 2, c->w, c->h, 0, bd, prm->tile_cols, prm->tile_rows, ref);
        const pix *s = c->src[0] + yp * c->sstride[0] + xp;
        uint64_t best = ~0ull;
        for (int mode = 0; mode < 35; mode++) {
            orc_intra_filter_ref(ref, filt, 2, mode, 0, bd, 1);         /* 4x4: never filtered */
            orc_intra_pred(filt, pred, 4, 2, mode, 0, bd);
            uint32_t cst = ((uint32_t)orc_satd(s, c->sstride[0], pred, 4, 4, 4) << 4) + (uint32_t)(prm->lambda_sad_q4 * intra_mode_bits(cand, mode));
            uint64_t key = ((uint64_t)cst << 6) | (uint32_t)mode;
            if (key < best) best = key;
        }
        int mode = (int)(best & 63);
        r->intra_mode[k] = (uint8_t)mode;
        orc_intra_filter_ref(ref, filt, 2, mode, 0, bd, 1);
        orc_intra_pred(filt, pred, 4, 2, mode, 0, bd);
        if (code_tu(s, c->sstride[0], pred, 4, c->rec[0] + yp * c->rstride[0] + xp, c->rstride[0], c->coef[0] + yp * c->w + xp, c->w,
                    2, prm->qp, bd, 1, 1 /* DST-VII */, &sse, &bits, 0)) { r->cbf_y4 |= (uint8_t)(1 << k); r->flags |= ORC_F_CBF_Y; }
        sse_total += sse; bits_total += 16 * intra_mode_bits(cand, mode) + bits;
    }
    int cmode = r->intra_mode[0];
    r->chroma_mode = (uint8_t)cmode;
    for (int ci = 1; ci < 3; ci++) {
        int xc = x >> 1, yc = y >> 1;
        orc_intra_build_ref_tiles(c->rec[ci], c->rstride[ci], xc, yc, 2, c->w >> 1, c->h >> 1, ci, bd, prm->tile_cols, prm->tile_rows, ref);
        orc_intra_pred(ref, pred, 4, 2, cmode, ci, bd);
        if (code_tu(c->src[ci] + yc * c->sstride[ci] + xc, c->sstride[ci], pred, 4, c->rec[ci] + yc * c->rstride[ci] + xc, c->rstride[ci],
                    c->coef[ci] + yc * (c->w >> 1) + xc, c->w >> 1, 2, prm->qp_c, bd, 1, 0, &sse, &bits, 0)) r->flags |= ci == 1 ? ORC_F_CBF_CB : ORC_F_CBF_CR;
        sse_total += sse; bits_total += bits;
    }
    return ((uint64_t)sse_total << 4) + (((uint64_t)prm->lambda_q4 * (uint64_t)bits_total) >> 4);
}

static void region_copy(pix *dst, int ds, const pix *src, int ss, int w, int h)
{
    for (int y = 0; y < h; y++) memcpy(dst + y * ds, src + y * ss, w * sizeof(pix));
}
static void region_copy16(int16_t *dst, int ds, const int16_t *src, int ss, int w, int h)
{
    for (int y = 0; y < h; y++) memcpy(dst + y * ds, src + y * ss, w * sizeof(int16_t));
}

/* stage B of one CTU: the plan's CUs in decoding order (z-order), NxN trial on top of a planned 8x8 CU; returns the CTU's RD cost
 * (sum of the CU costs + lambda per quadtree node above the leaves), which the P pictures' second pass holds against the inter cost */
static uint64_t intra_code_ctu(intra_ctx *c, int x0, int y0, const intra_plan *pl)
{
    const uint64_t lam_split = ((uint64_t)c->prm->lambda_q4 * 16) >> 4;
    uint64_t j = lam_split;
    for (int q = 0; q < 4; q++) {
        int any = 0;
        for (int k = 0; k < 4; k++) {
            int nd = 5 + 4 * q + k, x, y, l;
            node_geom(nd, &x, &y, &l);
            x += x0; y += y0;
            if (!pl->chosen[nd]) continue;
            any = 1;
            uint64_t j2n = intra_cu(c, x, y, 3, pl->mode[nd], pl->cmode[nd]);
            /* NxN trial: only when the 2Nx2N CU left a luma residual; 2Nx2N wins ties */
            if (c->prm->intra_nxn && (c->cu[(y >> 3) * c->w8 + (x >> 3)].flags & ORC_F_CBF_Y)) {
                pix sv[3][64];
                int16_t sc[3][64];
                orc_cu_rec scu = c->cu[(y >> 3) * c->w8 + (x >> 3)];
                for (int ci = 0; ci < 3; ci++) {
                    int sh = ci ? 1 : 0;
                    region_copy(sv[ci], 8 >> sh, c->rec[ci] + (y >> sh) * c->rstride[ci] + (x >> sh), c->rstride[ci], 8 >> sh, 8 >> sh);
                    region_copy16(sc[ci], 8 >> sh, c->coef[ci] + (y >> sh) * (c->w >> sh) + (x >> sh), c->w >> sh, 8 >> sh, 8 >> sh);
                }
                uint64_t jnxn = intra_cu_nxn(c, x, y);
                if (jnxn < j2n) j2n = jnxn;
                else {
                    for (int ci = 0; ci < 3; ci++) {
                        int sh = ci ? 1 : 0;
                        region_copy(c->rec[ci] + (y >> sh) * c->rstride[ci] + (x >> sh), c->rstride[ci], sv[ci], 8 >> sh, 8 >> sh, 8 >> sh);
                        region_copy16(c->coef[ci] + (y >> sh) * (c->w >> sh) + (x >> sh), c->w >> sh, sc[ci], 8 >> sh, 8 >> sh, 8 >> sh);
                    }
                    c->cu[(y >> 3) * c->w8 + (x >> 3)] = scu;
                }
            }
            j += j2n;
        }
        if (any) j += lam_split;
        if (pl->chosen[1 + q]) {
            int x, y, l;
            node_geom(1 + q, &x, &y, &l);
            j += intra_cu(c, x0 + x, y0 + y, 4, pl->mode[1 + q], pl->cmode[1 + q]) + lam_split;
        }
    }
    if (pl->chosen[0]) j += intra_cu(c, x0, y0, 5, pl->mode[0], pl->cmode[0]);
    return j;
}

/* Intra second pass of a P picture.  The first pass coded every CTU inter; CTUs flagged in cand[] are re-coded as intra
 * (the I-picture quadtree search, predicting from the reconstruction around them) and keep the intra version when its
 * J is lower than the inter J.  An intra CTU predicts from its left / top-left / top / top-right neighbours, so only CTUs
 * whose candidate neighbours are settled may run together: round A takes candidates none of whose four causal
 * neighbours is a candidate, round B those whose candidate neighbours all ran in round A; the rest stay inter.  CTUs of a
 * round are mutually independent -- the device runs each round as one launch. */
static int ip_eligible_a(const uint8_t *cand, int wc, int hc, int cx, int cy)
{
    static const int nb[4][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    if (!cand[cy * wc + cx]) return 0;
    for (int k = 0; k < 4; k++) {
        int x = cx + nb[k][0], y = cy + nb[k][1];
        if (x >= 0 && y >= 0 && x < wc && y < hc && cand[y * wc + x]) return 0;
    }
    return 1;
}
static int ip_eligible_b(const uint8_t *cand, int wc, int hc, int cx, int cy)
{
    static const int nb[4][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    if (!cand[cy * wc + cx] || ip_eligible_a(cand, wc, hc, cx, cy)) return 0;
    for (int k = 0; k < 4; k++) {
        int x = cx + nb[k][0], y = cy + nb[k][1];
        if (x >= 0 && y >= 0 && x < wc && y < hc && cand[y * wc + x] && !ip_eligible_a(cand, wc, hc, x, y)) return 0;
    }
    return 1;
}
static void intra_in_p_pass(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride, int w, int h,
                            const orc_params *prm, pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                            orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, const uint8_t *cand, const uint64_t *jinter)
{
    orc_params ip = *prm;                          /* prm->tile_cols / tile_rows: the P pictures' own grid (PPS 0; 1x1 unless the session codes P pictures as tiles) */
    intra_ctx c;
    c.src[0] = src_y; c.src[1] = src_u; c.src[2] = src_v;
    c.sstride[0] = src_stride; c.sstride[1] = c.sstride[2] = src_cstride;
    c.rec[0] = rec_y; c.rec[1] = rec_u; c.rec[2] = rec_v;
    c.rstride[0] = rec_stride; c.rstride[1] = c.rstride[2] = rec_cstride;
    c.coef[0] = coef_y; c.coef[1] = coef_u; c.coef[2] = coef_v;
    c.cu = cu; c.w = w; c.h = h; c.w8 = w >> 3; c.prm = &ip;
    int wc = (w + ORC_CTU - 1) / ORC_CTU, hc = (h + ORC_CTU - 1) / ORC_CTU;
    for (int round = 0; round < 2; round++)
        for (int cy = 0; cy < hc; cy++)
            for (int cx = 0; cx < wc; cx++) {
                if (!(round == 0 ? ip_eligible_a(cand, wc, hc, cx, cy) : ip_eligible_b(cand, wc, hc, cx, cy))) continue;
                int x0 = cx * ORC_CTU, y0 = cy * ORC_CTU;
                int bw = w - x0 < ORC_CTU ? w - x0 : ORC_CTU, bh = h - y0 < ORC_CTU ? h - y0 : ORC_CTU;
                pix sv[3][32 * 32];
                int16_t sc[3][32 * 32];
                orc_cu_rec scu[16];
                for (int ci = 0; ci < 3; ci++) {
                    int sh = ci ? 1 : 0;
                    region_copy(sv[ci], 32 >> sh, c.rec[ci] + (y0 >> sh) * c.rstride[ci] + (x0 >> sh), c.rstride[ci], bw >> sh, bh >> sh);
                    region_copy16(sc[ci], 32 >> sh, c.coef[ci] + (y0 >> sh) * (w >> sh) + (x0 >> sh), w >> sh, bw >> sh, bh >> sh);
                }
                for (int yy = 0; yy < bh / 8; yy++)
                    for (int xx = 0; xx < bw / 8; xx++) scu[yy * 4 + xx] = cu[((y0 >> 3) + yy) * c.w8 + (x0 >> 3) + xx];
                intra_plan pl;
                intra_plan_ctu(&c, x0, y0, &pl);
                uint64_t jintra = intra_code_ctu(&c, x0, y0, &pl);
                if (jintra < jinter[cy * wc + cx]) continue;
                for (int ci = 0; ci < 3; ci++) {
                    int sh = ci ? 1 : 0;
                    region_copy(c.rec[ci] + (y0 >> sh) * c.rstride[ci] + (x0 >> sh), c.rstride[ci], sv[ci], 32 >> sh, bw >> sh, bh >> sh);
                    region_copy16(c.coef[ci] + (y0 >> sh) * (w >> sh) + (x0 >> sh), w >> sh, sc[ci], 32 >> sh, bw >> sh, bh >> sh);
                }
                for (int yy = 0; yy < bh / 8; yy++)
                    for (int xx = 0; xx < bw / 8; xx++) cu[((y0 >> 3) + yy) * c.w8 + (x0 >> 3) + xx] = scu[yy * 4 + xx];
            }
}

void orc_analyze_intra_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                             int w, int h, const orc_params *prm,
                             pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                             orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint64_t *est)
{
    intra_ctx c;
    c.src[0] = src_y; c.src[1] = src_u; c.src[2] = src_v;
    c.sstride[0] = src_stride; c.sstride[1] = c.sstride[2] = src_cstride;
    c.rec[0] = rec_y; c.rec[1] = rec_u; c.rec[2] = rec_v;
    c.rstride[0] = rec_stride; c.rstride[1] = c.rstride[2] = rec_cstride;
    c.coef[0] = coef_y; c.coef[1] = coef_u; c.coef[2] = coef_v;
    c.cu = cu; c.w = w; c.h = h; c.w8 = w >> 3; c.prm = prm;
    for (int y = 0; y < h; y += ORC_CTU)
        for (int x = 0; x < w; x += ORC_CTU) {
            intra_plan pl;
            intra_plan_ctu(&c, x, y, &pl);
            intra_code_ctu(&c, x, y, &pl);
        }
    if (est) *est = estimate_bits(cu, coef_y, coef_u, coef_v, w, h, NULL);
}

/* ================================================================================================
 * K4a : deblocking — H.265 8.7.2
 * ================================================================================================ */
static const uint8_t kBeta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18,
                                  20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static const uint8_t kTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

/* 8.7.2.4 boundary strength between the 8x8 blocks P and Q across a CU(=TU=PU) edge */
static int edge_bs(const orc_cu_rec *p, const orc_cu_rec *q)
{
    if (!(p->flags & ORC_F_INTER) || !(q->flags & ORC_F_INTER)) return 2;
    if ((p->flags & ORC_F_CBF_Y) || (q->flags & ORC_F_CBF_Y)) return 1;
    /* 8.7.2.4, motion: different reference pictures or numbers of vectors -> 1; else a vector component differing by >= 4 quarter samples -> 1.
     * Every list holds ONE picture and the two lists' pictures differ (the anchors before / after a B picture), so "the same reference pictures"
     * means "the same lists", and the vectors to compare are those of the same list */
    const int pu = p->flags & (ORC_F_L1 | ORC_F_NOL0), qu = q->flags & (ORC_F_L1 | ORC_F_NOL0);
    if (pu != qu) return 1;
    if (!(pu & ORC_F_NOL0) && (iabs(p->mvx - q->mvx) >= 4 || iabs(p->mvy - q->mvy) >= 4)) return 1;
    if ((pu & ORC_F_L1) && (iabs(orc_mv1x(p) - orc_mv1x(q)) >= 4 || iabs(orc_mv1y(p) - orc_mv1y(q)) >= 4)) return 1;
    return 0;
}

/* filter one 4-sample luma segment; `s` steps across the edge, `t` along it (8.7.2.5.3 / .6 / .7) */
static void luma_segment(pix *q0p, int s, int t, int beta, int tc, int maxv)
{
#define P(i, k) q0p[-(i + 1) * s + (k) * t]
#define Q(i, k) q0p[(i) * s + (k) * t]
    int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
    int dq0 = iabs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = iabs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
    int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
    if (d >= beta) return;
    int sam0 = 2 * dpq0 < (beta >> 2) && iabs(P(3, 0) - P(0, 0)) + iabs(Q(0, 0) - Q(3, 0)) < (beta >> 3) &&
               iabs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
    int sam3 = 2 * dpq3 < (beta >> 2) && iabs(P(3, 3) - P(0, 3)) + iabs(Q(0, 3) - Q(3, 3)) < (beta >> 3) &&
               iabs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
    int strong = sam0 && sam3;
    int dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
    for (int k = 0; k < 4; k++) {
        int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k), q0 = Q(0, k), q1 = Q(1, k), q2 = Q(2, k), q3 = Q(3, k);
        if (strong) {
            P(0, k) = (pix)CLIP3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            P(1, k) = (pix)CLIP3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
            P(2, k) = (pix)CLIP3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            Q(0, k) = (pix)CLIP3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            Q(1, k) = (pix)CLIP3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
            Q(2, k) = (pix)CLIP3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
        } else {
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (iabs(delta) >= tc * 10) continue;
            delta = CLIP3(-tc, tc, delta);
            P(0, k) = (pix)CLIP3(0, maxv, p0 + delta);
            Q(0, k) = (pix)CLIP3(0, maxv, q0 - delta);
            if (dep) { int dl = CLIP3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1); P(1, k) = (pix)CLIP3(0, maxv, p1 + dl); }
            if (deq) { int dl = CLIP3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1); Q(1, k) = (pix)CLIP3(0, maxv, q1 + dl); }
        }
    }
#undef P
#undef Q
}

void orc_deblock_frame(pix *rec_y, pix *rec_u, pix *rec_v, int stride, int cstride, int w, int h,
                       const orc_cu_rec *cu, int bit_depth, int cqp_off)
{
    int w8 = w >> 3, h8 = h >> 3, maxv = (1 << bit_depth) - 1, sc = 1 << (bit_depth - 8);
    for (int dir = 0; dir < 2; dir++) {      /* 0: vertical edges (filter across x), 1: horizontal edges */
        for (int by = 0; by < h8; by++)
            for (int bx = 0; bx < w8; bx++) {
                const orc_cu_rec *q = &cu[by * w8 + bx];
                int x = bx * 8, y = by * 8;
                int cu_mask = (1 << q->log2_size) - 1;
                if (dir == 0 ? (x == 0 || (x & cu_mask)) : (y == 0 || (y & cu_mask))) continue; /* not a CU edge */
                const orc_cu_rec *p = dir == 0 ? q - 1 : q - w8;
                int bs = edge_bs(p, q);
                if (!bs) continue;
                int qpl = (p->qp + q->qp + 1) >> 1;
                int beta = kBeta[CLIP3(0, 51, qpl)] * sc;
                int tc = kTc[CLIP3(0, 53, qpl + 2 * (bs - 1))] * sc;
                pix *e = rec_y + y * stride + x;
                for (int seg = 0; seg < 2; seg++)
                    luma_segment(dir == 0 ? e + seg * 4 * stride : e + seg * 4, dir == 0 ? 1 : stride, dir == 0 ? stride : 1, beta, tc, maxv);
                /* chroma: bS == 2 edges on the 8-sample chroma grid (8.7.2.5.5) */
                if (bs == 2 && ((dir == 0 ? x : y) & 15) == 0) {
                    int qpc = orc_chroma_qp(qpl + cqp_off);
                    int tcc = kTc[CLIP3(0, 53, qpc + 2)] * sc;
                    for (int ci = 0; ci < 2; ci++) {
                        pix *c = (ci ? rec_v : rec_u) + (y >> 1) * cstride + (x >> 1);
                        int s = dir == 0 ? 1 : cstride, t = dir == 0 ? cstride : 1;
                        for (int k = 0; k < 4; k++) {
                            int p0 = c[-s + k * t], p1 = c[-2 * s + k * t], q0 = c[k * t], q1 = c[s + k * t];
                            int delta = CLIP3(-tcc, tcc, (((q0 - p0) * 4 + p1 - q1 + 4) >> 3));
                            c[-s + k * t] = (pix)CLIP3(0, maxv, p0 + delta);
                            c[k * t] = (pix)CLIP3(0, maxv, q0 - delta);
                        }
                    }
                }
            }
    }
}

/* ================================================================================================
 * K4b : SAO — H.265 8.7.3 (apply) + encoder-side statistics/decision
 * ================================================================================================ */
static const int8_t kEoDx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}};
static const int8_t kEoDy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
static inline int sgn(int v) { return (v > 0) - (v < 0); }
static inline int eo_cat(const pix *p, int stride, int x, int y, int w, int h, int cls)
{
    int xa = x + kEoDx[cls][0], ya = y + kEoDy[cls][0], xb = x + kEoDx[cls][1], yb = y + kEoDy[cls][1];
    if (xa < 0 || xb < 0 || ya < 0 || yb < 0 || xa >= w || xb >= w || ya >= h || yb >= h) return 0;
    int c = p[y * stride + x];
    int e = 2 + sgn(c - p[ya * stride + xa]) + sgn(c - p[yb * stride + xb]);
    return e == 2 ? 0 : e < 2 ? e + 1 : e;     /* 0->1, 1->2, 2->0, 3->3, 4->4 */
}

/* best offset for (count n, sum s): start from the rounded mean (sign-constrained, |o|<=maxoff: 7 at 8 bit, 31 at 10) and walk toward 0
 * minimising  (n*o*o - 2*o*s)*16 + lambda_q4 * rate(o) ; returns offset, adds its cost to *cost */
static int sao_offset_rd(int n, int s, int sign_rule, int lam_q4, int band, int maxoff, int64_t *cost)
{
    if (n == 0) { *cost += 0 + (int64_t)lam_q4 * 1; return 0; }
    int o = (int)((2 * (int64_t)iabs(s) + n) / (2 * n));   /* round(|s|/n) */
    if (s < 0) o = -o;
    if (sign_rule > 0 && o < 0) o = 0;
    if (sign_rule < 0 && o > 0) o = 0;
    o = CLIP3(-maxoff, maxoff, o);
    int best_o = 0;
    int64_t best = (int64_t)lam_q4 * 1;        /* o = 0: one bin */
    int step = o > 0 ? 1 : -1;
    for (int t = step; o != 0 && t != o + step; t += step) {
        int a = iabs(t);
        int rate = (a < maxoff ? a + 1 : maxoff) + (band ? 1 : 0);
        int64_t c = (((int64_t)n * t * t - 2 * (int64_t)t * s) * 16) + (int64_t)lam_q4 * rate;
        if (c < best) { best = c; best_o = t; }
    }
    *cost += best;
    return best_o;
}

typedef struct { int type, cls, band; int8_t off[4]; int64_t cost; } sao_choice;

static void sao_stats(const pix *src, int sstride, const pix *dbk, int stride, int x0, int y0, int cw, int ch,
                      int w, int h, int bit_depth, int32_t eo_n[4][5], int32_t eo_s[4][5], int32_t bo_n[32], int32_t bo_s[32])
{
    memset(eo_n, 0, sizeof(int32_t) * 20); memset(eo_s, 0, sizeof(int32_t) * 20);
    memset(bo_n, 0, sizeof(int32_t) * 32); memset(bo_s, 0, sizeof(int32_t) * 32);
    for (int y = y0; y < y0 + ch && y < h; y++)
        for (int x = x0; x < x0 + cw && x < w; x++) {
            int d = src[y * sstride + x] - dbk[y * stride + x];
            int b = dbk[y * stride + x] >> (bit_depth - 5);
            bo_n[b]++; bo_s[b] += d;
            for (int c = 0; c < 4; c++) { int k = eo_cat(dbk, stride, x, y, w, h, c); eo_n[c][k]++; eo_s[c][k] += d; }
        }
}

/* candidates in fixed order: off, band, edge class 0..3; strict < keeps the earliest on ties */
static void sao_eval(int32_t eo_n[4][5], int32_t eo_s[4][5], int32_t bo_n[32], int32_t bo_s[32], int lam_q4, int bit_depth, sao_choice out[6])
{
    int maxoff = (1 << ((bit_depth < 10 ? bit_depth : 10) - 5)) - 1;
    memset(out, 0, sizeof(sao_choice) * 6);
    out[0].type = 0; out[0].cost = 0;
    /* band */
    int8_t bo_off[32]; int64_t bo_cost[32];
    for (int b = 0; b < 32; b++) { bo_cost[b] = 0; bo_off[b] = (int8_t)sao_offset_rd(bo_n[b], bo_s[b], 0, lam_q4, 1, maxoff, &bo_cost[b]); }
    int64_t bestb = 0; int pos = -1;
    for (int p = 0; p <= 28; p++) {
        int64_t c = bo_cost[p] + bo_cost[p + 1] + bo_cost[p + 2] + bo_cost[p + 3];
        if (pos < 0 || c < bestb) { bestb = c; pos = p; }
    }
    out[1].type = 1; out[1].band = pos; out[1].cost = bestb + (int64_t)lam_q4 * 7;
    for (int i = 0; i < 4; i++) out[1].off[i] = bo_off[pos + i];
    for (int c = 0; c < 4; c++) {
        sao_choice *o = &out[2 + c];
        o->type = 2; o->cls = c; o->cost = (int64_t)lam_q4 * 4;
        for (int k = 1; k <= 4; k++) o->off[k - 1] = (int8_t)sao_offset_rd(eo_n[c][k], eo_s[c][k], k <= 2 ? 1 : -1, lam_q4, 0, maxoff, &o->cost);
    }
}

static void sao_apply_ctb(const pix *dbk, int stride, pix *out, int ostride, int x0, int y0, int cw, int ch, int w, int h,
                          int bit_depth, int type, int cls, int band, const int8_t off[4])
{
    int maxv = (1 << bit_depth) - 1;
    for (int y = y0; y < y0 + ch && y < h; y++)
        for (int x = x0; x < x0 + cw && x < w; x++) {
            int v = dbk[y * stride + x];
            if (type == 2) {
                int k = eo_cat(dbk, stride, x, y, w, h, cls);
                if (k) v = CLIP3(0, maxv, v + off[k - 1]);
            } else if (type == 1) {
                int k = ((v >> (bit_depth - 5)) - band) & 31;
                if (k < 4) v = CLIP3(0, maxv, v + off[k]);
            }
            out[y * ostride + x] = (pix)v;
        }
}

void orc_sao_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                   const pix *dbk_y, const pix *dbk_u, const pix *dbk_v, int stride, int cstride,
                   pix *out_y, pix *out_u, pix *out_v, int ostride, int ocstride,
                   int w, int h, const orc_params *prm, orc_sao_ctu *sao)
{
    int wc = (w + ORC_CTU - 1) / ORC_CTU, hc = (h + ORC_CTU - 1) / ORC_CTU, bd = prm->bit_depth;
    int32_t eo_n[4][5], eo_s[4][5], bo_n[32], bo_s[32];
    for (int cy = 0; cy < hc; cy++)
        for (int cx = 0; cx < wc; cx++) {
            orc_sao_ctu *o = &sao[cy * wc + cx];
            memset(o, 0, sizeof *o);
            sao_choice ch[3][6];
            sao_stats(src_y, src_stride, dbk_y, stride, cx * ORC_CTU, cy * ORC_CTU, ORC_CTU, ORC_CTU, w, h, bd, eo_n, eo_s, bo_n, bo_s);
            sao_eval(eo_n, eo_s, bo_n, bo_s, prm->lambda_q4, bd, ch[0]);
            sao_stats(src_u, src_cstride, dbk_u, cstride, cx * 16, cy * 16, 16, 16, w / 2, h / 2, bd, eo_n, eo_s, bo_n, bo_s);
            sao_eval(eo_n, eo_s, bo_n, bo_s, prm->lambda_q4, bd, ch[1]);
            sao_stats(src_v, src_cstride, dbk_v, cstride, cx * 16, cy * 16, 16, 16, w / 2, h / 2, bd, eo_n, eo_s, bo_n, bo_s);
            sao_eval(eo_n, eo_s, bo_n, bo_s, prm->lambda_q4, bd, ch[2]);
            int bl = 0, bc = 0;
            for (int k = 1; k < 6; k++) {
                if (ch[0][k].cost < ch[0][bl].cost) bl = k;
                if (ch[1][k].cost + ch[2][k].cost < ch[1][bc].cost + ch[2][bc].cost) bc = k;
            }
            o->type[0] = (uint8_t)ch[0][bl].type; o->eo_class[0] = (uint8_t)ch[0][bl].cls; o->band_pos[0] = (uint8_t)ch[0][bl].band;
            memcpy(o->offset[0], ch[0][bl].off, 4);
            o->type[1] = (uint8_t)ch[1][bc].type; o->eo_class[1] = (uint8_t)ch[1][bc].cls;
            o->band_pos[1] = (uint8_t)ch[1][bc].band; o->band_pos[2] = (uint8_t)ch[2][bc].band;
            memcpy(o->offset[1], ch[1][bc].off, 4); memcpy(o->offset[2], ch[2][bc].off, 4);
        }
    orc_sao_apply_frame(dbk_y, dbk_u, dbk_v, stride, cstride, out_y, out_u, out_v, ostride, ocstride, w, h, bd, sao);
}

void orc_sao_apply_frame(const pix *dbk_y, const pix *dbk_u, const pix *dbk_v, int stride, int cstride,
                         pix *out_y, pix *out_u, pix *out_v, int ostride, int ocstride,
                         int w, int h, int bit_depth, const orc_sao_ctu *sao)
{
    int wc = (w + ORC_CTU - 1) / ORC_CTU, hc = (h + ORC_CTU - 1) / ORC_CTU;
    for (int cy = 0; cy < hc; cy++)
        for (int cx = 0; cx < wc; cx++) {
            const orc_sao_ctu *o = &sao[cy * wc + cx];
            sao_apply_ctb(dbk_y, stride, out_y, ostride, cx * ORC_CTU, cy * ORC_CTU, ORC_CTU, ORC_CTU, w, h, bit_depth,
                          o->type[0], o->eo_class[0], o->band_pos[0], o->offset[0]);
            sao_apply_ctb(dbk_u, cstride, out_u, ocstride, cx * 16, cy * 16, 16, 16, w / 2, h / 2, bit_depth,
                          o->type[1], o->eo_class[1], o->band_pos[1], o->offset[1]);
            sao_apply_ctb(dbk_v, cstride, out_v, ocstride, cx * 16, cy * 16, 16, 16, w / 2, h / 2, bit_depth,
                          o->type[1], o->eo_class[1], o->band_pos[2], o->offset[2]);
        }
}
