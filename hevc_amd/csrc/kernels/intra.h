// hevc_amd/csrc/kernels/intra.h — K2: intra prediction + mode decision (35 modes, SATD) + K3 residual for the CTUs
// of an I picture.  One 256-thread workgroup per 32x32 CTU; CTUs are launched one anti-diagonal (x + 2y = d) at a
// time because a CTU predicts from its left, top-left, top and top-right neighbours' reconstructions.  With a tile
// grid (prm.tile_cols x tile_rows, PPS 1) prediction stops at tile boundaries, so every tile runs its own wavefront and
// the launch count drops from W + 2(H-1) to w + 2(h-1) CTUs of one tile.
//
// Inside the CTU the quadtree is walked depth first exactly like oracle/hevc_oracle.c intra_tree: the four 8x8
// children of a 16x16 block first, then the 16x16 block itself, keep the cheaper (whole wins ties); then the same
// for 32x32.  The CTU's reconstruction, levels and CU records live in LDS until the CTU is final.
// Prediction is H.265 8.4.4.2 (reference availability by z-scan order, substitution, [1 2 1] / strong smoothing,
// planar, DC, angular with the boundary filters); MPM list per 8.4.2.
#pragma once
#include "common.h"
#include "residual.h"

namespace mihevc {

template <typename T> struct IntraArgs {
    Plane<const T> src[3];
    Plane<T> rec[3];             // pre-deblock reconstruction (read for neighbours, written for this CTU)
    int w, h, ctus_w, ctus_h;
    CostParams prm;
    mihevc_cu_rec *cu;
    int16_t *coef[3];
    int diagonal;                // informational: the launch passes the diagonal as a kernel argument
    unsigned long long *est;     // optional: picture-level rate estimate accumulator (1/16 bit)
    int sparse_coef;             // 1: store levels only for TUs with a non-zero level (see InterArgs)
    const IpInfo *ip;            // P pictures' intra second pass: per-CTU hand-over from the inter pass; nullptr in I pictures
};

// Second pass of a P picture (oracle: intra_in_p_pass).  Candidates may only run together when the CTUs they predict from are
// settled: round 0 takes candidates none of whose four causal neighbours (left, top-left, top, top-right) is a candidate,
// round 1 those whose candidate neighbours all ran in round 0.  Pure functions of the candidate map, so a round is one launch.
DEV bool ip_cand_at(const IpInfo *ip, int ctus_w, int ctus_h, int cx, int cy) { return cx >= 0 && cy >= 0 && cx < ctus_w && cy < ctus_h && ip[cy * ctus_w + cx].cand; }
DEV bool ip_eligible_a(const IpInfo *ip, int ctus_w, int ctus_h, int cx, int cy)
{
    return ip_cand_at(ip, ctus_w, ctus_h, cx, cy) && !ip_cand_at(ip, ctus_w, ctus_h, cx - 1, cy) && !ip_cand_at(ip, ctus_w, ctus_h, cx - 1, cy - 1) &&
           !ip_cand_at(ip, ctus_w, ctus_h, cx, cy - 1) && !ip_cand_at(ip, ctus_w, ctus_h, cx + 1, cy - 1);
}
DEV bool ip_eligible(const IpInfo *ip, int ctus_w, int ctus_h, int cx, int cy, int round)
{
    if (round == 0) return ip_eligible_a(ip, ctus_w, ctus_h, cx, cy);
    if (!ip_cand_at(ip, ctus_w, ctus_h, cx, cy) || ip_eligible_a(ip, ctus_w, ctus_h, cx, cy)) return false;
    const int nb[4][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
    for (int k = 0; k < 4; k++)
        if (ip_cand_at(ip, ctus_w, ctus_h, cx + nb[k][0], cy + nb[k][1]) && !ip_eligible_a(ip, ctus_w, ctus_h, cx + nb[k][0], cy + nb[k][1])) return false;
    return true;
}

constexpr int RY_STRIDE = 68;    // LDS luma neighbourhood: rows -1..31, cols -1..63 (+ pad)
constexpr int RC_STRIDE = 36;    // chroma: rows -1..15, cols -1..31

template <typename T> struct Nx4 {       // per block group g: 0 = luma PU or Cb, 1 = Cr
    T ref_raw[2][17], ref[2][17];
    unsigned av[2], nz[2];
    int16_t pred[2][16], res[2][16], lvl[2][16];
    int tmp[2][16];
};

template <typename T> struct IntraShared {
    ResidualShared rs;
    T src[1536];
    T pred[1536];
    T rec_y[33 * RY_STRIDE];
    T rec_c[2][17 * RC_STRIDE];
    T save_y[32 * 32];
    T save_c[2][16 * 16];
    int16_t coef_acc[1536], coef_save[1536];
    mihevc_cu_rec cu_acc[16], cu_save[16];
    mihevc_cu_rec left_cu[4];    // CU records of the left CTU's right column (MPM derivation), fetched once per CTU
    T ref_raw[3][132], ref[3][132], filt[132];   // [plane]: 4N+1 reference samples (raw, substituted); filtered luma
    uint8_t avail[3][132];
    int satd[35][16];
    int16_t hrow[35][64];        // 8x8 CUs: horizontally transformed difference rows of every mode (row-split SATD)
    int satd8[35];               // 8x8 CUs: sum |H d H| per mode (before the (s+2)>>2 normalisation)
    unsigned long long mode_key;  // min over modes of (cost << 6 | mode)
    int cand[3], dc_val[3];
    unsigned sse;
    int bits[3];
    unsigned long long j_cu;
    unsigned est;
    // NxN trial of an 8x8 CU: the 2Nx2N result parked here while four 4x4 PUs are coded in place
    T nx_rec[96];                // Y 8x8, Cb 4x4, Cr 4x4
    int16_t nx_coef[96];
    mihevc_cu_rec nx_cu;
    unsigned long long nx_j2n;
    unsigned nx_sse;
    int nx_bits, nx_keep;
    int nx_try;                  // set by intra_cu: the 2Nx2N CU left a luma residual (the NxN trial's condition)
    unsigned nx_cbf_c;           // CU_CBF_CB / CU_CBF_CR of the NxN trial
    unsigned csatd[5];           // chroma mode candidates (0 = DM, 1..4 = planar / 26 / 10 / DC): SATD over Cb + Cr
    int cmode, cmode_k;          // chosen chroma prediction mode and its candidate index
    int16_t nx_mat[2][16];       // 4x4 DST-VII and DCT matrices [k * 4 + n]
    int16_t tab_angle[35], tab_inv[35];   // Tables 8-4 / 8-5 by mode, LDS copies: a global read per use sat on the serial chain
    int16_t tab_qs[6], tab_ls[6];
    Nx4<T> nx;
};

// p[x][y] accessors on the linear 4N+1 layout: L[0] = p[-1][2N-1] ... L[2N] = p[-1][-1] ... L[4N] = p[2N-1][-1]
template <typename T> DEV int ref_left(const T *L, int n, int y) { return L[2 * n - 1 - y]; }
template <typename T> DEV int ref_top(const T *L, int n, int x) { return L[2 * n + 1 + x]; }

// one predicted sample — 8.4.4.2.4 (planar), .5 (DC incl. edge smoothing), .6 (angular incl. modes 10/26 edge filter)
// `angle` / `inv` are the mode's Table 8-4 / 8-5 entries, fetched once per lane by the caller (a global-memory read per
// sample sat on the critical path of the latency-bound intra wavefront)
template <typename T> DEV int intra_sample(const T *L, int log2n, int mode, int angle, int inv, int x, int y, int c_idx, int bit_depth, int dc)
{
    const int n = 1 << log2n;
    if (mode == 0)
        return ((n - 1 - x) * ref_left(L, n, y) + (x + 1) * ref_top(L, n, n) + (n - 1 - y) * ref_top(L, n, x) + (y + 1) * ref_left(L, n, n) + n) >> (log2n + 1);
    if (mode == 1) {
        if (c_idx == 0 && n < 32) {
            if (x == 0 && y == 0) return (ref_left(L, n, 0) + 2 * dc + ref_top(L, n, 0) + 2) >> 2;
            if (y == 0) return (ref_top(L, n, x) + 3 * dc + 2) >> 2;
            if (x == 0) return (ref_left(L, n, y) + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    const int vertical = mode >= 18;
    const int a = vertical ? y : x, b = vertical ? x : y;     // a: along the prediction direction
    if (angle == 0 && c_idx == 0 && n < 32 && b == 0) {
        int corner = L[2 * n];
        int v = vertical ? ref_top(L, n, 0) + ((ref_left(L, n, a) - corner) >> 1) : ref_left(L, n, 0) + ((ref_top(L, n, a) - corner) >> 1);
        return clip3(0, (1 << bit_depth) - 1, v);
    }
    const int idx = ((a + 1) * angle) >> 5, f = ((a + 1) * angle) & 31;
    auto r = [&](int i) -> int {
        if (i >= 0) return vertical ? L[2 * n + i] : L[2 * n - i];
        int k = -1 + ((i * inv + 128) >> 8);
        return vertical ? L[2 * n - 1 - k] : L[2 * n + 1 + k];
    };
    int v0 = r(b + idx + 1);
    return f ? ((32 - f) * v0 + f * r(b + idx + 2) + 16) >> 5 : v0;
}

DEV int mode_angle(int mode) { return g_tab.intra_angle[mode]; }
DEV int mode_inv_angle(int mode) { return (mode >= 11 && mode <= 25) ? g_tab.inv_angle[mode - 11] : 0; }

// difference (source - prediction) of one 8x8 tile of an NxN luma block for `mode`: the same arithmetic as
// intra_sample, with everything that depends only on the mode / row hoisted out of the sample loop (the SATD mode
// search runs 35 x (N/8)^2 of these per CU and dominated k_intra_diag: profiles/r01_a_first)
template <typename T>
DEV void intra_tile_diff(const T *L, int log2n, int mode, int angle, int inv, int tx, int ty, int bit_depth, int dc, const T *src, int src_stride, int (&m)[8][8],
                         bool luma = true)      // chroma blocks get no DC / mode 10 / mode 26 edge smoothing (8.4.4.2.5, 8.4.4.2.6)
{
    const int n = 1 << log2n;
    if (mode == 0) {
        const int tr = ref_top(L, n, n), bl = ref_left(L, n, n);
        int top[8];
#pragma unroll
        for (int i = 0; i < 8; i++) top[i] = ref_top(L, n, tx + i);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int y = ty + j, lf = ref_left(L, n, y);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int x = tx + i;
                m[j][i] = (int)src[j * src_stride + i] - (((n - 1 - x) * lf + (x + 1) * tr + (n - 1 - y) * top[i] + (y + 1) * bl + n) >> (log2n + 1));
            }
        }
        return;
    }
    if (mode == 1) {
        const bool edge = luma && n < 32;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int x = tx + i, y = ty + j;
                int v = dc;
                if (edge && (x == 0 || y == 0)) {
                    if (x == 0 && y == 0) v = (ref_left(L, n, 0) + 2 * dc + ref_top(L, n, 0) + 2) >> 2;
                    else if (y == 0) v = (ref_top(L, n, x) + 3 * dc + 2) >> 2;
                    else v = (ref_left(L, n, y) + 3 * dc + 2) >> 2;
                }
                m[j][i] = (int)src[j * src_stride + i] - v;
            }
        return;
    }
    const int vertical = mode >= 18;
    const int a0 = vertical ? ty : tx, b0 = vertical ? tx : ty;
    const int maxv = (1 << bit_depth) - 1;
#pragma unroll
    for (int ai = 0; ai < 8; ai++) {                 // along the prediction direction
        const int a = a0 + ai, idx = ((a + 1) * angle) >> 5, f = ((a + 1) * angle) & 31;
        int r[9];
#pragma unroll
        for (int e = 0; e < 9; e++) {
            const int i = b0 + e + idx + 1;
            if (i >= 0) r[e] = vertical ? L[2 * n + i] : L[2 * n - i];
            else { const int k = -1 + ((i * inv + 128) >> 8); r[e] = vertical ? L[2 * n - 1 - k] : L[2 * n + 1 + k]; }
            if (e == 8 && !f) r[e] = 0;              // never read when the fraction is zero (may lie past the array)
        }
#pragma unroll
        for (int bi = 0; bi < 8; bi++) {
            int v = f ? ((32 - f) * r[bi] + f * r[bi + 1] + 16) >> 5 : r[bi];
            if (luma && angle == 0 && n < 32 && b0 + bi == 0) {
                const int corner = L[2 * n];
                v = vertical ? ref_top(L, n, 0) + ((ref_left(L, n, a) - corner) >> 1) : ref_left(L, n, 0) + ((ref_top(L, n, a) - corner) >> 1);
                v = clip3(0, maxv, v);
            }
            if (vertical) m[ai][bi] = (int)src[ai * src_stride + bi] - v;
            else m[bi][ai] = (int)src[bi * src_stride + ai] - v;
        }
    }
}

// 8.4.2 candModeList of the PU whose top-left luma sample is (px, py) in CTU coordinates (a CU, or a 4x4 PU of an NxN
// CU).  A neighbouring NxN CU answers with the mode of the 4x4 PU that holds the neighbouring sample; the left CTU's
// right column comes from s.left_cu; the CTU above is never consulted (8.4.2: DC).
template <typename T> DEV void mpm_cand(const IntraShared<T> &s, int px, int py, bool left_ok, int (&cand)[3])
{
    int ma = 1, mb = 1;
    if (px > 0 || left_ok) {
        const mihevc_cu_rec &r = px > 0 ? s.cu_acc[(py >> 3) * 4 + ((px - 1) >> 3)] : s.left_cu[py >> 3];
        if (!(r.flags & CU_INTER)) ma = r.intra_mode[(r.flags & CU_NXN) ? ((py >> 2) & 1) * 2 + (((px - 1) >> 2) & 1) : 0];
    }
    if (py > 0) {
        const mihevc_cu_rec &r = s.cu_acc[((py - 1) >> 3) * 4 + (px >> 3)];
        if (!(r.flags & CU_INTER)) mb = r.intra_mode[(r.flags & CU_NXN) ? (((py - 1) >> 2) & 1) * 2 + ((px >> 2) & 1) : 0];
    }
    if (ma == mb) {
        if (ma < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
        else { cand[0] = ma; cand[1] = 2 + ((ma + 29) & 31); cand[2] = 2 + ((ma - 2 + 1) & 31); }
    } else {
        cand[0] = ma; cand[1] = mb;
        cand[2] = (ma != 0 && mb != 0) ? 0 : (ma != 1 && mb != 1) ? 1 : 26;
    }
}

DEV bool intra_filter_on(int log2n, int mode)
{
    if (mode == 1 || log2n == 2) return false;
    int d = imin(iabs(mode - 26), iabs(mode - 10));
    return d > (log2n == 3 ? 7 : log2n == 4 ? 1 : 0);
}

template <typename T, class Ex>
DEV void intra_cu(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int x0, int y0, int cx, int cy, int log2n)
{
    const int n = 1 << log2n, bd = a.prm.bit_depth, tiles = n >> 3, ntile = tiles * tiles;
    const Region rg{cx, cy, log2n};
    const int rcnt = rg.count();
    const int gx = x0 + cx, gy = y0 + cy;        // picture coordinates of the CU
    // tile of this CTU as CTB bounds (neighbours outside it are unavailable, 6.4.1); one tile = the whole picture
    const int ctu_x = x0 >> CTU_LOG2, ctu_y = y0 >> CTU_LOG2;
    const int tcn = a.prm.tile_cols > 1 ? a.prm.tile_cols : 1, trn = a.prm.tile_rows > 1 ? a.prm.tile_rows : 1;
    const int tci = tile_of(ctu_x, tcn, a.ctus_w), tri = tile_of(ctu_y, trn, a.ctus_h);
    const int tx_lo = tile_bd(tci, tcn, a.ctus_w) << CTU_LOG2, tx_hi = tile_bd(tci + 1, tcn, a.ctus_w) << CTU_LOG2;
    const int ty_lo = tile_bd(tri, trn, a.ctus_h) << CTU_LOG2;
    // reference samples: availability + raw values for the three planes
    ex.phase([&](int tid) {
        for (int u = tid; u < 3 * 129; u += NT) {
            int pl = u / 129, i = u % 129, np = pl ? n >> 1 : n, total = 4 * np + 1;
            if (i >= total) continue;
            int px = pl ? cx >> 1 : cx, py = pl ? cy >> 1 : cy, xn, yn;
            if (i < 2 * np) { xn = px - 1; yn = py + 2 * np - 1 - i; }
            else if (i == 2 * np) { xn = px - 1; yn = py - 1; }
            else { xn = px + (i - 2 * np - 1); yn = py - 1; }
            int sh = pl ? 1 : 0, lx = (xn << sh) + x0, ly = (yn << sh) + y0;      // luma picture position of the neighbour
            bool ok = lx >= tx_lo && ly >= ty_lo && lx < a.w && lx < tx_hi && ly < a.h && zaddr(lx, ly, a.ctus_w) < zaddr(gx, gy, a.ctus_w);
            s.avail[pl][i] = ok;
            s.ref_raw[pl][i] = ok ? (pl ? s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] : s.rec_y[(yn + 1) * RY_STRIDE + xn + 1]) : (T)0;
        }
        if (tid == 0) {   // 8.4.2 candModeList
            int cand[3];
            mpm_cand(s, cx, cy, x0 > tx_lo, cand);
            s.cand[0] = cand[0]; s.cand[1] = cand[1]; s.cand[2] = cand[2];
            s.sse = 0; s.bits[0] = s.bits[1] = s.bits[2] = 0;
            s.mode_key = ~0ull;
        }
    });
    // substitution (8.4.4.2.2): nearest available sample at a lower index, else the first available above
    ex.phase([&](int tid) {
        for (int u = tid; u < 3 * 129; u += NT) {
            int pl = u / 129, i = u % 129, np = pl ? n >> 1 : n, total = 4 * np + 1;
            if (i >= total) continue;
            int j = i;
            while (j >= 0 && !s.avail[pl][j]) j--;
            if (j < 0) { j = i + 1; while (j < total && !s.avail[pl][j]) j++; }
            s.ref[pl][i] = j < total ? s.ref_raw[pl][j] : (T)(1 << (bd - 1));
        }
    });
    // smoothing filter for luma (8.4.4.2.3) + DC values
    ex.phase([&](int tid) {
        const T *L = s.ref[0];
        const int total = 4 * n + 1;
        bool strong = false;
        if (n == 32) {
            int thr = 1 << (bd - 5), c = L[64];
            strong = iabs(c + L[128] - 2 * L[96]) < thr && iabs(c + L[0] - 2 * L[32]) < thr;
        }
        for (int i = tid; i < total; i += NT) {
            int v;
            if (i == 0 || i == total - 1) v = L[i];
            else if (strong) v = i == 64 ? L[64] : i < 64 ? (i * L[64] + (64 - i) * L[0] + 32) >> 6 : ((128 - i) * L[64] + (i - 64) * L[128] + 32) >> 6;
            else v = (L[i - 1] + 2 * L[i] + L[i + 1] + 2) >> 2;
            s.filt[i] = (T)v;
        }
        if (tid < 3) {
            int np = tid ? n >> 1 : n, lg = tid ? log2n - 1 : log2n, sum = np;
            for (int i = 0; i < np; i++) sum += ref_top(s.ref[tid], np, i) + ref_left(s.ref[tid], np, i);
            s.dc_val[tid] = sum >> (lg + 1);
        }
    });
    // 35 modes x 8x8 tiles: prediction and SATD against the source
    if (log2n == 3) {
        // an 8x8 CU has one tile per mode: 35 busy lanes would leave the workgroup idle while this CU blocks the rest of
        // the CTU.  Split every tile by rows: 280 lanes each predict one row and transform it horizontally, then
        // 280 lanes each finish one column (the 2-D Hadamard is separable, so the sum is unchanged).
        ex.phase([&](int tid) {
            for (int u = tid; u < 35 * 8; u += NT) {
                const int mode = u >> 3, y = u & 7;
                const T *L = intra_filter_on(3, mode) ? s.filt : s.ref[0];
                const int ang = s.tab_angle[mode], inv = s.tab_inv[mode];
                int d[8];
#pragma unroll
                for (int x = 0; x < 8; x++) d[x] = (int)s.src[(cy + y) * 32 + cx + x] - intra_sample<T>(L, 3, mode, ang, inv, x, y, 0, bd, s.dc_val[0]);
#pragma unroll
                for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (!(i & st)) { int p = d[i], q = d[i + st]; d[i] = p + q; d[i + st] = p - q; }
#pragma unroll
                for (int x = 0; x < 8; x++) s.hrow[mode][y * 8 + x] = (int16_t)d[x];
            }
            if (tid < 35) s.satd8[tid] = 0;
        });
        ex.phase([&](int tid) {
            for (int u = tid; u < 35 * 8; u += NT) {
                const int mode = u >> 3, x = u & 7;
                int d[8];
#pragma unroll
                for (int y = 0; y < 8; y++) d[y] = s.hrow[mode][y * 8 + x];
#pragma unroll
                for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (!(i & st)) { int p = d[i], q = d[i + st]; d[i] = p + q; d[i + st] = p - q; }
                int sum = 0;
#pragma unroll
                for (int y = 0; y < 8; y++) sum += iabs(d[y]);
                ex.atomic_add(&s.satd8[mode], sum);
            }
        });
    } else {
    ex.phase([&](int tid) {
        for (int u = tid; u < 35 * ntile; u += NT) {
            int mode = u / ntile, t = u % ntile, tx = (t % tiles) * 8, ty = (t / tiles) * 8;
            const T *L = intra_filter_on(log2n, mode) ? s.filt : s.ref[0];
            int m[8][8];
            intra_tile_diff<T>(L, log2n, mode, s.tab_angle[mode], s.tab_inv[mode], tx, ty, bd, s.dc_val[0], s.src + (cy + ty) * 32 + cx + tx, 32, m);
            s.satd[mode][t] = hadamard8_satd(m);
        }
    });
    }
    ex.phase([&](int tid) {
        if (tid < 35) {
            unsigned satd = 0;
            if (log2n == 3) satd = (unsigned)((s.satd8[tid] + 2) >> 2);
            else for (int t = 0; t < ntile; t++) satd += (unsigned)s.satd[tid][t];
            int bits = tid == s.cand[0] ? 2 : (tid == s.cand[1] || tid == s.cand[2]) ? 3 : 6;
            unsigned cost = (satd << 4) + (unsigned)(a.prm.lambda_sad_q4 * bits);
            ex.atomic_min(&s.mode_key, ((unsigned long long)cost << 6) | (unsigned)tid);      // ties -> lowest mode
        }
        if (tid >= 64 && tid < 80) {
            int t = tid - 64, tx = t & 3, ty = t >> 2;
            bool in = tx * 8 >= cx && tx * 8 < cx + n && ty * 8 >= cy && ty * 8 < cy + n;
            s.rs.tu_log2[t] = in ? (uint8_t)log2n : 0;
            s.rs.tu_intra[t] = 1;
        }
        if (tid >= 80 && tid < 83) s.rs.cbf[tid - 80] = 0;
        if (tid >= 96 && tid < 101) s.csatd[tid - 96] = 0;
        if (tid == 101) { s.cmode_k = 0; }
    });
    // intra_chroma_pred_mode (oracle intra_cu): DM or planar / 26 / 10 / DC (a candidate equal to the luma mode stands for 34), by SATD
    // over Cb + Cr + lambda * (1 bit DM, 3 bits otherwise); DM wins ties
    if (a.prm.chroma_modes) {
        ex.phase([&](int tid) {
            const int mode = (int)(s.mode_key & 63), l2c = log2n - 1, nc = n >> 1, ct = nc >= 8 ? nc >> 3 : 1, ntc = ct * ct;
            for (int u = tid; u < 5 * 2 * ntc; u += NT) {
                const int k = u / (2 * ntc), pl = 1 + (u / ntc) % 2, t = u % ntc;
                const int base = k == 1 ? 0 : k == 2 ? 26 : k == 3 ? 10 : 1, m = k == 0 ? mode : (base == mode ? 34 : base);
                const T *L = s.ref[pl];
                const int sbase = 1024 + (pl - 1) * 256 + (cy >> 1) * 16 + (cx >> 1);
                int satd;
                if (nc == 4) {
                    int d[16];
                    const int ang = s.tab_angle[m], inv = s.tab_inv[m];
#pragma unroll
                    for (int i = 0; i < 16; i++) d[i] = (int)s.src[sbase + (i >> 2) * 16 + (i & 3)] - intra_sample<T>(L, 2, m, ang, inv, i & 3, i >> 2, pl, bd, s.dc_val[pl]);
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        int p0 = d[y * 4] + d[y * 4 + 1], p1 = d[y * 4] - d[y * 4 + 1], p2 = d[y * 4 + 2] + d[y * 4 + 3], p3 = d[y * 4 + 2] - d[y * 4 + 3];
                        d[y * 4] = p0 + p2; d[y * 4 + 1] = p1 + p3; d[y * 4 + 2] = p0 - p2; d[y * 4 + 3] = p1 - p3;
                    }
                    int sum = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        int p0 = d[x] + d[4 + x], p1 = d[x] - d[4 + x], p2 = d[8 + x] + d[12 + x], p3 = d[8 + x] - d[12 + x];
                        sum += iabs(p0 + p2) + iabs(p1 + p3) + iabs(p0 - p2) + iabs(p1 - p3);
                    }
                    satd = (sum + 1) >> 1;
                } else {
                    int mm[8][8];
                    const int tx = (t % ct) * 8, ty = (t / ct) * 8;
                    intra_tile_diff<T>(L, l2c, m, s.tab_angle[m], s.tab_inv[m], tx, ty, bd, s.dc_val[pl], s.src + sbase + ty * 16 + tx, 16, mm, false);
                    satd = hadamard8_satd(mm);
                }
                ex.atomic_add(&s.csatd[k], (unsigned)satd);
            }
        });
        ex.phase([&](int tid) {
            if (tid != 0) return;
            const int mode = (int)(s.mode_key & 63);
            unsigned long long best = ~0ull;
            for (int k = 0; k < 5; k++) {
                const unsigned long long key = ((unsigned long long)((s.csatd[k] << 4) + (unsigned)(a.prm.lambda_sad_q4 * (k == 0 ? 1 : 3))) << 3) | (unsigned)k;
                best = key < best ? key : best;
            }
            const int k = (int)(best & 7), base = k == 1 ? 0 : k == 2 ? 26 : k == 3 ? 10 : 1;
            s.cmode_k = k;
            s.cmode = k == 0 ? mode : (base == mode ? 34 : base);
        });
    }
    // prediction of the chosen mode (luma) and the chosen chroma mode (DM unless prm.chroma_modes), residual
    ex.phase([&](int tid) {
        const int mode = (int)(s.mode_key & 63), cmode = s.cmode_k ? s.cmode : mode;
        const int cang = s.tab_angle[cmode], cinv = s.tab_inv[cmode];
        const T *L = intra_filter_on(log2n, mode) ? s.filt : s.ref[0];
        const int ang = s.tab_angle[mode], inv = s.tab_inv[mode];
        for (int k = tid; k < rcnt; k += NT) {
            const int i = rg.index(k);
            SampleLoc l = locate(s.rs, i);
            s.rs.desc[i] = pack_loc(l);
            if (!l.log2n) continue;
            int v;
            if (l.plane == 0) v = intra_sample<T>(L, log2n, mode, ang, inv, l.x - cx, l.y - cy, 0, bd, s.dc_val[0]);
            else v = intra_sample<T>(s.ref[l.plane], log2n - 1, cmode, cang, cinv, l.x - (cx >> 1), l.y - (cy >> 1), l.plane, bd, s.dc_val[l.plane]);
            s.pred[i] = (T)v;
            s.rs.res[i] = (int16_t)((int)s.src[i] - v);
        }
    });
    residual_pipeline(ex, s.rs, a.prm.qp, a.prm.qp_c, bd, rg);
    // reconstruction into the LDS neighbourhood, distortion, rate estimate
    ex.phase([&](int tid) {
        const int maxv = (1 << bd) - 1;
        unsigned sse = 0;
        for (int k = tid; k < rcnt; k += NT) {
            const int i = rg.index(k);
            SampleLoc l = locate(s.rs, i);
            if (!l.log2n) continue;
            int v = clip3(0, maxv, (int)s.pred[i] + s.rs.res[i]);
            if (l.plane == 0) s.rec_y[(l.y + 1) * RY_STRIDE + l.x + 1] = (T)v;
            else s.rec_c[l.plane - 1][(l.y + 1) * RC_STRIDE + l.x + 1] = (T)v;
            int d = (int)s.src[i] - v;
            sse += (unsigned)(d * d);
            s.coef_acc[i] = s.rs.lvl[i];
        }
        if (sse) ex.atomic_add(&s.sse, sse);
        for (int sb = tid; sb < 96; sb += NT) {       // 64 luma + 16 + 16 chroma 4x4 sub-blocks
            int pl = sb < 64 ? 0 : 1 + ((sb - 64) >> 4), k = sb < 64 ? sb : (sb - 64) & 15;
            int per = pl ? 4 : 8, bx = (k % per) * 4, by = (k / per) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0;
            int sh = pl ? 2 : 3;
            if (!s.rs.tu_log2[(by >> sh) * 4 + (bx >> sh)]) continue;
            int b = subblock_bits_q4(s.rs.lvl + base + by * stride + bx, stride);
            if (b) ex.atomic_add(&s.bits[pl], b);
        }
    });
    ex.phase([&](int tid) {
        const int mode = (int)(s.mode_key & 63);
        int t0 = (cy >> 3) * 4 + (cx >> 3);
        if (tid < 16 && s.rs.tu_log2[tid]) {
            mihevc_cu_rec r;
            r.log2_size = (uint8_t)log2n;
            r.flags = (uint8_t)(((s.rs.cbf[0] >> t0) & 1 ? CU_CBF_Y : 0) | ((s.rs.cbf[1] >> t0) & 1 ? CU_CBF_CB : 0) | ((s.rs.cbf[2] >> t0) & 1 ? CU_CBF_CR : 0));
            r.chroma_mode = (uint8_t)(s.cmode_k ? s.cmode : mode); r.qp = (uint8_t)a.prm.qp;
            r.intra_mode[0] = r.intra_mode[1] = r.intra_mode[2] = r.intra_mode[3] = (uint8_t)mode;
            r.mvx = r.mvy = 0; r.cbf_y4 = 0; r.pad[0] = r.pad[1] = r.pad[2] = 0;
            s.cu_acc[tid] = r;
        }
        if (tid == 0) {
            int mb = mode == s.cand[0] ? 2 : (mode == s.cand[1] || mode == s.cand[2]) ? 3 : 6;
            int bits = 16 * mb + 16 + 24 + (s.cmode_k ? 32 : 0);
            for (int p = 0; p < 3; p++) bits += s.bits[p] ? s.bits[p] + R_TU : 0;
            s.j_cu = ((unsigned long long)s.sse << 4) + (((unsigned long long)a.prm.lambda_q4 * (unsigned long long)bits) >> 4);
            s.nx_try = (int)((s.rs.cbf[0] >> t0) & 1);
        }
    });
}

// ------------------------------------------------------------------------------------------ NxN (four 4x4 PUs)
// part_mode NxN trial of the 8x8 CU at (cx, cy), run right after its 2Nx2N evaluation (whose result sits in the
// accumulated state): park the 2Nx2N result, code four 4x4 PUs in z-order in place, then the two 4x4 chroma TUs with
// PU 0's mode (DM), and keep NxN only if its cost is lower.  The 4x4 blocks depend on each other, so the whole trial runs
// on wave 0 as wave-local steps (ex.wave_step: no workgroup barriers): 17 lanes build the reference samples, 35 lanes rank
// the modes by 4x4-Hadamard SATD, 16 lanes (one per sample) predict, transform (DST-VII luma / DCT chroma), quantise and
// reconstruct.  Arithmetic as residual_pipeline / oracle code_tu.
template <typename T, class Ex>
DEV void intra_cu_nxn(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int x0, int y0, int cx, int cy)
{
    const int ctu_x = x0 >> CTU_LOG2, ctu_y = y0 >> CTU_LOG2, bd = a.prm.bit_depth, maxv = (1 << bd) - 1;
    const int tcn = a.prm.tile_cols > 1 ? a.prm.tile_cols : 1, trn = a.prm.tile_rows > 1 ? a.prm.tile_rows : 1;
    const int tci = tile_of(ctu_x, tcn, a.ctus_w), tri = tile_of(ctu_y, trn, a.ctus_h);
    const int tx_lo = tile_bd(tci, tcn, a.ctus_w) << CTU_LOG2, tx_hi = tile_bd(tci + 1, tcn, a.ctus_w) << CTU_LOG2;
    const int ty_lo = tile_bd(tri, trn, a.ctus_h) << CTU_LOG2;
    const int tile = (cy >> 3) * 4 + (cx >> 3);
    Nx4<T> &w = s.nx;
    // the trial of CU column c runs on wave c: wave 0 of every workgroup tends to share one SIMD, and a serial chain pinned to
    // it left the other three SIMDs of the CU idle
    const int wbase = ((cx >> 3) & 3) * 64;
    ex.phase([&](int tid) {
        if (tid < 96) {
            int pl, x, y;
            if (tid < 64) { pl = 0; x = cx + (tid & 7); y = cy + (tid >> 3); } else { int k = tid - 64; pl = 1 + (k >> 4); k &= 15; x = (cx >> 1) + (k & 3); y = (cy >> 1) + (k >> 2); }
            s.nx_rec[tid] = pl ? s.rec_c[pl - 1][(y + 1) * RC_STRIDE + x + 1] : s.rec_y[(y + 1) * RY_STRIDE + x + 1];
            s.nx_coef[tid] = s.coef_acc[(pl ? 1024 + (pl - 1) * 256 + y * 16 : y * 32) + x];
        }
        if (tid == 0) {
            s.nx_cu = s.cu_acc[tile];
            s.nx_j2n = s.j_cu;
            mihevc_cu_rec r = s.cu_acc[tile];
            r.flags = CU_NXN; r.cbf_y4 = 0;
            r.intra_mode[0] = r.intra_mode[1] = r.intra_mode[2] = r.intra_mode[3] = 1;
            s.cu_acc[tile] = r;
            s.nx_sse = 0; s.nx_bits = 16 + 24; s.nx_cbf_c = 0;
            s.mode_key = ~0ull;
            w.av[0] = w.av[1] = 0; w.nz[0] = w.nz[1] = 0;
        }
    });
    // one pass codes `nb` 4x4 blocks of the same kind side by side (lane group g = lane / 32): a luma PU (nb = 1) or Cb + Cr (nb = 2)
    auto code_blocks = [&](int nb, int pl0, int bx, int by, int k) {
        const bool luma = pl0 == 0;
        const int sh = luma ? 0 : 1, qp = luma ? a.prm.qp : a.prm.qp_c, q = qp + 6 * (bd - 8);
        const int qbits = 14 + q / 6 + (15 - bd - 2), qs = s.tab_qs[q % 6], ls = s.tab_ls[q % 6], bsh = bd + 2 - 5;
        const int16_t *M = s.nx_mat[luma ? 0 : 1];
        const int zc = zaddr(x0 + (bx << sh), y0 + (by << sh), a.ctus_w);
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // reference samples: availability (6.4.1 incl. tiles) + raw values
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 17) return;
            const int pl = pl0 + g;
            int xn, yn;
            if (i < 8) { xn = bx - 1; yn = by + 7 - i; } else if (i == 8) { xn = bx - 1; yn = by - 1; } else { xn = bx + i - 9; yn = by - 1; }
            const int lx = x0 + xn * (1 << sh), ly = y0 + yn * (1 << sh);
            const bool ok = lx >= tx_lo && ly >= ty_lo && lx < a.w && lx < tx_hi && ly < a.h && zaddr(lx, ly, a.ctus_w) < zc;
            w.ref_raw[g][i] = ok ? (pl ? s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] : s.rec_y[(yn + 1) * RY_STRIDE + xn + 1]) : (T)0;
            if (ok) ex.atomic_or(&w.av[g], 1u << i);
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // substitution 8.4.4.2.2: nearest available below, else the first available
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 17) return;
            const unsigned av = w.av[g], below = av & ((2u << i) - 1);
            w.ref[g][i] = !av ? (T)(1 << (bd - 1)) : below ? w.ref_raw[g][31 - __builtin_clz(below)] : w.ref_raw[g][__builtin_ctz(av)];
        });
        if (luma) ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;   // 35 modes ranked by SATD + MPM-aware mode bits
            if (tid >= 35) return;
            const T *ref = w.ref[0];
            int cand[3], m[16], sum = 4;
            mpm_cand(s, bx, by, x0 > tx_lo, cand);
#pragma unroll
            for (int i = 0; i < 4; i++) sum += ref_top(ref, 4, i) + ref_left(ref, 4, i);
            const int ang = s.tab_angle[tid], inv = s.tab_inv[tid], dc = sum >> 3;
#pragma unroll
            for (int i = 0; i < 16; i++) m[i] = (int)s.src[(by + (i >> 2)) * 32 + bx + (i & 3)] - intra_sample<T>(ref, 2, tid, ang, inv, i & 3, i >> 2, 0, bd, dc);
#pragma unroll
            for (int y = 0; y < 4; y++) {
                int p0 = m[y * 4] + m[y * 4 + 1], p1 = m[y * 4] - m[y * 4 + 1], p2 = m[y * 4 + 2] + m[y * 4 + 3], p3 = m[y * 4 + 2] - m[y * 4 + 3];
                m[y * 4] = p0 + p2; m[y * 4 + 1] = p1 + p3; m[y * 4 + 2] = p0 - p2; m[y * 4 + 3] = p1 - p3;
            }
            int sat = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                int p0 = m[x] + m[4 + x], p1 = m[x] - m[4 + x], p2 = m[8 + x] + m[12 + x], p3 = m[8 + x] - m[12 + x];
                sat += iabs(p0 + p2) + iabs(p1 + p3) + iabs(p0 - p2) + iabs(p1 - p3);
            }
            const int bits = tid == cand[0] ? 2 : (tid == cand[1] || tid == cand[2]) ? 3 : 6;
            const unsigned cost = ((unsigned)((sat + 1) >> 1) << 4) + (unsigned)(a.prm.lambda_sad_q4 * bits);
            ex.atomic_min(&s.mode_key, ((unsigned long long)cost << 6) | (unsigned)tid);
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // prediction + residual, one lane per sample
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 16) return;
            const int pl = pl0 + g, mode = luma ? (int)(s.mode_key & 63) : (int)s.cu_acc[tile].intra_mode[0];
            const T *ref = w.ref[g];
            int sum = 4;
#pragma unroll
            for (int j = 0; j < 4; j++) sum += ref_top(ref, 4, j) + ref_left(ref, 4, j);
            const int x = i & 3, y = i >> 2;
            const int v = intra_sample<T>(ref, 2, mode, s.tab_angle[mode], s.tab_inv[mode], x, y, pl, bd, sum >> 3);
            const int sidx = luma ? (by + y) * 32 + bx + x : 1024 + (pl - 1) * 256 + (by + y) * 16 + bx + x;
            w.pred[g][i] = (int16_t)v;
            w.res[g][i] = (int16_t)((int)s.src[sidx] - v);
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // forward stage 1 (rows)
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 16) return;
            const int u = i & 3, y = i >> 2, s1 = bd - 7;
            int acc = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) acc += M[u * 4 + x] * w.res[g][y * 4 + x];
            w.tmp[g][i] = (acc + (1 << (s1 - 1))) >> s1;
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // forward stage 2 (columns) + quantisation (intra dead zone 171/512)
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 16) return;
            const int u = i & 3, v = i >> 2;
            int acc = 0;
#pragma unroll
            for (int y = 0; y < 4; y++) acc += M[v * 4 + y] * w.tmp[g][y * 4 + u];
            const int c = clip3(-32768, 32767, (acc + 128) >> 8);
            long long l = ((long long)iabs(c) * qs + ((long long)171 << (qbits - 9))) >> qbits;
            if (l > 32767) l = 32767;
            w.lvl[g][i] = (int16_t)(c < 0 ? -(int)l : (int)l);
            if (l) ex.atomic_or(&w.nz[g], 1u);
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // scaling + inverse stage 1 (columns, 16-bit clip)
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 16 || !w.nz[g]) return;
            const int x = i & 3, y = i >> 2;
            const long long scale = (long long)16 * ls << (q / 6);
            int acc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) acc += M[j * 4 + y] * clip3(-32768, 32767, (int)((w.lvl[g][j * 4 + x] * scale + ((long long)1 << (bsh - 1))) >> bsh));
            w.tmp[g][i] = clip3(-32768, 32767, (acc + 64) >> 7);
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // inverse stage 2 (rows), reconstruction, distortion, rate, records
            const int g = tid >> 5, i = tid & 31;
            if (g >= nb || i >= 16) return;
            const int pl = pl0 + g, x = i & 3, y = i >> 2, s3 = 20 - bd, nz = (int)w.nz[g];
            int r = 0;
            if (nz) {
                int acc = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) acc += M[j * 4 + x] * w.tmp[g][y * 4 + j];
                r = (int)(int16_t)((acc + (1 << (s3 - 1))) >> s3);
            }
            const int v = clip3(0, maxv, (int)w.pred[g][i] + r);
            const int sidx = luma ? (by + y) * 32 + bx + x : 1024 + (pl - 1) * 256 + (by + y) * 16 + bx + x;
            if (luma) s.rec_y[(by + y + 1) * RY_STRIDE + bx + x + 1] = (T)v; else s.rec_c[pl - 1][(by + y + 1) * RC_STRIDE + bx + x + 1] = (T)v;
            s.coef_acc[sidx] = w.lvl[g][i];
            const int d = (int)s.src[sidx] - v, al = iabs((int)w.lvl[g][i]);
            int bits = al ? rate_level(al) : 0;
            if (i == 0) {
                if (nz) bits += R_SB + R_TU;
                if (luma) {
                    const int mode = (int)(s.mode_key & 63);
                    int cand[3];
                    mpm_cand(s, bx, by, x0 > tx_lo, cand);
                    bits += 16 * (mode == cand[0] ? 2 : (mode == cand[1] || mode == cand[2]) ? 3 : 6);
                    s.cu_acc[tile].intra_mode[k] = (uint8_t)mode;
                    if (k == 0) s.cu_acc[tile].chroma_mode = (uint8_t)mode;
                    if (nz) { s.cu_acc[tile].cbf_y4 |= (uint8_t)(1 << k); s.cu_acc[tile].flags |= CU_CBF_Y; }
                } else if (nz) {
                    ex.atomic_or(&s.nx_cbf_c, (unsigned)(pl == 1 ? CU_CBF_CB : CU_CBF_CR));
                }
            }
            if (d) ex.atomic_add(&s.nx_sse, (unsigned)(d * d));
            if (bits) ex.atomic_add(&s.nx_bits, bits);
        });
        ex.wave_step([&](int tid0) {
            const int tid = tid0 - wbase;
            if (tid < 0 || tid >= 64) return;          // reset the per-block scratch for the next block
            if (tid == 0) { s.mode_key = ~0ull; w.av[0] = w.av[1] = 0; w.nz[0] = w.nz[1] = 0; }
        });
    };
    for (int k = 0; k < 4; k++) code_blocks(1, 0, cx + (k & 1) * 4, cy + (k >> 1) * 4, k);
    code_blocks(2, 1, cx >> 1, cy >> 1, 0);
    ex.phase([&](int) {});      // the wave steps carry no barrier: the other waves wait here for the trial to finish
    ex.phase([&](int tid) {
        if (tid != 0) return;
        s.cu_acc[tile].flags |= (uint8_t)s.nx_cbf_c;
        const unsigned long long j = ((unsigned long long)s.nx_sse << 4) + (((unsigned long long)a.prm.lambda_q4 * (unsigned long long)s.nx_bits) >> 4);
        s.nx_keep = j < s.nx_j2n;
        if (s.nx_keep) s.j_cu = j;
    });
    ex.phase([&](int tid) {
        if (s.nx_keep) return;
        if (tid < 96) {
            int pl, x, y;
            if (tid < 64) { pl = 0; x = cx + (tid & 7); y = cy + (tid >> 3); } else { int k = tid - 64; pl = 1 + (k >> 4); k &= 15; x = (cx >> 1) + (k & 3); y = (cy >> 1) + (k >> 2); }
            if (pl) s.rec_c[pl - 1][(y + 1) * RC_STRIDE + x + 1] = s.nx_rec[tid]; else s.rec_y[(y + 1) * RY_STRIDE + x + 1] = s.nx_rec[tid];
            s.coef_acc[(pl ? 1024 + (pl - 1) * 256 + y * 16 : y * 32) + x] = s.nx_coef[tid];
        }
        if (tid == 0) s.cu_acc[tile] = s.nx_cu;
    });
}

// copy the region (cx,cy,n) of the accumulated state to the save area (dir = 0) or back (dir = 1)
template <typename T, class Ex> DEV void intra_save_restore(Ex &ex, IntraShared<T> &s, int cx, int cy, int n, int dir)
{
    const Region rg{cx, cy, n == 32 ? 5 : 4};
    ex.phase([&](int tid) {
        for (int k = tid; k < rg.count(); k += NT) {
            const int i = rg.index(k);
            int pl, x, y;
            if (i < 1024) { pl = 0; x = i & 31; y = i >> 5; } else { int kk = i - 1024; pl = 1 + (kk >> 8); kk &= 255; x = kk & 15; y = kk >> 4; }
            T *live = pl ? &s.rec_c[pl - 1][(y + 1) * RC_STRIDE + x + 1] : &s.rec_y[(y + 1) * RY_STRIDE + x + 1];
            T *save = pl ? &s.save_c[pl - 1][y * 16 + x] : &s.save_y[y * 32 + x];
            if (dir == 0) { *save = *live; s.coef_save[i] = s.coef_acc[i]; }
            else { *live = *save; s.coef_acc[i] = s.coef_save[i]; }
        }
        if (tid < 16) {
            int tx = (tid & 3) * 8, ty = (tid >> 2) * 8;
            if (tx >= cx && tx < cx + n && ty >= cy && ty < cy + n) {
                if (dir == 0) s.cu_save[tid] = s.cu_acc[tid]; else s.cu_acc[tid] = s.cu_save[tid];
            }
        }
    });
}

template <typename T, class Ex>
DEV void intra_ctu_program(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int ctu_x, int ctu_y)
{
    const int x0 = ctu_x * CTU, y0 = ctu_y * CTU;
    const unsigned long long lam_split = (unsigned long long)a.prm.lambda_q4;     // (lambda_q4 * 16) >> 4
    residual_init(ex, s.rs);
    ex.phase([&](int tid) {
        for (int i = tid; i < 1536; i += NT) {
            int pl, x, y;
            if (i < 1024) { pl = 0; x = i & 31; y = i >> 5; } else { int k = i - 1024; pl = 1 + (k >> 8); k &= 255; x = k & 15; y = k >> 4; }
            int gx = (pl ? x0 >> 1 : x0) + x, gy = (pl ? y0 >> 1 : y0) + y, pw = pl ? a.w >> 1 : a.w, ph = pl ? a.h >> 1 : a.h;
            s.src[i] = (gx < pw && gy < ph) ? a.src[pl].p[(size_t)gy * a.src[pl].stride + gx] : (T)0;
            s.coef_acc[i] = 0;
        }
        if (tid == 0) s.est = 0;
        if (tid >= 64 && tid < 99) { s.tab_angle[tid - 64] = (int16_t)mode_angle(tid - 64); s.tab_inv[tid - 64] = (int16_t)mode_inv_angle(tid - 64); }
        if (tid >= 128 && tid < 134) { s.tab_qs[tid - 128] = g_tab.quant_scale[tid - 128]; s.tab_ls[tid - 128] = g_tab.level_scale[tid - 128]; }
        if (tid < 32) s.nx_mat[tid >> 4][tid & 15] = tid < 16 ? g_tab.dst4[(tid >> 2) & 3][tid & 3] : g_tab.mat[((tid >> 2) & 3) * 8][tid & 3];
        if (tid < 4 && x0 > 0 && y0 + tid * 8 < a.h) s.left_cu[tid] = a.cu[(size_t)((y0 >> 3) + tid) * (a.w >> 3) + ((x0 - 1) >> 3)];
        // neighbourhood: row -1 (cols -1..63 luma / -1..31 chroma) and column -1 (rows 0..31 / 0..15) from the picture
        for (int u = tid; u < 65 + 32 + 2 * (33 + 16); u += NT) {
            int pl, k, row_len, col_len;
            if (u < 97) { pl = 0; k = u; row_len = 65; col_len = 32; }
            else { pl = 1 + (u - 97) / 49; k = (u - 97) % 49; row_len = 33; col_len = 16; }
            int xn, yn;
            if (k < row_len) { xn = k - 1; yn = -1; } else { xn = -1; yn = k - row_len; }
            (void)col_len;
            int pw = pl ? a.w >> 1 : a.w, ph = pl ? a.h >> 1 : a.h;
            int gx = (pl ? x0 >> 1 : x0) + xn, gy = (pl ? y0 >> 1 : y0) + yn;
            T v = 0;
            if (gx >= 0 && gy >= 0 && gx < pw && gy < ph) v = a.rec[pl].p[(ptrdiff_t)gy * a.rec[pl].stride + gx];
            if (pl == 0) s.rec_y[(yn + 1) * RY_STRIDE + xn + 1] = v; else s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] = v;
        }
    });
    unsigned long long j16[4];
    for (int q = 0; q < 4; q++) {
        const int qx = (q & 1) * 16, qy = (q >> 1) * 16;
        j16[q] = 0;
        if (x0 + qx >= a.w || y0 + qy >= a.h) continue;
        unsigned long long jsplit = lam_split;
        for (int b = 0; b < 4; b++) {
            const int bx = qx + (b & 1) * 8, by = qy + (b >> 1) * 8;
            if (x0 + bx >= a.w || y0 + by >= a.h) continue;
            intra_cu(ex, s, a, x0, y0, bx, by, 3);
            // NxN trial only when the 2Nx2N CU left a luma residual.  The condition must be uniform over the workgroup, so it is read from
            // a word nobody writes before the next intra_cu ends — NOT from cu_acc[].flags, which the trial's first phase rewrites while
            // slower waves may still be evaluating this line (that race made waves skip the trial and its barriers on loaded devices)
            if (a.prm.intra_nxn && s.nx_try) intra_cu_nxn(ex, s, a, x0, y0, bx, by);
            jsplit += s.j_cu;
        }
        const bool fits = x0 + qx + 16 <= a.w && y0 + qy + 16 <= a.h;
        if (!fits) { j16[q] = jsplit; continue; }
        intra_save_restore(ex, s, qx, qy, 16, 0);
        intra_cu(ex, s, a, x0, y0, qx, qy, 4);
        const unsigned long long jwhole = s.j_cu + lam_split;
        if (jwhole <= jsplit) j16[q] = jwhole;
        else { intra_save_restore(ex, s, qx, qy, 16, 1); j16[q] = jsplit; }
    }
    unsigned long long jctu = lam_split + j16[0] + j16[1] + j16[2] + j16[3];
    if (x0 + 32 <= a.w && y0 + 32 <= a.h) {
        const unsigned long long jsplit = jctu;
        intra_save_restore(ex, s, 0, 0, 32, 0);
        intra_cu(ex, s, a, x0, y0, 0, 0, 5);
        if (s.j_cu + lam_split > jsplit) intra_save_restore(ex, s, 0, 0, 32, 1);
        else jctu = s.j_cu + lam_split;
    }
    // P picture: the intra version replaces the inter one only when it is cheaper (uniform over the workgroup)
    if (a.ip && jctu >= a.ip[ctu_y * a.ctus_w + ctu_x].jinter) return;
    // the CTU is final: reconstruction, levels and CU records to memory
    ex.phase([&](int tid) {
        for (int i = 4 * tid; i < 1536; i += 4 * NT) {      // four samples of one row per lane, dword stores
            int pl, x, y;
            if (i < 1024) { pl = 0; x = i & 31; y = i >> 5; } else { int k = i - 1024; pl = 1 + (k >> 8); k &= 255; x = k & 15; y = k >> 4; }
            int gx = (pl ? x0 >> 1 : x0) + x, gy = (pl ? y0 >> 1 : y0) + y, pw = pl ? a.w >> 1 : a.w, ph = pl ? a.h >> 1 : a.h;
            if (gx >= pw || gy >= ph) continue;            // widths are multiples of 4 in both planes: a quad is inside or outside
            const T *r = pl ? &s.rec_c[pl - 1][(y + 1) * RC_STRIDE + x + 1] : &s.rec_y[(y + 1) * RY_STRIDE + x + 1];
            store4(a.rec[pl].p + (ptrdiff_t)gy * a.rec[pl].stride + gx, r[0], r[1], r[2], r[3]);
            {
                const int sh = pl ? 2 : 3, fl = s.cu_acc[(y >> sh) * 4 + (x >> sh)].flags;
                if (!a.sparse_coef || (fl & (pl == 0 ? CU_CBF_Y : pl == 1 ? CU_CBF_CB : CU_CBF_CR)))
                    store4(a.coef[pl] + (size_t)gy * pw + gx, s.coef_acc[i], s.coef_acc[i + 1], s.coef_acc[i + 2], s.coef_acc[i + 3]);
            }
        }
        if (tid < 16) {
            int tx = (tid & 3) * 8, ty = (tid >> 2) * 8;
            if (x0 + tx < a.w && y0 + ty < a.h) a.cu[(size_t)((y0 + ty) >> 3) * (a.w >> 3) + ((x0 + tx) >> 3)] = s.cu_acc[tid];
        }
        if (a.est) {       // rate estimate of the final CTU: coefficient sub-block costs + 8 bits of header per CU
            unsigned e = 0;
            for (int sb = tid; sb < 96; sb += NT) {
                int pl = sb < 64 ? 0 : 1 + ((sb - 64) >> 4), k = sb < 64 ? sb : (sb - 64) & 15;
                int per = pl ? 4 : 8, bx = (k % per) * 4, by = (k / per) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0;
                int lx = pl ? bx * 2 : bx, ly = pl ? by * 2 : by;
                if (x0 + lx < a.w && y0 + ly < a.h) e += (unsigned)subblock_bits_q4(s.coef_acc + base + by * stride + bx, stride);
            }
            if (tid < 16) {
                int tx = (tid & 3) * 8, ty = (tid >> 2) * 8;
                const mihevc_cu_rec &r = s.cu_acc[tid];
                if (x0 + tx < a.w && y0 + ty < a.h && !(tx & ((1 << r.log2_size) - 1)) && !(ty & ((1 << r.log2_size) - 1))) {
                    int ncbf = ((r.flags & CU_CBF_CB) != 0) + ((r.flags & CU_CBF_CR) != 0);
                    if (r.flags & CU_NXN) { for (int k = 0; k < 4; k++) ncbf += (r.cbf_y4 >> k) & 1; } else ncbf += (r.flags & CU_CBF_Y) != 0;
                    e += R_INTRA_CU + (unsigned)(R_TU * ncbf);
                }
            }
            if (e) ex.atomic_add(&s.est, e);
        }
    });
    // in a P picture the CTU's inter estimate is already in the picture total: add the difference (modulo 2^64)
    if (a.est) ex.phase([&](int tid) { if (tid == 0) ex.atomic_add_global(a.est, (unsigned long long)s.est - (a.ip ? (unsigned long long)a.ip[ctu_y * a.ctus_w + ctu_x].est : 0ull)); });
}

}  // namespace mihevc
