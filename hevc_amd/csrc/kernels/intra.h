// hevc_amd/csrc/kernels/intra.h — K2: intra prediction + mode decision (35 modes, SATD) + K3 residual for the CTUs
// of an I picture (and of the P pictures' intra second pass), in two stages (oracle/hevc_oracle.c intra_plan_ctu / intra_code_ctu):
//
//   A. PLAN (intra_plan_program, k_intra_plan): luma / chroma modes of all 21 quadtree nodes and the quadtree itself, decided on the
//      SOURCE picture — the source neighbours stand in for the reconstruction under the real availability rules (picture, tile,
//      z-order), node costs are real RD costs (K3 residual coding, SSE + lambda x estimated bits).  Nothing depends on another CTU, so
//      one launch plans every CTU of every picture in flight.
//   B. CODE (intra_code_program, k_intra_diag): the planned CUs only, in decoding order, predicted from the real reconstruction of
//      their left / top-left / top / top-right neighbours: CTUs are launched one anti-diagonal (x + 2y = d) at a time; with a tile grid
//      (prm.tile_cols x tile_rows, PPS 1) prediction stops at tile boundaries, so every tile runs its own wavefront and the launch
//      count drops from W + 2(H-1) to w + 2(h-1) CTUs of one tile.
//
// Round 1 searched modes and tree depth first on the reconstruction inside the wavefront (21 CUs x ~16 barrier phases per CTU program,
// 0.6 ms of latency per anti-diagonal launch: 23 % of the device time for 4 of 300 pictures); the plan costs +0.07 % bits at -0.003 dB on
// the bench clip's IDR pictures (QP 23) against that search.  The CTU's reconstruction, levels and CU records live in LDS until the CTU
// is final.  Prediction is H.265 8.4.4.2 (reference availability by z-scan order, substitution, [1 2 1] / strong smoothing,
// planar, DC, angular with the boundary filters); MPM list per 8.4.2.
#pragma once
#include "common.h"
#include "residual.h"

namespace mihevc {

struct IntraPlan;

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
    struct IntraPlan *plan;      // per CTU: k_intra_plan -> k_intra_diag (I pictures); unused when one workgroup runs both stages
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

// what the plan stage hands to the code stage for one CTU: the quadtree's leaves and their modes (k_intra_plan -> k_intra_diag through
// global memory in I pictures; inside one workgroup in the P pictures' second pass)
struct IntraPlan {
    uint8_t chosen[21], mode[21], cmode[21], pad;
};

template <typename T> struct IntraShared {
    ResidualShared rs;
    T src[1536];
    T pred[1536];
    T rec_y[33 * RY_STRIDE];     // neighbourhood image: the SOURCE during the plan stage, the reconstruction during the code stage
    T rec_c[2][17 * RC_STRIDE];
    mihevc_cu_rec cu_acc[16];
    mihevc_cu_rec left_cu[4];    // CU records of the left CTU's right column (MPM derivation), fetched once per CTU
    IntraPlan plan;
    int16_t tab_angle[35], tab_inv[35];   // Tables 8-4 / 8-5 by mode, LDS copies: a global read per use sat on the serial chain
    int16_t tab_qs[6], tab_ls[6];
    int16_t nx_mat[2][16];       // 4x4 DST-VII and DCT matrices [k * 4 + n]
    unsigned est;
    union {      // the two stages never need each other's working state: one workgroup runs them one after the other at most
    struct {
    // ---- code stage: one CU at a time
    int16_t coef_acc[1536];
    T ref_raw[3][132], ref[3][132], filt[132];   // [plane]: 4N+1 reference samples (raw, substituted); filtered luma
    unsigned avmask[3][5];       // [plane]: bit i = reference sample i is available (zeroed by the previous CU's last phase / the CTU's first)
    unsigned long long mode_key; // NxN trial: min over modes of (cost << 6 | mode)
    int cand[3], dc_val[3];
    unsigned sse;
    int bits[3];
    unsigned long long j_cu, j_ctu;
    // NxN trial of an 8x8 CU: the 2Nx2N result parked here while four 4x4 PUs are coded in place
    T nx_rec[96];                // Y 8x8, Cb 4x4, Cr 4x4
    int16_t nx_coef[96];
    mihevc_cu_rec nx_cu;
    unsigned long long nx_j2n;
    unsigned nx_sse;
    int nx_bits, nx_keep;
    int nx_try;                  // set by intra_code_cu: the 2Nx2N CU left a luma residual (the NxN trial's condition)
    unsigned nx_cbf_c;           // CU_CBF_CB / CU_CBF_CR of the NxN trial
    Nx4<T> nx;
    };
    struct {
    // ---- plan stage: all 21 quadtree nodes at once
    T p_ref[21][132], p_filt[21][132];           // luma reference samples of every node (substituted; filtered)
    T p_refc[21][2][68];                         // chroma reference samples
    unsigned p_av[21][3][5];
    int p_dc[21][3];
    unsigned p_msum[21][35];                     // luma SATD of every (node, mode), summed over the node's 8x8 tiles
    unsigned long long p_key[21];                // min over modes of (cost << 6 | mode)
    unsigned p_craw[160];                        // chroma candidates of one level: [node][k][plane][tile] sum |H d H|
    int16_t p_crow[2560];                        // the same: row-transformed difference lines
    uint8_t p_valid[21], p_mode[21], p_cmode[21];
    unsigned p_sse[21];
    unsigned p_bits[21][3];
    unsigned long long p_j[21], p_j16[4];
    int p_use16[4], p_use32;
    };
    };
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

// tile of a CTU as luma sample bounds: neighbours outside it are unavailable (6.4.1); one tile = the whole picture
struct TileBox { int x_lo, x_hi, y_lo; };
template <typename T> DEV TileBox tile_box(const IntraArgs<T> &a, int ctu_x, int ctu_y)
{
    const int tcn = a.prm.tile_cols > 1 ? a.prm.tile_cols : 1, trn = a.prm.tile_rows > 1 ? a.prm.tile_rows : 1;
    const int tci = tile_of(ctu_x, tcn, a.ctus_w), tri = tile_of(ctu_y, trn, a.ctus_h);
    return TileBox{tile_bd(tci, tcn, a.ctus_w) << CTU_LOG2, tile_bd(tci + 1, tcn, a.ctus_w) << CTU_LOG2, tile_bd(tri, trn, a.ctus_h) << CTU_LOG2};
}
// position, in its plane, of reference sample i of a block at (px, py) with np samples per side
DEV void ref_pos(int px, int py, int np, int i, int &xn, int &yn)
{
    if (i < 2 * np) { xn = px - 1; yn = py + 2 * np - 1 - i; }
    else if (i == 2 * np) { xn = px - 1; yn = py - 1; }
    else { xn = px + (i - 2 * np - 1); yn = py - 1; }
}
// 8.4.4.2.2 substitution source of sample i: the nearest available sample at a lower index, else the first available above, else
// `total` (nothing available).  Bit scans over the availability words instead of a sample-by-sample walk: a bottom-left run of 2N
// unavailable samples made this a chain of up to 64 dependent LDS reads.
DEV int ref_source(const unsigned *m, int i, int total)
{
    const unsigned upto = (2u << (i & 31)) - 1u;           // bits 0..(i & 31); all ones when (i & 31) == 31
    int w = i >> 5;
    unsigned bits = m[w] & upto;
    for (;;) {
        if (bits) return 32 * w + 31 - __builtin_clz(bits);
        if (--w < 0) break;
        bits = m[w];
    }
    w = i >> 5;
    bits = m[w] & ~upto;
    for (;;) {
        if (bits) return 32 * w + __builtin_ctz(bits);
        if (++w > 4) break;
        bits = m[w];
    }
    return total;
}
DEV int level_first(int level) { return level == 0 ? 0 : level == 1 ? 1 : 5; }      // level 0 / 1 / 2 = the 32x32 / 16x16 / 8x8 nodes
DEV int mode_bits_for(const int (&cand)[3], int mode) { return mode == cand[0] ? 2 : (mode == cand[1] || mode == cand[2]) ? 3 : 6; }
DEV void cand_from(int ma, int mb, int (&cand)[3])       // 8.4.2 candModeList from the left (ma) and above (mb) modes
{
    if (ma == mb) {
        if (ma < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
        else { cand[0] = ma; cand[1] = 2 + ((ma + 29) & 31); cand[2] = 2 + ((ma - 2 + 1) & 31); }
    } else {
        cand[0] = ma; cand[1] = mb;
        cand[2] = (ma != 0 && mb != 0) ? 0 : (ma != 1 && mb != 1) ? 1 : 26;
    }
}
// the plan's candidate list of a node: planned modes of the same-level nodes left of and above it inside the CTU, DC outside
template <typename T> DEV void plan_cand(const IntraShared<T> &s, int node, int level, int (&cand)[3])
{
    int cx, cy, l2;
    node_geom(node, cx, cy, l2);
    const int n = 1 << l2;
    const int ma = cx > 0 ? (int)(s.p_key[node_of_tile(level, (cx - n) >> 3, cy >> 3)] & 63) : 1;
    const int mb = cy > 0 ? (int)(s.p_key[node_of_tile(level, cx >> 3, (cy - n) >> 3)] & 63) : 1;
    cand_from(ma, mb, cand);
}

// ------------------------------------------------------------------------------------------ stage A: the plan
// oracle/hevc_oracle.c intra_plan_ctu: every quadtree node of the CTU is costed on the SOURCE picture (its neighbours stand in for the
// reconstruction, under the real availability rules), so the CTUs of a picture are independent: one launch plans them all, and the
// 21 nodes of a CTU are worked on side by side — reference samples of all nodes, then 1680 (node, mode, tile) SATD units over the 256
// lanes, the MPM-aware mode pick (a z-order chain per level, run by one wave per level), and per level one pass of prediction, K3
// residual coding of the WHOLE CTU and distortion + rate per node; the tree is then decided bottom-up on those RD costs.
template <typename T, class Ex>
DEV void intra_plan_program(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int ctu_x, int ctu_y)
{
    const int x0 = ctu_x * CTU, y0 = ctu_y * CTU, bd = a.prm.bit_depth;
    const TileBox tb = tile_box(a, ctu_x, ctu_y);
    residual_init(ex, s.rs);
    ex.phase([&](int tid) {
        load_ctu_source<T>(s.src, a.src, x0, y0, a.w, a.h, tid);
        if (tid >= 64 && tid < 99) { s.tab_angle[tid - 64] = (int16_t)mode_angle(tid - 64); s.tab_inv[tid - 64] = (int16_t)mode_inv_angle(tid - 64); }
        if (tid >= 128 && tid < 134) { s.tab_qs[tid - 128] = g_tab.quant_scale[tid - 128]; s.tab_ls[tid - 128] = g_tab.level_scale[tid - 128]; }
        if (tid < 32) s.nx_mat[tid >> 4][tid & 15] = tid < 16 ? g_tab.dst4[(tid >> 2) & 3][tid & 3] : g_tab.mat[((tid >> 2) & 3) * 8][tid & 3];
        if (tid >= 160 && tid < 181) {
            int nx, ny, nl;
            node_geom(tid - 160, nx, ny, nl);
            s.p_valid[tid - 160] = x0 + nx + (1 << nl) <= a.w && y0 + ny + (1 << nl) <= a.h;
            s.p_key[tid - 160] = ~0ull;
        }
        for (int i = tid; i < 21 * 35; i += NT) s.p_msum[i / 35][i % 35] = 0;
        for (int i = tid; i < 21 * 15; i += NT) s.p_av[i / 15][(i % 15) / 5][i % 5] = 0;
        // source neighbourhood ring: row -1 (cols -1..63 luma / -1..31 chroma) and column -1 (rows 0..31 / 0..15)
        for (int u = tid; u < 65 + 32 + 2 * (33 + 16); u += NT) {
            int pl, k, row_len;
            if (u < 97) { pl = 0; k = u; row_len = 65; } else { pl = 1 + (u - 97) / 49; k = (u - 97) % 49; row_len = 33; }
            int xn, yn;
            if (k < row_len) { xn = k - 1; yn = -1; } else { xn = -1; yn = k - row_len; }
            const int pw = pl ? a.w >> 1 : a.w, ph = pl ? a.h >> 1 : a.h, gx = (pl ? x0 >> 1 : x0) + xn, gy = (pl ? y0 >> 1 : y0) + yn;
            T v = 0;
            if (gx >= 0 && gy >= 0 && gx < pw && gy < ph) v = a.src[pl].p[(ptrdiff_t)gy * a.src[pl].stride + gx];
            if (pl == 0) s.rec_y[(yn + 1) * RY_STRIDE + xn + 1] = v; else s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] = v;
        }
    });
    ex.phase([&](int tid) {      // the CTU's own source samples complete the neighbourhood image
        for (int i = tid; i < 1536; i += NT) {
            if (i < 1024) s.rec_y[((i >> 5) + 1) * RY_STRIDE + (i & 31) + 1] = s.src[i];
            else { const int k = (i - 1024) & 255; s.rec_c[(i - 1024) >> 8][((k >> 4) + 1) * RC_STRIDE + (k & 15) + 1] = s.src[i]; }
        }
    });
    // reference samples of every node and plane: availability + raw values, substitution (in place: an available sample keeps its value
    // and only those are read), smoothing + DC
    auto for_refs = [&](int tid, auto &&f) {
        for (int level = 0; level < 3; level++) {
            const int n = 32 >> level, ty = 4 * n + 1, tc = 2 * n + 1, per = ty + 2 * tc, cnt = 1 << (2 * level);
            for (int u = tid; u < cnt * per; u += NT) {
                const int r = u % per, nd = level_first(level) + u / per;
                if (!s.p_valid[nd]) continue;
                const int pl = r < ty ? 0 : r < ty + tc ? 1 : 2, i = pl == 0 ? r : pl == 1 ? r - ty : r - ty - tc;
                f(nd, n, pl, i);
            }
        }
    };
    ex.phase([&](int tid) {
        for_refs(tid, [&](int nd, int n, int pl, int i) {
            int cx, cy, l2, xn, yn;
            node_geom(nd, cx, cy, l2);
            ref_pos(pl ? cx >> 1 : cx, pl ? cy >> 1 : cy, pl ? n >> 1 : n, i, xn, yn);
            const int sh = pl ? 1 : 0, lx = xn * (1 << sh) + x0, ly = yn * (1 << sh) + y0;
            const bool ok = lx >= tb.x_lo && ly >= tb.y_lo && lx < a.w && lx < tb.x_hi && ly < a.h && zaddr(lx, ly, a.ctus_w) < zaddr(x0 + cx, y0 + cy, a.ctus_w);
            const T v = ok ? (pl ? s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] : s.rec_y[(yn + 1) * RY_STRIDE + xn + 1]) : (T)0;
            if (pl) s.p_refc[nd][pl - 1][i] = v; else s.p_ref[nd][i] = v;
            if (ok) ex.atomic_or(&s.p_av[nd][pl][i >> 5], 1u << (i & 31));
        });
    });
    ex.phase([&](int tid) {
        for_refs(tid, [&](int nd, int n, int pl, int i) {
            const int total = (pl ? 2 * n : 4 * n) + 1, j = ref_source(s.p_av[nd][pl], i, total);
            if (j == i) return;
            T *R = pl ? s.p_refc[nd][pl - 1] : s.p_ref[nd];
            R[i] = j < total ? R[j] : (T)(1 << (bd - 1));
        });
    });
    ex.phase([&](int tid) {
        for (int level = 0; level < 3; level++) {
            const int n = 32 >> level, total = 4 * n + 1, cnt = 1 << (2 * level);
            for (int u = tid; u < cnt * total; u += NT) {
                const int i = u % total, nd = level_first(level) + u / total;
                if (!s.p_valid[nd]) continue;
                const T *L = s.p_ref[nd];
                int v;
                if (i == 0 || i == total - 1) v = L[i];
                else if (n == 32 && iabs(L[64] + L[128] - 2 * L[96]) < (1 << (bd - 5)) && iabs(L[64] + L[0] - 2 * L[32]) < (1 << (bd - 5)))
                    v = i == 64 ? L[64] : i < 64 ? (i * L[64] + (64 - i) * L[0] + 32) >> 6 : ((128 - i) * L[64] + (i - 64) * L[128] + 32) >> 6;
                else v = (L[i - 1] + 2 * L[i] + L[i + 1] + 2) >> 2;
                s.p_filt[nd][i] = (T)v;
            }
        }
        for (int u = tid; u < 63; u += NT) {
            const int nd = u / 3, pl = u % 3;
            if (!s.p_valid[nd]) continue;
            int cx, cy, l2;
            node_geom(nd, cx, cy, l2);
            const int np = pl ? 1 << (l2 - 1) : 1 << l2, lg = pl ? l2 - 1 : l2;
            const T *L = pl ? s.p_refc[nd][pl - 1] : s.p_ref[nd];
            int sum = np;
            for (int i = 0; i < np; i++) sum += ref_top(L, np, i) + ref_left(L, np, i);
            s.p_dc[nd][pl] = sum >> (lg + 1);
        }
    });
    // 35 modes x every 8x8 tile of every node: 3 x 560 SATD units
    ex.phase([&](int tid) {
        for (int u = tid; u < 3 * 560; u += NT) {
            const int level = 2 - u / 560, v = u % 560;        // the 8x8 level first: its lanes diverge least
            const int k = level == 2 ? v / 35 : level == 1 ? v / 140 : 0, mode = level == 2 ? v % 35 : level == 1 ? (v % 140) >> 2 : v >> 4;
            const int t = level == 2 ? 0 : level == 1 ? v & 3 : v & 15, nd = level_first(level) + k;
            if (!s.p_valid[nd]) continue;
            int cx, cy, l2;
            node_geom(nd, cx, cy, l2);
            const int tiles = 1 << (l2 - 3), tx = (t % tiles) * 8, ty = (t / tiles) * 8;
            const T *L = intra_filter_on(l2, mode) ? s.p_filt[nd] : s.p_ref[nd];
            int m[8][8];
            intra_tile_diff<T>(L, l2, mode, s.tab_angle[mode], s.tab_inv[mode], tx, ty, bd, s.p_dc[nd][0], s.src + (cy + ty) * 32 + cx + tx, 32, m);
            ex.atomic_add(&s.p_msum[nd][mode], (unsigned)hadamard8_satd(m));
        }
    });
    // mode pick: inside a level a node's candidate list needs its left / top neighbours' picks, so each level is a z-order chain; wave w
    // runs level 2 - w with wave-local steps (35 lanes = the modes), all three chains side by side
    for (int i = 0; i < 16; i++)
        ex.wave_step([&](int tid) {
            const int level = 2 - (tid >> 6), mode = tid & 63;
            if (level < 0 || mode >= 35 || i >= (1 << (2 * level))) return;
            const int nd = level_first(level) + i;
            if (!s.p_valid[nd]) return;
            int cand[3];
            plan_cand(s, nd, level, cand);
            const unsigned cost = (s.p_msum[nd][mode] << 4) + (unsigned)(a.prm.lambda_sad_q4 * mode_bits_for(cand, mode));
            ex.atomic_min(&s.p_key[nd], ((unsigned long long)cost << 6) | (unsigned)mode);      // ties -> lowest mode
        });
    ex.phase([&](int) {});       // the wave steps carry no barrier: everybody waits here for the three chains
    for (int level = 0; level < 3; level++) {
        const int log2n = 5 - level, n = 1 << log2n, nc = n >> 1, l2c = log2n - 1, tw = nc >= 8 ? 8 : 4, ct = nc / tw, ntc = ct * ct;
        const int cnt = 1 << (2 * level), first = level_first(level);
        // intra_chroma_pred_mode: DM or planar / 26 / 10 / DC (a candidate equal to the luma mode stands for 34) by SATD over Cb + Cr +
        // lambda * (1 bit DM, 3 bits otherwise), DM wins ties; every tile split into lines: one lane predicts a row and transforms it,
        // a second phase finishes one column each
        if (a.prm.chroma_modes) {
            const int nline = cnt * 5 * 2 * ntc * tw;
            ex.phase([&](int tid) {
                for (int u = tid; u < nline; u += NT) {
                    const int y = u % tw, t = (u / tw) % ntc, pl = 1 + (u / (tw * ntc)) % 2, k = (u / (tw * ntc * 2)) % 5, nk = u / (tw * ntc * 10), nd = first + nk;
                    if (!s.p_valid[nd]) continue;
                    const int mode = (int)(s.p_key[nd] & 63), base = k == 1 ? 0 : k == 2 ? 26 : k == 3 ? 10 : 1, m = k == 0 ? mode : (base == mode ? 34 : base);
                    int cx, cy, l2;
                    node_geom(nd, cx, cy, l2);
                    const T *L = s.p_refc[nd][pl - 1];
                    const int tx = (t % ct) * tw, ty = (t / ct) * tw, ang = s.tab_angle[m], inv = s.tab_inv[m];
                    const int sbase = 1024 + (pl - 1) * 256 + ((cy >> 1) + ty + y) * 16 + (cx >> 1) + tx;
                    int d[8];
#pragma unroll
                    for (int x = 0; x < 8; x++) d[x] = x < tw ? (int)s.src[sbase + x] - intra_sample<T>(L, l2c, m, ang, inv, tx + x, ty + y, pl, bd, s.p_dc[nd][pl]) : 0;
                    if (tw == 8) {
#pragma unroll
                        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                if (!(i & st)) { int p = d[i], q = d[i + st]; d[i] = p + q; d[i + st] = p - q; }
                    } else {
                        const int p0 = d[0] + d[1], p1 = d[0] - d[1], p2 = d[2] + d[3], p3 = d[2] - d[3];
                        d[0] = p0 + p2; d[1] = p1 + p3; d[2] = p0 - p2; d[3] = p1 - p3;
                    }
                    int16_t *o = s.p_crow + ((((nk * 5 + k) * 2 + (pl - 1)) * ntc + t) * tw + y) * tw;
#pragma unroll
                    for (int x = 0; x < 8; x++) if (x < tw) o[x] = (int16_t)d[x];
                }
                for (int i = tid; i < 160; i += NT) s.p_craw[i] = 0;
            });
            ex.phase([&](int tid) {
                for (int u = tid; u < nline; u += NT) {
                    const int x = u % tw, t = (u / tw) % ntc, pl = (u / (tw * ntc)) % 2, k = (u / (tw * ntc * 2)) % 5, nk = u / (tw * ntc * 10);
                    if (!s.p_valid[first + nk]) continue;
                    const int16_t *c = s.p_crow + (((nk * 5 + k) * 2 + pl) * ntc + t) * tw * tw + x;
                    int sum;
                    if (tw == 8) {
                        int d[8];
#pragma unroll
                        for (int y = 0; y < 8; y++) d[y] = c[y * 8];
#pragma unroll
                        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                if (!(i & st)) { int p = d[i], q = d[i + st]; d[i] = p + q; d[i + st] = p - q; }
                        sum = 0;
#pragma unroll
                        for (int y = 0; y < 8; y++) sum += iabs(d[y]);
                    } else {
                        const int p0 = c[0] + c[4], p1 = c[0] - c[4], p2 = c[8] + c[12], p3 = c[8] - c[12];
                        sum = iabs(p0 + p2) + iabs(p1 + p3) + iabs(p0 - p2) + iabs(p1 - p3);
                    }
                    ex.atomic_add(&s.p_craw[((nk * 5 + k) * 2 + pl) * ntc + t], (unsigned)sum);
                }
            });
        }
        ex.phase([&](int tid) {
            if (tid < cnt && s.p_valid[first + tid]) {
                const int nd = first + tid, mode = (int)(s.p_key[nd] & 63);
                int cmode = mode;
                if (a.prm.chroma_modes) {
                    unsigned long long best = ~0ull;
                    for (int k = 0; k < 5; k++) {
                        unsigned satd = 0;          // per tile (sum |H d H| + 2) >> 2 for 8x8 tiles, (sum + 1) >> 1 for the 4x4 blocks
                        for (int p = 0; p < 2 * ntc; p++) { const unsigned r = s.p_craw[(tid * 5 + k) * 2 * ntc + p]; satd += tw == 8 ? (r + 2) >> 2 : (r + 1) >> 1; }
                        const unsigned long long key = ((unsigned long long)((satd << 4) + (unsigned)(a.prm.lambda_sad_q4 * (k == 0 ? 1 : 3))) << 3) | (unsigned)k;
                        best = key < best ? key : best;
                    }
                    const int k = (int)(best & 7), base = k == 1 ? 0 : k == 2 ? 26 : k == 3 ? 10 : 1;
                    if (k) cmode = base == mode ? 34 : base;
                }
                s.p_mode[nd] = (uint8_t)mode; s.p_cmode[nd] = (uint8_t)cmode;
                s.p_sse[nd] = 0; s.p_bits[nd][0] = s.p_bits[nd][1] = s.p_bits[nd][2] = 0;
            }
            if (tid >= 64 && tid < 80) {
                const int t = tid - 64;
                s.rs.tu_log2[t] = s.p_valid[node_of_tile(level, t & 3, t >> 2)] ? (uint8_t)log2n : 0;
                s.rs.tu_intra[t] = 1;
            }
            if (tid >= 96 && tid < 99) s.rs.cbf[tid - 96] = 0;
        });
        // RD cost of every node of the level with its modes: prediction (still from the source neighbourhood), K3 over the whole CTU,
        // distortion and rate per node
        ex.phase([&](int tid) {
            for (int i = tid; i < 1536; i += NT) {
                SampleLoc l = locate(s.rs, i);
                s.rs.desc[i] = pack_loc(l);
                if (!l.log2n) continue;
                const int sh = l.plane ? 2 : 3, nd = node_of_tile(level, l.x >> sh, l.y >> sh);
                int cx, cy, l2, v;
                node_geom(nd, cx, cy, l2);
                if (l.plane == 0) {
                    const int mode = s.p_mode[nd];
                    const T *L = intra_filter_on(log2n, mode) ? s.p_filt[nd] : s.p_ref[nd];
                    v = intra_sample<T>(L, log2n, mode, s.tab_angle[mode], s.tab_inv[mode], l.x - cx, l.y - cy, 0, bd, s.p_dc[nd][0]);
                } else {
                    const int cm = s.p_cmode[nd];
                    v = intra_sample<T>(s.p_refc[nd][l.plane - 1], l2c, cm, s.tab_angle[cm], s.tab_inv[cm], l.x - (cx >> 1), l.y - (cy >> 1), l.plane, bd, s.p_dc[nd][l.plane]);
                }
                s.pred[i] = (T)v;
                s.rs.res[i] = (int16_t)((int)s.src[i] - v);
            }
        });
        residual_pipeline(ex, s.rs, a.prm.qp, a.prm.qp_c, bd, whole_ctu());
        ex.phase([&](int tid) {
            const int maxv = (1 << bd) - 1;
            for (int i = 4 * tid; i < 1536; i += 4 * NT) {
                SampleLoc l = locate(s.rs, i);
                if (!l.log2n) continue;
                const int sh = l.plane ? 2 : 3;
                unsigned sse = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { const int d = (int)s.src[i + j] - clip3(0, maxv, (int)s.pred[i + j] + s.rs.res[i + j]); sse += (unsigned)(d * d); }
                if (sse) ex.atomic_add(&s.p_sse[node_of_tile(level, l.x >> sh, l.y >> sh)], sse);
            }
            for (int sb = tid; sb < 96; sb += NT) {       // 64 luma + 16 + 16 chroma 4x4 sub-blocks
                const int pl = sb < 64 ? 0 : 1 + ((sb - 64) >> 4), k = sb < 64 ? sb : (sb - 64) & 15;
                const int per = pl ? 4 : 8, bx = (k & (per - 1)) * 4, by = (k >> (pl ? 2 : 3)) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0, sh = pl ? 2 : 3;
                const int tile = (by >> sh) * 4 + (bx >> sh);
                if (!s.rs.tu_log2[tile]) continue;
                const int b = subblock_bits_q4(s.rs.lvl + base + by * stride + bx, stride);
                if (b) ex.atomic_add(&s.p_bits[node_of_tile(level, tile & 3, tile >> 2)][pl], (unsigned)b);
            }
        });
        ex.phase([&](int tid) {
            if (tid >= cnt) return;
            const int nd = first + tid;
            if (!s.p_valid[nd]) { s.p_j[nd] = 0; return; }
            int cand[3];
            plan_cand(s, nd, level, cand);
            unsigned bits = 16u * (unsigned)mode_bits_for(cand, s.p_mode[nd]) + 16 + 24 + (s.p_cmode[nd] != s.p_mode[nd] ? 32 : 0);
            for (int p = 0; p < 3; p++) bits += s.p_bits[nd][p] ? s.p_bits[nd][p] + R_TU : 0;
            s.p_j[nd] = ((unsigned long long)s.p_sse[nd] << 4) + (((unsigned long long)a.prm.lambda_q4 * (unsigned long long)bits) >> 4);
        });
    }
    // the tree, bottom-up (wave 0): split = lambda + children, whole = own + lambda, whole wins ties; a node that does not fit is split
    const unsigned long long lam_split = (unsigned long long)a.prm.lambda_q4;     // (lambda_q4 * 16) >> 4
    ex.wave_step([&](int tid) {
        if (tid >= 4) return;
        const int q = tid;
        unsigned long long js = lam_split;
        int any = 0;
        for (int k = 0; k < 4; k++) if (s.p_valid[5 + 4 * q + k]) { js += s.p_j[5 + 4 * q + k]; any = 1; }
        const unsigned long long jw = s.p_j[1 + q] + lam_split;
        s.p_use16[q] = s.p_valid[1 + q] && jw <= js;
        s.p_j16[q] = !any ? 0 : s.p_use16[q] ? jw : js;
    });
    ex.wave_step([&](int tid) {
        if (tid != 0) return;
        s.p_use32 = s.p_valid[0] && s.p_j[0] + lam_split <= lam_split + s.p_j16[0] + s.p_j16[1] + s.p_j16[2] + s.p_j16[3];
    });
    ex.phase([&](int tid) {
        if (tid >= 21) return;
        const int nd = tid;
        s.plan.chosen[nd] = (uint8_t)(s.p_valid[nd] && (nd == 0 ? s.p_use32 : nd < 5 ? (!s.p_use32 && s.p_use16[nd - 1]) : (!s.p_use32 && !s.p_use16[(nd - 5) >> 2])));
        s.plan.mode[nd] = s.p_mode[nd]; s.plan.cmode[nd] = s.p_cmode[nd];
        if (a.plan) {       // I pictures: k_intra_diag picks it up from memory
            IntraPlan &o = a.plan[ctu_y * a.ctus_w + ctu_x];
            o.chosen[nd] = s.plan.chosen[nd]; o.mode[nd] = s.p_mode[nd]; o.cmode[nd] = s.p_cmode[nd];
        }
    });
}

// ------------------------------------------------------------------------------------------ stage B: coding the planned CUs
// one planned 2Nx2N CU (oracle intra_cu): reference samples from the reconstruction, prediction with the planned modes, K3, reconstruction
// into the LDS neighbourhood, distortion + rate, records
template <typename T, class Ex>
DEV void intra_code_cu(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int x0, int y0, const TileBox &tb, int cx, int cy, int log2n, int mode, int cmode)
{
    const int n = 1 << log2n, bd = a.prm.bit_depth;
    const Region rg{cx, cy, log2n};
    const int rcnt = rg.count();
    const int gx = x0 + cx, gy = y0 + cy;        // picture coordinates of the CU
    ex.phase([&](int tid) {      // reference samples: availability + raw values for the three planes
        for (int u = tid; u < 3 * 129; u += NT) {
            const int pl = u / 129, i = u % 129, np = pl ? n >> 1 : n;
            if (i >= 4 * np + 1) continue;
            int xn, yn;
            ref_pos(pl ? cx >> 1 : cx, pl ? cy >> 1 : cy, np, i, xn, yn);
            const int sh = pl ? 1 : 0, lx = xn * (1 << sh) + x0, ly = yn * (1 << sh) + y0;      // luma picture position of the neighbour
            const bool ok = lx >= tb.x_lo && ly >= tb.y_lo && lx < a.w && lx < tb.x_hi && ly < a.h && zaddr(lx, ly, a.ctus_w) < zaddr(gx, gy, a.ctus_w);
            if (ok) ex.atomic_or(&s.avmask[pl][i >> 5], 1u << (i & 31));
            s.ref_raw[pl][i] = ok ? (pl ? s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] : s.rec_y[(yn + 1) * RY_STRIDE + xn + 1]) : (T)0;
        }
        if (tid == 0) {   // 8.4.2 candModeList from the CUs actually coded around this one (rate estimate only)
            int cand[3];
            mpm_cand(s, cx, cy, x0 > tb.x_lo, cand);
            s.cand[0] = cand[0]; s.cand[1] = cand[1]; s.cand[2] = cand[2];
            s.sse = 0; s.bits[0] = s.bits[1] = s.bits[2] = 0;
        }
    });
    ex.phase([&](int tid) {      // substitution (8.4.4.2.2)
        for (int u = tid; u < 3 * 129; u += NT) {
            const int pl = u / 129, i = u % 129, np = pl ? n >> 1 : n, total = 4 * np + 1;
            if (i >= total) continue;
            const int j = ref_source(s.avmask[pl], i, total);
            s.ref[pl][i] = j < total ? s.ref_raw[pl][j] : (T)(1 << (bd - 1));
        }
    });
    ex.phase([&](int tid) {      // smoothing filter for luma (8.4.4.2.3) + DC values + the CU's transform units
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
        if (tid >= 192 && tid < 195) {
            const int pl = tid - 192, np = pl ? n >> 1 : n, lg = pl ? log2n - 1 : log2n;
            int sum = np;
            for (int i = 0; i < np; i++) sum += ref_top(s.ref[pl], np, i) + ref_left(s.ref[pl], np, i);
            s.dc_val[pl] = sum >> (lg + 1);
        }
        if (tid >= 208 && tid < 224) {
            int t = tid - 208, tx = t & 3, ty = t >> 2;
            bool in = tx * 8 >= cx && tx * 8 < cx + n && ty * 8 >= cy && ty * 8 < cy + n;
            s.rs.tu_log2[t] = in ? (uint8_t)log2n : 0;
            s.rs.tu_intra[t] = 1;
        }
        if (tid >= 224 && tid < 227) s.rs.cbf[tid - 224] = 0;
    });
    ex.phase([&](int tid) {      // prediction with the planned modes, residual
        const int cang = s.tab_angle[cmode], cinv = s.tab_inv[cmode], ang = s.tab_angle[mode], inv = s.tab_inv[mode];
        const T *L = intra_filter_on(log2n, mode) ? s.filt : s.ref[0];
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
    ex.phase([&](int tid) {      // reconstruction into the LDS neighbourhood, distortion, rate estimate
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
            int per = pl ? 4 : 8, bx = (k & (per - 1)) * 4, by = (k >> (pl ? 2 : 3)) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0;
            int sh = pl ? 2 : 3;
            if (!s.rs.tu_log2[(by >> sh) * 4 + (bx >> sh)]) continue;
            int b = subblock_bits_q4(s.rs.lvl + base + by * stride + bx, stride);
            if (b) ex.atomic_add(&s.bits[pl], b);
        }
    });
    ex.phase([&](int tid) {
        int t0 = (cy >> 3) * 4 + (cx >> 3);
        if (tid < 16 && s.rs.tu_log2[tid]) {
            mihevc_cu_rec r;
            r.log2_size = (uint8_t)log2n;
            r.flags = (uint8_t)(((s.rs.cbf[0] >> t0) & 1 ? CU_CBF_Y : 0) | ((s.rs.cbf[1] >> t0) & 1 ? CU_CBF_CB : 0) | ((s.rs.cbf[2] >> t0) & 1 ? CU_CBF_CR : 0));
            r.chroma_mode = (uint8_t)cmode; r.qp = (uint8_t)a.prm.qp;
            r.intra_mode[0] = r.intra_mode[1] = r.intra_mode[2] = r.intra_mode[3] = (uint8_t)mode;
            r.mvx = r.mvy = 0; r.cbf_y4 = 0; r.pad[0] = r.pad[1] = r.pad[2] = 0;
            s.cu_acc[tid] = r;
        }
        if (tid >= 64 && tid < 79) s.avmask[(tid - 64) / 5][(tid - 64) % 5] = 0;      // for the next CU's reference samples
        if (tid == 0) {
            int bits = 16 * mode_bits_for(s.cand, mode) + 16 + 24 + (cmode != mode ? 32 : 0);
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

// Stage B of one CTU (oracle intra_code_ctu): the plan's CUs in decoding order, the NxN trial on top of a planned 8x8 CU, then the
// CTU's reconstruction, levels, records and rate estimate to memory.  `fresh`: the workgroup did not run the plan stage itself (I
// pictures: k_intra_plan left the plan in a.plan), so tables, source image and plan are loaded here.
template <typename T, class Ex>
DEV void intra_code_program(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int ctu_x, int ctu_y, bool fresh)
{
    const int x0 = ctu_x * CTU, y0 = ctu_y * CTU;
    const TileBox tb = tile_box(a, ctu_x, ctu_y);
    const unsigned long long lam_split = (unsigned long long)a.prm.lambda_q4;     // (lambda_q4 * 16) >> 4
    if (fresh) residual_init(ex, s.rs);
    ex.phase([&](int tid) {
        if (fresh) {
            load_ctu_source<T>(s.src, a.src, x0, y0, a.w, a.h, tid);
            if (tid >= 64 && tid < 99) { s.tab_angle[tid - 64] = (int16_t)mode_angle(tid - 64); s.tab_inv[tid - 64] = (int16_t)mode_inv_angle(tid - 64); }
            if (tid >= 128 && tid < 134) { s.tab_qs[tid - 128] = g_tab.quant_scale[tid - 128]; s.tab_ls[tid - 128] = g_tab.level_scale[tid - 128]; }
            if (tid < 32) s.nx_mat[tid >> 4][tid & 15] = tid < 16 ? g_tab.dst4[(tid >> 2) & 3][tid & 3] : g_tab.mat[((tid >> 2) & 3) * 8][tid & 3];
            if (tid >= 160 && tid < 181) {
                const IntraPlan &p = a.plan[ctu_y * a.ctus_w + ctu_x];
                s.plan.chosen[tid - 160] = p.chosen[tid - 160]; s.plan.mode[tid - 160] = p.mode[tid - 160]; s.plan.cmode[tid - 160] = p.cmode[tid - 160];
            }
        }
        for (int i = tid; i < 1536; i += NT) s.coef_acc[i] = 0;
        if (tid == 0) s.est = 0;
        if (tid >= 224 && tid < 239) s.avmask[(tid - 224) / 5][(tid - 224) % 5] = 0;
        if (tid < 4 && x0 > 0 && y0 + tid * 8 < a.h) s.left_cu[tid] = a.cu[(size_t)((y0 >> 3) + tid) * (a.w >> 3) + ((x0 - 1) >> 3)];
        // neighbourhood: row -1 (cols -1..63 luma / -1..31 chroma) and column -1 (rows 0..31 / 0..15) from the picture
        for (int u = tid; u < 65 + 32 + 2 * (33 + 16); u += NT) {
            int pl, k, row_len;
            if (u < 97) { pl = 0; k = u; row_len = 65; } else { pl = 1 + (u - 97) / 49; k = (u - 97) % 49; row_len = 33; }
            int xn, yn;
            if (k < row_len) { xn = k - 1; yn = -1; } else { xn = -1; yn = k - row_len; }
            int pw = pl ? a.w >> 1 : a.w, ph = pl ? a.h >> 1 : a.h;
            int gx = (pl ? x0 >> 1 : x0) + xn, gy = (pl ? y0 >> 1 : y0) + yn;
            T v = 0;
            if (gx >= 0 && gy >= 0 && gx < pw && gy < ph) v = a.rec[pl].p[(ptrdiff_t)gy * a.rec[pl].stride + gx];
            if (pl == 0) s.rec_y[(yn + 1) * RY_STRIDE + xn + 1] = v; else s.rec_c[pl - 1][(yn + 1) * RC_STRIDE + xn + 1] = v;
        }
    });
    // Which CUs run is uniform over the workgroup: plan.chosen is written once (above, or by the plan stage behind a barrier) and never
    // again; j_cu / nx_try are words the CU's last phase writes and nobody rewrites before the next CU's last phase.
    unsigned long long jctu = lam_split;
    for (int q = 0; q < 4; q++) {
        const int qx = (q & 1) * 16, qy = (q >> 1) * 16;
        bool any = false;
        for (int b = 0; b < 4; b++) {
            if (!s.plan.chosen[5 + 4 * q + b]) continue;
            const int bx = qx + (b & 1) * 8, by = qy + (b >> 1) * 8;
            any = true;
            intra_code_cu(ex, s, a, x0, y0, tb, bx, by, 3, s.plan.mode[5 + 4 * q + b], s.plan.cmode[5 + 4 * q + b]);
            if (a.prm.intra_nxn && s.nx_try) intra_cu_nxn(ex, s, a, x0, y0, bx, by);
            jctu += s.j_cu;
        }
        if (any) jctu += lam_split;
        if (s.plan.chosen[1 + q]) {
            intra_code_cu(ex, s, a, x0, y0, tb, qx, qy, 4, s.plan.mode[1 + q], s.plan.cmode[1 + q]);
            jctu += s.j_cu + lam_split;
        }
    }
    if (s.plan.chosen[0]) {
        intra_code_cu(ex, s, a, x0, y0, tb, 0, 0, 5, s.plan.mode[0], s.plan.cmode[0]);
        jctu += s.j_cu;
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
                int per = pl ? 4 : 8, bx = (k & (per - 1)) * 4, by = (k >> (pl ? 2 : 3)) * 4, stride = pl ? 16 : 32, base = pl ? 1024 + (pl - 1) * 256 : 0;
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

// plan + code of one CTU in one workgroup: the P pictures' second pass (k_intra_p) and the CPU stepping of the kernel source
template <typename T, class Ex>
DEV void intra_ctu_program(Ex &ex, IntraShared<T> &s, const IntraArgs<T> &a, int ctu_x, int ctu_y)
{
    intra_plan_program<T>(ex, s, a, ctu_x, ctu_y);
    intra_code_program<T>(ex, s, a, ctu_x, ctu_y, false);
}

}  // namespace mihevc
