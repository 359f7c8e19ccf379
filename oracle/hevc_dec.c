/* oracle/hevc_dec.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A small HEVC decoder written from ITU-T H.265 clauses 7.3 (syntax), 9.3 (CABAC), 8.3-8.7 (decoding), for the
 * syntax subset the MI355X encoder can emit: Main / Main10, 4:2:0, one slice per picture, I and P slices with one
 * reference picture, 2Nx2N (+ intra NxN) CUs, merge / skip / AMVP without temporal candidates, residual coding
 * without transform-skip, deblocking with default parameters, SAO.  Anything else is reported as an error instead
 * of being guessed.  It exists because no third-party HEVC decoder is available in this image (SURVEY.md §7 hard
 * part 4): "encode -> bitstream -> this decoder == encoder reconstruction" is the conformance gate we can run.
 * The reference (uingei/hevc) has no decoder of its own; it relies on ffmpeg (core/transcoder.py:506).
 */
#include "hevc_oracle.h"
#include "hevc_dec_recon.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>

#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : (v) > (hi) ? (hi) : (v))
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------ bit reader (RBSP) */
typedef struct { const uint8_t *p; size_t n, pos; int err; } bitrd;   /* pos in bits */
static int br_bit(bitrd *b)
{
    if (b->pos >= b->n * 8) { b->err = 1; return 0; }
    int v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
    b->pos++;
    return v;
}
static uint32_t br_u(bitrd *b, int n) { uint32_t v = 0; while (n--) v = (v << 1) | (uint32_t)br_bit(b); return v; }
static uint32_t br_ue(bitrd *b)
{
    int z = 0;
    while (!br_bit(b) && z < 32 && !b->err) z++;
    return z ? ((1u << z) - 1 + br_u(b, z)) : 0;
}
static int32_t br_se(bitrd *b) { uint32_t k = br_ue(b); return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1); }
static int br_more_data(const bitrd *b)
{   /* more_rbsp_data(): something before the last 1 bit */
    size_t last = b->n * 8;
    while (last > b->pos) { size_t i = last - 1; if ((b->p[i >> 3] >> (7 - (i & 7))) & 1) break; last--; }
    return last > b->pos + 1;
}
static int br_trailing_ok(bitrd *b)
{
    if (!br_bit(b)) return 0;
    while (b->pos & 7) if (br_bit(b)) return 0;
    return b->pos == b->n * 8;
}

/* ------------------------------------------------------------------ CABAC decoder — 9.3.4.3 */
static const uint8_t kRangeLps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185},
    {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},  {85, 104, 123, 142},  {81, 99, 117, 135},
    {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},   {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},
    {56, 69, 81, 94},     {53, 65, 77, 89},     {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},
    {41, 50, 59, 69},     {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},     {23, 28, 33, 39},
    {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},     {18, 22, 26, 30},     {17, 21, 25, 28},
    {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},     {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},
    {12, 14, 17, 20},     {11, 14, 16, 19},     {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},
    {9, 11, 12, 14},      {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
static const uint8_t kTransLps[64] = {0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                      24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

/* context table layout (offsets into one array) and initValues — Tables 9-5..9-37; column = initType 0 (I), 1 (P), 2 (B) */
enum {
    CX_SAO_MERGE = 0, CX_SAO_TYPE = 1, CX_SPLIT_CU = 2, CX_SKIP = 5, CX_PRED_MODE = 8, CX_PART_MODE = 9, CX_PREV_INTRA = 13,
    CX_CHROMA_MODE = 14, CX_RQT_ROOT = 15, CX_MERGE_FLAG = 16, CX_MERGE_IDX = 17, CX_MVP = 18, CX_SPLIT_TU = 19, CX_CBF_LUMA = 22,
    CX_CBF_CHROMA = 24, CX_MVD0 = 29, CX_MVD1 = 30, CX_LAST_X = 31, CX_LAST_Y = 49, CX_CSBF = 67, CX_SIG = 71, CX_G1 = 115,
    CX_G2 = 139, CX_QP_DELTA = 145, CX_INTER_DIR = 147, CX_COUNT = 152
};
#define CNU 154
static const uint8_t kInit[3][CX_COUNT] = {
    { /* I */ 153, 200, 139, 141, 157, CNU, CNU, CNU, CNU, 184, CNU, CNU, CNU, 184, 63, CNU, CNU, CNU, CNU, 153, 138, 138, 111, 141,
      94, 138, 182, 154, 154, CNU, CNU,
      110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
      110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
      91, 171, 134, 141,
      111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
      140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 141, 111,
      140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
      138, 153, 136, 167, 152, 152, 154, 154, CNU, CNU, CNU, CNU, CNU},
    { /* P */ 153, 185, 107, 139, 126, 197, 185, 201, 149, 154, 139, 154, 154, 154, 152, 79, 110, 122, 168, 124, 138, 94, 153, 111,
      149, 107, 167, 154, 154, 140, 198,
      125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
      125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
      121, 140, 61, 154,
      155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
      170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140, 140, 140,
      154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
      107, 167, 91, 122, 107, 167, 154, 154, 95, 79, 63, 31, 31},
    { /* B */ 153, 160, 107, 139, 126, 197, 185, 201, 134, 154, 139, 154, 154, 183, 152, 79, 154, 137, 168, 224, 167, 122, 153, 111,
      149, 92, 167, 154, 154, 169, 198,
      125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
      125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
      121, 140, 61, 154,
      170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
      170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140, 140, 140,
      154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182,
      107, 167, 91, 107, 107, 167, 154, 154, 95, 79, 63, 31, 31}};

typedef struct {
    const uint8_t *p; size_t n, pos;   /* byte position in the slice data */
    uint32_t range, offset; int bits_left;
    uint8_t state[CX_COUNT], mps[CX_COUNT];
    int err;
} cabac;
static int cb_bit(cabac *c)
{
    if (c->bits_left == 0) {
        if (c->pos >= c->n) { c->err = 1; return 0; }
        c->bits_left = 8;
    }
    int v = (c->p[c->pos] >> (c->bits_left - 1)) & 1;
    if (--c->bits_left == 0) c->pos++;
    return v;
}
static void cb_init(cabac *c, const uint8_t *p, size_t n, int init_type, int qp)
{
    memset(c, 0, sizeof *c);
    c->p = p; c->n = n; c->range = 510;
    for (int i = 0; i < 9; i++) c->offset = (c->offset << 1) | (uint32_t)cb_bit(c);
    qp = CLIP3(0, 51, qp);
    for (int i = 0; i < CX_COUNT; i++) {
        int iv = kInit[init_type][i];
        int m = (iv >> 4) * 5 - 45, nn = ((iv & 15) << 3) - 16;
        int pre = CLIP3(1, 126, ((m * qp) >> 4) + nn);
        c->mps[i] = pre > 63;
        c->state[i] = (uint8_t)(c->mps[i] ? pre - 64 : 63 - pre);
    }
}
static int cb_decision(cabac *c, int ctx)
{
    uint32_t lps = kRangeLps[c->state[ctx]][(c->range >> 6) & 3];
    int bin;
    c->range -= lps;
    if (c->offset >= c->range) {
        bin = !c->mps[ctx];
        c->offset -= c->range; c->range = lps;
        if (c->state[ctx] == 0) c->mps[ctx] = !c->mps[ctx];
        c->state[ctx] = kTransLps[c->state[ctx]];
    } else {
        bin = c->mps[ctx];
        if (c->state[ctx] < 62) c->state[ctx]++;
    }
    while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | (uint32_t)cb_bit(c); }
    return bin;
}
static int cb_bypass(cabac *c)
{
    c->offset = (c->offset << 1) | (uint32_t)cb_bit(c);
    if (c->offset >= c->range) { c->offset -= c->range; return 1; }
    return 0;
}
static uint32_t cb_bypass_n(cabac *c, int n) { uint32_t v = 0; while (n--) v = (v << 1) | (uint32_t)cb_bypass(c); return v; }
static int cb_terminate(cabac *c)
{
    c->range -= 2;
    if (c->offset >= c->range) return 1;
    while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | (uint32_t)cb_bit(c); }
    return 0;
}

/* ------------------------------------------------------------------ decoder state */
typedef struct { pix *base[3]; pix *pl[3]; int stride[3]; int poc; int seq, in_dpb; } picture;      /* seq: coded video sequence (counts IDR pictures); in_dpb: still a possible reference */

typedef struct { char name[40]; long long val; } kv;

struct orc_decoder {
    /* SPS */
    int have_sps, have_pps, have_vps;
    int w, h, bit_depth, conf[4], log2_ctb, log2_min_cb, log2_min_tb, log2_max_tb, th_inter, th_intra;
    int sao_on, strong_intra, poc_bits, num_strps, strps_neg[64], strps_delta[64][4], strps_used[64][4], amp, tmvp;
    int strps_pos[64], strps_pdelta[64][4], strps_pused[64][4];      /* the positive (later in output order) pictures of every short-term RPS */
    /* PPS (the active one; parse_pps files a copy under its id, decode_slice activates it) */
    int init_qp, sign_hiding, cu_qp_delta, cb_off, cr_off, lf_across, dbk_control, dbk_override_en, pps_dbk_disabled,
        cabac_init_present, par_mrg_level, transform_skip;
    int tiles, tile_cols, tile_rows, lf_across_tiles, col_bd[22], row_bd[24];
    int hrd_nal, hrd_init_len, hrd_au_len, hrd_dpb_len;     /* E.2.2, for the buffering period / picture timing SEI */
    struct pps_copy { int valid, v[12], tiles, tile_cols, tile_rows, lf_across_tiles, col_bd[22], row_bd[24]; } pps[4];
    /* pictures */
    picture *pics; int n_pics, cap_pics;
    picture cur; int cur_valid;
    orc_cu_rec *cu; uint8_t *depth8; uint8_t *skip8; orc_sao_ctu *sao;
    /* slice */
    int slice_type, slice_qp, sao_luma, sao_chroma, max_merge, ref_idx;
    /* slices of the current picture: full-width bands of CTU rows [band_row0[k], band_row0[k + 1]); the slice being decoded starts at luma row slice_y0 */
    int n_bands, band_row0[24], slice_y0, pic_sao, pic_lf_across;
    unsigned char band_lf_across[24];     /* per slice: slice_loop_filter_across_slices_enabled_flag (inferred from the PPS flag when absent) */
    size_t *epb; int n_epb, cap_epb;   /* positions (escaped payload offsets after the NAL header) of the removed 0x03 bytes */
    int poc, seq, prev_tid0_poc, cur_nal;       /* prev_tid0_poc: POC of the previous picture that is not a sub-layer non-reference picture (8.3.1) */
    const picture *ref[2];             /* RefPicList0[0], RefPicList1[0] of the current slice */
    int ref_poc[2], mvd_l1_zero;
    int *out_order; int n_out_order;   /* decode index of every picture in output order (coded video sequence, then POC) */
    cabac cb;
    char err[256];
    kv kvs[160]; int n_kv;
};

static void set_err(orc_decoder *d, const char *fmt, ...)
{
    if (d->err[0]) return;
    va_list ap; va_start(ap, fmt); vsnprintf(d->err, sizeof d->err, fmt, ap); va_end(ap);
}
static void put_kv(orc_decoder *d, const char *name, long long v)
{
    for (int i = 0; i < d->n_kv; i++) if (!strcmp(d->kvs[i].name, name)) { d->kvs[i].val = v; return; }
    if (d->n_kv < 160) { snprintf(d->kvs[d->n_kv].name, 40, "%s", name); d->kvs[d->n_kv++].val = v; }
}

orc_decoder *orc_dec_open(void) { return (orc_decoder *)calloc(1, sizeof(orc_decoder)); }
const char *orc_dec_error(const orc_decoder *d) { return d->err; }
int orc_dec_query(const orc_decoder *d, const char *f, long long *v)
{
    for (int i = 0; i < d->n_kv; i++) if (!strcmp(d->kvs[i].name, f)) { *v = d->kvs[i].val; return 1; }
    return 0;
}
static void free_pic(picture *p) { for (int i = 0; i < 3; i++) free(p->base[i]); memset(p, 0, sizeof *p); }
void orc_dec_close(orc_decoder *d)
{
    if (!d) return;
    for (int i = 0; i < d->n_pics; i++) free_pic(&d->pics[i]);
    free(d->pics); free(d->cu); free(d->depth8); free(d->skip8); free(d->sao); free(d->epb); free(d->out_order);
    if (d->cur_valid) free_pic(&d->cur);
    free(d);
}
int orc_dec_info(const orc_decoder *d, int *w, int *h, int *bd, int *cw, int *ch)
{
    if (!d->have_sps) return -1;
    *w = d->w; *h = d->h; *bd = d->bit_depth;
    *cw = d->w - 2 * (d->conf[0] + d->conf[1]); *ch = d->h - 2 * (d->conf[2] + d->conf[3]);
    return 0;
}
/* pictures come back in OUTPUT order (C.5.2: by coded video sequence, then by picture order count); without B pictures that is the decoding order */
int orc_dec_get_frame(const orc_decoder *d0, int idx, pix *y, pix *u, pix *v)
{
    orc_decoder *d = (orc_decoder *)d0;
    if (idx < 0 || idx >= d->n_pics) return -1;
    if (d->n_out_order != d->n_pics) {
        d->out_order = (int *)realloc(d->out_order, sizeof(int) * (size_t)d->n_pics);
        for (int i = 0; i < d->n_pics; i++) d->out_order[i] = i;
        for (int i = 1; i < d->n_pics; i++) {          /* insertion sort: the disorder is local (one picture) */
            int k = d->out_order[i], j = i - 1;
            while (j >= 0 && (d->pics[d->out_order[j]].seq > d->pics[k].seq ||
                              (d->pics[d->out_order[j]].seq == d->pics[k].seq && d->pics[d->out_order[j]].poc > d->pics[k].poc))) { d->out_order[j + 1] = d->out_order[j]; j--; }
            d->out_order[j + 1] = k;
        }
        d->n_out_order = d->n_pics;
    }
    const picture *p = &d->pics[d->out_order[idx]];
    pix *dst[3] = {y, u, v};
    for (int c = 0; c < 3; c++) {
        int w = c ? d->w / 2 : d->w, h = c ? d->h / 2 : d->h;
        for (int r = 0; r < h; r++) memcpy(dst[c] + (size_t)r * w, p->pl[c] + (size_t)r * p->stride[c], w * sizeof(pix));
    }
    return p->poc;
}
static void alloc_pic(orc_decoder *d, picture *p)
{
    for (int c = 0; c < 3; c++) {
        int pad = c ? ORC_PAD / 2 : ORC_PAD, w = c ? d->w / 2 : d->w, h = c ? d->h / 2 : d->h;
        p->stride[c] = w + 2 * pad;
        p->base[c] = (pix *)calloc((size_t)p->stride[c] * (h + 2 * pad), sizeof(pix));
        p->pl[c] = p->base[c] + (size_t)pad * p->stride[c] + pad;
    }
}

/* ------------------------------------------------------------------ parameter sets — 7.3.2 */
static void parse_ptl(orc_decoder *d, bitrd *b, const char *pfx)
{
    char nm[40];
    br_u(b, 2);
    int tier = (int)br_u(b, 1), profile = (int)br_u(b, 5);
    uint32_t compat = br_u(b, 32);
    int prog = br_bit(b), inter = br_bit(b), nonpacked = br_bit(b), frameonly = br_bit(b);
    br_u(b, 32); br_u(b, 11); br_bit(b);
    int level = (int)br_u(b, 8);
    snprintf(nm, 40, "%s.profile_idc", pfx); put_kv(d, nm, profile);
    snprintf(nm, 40, "%s.tier_flag", pfx); put_kv(d, nm, tier);
    snprintf(nm, 40, "%s.level_idc", pfx); put_kv(d, nm, level);
    snprintf(nm, 40, "%s.compat", pfx); put_kv(d, nm, compat);
    snprintf(nm, 40, "%s.progressive", pfx); put_kv(d, nm, prog);
    snprintf(nm, 40, "%s.frame_only", pfx); put_kv(d, nm, frameonly);
    (void)inter; (void)nonpacked;
}

static int parse_vps(orc_decoder *d, bitrd *b)
{
    put_kv(d, "vps.id", br_u(b, 4));
    br_u(b, 2);
    if (br_u(b, 6) != 0) { set_err(d, "vps: layers"); return -1; }
    if (br_u(b, 3) != 0) { set_err(d, "vps: sub-layers unsupported"); return -1; }
    br_bit(b);
    if (br_u(b, 16) != 0xffff) { set_err(d, "vps: reserved_0xffff"); return -1; }
    parse_ptl(d, b, "vps");
    int ordering = br_bit(b); (void)ordering;
    put_kv(d, "vps.max_dec_pic_buffering_minus1", br_ue(b));
    put_kv(d, "vps.max_num_reorder", br_ue(b));
    br_ue(b);
    br_u(b, 6);
    if (br_ue(b) != 0) { set_err(d, "vps: layer sets"); return -1; }
    int timing = br_bit(b);
    put_kv(d, "vps.timing_info_present", timing);
    if (timing) {
        put_kv(d, "vps.num_units_in_tick", br_u(b, 32));
        put_kv(d, "vps.time_scale", br_u(b, 32));
        if (br_bit(b)) br_ue(b);
        if (br_ue(b) != 0) { set_err(d, "vps: hrd"); return -1; }
    }
    if (br_bit(b)) { set_err(d, "vps: extension"); return -1; }
    if (!br_trailing_ok(b) || b->err) { set_err(d, "vps: trailing bits"); return -1; }
    d->have_vps = 1;
    return 0;
}

static int parse_hrd(orc_decoder *d, bitrd *b)
{
    int nal = br_bit(b), vcl = br_bit(b), subpic = 0;
    put_kv(d, "hrd.nal_present", nal);
    if (nal || vcl) {
        subpic = br_bit(b);
        if (subpic) { set_err(d, "hrd: sub-pic"); return -1; }
        put_kv(d, "hrd.bit_rate_scale", br_u(b, 4));
        put_kv(d, "hrd.cpb_size_scale", br_u(b, 4));
        d->hrd_nal = nal;
        d->hrd_init_len = 1 + (int)br_u(b, 5); d->hrd_au_len = 1 + (int)br_u(b, 5); d->hrd_dpb_len = 1 + (int)br_u(b, 5);
        put_kv(d, "hrd.initial_cpb_removal_delay_length_minus1", d->hrd_init_len - 1);
        put_kv(d, "hrd.au_cpb_removal_delay_length_minus1", d->hrd_au_len - 1);
        put_kv(d, "hrd.dpb_output_delay_length_minus1", d->hrd_dpb_len - 1);
    }
    int fixed_general = br_bit(b), fixed_cvs = 1, low_delay = 0, cpb_cnt = 0;
    if (!fixed_general) fixed_cvs = br_bit(b);
    if (fixed_cvs) br_ue(b); else low_delay = br_bit(b);
    if (!low_delay) cpb_cnt = (int)br_ue(b);
    for (int k = 0; k < nal + vcl; k++)
        for (int i = 0; i <= cpb_cnt; i++) {
            long long br = br_ue(b), cs = br_ue(b);
            if (k == 0 && i == 0) { put_kv(d, "hrd.bit_rate_value_minus1", br); put_kv(d, "hrd.cpb_size_value_minus1", cs); }
            put_kv(d, "hrd.cbr_flag", br_bit(b));
        }
    return 0;
}

static int parse_sps(orc_decoder *d, bitrd *b)
{
    br_u(b, 4);
    if (br_u(b, 3) != 0) { set_err(d, "sps: sub-layers unsupported"); return -1; }
    br_bit(b);
    parse_ptl(d, b, "sps");
    if (br_ue(b) != 0) { set_err(d, "sps: id"); return -1; }
    if (br_ue(b) != 1) { set_err(d, "sps: chroma_format_idc != 1"); return -1; }
    d->w = (int)br_ue(b); d->h = (int)br_ue(b);
    memset(d->conf, 0, sizeof d->conf);
    if (br_bit(b)) for (int i = 0; i < 4; i++) d->conf[i] = (int)br_ue(b);
    d->bit_depth = 8 + (int)br_ue(b);
    if ((int)br_ue(b) + 8 != d->bit_depth) { set_err(d, "sps: chroma bit depth differs"); return -1; }
    d->poc_bits = 4 + (int)br_ue(b);
    br_bit(b);
    put_kv(d, "sps.max_dec_pic_buffering_minus1", br_ue(b));
    put_kv(d, "sps.max_num_reorder", br_ue(b));
    br_ue(b);
    d->log2_min_cb = 3 + (int)br_ue(b);
    d->log2_ctb = d->log2_min_cb + (int)br_ue(b);
    d->log2_min_tb = 2 + (int)br_ue(b);
    d->log2_max_tb = d->log2_min_tb + (int)br_ue(b);
    d->th_inter = (int)br_ue(b); d->th_intra = (int)br_ue(b);
    if (br_bit(b)) { set_err(d, "sps: scaling lists"); return -1; }
    d->amp = br_bit(b);
    d->sao_on = br_bit(b);
    if (br_bit(b)) { set_err(d, "sps: pcm"); return -1; }
    d->num_strps = (int)br_ue(b);
    if (d->num_strps > 64) { set_err(d, "sps: strps count"); return -1; }
    for (int i = 0; i < d->num_strps; i++) {
        if (i && br_bit(b)) { set_err(d, "sps: inter RPS prediction"); return -1; }
        int neg = (int)br_ue(b), posn = (int)br_ue(b);
        if (posn > 4 || neg > 4) { set_err(d, "sps: rps shape"); return -1; }
        d->strps_neg[i] = neg; d->strps_pos[i] = posn;
        int acc = 0;
        for (int k = 0; k < neg; k++) { acc -= (int)br_ue(b) + 1; d->strps_delta[i][k] = acc; d->strps_used[i][k] = br_bit(b); }
        acc = 0;
        for (int k = 0; k < posn; k++) { acc += (int)br_ue(b) + 1; d->strps_pdelta[i][k] = acc; d->strps_pused[i][k] = br_bit(b); }
    }
    if (br_bit(b)) { set_err(d, "sps: long-term refs"); return -1; }
    d->tmvp = br_bit(b);
    if (d->tmvp) { set_err(d, "sps: temporal mvp"); return -1; }
    d->strong_intra = br_bit(b);
    int vui = br_bit(b);
    put_kv(d, "sps.vui_present", vui);
    if (vui) {
        if (br_bit(b)) { int idc = (int)br_u(b, 8); put_kv(d, "vui.aspect_ratio_idc", idc); if (idc == 255) { br_u(b, 16); br_u(b, 16); } }
        if (br_bit(b)) br_bit(b);
        int vs = br_bit(b);
        put_kv(d, "vui.video_signal_type_present", vs);
        if (vs) {
            put_kv(d, "vui.video_format", br_u(b, 3));
            put_kv(d, "vui.full_range", br_bit(b));
            int cd = br_bit(b);
            put_kv(d, "vui.colour_description_present", cd);
            if (cd) { put_kv(d, "vui.colour_primaries", br_u(b, 8)); put_kv(d, "vui.transfer", br_u(b, 8)); put_kv(d, "vui.matrix", br_u(b, 8)); }
        }
        int cl = br_bit(b);
        put_kv(d, "vui.chroma_loc_present", cl);
        if (cl) { put_kv(d, "vui.chroma_loc_top", br_ue(b)); br_ue(b); }
        br_bit(b); br_bit(b); br_bit(b);
        if (br_bit(b)) { br_ue(b); br_ue(b); br_ue(b); br_ue(b); }
        int ti = br_bit(b);
        put_kv(d, "vui.timing_info_present", ti);
        if (ti) {
            put_kv(d, "vui.num_units_in_tick", br_u(b, 32));
            put_kv(d, "vui.time_scale", br_u(b, 32));
            if (br_bit(b)) br_ue(b);
            int hrd = br_bit(b);
            put_kv(d, "vui.hrd_present", hrd);
            if (hrd && parse_hrd(d, b)) return -1;
        }
        if (br_bit(b)) { br_bit(b); br_bit(b); br_bit(b); br_ue(b); br_ue(b); br_ue(b); br_ue(b); br_ue(b); }
    }
    if (br_bit(b)) { set_err(d, "sps: extension"); return -1; }
    if (!br_trailing_ok(b) || b->err) { set_err(d, "sps: trailing bits"); return -1; }
    if (d->log2_ctb != ORC_CTU_LOG2 || d->log2_min_cb != 3 || d->log2_min_tb != 2 || d->log2_max_tb != 5) {
        set_err(d, "sps: block sizes ctb=%d mincb=%d tb=%d..%d unsupported", d->log2_ctb, d->log2_min_cb, d->log2_min_tb, d->log2_max_tb);
        return -1;
    }
    if ((d->w & 7) || (d->h & 7)) { set_err(d, "sps: size not multiple of MinCb"); return -1; }
    put_kv(d, "sps.width", d->w); put_kv(d, "sps.height", d->h); put_kv(d, "sps.bit_depth", d->bit_depth);
    put_kv(d, "sps.sao", d->sao_on); put_kv(d, "sps.strong_intra", d->strong_intra); put_kv(d, "sps.amp", d->amp);
    put_kv(d, "sps.conf_right", d->conf[1]); put_kv(d, "sps.conf_bottom", d->conf[3]);
    put_kv(d, "sps.log2_ctb", d->log2_ctb); put_kv(d, "sps.poc_bits", d->poc_bits);
    d->have_sps = 1;
    return 0;
}

static void save_pps(orc_decoder *d, int id)
{
    struct pps_copy *p = &d->pps[id];
    int v[12] = {d->init_qp, d->sign_hiding, d->cu_qp_delta, d->cb_off, d->cr_off, d->lf_across, d->dbk_control, d->dbk_override_en,
                 d->pps_dbk_disabled, d->cabac_init_present, d->par_mrg_level, d->transform_skip};
    memcpy(p->v, v, sizeof v);
    p->valid = 1; p->tiles = d->tiles; p->tile_cols = d->tile_cols; p->tile_rows = d->tile_rows; p->lf_across_tiles = d->lf_across_tiles;
    memcpy(p->col_bd, d->col_bd, sizeof p->col_bd); memcpy(p->row_bd, d->row_bd, sizeof p->row_bd);
}
static int activate_pps(orc_decoder *d, int id)
{
    if (id < 0 || id > 3 || !d->pps[id].valid) return -1;
    const struct pps_copy *p = &d->pps[id];
    d->init_qp = p->v[0]; d->sign_hiding = p->v[1]; d->cu_qp_delta = p->v[2]; d->cb_off = p->v[3]; d->cr_off = p->v[4]; d->lf_across = p->v[5];
    d->dbk_control = p->v[6]; d->dbk_override_en = p->v[7]; d->pps_dbk_disabled = p->v[8]; d->cabac_init_present = p->v[9];
    d->par_mrg_level = p->v[10]; d->transform_skip = p->v[11];
    d->tiles = p->tiles; d->tile_cols = p->tile_cols; d->tile_rows = p->tile_rows; d->lf_across_tiles = p->lf_across_tiles;
    memcpy(d->col_bd, p->col_bd, sizeof p->col_bd); memcpy(d->row_bd, p->row_bd, sizeof p->row_bd);
    return 0;
}

static int parse_pps(orc_decoder *d, bitrd *b)
{
    int pps_id = (int)br_ue(b);
    if (pps_id > 3 || br_ue(b)) { set_err(d, "pps: ids"); return -1; }
    if (!d->have_sps) { set_err(d, "pps before sps"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: dependent slices"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: output_flag_present"); return -1; }
    if (br_u(b, 3)) { set_err(d, "pps: extra slice header bits"); return -1; }
    d->sign_hiding = br_bit(b);
    d->cabac_init_present = br_bit(b);
    if (br_ue(b) || br_ue(b)) { set_err(d, "pps: default ref idx counts"); return -1; }
    d->init_qp = 26 + br_se(b);
    if (br_bit(b)) { set_err(d, "pps: constrained intra"); return -1; }
    d->transform_skip = br_bit(b);
    if (d->transform_skip) { set_err(d, "pps: transform skip"); return -1; }
    d->cu_qp_delta = br_bit(b);
    if (d->cu_qp_delta) { set_err(d, "pps: cu_qp_delta"); return -1; }
    d->cb_off = br_se(b); d->cr_off = br_se(b);
    if (d->cb_off || d->cr_off) { set_err(d, "pps: chroma qp offsets"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: slice chroma offsets"); return -1; }
    if (br_bit(b) || br_bit(b)) { set_err(d, "pps: weighted prediction"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: transquant bypass"); return -1; }
    d->tiles = br_bit(b);
    if (br_bit(b)) { set_err(d, "pps: wpp"); return -1; }
    {   /* 7.3.2.3 tiles + 6.5.1 column / row boundaries */
        int wc = (d->w + (1 << d->log2_ctb) - 1) >> d->log2_ctb, hc = (d->h + (1 << d->log2_ctb) - 1) >> d->log2_ctb;
        d->tile_cols = d->tile_rows = 1; d->lf_across_tiles = 1;
        d->col_bd[0] = d->row_bd[0] = 0;
        if (d->tiles) {
            d->tile_cols = 1 + (int)br_ue(b); d->tile_rows = 1 + (int)br_ue(b);
            if (d->tile_cols > 20 || d->tile_rows > 22 || d->tile_cols > wc || d->tile_rows > hc || (d->tile_cols == 1 && d->tile_rows == 1)) {
                set_err(d, "pps: tile grid %dx%d", d->tile_cols, d->tile_rows); return -1;
            }
            if (br_bit(b)) {      /* uniform_spacing_flag */
                for (int i = 0; i <= d->tile_cols; i++) d->col_bd[i] = i * wc / d->tile_cols;
                for (int i = 0; i <= d->tile_rows; i++) d->row_bd[i] = i * hc / d->tile_rows;
            } else {
                for (int i = 0; i < d->tile_cols - 1; i++) d->col_bd[i + 1] = d->col_bd[i] + 1 + (int)br_ue(b);
                for (int i = 0; i < d->tile_rows - 1; i++) d->row_bd[i + 1] = d->row_bd[i] + 1 + (int)br_ue(b);
                d->col_bd[d->tile_cols] = wc; d->row_bd[d->tile_rows] = hc;
            }
            for (int i = 0; i < d->tile_cols; i++) if ((d->col_bd[i + 1] - d->col_bd[i]) << d->log2_ctb < 256) { set_err(d, "pps: tile column narrower than 256 luma samples (A.4.1)"); return -1; }
            for (int i = 0; i < d->tile_rows; i++) if ((d->row_bd[i + 1] - d->row_bd[i]) << d->log2_ctb < 64) { set_err(d, "pps: tile row lower than 64 luma samples (A.4.1)"); return -1; }
            d->lf_across_tiles = br_bit(b);
            if (!d->lf_across_tiles) { set_err(d, "pps: loop filter across tiles disabled"); return -1; }
        }
        d->col_bd[d->tile_cols] = wc; d->row_bd[d->tile_rows] = hc;
    }
    d->lf_across = br_bit(b);
    d->dbk_control = br_bit(b);
    if (d->dbk_control) { set_err(d, "pps: deblocking control"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: scaling list"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: lists modification"); return -1; }
    d->par_mrg_level = 2 + (int)br_ue(b);
    if (br_bit(b)) { set_err(d, "pps: slice header ext"); return -1; }
    if (br_bit(b)) { set_err(d, "pps: extension"); return -1; }
    if (!br_trailing_ok(b) || b->err) { set_err(d, "pps: trailing bits"); return -1; }
    put_kv(d, "pps.init_qp", d->init_qp); put_kv(d, "pps.sign_hiding", d->sign_hiding);
    if (d->tiles) { put_kv(d, "pps.tile_cols", d->tile_cols); put_kv(d, "pps.tile_rows", d->tile_rows); }
    save_pps(d, pps_id);
    d->have_pps = 1;
    return 0;
}

static int parse_sei(orc_decoder *d, bitrd *b)
{
    while (br_more_data(b) && !b->err) {
        int type = 0, size = 0, v;
        while ((v = (int)br_u(b, 8)) == 255) type += 255;
        type += v;
        while ((v = (int)br_u(b, 8)) == 255) size += 255;
        size += v;
        size_t end = b->pos + (size_t)size * 8;
        char nm[40];
        snprintf(nm, 40, "sei.%d.size", type); put_kv(d, nm, size);
        if (type == 137 && size == 24) {
            const char *n3[3] = {"g", "b", "r"};
            for (int i = 0; i < 3; i++) {
                snprintf(nm, 40, "sei.mdcv.%sx", n3[i]); put_kv(d, nm, br_u(b, 16));
                snprintf(nm, 40, "sei.mdcv.%sy", n3[i]); put_kv(d, nm, br_u(b, 16));
            }
            put_kv(d, "sei.mdcv.wpx", br_u(b, 16)); put_kv(d, "sei.mdcv.wpy", br_u(b, 16));
            put_kv(d, "sei.mdcv.max_lum", br_u(b, 32)); put_kv(d, "sei.mdcv.min_lum", br_u(b, 32));
        } else if (type == 144 && size == 4) {
            put_kv(d, "sei.cll.max_cll", br_u(b, 16)); put_kv(d, "sei.cll.max_fall", br_u(b, 16));
        } else if (type == 0) {           /* D.2.2 buffering period (needs the SPS's HRD lengths) */
            if (!d->hrd_nal) { set_err(d, "buffering period SEI without NAL HRD parameters"); return -1; }
            if (br_ue(b)) { set_err(d, "bp: sps id"); return -1; }
            int irap = br_bit(b);
            if (irap) { br_u(b, d->hrd_au_len); br_u(b, d->hrd_dpb_len); }
            put_kv(d, "sei.bp.concatenation", br_bit(b));
            br_u(b, d->hrd_au_len);
            put_kv(d, "sei.bp.initial_delay", br_u(b, d->hrd_init_len));
            put_kv(d, "sei.bp.initial_offset", br_u(b, d->hrd_init_len));
            long long n_bp = 0; orc_dec_query(d, "count.sei_bp", &n_bp); put_kv(d, "count.sei_bp", n_bp + 1);
        } else if (type == 1) {           /* D.2.3 picture timing, frame_field_info_present_flag = 0 */
            if (!d->hrd_nal) { set_err(d, "picture timing SEI without HRD parameters"); return -1; }
            put_kv(d, "sei.pt.au_cpb_removal_delay_minus1", br_u(b, d->hrd_au_len));
            put_kv(d, "sei.pt.dpb_output_delay", br_u(b, d->hrd_dpb_len));
            long long n_pt = 0; orc_dec_query(d, "count.sei_pt", &n_pt); put_kv(d, "count.sei_pt", n_pt + 1);
        }
        if (b->pos > end) { set_err(d, "sei %d: payload overrun", type); return -1; }
        b->pos = end;
    }
    if (!br_trailing_ok(b) || b->err) { set_err(d, "sei: trailing bits"); return -1; }
    return 0;
}

/* ------------------------------------------------------------------ slice data helpers */
static inline int zorder6(int bx, int by)
{
    int z = 0;
    for (int i = 0; i < 3; i++) z |= ((bx >> i) & 1) << (2 * i) | ((by >> i) & 1) << (2 * i + 1);
    return z;
}
static inline int zaddr(const orc_decoder *d, int x, int y)
{
    int wc = (d->w + ORC_CTU - 1) >> ORC_CTU_LOG2;
    return (((y >> ORC_CTU_LOG2) * wc + (x >> ORC_CTU_LOG2)) << 6) | zorder6((x & 31) >> 2, (y & 31) >> 2);
}
/* 6.4.1 z-scan availability of (xn,yn) seen from (xc,yc) */
static inline int tile_col_of(const orc_decoder *d, int ctb_x) { int i = 0; while (i + 1 < d->tile_cols && d->col_bd[i + 1] <= ctb_x) i++; return i; }
static inline int tile_row_of(const orc_decoder *d, int ctb_y) { int i = 0; while (i + 1 < d->tile_rows && d->row_bd[i + 1] <= ctb_y) i++; return i; }
static inline int same_tile_d(const orc_decoder *d, int xa, int ya, int xb, int yb)
{
    return tile_col_of(d, xa >> d->log2_ctb) == tile_col_of(d, xb >> d->log2_ctb) && tile_row_of(d, ya >> d->log2_ctb) == tile_row_of(d, yb >> d->log2_ctb);
}
/* inside one tile the tile scan is the raster scan of its CTBs, so "earlier in decoding order" is the picture-raster
 * z-order comparison restricted to the same tile (6.4.1 with 6.5.1) */
static inline int avail_z(const orc_decoder *d, int xc, int yc, int xn, int yn)
{
    return xn >= 0 && yn >= d->slice_y0 && xn < d->w && yn < d->h && zaddr(d, xn, yn) <= zaddr(d, xc, yc) && same_tile_d(d, xn, yn, xc, yc);
}
static inline orc_cu_rec *cu_at(orc_decoder *d, int x, int y) { return &d->cu[(y >> 3) * (d->w >> 3) + (x >> 3)]; }

/* scan order tables — 6.5.3..6.5.5, generated */
static uint8_t g_scan[3][4][64][2];   /* [scanIdx][log2 blk size 0..3][pos][x,y] */
static int g_scan_ready;
static void build_scans(void)
{
    if (g_scan_ready) return;
    for (int l = 0; l < 4; l++) {
        int n = 1 << l, i = 0, x = 0, y = 0;
        for (;;) {                                  /* up-right diagonal */
            while (y >= 0) { if (x < n && y < n) { g_scan[0][l][i][0] = (uint8_t)x; g_scan[0][l][i][1] = (uint8_t)y; i++; } y--; x++; }
            y = x; x = 0;
            if (i >= n * n) break;
        }
        i = 0;
        for (y = 0; y < n; y++) for (x = 0; x < n; x++) { g_scan[1][l][i][0] = (uint8_t)x; g_scan[1][l][i][1] = (uint8_t)y; i++; }
        i = 0;
        for (x = 0; x < n; x++) for (y = 0; y < n; y++) { g_scan[2][l][i][0] = (uint8_t)x; g_scan[2][l][i][1] = (uint8_t)y; i++; }
    }
    g_scan_ready = 1;
}

/* 7.3.8.11 + 9.3.4.2.4-.7 residual_coding; writes TransCoeffLevel into lvl (n*n raster), returns 0 / -1 */
static int residual_coding(orc_decoder *d, int log2n, int c_idx, int scan_idx, int16_t *lvl)
{
    cabac *c = &d->cb;
    int n = 1 << log2n;
    memset(lvl, 0, sizeof(int16_t) * n * n);
    /* last position prefixes */
    int off, shift;
    if (c_idx == 0) { off = 3 * (log2n - 2) + ((log2n - 1) >> 2); shift = (log2n + 1) >> 2; }
    else { off = 15; shift = log2n - 2; }
    int cmax = (log2n << 1) - 1, px = 0, py = 0;
    while (px < cmax && cb_decision(c, CX_LAST_X + off + (px >> shift))) px++;
    while (py < cmax && cb_decision(c, CX_LAST_Y + off + (py >> shift))) py++;
    int lx = px, ly = py;
    if (px > 3) { int nb = (px >> 1) - 1; lx = (1 << nb) * (2 + (px & 1)) + (int)cb_bypass_n(c, nb); }
    if (py > 3) { int nb = (py >> 1) - 1; ly = (1 << nb) * (2 + (py & 1)) + (int)cb_bypass_n(c, nb); }
    if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
    if (lx >= n || ly >= n) { set_err(d, "residual: last position out of block"); return -1; }
    /* locate last sub-block / position */
    int l2sb = log2n - 2, last_sb = (1 << (2 * l2sb)) - 1, last_pos = 16, xs, ys, xc, yc;
    do {
        if (last_pos == 0) { last_pos = 16; last_sb--; }
        last_pos--;
        xs = g_scan[scan_idx][l2sb][last_sb][0]; ys = g_scan[scan_idx][l2sb][last_sb][1];
        xc = (xs << 2) + g_scan[scan_idx][2][last_pos][0]; yc = (ys << 2) + g_scan[scan_idx][2][last_pos][1];
    } while (xc != lx || yc != ly);
    uint8_t csbf[8][8];
    memset(csbf, 0, sizeof csbf);
    int g1_carry = 1;       /* HM's c1 carried across sub-blocks == spec lastGreater1Ctx logic (9.3.4.2.6) */
    int nsb = 1 << l2sb;
    for (int i = last_sb; i >= 0; i--) {
        xs = g_scan[scan_idx][l2sb][i][0]; ys = g_scan[scan_idx][l2sb][i][1];
        int infer_dc = 0;
        int right = xs + 1 < nsb ? csbf[ys][xs + 1] : 0, below = ys + 1 < nsb ? csbf[ys + 1][xs] : 0;
        if (i < last_sb && i > 0) {
            csbf[ys][xs] = (uint8_t)cb_decision(c, CX_CSBF + ((right | below) ? 1 : 0) + (c_idx ? 2 : 0));
            infer_dc = 1;
        } else csbf[ys][xs] = 1;
        uint8_t sig[16];
        memset(sig, 0, 16);
        int prev_csbf = right + 2 * below;
        for (int k = (i == last_sb) ? last_pos : 15; k >= 0; k--) {
            int xp = g_scan[scan_idx][2][k][0], yp = g_scan[scan_idx][2][k][1];
            xc = (xs << 2) + xp; yc = (ys << 2) + yp;
            if (i == last_sb && k == last_pos) { sig[k] = 1; continue; }
            if (!csbf[ys][xs]) continue;
            if (k == 0 && infer_dc) { sig[0] = 1; continue; }
            int sc;
            if (log2n == 2) { static const uint8_t m[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}; sc = m[(yc << 2) + xc]; }
            else if (xc + yc == 0) sc = 0;
            else {
                if (prev_csbf == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
                else if (prev_csbf == 1) sc = yp == 0 ? 2 : yp == 1 ? 1 : 0;
                else if (prev_csbf == 2) sc = xp == 0 ? 2 : xp == 1 ? 1 : 0;
                else sc = 2;
                if (c_idx == 0) { if (xs || ys) sc += 3; sc += log2n == 3 ? (scan_idx == 0 ? 9 : 15) : 21; }
                else sc += log2n == 3 ? 9 : 12;
            }
            sig[k] = (uint8_t)cb_decision(c, CX_SIG + (c_idx ? 27 + sc : sc));
            if (sig[k]) infer_dc = 0;
        }
        int nsig = 0;
        for (int k = 0; k < 16; k++) nsig += sig[k];
        if (!nsig) continue;
        int ctx_set = (i > 0 && c_idx == 0) ? 2 : 0;
        if (g1_carry == 0) ctx_set++;
        int c1 = 1, ng1 = 0, last_g1_pos = -1, first_sig = 16, last_sig = -1;
        uint8_t g1[16], g2[16];
        memset(g1, 0, 16); memset(g2, 0, 16);
        for (int k = 15; k >= 0; k--) {
            if (!sig[k]) continue;
            if (ng1 < 8) {
                g1[k] = (uint8_t)cb_decision(c, CX_G1 + ctx_set * 4 + c1 + (c_idx ? 16 : 0));
                ng1++;
                if (g1[k]) { c1 = 0; if (last_g1_pos < 0) last_g1_pos = k; }
                else if (c1 > 0 && c1 < 3) c1++;
            }
            if (last_sig < 0) last_sig = k;
            first_sig = k;
        }
        g1_carry = c1;
        if (last_g1_pos >= 0) g2[last_g1_pos] = (uint8_t)cb_decision(c, CX_G2 + ctx_set + (c_idx ? 4 : 0));
        int hidden = d->sign_hiding && (last_sig - first_sig > 3);
        uint32_t signs = 0; int nsigns = 0;
        for (int k = 15; k >= 0; k--) if (sig[k] && !(hidden && k == first_sig)) nsigns++;
        signs = cb_bypass_n(c, nsigns) << (32 - nsigns > 31 ? 0 : 32 - nsigns);
        if (nsigns == 0) signs = 0;
        int num = 0, sum = 0, rice = 0;
        for (int k = 15; k >= 0; k--) {
            if (!sig[k]) continue;
            int base = 1 + g1[k] + g2[k];
            int thresh = num < 8 ? (k == last_g1_pos ? 3 : 2) : 1;
            int a = base;
            if (base == thresh) {
                int pfx = 0;
                while (pfx < 4 && cb_bypass(c)) pfx++;
                int rem;
                if (pfx < 4) rem = (pfx << rice) + (int)cb_bypass_n(c, rice);
                else {
                    int kk = rice + 1; rem = 4 << rice;
                    while (cb_bypass(c)) { rem += 1 << kk; kk++; if (kk > 30) { set_err(d, "residual: escape too long"); return -1; } }
                    rem += (int)cb_bypass_n(c, kk);
                }
                a = base + rem;
                if (a > 3 * (1 << rice)) rice = rice < 4 ? rice + 1 : 4;
            }
            int neg;
            if (hidden && k == first_sig) neg = 0; /* fixed after the loop */
            else { neg = (int)(signs >> 31); signs <<= 1; }
            xc = (xs << 2) + g_scan[scan_idx][2][k][0]; yc = (ys << 2) + g_scan[scan_idx][2][k][1];
            if (a > 32767) a = 32767;
            lvl[yc * n + xc] = (int16_t)(neg ? -a : a);
            sum += a; num++;
        }
        if (hidden && (sum & 1)) {
            xc = (xs << 2) + g_scan[scan_idx][2][first_sig][0]; yc = (ys << 2) + g_scan[scan_idx][2][first_sig][1];
            lvl[yc * n + xc] = (int16_t)-lvl[yc * n + xc];
        }
    }
    return c->err ? -1 : 0;
}

/* reconstruct one TU: scaling + inverse transform + add to the prediction already stored in the picture — the decoder's own arithmetic
 * (hevc_dec_recon.c), not the oracle's */
static void add_residual(orc_decoder *d, int c_idx, int x, int y, int log2n, const int16_t *lvl, int qp, int dst)
{
    d2_residual_add(d->cur.pl[c_idx] + (size_t)y * d->cur.stride[c_idx] + x, d->cur.stride[c_idx], lvl, log2n, qp, d->bit_depth, dst);
}

static void intra_predict_block(orc_decoder *d, int c_idx, int x, int y, int log2n, int mode)
{
    pix ref[129];
    int s = c_idx ? 1 : 0;
    /* 8.4.4.2.2 with 6.4.1: a neighbour is available when it lies in the picture, in the current slice and tile, and earlier in decoding
     * order (avail_z: explicit tile boundaries, the slice's first row); then the substitution process */
    {
        const int n = 1 << log2n, total = 4 * n + 1, stride = d->cur.stride[c_idx];
        const pix *rec = d->cur.pl[c_idx];
        uint8_t av[4 * 32 + 1];
        int first = -1;
        for (int i = 0; i < total; i++) {
            int xn, yn;
            if (i < 2 * n) { xn = x - 1; yn = y + 2 * n - 1 - i; }
            else if (i == 2 * n) { xn = x - 1; yn = y - 1; }
            else { xn = x + (i - 2 * n - 1); yn = y - 1; }
            int ok = xn >= 0 && yn >= 0 && avail_z(d, x << s, y << s, xn << s, yn << s) && zaddr(d, xn << s, yn << s) < zaddr(d, x << s, y << s);
            av[i] = (uint8_t)ok;
            ref[i] = ok ? rec[(size_t)yn * stride + xn] : 0;
            if (ok && first < 0) first = i;
        }
        if (first < 0) for (int i = 0; i < total; i++) ref[i] = (pix)(1 << (d->bit_depth - 1));
        else {
            if (!av[0]) ref[0] = ref[first];
            for (int i = 1; i < total; i++) if (!av[i]) ref[i] = ref[i - 1];
        }
    }
    d2_intra_pred(ref, d->cur.pl[c_idx] + (size_t)y * d->cur.stride[c_idx] + x, d->cur.stride[c_idx], log2n, mode, c_idx, d->bit_depth, d->strong_intra);
}

static int intra_scan_idx(int log2n_block, int c_idx, int mode)
{
    if (log2n_block == 2 || (log2n_block == 3 && c_idx == 0)) {
        if (mode >= 6 && mode <= 14) return 2;
        if (mode >= 22 && mode <= 30) return 1;
    }
    return 0;
}

/* 7.3.8.8 transform_tree / 7.3.8.10 transform_unit (for the depths the encoder can produce) */
static int transform_tree(orc_decoder *d, int x0, int y0, int xb, int yb, int log2n, int depth, int blk,
                          int intra, int nxn, const uint8_t *luma_modes, int chroma_mode, int pcb, int pcr)
{
    cabac *c = &d->cb;
    int max_depth = intra ? d->th_intra + nxn : d->th_inter;
    int split;
    if (log2n <= d->log2_max_tb && log2n > d->log2_min_tb && depth < max_depth && !(nxn && depth == 0))
        split = cb_decision(c, CX_SPLIT_TU + 5 - log2n);
    else
        split = log2n > d->log2_max_tb || (nxn && depth == 0);
    int cbf_cb = 0, cbf_cr = 0;
    if (log2n > 2) {
        if (depth == 0 || pcb) cbf_cb = cb_decision(c, CX_CBF_CHROMA + depth);
        if (depth == 0 || pcr) cbf_cr = cb_decision(c, CX_CBF_CHROMA + depth);
    } else { cbf_cb = pcb; cbf_cr = pcr; }
    if (split) {
        int h = 1 << (log2n - 1);
        for (int k = 0; k < 4; k++)
            if (transform_tree(d, x0 + (k & 1) * h, y0 + (k >> 1) * h, x0, y0, log2n - 1, depth + 1, k, intra, nxn, luma_modes, chroma_mode, cbf_cb, cbf_cr))
                return -1;
        return 0;
    }
    int cbf_luma = 1;
    if (intra || depth != 0 || cbf_cb || cbf_cr) cbf_luma = cb_decision(c, CX_CBF_LUMA + (depth == 0 ? 1 : 0));
    int16_t lvl[32 * 32];
    orc_cu_rec *r = cu_at(d, x0, y0);
    int lmode = intra ? luma_modes[nxn ? blk : 0] : 1;
    /* luma */
    if (intra) intra_predict_block(d, 0, x0, y0, log2n, lmode);
    if (cbf_luma) {
        if (residual_coding(d, log2n, 0, intra ? intra_scan_idx(log2n, 0, lmode) : 0, lvl)) return -1;
        add_residual(d, 0, x0, y0, log2n, lvl, d->slice_qp, intra && log2n == 2);
        if (nxn) r->cbf_y4 |= (uint8_t)(1 << blk); else r->flags |= ORC_F_CBF_Y;
        if (nxn) r->flags |= ORC_F_CBF_Y;
    }
    /* chroma: with the luma TU unless luma is 4x4, then once after blkIdx 3 at the parent position */
    int do_chroma = log2n > 2 || blk == 3;
    if (do_chroma) {
        int xc = (log2n > 2 ? x0 : xb) >> 1, yc = (log2n > 2 ? y0 : yb) >> 1, l2c = log2n > 2 ? log2n - 1 : 2;
        int qpc = d2_chroma_qp(d->slice_qp);
        for (int ci = 1; ci < 3; ci++) {
            int cbf = ci == 1 ? cbf_cb : cbf_cr;
            if (intra) intra_predict_block(d, ci, xc, yc, l2c, chroma_mode);
            if (cbf) {
                if (residual_coding(d, l2c, ci, intra ? intra_scan_idx(l2c, ci, chroma_mode) : 0, lvl)) return -1;
                add_residual(d, ci, xc, yc, l2c, lvl, qpc, 0);
                cu_at(d, log2n > 2 ? x0 : xb, log2n > 2 ? y0 : yb)->flags |= ci == 1 ? ORC_F_CBF_CB : ORC_F_CBF_CR;
            }
        }
    }
    return 0;
}

/* spec-literal motion compensation with coordinate clamping — 8.5.3.3.3.1 */
static const int8_t kLT[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t kCT[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
/* predSamplesLX of one block: the 14-bit intermediate samples of 8.5.3.3.3.1 / .2 with reference coordinates clamped to the picture (8.5.3.3.3.1 xInt / yInt) */
static void mc_block14(orc_decoder *d, const picture *ref, int c_idx, int x, int y, int n, int mvx, int mvy, int16_t *out)
{
    int chroma = c_idx != 0, taps = chroma ? 4 : 8, half = chroma ? 1 : 3;
    int w = chroma ? d->w / 2 : d->w, h = chroma ? d->h / 2 : d->h;
    int fx = chroma ? mvx & 7 : mvx & 3, fy = chroma ? mvy & 7 : mvy & 3;
    int xi = x + (chroma ? mvx >> 3 : mvx >> 2), yi = y + (chroma ? mvy >> 3 : mvy >> 2);
    int shift1 = d->bit_depth - 8 < 4 ? d->bit_depth - 8 : 4, shift3 = 14 - d->bit_depth;
    const pix *rp = ref->pl[c_idx]; int rs = ref->stride[c_idx];
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            int v;
#define RS(xx, yy) rp[(size_t)CLIP3(0, h - 1, (yy)) * rs + CLIP3(0, w - 1, (xx))]
#define TAP(f, k) (chroma ? kCT[f][k] : kLT[f][k])
            if (!fx && !fy) v = RS(xi + i, yi + j) << shift3;
            else if (!fy) { int a = 0; for (int k = 0; k < taps; k++) a += TAP(fx, k) * RS(xi + i + k - half, yi + j); v = a >> shift1; }
            else if (!fx) { int a = 0; for (int k = 0; k < taps; k++) a += TAP(fy, k) * RS(xi + i, yi + j + k - half); v = a >> shift1; }
            else {
                int a = 0;
                for (int r = 0; r < taps; r++) {
                    int t = 0;
                    for (int k = 0; k < taps; k++) t += TAP(fx, k) * RS(xi + i + k - half, yi + j + r - half);
                    a += TAP(fy, r) * (t >> shift1);
                }
                v = a >> 6;
            }
#undef RS
#undef TAP
            out[j * n + i] = (int16_t)v;
        }
}
/* 8.5.3.3.4.2 default weighted sample prediction into the current picture: one list ((p + offset1) >> shift1) or the average of both */
static void mc_predict(orc_decoder *d, int c_idx, int x, int y, int n, int use0, int mv0x, int mv0y, int use1, int mv1x, int mv1y)
{
    int16_t p0[32 * 32], p1[32 * 32];
    const int maxv = (1 << d->bit_depth) - 1, shift1 = 14 - d->bit_depth, shift2 = 15 - d->bit_depth;
    if (use0) mc_block14(d, d->ref[0], c_idx, x, y, n, mv0x, mv0y, p0);
    if (use1) mc_block14(d, d->ref[1], c_idx, x, y, n, mv1x, mv1y, p1);
    pix *dp = d->cur.pl[c_idx] + (size_t)y * d->cur.stride[c_idx] + x;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            int v;
            if (use0 && use1) v = (p0[j * n + i] + p1[j * n + i] + (1 << (shift2 - 1))) >> shift2;
            else v = ((use0 ? p0 : p1)[j * n + i] + (1 << (shift1 - 1))) >> shift1;
            dp[j * d->cur.stride[c_idx] + i] = (pix)CLIP3(0, maxv, v);
        }
}

/* motion of a prediction block: prediction list utilization flags and the two vectors (reference indices are 0: one picture per list) */
typedef struct { int ok, f[2], mv[2][2]; } motion;
typedef struct { int ok, mvx, mvy; } mvcand;
static motion rec_motion(const orc_cu_rec *r)
{
    motion m;
    memset(&m, 0, sizeof m);
    m.ok = 1; m.f[0] = !(r->flags & ORC_F_NOL0); m.f[1] = (r->flags & ORC_F_L1) != 0;
    if (m.f[0]) { m.mv[0][0] = r->mvx; m.mv[0][1] = r->mvy; }
    if (m.f[1]) { m.mv[1][0] = orc_mv1x(r); m.mv[1][1] = orc_mv1y(r); }
    return m;
}
static motion nb_motion(orc_decoder *d, int xc, int yc, int xn, int yn)
{
    motion m;
    memset(&m, 0, sizeof m);
    if (!avail_z(d, xc, yc, xn, yn)) return m;
    const orc_cu_rec *r = cu_at(d, xn, yn);
    if (!(r->flags & ORC_F_INTER)) return m;
    return rec_motion(r);
}
static int same_motion(const motion *a, const motion *b)
{
    if (!a->ok || !b->ok || a->f[0] != b->f[0] || a->f[1] != b->f[1]) return 0;
    for (int l = 0; l < 2; l++) if (a->f[l] && (a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1])) return 0;
    return 1;
}
/* 8.5.3.2.2 - 8.5.3.2.5 merge candidates: spatial, combined bi-predictive (B slices), zero; no temporal candidate (the SPS switches it off) */
static void merge_list(orc_decoder *d, int x, int y, int n, motion out[5])
{
    motion a1 = nb_motion(d, x, y, x - 1, y + n - 1), b1 = nb_motion(d, x, y, x + n - 1, y - 1);
    motion b0 = nb_motion(d, x, y, x + n, y - 1), a0 = nb_motion(d, x, y, x - 1, y + n), b2 = nb_motion(d, x, y, x - 1, y - 1);
    /* pruning compares against the neighbour's raw availability, not its post-pruning flag (8.5.3.2.3) */
    int fa1 = a1.ok, fb1 = b1.ok && !same_motion(&b1, &a1), fb0 = b0.ok && !same_motion(&b0, &b1), fa0 = a0.ok && !same_motion(&a0, &a1);
    int fb2 = b2.ok && !same_motion(&b2, &a1) && !same_motion(&b2, &b1) && (fa0 + fa1 + fb0 + fb1 != 4);
    a1.ok = fa1; b1.ok = fb1; b0.ok = fb0; a0.ok = fa0; b2.ok = fb2;
    int k = 0;
    motion order[5] = {a1, b1, b0, a0, b2};
    for (int i = 0; i < 5 && k < d->max_merge; i++) if (order[i].ok) out[k++] = order[i];
    if (d->slice_type == 0 && k > 1 && k < d->max_merge) {          /* 8.5.3.2.4, Table 8-6 */
        static const uint8_t l0c[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1c[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
        const int orig = k;
        for (int c = 0; c < orig * (orig - 1) && k < d->max_merge; c++) {
            const motion *p = &out[l0c[c]], *q = &out[l1c[c]];
            if (!p->f[0] || !q->f[1]) continue;
            /* DiffPicOrderCnt(RefPicList0[refIdxL0], RefPicList1[refIdxL1]) != 0 || mvL0 != mvL1 */
            if (d->ref_poc[0] == d->ref_poc[1] && p->mv[0][0] == q->mv[1][0] && p->mv[0][1] == q->mv[1][1]) continue;
            motion m;
            memset(&m, 0, sizeof m);
            m.ok = 1; m.f[0] = m.f[1] = 1; m.mv[0][0] = p->mv[0][0]; m.mv[0][1] = p->mv[0][1]; m.mv[1][0] = q->mv[1][0]; m.mv[1][1] = q->mv[1][1];
            out[k++] = m;
        }
    }
    while (k < 5) { memset(&out[k], 0, sizeof out[k]); out[k].ok = 1; out[k].f[0] = 1; out[k].f[1] = d->slice_type == 0; k++; }      /* 8.5.3.2.5 */
}
/* 8.5.3.2.7, one neighbour for list X: first the vector that refers to the SAME picture as RefPicListX[0] (its list-X vector, else its list-Y vector),
 * in the second pass any vector, scaled by the ratio of the POC distances (equations 8-179 .. 8-183) */
static mvcand amvp_from(const orc_decoder *d, const motion *m, int lx, int scaled_pass)
{
    mvcand c = {0, 0, 0};
    if (!m->ok) return c;
    for (int t = 0; t < 2; t++) {
        const int l = t ? !lx : lx;
        if (!m->f[l]) continue;
        if (d->ref_poc[l] == d->ref_poc[lx]) { c.ok = 1; c.mvx = m->mv[l][0]; c.mvy = m->mv[l][1]; return c; }
        if (!scaled_pass) continue;
        const int td = CLIP3(-128, 127, d->poc - d->ref_poc[l]), tb = CLIP3(-128, 127, d->poc - d->ref_poc[lx]);
        const int tx = (16384 + (iabs(td) >> 1)) / td, dsf = CLIP3(-4096, 4095, (tb * tx + 32) >> 6);
        for (int k = 0; k < 2; k++) {
            const int p = dsf * m->mv[l][k], r = (iabs(p) + 127) >> 8, v = CLIP3(-32768, 32767, p < 0 ? -r : r);
            if (k) c.mvy = v; else c.mvx = v;
        }
        c.ok = 1;
        return c;
    }
    return c;
}
/* 8.5.3.2.6 - 8.5.3.2.7 AMVP candidates of list X (spatial; the SPS switches the temporal one off) */
static void amvp_list(orc_decoder *d, int x, int y, int n, int lx, mvcand out[2])
{
    motion a0 = nb_motion(d, x, y, x - 1, y + n), a1 = nb_motion(d, x, y, x - 1, y + n - 1);
    motion b0 = nb_motion(d, x, y, x + n, y - 1), b1 = nb_motion(d, x, y, x + n - 1, y - 1), b2 = nb_motion(d, x, y, x - 1, y - 1);
    const int scaled = a0.ok || a1.ok;   /* isScaledFlagLX; 6.4.2 availability already excludes intra neighbours */
    const motion *as[2] = {&a0, &a1}, *bs[3] = {&b0, &b1, &b2};
    mvcand a = {0, 0, 0}, b = {0, 0, 0};
    for (int k = 0; k < 2 && !a.ok; k++) a = amvp_from(d, as[k], lx, 0);
    for (int k = 0; k < 2 && !a.ok; k++) a = amvp_from(d, as[k], lx, 1);
    for (int k = 0; k < 3 && !b.ok; k++) b = amvp_from(d, bs[k], lx, 0);
    if (!scaled && b.ok) a = b;
    if (!scaled) {
        b.ok = 0;
        for (int k = 0; k < 3 && !b.ok; k++) b = amvp_from(d, bs[k], lx, 1);
    }
    int k = 0;
    if (a.ok) out[k++] = a;
    if (b.ok && !(a.ok && a.mvx == b.mvx && a.mvy == b.mvy)) out[k++] = b;
    while (k < 2) { out[k].ok = 1; out[k].mvx = 0; out[k].mvy = 0; k++; }
}

static int read_mvd(cabac *c, int *dx, int *dy)
{
    int g0x = cb_decision(c, CX_MVD0), g0y = cb_decision(c, CX_MVD0), g1x = 0, g1y = 0;
    if (g0x) g1x = cb_decision(c, CX_MVD1);
    if (g0y) g1y = cb_decision(c, CX_MVD1);
    int v[2] = {0, 0}, g0[2] = {g0x, g0y}, g1[2] = {g1x, g1y};
    for (int i = 0; i < 2; i++) {
        if (!g0[i]) continue;
        int a = 1;
        if (g1[i]) { int k = 1, r = 0; while (cb_bypass(c)) { r += 1 << k; k++; if (k > 20) return -1; } r += (int)cb_bypass_n(c, k); a = r + 2; }
        v[i] = cb_bypass(c) ? -a : a;
    }
    *dx = v[0]; *dy = v[1];
    return 0;
}

static int coding_unit(orc_decoder *d, int x0, int y0, int log2n)
{
    cabac *c = &d->cb;
    int n = 1 << log2n, w8 = d->w >> 3;
    int skip = 0, intra = d->slice_type == 2, nxn = 0, merge = 0, merge_idx = 0;
    orc_cu_rec rec;
    memset(&rec, 0, sizeof rec);
    rec.log2_size = (uint8_t)log2n; rec.qp = (uint8_t)d->slice_qp;
    if (d->slice_type != 2) {
        int l = avail_z(d, x0, y0, x0 - 1, y0) && d->skip8[(y0 >> 3) * w8 + ((x0 - 1) >> 3)];
        int a = avail_z(d, x0, y0, x0, y0 - 1) && d->skip8[((y0 - 1) >> 3) * w8 + (x0 >> 3)];
        skip = cb_decision(c, CX_SKIP + l + a);
    }
    uint8_t lmodes[4] = {1, 1, 1, 1};
    int chroma_mode = 1;
    if (skip) {
        merge = 1;
        if (d->max_merge > 1) { while (merge_idx < d->max_merge - 1 && (merge_idx == 0 ? cb_decision(c, CX_MERGE_IDX) : cb_bypass(c))) merge_idx++; }
    } else {
        if (d->slice_type != 2) intra = cb_decision(c, CX_PRED_MODE);
        if (!intra || log2n == d->log2_min_cb) {
            int bin = cb_decision(c, CX_PART_MODE);
            if (intra) nxn = !bin;
            else if (!bin) { set_err(d, "inter part_mode != 2Nx2N unsupported"); return -1; }
        }
        if (intra) {
            int parts = nxn ? 4 : 1, pn = nxn ? n / 2 : n;
            int prev[4], mpm_idx[4], rem[4];
            for (int k = 0; k < parts; k++) prev[k] = cb_decision(c, CX_PREV_INTRA);
            for (int k = 0; k < parts; k++) {
                mpm_idx[k] = rem[k] = 0;
                if (prev[k]) { if (cb_bypass(c)) mpm_idx[k] = 1 + cb_bypass(c); }
                else rem[k] = (int)cb_bypass_n(c, 5);
            }
            int cm = 4;
            for (int k = 0; k < parts; k++) {
                int xp = x0 + (k & 1) * pn, yp = y0 + (k >> 1) * pn;
                /* 8.4.2 */
                int ca = 1, cbm = 1;
                if (avail_z(d, xp, yp, xp - 1, yp)) {
                    const orc_cu_rec *r = (xp - 1 >= x0 && yp >= y0) ? &rec : cu_at(d, xp - 1, yp);
                    if (!(r->flags & ORC_F_INTER)) ca = (xp - 1 >= x0 && yp >= y0) ? lmodes[(k & 2) | 0] : r->intra_mode[r->flags & ORC_F_NXN ? ((((yp) >> 2) & 1) * 2 + 1) : 0];
                }
                if (avail_z(d, xp, yp, xp, yp - 1) && ((yp - 1) >> d->log2_ctb) == (yp >> d->log2_ctb)) {
                    const orc_cu_rec *r = (yp - 1 >= y0 && xp >= x0) ? &rec : cu_at(d, xp, yp - 1);
                    if (!(r->flags & ORC_F_INTER)) cbm = (yp - 1 >= y0 && xp >= x0) ? lmodes[k & 1] : r->intra_mode[r->flags & ORC_F_NXN ? (2 + (((xp) >> 2) & 1)) : 0];
                }
                int cand[3];
                if (ca == cbm) {
                    if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
                    else { cand[0] = ca; cand[1] = 2 + ((ca + 29) & 31); cand[2] = 2 + ((ca - 2 + 1) & 31); }
                } else { cand[0] = ca; cand[1] = cbm; cand[2] = (ca != 0 && cbm != 0) ? 0 : (ca != 1 && cbm != 1) ? 1 : 26; }
                int mode;
                if (prev[k]) mode = cand[mpm_idx[k]];
                else {
                    if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
                    if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
                    if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
                    mode = rem[k];
                    for (int i = 0; i < 3; i++) if (mode >= cand[i]) mode++;
                }
                lmodes[k] = (uint8_t)mode;
                if (!nxn) lmodes[1] = lmodes[2] = lmodes[3] = (uint8_t)mode;
            }
            if (cb_decision(c, CX_CHROMA_MODE)) cm = (int)cb_bypass_n(c, 2);
            static const uint8_t cmap[4] = {0, 26, 10, 1};
            if (cm == 4) chroma_mode = lmodes[0];
            else { chroma_mode = cmap[cm]; if (chroma_mode == lmodes[0]) chroma_mode = 34; }
        } else {
            merge = cb_decision(c, CX_MERGE_FLAG);
            if (merge) { if (d->max_merge > 1) while (merge_idx < d->max_merge - 1 && (merge_idx == 0 ? cb_decision(c, CX_MERGE_IDX) : cb_bypass(c))) merge_idx++; }
        }
    }
    rec.flags = (uint8_t)((intra ? 0 : ORC_F_INTER) | (nxn ? ORC_F_NXN : 0));
    memcpy(rec.intra_mode, lmodes, 4); rec.chroma_mode = (uint8_t)chroma_mode;
    if (!intra) {
        motion m;
        memset(&m, 0, sizeof m);
        if (merge) { motion l[5]; merge_list(d, x0, y0, n, l); m = l[merge_idx]; }
        else {
            /* 7.3.8.6 prediction_unit: inter_pred_idc (B slices; 9.3.4.2.2: first bin by CtDepth since nPbW + nPbH != 12), then per list mvd_coding and
             * mvp_lX_flag (ref_idx_lX is absent: one active picture per list) */
            m.ok = 1; m.f[0] = 1; m.f[1] = 0;
            if (d->slice_type == 0) {
                if (cb_decision(c, CX_INTER_DIR + (d->log2_ctb - log2n))) m.f[1] = 1;               /* PRED_BI */
                else if (cb_decision(c, CX_INTER_DIR + 4)) { m.f[0] = 0; m.f[1] = 1; }             /* PRED_L1 */
            }
            for (int lx = 0; lx < 2; lx++) {
                if (!m.f[lx]) continue;
                int dx = 0, dy = 0;
                if (!(lx == 1 && d->mvd_l1_zero && m.f[0]) && read_mvd(c, &dx, &dy)) { set_err(d, "mvd escape too long"); return -1; }
                int flag = cb_decision(c, CX_MVP);
                mvcand l[2]; amvp_list(d, x0, y0, n, lx, l);
                m.mv[lx][0] = l[flag].mvx + dx; m.mv[lx][1] = l[flag].mvy + dy;
            }
        }
        if (m.f[0]) { rec.mvx = (int16_t)m.mv[0][0]; rec.mvy = (int16_t)m.mv[0][1]; } else rec.flags |= ORC_F_NOL0;
        if (m.f[1]) { rec.flags |= ORC_F_L1; orc_set_mv1(&rec, m.mv[1][0], m.mv[1][1]); }
        if ((m.f[0] && !d->ref[0]) || (m.f[1] && !d->ref[1])) { set_err(d, "inter block without its reference picture"); return -1; }
        mc_predict(d, 0, x0, y0, n, m.f[0], m.mv[0][0], m.mv[0][1], m.f[1], m.mv[1][0], m.mv[1][1]);
        mc_predict(d, 1, x0 >> 1, y0 >> 1, n >> 1, m.f[0], m.mv[0][0], m.mv[0][1], m.f[1], m.mv[1][0], m.mv[1][1]);
        mc_predict(d, 2, x0 >> 1, y0 >> 1, n >> 1, m.f[0], m.mv[0][0], m.mv[0][1], m.f[1], m.mv[1][0], m.mv[1][1]);
    }
    for (int yy = 0; yy < n; yy += 8)
        for (int xx = 0; xx < n; xx += 8) { *cu_at(d, x0 + xx, y0 + yy) = rec; d->skip8[((y0 + yy) >> 3) * w8 + ((x0 + xx) >> 3)] = (uint8_t)skip; }
    int root_cbf = 1;
    if (skip) root_cbf = 0;
    else if (!intra && !merge) root_cbf = cb_decision(c, CX_RQT_ROOT);   /* merge 2Nx2N: inferred 1 (7.3.8.5) */
    if (root_cbf) {
        if (transform_tree(d, x0, y0, x0, y0, log2n, 0, 0, intra, nxn, lmodes, chroma_mode, 0, 0)) return -1;
        orc_cu_rec *first = cu_at(d, x0, y0);
        for (int yy = 0; yy < n; yy += 8)
            for (int xx = 0; xx < n; xx += 8) { orc_cu_rec *r = cu_at(d, x0 + xx, y0 + yy); r->flags = first->flags; r->cbf_y4 = first->cbf_y4; }
    }
    return c->err ? -1 : 0;
}

static int coding_quadtree(orc_decoder *d, int x0, int y0, int log2n, int depth)
{
    int n = 1 << log2n, split, w8 = d->w >> 3;
    if (x0 + n <= d->w && y0 + n <= d->h && log2n > d->log2_min_cb) {
        int l = avail_z(d, x0, y0, x0 - 1, y0) && d->depth8[(y0 >> 3) * w8 + ((x0 - 1) >> 3)] > depth;
        int a = avail_z(d, x0, y0, x0, y0 - 1) && d->depth8[((y0 - 1) >> 3) * w8 + (x0 >> 3)] > depth;
        split = cb_decision(&d->cb, CX_SPLIT_CU + l + a);
    } else split = log2n > d->log2_min_cb;
    if (split) {
        int h = n >> 1;
        for (int k = 0; k < 4; k++) {
            int x1 = x0 + (k & 1) * h, y1 = y0 + (k >> 1) * h;
            if (x1 < d->w && y1 < d->h && coding_quadtree(d, x1, y1, log2n - 1, depth + 1)) return -1;
        }
        return 0;
    }
    for (int yy = 0; yy < n; yy += 8)
        for (int xx = 0; xx < n; xx += 8) d->depth8[((y0 + yy) >> 3) * w8 + ((x0 + xx) >> 3)] = (uint8_t)depth;
    return coding_unit(d, x0, y0, log2n);
}

/* 7.3.8.3 sao() */
static void parse_sao(orc_decoder *d, int rx, int ry)
{
    cabac *c = &d->cb;
    int wc = (d->w + ORC_CTU - 1) >> ORC_CTU_LOG2;
    orc_sao_ctu *o = &d->sao[ry * wc + rx];
    memset(o, 0, sizeof *o);
    /* merge candidates must lie in the same slice and tile (7.3.8.3) */
    if (rx > d->col_bd[tile_col_of(d, rx)] && cb_decision(c, CX_SAO_MERGE)) { *o = d->sao[ry * wc + rx - 1]; goto mask; }
    if (ry > d->row_bd[tile_row_of(d, ry)] && ry > (d->slice_y0 >> ORC_CTU_LOG2) && cb_decision(c, CX_SAO_MERGE)) { *o = d->sao[(ry - 1) * wc + rx]; goto mask; }   /* the CTB above must be in this slice and tile */
    for (int ci = 0; ci < 3; ci++) {
        if ((ci == 0 && !d->sao_luma) || (ci > 0 && !d->sao_chroma)) continue;
        int t = ci ? 1 : 0;
        if (ci < 2) { o->type[t] = 0; if (cb_decision(c, CX_SAO_TYPE)) o->type[t] = (uint8_t)(cb_bypass(c) ? 2 : 1); }
        if (!o->type[t]) continue;
        int a[4], cmax = (1 << ((d->bit_depth < 10 ? d->bit_depth : 10) - 5)) - 1;
        for (int i = 0; i < 4; i++) { a[i] = 0; while (a[i] < cmax && cb_bypass(c)) a[i]++; }
        if (o->type[t] == 1) {
            for (int i = 0; i < 4; i++) if (a[i] && cb_bypass(c)) a[i] = -a[i];
            o->band_pos[ci] = (uint8_t)cb_bypass_n(c, 5);
        } else {
            if (ci == 0) o->eo_class[0] = (uint8_t)cb_bypass_n(c, 2);
            if (ci == 1) o->eo_class[1] = (uint8_t)cb_bypass_n(c, 2);
            a[2] = -a[2]; a[3] = -a[3];
        }
        for (int i = 0; i < 4; i++) o->offset[ci][i] = (int8_t)a[i];
    }
mask:
    if (!d->sao_luma) o->type[0] = 0;
    if (!d->sao_chroma) o->type[1] = 0;
}

/* number of emulation prevention bytes removed before rbsp offset r of the current NAL */
static size_t epb_before(const orc_decoder *d, size_t r)
{
    size_t k = 0;
    while ((int)k < d->n_epb && d->epb[k] <= r + k) k++;
    return k;
}

/* the picture under construction is complete: in-loop filters, then it becomes a reference.  Slices are independent bands of CTU rows
 * (pps_loop_filter_across_slices_enabled_flag = 0 whenever a picture has more than one): deblocking leaves the edges on a slice boundary
 * alone and SAO sees samples of another slice as absent (8.7.2 / 8.7.3), which is what filtering every band as a picture of its own does. */
static void finish_picture(orc_decoder *d)
{
    if (!d->cur_valid) return;
    picture out; memset(&out, 0, sizeof out);
    if (d->pic_sao) { alloc_pic(d, &out); out.poc = d->cur.poc; out.seq = d->cur.seq; out.in_dpb = d->cur.in_dpb; }
    {   /* in-loop filters over the whole picture with the slice / tile boundary rules of 8.7.2 / 8.7.3 (hevc_dec_recon.c: the decoder's own) */
        d2_picture_info pi;
        memset(&pi, 0, sizeof pi);
        pi.w = d->w; pi.h = d->h; pi.bit_depth = d->bit_depth; pi.cu = d->cu; pi.sao = d->sao;
        pi.cb_qp_offset = d->cb_off; pi.cr_qp_offset = d->cr_off;
        pi.n_bands = d->n_bands; pi.band_row0 = d->band_row0; pi.band_lf_across = d->band_lf_across;
        pi.tile_cols = d->tiles ? d->tile_cols : 1; pi.tile_rows = d->tiles ? d->tile_rows : 1; pi.lf_across_tiles = d->tiles ? d->lf_across_tiles : 1;
        pi.col_bd = d->col_bd; pi.row_bd = d->row_bd;
        d2_deblock_picture(&pi, d->cur.pl[0], d->cur.pl[1], d->cur.pl[2], d->cur.stride[0], d->cur.stride[1]);
        if (d->pic_sao)
            d2_sao_picture(&pi, d->cur.pl[0], d->cur.pl[1], d->cur.pl[2], d->cur.stride[0], d->cur.stride[1], out.pl[0], out.pl[1], out.pl[2], out.stride[0], out.stride[1]);
    }
    if (d->pic_sao) { free_pic(&d->cur); d->cur = out; }
    if (d->band_row0[d->n_bands] != ((d->h + ORC_CTU - 1) >> ORC_CTU_LOG2)) set_err(d, "picture incomplete: slices cover %d of %d CTU rows", d->band_row0[d->n_bands], (d->h + ORC_CTU - 1) >> ORC_CTU_LOG2);
    if (d->n_pics == d->cap_pics) { d->cap_pics = d->cap_pics ? d->cap_pics * 2 : 16; d->pics = (picture *)realloc(d->pics, sizeof(picture) * d->cap_pics); }
    d->pics[d->n_pics++] = d->cur; d->cur_valid = 0;
}

static int decode_slice(orc_decoder *d, const uint8_t *rbsp, size_t n, int nal_type)
{
    if (!d->have_sps || !d->have_pps) { set_err(d, "slice before parameter sets"); return -1; }
    bitrd b = {rbsp, n, 0, 0};
    const int first = br_bit(&b);        /* first_slice_segment_in_pic_flag */
    d->cur_nal = nal_type;
    int irap = nal_type >= 16 && nal_type <= 23, idr = nal_type == 19 || nal_type == 20;
    if (irap) br_bit(&b);
    if (activate_pps(d, (int)br_ue(&b))) { set_err(d, "slice: pps id"); return -1; }
    const int wc = (d->w + ORC_CTU - 1) >> ORC_CTU_LOG2, hc = (d->h + ORC_CTU - 1) >> ORC_CTU_LOG2;
    int addr = 0;
    if (!first) {                        /* dependent_slice_segments_enabled_flag is 0: slice_segment_address follows, Ceil(Log2(PicSizeInCtbsY)) bits */
        int nb = 0;
        while ((1 << nb) < wc * hc) nb++;
        addr = (int)br_u(&b, nb);
        if (!d->cur_valid) { set_err(d, "slice segment address %d without a first slice", addr); return -1; }
        if (addr % wc || addr / wc != d->band_row0[d->n_bands]) { set_err(d, "slice at CTB %d: only consecutive slices of whole CTB rows are decoded", addr); return -1; }
    } else finish_picture(d);
    d->slice_type = (int)br_ue(&b);
    if (d->slice_type < 0 || d->slice_type > 2) { set_err(d, "slice_type %d", d->slice_type); return -1; }
    if (irap && d->slice_type != 2) { set_err(d, "IRAP picture with a P / B slice"); return -1; }
    int poc = 0;
    if (!idr) {
        /* 8.3.1: POC from the previous picture at TemporalId 0 that is not a sub-layer non-reference picture (TRAIL_N is one) */
        int lsb = (int)br_u(&b, d->poc_bits);
        int prev = d->prev_tid0_poc;
        int maxl = 1 << d->poc_bits, prev_lsb = prev & (maxl - 1), prev_msb = prev - prev_lsb, msb;
        if (lsb < prev_lsb && prev_lsb - lsb >= maxl / 2) msb = prev_msb + maxl;
        else if (lsb > prev_lsb && lsb - prev_lsb > maxl / 2) msb = prev_msb - maxl;
        else msb = prev_msb;
        poc = msb + lsb;
        if (!br_bit(&b)) { set_err(d, "slice: explicit RPS unsupported"); return -1; }
        if (d->num_strps > 1) { int nb = 0; while ((1 << nb) < d->num_strps) nb++; d->ref_idx = (int)br_u(&b, nb); } else d->ref_idx = 0;
        if (d->ref_idx >= d->num_strps) { set_err(d, "slice: short_term_ref_pic_set_idx"); return -1; }
    }
    if (!first && poc != d->poc) { set_err(d, "slices of one picture disagree on the picture order count"); return -1; }
    d->poc = poc;
    if (first) {
        if (idr) d->seq++;
        /* 8.3.2 reference picture set: pictures of this coded video sequence outside the set can never be referenced again; the set's pictures must
         * all be present (no picture is ever missing in these streams: a missing one is an error).  8.3.4 reference picture lists with one active entry
         * each: RefPicList0 = StCurrBefore then StCurrAfter, RefPicList1 = StCurrAfter then StCurrBefore */
        const picture *before = NULL, *after = NULL;
        int before_poc = 0, after_poc = 0;
        for (int i = 0; i < d->n_pics; i++) {
            picture *p = &d->pics[i];
            if (!p->in_dpb) continue;
            int keep = 0;
            if (!idr && p->seq == d->seq) {
                for (int k = 0; k < d->strps_neg[d->ref_idx]; k++)
                    if (p->poc == poc + d->strps_delta[d->ref_idx][k]) { keep = 1; if (d->strps_used[d->ref_idx][k] && !before) { before = p; before_poc = p->poc; } }
                for (int k = 0; k < d->strps_pos[d->ref_idx]; k++)
                    if (p->poc == poc + d->strps_pdelta[d->ref_idx][k]) { keep = 1; if (d->strps_pused[d->ref_idx][k] && !after) { after = p; after_poc = p->poc; } }
            }
            p->in_dpb = keep;
        }
        /* the nearest picture first: negative deltas are listed closest first, so "first match in DPB order" is not enough when a set holds several */
        if (!idr) {
            for (int k = 0; k < d->strps_neg[d->ref_idx] && d->strps_used[d->ref_idx][k]; k++) {
                int found = 0;
                for (int i = 0; i < d->n_pics; i++) if (d->pics[i].in_dpb && d->pics[i].seq == d->seq && d->pics[i].poc == poc + d->strps_delta[d->ref_idx][k]) { if (k == 0) { before = &d->pics[i]; before_poc = d->pics[i].poc; } found = 1; }
                if (!found) { set_err(d, "reference picture poc %d of picture %d is not in the DPB", poc + d->strps_delta[d->ref_idx][k], poc); return -1; }
            }
            for (int k = 0; k < d->strps_pos[d->ref_idx] && d->strps_pused[d->ref_idx][k]; k++) {
                int found = 0;
                for (int i = 0; i < d->n_pics; i++) if (d->pics[i].in_dpb && d->pics[i].seq == d->seq && d->pics[i].poc == poc + d->strps_pdelta[d->ref_idx][k]) { if (k == 0) { after = &d->pics[i]; after_poc = d->pics[i].poc; } found = 1; }
                if (!found) { set_err(d, "reference picture poc %d of picture %d is not in the DPB", poc + d->strps_pdelta[d->ref_idx][k], poc); return -1; }
            }
        }
        d->ref[0] = before ? before : after; d->ref_poc[0] = before ? before_poc : after_poc;
        d->ref[1] = after ? after : before; d->ref_poc[1] = after ? after_poc : before_poc;
        if (nal_type != 0) d->prev_tid0_poc = poc;          /* TRAIL_N (0) is a sub-layer non-reference picture: it does not anchor later POCs */
    }
    d->sao_luma = d->sao_chroma = 0;
    if (d->sao_on) { d->sao_luma = br_bit(&b); d->sao_chroma = br_bit(&b); }
    d->max_merge = 5;
    d->mvd_l1_zero = 0;
    if (d->slice_type != 2) {
        if (br_bit(&b)) { set_err(d, "slice: num_ref_idx override"); return -1; }
        if (d->slice_type == 0) d->mvd_l1_zero = br_bit(&b);      /* (lists_modification / cabac_init / collocated / pred_weight_table: absent by PPS and SPS) */
        d->max_merge = 5 - (int)br_ue(&b);
        if (!d->ref[0] || (d->slice_type == 0 && !d->ref[1])) { set_err(d, "slice: empty reference picture list"); return -1; }
    }
    d->slice_qp = d->init_qp + br_se(&b);
    int slice_lf_across = d->lf_across;      /* 7.4.7.1: when absent, inferred equal to pps_loop_filter_across_slices_enabled_flag */
    if (d->lf_across) slice_lf_across = br_bit(&b);      /* slice_loop_filter_across_slices_enabled_flag: deblocking is on, so present */
    int n_entry = 0;
    uint32_t entry[20 * 22];
    if (d->tiles) {
        n_entry = (int)br_ue(&b);
        if (n_entry >= d->tile_cols * d->tile_rows || (n_entry + 1) % d->tile_cols) { set_err(d, "slice: %d entry points, tile grid %dx%d", n_entry, d->tile_cols, d->tile_rows); return -1; }
        if (n_entry > 0) {
            int len = 1 + (int)br_ue(&b);
            if (len > 32) { set_err(d, "slice: offset_len"); return -1; }
            for (int i = 0; i < n_entry; i++) entry[i] = 1 + br_u(&b, len);
        }
    }
    if (!br_bit(&b)) { set_err(d, "slice: byte_alignment bit"); return -1; }
    while (b.pos & 7) if (br_bit(&b)) { set_err(d, "slice: alignment zero bits"); return -1; }
    if (b.err) { set_err(d, "slice header truncated"); return -1; }
    put_kv(d, "slice.last_qp", d->slice_qp); put_kv(d, "slice.last_type", d->slice_type); put_kv(d, "slice.max_merge", d->max_merge);

    int w8 = d->w >> 3, h8 = d->h >> 3;
    if (first) {
        if (!d->cu) {
            d->cu = (orc_cu_rec *)calloc((size_t)w8 * h8, sizeof(orc_cu_rec));
            d->depth8 = (uint8_t *)calloc((size_t)w8 * h8, 1); d->skip8 = (uint8_t *)calloc((size_t)w8 * h8, 1);
            d->sao = (orc_sao_ctu *)calloc((size_t)wc * hc, sizeof(orc_sao_ctu));
        }
        memset(d->cu, 0, sizeof(orc_cu_rec) * w8 * h8); memset(d->depth8, 0, (size_t)w8 * h8); memset(d->skip8, 0, (size_t)w8 * h8);
        memset(d->sao, 0, sizeof(orc_sao_ctu) * wc * hc);
        alloc_pic(d, &d->cur); d->cur_valid = 1; d->cur.poc = d->poc; d->cur.seq = d->seq; d->cur.in_dpb = d->cur_nal != 0;
        d->n_bands = 0; d->band_row0[0] = 0;
        d->pic_sao = d->sao_luma || d->sao_chroma; d->pic_lf_across = d->lf_across;
    } else if ((d->sao_luma || d->sao_chroma) != d->pic_sao) { set_err(d, "slices of one picture disagree on SAO"); return -1; }
    if (d->n_bands >= 23) { set_err(d, "too many slices"); return -1; }
    build_scans();
    const int row0 = addr / wc;
    d->slice_y0 = row0 << ORC_CTU_LOG2;
    int ok = -1, row_end = row0;
    /* slice_segment_data (7.3.8.1) in tile scan; every tile is one CABAC substream that starts at its entry point.
     * Entry points count bytes of the NAL payload WITH emulation prevention bytes (7.4.7.1): map through epb[].
     * The slice holds whole tile rows (with tiles) or whole CTB rows (one tile): tiles t_first .. t_first + n_entry. */
    size_t data0 = b.pos >> 3;            /* rbsp offset of the slice segment data */
    size_t sub_start = data0;             /* rbsp offset of the current substream */
    int t_first = 0;
    if (d->tiles) {
        int ty0 = tile_row_of(d, row0);
        if (d->row_bd[ty0] != row0) { set_err(d, "slice at CTB row %d does not start a tile row", row0); goto done; }
        t_first = ty0 * d->tile_cols;
    }
    const int n_sub = n_entry + 1;
    for (int ti = 0; ti < n_sub; ti++) {
        int t = t_first + ti, tx = t % d->tile_cols, ty = t / d->tile_cols;
        size_t sub_end = n;
        if (ti < n_entry) {
            /* escaped start of this substream = its rbsp start + EPBs before it; add the signalled size; map back */
            size_t esc_start = sub_start + epb_before(d, sub_start), esc_end = esc_start + entry[ti];
            size_t e = esc_end;           /* rbsp position r with r + epb_before(r) == esc_end */
            for (int k = 0; k < d->n_epb && d->epb[k] < esc_end; k++) e--;
            sub_end = e;
            if (sub_end > n || sub_end <= sub_start) { set_err(d, "slice: entry point %d out of range", ti); goto done; }
        }
        cb_init(&d->cb, rbsp + sub_start, sub_end - sub_start, d->slice_type == 2 ? 0 : d->slice_type == 1 ? 1 : 2, d->slice_qp);      /* 9.3.2.2: initType with cabac_init_flag 0 */
        /* one tile (no tiles in the PPS): CTB rows from row0 until end_of_slice_segment_flag */
        const int ry0 = d->tiles ? d->row_bd[ty] : row0, ry1 = d->tiles ? d->row_bd[ty + 1] : hc;
        int ended = 0;
        for (int ry = ry0; ry < ry1 && !ended; ry++)
            for (int rx = d->col_bd[tx]; rx < d->col_bd[tx + 1]; rx++) {
                if (d->sao_luma || d->sao_chroma) parse_sao(d, rx, ry);
                if (coding_quadtree(d, rx << ORC_CTU_LOG2, ry << ORC_CTU_LOG2, ORC_CTU_LOG2, 0)) goto done;
                int end = cb_terminate(&d->cb);
                int tile_last = ry == ry1 - 1 && rx == d->col_bd[tx + 1] - 1;
                if (end) {
                    /* legal only at the end of a CTB row that ends the slice's last substream */
                    if (ti != n_sub - 1 || rx != d->col_bd[tx + 1] - 1 || (d->tiles && !tile_last)) { set_err(d, "end_of_slice_segment_flag=1 at ctu (%d,%d)", rx, ry); goto done; }
                    ended = 1; row_end = ry + 1;
                    break;
                }
                if (tile_last && ti == n_sub - 1) { set_err(d, "end_of_slice_segment_flag=0 at ctu (%d,%d)", rx, ry); goto done; }
            }
        if (d->cb.err) { set_err(d, "slice data truncated (tile %d)", t); goto done; }
        if (ti < n_sub - 1) {
            if (ended) { set_err(d, "slice ended before its last substream"); goto done; }
            /* end_of_subset_one_bit, then byte_alignment().  The encoder's flush (9.3.4.5: 7-bit renormalisation, put_bits(1),
             * write_bits(((low >> 7) & 3) | 1, 2)) puts its final '1' -- which IS alignment_bit_equal_to_one -- on the last bit of
             * the 9-bit window the arithmetic decoder holds when it decodes the terminating bin (no renormalisation follows a
             * terminating 1).  So: the last bit read is 1, only zero bits follow in that byte, and the next byte is the entry point. */
            if (!cb_terminate(&d->cb)) { set_err(d, "end_of_subset_one_bit != 1 after tile %d", t); goto done; }
            {
                const uint8_t *sp = rbsp + sub_start;
                size_t next_bit = d->cb.pos * 8 + (d->cb.bits_left ? 8 - (size_t)d->cb.bits_left : 0), last = next_bit - 1;
                if (d->cb.err || last / 8 >= sub_end - sub_start) { set_err(d, "tile %d: substream overrun", t); goto done; }
                if (!((sp[last >> 3] >> (7 - (last & 7))) & 1)) { set_err(d, "tile %d: alignment_bit_equal_to_one missing", t); goto done; }
                if (sp[last >> 3] & ((1u << (7 - (last & 7))) - 1)) { set_err(d, "tile %d: non-zero alignment bits", t); goto done; }
                if ((last >> 3) + 1 != sub_end - sub_start) { set_err(d, "tile %d: substream is %zu bytes, entry point says %zu", t, (last >> 3) + 1, sub_end - sub_start); goto done; }
            }
        } else if (!ended) { set_err(d, "slice data ends without end_of_slice_segment_flag"); goto done; }
        sub_start = sub_end;
    }
    d->band_lf_across[d->n_bands] = (unsigned char)slice_lf_across;
    d->n_bands++;
    d->band_row0[d->n_bands] = row_end;
    ok = 0;
done:
    if (ok && d->cur_valid) { free_pic(&d->cur); d->cur_valid = 0; }
    return ok;
}

int orc_dec_decode(orc_decoder *d, const uint8_t *data, size_t size)
{
    size_t i = 0;
    int n_aud = 0, n_slices = 0, au_open = 0;
    uint8_t *rbsp = (uint8_t *)malloc(size + 8);
    while (i + 3 < size) {
        /* find start code */
        if (!(data[i] == 0 && data[i + 1] == 0 && data[i + 2] == 1)) { i++; continue; }
        size_t s = i + 3, e = s;
        while (e + 2 < size && !(data[e] == 0 && data[e + 1] == 0 && (data[e + 2] == 1 || (data[e + 2] == 0 && e + 3 < size && data[e + 3] == 1)))) e++;
        if (e + 2 >= size) e = size;
        /* NAL header + emulation prevention removal (7.3.1.1) */
        if (e - s < 2) { set_err(d, "short NAL"); break; }
        if (data[s] & 0x80) { set_err(d, "forbidden_zero_bit"); break; }
        int type = (data[s] >> 1) & 63, tid = data[s + 1] & 7;
        if (tid != 1 || ((data[s] & 1) << 5 | data[s + 1] >> 3) != 0) { set_err(d, "nuh layer/temporal id"); break; }
        size_t m = 0; int zeros = 0;
        d->n_epb = 0;
        for (size_t k = s + 2; k < e; k++) {
            if (zeros >= 2 && data[k] == 3) {
                if (d->n_epb == d->cap_epb) { d->cap_epb = d->cap_epb ? d->cap_epb * 2 : 64; d->epb = (size_t *)realloc(d->epb, sizeof(size_t) * d->cap_epb); }
                d->epb[d->n_epb++] = k - (s + 2);
                zeros = 0; continue;
            }
            if (zeros >= 2 && data[k] < 3) { set_err(d, "start-code emulation inside NAL type %d", type); goto out; }
            rbsp[m++] = data[k];
            zeros = data[k] == 0 ? zeros + 1 : 0;
        }
        bitrd b = {rbsp, m, 0, 0};
        int rc = 0;
        /* 7.4.2.4.4: an access unit delimiter, when present, is the first NAL unit of its access unit (one slice per picture here) */
        if (type == 35 && au_open) { set_err(d, "access unit delimiter is not the first NAL unit of its access unit"); break; }
        au_open = !(type == 0 || type == 1 || type == 19 || type == 20);
        if (au_open) finish_picture(d);          /* a non-VCL NAL unit after the slices: the picture before it is complete */
        if (type == 32) rc = parse_vps(d, &b);
        else if (type == 33) rc = parse_sps(d, &b);
        else if (type == 34) rc = parse_pps(d, &b);
        else if (type == 35) { n_aud++; put_kv(d, "aud.last_pic_type", br_u(&b, 3)); if (!br_trailing_ok(&b)) { set_err(d, "aud trailing"); rc = -1; } }
        else if (type == 39 || type == 40) rc = parse_sei(d, &b);
        else if (type == 0 || type == 1 || type == 19 || type == 20) { rc = decode_slice(d, rbsp, m, type); n_slices++; }
        else { set_err(d, "unsupported NAL type %d", type); rc = -1; }
        if (rc) break;
        i = e;
    }
out:
    if (!d->err[0]) finish_picture(d);
    free(rbsp);
    put_kv(d, "count.aud", n_aud); put_kv(d, "count.slices", n_slices);
    return d->err[0] ? -1 : d->n_pics;
}
