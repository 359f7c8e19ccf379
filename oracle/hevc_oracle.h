/* oracle/hevc_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar CPU restatement of the HEVC encode hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product (hevc_amd/)
 * never links or calls it.
 *
 * What it follows.  The reference (uingei/hevc) reaches this arithmetic through an external
 * `ffmpeg -c:v libx265` child (core/transcoder.py:412 codec name, :398-411 params, :463, :506); libx265 is
 * third-party, unpinned and absent from /root/reference and from this image, and the reference's own tests hold
 * no golden vector for it (tests/test_transcoder.py:32,47 assert only a status string).  PARITY UNPINNED against
 * libx265.  The normative (decoder-side) arithmetic therefore follows ITU-T H.265 (04/2013 and later) clause by
 * clause — each function names its clause — and is pinned by the closed-form known-answer tests in
 * tests/test_oracle_kat.py and by the independent decoder in hevc_dec.c (encode -> bitstream -> decode must
 * reproduce the encoder's reconstruction bit for bit).  Encoder-side choices (search, costs) are this build's own
 * and are DEFINED here: the HIP kernels must reproduce them exactly.
 *
 * All sample planes are uint16_t containers (8- and 10-bit alike) to keep the restatement single-typed.
 */
#ifndef HEVC_ORACLE_H
#define HEVC_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t pix;

#define ORC_CTU_LOG2 5
#define ORC_CTU 32
#define ORC_MINCU_LOG2 3
#define ORC_PAD 80            /* luma border of reference planes (chroma: half) */

/* per-8x8 block record, identical to the product's mihevc_cu_rec (include/mihevc.h) */
typedef struct {
    uint8_t log2_size;   /* CU size: 3,4,5 */
    uint8_t flags;       /* bit0 inter, bit1 cbf_y, bit2 cbf_cb, bit3 cbf_cr, bit4 intra NxN, bit5 (host) skip */
    uint8_t chroma_mode; /* actual chroma intra mode 0..34 */
    uint8_t qp;
    uint8_t intra_mode[4];
    int16_t mvx, mvy;    /* quarter-pel */
    uint8_t cbf_y4;      /* NxN: luma cbf of the four 4x4 TUs */
    uint8_t pad[3];
} orc_cu_rec;

#define ORC_F_INTER 1
#define ORC_F_CBF_Y 2
#define ORC_F_CBF_CB 4
#define ORC_F_CBF_CR 8
#define ORC_F_NXN 16
#define ORC_F_L1 32      /* inter CU of a B picture: list 1 is used; its vector lives in intra_mode[0..3] (unused by inter CUs) as two little-endian int16 */
#define ORC_F_NOL0 64    /* inter CU of a B picture: list 0 is NOT used (L1 only); neither flag: list 0 only, as in P pictures; ORC_F_L1 alone: bi-prediction */
static inline int orc_mv1x(const orc_cu_rec *r) { return (int16_t)(r->intra_mode[0] | (r->intra_mode[1] << 8)); }
static inline int orc_mv1y(const orc_cu_rec *r) { return (int16_t)(r->intra_mode[2] | (r->intra_mode[3] << 8)); }
static inline void orc_set_mv1(orc_cu_rec *r, int x, int y)
{
    r->intra_mode[0] = (uint8_t)(x & 255); r->intra_mode[1] = (uint8_t)((x >> 8) & 255); r->intra_mode[2] = (uint8_t)(y & 255); r->intra_mode[3] = (uint8_t)((y >> 8) & 255);
}

typedef struct {
    uint8_t type[2];      /* [0] luma, [1] chroma: 0 off, 1 band, 2 edge */
    uint8_t eo_class[2];
    uint8_t band_pos[3];
    int8_t  offset[3][4]; /* signed final SaoOffsetVal[1..4] */
    uint8_t pad;
} orc_sao_ctu;

/* integer cost parameters (computed on the host from QP; see hevc_amd/csrc/ratectl) */
typedef struct {
    int qp;            /* luma QP of the frame */
    int qp_c;          /* chroma QP (table-mapped, cb/cr offsets 0) */
    int bit_depth;
    int lambda_sad_q4; /* sqrt(lambda_mode) * 16 */
    int lambda_q4;     /* lambda_mode * 16  (SSE domain) */
    int me_range;      /* integer search +-range, <= 64 */
    int tile_cols, tile_rows; /* intra pictures: uniform tile grid (6.5.1); 0 or 1 = one tile.  Neighbours in another tile
                                * are unavailable for prediction (6.4.1), which is what shortens the CTU wavefront */
    int intra_nxn;            /* 1: every 8x8 intra CU is also tried as four 4x4 PUs (part_mode NxN, DST-VII luma TUs) */
    int intra_in_p;           /* 1: P pictures get a second pass that re-codes badly predicted CTUs as intra (see orc_analyze_inter_frame) */
    int pre_search;           /* 1: when no search centres are given, take them from a +-14 full search on the 1/4-size pictures (+-56 samples) */
    int rdo_zero;             /* 1: an inter TU whose levels cost more (lambda * bits) than the distortion they remove is coded as all-zero */
    int chroma_modes;         /* 1: 2Nx2N intra CUs choose intra_chroma_pred_mode among planar / vertical / horizontal / DC / DM by SATD */
    int mc_top, mc_bottom;    /* 1: the picture is a slice whose upper / lower neighbour is coded elsewhere: motion compensation must not read across that edge */
    int rdo_cg;               /* k > 0: RD zero-out of the 4x4 coefficient groups of inter TUs with lambda x k / 2 (see code_tu); 0: off */
} orc_params;

/* ---- primitives (clauses of H.265 in the .c) ---- */
void orc_fwd_transform(const int16_t *res, int rstride, int16_t *coef, int log2n, int dst, int bit_depth);
void orc_inv_transform(const int16_t *coef, int16_t *res, int rstride, int log2n, int dst, int bit_depth);
int  orc_quant(const int16_t *coef, int16_t *lvl, int log2n, int qp, int bit_depth, int intra);
void orc_dequant(const int16_t *lvl, int16_t *coef, int log2n, int qp, int bit_depth);
int  orc_chroma_qp(int qp_y);
void orc_transform_matrix(int16_t *out32x32);

void orc_intra_build_ref(const pix *rec, int stride, int x0, int y0, int log2n, int pic_w, int pic_h,
                         const uint8_t *avail_map, int map_stride, int c_idx, int bit_depth, pix *ref /*4N+1*/);
/* same with a uniform tile grid: samples of another tile are unavailable */
void orc_intra_build_ref_tiles(const pix *rec, int stride, int x0, int y0, int log2n, int pic_w, int pic_h,
                               int c_idx, int bit_depth, int tile_cols, int tile_rows, pix *ref);
/* first CTB column (row) of tile column (row) i of n over n_ctb CTBs, uniform spacing (6.5.1) */
static inline int orc_tile_bd(int i, int n, int n_ctb) { return i * n_ctb / n; }
static inline int orc_tile_of(int ctb, int n, int n_ctb) { int i = 0; while (i + 1 < n && orc_tile_bd(i + 1, n, n_ctb) <= ctb) i++; return i; }
void orc_intra_filter_ref(const pix *ref, pix *filt, int log2n, int mode, int c_idx, int bit_depth, int strong);
void orc_intra_pred(const pix *ref, pix *dst, int dstride, int log2n, int mode, int c_idx, int bit_depth);

void orc_interp_luma(const pix *ref, int rstride, int x, int y, int mvx, int mvy, int w, int h, int bit_depth,
                     pix *dst, int dstride);
void orc_interp_chroma(const pix *ref, int rstride, int xc, int yc, int mvx, int mvy, int wc, int hc, int bit_depth,
                       pix *dst, int dstride);
int  orc_sad(const pix *a, int as, const pix *b, int bs, int w, int h);
int  orc_satd(const pix *a, int as, const pix *b, int bs, int w, int h);
int  orc_mvd_bits(int d);

void orc_pad_plane(pix *p, int stride, int w, int h, int pad);

/* ---- frame-level stages: the definition of what each HIP stage must output ---- */
/* K1+K3: inter (P) frame.  ref_* are padded planes (ORC_PAD / ORC_PAD/2); rec_* receive the pre-deblock
 * reconstruction; coef_* receive levels in TU-local raster at picture coordinates. */
void orc_analyze_inter_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                             const pix *ref_y, const pix *ref_u, const pix *ref_v, int ref_stride, int ref_cstride,
                             int w, int h, const orc_params *prm, const int16_t *centers /*2 per CTU or NULL*/,
                             pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                             orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                             int32_t *me_dump /* optional: per CTU 21*(mvx,mvy,cost) after integer search, or NULL */,
                             uint64_t *est /* optional: picture rate estimate in 1/16 bit */);
/* K1+K3 for a B picture between two anchors: ref0_* = the anchor before it in display order (list 0), ref1_* = the one after it (list 1), both padded.
 * The quadtree is decided on the list-0 search exactly as in a P picture; every CU of the tree then also refines its list-1 vector and tries the
 * bi-prediction of the two refined vectors (8.5.3.3.4.2 default weighted average of the 14-bit predictions), and takes the cheapest of
 * SATD << 4 + lambda * (mvd bits + inter_pred_idc bins): list 0 (2 bins), list 1 (2), both (1); ties in that order.  centers0 / centers1: search
 * centres against the two anchors.  me_dump0 / me_dump1 as me_dump above. */
void orc_analyze_b_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                         const pix *ref0_y, const pix *ref0_u, const pix *ref0_v, const pix *ref1_y, const pix *ref1_u, const pix *ref1_v,
                         int ref_stride, int ref_cstride, int w, int h, const orc_params *prm, const int16_t *centers0, const int16_t *centers1,
                         pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                         orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, int32_t *me_dump0, int32_t *me_dump1, uint64_t *est);
/* 1/4-size picture: every sample the rounded mean of a 4x4 luma block reduced to 8 bits (w, h multiples of 4) */
void orc_lowres(const pix *src, int stride, int w, int h, int bit_depth, pix *dst /* (w/4) x (h/4), stride w/4 */);
/* per CTU search centre (integer luma samples, 2 per CTU) from a +-14 full search of its 8x8 low-resolution block */
void orc_pre_search(const pix *lsrc, const pix *lref, int lw, int lh, int16_t *centers);
void orc_pre_search_cost(const pix *lsrc, const pix *lref, int lw, int lh, int16_t *centers, uint32_t *costs);
/* K2+K3: intra (I) frame */
void orc_analyze_intra_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                             int w, int h, const orc_params *prm,
                             pix *rec_y, pix *rec_u, pix *rec_v, int rec_stride, int rec_cstride,
                             orc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint64_t *est);
/* K4a: deblocking, in place on rec_* (clause 8.7.2) */
void orc_deblock_frame(pix *rec_y, pix *rec_u, pix *rec_v, int stride, int cstride, int w, int h,
                       const orc_cu_rec *cu, int bit_depth, int deblock_chroma_qp_offset);
/* K4b: SAO decision + apply: reads deblocked dbk_*, writes out_* (clause 8.7.3) and sao params */
void orc_sao_frame(const pix *src_y, const pix *src_u, const pix *src_v, int src_stride, int src_cstride,
                   const pix *dbk_y, const pix *dbk_u, const pix *dbk_v, int stride, int cstride,
                   pix *out_y, pix *out_u, pix *out_v, int ostride, int ocstride,
                   int w, int h, const orc_params *prm, orc_sao_ctu *sao);
void orc_sao_apply_frame(const pix *dbk_y, const pix *dbk_u, const pix *dbk_v, int stride, int cstride,
                         pix *out_y, pix *out_u, pix *out_v, int ostride, int ocstride,
                         int w, int h, int bit_depth, const orc_sao_ctu *sao);

/* ---- hevc_dec.c: independent decoder for the syntax subset the encoder emits ---- */
typedef struct orc_decoder orc_decoder;
orc_decoder *orc_dec_open(void);
/* feed a whole Annex-B stream; returns number of decoded pictures or <0 on syntax error */
int  orc_dec_decode(orc_decoder *d, const uint8_t *data, size_t size);
int  orc_dec_info(const orc_decoder *d, int *w, int *h, int *bit_depth, int *conf_w, int *conf_h);
/* copy decoded picture idx (output order == decode order here) into 16-bit planes of the coded size */
int  orc_dec_get_frame(const orc_decoder *d, int idx, pix *y, pix *u, pix *v);
const char *orc_dec_error(const orc_decoder *d);
/* parsed header fields for conformance assertions in tests: name -> value, returns 0 if unknown */
int  orc_dec_query(const orc_decoder *d, const char *field, long long *value);
void orc_dec_close(orc_decoder *d);

#ifdef __cplusplus
}
#endif
#endif
