/* oracle/hevc_dec_recon.h — TEST INFRASTRUCTURE, NOT PRODUCT.  The decoder's own normative arithmetic (hevc_dec_recon.c): nothing here
 * calls into hevc_oracle.c.  The record types (orc_cu_rec, orc_sao_ctu, pix) are plain data shared through hevc_oracle.h. */
#ifndef HEVC_DEC_RECON_H
#define HEVC_DEC_RECON_H
#include "hevc_oracle.h"

/* what the in-loop filters need to know about the decoded picture */
typedef struct {
    int w, h, bit_depth;
    const orc_cu_rec *cu;              /* per 8x8 block, as parsed */
    const orc_sao_ctu *sao;            /* per CTB, as parsed (type 0 where the slice switched SAO off) */
    int cb_qp_offset, cr_qp_offset;    /* pps_cb_qp_offset / pps_cr_qp_offset (cQpPicOffset of 8.7.2.5.5) */
    int beta_offset_div2, tc_offset_div2;
    int n_bands;                       /* slices: full-width bands of CTB rows [band_row0[k], band_row0[k + 1]) */
    const int *band_row0;
    const unsigned char *band_lf_across;   /* per slice: slice_loop_filter_across_slices_enabled_flag */
    int tile_cols, tile_rows, lf_across_tiles;
    const int *col_bd, *row_bd;
    const int *poc_of_ref;             /* reserved for B slices: picture order counts of RefPicList0[0], RefPicList1[0] */
} d2_picture_info;

int  d2_chroma_qp(int qpi);
void d2_residual_add(pix *dst, int stride, const int16_t *lvl, int log2n, int qp, int bit_depth, int dst4);
void d2_residual(const int16_t *lvl, int32_t *res, int log2n, int qp, int bit_depth, int dst4);
void d2_intra_pred(const pix *ref, pix *dst, int stride, int log2n, int mode, int c_idx, int bit_depth, int strong_enabled);
int  d2_motion_differs(const orc_cu_rec *p, const orc_cu_rec *q, const int *poc_of_ref);
void d2_deblock_picture(const d2_picture_info *pi, pix *y, pix *u, pix *v, int stride, int cstride);
void d2_sao_picture(const d2_picture_info *pi, const pix *y, const pix *u, const pix *v, int stride, int cstride, pix *oy, pix *ou, pix *ov, int ostride, int ocstride);
#endif
