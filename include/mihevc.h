/* include/mihevc.h — C ABI of the MI355X-native HEVC encode path (libmihevc.so).
 *
 * This is the drop-in boundary.  The reference has no FFI: its encode step is the child process
 * `ffmpeg -c:v libx265|hevc_nvenc` spawned by run_ffmpeg (reference core/transcoder.py:497-535) with the operating
 * point built by build_ffmpeg_params (core/transcoder.py:357-412).  The entry points below are what a binding for
 * that step would call instead; hevc_amd/encoder.py is that binding (ctypes), INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions: plain C types only; every int-returning function returns 0 (MIHEVC_OK) or a negative mihevc_err;
 * nothing throws or aborts; distinct sessions may be used from distinct threads concurrently (the reference calls
 * convert_video from N QThreads, gui/mainwindow.py:289-301); one session is driven by one thread at a time.
 * ctypes releases the GIL during calls, which that threading model relies on.
 */
#ifndef MIHEVC_H
#define MIHEVC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIHEVC_ABI_VERSION 3

typedef enum {
    MIHEVC_OK = 0,
    MIHEVC_EAGAIN = -1,     /* receive_packet: nothing ready yet (send more frames or flush) */
    MIHEVC_EOF = -2,        /* receive_packet after flush: stream complete */
    MIHEVC_EINVAL = -3,     /* bad argument / unsupported configuration */
    MIHEVC_ENODEV = -4,     /* no gfx950 device (or HIP runtime unusable) */
    MIHEVC_ENOMEM = -5,
    MIHEVC_EDEVICE = -6,    /* HIP call failed; see mihevc_last_error */
    MIHEVC_ESTATE = -7      /* call sequence error (e.g. send after flush) */
} mihevc_err;

/* Encoder operating point.  Field-for-field this is what the reference passes to libx265 through
 * `-x265-params` (core/transcoder.py:398-411) plus the HDR10 set of core/utils.py:58-69. */
typedef struct mihevc_config {
    int32_t width, height;            /* display size; coded size is rounded up to a multiple of 8 + conformance window */
    int32_t fps_num, fps_den;         /* VUI/VPS timing */
    int32_t bit_depth;                /* 8 (Main) or 10 (Main10) */
    int32_t level_idc;                /* general_level_idc = 30 * level  (x265 level-idc) */
    int32_t tier;                     /* 0 main, 1 high                  (x265 tier) */
    int32_t crf;                      /* constant-quality target         (x265 crf); frame QP is derived from it */
    int32_t qp;                       /* >= 0: force this QP for P frames (I frames qp-3); -1: derive from crf */
    int32_t vbv_maxrate_kbps;         /* x265 vbv-maxrate; 0 = unconstrained */
    int32_t vbv_bufsize_kbits;        /* x265 vbv-bufsize */
    int32_t keyint, min_keyint;       /* closed GOP length (IDR period)  (x265 keyint / min-keyint) */
    int32_t colour_primaries, transfer, matrix;   /* VUI code points: 1/1/1 bt709, 9/16/9 HDR10 (x265 colorprim/transfer/colormatrix) */
    int32_t full_range;               /* 0: `-color_range tv` (core/transcoder.py:490) */
    int32_t chroma_loc;               /* -1: not signalled; 0..5 chroma_sample_loc_type (x265 chromaloc) */
    int32_t aud;                      /* emit access unit delimiters     (x265 aud) */
    int32_t repeat_headers;           /* VPS/SPS/PPS before every IDR    (x265 repeat-headers); the first IDR always has them */
    int32_t hdr10;                    /* emit SEI 137 + 144              (x265 hdr10 / master-display / max-cll) */
    uint16_t md_primaries[3][2];      /* G,B,R (x,y) in 0.00002 units */
    uint16_t md_white[2];
    uint32_t md_max_lum, md_min_lum;  /* 0.0001 cd/m2 */
    uint16_t max_cll, max_fall;
    int32_t me_range;                 /* integer search +-range (<= 64); 0 = default */
    int32_t gops_in_flight;           /* closed GOPs encoded in lock-step on the device; 0 = default */
    int32_t host_threads;             /* CABAC worker threads; 0 = default */
    int32_t sao;                      /* 1 (default -1 -> 1) enable SAO */
    int32_t profile_stages;           /* 1: bracket every stage launch with HIP events on the compute stream (mihevc_stats.stage_ms); 2: only the
                                       * dominant stage (inter_ctu) — ~180 instead of ~1250 events per 300 pictures, which cost 5 % of the throughput */
    int32_t intra_tiles;              /* 1 (default): IDR pictures use the largest uniform tile grid the level allows (PPS 1), which
                                       * cuts the intra CTU wavefront from W+2H to w+2h CTUs of one tile; 0: one tile */
    int32_t intra_nxn;                /* 1: 8x8 intra CUs are also tried as four 4x4 PUs (part_mode NxN, DST-VII 4x4 luma TUs).  Default 0:
                                       * on the bench clip the trial costs 2 ms per IDR picture (-15 % fps) and wins in 0.4 % of the CUs */
    int32_t intra_in_p;               /* 1: P pictures run an intra second pass over CTUs the reference predicts badly (two independent-set
                                       * rounds, so isolated CTUs and the first CTUs of a blob).  Default 0: each round costs one
                                       * serial CTU-program latency, 0.12 ms per 1080p picture, and the bench clip gains nothing from it */
    int32_t hrd;                      /* 1: HRD parameters in the VUI + buffering-period SEI at every IDR + picture-timing SEI per picture
                                       * (x265 hrd=1, part of the reference's HDR10 set, core/utils.py:66); needs vbv_maxrate/bufsize */
    int32_t pre_search;               /* 1 (default): search centres from a +-14 full search on the 1/4-size pictures, so the +-me_range
                                       * integer search follows motion up to +-56 samples; 0: centres at zero.  A session searches the 1/4-size SOURCE
                                       * picture against the SOURCE picture before it, for a whole chunk at once (beside the IDR step) */
    int32_t rdo_zero;                 /* 1 (default): inter TUs whose levels cost more (lambda x bits) than the distortion they remove are
                                       * coded as all-zero (whole 300-picture bench clip: -12 % bits at -0.14 dB at fixed QP 27, +0.02 dB at equal bitrate
                                       * under the rate controller) */
    int32_t chroma_modes;             /* 1 (default): 2Nx2N intra CUs choose intra_chroma_pred_mode among DM / planar / vertical / horizontal /
                                       * DC by SATD over Cb + Cr; 0: always DM */
    /* ---- one picture over several devices (BASELINE configs[4]): a session may code ONE SLICE — a full-width band of CTU rows — of
     * pictures pic_height high.  `height` is then the band's own height (32 x its CTU rows, the last band takes what is left), the
     * parameter sets describe the whole picture and are identical in every slice's session, the slice header carries the band's
     * slice_segment_address, in-loop filters stop at slice boundaries (pps_loop_filter_across_slices_enabled_flag = 0) and motion
     * vectors never reach across them (nothing is exchanged between the devices).  slice_count = 0 / 1: whole pictures. */
    int32_t pic_height;               /* height of the whole picture (display); used when slice_count > 1 */
    int32_t slice_count, slice_index;
    int32_t slice_ctu_rows[16];       /* CTU rows of every slice, top to bottom (sum = ceil(pic_height / 32)) */
    int32_t rate_share_q16;           /* share of vbv-maxrate / vbv-bufsize this slice plans with, 65536 = all (0 = all) */
    int32_t scenecut;                 /* 1 (default): an IDR picture where the picture changes (mean absolute difference of consecutive source pictures),
                                       * at least min_keyint pictures after the last one (x265 scenecut / min-keyint); 0: IDR every keyint pictures only */
    int32_t gop_balance;              /* 1 (default): a run of pictures between scene cuts / chunk ends is coded as the fewest GOPs keyint allows, of
                                       * near-equal length (300 pictures at keyint 90: IDR at 0, 75, 150, 225), so the GOP lanes of the device pipeline finish
                                       * together; 0: IDR every keyint pictures (0, 90, 180, 270).  The number of IDR pictures is the same either way */
    int32_t rdo_cg;                   /* k > 0: RD zero-out of the 4x4 coefficient groups of inter TUs ("RDOQ-lite"): a group of levels is dropped when the
                                       * squared error it removes is worth less than lambda x k / 2 x its bits.  Default 0 (off): over whole GOPs of the bench
                                       * clip it buys nothing that a higher QP would not (k = 2: -0.6 % bits / -0.006 dB, k = 5: -8.5 % / -0.16 dB at QP 27) */
    /* ---- ABI 3 ---- */
    int32_t p_tiles;                  /* P pictures as a uniform tile grid of their own (PPS 0), one CABAC substream and ONE HOST JOB per tile: what lets the host keep up
                                       * with the device when few pictures are in flight (4320p: one GOP lane, 1.5 Mbit of CABAC per picture; x265 gets the same from WPP
                                       * under `-threads 0`, core/transcoder.py:410-411).  -1 (default): one tile per 1920x1080 of picture (4320p 4x4, 2160p 2x2, up to
                                       * 1080p-class none: tiles that large cost ~0.2 % bits); 0: off; 1: as -1 but at least 2x2 when the level allows.  Motion
                                       * compensation, deblocking and SAO cross tile boundaries; merge / AMVP candidates and CABAC contexts do not */
    int32_t bframes;                  /* 0 (default) / 1 / -1.  1: every second picture of a closed GOP is a B picture between two anchors (x265 bframes; the reference's
                                       * preset=slow runs 4 with b-adapt, core/transcoder.py:399): coding order I0 P2 b1 P4 b3 ..., a B picture predicts from the anchor
                                       * before it (list 0), the one after it (list 1) or both (8.5.3.3.4.2), is never a reference itself (TRAIL_N) and takes QP + 2.
                                       * Packets come out in DECODING order with dts <= pts (mihevc_receive_packet), the MP4 writer adds the ctts box.
                                       * -1: decided per chunk (x265 b-adapt in spirit): a probe on the 1/4-size SOURCE pictures compares how well a picture is
                                       * matched by the picture one place and two places before it; B pictures where two places back is nearly as good (static and
                                       * translating content), none where it is not (zoom, fades: there anchors two pictures apart cost more than B pictures save).
                                       * mihevc_stats.reserved[0..2]: the last probe's two costs (1/1000 per CTU) and its decision */
    int32_t b_qp_offset;              /* QP of a B picture above the anchors around it; -1 (default): 2 (x265 pbratio 1.3) */
    int32_t slice_halo;               /* slice_count > 1 only.  1: the sessions of one picture's slices EXCHANGE rows (they find each other through slice_group and must
                                       * live in one process): the PAD rows of the final reconstruction either side of every seam, so motion vectors cross seams as in a
                                       * whole picture, and 8 rows of the pre-deblock reconstruction + one row of CU records, so deblocking and SAO run across the seams
                                       * (pps_loop_filter_across_slices_enabled_flag = 1); rate control takes its inputs summed over the slices: one plan per picture.
                                       * Point-to-point pulls out of the neighbour's device memory (xGMI peer access), no collective.  0: nothing is exchanged — motion
                                       * constrained slices, filters stop at the seams, every slice its own rate controller (round 2: -1.5 dB at 4320p over 8) */
    int32_t slice_group;              /* slice_halo: any non-zero number shared by the sessions of one picture's slices and by nobody else in the process */
} mihevc_config;

typedef struct mihevc_session mihevc_session;

typedef struct mihevc_stats {
    int64_t frames_in, frames_out, bytes_out;
    double  sse_y, sse_u, sse_v;      /* encoder reconstruction vs source (summed over frames_out), for PSNR */
    double  device_ms, entropy_ms;    /* accumulated device time (HIP events) and host CABAC time (sum over threads) */
    int32_t last_qp;
    int32_t reserved[7];              /* [0..2]: cfg.bframes = -1, the last probe; [3..5]: host microseconds of the chunks in front of their first launch, behind their last
                                       * kernel (last symbol copies + the entropy coding still open), and in all: where wall time that is not device time goes */
    /* per-stage device time, filled when cfg.profile_stages: sum of HIP-event intervals and number of launches.
     * index: 0 intra (plan + the dataflow launch of a step), 1 me_search, 2 inter_ctu, 3 deblock (V+H; only without SAO: with SAO the loop filter is one kernel, counted
     * under 4), 4 sao (the whole loop filter: deblock of the CTU's tile + decide + apply + squared error), 5 border pad, 6 unused since ABI 2 (the SSE fold runs on the copy
     * stream), 7 intra second pass of P pictures (two rounds).  One launch covers `pictures` pictures (the lock-step batch). */
    double  stage_ms[8];
    int64_t stage_launches[8];
    int64_t stage_pictures[8];
} mihevc_stats;

int  mihevc_abi_version(void);
int  mihevc_device_count(void);                      /* gfx950 devices visible; 0 when none / no runtime */
/* NUMA node of the host memory closest to `device` (its PCI function's numa_node in sysfs); -1 when the platform does not say.  A process that
 * drives one device (one process per GPU: bench.py, hevc_amd/batch.py) binds itself to that node's CPUs BEFORE its first mihevc_open, so the pinned
 * symbol buffers and the CABAC workers that read them sit next to the device (hevc_amd/utils.py: bind_to_device_node). */
int  mihevc_device_numa_node(int device);
void mihevc_config_default(mihevc_config *cfg);      /* 1080p30 8-bit SDR at the reference's operating point */
int  mihevc_open(const mihevc_config *cfg, int device, mihevc_session **out);
/* Caller-owned host planes (8-bit: 1 byte/sample, 10-bit: 2 bytes little endian), copied/uploaded before return. */
int  mihevc_send_frame(mihevc_session *s, const void *y, const void *u, const void *v,
                       int pitch_y, int pitch_c, int64_t pts);
/* The same without waiting for the upload: the copies are enqueued on the session's side stream and the call returns.  The caller's planes must stay
 * valid and unmodified until mihevc_sync_uploads() or mihevc_flush() has returned (a decoder that feeds the session from a ring of N frame buffers calls
 * mihevc_sync_uploads before it reuses the oldest).  For the copies to run as DMA beside the caller the planes have to be page-locked host memory
 * (hipHostMalloc / hipHostRegister; torch: pin_memory()); with pageable memory the call is correct but as slow as mihevc_send_frame. */
int  mihevc_send_frame_async(mihevc_session *s, const void *y, const void *u, const void *v,
                             int pitch_y, int pitch_c, int64_t pts);
int  mihevc_sync_uploads(mihevc_session *s);         /* every frame handed over so far has left the caller's buffers */
/* Frames already resident in device memory (same layout, device pointers): the benchmark path.  Stream ordering contract: the copy
 * into the session's own pitch-aligned picture is ENQUEUED on the session's stream and the call returns before it has run, and it is not
 * ordered against any stream of the caller.  So (1) the producer of y/u/v must have finished before the call (synchronise its stream or
 * event first), and (2) the three planes must stay valid and unmodified until the picture's packet has been received, or mihevc_flush
 * has returned, whichever comes first (host buffers of mihevc_send_frame may be reused as soon as that call returns).  When the display size
 * already is the coded size (multiples of 8) and planes / pitches are 4-byte aligned the session codes straight from the caller's planes. */
int  mihevc_send_frame_device(mihevc_session *s, const void *y, const void *u, const void *v,
                              int pitch_y, int pitch_c, int64_t pts);
/* n pictures that are ALREADY in device memory in one call (pts = first_pts, first_pts + 1, ...): mihevc_send_frame_device n times, for a caller that holds the pointers
 * in arrays anyway — a host language's call overhead (ctypes: ~3.5 us per call, 1 ms per 300-frame clip at 5000 fps) stays out of the loop.  Same ownership rules; stops
 * at the first error.  Chunks that fill up on the way are coded inside the call, as with the one-picture form. */
int  mihevc_send_frames_device(mihevc_session *s, int n, const void *const *y, const void *const *u, const void *const *v,
                               int pitch_y, int pitch_c, int64_t first_pts);
/* One access unit (Annex-B NAL units) in session-owned memory, valid until the next receive/close. */
int  mihevc_receive_packet(mihevc_session *s, const uint8_t **data, size_t *size,
                           int64_t *pts, int64_t *dts, int *keyframe);
int  mihevc_flush(mihevc_session *s);                /* no more input; drain with receive_packet until MIHEVC_EOF */
/* Give the session up: it fails every later call, and if it codes one slice of a picture with slice_halo, the sessions of the other slices
 * stop waiting for it (their calls return MIHEVC_EDEVICE).  For a driver whose thread hit an error; mihevc_close is still due. */
int  mihevc_abort(mihevc_session *s);
void mihevc_close(mihevc_session *s);
int  mihevc_get_stats(const mihevc_session *s, mihevc_stats *out);
/* VPS+SPS+PPS (Annex-B) for the muxer's hvcC box; valid until close. */
int  mihevc_get_headers(mihevc_session *s, const uint8_t **data, size_t *size);
/* Copy the reconstruction of output frame `index` (display order) as 16-bit planes of the CODED size; only
 * available when the session was opened with keep_recon (tests): returns MIHEVC_ESTATE otherwise. */
int  mihevc_set_keep_recon(mihevc_session *s, int keep);
int  mihevc_get_recon(mihevc_session *s, int64_t index, uint16_t *y, uint16_t *u, uint16_t *v);
int  mihevc_coded_size(const mihevc_session *s, int *w, int *h);
/* QP, slice type (2 = IDR, 1 = P) and coded size in bits (-1 while CABAC is still running) of output picture `index` */
int  mihevc_get_frame_info(mihevc_session *s, int64_t index, int *qp, int *slice_type, int64_t *bits);
const char *mihevc_strerror(int err);
const char *mihevc_last_error(const mihevc_session *s);

/* ---- integer cost parameters derived from a QP (shared by every stage; exported so tests can hand the same
 *      numbers to the oracle) ---- */
typedef struct mihevc_cost_params {
    int32_t qp, qp_c, bit_depth, lambda_sad_q4, lambda_q4, me_range;
    int32_t tile_cols, tile_rows;     /* intra pictures: uniform tile grid (0/1 = one tile); see mihevc_tile_grid */
    int32_t intra_nxn;                /* 1: try part_mode NxN (four 4x4 PUs, DST-VII) for 8x8 intra CUs */
    int32_t intra_in_p;               /* 1: mihevc_k_inter_frame also runs the intra second pass of P pictures */
    int32_t pre_search;               /* 1: without explicit centres, mihevc_k_inter_frame derives them from the 1/4-size pictures */
    int32_t rdo_zero;                 /* 1: RD zero-out of inter TUs */
    int32_t chroma_modes;             /* 1: chroma intra mode decision (else DM) */
    int32_t mc_top, mc_bottom;        /* 1: motion compensation must not read above row 0 / below the last row (the picture is a slice whose
                                       * neighbour lives on another device); 0: the padded border is the picture's own */
    int32_t rdo_cg;                   /* k > 0: RD zero-out of 4x4 coefficient groups of inter TUs with lambda x k / 2; 0: off */
} mihevc_cost_params;
void mihevc_cost_params_for_qp(int qp, int bit_depth, int me_range, mihevc_cost_params *out);   /* tile grid 1x1, every analysis knob 0 */
/* Tile grid of IDR pictures for this configuration: the most columns/rows Table A.8 allows at cfg->level_idc with every
 * column >= 256 and every row >= 64 luma samples (A.4.1), uniform spacing; 1x1 when cfg->intra_tiles == 0. */
int  mihevc_tile_grid(const mihevc_config *cfg, int *cols, int *rows);
/* the same for P pictures (cfg->p_tiles): 1x1 when off */
int  mihevc_p_tile_grid(const mihevc_config *cfg, int *cols, int *rows);

/* per-8x8-block record produced by the analysis kernels and consumed by deblocking and the host entropy coder */
typedef struct mihevc_cu_rec {
    uint8_t log2_size;     /* CU size 3..5 (CU = PU = TU except intra NxN) */
    uint8_t flags;         /* bit0 inter, bit1 cbf_y, bit2 cbf_cb, bit3 cbf_cr, bit4 intra NxN; inter CUs of B pictures: bit5 list 1 is used (its vector: two
                            * little-endian int16 in intra_mode[0..3], which inter CUs do not use), bit6 list 0 is NOT used; neither: list 0 only (P pictures) */
    uint8_t chroma_mode;   /* chroma intra prediction mode 0..34 */
    uint8_t qp;
    uint8_t intra_mode[4];
    int16_t mvx, mvy;      /* quarter-sample units */
    uint8_t cbf_y4;
    uint8_t pad[3];
} mihevc_cu_rec;

typedef struct mihevc_sao_ctu {
    uint8_t type[2];       /* luma, chroma: 0 off, 1 band, 2 edge */
    uint8_t eo_class[2];
    uint8_t band_pos[3];
    int8_t  offset[3][4];
    uint8_t pad;
} mihevc_sao_ctu;

/* ---- per-stage entry points: host buffers in, host buffers out, one device pass each.  They run the SAME kernels
 *      the session runs, so the parity tests (tests/test_gpu_parity.py) hit each stage alone.  Sample planes are
 *      uint8_t when bit_depth == 8 and uint16_t otherwise; pitches are in samples. ---- */
/* K3: forward transform + quantisation + dequantisation + inverse transform of n_blocks residual blocks; log2n 2..5 (4/8/16/32-point
 * DCT); dst4 = 1 with log2n = 2 selects the 4x4 DST-VII of intra luma TUs (8.6.4.2) */
int mihevc_k_transform(int device, const int16_t *residual, int16_t *levels, int16_t *recon_residual,
                       int n_blocks, int log2n, int qp, int bit_depth, int intra, int dst4);
/* K2+K3: intra picture analysis -> pre-deblock reconstruction, CU records, levels */
int mihevc_k_intra_frame(int device, const void *src_y, const void *src_u, const void *src_v, int width, int height,
                         const mihevc_cost_params *prm, void *rec_y, void *rec_u, void *rec_v,
                         mihevc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                         uint64_t *est_bits_q4 /* optional: the picture's rate estimate in 1/16 bit (rate control input) */);
/* K1+K3: inter picture analysis against one (unpadded) reference reconstruction */
int mihevc_k_inter_frame(int device, const void *src_y, const void *src_u, const void *src_v,
                         const void *ref_y, const void *ref_u, const void *ref_v, int width, int height,
                         const mihevc_cost_params *prm, const int16_t *centers, void *rec_y, void *rec_u, void *rec_v,
                         mihevc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, int32_t *me_dump,
                         uint64_t *est_bits_q4);
/* K1+K3 for a B picture (cfg.bframes) between two anchors: ref0 = the (unpadded) reconstruction of the anchor before it in display order (list 0), ref1 = the
 * anchor after it (list 1); both integer searches, list-0 tree, list-1 refinement, bi-prediction trial.  centers0 / centers1, me_dump0 / me_dump1: per list */
int mihevc_k_b_frame(int device, const void *src_y, const void *src_u, const void *src_v,
                     const void *ref0_y, const void *ref0_u, const void *ref0_v, const void *ref1_y, const void *ref1_u, const void *ref1_v, int width, int height,
                     const mihevc_cost_params *prm, const int16_t *centers0, const int16_t *centers1, void *rec_y, void *rec_u, void *rec_v,
                     mihevc_cu_rec *cu, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, int32_t *me_dump0, int32_t *me_dump1, uint64_t *est_bits_q4);
/* K4a: deblocking in place (the picture passes: a session runs them only without SAO, see mihevc_k_loop_filter) */
int mihevc_k_deblock(int device, void *rec_y, void *rec_u, void *rec_v, int width, int height,
                     const mihevc_cu_rec *cu, int bit_depth);
/* K4b: SAO statistics + decision + apply */
int mihevc_k_sao(int device, const void *src_y, const void *src_u, const void *src_v,
                 const void *dbk_y, const void *dbk_u, const void *dbk_v, int width, int height,
                 const mihevc_cost_params *prm, void *out_y, void *out_u, void *out_v, mihevc_sao_ctu *sao);
/* K4 in one pass (what a session runs): the CTU programs deblock their own tile of the PRE-deblock reconstruction `rec_*` (halo of 4 luma / 2 chroma samples),
 * then decide and apply SAO.  Result = mihevc_k_deblock followed by mihevc_k_sao, bit for bit. */
int mihevc_k_loop_filter(int device, const void *src_y, const void *src_u, const void *src_v,
                         const void *rec_y, const void *rec_u, const void *rec_v, int width, int height, const mihevc_cu_rec *cu,
                         const mihevc_cost_params *prm, void *out_y, void *out_u, void *out_v, mihevc_sao_ctu *sao);

/* ---- host-only stages (no device needed): bitstream ---- */
/* VPS+SPS+PPS (+SEI when hdr10) as Annex-B into buf; returns size or negative error */
int mihevc_write_parameter_sets(const mihevc_config *cfg, uint8_t *buf, size_t cap);
/* CABAC-code one picture from its symbols into one slice NAL (+AUD when cfg->aud); returns size or negative error.
 * slice_type 2 = I (IDR), 1 = P, 0 = B (cfg->bframes); poc = position inside the closed GOP in display order. */
int mihevc_encode_picture_host(const mihevc_config *cfg, int slice_type, int poc, int qp,
                               const mihevc_cu_rec *cu, const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v,
                               const mihevc_sao_ctu *sao, uint8_t *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MIHEVC_H */
