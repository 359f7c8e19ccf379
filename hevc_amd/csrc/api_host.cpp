// hevc_amd/csrc/api_host.cpp — C-ABI entry points that need no device: defaults, cost parameters, bitstream.
#include <cmath>
#include <cstring>

#include "bitstream.h"

using namespace mihevc;

extern "C" {

int mihevc_abi_version(void) { return MIHEVC_ABI_VERSION; }

const char *mihevc_strerror(int err)
{
    switch (err) {
    case MIHEVC_OK: return "ok";
    case MIHEVC_EAGAIN: return "no packet ready";
    case MIHEVC_EOF: return "end of stream";
    case MIHEVC_EINVAL: return "invalid argument or unsupported configuration";
    case MIHEVC_ENODEV: return "no gfx950 device available";
    case MIHEVC_ENOMEM: return "out of memory";
    case MIHEVC_EDEVICE: return "HIP runtime error";
    case MIHEVC_ESTATE: return "call not valid in this session state";
    default: return "unknown error";
    }
}

// Defaults = the reference's 1080p30 SDR operating point (reference core/transcoder.py:398-411 evaluated for
// 1920x1080@30, 10 s: crf 19, vbv 2940/3528, keyint 90, min-keyint 45, level 4, main tier).
void mihevc_config_default(mihevc_config *c)
{
    memset(c, 0, sizeof *c);
    c->width = 1920; c->height = 1080; c->fps_num = 30; c->fps_den = 1; c->bit_depth = 8;
    c->level_idc = 120; c->tier = 0; c->crf = 19; c->qp = -1;
    c->vbv_maxrate_kbps = 2940; c->vbv_bufsize_kbits = 3528; c->keyint = 90; c->min_keyint = 45;
    c->colour_primaries = 1; c->transfer = 1; c->matrix = 1; c->full_range = 0; c->chroma_loc = -1;
    c->aud = 0; c->repeat_headers = 0; c->hdr10 = 0;
    // core/utils.py:38,40 defaults, used when hdr10 is switched on without explicit metadata
    const uint16_t prim[3][2] = {{13250, 34500}, {7500, 3000}, {34000, 16000}};
    memcpy(c->md_primaries, prim, sizeof prim);
    c->md_white[0] = 15635; c->md_white[1] = 16450; c->md_max_lum = 10000000; c->md_min_lum = 50;
    c->max_cll = 1000; c->max_fall = 400;
    c->me_range = 0; c->gops_in_flight = 0; c->host_threads = 0; c->sao = 1; c->intra_tiles = 1; c->intra_nxn = 0; c->intra_in_p = 0; c->pre_search = 1; c->rdo_zero = 1; c->chroma_modes = 1; c->scenecut = 1; c->gop_balance = 1; c->rdo_cg = 0;
    c->p_tiles = -1; c->b_qp_offset = -1;
}

// lambda_mode = 0.57 * 2^((qp-12)/3) (the usual HM/x265 relation); SAD/SATD-domain lambda is its square root.
void mihevc_cost_params_for_qp(int qp, int bit_depth, int me_range, mihevc_cost_params *out)
{
    static const int8_t tab[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
    double lam = 0.57 * std::pow(2.0, (qp - 12) / 3.0);
    out->qp = qp;
    int qpi = qp < -12 ? -12 : qp > 57 ? 57 : qp;
    out->qp_c = qpi < 30 ? qpi : qpi > 43 ? qpi - 6 : tab[qpi - 30];   // Table 8-10, cb/cr offsets 0
    out->bit_depth = bit_depth;
    // distortions are measured on the samples as they are, so at 10 bit an SSE is 16x and a SAD/SATD 4x its 8-bit value: scale the
    // multipliers instead of shifting every distortion (HM shifts the distortion; same trade-off).  Without this Main10 decisions ran
    // with a 16x too small lambda: SAO parameters alone cost 72 bits per CTU on the 2160p bench clip.
    const int sh = bit_depth - 8;
    out->lambda_sad_q4 = (int)std::lround(16.0 * std::sqrt(lam)) << sh;
    out->lambda_q4 = (int)std::lround(16.0 * lam) << (2 * sh);
    out->me_range = me_range > 0 ? me_range : 15;
    out->tile_cols = out->tile_rows = 1;
    out->intra_nxn = 0;
    out->intra_in_p = 0;
    out->pre_search = 0;
    out->rdo_zero = 0;
    out->chroma_modes = 0;
    out->mc_top = out->mc_bottom = 0;
    out->rdo_cg = 0;
}

static int copy_out(const std::vector<uint8_t> &v, uint8_t *buf, size_t cap)
{
    if (v.size() > cap) return MIHEVC_ENOMEM;
    memcpy(buf, v.data(), v.size());
    return (int)v.size();
}

static bool config_ok(const mihevc_config *c)
{
    return c && c->width >= 16 && c->height >= 16 && c->width <= 8192 && c->height <= 4352 && !(c->width & 1) && !(c->height & 1) &&
           (c->bit_depth == 8 || c->bit_depth == 10) && c->fps_num > 0 && c->fps_den > 0;
}

int mihevc_tile_grid(const mihevc_config *cfg, int *cols, int *rows)
{
    if (!config_ok(cfg) || !cols || !rows) return MIHEVC_EINVAL;
    TileGrid g = tile_grid(*cfg);
    *cols = g.cols; *rows = g.rows;
    return MIHEVC_OK;
}

int mihevc_p_tile_grid(const mihevc_config *cfg, int *cols, int *rows)
{
    if (!config_ok(cfg) || !cols || !rows) return MIHEVC_EINVAL;
    TileGrid g = p_tile_grid(*cfg);
    *cols = g.cols; *rows = g.rows;
    return MIHEVC_OK;
}

int mihevc_write_parameter_sets(const mihevc_config *cfg, uint8_t *buf, size_t cap)
{
    if (!config_ok(cfg) || !buf) return MIHEVC_EINVAL;
    std::vector<uint8_t> v;
    write_parameter_sets(*cfg, v);
    return copy_out(v, buf, cap);
}

int mihevc_encode_picture_host(const mihevc_config *cfg, int slice_type, int poc, int qp, const mihevc_cu_rec *cu,
                               const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v,
                               const mihevc_sao_ctu *sao, uint8_t *buf, size_t cap)
{
    if (!config_ok(cfg) || !cu || !coef_y || !coef_u || !coef_v || !buf) return MIHEVC_EINVAL;
    if (slice_type != 1 && slice_type != 2 && !(slice_type == 0 && cfg->bframes > 0)) return MIHEVC_EINVAL;
    PictureSyms p{slice_type, poc, qp, cu, {coef_y, coef_u, coef_v}, cfg->sao ? sao : nullptr, 0};
    std::vector<uint8_t> v;
    encode_picture(*cfg, p, v);
    return copy_out(v, buf, cap);
}

}  // extern "C"
