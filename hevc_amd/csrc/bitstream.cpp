// hevc_amd/csrc/bitstream.cpp — parameter sets, slice header and CABAC slice data on host cores.
// See bitstream.h for scope.  Clause numbers: ITU-T H.265.
#include "bitstream.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mihevc {

// ------------------------------------------------------------------------------------------ bit writer
void BitWriter::put(uint32_t v, int n)
{
    while (n > 0) {
        int take = std::min(n, 8 - nbits_);
        uint32_t chunk = (n == 32 && take == 32) ? v : ((v >> (n - take)) & ((1u << take) - 1));
        acc_ = (acc_ << take) | chunk;
        nbits_ += take;
        n -= take;
        if (nbits_ == 8) {
            buf_.push_back((uint8_t)acc_);
            acc_ = 0;
            nbits_ = 0;
        }
    }
}
void BitWriter::ue(uint32_t v)
{
    uint32_t x = v + 1;
    int len = 0;
    while ((x >> len) > 1) len++;
    put(0, len);
    put(x, len + 1);
}
void BitWriter::se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
void BitWriter::align_zero()
{
    if (nbits_) put(0, 8 - nbits_);
}
void BitWriter::trailing()
{
    put(1, 1);
    align_zero();
}

void append_nal(std::vector<uint8_t> &out, int nal_type, const std::vector<uint8_t> &rbsp)
{
    static const uint8_t sc[4] = {0, 0, 0, 1};
    out.insert(out.end(), sc, sc + 4);
    out.push_back((uint8_t)(nal_type << 1));   // forbidden_zero_bit, type, nuh_layer_id high bit (0)
    out.push_back(1);                          // nuh_layer_id low bits (0), nuh_temporal_id_plus1 = 1
    int zeros = 0;
    out.reserve(out.size() + rbsp.size() + rbsp.size() / 64 + 8);
    for (uint8_t b : rbsp) {
        if (zeros >= 2 && b <= 3) {
            out.push_back(3);                  // emulation_prevention_three_byte
            zeros = 0;
        }
        out.push_back(b);
        zeros = b == 0 ? zeros + 1 : 0;
    }
    // cabac_zero_words / trailing zero protection: an RBSP ending in 0x00 would need a 0x03; ours always end in the stop bit
}

// ------------------------------------------------------------------------------------------ parameter sets
static void profile_tier_level(BitWriter &w, const mihevc_config &c)
{
    int profile = c.bit_depth > 8 ? 2 : 1;
    w.put(0, 2);                    // general_profile_space
    w.put1(c.tier);                 // general_tier_flag
    w.put((uint32_t)profile, 5);
    w.put(profile == 1 ? 0x60000000u : 0x20000000u, 32);   // compatibility: Main streams also decode as Main10
    w.put1(1);                      // progressive_source
    w.put1(0);                      // interlaced_source
    w.put1(0);                      // non_packed_constraint
    w.put1(1);                      // frame_only_constraint
    w.put(0, 32);                   // 43 reserved zero bits + 1
    w.put(0, 12);
    w.put((uint32_t)c.level_idc, 8);
}

void write_vps(const mihevc_config &c, std::vector<uint8_t> &out)
{
    BitWriter w;
    w.put(0, 4);          // vps_video_parameter_set_id
    w.put1(1);            // vps_base_layer_internal_flag
    w.put1(1);            // vps_base_layer_available_flag
    w.put(0, 6);          // vps_max_layers_minus1
    w.put(0, 3);          // vps_max_sub_layers_minus1
    w.put1(1);            // vps_temporal_id_nesting_flag
    w.put(0xffff, 16);
    profile_tier_level(w, c);
    w.put1(1);            // vps_sub_layer_ordering_info_present_flag
    w.ue(c.bframes != 0 ? 2 : 1);   // vps_max_dec_pic_buffering_minus1: current + one reference (+ the second anchor a B picture predicts from)
    w.ue(c.bframes != 0 ? 1 : 0);   // vps_max_num_reorder_pics: an anchor precedes the B picture in front of it in decoding order
    w.ue(0);              // vps_max_latency_increase_plus1
    w.put(0, 6);          // vps_max_layer_id
    w.ue(0);              // vps_num_layer_sets_minus1
    w.put1(1);            // vps_timing_info_present_flag
    w.put((uint32_t)c.fps_den, 32);
    w.put((uint32_t)c.fps_num, 32);
    w.put1(0);            // vps_poc_proportional_to_timing_flag
    w.ue(0);              // vps_num_hrd_parameters
    w.put1(0);            // vps_extension_flag
    w.trailing();
    append_nal(out, 32, w.bytes());
}

void write_sps(const mihevc_config &c, std::vector<uint8_t> &out)
{
    CodedSize cs = coded_size(c.width, picture_height(c));
    BitWriter w;
    w.put(0, 4);          // sps_video_parameter_set_id
    w.put(0, 3);          // sps_max_sub_layers_minus1
    w.put1(1);            // sps_temporal_id_nesting_flag
    profile_tier_level(w, c);
    w.ue(0);              // sps_seq_parameter_set_id
    w.ue(1);              // chroma_format_idc 4:2:0
    w.ue((uint32_t)cs.w);
    w.ue((uint32_t)cs.h);
    bool crop = cs.crop_r || cs.crop_b;
    w.put1(crop);
    if (crop) {
        w.ue(0);
        w.ue((uint32_t)cs.crop_r / 2);    // units of SubWidthC
        w.ue(0);
        w.ue((uint32_t)cs.crop_b / 2);
    }
    w.ue((uint32_t)c.bit_depth - 8);
    w.ue((uint32_t)c.bit_depth - 8);
    w.ue(4);              // log2_max_pic_order_cnt_lsb_minus4 -> 8 bits (keyint <= 240)
    w.put1(1);            // sps_sub_layer_ordering_info_present_flag
    w.ue(c.bframes != 0 ? 2 : 1);   // sps_max_dec_pic_buffering_minus1
    w.ue(c.bframes != 0 ? 1 : 0);   // sps_max_num_reorder_pics
    w.ue(0);                       // sps_max_latency_increase_plus1
    w.ue(0);              // log2_min_luma_coding_block_size_minus3 -> 8
    w.ue(kCtuLog2 - 3);   // log2_diff_max_min_luma_coding_block_size -> CTB 32
    w.ue(0);              // log2_min_luma_transform_block_size_minus2 -> 4
    w.ue(3);              // log2_diff_max_min_luma_transform_block_size -> 32
    w.ue(0);              // max_transform_hierarchy_depth_inter
    w.ue(0);              // max_transform_hierarchy_depth_intra (NxN still splits once, 7.3.8.8)
    w.put1(0);            // scaling_list_enabled_flag
    w.put1(0);            // amp_enabled_flag
    w.put1(c.sao != 0);   // sample_adaptive_offset_enabled_flag
    w.put1(0);            // pcm_enabled_flag
    if (c.bframes != 0) {          // (bframes = -1: B pictures where the session's probe finds them worth it: the stream has to announce them)
        // three reference picture sets (7.3.7): 0 = {-1} (an anchor right behind its predecessor: the last picture of an even GOP), 1 = {-2} (an anchor
        // two pictures on), 2 = {-1, +1} (the B picture between two anchors)
        w.ue(3);          // num_short_term_ref_pic_sets
        w.ue(1); w.ue(0); w.ue(0); w.put1(1);                                   // set 0: one negative picture, delta -1, used
        w.put1(0); w.ue(1); w.ue(0); w.ue(1); w.put1(1);                        // set 1: inter_ref_pic_set_prediction_flag 0; one negative picture, delta -2, used
        w.put1(0); w.ue(1); w.ue(1); w.ue(0); w.put1(1); w.ue(0); w.put1(1);    // set 2: one negative (-1, used), one positive (+1, used)
    } else {
    w.ue(1);              // num_short_term_ref_pic_sets
    w.ue(1);              //   num_negative_pics
    w.ue(0);              //   num_positive_pics
    w.ue(0);              //   delta_poc_s0_minus1
    w.put1(1);            //   used_by_curr_pic_s0_flag
    }
    w.put1(0);            // long_term_ref_pics_present_flag
    w.put1(0);            // sps_temporal_mvp_enabled_flag
    w.put1(1);            // strong_intra_smoothing_enabled_flag
    w.put1(1);            // vui_parameters_present_flag
    {   // E.2.1
        w.put1(1);        // aspect_ratio_info_present_flag
        w.put(1, 8);      // square samples
        w.put1(0);        // overscan_info_present_flag
        w.put1(1);        // video_signal_type_present_flag
        w.put(5, 3);      // video_format unspecified
        w.put1(c.full_range);
        w.put1(1);        // colour_description_present_flag
        w.put((uint32_t)c.colour_primaries, 8);
        w.put((uint32_t)c.transfer, 8);
        w.put((uint32_t)c.matrix, 8);
        w.put1(c.chroma_loc >= 0);
        if (c.chroma_loc >= 0) {
            w.ue((uint32_t)c.chroma_loc);
            w.ue((uint32_t)c.chroma_loc);
        }
        w.put1(0);        // neutral_chroma_indication_flag
        w.put1(0);        // field_seq_flag
        w.put1(0);        // frame_field_info_present_flag
        w.put1(0);        // default_display_window_flag
        w.put1(1);        // vui_timing_info_present_flag
        w.put((uint32_t)c.fps_den, 32);
        w.put((uint32_t)c.fps_num, 32);
        w.put1(0);        // vui_poc_proportional_to_timing_flag
        const HrdInfo hi = hrd_info(c);
        w.put1(hi.on);    // vui_hrd_parameters_present_flag
        if (hi.on) {      // E.2.2 hrd_parameters(1, 0)
            w.put1(1);            // nal_hrd_parameters_present_flag
            w.put1(0);            // vcl_hrd_parameters_present_flag
            w.put1(0);            // sub_pic_hrd_params_present_flag
            w.put(0, 4);          // bit_rate_scale: 64 bit/s units
            w.put(0, 4);          // cpb_size_scale: 16 bit units
            w.put(23, 5);         // initial_cpb_removal_delay_length_minus1
            w.put(23, 5);         // au_cpb_removal_delay_length_minus1
            w.put(4, 5);          // dpb_output_delay_length_minus1
            w.put1(1);            // fixed_pic_rate_general_flag (within_cvs inferred 1)
            w.ue(0);              // elemental_duration_in_tc_minus1
            w.ue(0);              // cpb_cnt_minus1 (low_delay_hrd_flag inferred 0)
            w.ue(hi.bit_rate_value_minus1);
            w.ue(hi.cpb_size_value_minus1);
            w.put1(0);            // cbr_flag
        }
        w.put1(0);        // bitstream_restriction_flag
    }
    w.put1(0);            // sps_extension_present_flag
    w.trailing();
    append_nal(out, 33, w.bytes());
}

TileGrid tile_grid(const mihevc_config &cfg);
// Table A.8 (MaxTileCols / MaxTileRows) by general_level_idc
static void level_tile_limits(int level_idc, int &max_cols, int &max_rows)
{
    max_cols = max_rows = 1;
    if (level_idc >= 180) { max_cols = 20; max_rows = 22; }
    else if (level_idc >= 150) { max_cols = 10; max_rows = 11; }
    else if (level_idc >= 120) { max_cols = 5; max_rows = 5; }
    else if (level_idc >= 93) { max_cols = 3; max_rows = 3; }
    else if (level_idc >= 90) { max_cols = 2; max_rows = 2; }
}
int slice_first_row(const mihevc_config &cfg, int k)
{
    int r = 0;
    for (int j = 0; j < k && j < 16; j++) r += cfg.slice_ctu_rows[j];
    return r;
}
// sliced pictures: tiles at all?  Only when a column can be 256 luma samples wide (A.4.1) and the level's tile rows (Table A.8) reach
// around the slices — every slice is at least one tile row then
static bool sliced_tiles_allowed(const mihevc_config &cfg)
{
    int max_cols, max_rows;
    level_tile_limits(cfg.level_idc, max_cols, max_rows);
    return cfg.intra_tiles && ((cfg.width + 7) & ~7) >= 256 && max_rows >= cfg.slice_count;
}
// tile rows of slice k: the level's row budget shared evenly by the slices, every row >= 64 luma samples (A.4.1)
int slice_tile_rows(const mihevc_config &cfg, int k)
{
    int max_cols, max_rows;
    level_tile_limits(cfg.level_idc, max_cols, max_rows);
    if (!sliced_tiles_allowed(cfg)) return 1;
    return std::max(1, std::min(max_rows / std::max(1, cfg.slice_count), cfg.slice_ctu_rows[k] / (64 / kCtu)));
}

// does PPS 1 (IDR pictures) enable tiles for the PICTURE?  (a slice of it may still be a single tile)
bool idr_tiles_on(const mihevc_config &cfg)
{
    TileGrid g = tile_grid(cfg);
    if (!sliced(cfg)) return g.on();
    if (!sliced_tiles_allowed(cfg)) return false;
    int rows = 0;
    for (int k = 0; k < cfg.slice_count; k++) rows += slice_tile_rows(cfg, k);
    return g.cols > 1 || rows > 1;
}

TileGrid tile_grid(const mihevc_config &cfg)
{
    TileGrid g;
    CodedSize cs = coded_size(cfg.width, cfg.height);
    g.wc = (cs.w + kCtu - 1) >> kCtuLog2; g.hc = (cs.h + kCtu - 1) >> kCtuLog2;
    if (!cfg.intra_tiles || cs.w < 256 || cs.h < 64) return g;     // A.4.1 bounds every tile, so small pictures stay untiled
    if (sliced(cfg) && !sliced_tiles_allowed(cfg)) return g;
    int max_cols, max_rows;
    level_tile_limits(cfg.level_idc, max_cols, max_rows);
    // A.4.1: columns >= 256 and rows >= 64 luma samples; with uniform spacing the narrowest column is floor(wc / cols) CTBs
    g.cols = std::max(1, std::min(max_cols, g.wc / (256 / kCtu)));
    g.rows = sliced(cfg) ? slice_tile_rows(cfg, cfg.slice_index) : std::max(1, std::min(max_rows, g.hc / (64 / kCtu)));
    return g;
}

// P pictures (PPS 0): a coarse uniform grid, one tile per 1920x1080 of picture — large tiles cost next to nothing in bits (contexts restart and
// merge / AMVP candidates end at 16 boundaries of a 4320p picture instead of 440) and give the host one CABAC job per tile
TileGrid p_tile_grid(const mihevc_config &cfg)
{
    TileGrid g;
    CodedSize cs = coded_size(cfg.width, cfg.height);
    g.wc = (cs.w + kCtu - 1) >> kCtuLog2; g.hc = (cs.h + kCtu - 1) >> kCtuLog2;
    if (cfg.p_tiles == 0 || sliced(cfg)) return g;        // the slices of a sliced picture already are one job each
    if (cs.w < 256 || cs.h < 64) return g;                // A.4.1 bounds EVERY tile once tiles are on — a single column too: small pictures stay untiled
    int max_cols, max_rows;
    level_tile_limits(cfg.level_idc, max_cols, max_rows);
    const int floor_ = cfg.p_tiles > 0 ? 2 : 1;
    g.cols = std::max(1, std::min({max_cols, g.wc / (256 / kCtu), std::max(floor_, cs.w / 1920)}));
    g.rows = std::max(1, std::min({max_rows, g.hc / (64 / kCtu), std::max(floor_, cs.h / 1080)}));
    return g;
}

void write_pps(const mihevc_config &cfg, int pps_id, std::vector<uint8_t> &out)
{
    BitWriter w;
    TileGrid g = pps_id == 1 ? tile_grid(cfg) : p_tile_grid(cfg);
    int rows_total = g.rows;
    if (pps_id == 1 && sliced(cfg)) {          // every slice splits ITS rows uniformly: the picture's grid is the concatenation, spelled out
        rows_total = 0;
        for (int k = 0; k < cfg.slice_count; k++) rows_total += slice_tile_rows(cfg, k);
    }
    const bool tiles_on = g.cols > 1 || rows_total > 1;
    w.ue((uint32_t)pps_id); // pps_pic_parameter_set_id
    w.ue(0);              // pps_seq_parameter_set_id
    w.put1(0);            // dependent_slice_segments_enabled_flag
    w.put1(0);            // output_flag_present_flag
    w.put(0, 3);          // num_extra_slice_header_bits
    w.put1(0);            // sign_data_hiding_enabled_flag
    w.put1(0);            // cabac_init_present_flag
    w.ue(0);              // num_ref_idx_l0_default_active_minus1
    w.ue(0);              // num_ref_idx_l1_default_active_minus1
    w.se(0);              // init_qp_minus26
    w.put1(0);            // constrained_intra_pred_flag
    w.put1(0);            // transform_skip_enabled_flag
    w.put1(0);            // cu_qp_delta_enabled_flag
    w.se(0);              // pps_cb_qp_offset
    w.se(0);              // pps_cr_qp_offset
    w.put1(0);            // pps_slice_chroma_qp_offsets_present_flag
    w.put1(0);            // weighted_pred_flag
    w.put1(0);            // weighted_bipred_flag
    w.put1(0);            // transquant_bypass_enabled_flag
    w.put1(tiles_on);     // tiles_enabled_flag
    w.put1(0);            // entropy_coding_sync_enabled_flag
    if (tiles_on) {
        w.ue((uint32_t)g.cols - 1);       // num_tile_columns_minus1
        w.ue((uint32_t)rows_total - 1);   // num_tile_rows_minus1
        const bool uniform = !(pps_id == 1 && sliced(cfg));
        w.put1(uniform);                  // uniform_spacing_flag
        if (!uniform) {
            for (int i = 0; i + 1 < g.cols; i++) w.ue((uint32_t)(g.col_bd(i + 1) - g.col_bd(i) - 1));          // column_width_minus1
            int left = rows_total - 1;
            for (int k = 0; k < cfg.slice_count && left > 0; k++) {
                const int r = slice_tile_rows(cfg, k), n = cfg.slice_ctu_rows[k];
                for (int j = 0; j < r && left > 0; j++, left--) w.ue((uint32_t)((j + 1) * n / r - j * n / r - 1));   // row_height_minus1
            }
        }
        w.put1(1);                        // loop_filter_across_tiles_enabled_flag
    }
    w.put1(!sliced(cfg) || cfg.slice_halo != 0); // pps_loop_filter_across_slices_enabled_flag: the slices of a picture live on different devices; they filter across
                          // their seams only when they exchange rows (cfg.slice_halo)
    w.put1(0);            // deblocking_filter_control_present_flag
    w.put1(0);            // pps_scaling_list_data_present_flag
    w.put1(0);            // lists_modification_present_flag
    w.ue(0);              // log2_parallel_merge_level_minus2
    w.put1(0);            // slice_segment_header_extension_present_flag
    w.put1(0);            // pps_extension_present_flag
    w.trailing();
    append_nal(out, 34, w.bytes());
}

void write_sei_hdr10(const mihevc_config &c, std::vector<uint8_t> &out)
{
    {   // D.2.28 mastering display colour volume
        BitWriter w;
        w.put(137, 8);
        w.put(24, 8);
        for (int i = 0; i < 3; i++) {
            w.put(c.md_primaries[i][0], 16);
            w.put(c.md_primaries[i][1], 16);
        }
        w.put(c.md_white[0], 16);
        w.put(c.md_white[1], 16);
        w.put(c.md_max_lum, 32);
        w.put(c.md_min_lum, 32);
        w.trailing();
        append_nal(out, 39, w.bytes());
    }
    {   // D.2.35 content light level
        BitWriter w;
        w.put(144, 8);
        w.put(4, 8);
        w.put(c.max_cll, 16);
        w.put(c.max_fall, 16);
        w.trailing();
        append_nal(out, 39, w.bytes());
    }
}

HrdInfo hrd_info(const mihevc_config &c)
{
    HrdInfo h{};
    h.on = c.hrd && c.vbv_maxrate_kbps > 0 && c.vbv_bufsize_kbits > 0;
    if (!h.on) return h;
    const uint64_t rate = (uint64_t)c.vbv_maxrate_kbps * 1000, cpb = (uint64_t)c.vbv_bufsize_kbits * 1000;
    h.bit_rate_value_minus1 = (uint32_t)(rate / 64 - 1);
    h.cpb_size_value_minus1 = (uint32_t)(cpb / 16 - 1);
    const uint64_t full = 90000ull * (((uint64_t)h.cpb_size_value_minus1 + 1) * 16) / (((uint64_t)h.bit_rate_value_minus1 + 1) * 64);   // CPB in 90 kHz ticks
    h.initial_delay = (uint32_t)(full * 9 / 10);
    h.initial_offset = (uint32_t)(full - h.initial_delay);
    return h;
}

static void append_sei(int payload_type, BitWriter &body, std::vector<uint8_t> &out)
{
    // sei_payload: data, then payload_bit_equal_to_one + zero bits when the data is not byte aligned (D.2.1)
    if (!body.aligned()) { body.put1(1); body.align_zero(); }
    BitWriter w;
    w.put((uint32_t)payload_type, 8);
    w.put((uint32_t)body.bytes().size(), 8);
    w.append_bytes(body.bytes().data(), body.bytes().size());
    w.trailing();
    append_nal(out, 39, w.bytes());
}

void write_sei_buffering_period(const mihevc_config &c, std::vector<uint8_t> &out)
{
    const HrdInfo hi = hrd_info(c);
    if (!hi.on) return;
    BitWriter b;
    b.ue(0);                      // bp_seq_parameter_set_id
    b.put1(0);                    // irap_cpb_params_present_flag
    b.put1(0);                    // concatenation_flag
    b.put(0, 24);                 // au_cpb_removal_delay_delta_minus1
    b.put(hi.initial_delay, 24);  // nal_initial_cpb_removal_delay[0]
    b.put(hi.initial_offset, 24); // nal_initial_cpb_removal_offset[0]
    append_sei(0, b, out);
}

void write_sei_pic_timing(const mihevc_config &c, uint32_t au_cpb_removal_delay_minus1, std::vector<uint8_t> &out, uint32_t pic_dpb_output_delay)
{
    if (!hrd_info(c).on) return;
    BitWriter b;                  // frame_field_info_present_flag = 0: only the CPB / DPB delays
    b.put(au_cpb_removal_delay_minus1 & 0xffffffu, 24);
    b.put(pic_dpb_output_delay & 31u, 5);      // clock ticks between removal from the CPB and output: 0 without reordering
    append_sei(1, b, out);
}

void write_aud(int slice_type, std::vector<uint8_t> &out)
{
    BitWriter w;
    w.put(slice_type == 2 ? 0u : slice_type == 1 ? 1u : 2u, 3);   // pic_type: 0 = I, 1 = P,I, 2 = B,P,I
    w.trailing();
    append_nal(out, 35, w.bytes());
}

void write_parameter_sets(const mihevc_config &c, std::vector<uint8_t> &out)
{
    write_vps(c, out);
    write_sps(c, out);
    write_pps(c, 0, out);
    if (idr_tiles_on(c) || p_tile_grid(c).on()) write_pps(c, 1, out);      // IDR pictures never use PPS 0 when it carries the P pictures' tile grid
    if (c.hdr10) write_sei_hdr10(c, out);
}

// ------------------------------------------------------------------------------------------ CABAC encoder (9.3.4.x encoder side)
namespace {

const uint8_t kRangeLps[64][4] = {
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
const uint8_t kNextLps[64] = {0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                              24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

// context indices (one flat table) — Tables 9-5 .. 9-37
enum Ctx {
    kSaoMerge = 0, kSaoType = 1, kSplitCu = 2, kSkip = 5, kPredMode = 8, kPartMode = 9, kPrevIntra = 13, kChromaMode = 14,
    kRqtRoot = 15, kMergeFlag = 16, kMergeIdx = 17, kMvp = 18, kSplitTu = 19, kCbfLuma = 22, kCbfChroma = 24, kMvdG0 = 29, kMvdG1 = 30,
    kLastX = 31, kLastY = 49, kCsbf = 67, kSig = 71, kG1 = 115, kG2 = 139, kInterDir = 145, kCtxCount = 150
};
constexpr uint8_t U = 154;   // unused in this slice type
// initValue per context for initType 0 (I slices), 1 (P slices) and 2 (B slices), cabac_init_flag = 0 (Tables 9-5 .. 9-37)
const uint8_t kInitI[kCtxCount] = {
    153, 200, 139, 141, 157, U, U, U, U, 184, U, U, U, 184, 63, U, U, U, U, 153, 138, 138, 111, 141, 94, 138, 182, 154, 154, U, U,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
    91, 171, 134, 141,
    111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
    140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 141, 111,
    140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
    138, 153, 136, 167, 152, 152,
    U, U, U, U, U};
const uint8_t kInitP[kCtxCount] = {
    153, 185, 107, 139, 126, 197, 185, 201, 149, 154, 139, 154, 154, 154, 152, 79, 110, 122, 168, 124, 138, 94, 153, 111, 149, 107, 167, 154, 154, 140, 198,
    125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
    125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
    121, 140, 61, 154,
    155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
    170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140, 140, 140,
    154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
    107, 167, 91, 122, 107, 167,
    95, 79, 63, 31, 31};
const uint8_t kInitB[kCtxCount] = {
    153, 160, 107, 139, 126, 197, 185, 201, 134, 154, 139, 154, 154, 183, 152, 79, 154, 137, 168, 224, 167, 122, 153, 111, 149, 92, 167, 154, 154, 169, 198,
    125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
    125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
    121, 140, 61, 154,
    170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
    170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140, 140, 140,
    154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182,
    107, 167, 91, 107, 107, 167,
    95, 79, 63, 31, 31};

class Cabac {
public:
    explicit Cabac(std::vector<uint8_t> &out) : out_p_(&out) {}
    void redirect(std::vector<uint8_t> &out) { out_p_ = &out; }
    void init(int slice_type, int qp)          // slice_type 2 = I, 1 = P, 0 = B
    {
        const uint8_t *iv = slice_type == 2 ? kInitI : slice_type == 1 ? kInitP : kInitB;
        qp = std::min(51, std::max(0, qp));
        for (int i = 0; i < kCtxCount; i++) {
            int m = (iv[i] >> 4) * 5 - 45, n = ((iv[i] & 15) << 3) - 16;
            int pre = std::min(126, std::max(1, ((m * qp) >> 4) + n));
            int mps = pre > 63;
            state_[i] = (uint8_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps);
        }
        low_ = 0; range_ = 510; bits_left_ = 23; buffered_ = 0xff; n_buffered_ = 0; bins_ = 0;
    }
    inline void bin(int ctx, int b)
    {
        bins_++;
        uint8_t &st = state_[ctx];
        uint32_t s = st >> 1, mps = st & 1;
        uint32_t lps = kRangeLps[s][(range_ >> 6) & 3];
        range_ -= lps;
        if ((uint32_t)b != mps) {
            int nb = __builtin_clz(lps) - 23;
            low_ = (low_ + range_) << nb;
            range_ = lps << nb;
            if (s == 0) mps ^= 1;
            st = (uint8_t)((kNextLps[s] << 1) | mps);
            bits_left_ -= nb;
        } else {
            st = (uint8_t)(((s < 62 ? s + 1 : s) << 1) | mps);
            if (range_ >= 256) return;
            low_ <<= 1; range_ <<= 1; bits_left_--;
        }
        if (bits_left_ < 12) write_out();
    }
    inline void bypass(int b)
    {
        bins_++;
        low_ <<= 1;
        if (b) low_ += range_;
        if (--bits_left_ < 12) write_out();
    }
    // up to 8 bypass bins per renormalisation (9.3.4.4 unrolled: low = (low << n) + range * value)
    void bypass_bits(uint32_t v, int n)
    {
        bins_ += (size_t)n;
        while (n > 0) {
            int k = n > 8 ? 8 : n;
            uint32_t pat = (v >> (n - k)) & ((1u << k) - 1);
            low_ = (low_ << k) + range_ * pat;
            bits_left_ -= k;
            if (bits_left_ < 12) write_out();
            n -= k;
        }
    }
    void terminate(int b)
    {
        bins_++;
        range_ -= 2;
        if (b) {
            low_ += range_;
            low_ <<= 7; range_ = 2 << 7; bits_left_ -= 7;
        } else if (range_ >= 256) return;
        else { low_ <<= 1; range_ <<= 1; bits_left_--; }
        if (bits_left_ < 12) write_out();
    }
    // 9.3.4.5 flush after end_of_slice_segment_flag = 1; the final written bit doubles as rbsp_stop_one_bit
    void finish()
    {
        if (low_ >> (32 - bits_left_)) {
            out_p_->push_back((uint8_t)(buffered_ + 1));
            while (n_buffered_ > 1) { out_p_->push_back(0x00); n_buffered_--; }
            low_ -= 1u << (32 - bits_left_);
        } else {
            if (n_buffered_ > 0) out_p_->push_back((uint8_t)buffered_);
            while (n_buffered_ > 1) { out_p_->push_back(0xff); n_buffered_--; }
        }
        // remaining 24 - bits_left_ bits of low_, then the stop bit + alignment
        int n = 24 - bits_left_;
        uint32_t v = low_ >> 8;
        uint32_t acc = 0; int na = 0;
        for (int i = n - 1; i >= 0; i--) {
            acc = (acc << 1) | ((v >> i) & 1);
            if (++na == 8) { out_p_->push_back((uint8_t)acc); acc = 0; na = 0; }
        }
        acc = (acc << 1) | 1; na++;            // rbsp_stop_one_bit
        acc <<= (8 - na);
        out_p_->push_back((uint8_t)acc);
    }
    size_t bins() const { return bins_; }

private:
    void write_out()
    {
        uint32_t lead = low_ >> (24 - bits_left_);
        bits_left_ += 8;
        low_ &= 0xffffffffu >> bits_left_;
        if (lead == 0xff) { n_buffered_++; return; }
        if (n_buffered_ > 0) {
            uint32_t carry = lead >> 8;
            out_p_->push_back((uint8_t)(buffered_ + carry));
            buffered_ = lead & 0xff;
            uint8_t fill = (uint8_t)(0xff + carry);
            while (n_buffered_ > 1) { out_p_->push_back(fill); n_buffered_--; }
        } else {
            n_buffered_ = 1;
            buffered_ = lead;
        }
    }
    std::vector<uint8_t> *out_p_;
    uint8_t state_[kCtxCount];
    uint32_t low_ = 0, range_ = 510, buffered_ = 0xff;
    int bits_left_ = 23, n_buffered_ = 0;
    size_t bins_ = 0;
};

// scan tables (6.5.3 - 6.5.5): [scanIdx][log2 size 0..3][pos] -> (x,y)
struct Scans {
    uint8_t xy[3][4][64][2];
    Scans()
    {
        for (int l = 0; l < 4; l++) {
            int n = 1 << l, i = 0, x = 0, y = 0;
            for (;;) {
                while (y >= 0) {
                    if (x < n && y < n) { xy[0][l][i][0] = (uint8_t)x; xy[0][l][i][1] = (uint8_t)y; i++; }
                    y--; x++;
                }
                y = x; x = 0;
                if (i >= n * n) break;
            }
            i = 0;
            for (y = 0; y < n; y++) for (x = 0; x < n; x++) { xy[1][l][i][0] = (uint8_t)x; xy[1][l][i][1] = (uint8_t)y; i++; }
            i = 0;
            for (x = 0; x < n; x++) for (y = 0; y < n; y++) { xy[2][l][i][0] = (uint8_t)x; xy[2][l][i][1] = (uint8_t)y; i++; }
        }
    }
};
const Scans kScans;

// motion of a prediction block: which lists it uses and their vectors (every list holds one picture: reference indices are always 0)
struct Mot {
    int ok, f0, f1, x0, y0, x1, y1;
    bool same(const Mot &o) const { return f0 == o.f0 && f1 == o.f1 && (!f0 || (x0 == o.x0 && y0 == o.y0)) && (!f1 || (x1 == o.x1 && y1 == o.y1)); }
};
struct Mv { int ok, x, y; };

class SliceCoder {
public:
    SliceCoder(const mihevc_config &cfg, const PictureSyms &pic) : cfg_(cfg), pic_(pic), cabac_(scratch_)
    {
        CodedSize cs = coded_size(cfg.width, cfg.height);
        w_ = cs.w; h_ = cs.h; w8_ = w_ >> 3;
        wc_ = (w_ + kCtu - 1) >> kCtuLog2; hc_ = (h_ + kCtu - 1) >> kCtuLog2;
        skip_.assign((size_t)w8_ * (h_ >> 3), 0);
        depth_.assign((size_t)w8_ * (h_ >> 3), 0);
        grid_ = pic.slice_type == 2 ? tile_grid(cfg) : p_tile_grid(cfg);
        grid_.wc = wc_; grid_.hc = hc_;
    }
    const TileGrid &grid() const { return grid_; }
    // 7.3.8.1 slice_segment_data for tile (tx, ty) into `out`: one CABAC substream.  Every tile but the last ends with
    // end_of_subset_one_bit + byte_alignment(), the last with end_of_slice_segment_flag = 1 + trailing bits; both are the
    // same flush (the final written '1' is the alignment / stop bit).
    size_t run_tile(int tx, int ty, std::vector<uint8_t> &out)
    {
        cabac_.redirect(out);
        cabac_.init(pic_.slice_type, pic_.qp);
        const bool last_tile = tx == grid_.cols - 1 && ty == grid_.rows - 1;
        const int x1 = grid_.col_bd(tx + 1), y1 = grid_.row_bd(ty + 1);
        for (int ry = grid_.row_bd(ty); ry < y1; ry++)
            for (int rx = grid_.col_bd(tx); rx < x1; rx++) {
                if (pic_.sao) sao(rx, ry);
                quadtree(rx << kCtuLog2, ry << kCtuLog2, kCtuLog2, 0);
                cabac_.terminate(last_tile && ry == y1 - 1 && rx == x1 - 1);     // end_of_slice_segment_flag
            }
        if (!last_tile) cabac_.terminate(1);                                      // end_of_subset_one_bit
        cabac_.finish();
        return cabac_.bins();
    }

private:
    const mihevc_cu_rec &cu(int x, int y) const { return pic_.cu[(y >> 3) * w8_ + (x >> 3)]; }
    static int zorder6(int bx, int by)
    {
        int z = 0;
        for (int i = 0; i < 3; i++) z |= ((bx >> i) & 1) << (2 * i) | ((by >> i) & 1) << (2 * i + 1);
        return z;
    }
    int zaddr(int x, int y) const { return (((y >> kCtuLog2) * wc_ + (x >> kCtuLog2)) << 6) | zorder6((x & 31) >> 2, (y & 31) >> 2); }
    bool avail(int xc, int yc, int xn, int yn) const   // 6.4.1: earlier in decoding order and in the same tile
    {
        if (!(xn >= 0 && yn >= 0 && xn < w_ && yn < h_ && zaddr(xn, yn) <= zaddr(xc, yc))) return false;
        return !grid_.on() || (grid_.col_of(xn >> kCtuLog2) == grid_.col_of(xc >> kCtuLog2) && grid_.row_of(yn >> kCtuLog2) == grid_.row_of(yc >> kCtuLog2));
    }

    // 7.3.8.3
    void sao(int rx, int ry)
    {
        const mihevc_sao_ctu &s = pic_.sao[ry * wc_ + rx];
        auto same = [](const mihevc_sao_ctu &a, const mihevc_sao_ctu &b) {
            if (a.type[0] != b.type[0] || a.type[1] != b.type[1]) return false;
            for (int c = 0; c < 3; c++) {
                int t = a.type[c ? 1 : 0];
                if (!t) continue;
                if (memcmp(a.offset[c], b.offset[c], 4)) return false;
                if (t == 1 && a.band_pos[c] != b.band_pos[c]) return false;
                if (t == 2 && c < 2 && a.eo_class[c] != b.eo_class[c]) return false;
            }
            return true;
        };
        if (rx > grid_.col_bd(grid_.col_of(rx))) {       // merge candidates: same slice and tile
            bool m = same(s, pic_.sao[ry * wc_ + rx - 1]);
            cabac_.bin(kSaoMerge, m);
            if (m) return;
        }
        if (ry > grid_.row_bd(grid_.row_of(ry))) {
            bool m = same(s, pic_.sao[(ry - 1) * wc_ + rx]);
            cabac_.bin(kSaoMerge, m);
            if (m) return;
        }
        int cmax = (1 << (std::min(cfg_.bit_depth, 10) - 5)) - 1;
        for (int c = 0; c < 3; c++) {
            int t = s.type[c ? 1 : 0];
            if (c < 2) {
                cabac_.bin(kSaoType, t != 0);
                if (t) cabac_.bypass(t == 2);
            }
            if (!t) continue;
            for (int i = 0; i < 4; i++) {
                int a = std::abs((int)s.offset[c][i]);
                for (int k = 0; k < a; k++) cabac_.bypass(1);
                if (a < cmax) cabac_.bypass(0);
            }
            if (t == 1) {
                for (int i = 0; i < 4; i++) if (s.offset[c][i]) cabac_.bypass(s.offset[c][i] < 0);
                cabac_.bypass_bits(s.band_pos[c], 5);
            } else if (c < 2) {
                cabac_.bypass_bits(s.eo_class[c], 2);
            }
        }
    }

    // 7.3.8.4
    void quadtree(int x0, int y0, int log2n, int depth)
    {
        int n = 1 << log2n;
        bool split;
        if (x0 + n <= w_ && y0 + n <= h_ && log2n > 3) {
            split = cu(x0, y0).log2_size < log2n;
            int l = avail(x0, y0, x0 - 1, y0) && depth_[(y0 >> 3) * w8_ + ((x0 - 1) >> 3)] > depth;
            int a = avail(x0, y0, x0, y0 - 1) && depth_[((y0 - 1) >> 3) * w8_ + (x0 >> 3)] > depth;
            cabac_.bin(kSplitCu + l + a, split);
        } else {
            split = log2n > 3;
        }
        if (split) {
            int h = n >> 1;
            for (int k = 0; k < 4; k++) {
                int x1 = x0 + (k & 1) * h, y1 = y0 + (k >> 1) * h;
                if (x1 < w_ && y1 < h_) quadtree(x1, y1, log2n - 1, depth + 1);
            }
            return;
        }
        for (int yy = 0; yy < n; yy += 8)
            for (int xx = 0; xx < n; xx += 8) depth_[((y0 + yy) >> 3) * w8_ + ((x0 + xx) >> 3)] = (uint8_t)depth;
        coding_unit(x0, y0, log2n);
    }

    static Mot motion_of(const mihevc_cu_rec &r)
    {
        Mot m{1, !(r.flags & F_NOL0), (r.flags & F_L1) != 0, 0, 0, 0, 0};
        if (m.f0) { m.x0 = r.mvx; m.y0 = r.mvy; }
        if (m.f1) { m.x1 = (int16_t)(r.intra_mode[0] | (r.intra_mode[1] << 8)); m.y1 = (int16_t)(r.intra_mode[2] | (r.intra_mode[3] << 8)); }
        return m;
    }
    Mot nb(int xc, int yc, int xn, int yn) const
    {
        Mot m{0, 0, 0, 0, 0, 0, 0};
        if (!avail(xc, yc, xn, yn)) return m;
        const mihevc_cu_rec &r = cu(xn, yn);
        if (!(r.flags & F_INTER)) return m;
        return motion_of(r);
    }
    static bool same_mot(const Mot &a, const Mot &b) { return a.ok && b.ok && a.same(b); }
    // 8.5.3.2.2 - 8.5.3.2.5: spatial candidates, (B slices) combined bi-predictive candidates, zero candidates; no temporal candidate
    // (sps_temporal_mvp_enabled_flag = 0).  One reference picture per list, so every reference index is 0.
    void merge_list(int x, int y, int n, Mot out[kMaxMergeCand]) const
    {
        const bool bslice = pic_.slice_type == 0;
        Mot a1 = nb(x, y, x - 1, y + n - 1), b1 = nb(x, y, x + n - 1, y - 1), b0 = nb(x, y, x + n, y - 1);
        Mot a0 = nb(x, y, x - 1, y + n), b2 = nb(x, y, x - 1, y - 1);
        int fa1 = a1.ok, fb1 = b1.ok && !same_mot(b1, a1), fb0 = b0.ok && !same_mot(b0, b1), fa0 = a0.ok && !same_mot(a0, a1);
        int fb2 = b2.ok && !same_mot(b2, a1) && !same_mot(b2, b1) && (fa0 + fa1 + fb0 + fb1 != 4);
        const Mot *order[5] = {&a1, &b1, &b0, &a0, &b2};
        const int flag[5] = {fa1, fb1, fb0, fa0, fb2};
        int k = 0;
        for (int i = 0; i < 5 && k < kMaxMergeCand; i++) if (flag[i]) out[k++] = *order[i];
        if (bslice && k > 1 && k < kMaxMergeCand) {       // 8.5.3.2.4: list-0 motion of one candidate with list-1 motion of another
            static const uint8_t l0c[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1c[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
            const int orig = k;
            for (int c = 0; c < orig * (orig - 1) && k < kMaxMergeCand; c++) {
                const Mot &p = out[l0c[c]], &q = out[l1c[c]];
                // (the two lists hold different pictures, so DiffPicOrderCnt(RefPicList0[0], RefPicList1[0]) != 0 and the vectors need not differ)
                if (p.f0 && q.f1) out[k++] = Mot{1, 1, 1, p.x0, p.y0, q.x1, q.y1};
            }
        }
        while (k < kMaxMergeCand) out[k++] = Mot{1, 1, bslice ? 1 : 0, 0, 0, 0, 0};      // 8.5.3.2.5 (numRefIdx = 1: every zero candidate has reference index 0)
    }
    // 8.5.3.2.6 - 8.5.3.2.7 for list X: spatial candidates; a neighbour's vector into the OTHER list's picture is scaled by the ratio of the picture order
    // count distances (8.5.3.2.7 equations for tx / distScaleFactor), here always -1: the two lists' pictures sit one picture before and after this one
    Mv amvp_pick(const Mot &m, int lx, bool scaled_pass) const
    {
        if (!m.ok) return Mv{0, 0, 0};
        const int fx = lx ? m.f1 : m.f0, fy = lx ? m.f0 : m.f1;
        if (!scaled_pass) return fx ? Mv{1, lx ? m.x1 : m.x0, lx ? m.y1 : m.y0} : Mv{0, 0, 0};      // same picture: only the same list's vector qualifies
        if (fx) return Mv{1, lx ? m.x1 : m.x0, lx ? m.y1 : m.y0};
        if (fy) return Mv{1, scale_mv(lx ? m.x0 : m.x1), scale_mv(lx ? m.y0 : m.y1)};
        return Mv{0, 0, 0};
    }
    static int scale_mv(int v)      // td = -tb = +-1: tx = (16384 + (|td| >> 1)) / td, distScaleFactor = clip3(-4096, 4095, (tb * tx + 32) >> 6) = -256
    {
        const int dsf = -256, p = dsf * v;
        const int r = (std::abs(p) + 127) >> 8;
        return std::min(32767, std::max(-32768, p < 0 ? -r : r));
    }
    void amvp_list(int x, int y, int n, int lx, Mv out[2]) const
    {
        const Mot a0 = nb(x, y, x - 1, y + n), a1 = nb(x, y, x - 1, y + n - 1);
        const Mot b0 = nb(x, y, x + n, y - 1), b1 = nb(x, y, x + n - 1, y - 1), b2 = nb(x, y, x - 1, y - 1);
        const bool scaled = a0.ok || a1.ok;      // isScaledFlagLX (6.4.2 availability already excludes intra neighbours)
        Mv a{0, 0, 0}, b{0, 0, 0};
        for (const Mot *m : {&a0, &a1}) if (!a.ok) a = amvp_pick(*m, lx, false);
        for (const Mot *m : {&a0, &a1}) if (!a.ok) a = amvp_pick(*m, lx, true);
        for (const Mot *m : {&b0, &b1, &b2}) if (!b.ok) b = amvp_pick(*m, lx, false);
        if (!scaled && b.ok) a = b;
        if (!scaled) {
            b = Mv{0, 0, 0};
            for (const Mot *m : {&b0, &b1, &b2}) if (!b.ok) b = amvp_pick(*m, lx, true);
        }
        int k = 0;
        if (a.ok) out[k++] = a;
        if (b.ok && !(a.ok && a.x == b.x && a.y == b.y)) out[k++] = b;
        while (k < 2) out[k++] = Mv{1, 0, 0};
    }
    static int mvd_bits(int d)
    {
        int a = std::abs(d);
        if (a == 0) return 1;
        if (a == 1) return 3;
        return 3 + 2 * (31 - __builtin_clz((unsigned)a));
    }
    void code_mvd(int dx, int dy)   // 7.3.8.9
    {
        int ax = std::abs(dx), ay = std::abs(dy);
        cabac_.bin(kMvdG0, ax > 0);
        cabac_.bin(kMvdG0, ay > 0);
        if (ax > 0) cabac_.bin(kMvdG1, ax > 1);
        if (ay > 0) cabac_.bin(kMvdG1, ay > 1);
        const int a[2] = {ax, ay}, d[2] = {dx, dy};
        for (int i = 0; i < 2; i++) {
            if (!a[i]) continue;
            if (a[i] > 1) {   // EG1(|d| - 2)
                int v = a[i] - 2, k = 1;
                while (v >= (1 << k)) { cabac_.bypass(1); v -= 1 << k; k++; }
                cabac_.bypass(0);
                cabac_.bypass_bits((uint32_t)v, k);
            }
            cabac_.bypass(d[i] < 0);
        }
    }
    void code_merge_idx(int idx)
    {
        for (int i = 0; i < kMaxMergeCand - 1; i++) {
            int b = idx > i;
            if (i == 0) cabac_.bin(kMergeIdx, b); else cabac_.bypass(b);
            if (!b) break;
        }
    }

    void intra_mpm(int x, int y, int cand[3]) const   // 8.4.2
    {
        int a = 1, b = 1;
        if (avail(x, y, x - 1, y)) {
            const mihevc_cu_rec &r = cu(x - 1, y);
            if (!(r.flags & F_INTER)) a = r.intra_mode[(r.flags & F_NXN) ? ((y >> 2) & 1) * 2 + (((x - 1) >> 2) & 1) : 0];   // the PU holding (x-1, y)
        }
        if (avail(x, y, x, y - 1) && ((y - 1) >> kCtuLog2) == (y >> kCtuLog2)) {
            const mihevc_cu_rec &r = cu(x, y - 1);
            if (!(r.flags & F_INTER)) b = r.intra_mode[(r.flags & F_NXN) ? (((y - 1) >> 2) & 1) * 2 + ((x >> 2) & 1) : 0];   // the PU holding (x, y-1)
        }
        if (a == b) {
            if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
            else { cand[0] = a; cand[1] = 2 + ((a + 29) & 31); cand[2] = 2 + ((a - 2 + 1) & 31); }
        } else {
            cand[0] = a; cand[1] = b;
            cand[2] = (a != 0 && b != 0) ? 0 : (a != 1 && b != 1) ? 1 : 26;
        }
    }

    // 7.3.8.5 / 7.3.8.6
    void coding_unit(int x0, int y0, int log2n)
    {
        const mihevc_cu_rec &r = cu(x0, y0);
        int n = 1 << log2n;
        bool inter = r.flags & F_INTER, nxn = r.flags & F_NXN;
        int cbf_any = r.flags & (F_CBF_Y | F_CBF_CB | F_CBF_CR);
        if (pic_.slice_type != 2) {
            Mot ml[kMaxMergeCand];
            int merge_idx = -1;
            const Mot mine = inter ? motion_of(r) : Mot{0, 0, 0, 0, 0, 0, 0};
            if (inter) {
                merge_list(x0, y0, n, ml);
                for (int i = 0; i < kMaxMergeCand; i++)
                    if (ml[i].same(mine)) { merge_idx = i; break; }
            }
            bool skip = inter && merge_idx >= 0 && !cbf_any;
            int l = avail(x0, y0, x0 - 1, y0) && skip_[(y0 >> 3) * w8_ + ((x0 - 1) >> 3)];
            int a = avail(x0, y0, x0, y0 - 1) && skip_[((y0 - 1) >> 3) * w8_ + (x0 >> 3)];
            cabac_.bin(kSkip + l + a, skip);
            for (int yy = 0; yy < n; yy += 8)
                for (int xx = 0; xx < n; xx += 8) skip_[((y0 + yy) >> 3) * w8_ + ((x0 + xx) >> 3)] = skip;
            if (skip) { code_merge_idx(merge_idx); return; }
            cabac_.bin(kPredMode, !inter);
            if (inter) {
                cabac_.bin(kPartMode, 1);                       // PART_2Nx2N
                cabac_.bin(kMergeFlag, merge_idx >= 0);
                if (merge_idx >= 0) {
                    code_merge_idx(merge_idx);                   // rqt_root_cbf inferred 1 (cbf_any is set, else skip)
                } else {
                    if (pic_.slice_type == 0) {                  // inter_pred_idc (9.3.4.2.2: first bin by CtDepth, nPbW + nPbH != 12 here)
                        const bool bi = mine.f0 && mine.f1;
                        cabac_.bin(kInterDir + (kCtuLog2 - log2n), bi);
                        if (!bi) cabac_.bin(kInterDir + 4, mine.f1);
                    }
                    for (int lx = 0; lx < 2; lx++) {             // ref_idx_lX absent (one picture per list), mvd_coding, mvp_lX_flag
                        if (!(lx ? mine.f1 : mine.f0)) continue;
                        const int mx = lx ? mine.x1 : mine.x0, my = lx ? mine.y1 : mine.y0;
                        Mv al[2];
                        amvp_list(x0, y0, n, lx, al);
                        int c0 = mvd_bits(mx - al[0].x) + mvd_bits(my - al[0].y);
                        int c1 = mvd_bits(mx - al[1].x) + mvd_bits(my - al[1].y);
                        int f = c1 < c0;
                        code_mvd(mx - al[f].x, my - al[f].y);
                        cabac_.bin(kMvp, f);
                    }
                    cabac_.bin(kRqtRoot, cbf_any != 0);
                }
                if (cbf_any) transform_tree(x0, y0, x0, y0, log2n, 0, 0, r, 0, 0);
                return;
            }
        }
        // intra
        if (log2n == 3) cabac_.bin(kPartMode, !nxn);
        int parts = nxn ? 4 : 1, pn = nxn ? n / 2 : n;
        int prev[4], idx[4];
        for (int k = 0; k < parts; k++) {
            int cand[3];
            intra_mpm(x0 + (k & 1) * pn, y0 + (k >> 1) * pn, cand);
            int mode = r.intra_mode[k];
            prev[k] = mode == cand[0] || mode == cand[1] || mode == cand[2];
            if (prev[k]) idx[k] = mode == cand[0] ? 0 : mode == cand[1] ? 1 : 2;
            else {
                std::sort(cand, cand + 3);
                int rem = mode;
                for (int i = 2; i >= 0; i--) if (rem > cand[i]) rem--;
                idx[k] = rem;
            }
        }
        for (int k = 0; k < parts; k++) cabac_.bin(kPrevIntra, prev[k]);
        for (int k = 0; k < parts; k++) {
            if (prev[k]) { cabac_.bypass(idx[k] > 0); if (idx[k] > 0) cabac_.bypass(idx[k] > 1); }
            else cabac_.bypass_bits((uint32_t)idx[k], 5);
        }
        {   // intra_chroma_pred_mode (7.4.9.6 / Table 8-2)
            static const uint8_t base[4] = {0, 26, 10, 1};
            int cm = r.chroma_mode, lm = r.intra_mode[0], code = 4;
            if (cm != lm) {
                for (int i = 0; i < 4; i++) if ((base[i] == lm ? 34 : base[i]) == cm) code = i;
            }
            cabac_.bin(kChromaMode, code != 4);
            if (code != 4) cabac_.bypass_bits((uint32_t)code, 2);
        }
        transform_tree(x0, y0, x0, y0, log2n, 0, 0, r, 0, 0);
    }

    // 7.3.8.8 + 7.3.8.10 for the tree shapes the analysis produces (TU = CU, or one split for intra NxN)
    void transform_tree(int x0, int y0, int xb, int yb, int log2n, int depth, int blk, const mihevc_cu_rec &r, int pcb, int pcr)
    {
        bool intra = !(r.flags & F_INTER), nxn = r.flags & F_NXN;
        bool split = nxn && depth == 0;          // inferred, never signalled with max_transform_hierarchy_depth = 0
        int cbf_cb = pcb, cbf_cr = pcr;
        if (log2n > 2) {
            cbf_cb = (r.flags & F_CBF_CB) != 0;
            cbf_cr = (r.flags & F_CBF_CR) != 0;
            if (depth == 0 || pcb) cabac_.bin(kCbfChroma + depth, cbf_cb);
            if (depth == 0 || pcr) cabac_.bin(kCbfChroma + depth, cbf_cr);
        }
        if (split) {
            int h = 1 << (log2n - 1);
            for (int k = 0; k < 4; k++) transform_tree(x0 + (k & 1) * h, y0 + (k >> 1) * h, x0, y0, log2n - 1, depth + 1, k, r, cbf_cb, cbf_cr);
            return;
        }
        int cbf_luma = nxn ? (r.cbf_y4 >> blk) & 1 : (r.flags & F_CBF_Y) != 0;
        if (intra || depth != 0 || cbf_cb || cbf_cr) cabac_.bin(kCbfLuma + (depth == 0 ? 1 : 0), cbf_luma);
        int lmode = intra ? r.intra_mode[nxn ? blk : 0] : 1;
        if (cbf_luma) residual(pic_.coef[0] + (size_t)y0 * w_ + x0, w_, log2n, 0, intra ? scan_idx(log2n, 0, lmode) : 0);
        if (log2n > 2 || blk == 3) {
            int xc = (log2n > 2 ? x0 : xb) >> 1, yc = (log2n > 2 ? y0 : yb) >> 1, l2c = log2n > 2 ? log2n - 1 : 2;
            int sc = intra ? scan_idx(l2c, 1, r.chroma_mode) : 0;
            if (cbf_cb) residual(pic_.coef[1] + (size_t)yc * (w_ >> 1) + xc, w_ >> 1, l2c, 1, sc);
            if (cbf_cr) residual(pic_.coef[2] + (size_t)yc * (w_ >> 1) + xc, w_ >> 1, l2c, 2, sc);
        }
    }
    static int scan_idx(int log2n, int c_idx, int mode)   // 7.4.9.11
    {
        if (log2n == 2 || (log2n == 3 && c_idx == 0)) {
            if (mode >= 6 && mode <= 14) return 2;
            if (mode >= 22 && mode <= 30) return 1;
        }
        return 0;
    }

    // 7.3.8.11 residual_coding (no transform skip, no sign hiding)
    void residual(const int16_t *lv, int stride, int log2n, int c_idx, int scan)
    {
        const int l2sb = log2n - 2, nsb = 1 << l2sb, n = 1 << log2n;
        const uint8_t(*sbscan)[2] = kScans.xy[scan][l2sb];
        const uint8_t(*pscan)[2] = kScans.xy[scan][2];
        // a 4x4 sub-block without a level, seen in four 64-bit reads: most sub-blocks of a P picture's TUs are, and gathering their 16 levels in scan
        // order one by one was the larger part of this function's time
        auto empty = [stride](const int16_t *b) {
            uint64_t r[4];
            for (int j = 0; j < 4; j++) memcpy(&r[j], b + (size_t)j * stride, 8);
            return ((r[0] | r[1]) | (r[2] | r[3])) == 0;
        };
        // gather sub-blocks in scan order, find the last significant position
        int last_sb = -1, last_pos = -1;
        for (int i = (1 << (2 * l2sb)) - 1; i >= 0 && last_sb < 0; i--) {
            const int16_t *b = lv + (size_t)(sbscan[i][1] << 2) * stride + (sbscan[i][0] << 2);
            if (empty(b)) continue;
            for (int k = 15; k >= 0; k--)
                if (b[pscan[k][1] * stride + pscan[k][0]]) { last_sb = i; last_pos = k; break; }
        }
        if (last_sb < 0) return;   // cbf was set for an all-zero block: cannot happen (analysis derives cbf from the levels)
        int lx = (sbscan[last_sb][0] << 2) + pscan[last_pos][0], ly = (sbscan[last_sb][1] << 2) + pscan[last_pos][1];
        if (scan == 2) std::swap(lx, ly);
        // last_sig_coeff_{x,y}_prefix / suffix (9.3.4.2.3)
        int off, shift;
        if (c_idx == 0) { off = 3 * (log2n - 2) + ((log2n - 1) >> 2); shift = (log2n + 1) >> 2; }
        else { off = 15; shift = log2n - 2; }
        auto group = [](int v) { return v < 4 ? v : 2 * (31 - __builtin_clz((unsigned)v)) + ((v >> ((31 - __builtin_clz((unsigned)v)) - 1)) & 1); };
        int px = group(lx), py = group(ly), cmax = (log2n << 1) - 1;
        for (int i = 0; i < px; i++) cabac_.bin(kLastX + off + (i >> shift), 1);
        if (px < cmax) cabac_.bin(kLastX + off + (px >> shift), 0);
        for (int i = 0; i < py; i++) cabac_.bin(kLastY + off + (i >> shift), 1);
        if (py < cmax) cabac_.bin(kLastY + off + (py >> shift), 0);
        if (px > 3) { int nb = (px >> 1) - 1; cabac_.bypass_bits((uint32_t)(lx - ((2 + (px & 1)) << nb)), nb); }
        if (py > 3) { int nb = (py >> 1) - 1; cabac_.bypass_bits((uint32_t)(ly - ((2 + (py & 1)) << nb)), nb); }
        (void)n;
        uint8_t csbf[8][8];
        memset(csbf, 0, sizeof csbf);
        int c1_carry = 1;
        for (int i = last_sb; i >= 0; i--) {
            int xs = sbscan[i][0], ys = sbscan[i][1];
            const int16_t *b = lv + (size_t)(ys << 2) * stride + (xs << 2);
            int right = xs + 1 < nsb ? csbf[ys][xs + 1] : 0, below = ys + 1 < nsb ? csbf[ys + 1][xs] : 0;
            if (i < last_sb && i > 0 && empty(b)) {          // coded_sub_block_flag 0 and nothing else
                cabac_.bin(kCsbf + ((right | below) ? 1 : 0) + (c_idx ? 2 : 0), 0);
                continue;
            }
            int lev[16], nsig = 0;
            int start = i == last_sb ? last_pos : 15;
            for (int k = 0; k < 16; k++) { lev[k] = k <= start ? b[pscan[k][1] * stride + pscan[k][0]] : 0; nsig += lev[k] != 0; }
            bool infer_dc = false;
            if (i < last_sb && i > 0) {
                csbf[ys][xs] = nsig != 0;
                cabac_.bin(kCsbf + ((right | below) ? 1 : 0) + (c_idx ? 2 : 0), nsig != 0);
                infer_dc = true;
            } else csbf[ys][xs] = 1;
            if (!csbf[ys][xs]) continue;
            int prev = right + 2 * below;
            for (int k = (i == last_sb ? last_pos - 1 : 15); k >= 0; k--) {
                if (k == 0 && infer_dc) break;          // DC of a coded sub-block with no other coefficient: inferred 1
                int xp = pscan[k][0], yp = pscan[k][1], xc = (xs << 2) + xp, yc = (ys << 2) + yp, sc;
                if (log2n == 2) { static const uint8_t m[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}; sc = m[(yc << 2) + xc]; }
                else if (xc + yc == 0) sc = 0;
                else {
                    if (prev == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
                    else if (prev == 1) sc = yp == 0 ? 2 : yp == 1 ? 1 : 0;
                    else if (prev == 2) sc = xp == 0 ? 2 : xp == 1 ? 1 : 0;
                    else sc = 2;
                    if (c_idx == 0) { if (xs || ys) sc += 3; sc += log2n == 3 ? (scan == 0 ? 9 : 15) : 21; }
                    else sc += log2n == 3 ? 9 : 12;
                }
                cabac_.bin(kSig + (c_idx ? 27 + sc : sc), lev[k] != 0);
                if (lev[k]) infer_dc = false;
            }
            if (!nsig) continue;   // only possible for i == 0 (coded_sub_block_flag inferred 1, all sig flags 0)
            int ctx_set = (i > 0 && c_idx == 0) ? 2 : 0;
            if (c1_carry == 0) ctx_set++;
            int c1 = 1, ng1 = 0, g2_pos = -1;
            for (int k = 15; k >= 0; k--) {
                if (!lev[k]) continue;
                if (ng1 == 8) break;
                int g1 = std::abs(lev[k]) > 1;
                cabac_.bin(kG1 + ctx_set * 4 + c1 + (c_idx ? 16 : 0), g1);
                ng1++;
                if (g1) { c1 = 0; if (g2_pos < 0) g2_pos = k; }
                else if (c1 > 0 && c1 < 3) c1++;
            }
            c1_carry = c1;
            if (g2_pos >= 0) cabac_.bin(kG2 + ctx_set + (c_idx ? 4 : 0), std::abs(lev[g2_pos]) > 2);
            uint32_t signs = 0; int ns = 0;
            for (int k = 15; k >= 0; k--) if (lev[k]) { signs = (signs << 1) | (uint32_t)(lev[k] < 0); ns++; }
            cabac_.bypass_bits(signs, ns);
            int num = 0, rice = 0;
            for (int k = 15; k >= 0; k--) {
                if (!lev[k]) continue;
                int a = std::abs(lev[k]);
                int base = num < 8 ? (k == g2_pos ? 3 : 2) : 1;
                if (a >= base) {
                    remaining(a - base, rice);
                    if (a > 3 * (1 << rice)) rice = std::min(rice + 1, 4);
                }
                num++;
            }
        }
    }
    // 9.3.3.11 coeff_abs_level_remaining: TR prefix (up to 4 ones) + FL(rice) / EG(rice+1) suffix
    void remaining(int v, int rice)
    {
        int q = v >> rice;
        if (q < 4) {
            for (int i = 0; i < q; i++) cabac_.bypass(1);
            cabac_.bypass(0);
            cabac_.bypass_bits((uint32_t)(v & ((1 << rice) - 1)), rice);
        } else {
            for (int i = 0; i < 4; i++) cabac_.bypass(1);
            int k = rice + 1, r = v - (4 << rice);
            while (r >= (1 << k)) { cabac_.bypass(1); r -= 1 << k; k++; }
            cabac_.bypass(0);
            cabac_.bypass_bits((uint32_t)r, k);
        }
    }

    const mihevc_config &cfg_;
    const PictureSyms &pic_;
    std::vector<uint8_t> scratch_;
    Cabac cabac_;
    int w_, h_, w8_, wc_, hc_;
    TileGrid grid_;
    std::vector<uint8_t> skip_, depth_;
};

}  // namespace

static TileGrid picture_grid(const mihevc_config &cfg, const PictureSyms &pic) { return pic.slice_type == 2 ? tile_grid(cfg) : p_tile_grid(cfg); }

int picture_tiles(const mihevc_config &cfg, const PictureSyms &pic)
{
    const TileGrid g = picture_grid(cfg, pic);
    return g.cols * g.rows;
}

size_t encode_tiles(const mihevc_config &cfg, const PictureSyms &pic, int t0, int t1, std::vector<std::vector<uint8_t>> &sub)
{
    SliceCoder coder(cfg, pic);          // its neighbourhood state (skip flags, depths) is only ever read inside the tile that wrote it
    const TileGrid &grid = coder.grid();
    size_t bins = 0;
    for (int t = t0; t < t1; t++) {
        sub[(size_t)t].clear();
        sub[(size_t)t].reserve(1 << 14);
        bins += coder.run_tile(t % grid.cols, t / grid.cols, sub[(size_t)t]);
    }
    return bins;
}

void assemble_picture(const mihevc_config &cfg, const PictureSyms &pic, const std::vector<std::vector<uint8_t>> &sub, std::vector<uint8_t> &out, bool with_aud)
{
    if (cfg.aud && with_aud) write_aud(pic.slice_type, out);
    // slice_segment_header (7.3.6.1)
    BitWriter w;
    bool idr = pic.slice_type == 2;
    const bool first = !sliced(cfg) || cfg.slice_index == 0;
    w.put1(first);                   // first_slice_segment_in_pic_flag
    if (idr) w.put1(0);              // no_output_of_prior_pics_flag
    const TileGrid grid = picture_grid(cfg, pic);
    const bool pps_tiles = idr ? idr_tiles_on(cfg) : grid.on();
    w.ue(idr && (pps_tiles || p_tile_grid(cfg).on()) ? 1 : 0);  // slice_pic_parameter_set_id: PPS 1 carries the IDR tile grid, PPS 0 the P pictures' (or none)
    if (!first) {                    // slice_segment_address: the slice's first CTB in raster order, Ceil(Log2(PicSizeInCtbsY)) bits
        const CodedSize pc = coded_size(cfg.width, picture_height(cfg));
        const int wc = (pc.w + kCtu - 1) >> kCtuLog2, hc = (pc.h + kCtu - 1) >> kCtuLog2;
        int nb = 0;
        while ((1 << nb) < wc * hc) nb++;
        w.put((uint32_t)(slice_first_row(cfg, cfg.slice_index) * wc), nb);
    }
    w.ue((uint32_t)pic.slice_type);  // 2 = I, 1 = P, 0 = B
    if (!idr) {
        w.put((uint32_t)pic.poc & 0xff, 8);   // slice_pic_order_cnt_lsb
        w.put1(1);                   // short_term_ref_pic_set_sps_flag
        // short_term_ref_pic_set_idx, Ceil(Log2(num_short_term_ref_pic_sets)) bits: none with one set; with B pictures (three sets) the set follows from the
        // picture's place in its GOP: B pictures sit at odd positions between two anchors, an anchor at an odd position follows its predecessor directly
        // (PictureSyms::ref_dist says it outright when a session mixes chunks with and without B pictures: bframes = -1)
        if (cfg.bframes != 0) w.put(pic.slice_type == 0 ? 2u : pic.ref_dist ? (pic.ref_dist == 1 ? 0u : 1u) : (pic.poc & 1) ? 0u : 1u, 2);
    }
    if (cfg.sao != 0) {
        w.put1(pic.sao != nullptr);  // slice_sao_luma_flag
        w.put1(pic.sao != nullptr);  // slice_sao_chroma_flag
    }
    if (!idr) {
        w.put1(0);                   // num_ref_idx_active_override_flag
        if (pic.slice_type == 0) w.put1(0);   // mvd_l1_zero_flag
        w.ue(5 - kMaxMergeCand);     // five_minus_max_num_merge_cand
    }
    w.se(pic.qp - 26);               // slice_qp_delta
    if (!sliced(cfg) || cfg.slice_halo != 0) w.put1(1);     // slice_loop_filter_across_slices_enabled_flag (present only when the PPS flag is set)
    const int n_tiles = grid.cols * grid.rows;
    if (pps_tiles) {
        // entry points (7.4.7.1): substream sizes in bytes of the NAL payload, emulation prevention bytes included.  Every
        // substream (and the header) ends in a byte that holds its final '1' bit, so the zero run that triggers an 0x03
        // never crosses a boundary and each size can be counted on its own.
        std::vector<uint32_t> esc((size_t)n_tiles - 1);
        uint32_t max_off = 0;
        for (int t = 0; t + 1 < n_tiles; t++) {
            const std::vector<uint8_t> &v = sub[(size_t)t];
            uint32_t n = (uint32_t)v.size();
            int zeros = 0;
            for (uint8_t b : v) {
                if (zeros >= 2 && b <= 3) { n++; zeros = 0; }
                zeros = b == 0 ? zeros + 1 : 0;
            }
            esc[(size_t)t] = n;
            max_off = std::max(max_off, n - 1);
        }
        int len = 1;
        while (len < 32 && (max_off >> len)) len++;
        w.ue((uint32_t)n_tiles - 1);     // num_entry_point_offsets
        if (n_tiles > 1) {
            w.ue((uint32_t)len - 1);     // offset_len_minus1
            for (uint32_t e : esc) w.put(e - 1, len);   // entry_point_offset_minus1
        }
    }
    w.trailing();                    // byte_alignment(): same bit pattern as rbsp_trailing_bits
    std::vector<uint8_t> rbsp = std::move(w.bytes());
    size_t total = rbsp.size();
    for (const auto &v : sub) total += v.size();
    rbsp.reserve(total);
    for (const auto &v : sub) rbsp.insert(rbsp.end(), v.begin(), v.end());
    append_nal(out, idr ? 19 : pic.slice_type == 0 ? 0 : 1, rbsp);      // IDR_W_RADL, TRAIL_N (a B picture is never a reference), TRAIL_R
}

size_t encode_picture(const mihevc_config &cfg, const PictureSyms &pic, std::vector<uint8_t> &out, bool with_aud)
{
    const int n_tiles = picture_tiles(cfg, pic);
    std::vector<std::vector<uint8_t>> sub((size_t)n_tiles);
    const size_t bins = encode_tiles(cfg, pic, 0, n_tiles, sub);
    assemble_picture(cfg, pic, sub, out, with_aud);
    return bins;
}

}  // namespace mihevc
