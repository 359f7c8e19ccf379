// hevc_amd/csrc/bitstream.h — host-side HEVC bitstream writer (parameter sets, slice header, CABAC slice data).
//
// Stays on host cores by design (BASELINE.json north_star: "CABAC entropy coding and MP4 mux stay on host cores").
// Replaces what libx265 does behind `ffmpeg -c:v libx265` in the reference (core/transcoder.py:412,463,506) for the
// syntax subset the device analysis produces.  Clause numbers refer to ITU-T H.265.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/mihevc.h"

namespace mihevc {

constexpr int kCtuLog2 = 5;
constexpr int kCtu = 32;
constexpr int kMaxMergeCand = 5;

enum : uint8_t { F_INTER = 1, F_CBF_Y = 2, F_CBF_CB = 4, F_CBF_CR = 8, F_NXN = 16, F_L1 = 32, F_NOL0 = 64 };      // F_L1 / F_NOL0: inter CUs of B pictures, see mihevc_cu_rec

struct CodedSize {
    int w, h;          // multiples of 8
    int crop_r, crop_b; // luma samples cropped by the conformance window
};
inline CodedSize coded_size(int width, int height)
{
    CodedSize c;
    c.w = (width + 7) & ~7;
    c.h = (height + 7) & ~7;
    c.crop_r = c.w - width;
    c.crop_b = c.h - height;
    return c;
}

class BitWriter {
public:
    void put(uint32_t v, int n);            // n <= 32, MSB first
    void put1(int b) { put((uint32_t)(b != 0), 1); }
    void ue(uint32_t v);
    void se(int32_t v);
    void trailing();                         // rbsp_trailing_bits
    void align_zero();
    bool aligned() const { return nbits_ == 0; }
    std::vector<uint8_t> &bytes() { return buf_; }
    const std::vector<uint8_t> &bytes() const { return buf_; }
    void append_bytes(const uint8_t *p, size_t n) { buf_.insert(buf_.end(), p, p + n); }

private:
    std::vector<uint8_t> buf_;
    uint32_t acc_ = 0;
    int nbits_ = 0;
};

// append start code + 2-byte NAL header + escaped payload to out (7.3.1.1, Annex B)
void append_nal(std::vector<uint8_t> &out, int nal_type, const std::vector<uint8_t> &rbsp);

void write_vps(const mihevc_config &cfg, std::vector<uint8_t> &out);
void write_sps(const mihevc_config &cfg, std::vector<uint8_t> &out);
void write_pps(const mihevc_config &cfg, int pps_id, std::vector<uint8_t> &out);   // id 0: one tile (P pictures); id 1: the IDR tile grid

// uniform tile grid of IDR pictures (6.5.1); cols == rows == 1 when tiles are off
struct TileGrid {
    int cols = 1, rows = 1, wc = 0, hc = 0;
    int col_bd(int i) const { return i * wc / cols; }
    int row_bd(int j) const { return j * hc / rows; }
    int col_of(int ctb_x) const { int i = 0; while (i + 1 < cols && col_bd(i + 1) <= ctb_x) i++; return i; }
    int row_of(int ctb_y) const { int j = 0; while (j + 1 < rows && row_bd(j + 1) <= ctb_y) j++; return j; }
    bool on() const { return cols > 1 || rows > 1; }
};
TileGrid tile_grid(const mihevc_config &cfg);        // of what the session codes: the picture, or its slice (cfg.slice_count > 1)
TileGrid p_tile_grid(const mihevc_config &cfg);      // P pictures (PPS 0, cfg.p_tiles); 1x1 when off or when the picture is coded as several slices
// sliced pictures (cfg.slice_count > 1): height of the whole picture, first CTU row of a slice, tile rows PPS 1 gives a slice
inline bool sliced(const mihevc_config &c) { return c.slice_count > 1; }
inline int picture_height(const mihevc_config &c) { return sliced(c) ? c.pic_height : c.height; }
int slice_first_row(const mihevc_config &cfg, int k);
int slice_tile_rows(const mihevc_config &cfg, int k);
bool idr_tiles_on(const mihevc_config &cfg);          // does PPS 1 enable tiles for the picture (a slice of it may still be one tile)
void write_sei_hdr10(const mihevc_config &cfg, std::vector<uint8_t> &out);
void write_aud(int slice_type, std::vector<uint8_t> &out);
// HRD signalling (cfg.hrd with a VBV): E.2.2 hrd_parameters in the VUI; D.2.2 buffering period at IRAP pictures, D.2.3 picture timing
struct HrdInfo {
    bool on;
    uint32_t bit_rate_value_minus1, cpb_size_value_minus1;   // scales 0: units of 64 bit/s and 16 bits
    uint32_t initial_delay, initial_offset;                  // 90 kHz: 0.9 x CPB fullness at the first removal, and the rest of the CPB
};
HrdInfo hrd_info(const mihevc_config &cfg);
void write_sei_buffering_period(const mihevc_config &cfg, std::vector<uint8_t> &out);
void write_sei_pic_timing(const mihevc_config &cfg, uint32_t au_cpb_removal_delay_minus1, std::vector<uint8_t> &out, uint32_t pic_dpb_output_delay = 0);
void write_parameter_sets(const mihevc_config &cfg, std::vector<uint8_t> &out);

// One picture's symbols (pointers into pinned host copies of the device outputs).
struct PictureSyms {
    int slice_type;      // 2 = I (coded as IDR_W_RADL), 1 = P (TRAIL_R), 0 = B (TRAIL_N: between two anchors, cfg.bframes)
    int poc;             // position in the closed GOP in DISPLAY order (0 for the IDR)
    int qp;              // slice QP
    const mihevc_cu_rec *cu;      // (h/8) x (w/8)
    const int16_t *coef[3];       // TU-local raster at picture coordinates; strides w, w/2, w/2
    const mihevc_sao_ctu *sao;    // per CTU, or nullptr when SAO is off for the picture
    int ref_dist = 0;             // P pictures of a stream that announces B pictures (cfg.bframes != 0): pictures between this one and its reference + 1 (1 or 2);
                                  // 0: derive it from the place in the GOP (the fixed I0 P2 b1 P4 b3 ... layout)
};

// CABAC-code the picture into one slice-segment NAL appended to out; returns the number of bins coded (stats).
// with_aud: prepend the access unit delimiter when cfg.aud (callers that put parameter sets / SEI into the same access unit write
// the AUD themselves, it must come first: 7.4.2.4.4)
size_t encode_picture(const mihevc_config &cfg, const PictureSyms &pic, std::vector<uint8_t> &out, bool with_aud = true);
// The same in pieces, so that the tiles of one picture can be coded by several host threads: every tile is its own CABAC substream and looks at
// nothing outside itself.  picture_tiles: substreams of this picture; encode_tiles: tiles [t0, t1) into sub[t] (any thread, any order, each range
// once); assemble_picture: slice header + entry points + substreams -> one NAL.  encode_picture = all three in one thread.
int picture_tiles(const mihevc_config &cfg, const PictureSyms &pic);
size_t encode_tiles(const mihevc_config &cfg, const PictureSyms &pic, int t0, int t1, std::vector<std::vector<uint8_t>> &sub);
void assemble_picture(const mihevc_config &cfg, const PictureSyms &pic, const std::vector<std::vector<uint8_t>> &sub, std::vector<uint8_t> &out, bool with_aud);

}  // namespace mihevc
