// hevc_amd/csrc/kernels/common.h — shared definitions of the gfx950 kernels.
//
// Every per-CTU kernel is written as a "phase program": a sequence of phases, each a lambda run by all threads of
// the workgroup followed by a workgroup barrier (`ex.phase(...)`).  Thread-private values never live across a phase
// boundary; everything that does lives in the workgroup's LDS image (`Shared` structs).  The HIP executor
// (GpuExec, below) maps a phase to `f(threadIdx.x); __syncthreads();`.  tests/emu/ instantiates the same programs
// with a sequential executor to step the kernel source on a CPU — a structural test of the kernel logic, not a
// product path (the package never loads it).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DEV __device__ __forceinline__
#define HD __host__ __device__
#define HDI __host__ __device__ inline
#define DEVCONST __device__ const
#define MIHEVC_GPU 1
#else
#define DEV inline
#define HD
#define HDI inline
#define DEVCONST static const
#define MIHEVC_GPU 0
#endif

#include "../../../include/mihevc.h"

namespace mihevc {

constexpr int CTU = 32;
constexpr int CTU_LOG2 = 5;
constexpr int PAD_Y = 80;     // luma border of reference planes
constexpr int PAD_C = 40;
constexpr int NT = 256;       // threads per CTU workgroup
constexpr int MAX_RANGE = 64;

enum : uint8_t { CU_INTER = 1, CU_CBF_Y = 2, CU_CBF_CB = 4, CU_CBF_CR = 8, CU_NXN = 16, CU_L1 = 32, CU_NOL0 = 64 };      // CU_L1 / CU_NOL0: inter CUs of B pictures (include/mihevc.h)

template <typename T> struct PixTraits;
template <> struct PixTraits<uint8_t> { static constexpr int kBitDepth = 8; };
template <> struct PixTraits<uint16_t> { static constexpr int kBitDepth = 10; };

// a sample plane; `p` addresses sample (0,0); padded planes hold valid samples at negative coordinates
template <typename T> struct Plane {
    T *p;
    int stride;
};

struct CostParams {
    int qp, qp_c, bit_depth, lambda_sad_q4, lambda_q4, me_range;
    int tile_cols, tile_rows;     // intra pictures: uniform tile grid (6.5.1), 1x1 = no tiles
    int intra_nxn;                // 1: 8x8 intra CUs are also tried as four 4x4 PUs (NxN, DST-VII luma TUs)
    int intra_in_p;               // 1: P pictures run the intra second pass (kernels/intra.h intra_p_eligible)
    int pre_search;               // 1: search centres come from a +-PRE_RANGE full search on the 1/4-size pictures (kernels/inter.h)
    int rdo_zero;                 // 1: inter TUs whose levels cost more than the distortion they remove are zeroed (inter_ctu_program)
    int chroma_modes;             // 1: 2Nx2N intra CUs choose among DM / planar / 26 / 10 / DC for chroma (intra_cu)
    int mc_top, mc_bottom;        // 1: the picture is a slice whose upper / lower neighbour lives on another device: motion compensation stays inside
    int rdo_cg;                   // k > 0: RD zero-out of the 4x4 coefficient groups of inter TUs with lambda x k / 2 (residual_pipeline); 0: off
};
// rate model in 1/16 bit (oracle/hevc_oracle.c ORC_R_*, fitted to the host CABAC): a level, a 4x4 sub-block with a level, a TU with a level,
// an inter CU, an intra CU
DEV int rate_level(int a) { return a == 1 ? 33 : a == 2 ? 50 : 53 + 27 * (31 - __builtin_clz((unsigned)(a - 1) | 1)); }
constexpr int R_SB = 143, R_TU = 30;
constexpr unsigned R_INTER_CU = 80, R_INTRA_CU = 128;
constexpr int PRE_RANGE = 14;     // low-resolution samples: centres reach +-56 luma samples, window reads stay inside the 80-sample border
// per-CTU hand-over from the inter pass of a P picture to its intra second pass
struct IpInfo {
    unsigned long long jinter;    // J of the inter version: SSE << 4 + lambda * estimated bits
    unsigned est;                 // the inter version's share of the picture's rate estimate (1/16 bit)
    unsigned cand;                // 1: the inter cost exceeds the source's own activity -> try intra
};
// uniform tile spacing (6.5.1): first CTB of tile i of n over n_ctb CTBs, and the tile holding a CTB
HDI int tile_bd(int i, int n, int n_ctb) { return i * n_ctb / n; }
HDI int tile_of(int ctb, int n, int n_ctb) { int i = 0; while (i + 1 < n && tile_bd(i + 1, n, n_ctb) <= ctb) i++; return i; }

// ------------------------------------------------------------------------------------------ tables
struct Tables {
    int8_t mat[32][32];       // H.265 8.6.4.2 transMatrix (32-point; N-point = rows k*(32/N), first N columns)
    int8_t dst4[4][4];
    int8_t luma_tap[4][8];    // 8.5.3.3.3.1 Table 8-11
    int8_t chroma_tap[8][4];  // Table 8-12
    int8_t intra_angle[35];   // 8.4.4.2.6 Table 8-4
    int16_t inv_angle[15];    // Table 8-5 (modes 11..25)
    uint8_t beta[52], tc[54]; // 8.7.2.5.3 Table 8-12 (beta', tc')
    int16_t quant_scale[6];
    uint8_t level_scale[6];
    int8_t chroma_qp[14];     // Table 8-10 for qPi 30..43
    uint8_t zorder[16];       // 4x4 grid of 8x8 tiles: raster index -> z-order index
};

constexpr Tables make_tables()
{
    Tables t{};
    const int c64[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
    for (int k = 0; k < 32; k++)
        for (int n = 0; n < 32; n++) {
            int a = (k * (2 * n + 1)) % 128;
            if (a > 64) a = 128 - a;
            t.mat[k][n] = (int8_t)(a > 32 ? -c64[64 - a] : c64[a]);
        }
    const int d[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t.dst4[i][j] = (int8_t)d[i][j];
    const int lt[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) t.luma_tap[i][j] = (int8_t)lt[i][j];
    const int ct[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) t.chroma_tap[i][j] = (int8_t)ct[i][j];
    const int ang[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
    for (int i = 0; i < 35; i++) t.intra_angle[i] = (int8_t)ang[i];
    const int inv[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
    for (int i = 0; i < 15; i++) t.inv_angle[i] = (int16_t)inv[i];
    for (int q = 0; q < 52; q++) t.beta[q] = (uint8_t)(q < 16 ? 0 : q < 29 ? q - 10 : 2 * q - 38);
    const int tcv[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                         5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
    for (int i = 0; i < 54; i++) t.tc[i] = (uint8_t)tcv[i];
    const int qs[6] = {26214, 23302, 20560, 18396, 16384, 14564}, ls[6] = {40, 45, 51, 57, 64, 72};
    for (int i = 0; i < 6; i++) { t.quant_scale[i] = (int16_t)qs[i]; t.level_scale[i] = (uint8_t)ls[i]; }
    const int cq[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
    for (int i = 0; i < 14; i++) t.chroma_qp[i] = (int8_t)cq[i];
    for (int ty = 0; ty < 4; ty++)
        for (int tx = 0; tx < 4; tx++) t.zorder[ty * 4 + tx] = (uint8_t)(((ty >> 1) * 2 + (tx >> 1)) * 4 + (ty & 1) * 2 + (tx & 1));
    return t;
}
DEVCONST Tables g_tab = make_tables();

// ------------------------------------------------------------------------------------------ small helpers
DEV int iabs(int v) { return v < 0 ? -v : v; }
DEV int imin(int a, int b) { return a < b ? a : b; }
DEV int imax(int a, int b) { return a > b ? a : b; }
DEV int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
DEV int ilog2u(unsigned v) { return 31 - __builtin_clz(v | 1); }
DEV int mvd_bits(int d)
{
    int a = iabs(d);
    return a == 0 ? 1 : a == 1 ? 3 : 3 + 2 * ilog2u((unsigned)a);
}
DEV int chroma_qp_of(int qp)
{
    int q = clip3(-12, 57, qp);
    return q < 30 ? q : q > 43 ? q - 6 : g_tab.chroma_qp[q - 30];
}
// z-scan address of the 4x4 unit holding luma sample (x,y) (6.4.1 at min-TB granularity)
DEV int zaddr(int x, int y, int ctus_w)
{
    int bx = (x & 31) >> 2, by = (y & 31) >> 2, z = 0;
    for (int i = 0; i < 3; i++) z |= ((bx >> i) & 1) << (2 * i) | ((by >> i) & 1) << (2 * i + 1);
    return (((y >> CTU_LOG2) * ctus_w + (x >> CTU_LOG2)) << 6) | z;
}
// quadtree node geometry: 0 = 32x32, 1..4 = 16x16 (z-order), 5..20 = 8x8 (z-order inside each 16x16)
DEV void node_geom(int node, int &x, int &y, int &log2n)
{
    if (node == 0) { x = 0; y = 0; log2n = 5; }
    else if (node < 5) { int q = node - 1; x = (q & 1) * 16; y = (q >> 1) * 16; log2n = 4; }
    else { int q = (node - 5) >> 2, s = (node - 5) & 3; x = (q & 1) * 16 + (s & 1) * 8; y = (q >> 1) * 16 + (s >> 1) * 8; log2n = 3; }
}
// node that covers tile (tx,ty) of the 4x4 tile grid at level 0 (32), 1 (16), 2 (8)
DEV int node_of_tile(int level, int tx, int ty)
{
    if (level == 0) return 0;
    int q = (ty >> 1) * 2 + (tx >> 1);
    return level == 1 ? 1 + q : 5 + q * 4 + (ty & 1) * 2 + (tx & 1);
}

// sum of absolute differences of 4 (8-bit) / 2 (16-bit) packed samples
DEV uint32_t sad_packed_u8(uint32_t a, uint32_t b, uint32_t acc)
{
#if MIHEVC_GPU
    return __builtin_amdgcn_sad_u8(a, b, acc);
#else
    for (int i = 0; i < 4; i++) acc += (uint32_t)iabs((int)((a >> (8 * i)) & 255) - (int)((b >> (8 * i)) & 255));
    return acc;
#endif
}
DEV uint32_t sad_packed_u16(uint32_t a, uint32_t b, uint32_t acc)
{
#if MIHEVC_GPU
    return __builtin_amdgcn_sad_u16(a, b, acc);
#else
    for (int i = 0; i < 2; i++) acc += (uint32_t)iabs((int)((a >> (16 * i)) & 65535) - (int)((b >> (16 * i)) & 65535));
    return acc;
#endif
}
DEV uint32_t load_u32(const void *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// SAD of one 8-sample row
DEV uint32_t sad_row8(const uint8_t *a, const uint8_t *b, uint32_t acc)
{
    acc = sad_packed_u8(load_u32(a), load_u32(b), acc);
    return sad_packed_u8(load_u32(a + 4), load_u32(b + 4), acc);
}
DEV uint32_t sad_row8(const uint16_t *a, const uint16_t *b, uint32_t acc)
{
    for (int i = 0; i < 8; i += 2) acc = sad_packed_u16(load_u32(a + i), load_u32(b + i), acc);
    return acc;
}

// two 16-bit lanes per dword (VOP3P v_pk_add_u16 / v_pk_sub_i16 / v_pk_max_i16): the callers keep every value inside 16 bits
DEV uint32_t pk_add16(uint32_t a, uint32_t b)
{
#if MIHEVC_GPU
    typedef short v2s __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (v2s)(__builtin_bit_cast(v2s, a) + __builtin_bit_cast(v2s, b)));
#else
    return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16);
#endif
}
DEV uint32_t pk_sub16(uint32_t a, uint32_t b)
{
#if MIHEVC_GPU
    typedef short v2s __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (v2s)(__builtin_bit_cast(v2s, a) - __builtin_bit_cast(v2s, b)));
#else
    return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16);
#endif
}
DEV uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
#if MIHEVC_GPU
    uint32_t r;       // (the vector builtin is split into two 16-bit compare / select pairs by this compiler: four instructions instead of one)
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    const int al = (int16_t)(a & 0xffffu), ah = (int16_t)(a >> 16), bl = (int16_t)(b & 0xffffu), bh = (int16_t)(b >> 16);
    return (uint32_t)((al > bl ? al : bl) & 0xffff) | ((uint32_t)((ah > bh ? ah : bh) & 0xffff) << 16);
#endif
}

// v_perm_b32: byte i of the result is byte (sel >> 8i) & 7 of the eight bytes {a (4 .. 7), b (0 .. 3)}
DEV uint32_t perm_bytes(uint32_t a, uint32_t b, uint32_t sel)
{
#if MIHEVC_GPU
    return __builtin_amdgcn_perm(a, b, sel);
#else
    const uint64_t ab = ((uint64_t)a << 32) | b;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((ab >> (8 * ((sel >> (8 * i)) & 7))) & 255u) << (8 * i);
    return r;
#endif
}
// 4 x 4 bytes transposed: r[k] = four bytes of row k -> c[i] = byte i of rows 0 .. 3
DEV void transpose_bytes4(const uint32_t (&r)[4], uint32_t (&c)[4])
{
    const uint32_t t0 = perm_bytes(r[1], r[0], 0x05010400u), t1 = perm_bytes(r[1], r[0], 0x07030602u);
    const uint32_t t2 = perm_bytes(r[3], r[2], 0x05010400u), t3 = perm_bytes(r[3], r[2], 0x07030602u);
    c[0] = perm_bytes(t2, t0, 0x05040100u); c[1] = perm_bytes(t2, t0, 0x07060302u);
    c[2] = perm_bytes(t3, t1, 0x05040100u); c[3] = perm_bytes(t3, t1, 0x07060302u);
}

// true when the predicate holds in every ACTIVE lane of the wave: lets a function pick a cheaper, value-identical path without diverging (a wave executes both sides of a branch its
// lanes disagree on).  Stepped on the CPU a lane stands alone; the paths give the same values, so any choice is right there.
DEV bool wave_all(bool p)
{
#if MIHEVC_GPU
    return __all(p) != 0;
#else
    return p;
#endif
}

// v_qsad_pk_u16_u8: four SADs of 4 packed bytes `cur` against the 4-byte windows of `ref8` at byte offsets 0..3,
// each accumulated into its own 16-bit field of `acc` (the instruction motion estimation was given on GCN/CDNA)
DEV uint64_t qsad_u8(uint64_t ref8, uint32_t cur, uint64_t acc)
{
#if MIHEVC_GPU
    return __builtin_amdgcn_qsad_pk_u16_u8(ref8, cur, acc);
#else
    uint64_t out = 0;
    for (int j = 0; j < 4; j++) {
        uint32_t w = (uint32_t)(ref8 >> (8 * j)), a = (uint32_t)((acc >> (16 * j)) & 0xffff);
        for (int i = 0; i < 4; i++) a += (uint32_t)iabs((int)((w >> (8 * i)) & 255) - (int)((cur >> (8 * i)) & 255));
        out |= (uint64_t)(a & 0xffff) << (16 * j);
    }
    return out;
#endif
}
DEV uint32_t align_bytes(uint32_t hi, uint32_t lo, int shift_bytes)     // ({hi,lo} >> 8*shift) & 0xffffffff, shift 0..3
{
#if MIHEVC_GPU
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)shift_bytes);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * shift_bytes));
#endif
}
// Copy a w x h window (w a multiple of 4 samples... of 4 BYTES for 8-bit, 2 samples for 16-bit) of a padded plane into
// an LDS image with row stride ls, one dword per lane-iteration and no integer division: lanes are laid out as 32
// dword-columns x 8 rows.  Source coordinates are clamped into the plane's border so partial CTUs at the picture edge
// never read outside the allocation (the clamped samples are never used).
template <typename T>
DEV void copy_window(T *lds, int ls, const T *plane, int pstride, int ox, int oy, int w, int h, int lo_x, int hi_x, int lo_y, int hi_y, int tid)
{
    constexpr int per = 4 / (int)sizeof(T);            // samples per dword
    constexpr int CH = 6;                              // row groups per batch: all their loads are issued before the first LDS store, so a lane
                                                       // waits for memory once per batch, not once per row (the load phases of the CTU kernels
                                                       // were chains of dependent round trips: profiles/r02 phase table)
    const int dpr = w / per, col = tid & 31, rg = tid >> 5;
    for (int d = col; d < dpr; d += 32) {
        const int x = clip3(lo_x, hi_x - (per - 1), ox + d * per);
        for (int r0 = rg; r0 < h; r0 += CH * (NT / 32)) {
            uint32_t v[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = r0 + k * (NT / 32);
                if (r < h) v[k] = load_u32(plane + (ptrdiff_t)clip3(lo_y, hi_y, oy + r) * pstride + x);
            }
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = r0 + k * (NT / 32);
                if (r < h) __builtin_memcpy(__builtin_assume_aligned(lds + r * ls + d * per, 4), &v[k], 4);
            }
        }
    }
}

// the workgroup's CTU source image (Y 32x32, then U, V 16x16; zero outside the picture) — dword loads for whole CTUs, all of a lane's
// loads in flight together; sample-wise only for the partial CTUs at the right / bottom picture edge
template <typename T>
DEV void load_ctu_source(T *dst, const Plane<const T> (&src)[3], int x0, int y0, int w, int h, int tid)
{
    constexpr int per = 4 / (int)sizeof(T);
    if (x0 + CTU <= w && y0 + CTU <= h) {
        constexpr int ND = 1536 / per, IT = (ND + NT - 1) / NT;
        uint32_t v[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int d = tid + k * NT;
            if (d < ND) {
                const int i = d * per, pl = i < 1024 ? 0 : 1 + ((i - 1024) >> 8), kk = pl ? (i - 1024) & 255 : i, x = pl ? kk & 15 : kk & 31, y = pl ? kk >> 4 : kk >> 5;
                v[k] = load_u32(src[pl].p + (size_t)((pl ? y0 >> 1 : y0) + y) * src[pl].stride + (pl ? x0 >> 1 : x0) + x);
            }
        }
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int d = tid + k * NT;
            if (d < ND) __builtin_memcpy(__builtin_assume_aligned(dst + d * per, 4), &v[k], 4);
        }
        return;
    }
    for (int i = tid; i < 1536; i += NT) {
        int pl, x, y;
        if (i < 1024) { pl = 0; x = i & 31; y = i >> 5; } else { int k = i - 1024; pl = 1 + (k >> 8); k &= 255; x = k & 15; y = k >> 4; }
        const int gx = (pl ? x0 >> 1 : x0) + x, gy = (pl ? y0 >> 1 : y0) + y, pw = pl ? w >> 1 : w, ph = pl ? h >> 1 : h;
        dst[i] = (gx < pw && gy < ph) ? src[pl].p[(size_t)gy * src[pl].stride + gx] : (T)0;
    }
}

// returns 0 in a way the optimiser cannot see through: stops loop-invariant hoisting of whole LDS tiles into VGPRs
DEV int opaque_zero()
{
    int z = 0;
#if MIHEVC_GPU
    asm volatile("" : "+v"(z));
#endif
    return z;
}
DEV uint32_t load_u32_aligned(const void *p)
{
    uint32_t v;
    __builtin_memcpy(&v, __builtin_assume_aligned(p, 4), 4);
    return v;
}
DEV void store_u32_aligned(void *p, uint32_t v) { __builtin_memcpy(__builtin_assume_aligned(p, 4), &v, 4); }
// four horizontally adjacent samples / levels as dword stores (p is 4-sample aligned): single-byte global stores cost one
// instruction and one partial-line write request each
DEV void store4(uint8_t *p, int v0, int v1, int v2, int v3) { store_u32_aligned(p, (uint32_t)v0 | (uint32_t)v1 << 8 | (uint32_t)v2 << 16 | (uint32_t)v3 << 24); }
DEV void store4(uint16_t *p, int v0, int v1, int v2, int v3)
{
    store_u32_aligned(p, (uint32_t)v0 | (uint32_t)v1 << 16);
    store_u32_aligned(p + 2, (uint32_t)v2 | (uint32_t)v3 << 16);
}
DEV void store4(int16_t *p, int v0, int v1, int v2, int v3)
{
    store_u32_aligned(p, ((uint32_t)v0 & 0xffffu) | (uint32_t)v1 << 16);
    store_u32_aligned(p + 2, ((uint32_t)v2 & 0xffffu) | (uint32_t)v3 << 16);
}
// 15 consecutive samples base[idx .. idx+14] from an LDS image whose `base` is 4-byte aligned, fetched with aligned
// dword reads + v_alignbyte (unaligned ds_read_b64 stalls: SQ_LDS_UNALIGNED_STALL, profiles/r01_a_first)
DEV void load_row15(const uint8_t *base, int idx, int (&px)[15])
{
    const int off = idx & 3;
    const uint8_t *p = base + (idx - off);
    uint32_t d[5];
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = load_u32_aligned(p + 4 * k);
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = align_bytes(d[k + 1], d[k], off);
#pragma unroll
    for (int i = 0; i < 15; i++) px[i] = (int)((r[i >> 2] >> (8 * (i & 3))) & 255);
}
DEV void load_row15(const uint16_t *base, int idx, int (&px)[15])
{
    const int off = idx & 1;
    const uint16_t *p = base + (idx - off);
    uint32_t d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = load_u32_aligned(p + 2 * k);
    uint32_t r[8];
#pragma unroll
    for (int k = 0; k < 7; k++) r[k] = align_bytes(d[k + 1], d[k], 2 * off);
    r[7] = align_bytes(0, d[7], 2 * off);
#pragma unroll
    for (int i = 0; i < 15; i++) px[i] = (int)((r[i >> 1] >> (16 * (i & 1))) & 65535);
}

// 8 consecutive samples base[idx .. idx+7], same aligned-read scheme
DEV void load_row8(const uint8_t *base, int idx, int (&px)[8])
{
    const int off = idx & 3;
    const uint8_t *p = base + (idx - off);
    const uint32_t d0 = load_u32_aligned(p), d1 = load_u32_aligned(p + 4), d2 = load_u32_aligned(p + 8);
    const uint32_t r0 = align_bytes(d1, d0, off), r1 = align_bytes(d2, d1, off);
#pragma unroll
    for (int i = 0; i < 4; i++) { px[i] = (int)((r0 >> (8 * i)) & 255); px[4 + i] = (int)((r1 >> (8 * i)) & 255); }
}
DEV void load_row8(const uint16_t *base, int idx, int (&px)[8])
{
    const int off = idx & 1;
    const uint16_t *p = base + (idx - off);
    uint32_t d[5];
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = load_u32_aligned(p + 2 * k);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t r = align_bytes(d[k + 1], d[k], 2 * off);
        px[2 * k] = (int)(r & 65535); px[2 * k + 1] = (int)(r >> 16);
    }
}

// sum of four signed-byte products + acc (v_dot4_i32_i8)
DEV int dot4_i8(uint32_t a, uint32_t b, int acc)
{
#if MIHEVC_GPU
    return __builtin_amdgcn_sdot4((int)a, (int)b, acc, false);
#else
    for (int i = 0; i < 4; i++) acc += (int)(int8_t)(a >> (8 * i)) * (int)(int8_t)(b >> (8 * i));
    return acc;
#endif
}
// a.lo*b.lo + a.hi*b.hi + acc on packed signed 16-bit pairs (v_dot2_i32_i16)
DEV int dot2_i16(uint32_t a, uint32_t b, int acc)
{
#if MIHEVC_GPU
    typedef short v2s __attribute__((ext_vector_type(2)));
    v2s va, vb;
    __builtin_memcpy(&va, &a, 4);
    __builtin_memcpy(&vb, &b, 4);
    return __builtin_amdgcn_sdot2(va, vb, acc, false);
#else
    return acc + (int)(int16_t)(a & 0xffff) * (int)(int16_t)(b & 0xffff) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16);
#endif
}

// (lo & 0xffff) | (hi << 16): two 16-bit values into one dword (v_perm_b32)
DEV uint32_t pack_lo16(int lo, int hi)
{
#if MIHEVC_GPU
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
#else
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
#endif
}

// in-place 8x8 Hadamard SATD of a difference block held in registers: (sum |H d H| + 2) >> 2
DEV int hadamard8_satd(int (&m)[8][8])
{
#pragma unroll
    for (int y = 0; y < 8; y++) {
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[y][i], q = m[y][i + st]; m[y][i] = p + q; m[y][i + st] = p - q; }
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) {
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[i][x], q = m[i + st][x]; m[i][x] = p + q; m[i + st][x] = p - q; }
#pragma unroll
        for (int i = 0; i < 8; i++) s += iabs(m[i][x]);
    }
    return (s + 2) >> 2;
}

// the same transform of a source tile with the DC term left out of the sum: its activity around its own mean
DEV int hadamard8_ac(int (&m)[8][8])
{
#pragma unroll
    for (int y = 0; y < 8; y++) {
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[y][i], q = m[y][i + st]; m[y][i] = p + q; m[y][i + st] = p - q; }
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) {
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & st)) { int p = m[i][x], q = m[i + st][x]; m[i][x] = p + q; m[i + st][x] = p - q; }
#pragma unroll
        for (int i = 0; i < 8; i++) if (x || i) s += iabs(m[i][x]);
    }
    return (s + 2) >> 2;
}

// ------------------------------------------------------------------------------------------ executors
#if MIHEVC_GPU
#ifdef MIHEVC_PHASE_PROF
__device__ unsigned long long g_phase_prof[8 * 1024 * 3];
#endif
struct GpuExec {
    // The lane id reaches every phase through an opaque asm: without it LLVM hoists each phase's lane-index arithmetic to the
    // kernel entry and keeps it all live across the whole CTU program (k_intra_diag: 88 VGPR spill stores at entry, 92 MB of
    // scratch writes per launch, profiles/r01_c_bench); recomputing a few integer ops per phase is far cheaper.
    static DEV int lane_id()
    {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    }
#ifdef MIHEVC_PHASE_PROF
    // diagnostic build only (libmihevc_prof.so, tools/phase_profile.py): cycles of wave 0 per call site — its own work and the whole phase
    // including the barrier — summed into a table indexed by the source line of the ex.phase() call
    template <class F> DEV void phase(F &&f, int line = __builtin_LINE(), const char *file = __builtin_FILE())
    {
        // slot = (file id, line): the id comes from three characters of the header's name (inter.h 6, intra.h 7, residual.h 2, loopfilter.h 0, device.hip 5)
        int len = 0;
        while (file[len]) len++;
        const int fid = ((int)file[len - 3] + 2 * (int)file[len - 4] + 3 * (int)file[len - 6]) & 7;
        const int slot = (fid * 1024 + (line & 1023)) * 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        f(lane_id());
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_phase_prof[slot], t2 - t0);
            atomicAdd(&g_phase_prof[slot + 1], t1 - t0);
            atomicAdd(&g_phase_prof[slot + 2], 1ull);
        }
    }
#else
    template <class F> DEV void phase(F &&f)
    {
        f(lane_id());
        __syncthreads();
    }
#endif
    // A step whose producers and consumers all sit in ONE wave (lanes tid < 64): no workgroup barrier, only wave-scope
    // ordering — LDS operations of a wave execute in issue order, the fence keeps the compiler from moving them across.
    // The caller closes the sequence of wave steps with a full phase() before other waves look at the results.
    template <class F> DEV void wave_step(F &&f)
    {
        f(lane_id());
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    DEV void atomic_add(int *p, int v) { atomicAdd(p, v); }
    DEV void atomic_add(unsigned *p, unsigned v) { atomicAdd(p, v); }
    DEV void atomic_add(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
    DEV void atomic_or(unsigned *p, unsigned v) { atomicOr(p, v); }
    DEV void atomic_and(unsigned *p, unsigned v) { atomicAnd(p, v); }
    DEV void atomic_min(unsigned long long *p, unsigned long long v) { atomicMin(p, v); }
    // a plain read of a word other lanes update with atomic_min: only good for skipping an atomic that could not win
    DEV unsigned long long peek(const unsigned long long *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
    DEV void atomic_add_global(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
};
#endif
struct SeqExec {      // sequential stepping of a phase program (tests/emu)
    // order 0: threads 0..NT-1; 1: NT-1..0; 2: a fresh pseudo-random permutation per phase.  A program without races inside a phase
    // gives the same result in every order (tests/test_kernel_source_stepped.py).
    int order = 0;
    unsigned long long rnd = 0x2545F4914F6CDD1Dull;
    template <class F> void run(F &&f)
    {
        if (order == 0) { for (int t = 0; t < NT; t++) f(t); return; }
        if (order == 1) { for (int t = NT - 1; t >= 0; t--) f(t); return; }
        int perm[NT];
        for (int i = 0; i < NT; i++) perm[i] = i;
        for (int i = NT - 1; i > 0; i--) {
            rnd ^= rnd << 13; rnd ^= rnd >> 7; rnd ^= rnd << 17;
            int j = (int)((rnd >> 33) % (unsigned)(i + 1)), tmp = perm[i];
            perm[i] = perm[j]; perm[j] = tmp;
        }
        for (int i = 0; i < NT; i++) f(perm[i]);
    }
    template <class F> void phase(F &&f) { run(f); }
    template <class F> void wave_step(F &&f) { run(f); }
    void atomic_add(int *p, int v) { *p += v; }
    void atomic_add(unsigned *p, unsigned v) { *p += v; }
    void atomic_add(unsigned long long *p, unsigned long long v) { *p += v; }
    void atomic_or(unsigned *p, unsigned v) { *p |= v; }
    void atomic_and(unsigned *p, unsigned v) { *p &= v; }
    void atomic_min(unsigned long long *p, unsigned long long v) { if (v < *p) *p = v; }
    unsigned long long peek(const unsigned long long *p) { return *p; }
    void atomic_add_global(unsigned long long *p, unsigned long long v) { *p += v; }
};

}  // namespace mihevc
