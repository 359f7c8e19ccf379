// hevc_amd/csrc/device.hip — __global__ entry points, launchers and the per-stage C-ABI functions (mihevc_k_*).
#include "device.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace mihevc {

__host__ __device__ static inline size_t round16(size_t v) { return (v + 15) & ~(size_t)15; }

// XCD-aware block -> CTU map: blocks b and b+8 share an XCD (and its L2), so give each XCD one contiguous run of
// CTUs; neighbouring CTUs overlap in their search windows (MI355X_MICROARCH.md, workgroup dispatch).
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    int chunk = (n + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// list 1: the search of a B picture against the anchor after it (InterArgs::ref1 / centers1 / me1)
template <typename T> __global__ __launch_bounds__(NT, 5) void k_me_search(const InterArgs<T> *args, int n_ctu, int list)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int ctu = xcd_remap(blockIdx.x, n_ctu);
    if (ctu >= n_ctu) return;
    MeShared<T> &s = *reinterpret_cast<MeShared<T> *>(smem);
    uint8_t *win = smem + round16(sizeof(MeShared<T>));
    GpuExec ex;
    if (list) {
        const InterArgs<T> a = list1_view(args[blockIdx.y]);
        me_search_program<T>(ex, s, win, a, ctu);
    } else me_search_program<T>(ex, s, win, args[blockIdx.y], ctu);
}

// 4 workgroups per CU (97 / 116 VGPRs for 8 / 10 bit and no scratch since the fractional search runs two lanes per tile; LDS allows 4 at
// 8 bit, 3 at 10 bit)
template <typename T> __global__ __launch_bounds__(NT, 4) void k_inter_ctu(const InterArgs<T> *args, int n_ctu)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int ctu = xcd_remap(blockIdx.x, n_ctu);
    if (ctu >= n_ctu) return;
    const InterArgs<T> &a = args[blockIdx.y];
    const int R = a.prm.me_range;
    InterShared<T> &s = *reinterpret_cast<InterShared<T> *>(smem);
    size_t off = round16(sizeof(InterShared<T>));
    T *wy = reinterpret_cast<T *>(smem + off);
    off += round16(((size_t)mc_win_y(R) * mc_win_y_stride(R) + 16) * sizeof(T));
    T *wu = reinterpret_cast<T *>(smem + off);
    off += round16(((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16) * sizeof(T));
    T *wv = reinterpret_cast<T *>(smem + off);
    GpuExec ex;
    inter_ctu_program<T>(ex, s, wy, wu, wv, a, ctu);
}

// B pictures: the same CTU program with the list-1 refinement and the bi-prediction trial (BiShared sits behind the windows); 3 workgroups per CU
template <typename T> __global__ __launch_bounds__(NT, 3) void k_inter_ctu_b(const InterArgs<T> *args, int n_ctu)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int ctu = xcd_remap(blockIdx.x, n_ctu);
    if (ctu >= n_ctu) return;
    const InterArgs<T> &a = args[blockIdx.y];
    const int R = a.prm.me_range;
    InterShared<T> &s = *reinterpret_cast<InterShared<T> *>(smem);
    size_t off = round16(sizeof(InterShared<T>));
    T *wy = reinterpret_cast<T *>(smem + off);
    off += round16(((size_t)mc_win_y(R) * mc_win_y_stride(R) + 16) * sizeof(T));
    T *wu = reinterpret_cast<T *>(smem + off);
    off += round16(((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16) * sizeof(T));
    T *wv = reinterpret_cast<T *>(smem + off);
    off += round16(((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16) * sizeof(T));
    BiShared *b = reinterpret_cast<BiShared *>(smem + off);
    GpuExec ex;
    inter_ctu_program<T, GpuExec, true>(ex, s, wy, wu, wv, a, ctu, b);
}

// stage A of the intra pictures: every CTU of every picture in flight plans its quadtree and modes on the source picture (kernels/intra.h);
// a throughput kernel like k_inter_ctu: 3 workgroups per CU fit its 50 KB of LDS (8 bit)
#ifndef INTRA_OCC
#define INTRA_OCC 3
#endif
template <typename T> __global__ __launch_bounds__(NT, (sizeof(T) == 1 ? INTRA_OCC : 2)) void k_intra_plan(const IntraArgs<T> *args, int n_ctu)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int ctu = xcd_remap(blockIdx.x, n_ctu);
    if (ctu >= n_ctu) return;
    const IntraArgs<T> &a = args[blockIdx.y];
    IntraShared<T> &s = *reinterpret_cast<IntraShared<T> *>(smem);
    GpuExec ex;
    intra_plan_program<T>(ex, s, a, ctu % a.ctus_w, ctu / a.ctus_w);
}

// stage B: blockIdx.x = tile * rows_per_tile + row inside the tile: every tile row holds at most one CTU of a diagonal.  A launch holds
// few hundred CTU programs (pictures x tiles x rows), each a chain of barrier phases: latency bound, so the planned CUs are all it runs
template <typename T> __global__ __launch_bounds__(NT, 2) void k_intra_diag(const IntraArgs<T> *args, int diagonal, int rows_per_tile)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const IntraArgs<T> &a = args[blockIdx.y];
    const int tcn = a.prm.tile_cols > 1 ? a.prm.tile_cols : 1, trn = a.prm.tile_rows > 1 ? a.prm.tile_rows : 1;
    const int tile = (int)blockIdx.x / rows_per_tile, r = (int)blockIdx.x % rows_per_tile;
    if (tile >= tcn * trn) return;
    const int tx = tile % tcn, ty = tile / tcn;
    const int cx0 = tile_bd(tx, tcn, a.ctus_w), cx1 = tile_bd(tx + 1, tcn, a.ctus_w), cy0 = tile_bd(ty, trn, a.ctus_h), cy1 = tile_bd(ty + 1, trn, a.ctus_h);
    const int cy = cy0 + r, cx = cx0 + diagonal - 2 * r;
    if (cy >= cy1 || cx < cx0 || cx >= cx1) return;
    IntraShared<T> &s = *reinterpret_cast<IntraShared<T> *>(smem);
    GpuExec ex;
    intra_code_program<T>(ex, s, a, cx, cy, true);
}

// Stage B of an I picture in ONE launch.  Inside a tile a CTU predicts from its left, above-left, above and above-right neighbours; coded along anti-diagonals x + 2y
// the two of them on the diagonal before (left, above-right; above when there is no above-right) finish last, so waiting for those two is waiting for all four.
// Workgroup ids run through the CTUs in anti-diagonal order (IntraFlow::order; the lanes of a batch interleaved): everything a workgroup waits for has a LOWER id, and
// the dispatcher hands out ids in order, so the lowest unfinished id is always resident and never waits: no deadlock for any number of resident workgroups.  A
// workgroup publishes its CTU with a release store at device scope after all of its waves fenced their stores (reconstruction, CU records: the neighbours' reads);
// the waiting side polls with relaxed device-scope loads and fences once it saw the word.  With one launch per anti-diagonal (k_intra_diag) every step of the chain
// cost the SLOWEST CTU of the diagonal plus a launch boundary: 24 x 130 us per IDR step at 1080p whatever the content.
// OPT-IN (MIHEVC_INTRA_FLOW): safe for one such kernel on the device, not for several at once (session.cpp ensure_flow).
template <typename T> __global__ __launch_bounds__(NT, 2) void k_intra_flow(const IntraArgs<T> *args, int lanes, int n_ctu, IntraFlow f)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int id = (int)blockIdx.x, lane = id % lanes, slot = id / lanes;
    if (slot >= n_ctu) return;
    const IntraArgs<T> &a = args[lane];
    const IntraFlowSlot o = f.order[slot];
    int *fl = f.flags + (size_t)lane * n_ctu;
    if (threadIdx.x < 2) {
        const int dep = threadIdx.x == 0 ? o.dep0 : o.dep1;
        if (dep >= 0) {
            int spins = 0;
            while (__hip_atomic_load(fl + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != f.gen) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 21)) { atomicExch(f.err, 1); break; }      // ~1 s: a bug, not a wait
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // every wave: nothing it reads from here on may come from before the neighbours' release
    IntraShared<T> &s = *reinterpret_cast<IntraShared<T> *>(smem);
    GpuExec ex;
    intra_code_program<T>(ex, s, a, o.cx, o.cy, true);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // every wave's stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(fl + o.cy * a.ctus_w + o.cx, f.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> __global__ __launch_bounds__(256) void k_lowres(const PreArgs<T> *args)
{
    lowres_sample<T>(args[blockIdx.y], (int)(blockIdx.x * 256 + threadIdx.x));
}
template <typename T> __global__ __launch_bounds__(NT) void k_pre_search(const PreArgs<T> *args, int n_ctu)
{
    __shared__ __align__(16) PreShared s;
    if ((int)blockIdx.x >= n_ctu) return;
    GpuExec ex;
    pre_search_program<T>(ex, s, args[blockIdx.y], (int)blockIdx.x);
}

// intra second pass of P pictures: one workgroup per CTU, most of them leave at once (not a candidate of this round)
template <typename T> __global__ __launch_bounds__(NT, 2) void k_intra_p(const IntraArgs<T> *args, int n_ctu, int round)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const IntraArgs<T> &a = args[blockIdx.y];
    const int ctu = (int)blockIdx.x;
    if (ctu >= n_ctu || !a.ip) return;
    const int cx = ctu % a.ctus_w, cy = ctu / a.ctus_w;
    if (!ip_eligible(a.ip, a.ctus_w, a.ctus_h, cx, cy, round)) return;
    IntraShared<T> &s = *reinterpret_cast<IntraShared<T> *>(smem);
    GpuExec ex;
    intra_ctu_program<T>(ex, s, a, cx, cy);
}

template <typename T> __global__ __launch_bounds__(256) void k_deblock(const DeblockArgs<T> *args)
{
    deblock_segment<T>(args[blockIdx.y], blockIdx.x * 256 + threadIdx.x);
}

// 8 workgroups per CU = every wave slot (49 / 55 VGPRs, 17 / 21 KB of LDS at 8 / 10 bit: the 10-bit form fits 7)
template <typename T> __global__ __launch_bounds__(NT, (sizeof(T) == 1 ? 8 : 7)) void k_sao_decide(const SaoArgs<T> *args, int n_ctu)
{
    __shared__ SaoShared<T> s;
    const int ctu = xcd_remap(blockIdx.x, n_ctu);
    if (ctu >= n_ctu) return;
    const SaoArgs<T> &a = args[blockIdx.y];
    GpuExec ex;
    sao_ctu_program<T>(ex, s, a, ctu);
}

// one thread per four samples of a row; rows of a plane are walked by consecutive lanes (coalesced dword / qword accesses)
template <typename T> __global__ __launch_bounds__(256) void k_sao_apply(const SaoArgs<T> *args)
{
    const SaoArgs<T> &a = args[blockIdx.y];
    const int ql = (a.w * a.h) >> 2, qc = ql >> 2, qwl = a.w >> 2, qwc = a.w >> 3;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ql) { sao_apply_quad<T>(a, 0, (i % qwl) * 4, i / qwl); return; }
    i -= ql;
    if (i < qc) { sao_apply_quad<T>(a, 1, (i % qwc) * 4, i / qwc); return; }
    i -= qc;
    if (i < qc) sao_apply_quad<T>(a, 2, (i % qwc) * 4, i / qwc);
}

template <typename T> __global__ __launch_bounds__(256) void k_pad(const SaoArgs<T> *args)
{
    const SaoArgs<T> &a = args[blockIdx.y];
    const int ny = pad_border_quads(a.w, a.h, PAD_Y), ncp = pad_border_quads(a.w >> 1, a.h >> 1, PAD_C);      // border only, four samples a lane
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ny) { pad_border_quad<T>(a.out[0], a.w, a.h, PAD_Y, i); return; }
    i -= ny;
    if (i < ncp) { pad_border_quad<T>(a.out[1], a.w >> 1, a.h >> 1, PAD_C, i); return; }
    i -= ncp;
    if (i < ncp) pad_border_quad<T>(a.out[2], a.w >> 1, a.h >> 1, PAD_C, i);
}

// fill the coded-size margin of a source plane (columns sw..pw-1, rows sh..ph-1) by edge replication
template <typename T> __global__ __launch_bounds__(256) void k_extend_margin(Plane<T> p, int sw, int sh, int pw, int ph)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pw * ph) return;
    int x = i % pw, y = i / pw;
    if (x < sw && y < sh) return;
    p.p[(ptrdiff_t)y * p.stride + x] = p.p[(ptrdiff_t)(y < sh ? y : sh - 1) * p.stride + (x < sw ? x : sw - 1)];
}
template <typename T> hipError_t launch_extend_margin(hipStream_t st, Plane<T> p, int sw, int sh, int pw, int ph)
{
    hipLaunchKernelGGL(k_extend_margin<T>, dim3((unsigned)((pw * ph + 255) / 256)), dim3(256), 0, st, p, sw, sh, pw, ph);
    return hipGetLastError();
}

// scene-cut detector: sum of |a - b| over every 4th sample of every 4th row of the luma planes of pictures blockIdx.y - 1... the pair
// (blockIdx.y, blockIdx.y + 1) -> out[blockIdx.y + 1]; wave reduction by shuffles, one atomic per wave
template <typename T> __global__ __launch_bounds__(256) void k_scene_diff(const ScenePic<T> *pics, unsigned long long *out, int w, int h)
{
    const ScenePic<T> a = pics[blockIdx.y], b = pics[blockIdx.y + 1];
    const int qw = (w + 3) >> 2, nq = qw * ((h + 3) >> 2);
    unsigned acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const int x = (i % qw) * 4, y = (i / qw) * 4;
        const int d = (int)a.p[(ptrdiff_t)y * a.stride + x] - (int)b.p[(ptrdiff_t)y * b.stride + x];
        acc += (unsigned)(d < 0 ? -d : d);
    }
    unsigned long long v = acc;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out + blockIdx.y + 1, v);
}
template <typename T> hipError_t launch_scene_diff(hipStream_t st, const ScenePic<T> *pics, unsigned long long *out, int w, int h, int n)
{
    if (n < 2) return hipSuccess;
    hipLaunchKernelGGL(k_scene_diff<T>, dim3(32, (unsigned)(n - 1)), dim3(256), 0, st, pics, out, w, h);
    return hipGetLastError();
}

// sum of squared error between source and final reconstruction, per plane (encoder PSNR statistics)
template <typename T> __global__ __launch_bounds__(256) void k_frame_sse(const SaoArgs<T> *args)
{
    const SaoArgs<T> &a = args[blockIdx.y];
    // four samples of a row per lane and iteration (coded widths are multiples of 8, chroma of 4): dword / qword loads, one division per quad
    const int ql = (a.w * a.h) >> 2, qc = ql >> 2, nq = ql + 2 * qc;
    unsigned long long acc[3] = {0, 0, 0};
#pragma unroll 4
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const int pl = i < ql ? 0 : i < ql + qc ? 1 : 2, k = pl == 0 ? i : pl == 1 ? i - ql : i - ql - qc, qw = (pl ? a.w >> 1 : a.w) >> 2;
        const int x = (k % qw) * 4, y = k / qw;
        T s4[4], o4[4];
        __builtin_memcpy(s4, a.src[pl].p + (ptrdiff_t)y * a.src[pl].stride + x, sizeof s4);
        __builtin_memcpy(o4, a.out[pl].p + (ptrdiff_t)y * a.out[pl].stride + x, sizeof o4);
        unsigned e = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int d = (int)s4[j] - (int)o4[j]; e += (unsigned)(d * d); }
        acc[pl] += e;
    }
    __shared__ unsigned long long red[3];
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    __syncthreads();
    for (int pl = 0; pl < 3; pl++) {
        unsigned long long v = acc[pl];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&red[pl], v);
    }
    __syncthreads();
    if (threadIdx.x < 3 && red[threadIdx.x]) atomicAdd(a.sse + threadIdx.x, red[threadIdx.x]);
}

// the per-CTU squared errors the SAO programs left (SaoArgs::sse_ctu) -> the picture's three sums: one workgroup per picture, plain stores
template <typename T> __global__ __launch_bounds__(256) void k_sse_fold(const SaoArgs<T> *args, int n_ctu)
{
    const SaoArgs<T> &a = args[blockIdx.x];
    unsigned long long acc[3] = {0, 0, 0};
    for (int i = (int)threadIdx.x; i < n_ctu; i += 256)
        for (int pl = 0; pl < 3; pl++) acc[pl] += a.sse_ctu[3 * i + pl];
    __shared__ unsigned long long red[3][4];
    for (int pl = 0; pl < 3; pl++) {
        unsigned long long v = acc[pl];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) red[pl][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) a.sse[threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
template <typename T> hipError_t launch_sse_fold(hipStream_t st, const SaoArgs<T> *d_args, int n_ctu, int batch)
{
    hipLaunchKernelGGL(k_sse_fold<T>, dim3((unsigned)batch), dim3(256), 0, st, d_args, n_ctu);
    return hipGetLastError();
}

// Start of a P step, one tiny launch for every lane: the step's cost parameters (QP from the rate controller, by value in the kernel
// arguments) go into the lane's argument blocks — the rest of the blocks was uploaded with the chunk — and the per-picture accumulators
// (SSE per plane + the rate estimate behind them: four 64-bit words at SaoArgs::sse) are zeroed.
template <typename T> __global__ __launch_bounds__(64) void k_begin_p_step(IntraArgs<T> *ia, InterArgs<T> *ea, SaoArgs<T> *sa, StepParams p)
{
    const int g = (int)blockIdx.x;
    if (threadIdx.x == 0) {
        CostParams c = p.prm[g];
        ea[g].prm = c; sa[g].prm = c;
        c.tile_cols = p.p_tile_cols; c.tile_rows = p.p_tile_rows;      // P pictures use PPS 0 (one tile, or cfg.p_tiles' grid)
        ia[g].prm = c;
    }
    if (threadIdx.x < 4) sa[g].sse[threadIdx.x] = 0;
}

// Head of a P step in ONE launch (three tiny kernels before: every launch boundary on the compute stream costs ~6 us, a step had ten): the border pad of
// the picture the previous step finished (only the next picture's searches read the border), the 1/4-size pictures of this step's source and reference,
// and the step's cost parameters.  blockIdx.x selects the job, blockIdx.y the lane.
template <typename T> __global__ __launch_bounds__(256) void k_prep_p_step(const SaoArgs<T> *prev, int n_pad_blocks, const PreArgs<T> *pre, int n_low_blocks,
                                                                           IntraArgs<T> *ia, InterArgs<T> *ea, SaoArgs<T> *sa, StepParams p)
{
    int b = (int)blockIdx.x;
    const int g = (int)blockIdx.y;
    if (b < n_pad_blocks) {
        const SaoArgs<T> &a = prev[g];
        const int ny = pad_border_quads(a.w, a.h, PAD_Y), ncp = pad_border_quads(a.w >> 1, a.h >> 1, PAD_C);
        int i = b * 256 + (int)threadIdx.x;
        if (i < ny) { pad_border_quad<T>(a.out[0], a.w, a.h, PAD_Y, i, a.halo_top, a.halo_bottom); return; }
        i -= ny;
        if (i < ncp) { pad_border_quad<T>(a.out[1], a.w >> 1, a.h >> 1, PAD_C, i, a.halo_top >> 1, a.halo_bottom >> 1); return; }
        i -= ncp;
        if (i < ncp) pad_border_quad<T>(a.out[2], a.w >> 1, a.h >> 1, PAD_C, i, a.halo_top >> 1, a.halo_bottom >> 1);
        return;
    }
    b -= n_pad_blocks;
    if (b < n_low_blocks) { lowres_sample<T>(pre[g], b * 256 + (int)threadIdx.x); return; }
    if (threadIdx.x == 0) {
        CostParams c = p.prm[g];
        ea[g].prm = c; sa[g].prm = c;
        c.tile_cols = p.p_tile_cols; c.tile_rows = p.p_tile_rows;      // P pictures use PPS 0 (one tile, or cfg.p_tiles' grid)
        ia[g].prm = c;
    }
    if (threadIdx.x < 4) sa[g].sse[threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------ launchers
template <typename T> hipError_t launch_prep_p_step(hipStream_t st, const SaoArgs<T> *prev, const PreArgs<T> *pre, IntraArgs<T> *ia, InterArgs<T> *ea, SaoArgs<T> *sa,
                                                    const StepParams &p, int w, int h, int batch)
{
    if (batch > MAX_LANES) return hipErrorInvalidValue;
    const int n_pad = prev ? (pad_border_quads(w, h, PAD_Y) + 2 * pad_border_quads(w >> 1, h >> 1, PAD_C) + 255) / 256 : 0;
    const int n_low = pre ? (2 * (w >> 2) * (h >> 2) + 255) / 256 : 0;
    hipLaunchKernelGGL(k_prep_p_step<T>, dim3((unsigned)(n_pad + n_low + 1), (unsigned)batch), dim3(256), 0, st, prev, n_pad, pre, n_low, ia, ea, sa, p);
    return hipGetLastError();
}
template <typename T> hipError_t launch_begin_p_step(hipStream_t st, IntraArgs<T> *ia, InterArgs<T> *ea, SaoArgs<T> *sa, const StepParams &p, int batch)
{
    if (batch > MAX_LANES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_begin_p_step<T>, dim3((unsigned)batch), dim3(64), 0, st, ia, ea, sa, p);
    return hipGetLastError();
}
template <typename T> hipError_t launch_frame_sse(hipStream_t st, const SaoArgs<T> *d_args, int batch)
{
    hipLaunchKernelGGL(k_frame_sse<T>, dim3(256, (unsigned)batch), dim3(256), 0, st, d_args);      // 3 same-address atomics per block: few, fat blocks
    return hipGetLastError();
}
template <typename K> static hipError_t ensure_smem(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T> hipError_t launch_me_search(hipStream_t st, const InterArgs<T> *d_args, int n_ctu, int batch, int R, int list)
{
    size_t smem = round16(sizeof(MeShared<T>)) + round16((size_t)me_win_elems(R));      // the window holds 8-bit samples for every T
    hipError_t e = ensure_smem(k_me_search<T>, smem);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((n_ctu + 7) >> 3) << 3), (unsigned)batch);
    hipLaunchKernelGGL(k_me_search<T>, grid, dim3(NT), smem, st, d_args, n_ctu, list);
    return hipGetLastError();
}

template <typename T> hipError_t launch_inter_ctu(hipStream_t st, const InterArgs<T> *d_args, int n_ctu, int batch, int R)
{
    size_t smem = round16(sizeof(InterShared<T>)) + round16(((size_t)mc_win_y(R) * mc_win_y_stride(R) + 16) * sizeof(T)) +
                  2 * round16(((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16) * sizeof(T));
    hipError_t e = ensure_smem(k_inter_ctu<T>, smem);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((n_ctu + 7) >> 3) << 3), (unsigned)batch);
    hipLaunchKernelGGL(k_inter_ctu<T>, grid, dim3(NT), smem, st, d_args, n_ctu);
    return hipGetLastError();
}

template <typename T> hipError_t launch_inter_ctu_b(hipStream_t st, const InterArgs<T> *d_args, int n_ctu, int batch, int R)
{
    size_t smem = round16(sizeof(InterShared<T>)) + round16(((size_t)mc_win_y(R) * mc_win_y_stride(R) + 16) * sizeof(T)) +
                  2 * round16(((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16) * sizeof(T)) + round16(sizeof(BiShared));
    hipError_t e = ensure_smem(k_inter_ctu_b<T>, smem);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((n_ctu + 7) >> 3) << 3), (unsigned)batch);
    hipLaunchKernelGGL(k_inter_ctu_b<T>, grid, dim3(NT), smem, st, d_args, n_ctu);
    return hipGetLastError();
}

void build_intra_flow_order(int ctus_w, int ctus_h, int tile_cols, int tile_rows, IntraFlowSlot *out)
{
    if (tile_cols < 1) tile_cols = 1;
    if (tile_rows < 1) tile_rows = 1;
    const int colw = (ctus_w + tile_cols - 1) / tile_cols, rowh = (ctus_h + tile_rows - 1) / tile_rows;
    int n = 0;
    for (int d = 0; d <= (colw - 1) + 2 * (rowh - 1); d++)
        for (int tile = 0; tile < tile_cols * tile_rows; tile++) {
            const int tx = tile % tile_cols, ty = tile / tile_cols;
            const int cx0 = tile_bd(tx, tile_cols, ctus_w), cx1 = tile_bd(tx + 1, tile_cols, ctus_w), cy0 = tile_bd(ty, tile_rows, ctus_h), cy1 = tile_bd(ty + 1, tile_rows, ctus_h);
            for (int r = 0; r < cy1 - cy0; r++) {
                const int cy = cy0 + r, cx = cx0 + d - 2 * r;
                if (cx < cx0 || cx >= cx1) continue;
                IntraFlowSlot s{cx, cy, -1, -1};
                if (cx > cx0) s.dep0 = cy * ctus_w + cx - 1;
                if (cy > cy0) s.dep1 = (cy - 1) * ctus_w + (cx + 1 < cx1 ? cx + 1 : cx);
                out[n++] = s;
            }
        }
}

template <typename T> hipError_t launch_intra_picture(hipStream_t st, const IntraArgs<T> *d_args, int ctus_w, int ctus_h, int batch, int tile_cols, int tile_rows, hipEvent_t after_plan,
                                                      const IntraFlow &flow)
{
    size_t smem = round16(sizeof(IntraShared<T>));
    hipError_t e = ensure_smem(k_intra_plan<T>, smem);
    if (e != hipSuccess) return e;
    e = ensure_smem(k_intra_diag<T>, smem);
    if (e != hipSuccess) return e;
    e = ensure_smem(k_intra_flow<T>, smem);
    if (e != hipSuccess) return e;
    if (tile_cols < 1) tile_cols = 1;
    if (tile_rows < 1) tile_rows = 1;
    const int n_ctu = ctus_w * ctus_h;
    hipLaunchKernelGGL(k_intra_plan<T>, dim3((unsigned)(((n_ctu + 7) >> 3) << 3), (unsigned)batch), dim3(NT), smem, st, d_args, n_ctu);      // stage A: every CTU at once
    if (after_plan) { e = hipEventRecord(after_plan, st); if (e != hipSuccess) return e; }      // from here on the stream runs the latency-bound anti-diagonal chain
    if (flow.order) {      // stage B as one dataflow launch
        hipLaunchKernelGGL(k_intra_flow<T>, dim3((unsigned)(n_ctu * batch)), dim3(NT), smem, st, d_args, batch, n_ctu, flow);
        return hipGetLastError();
    }
    // stage B, per tile one anti-diagonal at a time; uniform spacing: the widest column / tallest row is ceil(n_ctb / n_tiles)
    const int colw = (ctus_w + tile_cols - 1) / tile_cols, rowh = (ctus_h + tile_rows - 1) / tile_rows;
    for (int d = 0; d <= (colw - 1) + 2 * (rowh - 1); d++)
        hipLaunchKernelGGL(k_intra_diag<T>, dim3((unsigned)(tile_cols * tile_rows * rowh), (unsigned)batch), dim3(NT), smem, st, d_args, d, rowh);
    return hipGetLastError();
}

template <typename T> hipError_t launch_pre_search(hipStream_t st, const PreArgs<T> *d_args, int w, int h, int n_ctu, int batch, bool with_lowres)
{
    const int n = 2 * (w >> 2) * (h >> 2);
    if (with_lowres) hipLaunchKernelGGL(k_lowres<T>, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, st, d_args);      // else: k_prep_p_step made the pictures
    hipLaunchKernelGGL(k_pre_search<T>, dim3((unsigned)n_ctu, (unsigned)batch), dim3(NT), 0, st, d_args, n_ctu);
    return hipGetLastError();
}

// every picture of a chunk at once (args[i]: picture i in stream order): 1/4-size pictures of the SOURCES, then the search centres of picture i from
// lsrc (its own) against lref (its predecessor's) — PreArgs::ref is not read
template <typename T> __global__ __launch_bounds__(256) void k_lowres_src(const PreArgs<T> *args)
{
    const PreArgs<T> &a = args[blockIdx.y];
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < (a.w >> 2) * (a.h >> 2)) lowres_sample<T>(a, i);
}
template <typename T> hipError_t launch_pre_search_chunk(hipStream_t st, const PreArgs<T> *d_args, int w, int h, int n_ctu, int n_pictures)
{
    if (n_pictures <= 0) return hipSuccess;
    const int n = (w >> 2) * (h >> 2);
    hipLaunchKernelGGL(k_lowres_src<T>, dim3((unsigned)((n + 255) / 256), (unsigned)n_pictures), dim3(256), 0, st, d_args);
    hipLaunchKernelGGL(k_pre_search<T>, dim3((unsigned)n_ctu, (unsigned)n_pictures), dim3(NT), 0, st, d_args, n_ctu);
    return hipGetLastError();
}

template <typename T> hipError_t launch_intra_p(hipStream_t st, const IntraArgs<T> *d_args, int n_ctu, int batch)
{
    size_t smem = round16(sizeof(IntraShared<T>));
    hipError_t e = ensure_smem(k_intra_p<T>, smem);
    if (e != hipSuccess) return e;
    for (int round = 0; round < 2; round++)
        hipLaunchKernelGGL(k_intra_p<T>, dim3((unsigned)n_ctu, (unsigned)batch), dim3(NT), smem, st, d_args, n_ctu, round);
    return hipGetLastError();
}

template <typename T> hipError_t launch_deblock(hipStream_t st, const DeblockArgs<T> *d_v, const DeblockArgs<T> *d_h, int w, int h, int batch)
{
    int segs = (w >> 3) * (h >> 3) * 2;
    dim3 grid((unsigned)((segs + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL(k_deblock<T>, grid, dim3(256), 0, st, d_v);
    hipLaunchKernelGGL(k_deblock<T>, grid, dim3(256), 0, st, d_h);
    return hipGetLastError();
}

template <typename T> hipError_t launch_sao(hipStream_t st, const SaoArgs<T> *d_args, int w, int h, int batch, bool decide)
{
    int n_ctu = ((w + CTU - 1) / CTU) * ((h + CTU - 1) / CTU);
    if (decide) {        // the CTU program decides AND applies (its deblocked tile is in LDS): no second pass
        hipLaunchKernelGGL(k_sao_decide<T>, dim3((unsigned)(((n_ctu + 7) >> 3) << 3), (unsigned)batch), dim3(NT), 0, st, d_args, n_ctu);
        return hipGetLastError();
    }
    int n = (w * h + (w * h >> 1)) >> 2;
    hipLaunchKernelGGL(k_sao_apply<T>, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, st, d_args);
    return hipGetLastError();
}

template <typename T> hipError_t launch_pad(hipStream_t st, const SaoArgs<T> *d_args, int w, int h, int batch)
{
    int n = pad_border_quads(w, h, PAD_Y) + 2 * pad_border_quads(w >> 1, h >> 1, PAD_C);
    hipLaunchKernelGGL(k_pad<T>, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, st, d_args);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_copy_rows(const RowCopy *jobs)
{
    copy_rows_item(jobs[blockIdx.y], (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}
hipError_t launch_copy_rows(hipStream_t st, const RowCopy *d_jobs, int n_jobs, int blocks_per_job)
{
    if (n_jobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)blocks_per_job, (unsigned)n_jobs), dim3(256), 0, st, d_jobs);
    return hipGetLastError();
}

template <typename T> hipError_t alloc_plane(DevPlane<T> &d, int w, int h, int pad)
{
    d.w = w; d.h = h; d.pad = pad;
    d.pl.stride = (w + 2 * pad + 63) & ~63;
    hipError_t e = hipMalloc((void **)&d.base, (size_t)d.pl.stride * (h + 2 * pad) * sizeof(T));
    if (e != hipSuccess) { d.base = nullptr; return e; }
    d.pl.p = d.base + (size_t)pad * d.pl.stride + pad;
    return hipSuccess;
}
template <typename T> void free_plane(DevPlane<T> &d)
{
    if (d.base) (void)hipFree(d.base);
    d.base = nullptr; d.pl.p = nullptr;
}

int gfx950_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

#define INSTANTIATE(T)                                                                                                   \
    template hipError_t launch_me_search<T>(hipStream_t, const InterArgs<T> *, int, int, int, int);                     \
    template hipError_t launch_inter_ctu_b<T>(hipStream_t, const InterArgs<T> *, int, int, int);                        \
    template hipError_t launch_inter_ctu<T>(hipStream_t, const InterArgs<T> *, int, int, int);                          \
    template hipError_t launch_intra_picture<T>(hipStream_t, const IntraArgs<T> *, int, int, int, int, int, hipEvent_t, const IntraFlow &);     \
    template hipError_t launch_intra_p<T>(hipStream_t, const IntraArgs<T> *, int, int);                                       \
    template hipError_t launch_pre_search<T>(hipStream_t, const PreArgs<T> *, int, int, int, int, bool);                \
    template hipError_t launch_pre_search_chunk<T>(hipStream_t, const PreArgs<T> *, int, int, int, int);                \
    template hipError_t launch_prep_p_step<T>(hipStream_t, const SaoArgs<T> *, const PreArgs<T> *, IntraArgs<T> *, InterArgs<T> *, SaoArgs<T> *, const StepParams &, int, int, int); \
    template hipError_t launch_deblock<T>(hipStream_t, const DeblockArgs<T> *, const DeblockArgs<T> *, int, int, int);  \
    template hipError_t launch_sao<T>(hipStream_t, const SaoArgs<T> *, int, int, int, bool);                            \
    template hipError_t launch_pad<T>(hipStream_t, const SaoArgs<T> *, int, int, int);                                  \
    template hipError_t launch_frame_sse<T>(hipStream_t, const SaoArgs<T> *, int);                                       \
    template hipError_t launch_sse_fold<T>(hipStream_t, const SaoArgs<T> *, int, int);                                   \
    template hipError_t launch_begin_p_step<T>(hipStream_t, IntraArgs<T> *, InterArgs<T> *, SaoArgs<T> *, const StepParams &, int); \
    template hipError_t launch_extend_margin<T>(hipStream_t, Plane<T>, int, int, int, int);                             \
    template hipError_t launch_scene_diff<T>(hipStream_t, const ScenePic<T> *, unsigned long long *, int, int, int);            \
    template hipError_t alloc_plane<T>(DevPlane<T> &, int, int, int);                                                   \
    template void free_plane<T>(DevPlane<T> &);
INSTANTIATE(uint8_t)
INSTANTIATE(uint16_t)

// ------------------------------------------------------------------------------------------ K3 alone (parity/bench entry)
// One workgroup per batch of residual blocks laid out as a pseudo-CTU: the blocks of one launch share log2n.
__global__ __launch_bounds__(NT) void k_transform_blocks(const int16_t *res, int16_t *lvl, int16_t *rec, int n_blocks, int log2n, int qp,
                                                         int bit_depth, int intra)
{
    __shared__ ResidualShared s;
    GpuExec ex;
    const int n = 1 << log2n, per = 1024 >> (2 * log2n);          // luma-area blocks per workgroup
    const int first = blockIdx.x * per;
    residual_init(ex, s);
    ex.phase([&](int tid) {
        if (tid < 16) {
            int tx = (tid & 3) * 8, ty = (tid >> 2) * 8;
            int blk = first + (ty >> (log2n < 3 ? 3 : log2n)) * (32 >> (log2n < 3 ? 3 : log2n)) + (tx >> (log2n < 3 ? 3 : log2n));
            s.tu_log2[tid] = (log2n >= 3 && blk < n_blocks) ? (uint8_t)log2n : 0;
            s.tu_intra[tid] = (uint8_t)intra;
        }
        for (int i = tid; i < 1536; i += NT) s.res[i] = 0;
    });
    ex.phase([&](int tid) {
        for (int i = tid; i < 1024; i += NT) {
            int x = i & 31, y = i >> 5, blk = first + (y >> log2n) * (32 >> log2n) + (x >> log2n);
            if (blk < n_blocks) s.res[i] = res[(size_t)blk * n * n + (y & (n - 1)) * n + (x & (n - 1))];
        }
        for (int i = tid; i < 1536; i += NT) s.desc[i] = pack_loc(locate(s, i));
    });
    residual_pipeline(ex, s, qp, qp, bit_depth, whole_ctu());
    ex.phase([&](int tid) {
        for (int i = tid; i < 1024; i += NT) {
            int x = i & 31, y = i >> 5, blk = first + (y >> log2n) * (32 >> log2n) + (x >> log2n);
            if (blk >= n_blocks) continue;
            size_t o = (size_t)blk * n * n + (y & (n - 1)) * n + (x & (n - 1));
            lvl[o] = s.lvl[i];
            rec[o] = s.res[i];
        }
    });
}

// 4x4 TUs (DCT, or DST-VII for intra luma): 16 lanes per block, four blocks per wave, the same step sequence and arithmetic as the
// NxN trial of k_intra_diag (kernels/intra.h code_blocks): rows, columns + quantisation, scaling + inverse columns, inverse rows.
__global__ __launch_bounds__(NT) void k_transform4_blocks(const int16_t *res, int16_t *lvl, int16_t *rec, int n_blocks, int qp, int bit_depth, int intra, int dst)
{
    __shared__ int16_t M[16], sres[NT], slvl[NT];
    __shared__ int tmp[NT];
    __shared__ unsigned nz[NT / 16];
    const int tid = (int)threadIdx.x, g = tid >> 4, i = tid & 15, blk = (int)blockIdx.x * (NT / 16) + g;
    if (tid < 16) M[tid] = dst ? g_tab.dst4[tid >> 2][tid & 3] : g_tab.mat[(tid >> 2) * 8][tid & 3];
    if (i == 0) nz[g] = 0;
    sres[tid] = blk < n_blocks ? res[(size_t)blk * 16 + i] : (int16_t)0;
    __syncthreads();
    const int bd = bit_depth, q = qp + 6 * (bd - 8), qbits = 14 + q / 6 + (15 - bd - 2), bsh = bd + 2 - 5, s1 = bd - 7, s3 = 20 - bd;
    const int16_t *r = sres + g * 16;
    int *t = tmp + g * 16;
    int16_t *l = slvl + g * 16;
    {   const int u = i & 3, y = i >> 2;
        int acc = 0;
        for (int x = 0; x < 4; x++) acc += M[u * 4 + x] * r[y * 4 + x];
        t[i] = (acc + (1 << (s1 - 1))) >> s1; }
    __syncthreads();
    {   const int u = i & 3, v = i >> 2;
        int acc = 0;
        for (int y = 0; y < 4; y++) acc += M[v * 4 + y] * t[y * 4 + u];
        const int c = clip3(-32768, 32767, (acc + 128) >> 8);
        long long a = ((long long)iabs(c) * g_tab.quant_scale[q % 6] + ((long long)(intra ? 171 : 85) << (qbits - 9))) >> qbits;
        if (a > 32767) a = 32767;
        l[i] = (int16_t)(c < 0 ? -(int)a : (int)a);
        if (a) atomicOr(&nz[g], 1u); }
    __syncthreads();
    {   const int x = i & 3, y = i >> 2;
        const long long scale = (long long)16 * g_tab.level_scale[q % 6] << (q / 6);
        int acc = 0;
        for (int j = 0; j < 4; j++) acc += M[j * 4 + y] * clip3(-32768, 32767, (int)((l[j * 4 + x] * scale + ((long long)1 << (bsh - 1))) >> bsh));
        __syncthreads();
        t[i] = clip3(-32768, 32767, (acc + 64) >> 7); }
    __syncthreads();
    {   const int x = i & 3, y = i >> 2;
        int acc = 0;
        for (int j = 0; j < 4; j++) acc += M[j * 4 + x] * t[y * 4 + j];
        if (blk < n_blocks) {
            lvl[(size_t)blk * 16 + i] = l[i];
            rec[(size_t)blk * 16 + i] = nz[g] ? (int16_t)((acc + (1 << (s3 - 1))) >> s3) : (int16_t)0;
        } }
}

}  // namespace mihevc

// ================================================================================================ C ABI: stages
using namespace mihevc;

namespace {

struct DevBuf {      // RAII device allocation
    void *p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename U> U *as() const { return static_cast<U *>(p); }
};
#define CK(expr)                                     \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess) return MIHEVC_EDEVICE; \
    } while (0)

template <typename T> struct Planes3 {
    DevPlane<T> p[3];
    int alloc(int w, int h, bool padded)
    {
        for (int i = 0; i < 3; i++)
            if (alloc_plane<T>(p[i], i ? w / 2 : w, i ? h / 2 : h, padded ? (i ? PAD_C : PAD_Y) : 0) != hipSuccess) return MIHEVC_ENOMEM;
        return 0;
    }
    int upload(const void *y, const void *u, const void *v)
    {
        const void *src[3] = {y, u, v};
        for (int i = 0; i < 3; i++)
            CK(hipMemcpy2D(p[i].pl.p, p[i].pl.stride * sizeof(T), src[i], p[i].w * sizeof(T), p[i].w * sizeof(T), p[i].h, hipMemcpyHostToDevice));
        return 0;
    }
    int download(void *y, void *u, void *v)
    {
        void *dst[3] = {y, u, v};
        for (int i = 0; i < 3; i++)
            CK(hipMemcpy2D(dst[i], p[i].w * sizeof(T), p[i].pl.p, p[i].pl.stride * sizeof(T), p[i].w * sizeof(T), p[i].h, hipMemcpyDeviceToHost));
        return 0;
    }
    ~Planes3() { for (auto &x : p) free_plane<T>(x); }
};

CostParams to_prm(const mihevc_cost_params *p) { return CostParams{p->qp, p->qp_c, p->bit_depth, p->lambda_sad_q4, p->lambda_q4, p->me_range, p->tile_cols, p->tile_rows, p->intra_nxn, p->intra_in_p, p->pre_search, p->rdo_zero, p->chroma_modes, p->mc_top, p->mc_bottom, p->rdo_cg}; }

bool geometry_ok(int w, int h) { return w >= 16 && h >= 16 && !(w & 7) && !(h & 7) && w <= 8192 && h <= 4352; }

template <typename T>
int stage_intra(const void *sy, const void *su, const void *sv, int w, int h, const mihevc_cost_params *prm, void *ry, void *ru, void *rv,
                mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_, int16_t *cv, uint64_t *est)
{
    Planes3<T> src, rec;
    if (src.alloc(w, h, false) || rec.alloc(w, h, false)) return MIHEVC_ENOMEM;
    if (int e = src.upload(sy, su, sv)) return e;
    const size_t n8 = (size_t)(w / 8) * (h / 8), ny = (size_t)w * h;
    DevBuf dcu, dc0, dc1, dc2, dargs, dest, dplan;
    CK(dcu.alloc(n8 * sizeof(mihevc_cu_rec))); CK(dc0.alloc(ny * 2)); CK(dc1.alloc(ny / 2)); CK(dc2.alloc(ny / 2)); CK(dargs.alloc(sizeof(IntraArgs<T>)));
    CK(dplan.alloc((size_t)((w + CTU - 1) / CTU) * ((h + CTU - 1) / CTU) * sizeof(IntraPlan)));
    CK(dest.alloc(8)); CK(hipMemset(dest.p, 0, 8));
    CK(hipMemset(dcu.p, 0, n8 * sizeof(mihevc_cu_rec)));
    IntraArgs<T> a;
    for (int i = 0; i < 3; i++) { a.src[i] = {src.p[i].pl.p, src.p[i].pl.stride}; a.rec[i] = rec.p[i].pl; }
    a.w = w; a.h = h; a.ctus_w = (w + CTU - 1) / CTU; a.ctus_h = (h + CTU - 1) / CTU; a.prm = to_prm(prm);
    a.cu = dcu.as<mihevc_cu_rec>(); a.coef[0] = dc0.as<int16_t>(); a.coef[1] = dc1.as<int16_t>(); a.coef[2] = dc2.as<int16_t>(); a.diagonal = 0; a.est = dest.as<unsigned long long>(); a.sparse_coef = 0; a.ip = nullptr; a.plan = dplan.as<IntraPlan>();
    CK(hipMemcpy(dargs.p, &a, sizeof a, hipMemcpyHostToDevice));
    IntraFlow flow;
    DevBuf dorder, dflags;
    if (getenv("MIHEVC_INTRA_FLOW")) {        // (opt-in: stage B as one dataflow launch; default: one launch per anti-diagonal — session.cpp ensure_flow says why)
        const int n_ctu = a.ctus_w * a.ctus_h;
        std::vector<IntraFlowSlot> order((size_t)n_ctu);
        build_intra_flow_order(a.ctus_w, a.ctus_h, a.prm.tile_cols, a.prm.tile_rows, order.data());
        CK(dorder.alloc(order.size() * sizeof(IntraFlowSlot))); CK(dflags.alloc(((size_t)n_ctu + 1) * sizeof(int)));
        CK(hipMemcpy(dorder.p, order.data(), order.size() * sizeof(IntraFlowSlot), hipMemcpyHostToDevice));
        CK(hipMemset(dflags.p, 0, ((size_t)n_ctu + 1) * sizeof(int)));
        flow.order = dorder.as<IntraFlowSlot>(); flow.flags = dflags.as<int>(); flow.err = dflags.as<int>() + n_ctu; flow.gen = 1;
    }
    CK(launch_intra_picture<T>(0, dargs.as<IntraArgs<T>>(), a.ctus_w, a.ctus_h, 1, a.prm.tile_cols, a.prm.tile_rows, nullptr, flow));
    CK(hipDeviceSynchronize());
    if (flow.order) { int bad = 0; CK(hipMemcpy(&bad, flow.err, sizeof bad, hipMemcpyDeviceToHost)); if (bad) return MIHEVC_EDEVICE; }
    if (int e = rec.download(ry, ru, rv)) return e;
    CK(hipMemcpy(cu, dcu.p, n8 * sizeof(mihevc_cu_rec), hipMemcpyDeviceToHost));
    CK(hipMemcpy(cy, dc0.p, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cu_, dc1.p, ny / 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cv, dc2.p, ny / 2, hipMemcpyDeviceToHost));
    if (est) CK(hipMemcpy(est, dest.p, 8, hipMemcpyDeviceToHost));
    return MIHEVC_OK;
}

template <typename T>
int stage_inter(const void *sy, const void *su, const void *sv, const void *fy, const void *fu, const void *fv, int w, int h,
                const mihevc_cost_params *prm, const int16_t *centers, void *ry, void *ru, void *rv, mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_,
                int16_t *cv, int32_t *me_dump, uint64_t *est)
{
    Planes3<T> src, ref, rec;
    if (src.alloc(w, h, false) || ref.alloc(w, h, true) || rec.alloc(w, h, false)) return MIHEVC_ENOMEM;
    if (int e = src.upload(sy, su, sv)) return e;
    if (int e = ref.upload(fy, fu, fv)) return e;
    const int ctus_w = (w + CTU - 1) / CTU, n_ctu = ctus_w * ((h + CTU - 1) / CTU);
    const size_t n8 = (size_t)(w / 8) * (h / 8), ny = (size_t)w * h;
    DevBuf dcu, dc0, dc1, dc2, dargs, dme, dcen, dpad, dest;
    CK(dest.alloc(8)); CK(hipMemset(dest.p, 0, 8));
    CK(dcu.alloc(n8 * sizeof(mihevc_cu_rec))); CK(dc0.alloc(ny * 2)); CK(dc1.alloc(ny / 2)); CK(dc2.alloc(ny / 2));
    CK(dargs.alloc(sizeof(InterArgs<T>))); CK(dme.alloc((size_t)n_ctu * 63 * 4)); CK(dcen.alloc((size_t)n_ctu * 4)); CK(dpad.alloc(sizeof(SaoArgs<T>)));
    CK(hipMemset(dcu.p, 0, n8 * sizeof(mihevc_cu_rec)));
    if (centers) CK(hipMemcpy(dcen.p, centers, (size_t)n_ctu * 4, hipMemcpyHostToDevice));
    // border extension of the uploaded reference (the pad kernel works on the `out` planes of a SaoArgs block)
    SaoArgs<T> pa;
    memset(&pa, 0, sizeof pa);
    for (int i = 0; i < 3; i++) pa.out[i] = ref.p[i].pl;
    pa.w = w; pa.h = h;
    CK(hipMemcpy(dpad.p, &pa, sizeof pa, hipMemcpyHostToDevice));
    CK(launch_pad<T>(0, dpad.as<SaoArgs<T>>(), w, h, 1));
    InterArgs<T> a;
    for (int i = 0; i < 3; i++) { a.src[i] = {src.p[i].pl.p, src.p[i].pl.stride}; a.ref[i] = {ref.p[i].pl.p, ref.p[i].pl.stride}; a.rec[i] = rec.p[i].pl; }
    a.w = w; a.h = h; a.ctus_w = ctus_w; a.prm = to_prm(prm); a.centers = centers ? dcen.as<int16_t>() : nullptr; a.me = dme.as<int32_t>();
    a.cu = dcu.as<mihevc_cu_rec>(); a.coef[0] = dc0.as<int16_t>(); a.coef[1] = dc1.as<int16_t>(); a.coef[2] = dc2.as<int16_t>();
    a.est = dest.as<unsigned long long>(); a.sparse_coef = 0;
    for (int i = 0; i < 3; i++) a.ref1[i] = {nullptr, 0};
    a.centers1 = nullptr; a.me1 = nullptr;
    DevBuf dip, diargs;
    a.ip = nullptr;
    if (a.prm.intra_in_p) { CK(dip.alloc((size_t)n_ctu * sizeof(IpInfo))); CK(hipMemset(dip.p, 0, (size_t)n_ctu * sizeof(IpInfo))); a.ip = dip.as<IpInfo>(); }
    DevBuf dls, dlr, dpre;
    if (a.prm.pre_search && !centers) {       // search centres from the 1/4-size pictures
        const size_t ln = (size_t)(w >> 2) * (h >> 2);
        CK(dls.alloc(ln)); CK(dlr.alloc(ln)); CK(dpre.alloc(sizeof(PreArgs<T>)));
        PreArgs<T> pa2;
        pa2.src = a.src[0]; pa2.ref = a.ref[0]; pa2.lsrc = dls.as<uint8_t>(); pa2.lref = dlr.as<uint8_t>(); pa2.w = w; pa2.h = h; pa2.bit_depth = a.prm.bit_depth; pa2.centers = dcen.as<int16_t>(); pa2.cost = nullptr;
        CK(hipMemcpy(dpre.p, &pa2, sizeof pa2, hipMemcpyHostToDevice));
        CK(launch_pre_search<T>(0, dpre.as<PreArgs<T>>(), w, h, n_ctu, 1, true));
        a.centers = dcen.as<int16_t>();
    }
    CK(hipMemcpy(dargs.p, &a, sizeof a, hipMemcpyHostToDevice));
    CK(launch_me_search<T>(0, dargs.as<InterArgs<T>>(), n_ctu, 1, a.prm.me_range, 0));
    CK(launch_inter_ctu<T>(0, dargs.as<InterArgs<T>>(), n_ctu, 1, a.prm.me_range));
    if (a.prm.intra_in_p) {       // intra second pass on the same reconstruction / records / levels
        IntraArgs<T> ia;
        for (int i = 0; i < 3; i++) { ia.src[i] = a.src[i]; ia.rec[i] = a.rec[i]; ia.coef[i] = a.coef[i]; }
        ia.w = w; ia.h = h; ia.ctus_w = ctus_w; ia.ctus_h = (h + CTU - 1) / CTU;
        ia.prm = a.prm;        // tile_cols / tile_rows: the P pictures' own grid (1x1 unless the caller passes cfg.p_tiles' grid)
        ia.cu = a.cu; ia.diagonal = 0; ia.est = a.est; ia.sparse_coef = 0; ia.ip = a.ip; ia.plan = nullptr;
        CK(diargs.alloc(sizeof ia));
        CK(hipMemcpy(diargs.p, &ia, sizeof ia, hipMemcpyHostToDevice));
        CK(launch_intra_p<T>(0, diargs.as<IntraArgs<T>>(), n_ctu, 1));
    }
    CK(hipDeviceSynchronize());
    if (int e = rec.download(ry, ru, rv)) return e;
    CK(hipMemcpy(cu, dcu.p, n8 * sizeof(mihevc_cu_rec), hipMemcpyDeviceToHost));
    CK(hipMemcpy(cy, dc0.p, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cu_, dc1.p, ny / 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cv, dc2.p, ny / 2, hipMemcpyDeviceToHost));
    if (me_dump) CK(hipMemcpy(me_dump, dme.p, (size_t)n_ctu * 63 * 4, hipMemcpyDeviceToHost));
    if (est) CK(hipMemcpy(est, dest.p, 8, hipMemcpyDeviceToHost));
    return MIHEVC_OK;
}

// B picture between two anchors: both integer searches, then the B form of the CTU program
template <typename T>
int stage_b(const void *sy, const void *su, const void *sv, const void *f0y, const void *f0u, const void *f0v, const void *f1y, const void *f1u, const void *f1v, int w, int h,
            const mihevc_cost_params *prm, const int16_t *centers0, const int16_t *centers1, void *ry, void *ru, void *rv, mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_,
            int16_t *cv, int32_t *me_dump0, int32_t *me_dump1, uint64_t *est)
{
    Planes3<T> src, ref0, ref1, rec;
    if (src.alloc(w, h, false) || ref0.alloc(w, h, true) || ref1.alloc(w, h, true) || rec.alloc(w, h, false)) return MIHEVC_ENOMEM;
    if (int e = src.upload(sy, su, sv)) return e;
    if (int e = ref0.upload(f0y, f0u, f0v)) return e;
    if (int e = ref1.upload(f1y, f1u, f1v)) return e;
    const int ctus_w = (w + CTU - 1) / CTU, n_ctu = ctus_w * ((h + CTU - 1) / CTU);
    const size_t n8 = (size_t)(w / 8) * (h / 8), ny = (size_t)w * h;
    DevBuf dcu, dc0, dc1, dc2, dargs, dme0, dme1, dcen0, dcen1, dpad, dest;
    CK(dest.alloc(8)); CK(hipMemset(dest.p, 0, 8));
    CK(dcu.alloc(n8 * sizeof(mihevc_cu_rec))); CK(dc0.alloc(ny * 2)); CK(dc1.alloc(ny / 2)); CK(dc2.alloc(ny / 2));
    CK(dargs.alloc(sizeof(InterArgs<T>))); CK(dme0.alloc((size_t)n_ctu * 63 * 4)); CK(dme1.alloc((size_t)n_ctu * 63 * 4)); CK(dcen0.alloc((size_t)n_ctu * 4)); CK(dcen1.alloc((size_t)n_ctu * 4));
    CK(dpad.alloc(2 * sizeof(SaoArgs<T>)));
    CK(hipMemset(dcu.p, 0, n8 * sizeof(mihevc_cu_rec)));
    if (centers0) CK(hipMemcpy(dcen0.p, centers0, (size_t)n_ctu * 4, hipMemcpyHostToDevice));
    if (centers1) CK(hipMemcpy(dcen1.p, centers1, (size_t)n_ctu * 4, hipMemcpyHostToDevice));
    SaoArgs<T> pa[2];
    memset(pa, 0, sizeof pa);
    for (int i = 0; i < 3; i++) { pa[0].out[i] = ref0.p[i].pl; pa[1].out[i] = ref1.p[i].pl; }
    pa[0].w = pa[1].w = w; pa[0].h = pa[1].h = h;
    CK(hipMemcpy(dpad.p, pa, sizeof pa, hipMemcpyHostToDevice));
    CK(launch_pad<T>(0, dpad.as<SaoArgs<T>>(), w, h, 2));
    InterArgs<T> a;
    for (int i = 0; i < 3; i++) {
        a.src[i] = {src.p[i].pl.p, src.p[i].pl.stride}; a.ref[i] = {ref0.p[i].pl.p, ref0.p[i].pl.stride}; a.ref1[i] = {ref1.p[i].pl.p, ref1.p[i].pl.stride}; a.rec[i] = rec.p[i].pl;
    }
    a.w = w; a.h = h; a.ctus_w = ctus_w; a.prm = to_prm(prm);
    a.centers = centers0 ? dcen0.as<int16_t>() : nullptr; a.centers1 = centers1 ? dcen1.as<int16_t>() : nullptr;
    a.me = dme0.as<int32_t>(); a.me1 = dme1.as<int32_t>();
    a.cu = dcu.as<mihevc_cu_rec>(); a.coef[0] = dc0.as<int16_t>(); a.coef[1] = dc1.as<int16_t>(); a.coef[2] = dc2.as<int16_t>();
    a.est = dest.as<unsigned long long>(); a.sparse_coef = 0; a.ip = nullptr;
    CK(hipMemcpy(dargs.p, &a, sizeof a, hipMemcpyHostToDevice));
    CK(launch_me_search<T>(0, dargs.as<InterArgs<T>>(), n_ctu, 1, a.prm.me_range, 0));
    CK(launch_me_search<T>(0, dargs.as<InterArgs<T>>(), n_ctu, 1, a.prm.me_range, 1));
    CK(launch_inter_ctu_b<T>(0, dargs.as<InterArgs<T>>(), n_ctu, 1, a.prm.me_range));
    CK(hipDeviceSynchronize());
    if (int e = rec.download(ry, ru, rv)) return e;
    CK(hipMemcpy(cu, dcu.p, n8 * sizeof(mihevc_cu_rec), hipMemcpyDeviceToHost));
    CK(hipMemcpy(cy, dc0.p, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cu_, dc1.p, ny / 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cv, dc2.p, ny / 2, hipMemcpyDeviceToHost));
    if (me_dump0) CK(hipMemcpy(me_dump0, dme0.p, (size_t)n_ctu * 63 * 4, hipMemcpyDeviceToHost));
    if (me_dump1) CK(hipMemcpy(me_dump1, dme1.p, (size_t)n_ctu * 63 * 4, hipMemcpyDeviceToHost));
    if (est) CK(hipMemcpy(est, dest.p, 8, hipMemcpyDeviceToHost));
    return MIHEVC_OK;
}

template <typename T> int stage_deblock(void *ry, void *ru, void *rv, int w, int h, const mihevc_cu_rec *cu, int bit_depth)
{
    Planes3<T> rec;
    if (rec.alloc(w, h, false)) return MIHEVC_ENOMEM;
    if (int e = rec.upload(ry, ru, rv)) return e;
    const size_t n8 = (size_t)(w / 8) * (h / 8);
    DevBuf dcu, dargs;
    CK(dcu.alloc(n8 * sizeof(mihevc_cu_rec))); CK(dargs.alloc(2 * sizeof(DeblockArgs<T>)));
    CK(hipMemcpy(dcu.p, cu, n8 * sizeof(mihevc_cu_rec), hipMemcpyHostToDevice));
    DeblockArgs<T> a[2];
    for (int d = 0; d < 2; d++) {
        for (int i = 0; i < 3; i++) a[d].rec[i] = rec.p[i].pl;
        a[d].w = w; a[d].h = h; a[d].cu = dcu.as<mihevc_cu_rec>(); a[d].bit_depth = bit_depth; a[d].dir = d; a[d].y_org = 0;
    }
    CK(hipMemcpy(dargs.p, a, sizeof a, hipMemcpyHostToDevice));
    CK(launch_deblock<T>(0, dargs.as<DeblockArgs<T>>(), dargs.as<DeblockArgs<T>>() + 1, w, h, 1));
    CK(hipDeviceSynchronize());
    return rec.download(ry, ru, rv);
}

template <typename T>
int stage_sao(const void *sy, const void *su, const void *sv, const void *dy, const void *du, const void *dv, int w, int h,
              const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *sao, const mihevc_cu_rec *cu = nullptr)
{
    Planes3<T> src, dbk, out;
    if (src.alloc(w, h, false) || dbk.alloc(w, h, false) || out.alloc(w, h, true)) return MIHEVC_ENOMEM;
    if (int e = src.upload(sy, su, sv)) return e;
    if (int e = dbk.upload(dy, du, dv)) return e;
    const int ctus_w = (w + CTU - 1) / CTU, n_ctu = ctus_w * ((h + CTU - 1) / CTU);
    DevBuf dsao, dargs, dcu;
    CK(dsao.alloc((size_t)n_ctu * sizeof(mihevc_sao_ctu))); CK(dargs.alloc(sizeof(SaoArgs<T>)));
    SaoArgs<T> a;
    for (int i = 0; i < 3; i++) { a.src[i] = {src.p[i].pl.p, src.p[i].pl.stride}; a.dbk[i] = {dbk.p[i].pl.p, dbk.p[i].pl.stride}; a.out[i] = out.p[i].pl; }
    a.w = w; a.h = h; a.ctus_w = ctus_w; a.prm = to_prm(prm); a.sao = dsao.as<mihevc_sao_ctu>(); a.sse = nullptr; a.sse_ctu = nullptr; a.halo_top = a.halo_bottom = 0;
    a.cu = nullptr;
    if (cu) {      // the fused loop filter: `dbk` is the pre-deblock reconstruction
        const size_t n8 = (size_t)(w / 8) * (h / 8);
        CK(dcu.alloc(n8 * sizeof(mihevc_cu_rec)));
        CK(hipMemcpy(dcu.p, cu, n8 * sizeof(mihevc_cu_rec), hipMemcpyHostToDevice));
        a.cu = dcu.as<mihevc_cu_rec>();
    }
    CK(hipMemcpy(dargs.p, &a, sizeof a, hipMemcpyHostToDevice));
    CK(launch_sao<T>(0, dargs.as<SaoArgs<T>>(), w, h, 1, true));
    CK(launch_pad<T>(0, dargs.as<SaoArgs<T>>(), w, h, 1));
    CK(hipDeviceSynchronize());
    if (int e = out.download(oy, ou, ov)) return e;
    CK(hipMemcpy(sao, dsao.p, (size_t)n_ctu * sizeof(mihevc_sao_ctu), hipMemcpyDeviceToHost));
    return MIHEVC_OK;
}

int select_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MIHEVC_ENODEV;
    if (device < 0 || device >= n) return MIHEVC_EINVAL;
    return hipSetDevice(device) == hipSuccess ? 0 : MIHEVC_EDEVICE;
}

}  // namespace

extern "C" {

int mihevc_device_count(void) { return gfx950_device_count(); }
int mihevc_device_numa_node(int device)
{
    char bus[64] = {0};
    if (device < 0 || device >= gfx950_device_count() || hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return -1;
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');      // sysfs names are lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

int mihevc_k_transform(int device, const int16_t *residual, int16_t *levels, int16_t *recon_residual, int n_blocks, int log2n, int qp,
                       int bit_depth, int intra, int dst4)
{
    if (!residual || !levels || !recon_residual || n_blocks <= 0 || log2n < 2 || log2n > 5 || (dst4 && log2n != 2)) return MIHEVC_EINVAL;
    if (bit_depth != 8 && bit_depth != 10) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    const size_t bytes = (size_t)n_blocks << (2 * log2n + 1);
    DevBuf dres, dlvl, drec;
    CK(dres.alloc(bytes)); CK(dlvl.alloc(bytes)); CK(drec.alloc(bytes));
    CK(hipMemcpy(dres.p, residual, bytes, hipMemcpyHostToDevice));
    if (log2n == 2) {     // 4-point DCT, or DST-VII (intra luma 4x4)
        hipLaunchKernelGGL(k_transform4_blocks, dim3((unsigned)((n_blocks + NT / 16 - 1) / (NT / 16))), dim3(NT), 0, 0, dres.as<int16_t>(), dlvl.as<int16_t>(),
                           drec.as<int16_t>(), n_blocks, qp, bit_depth, intra, dst4);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(levels, dlvl.p, bytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(recon_residual, drec.p, bytes, hipMemcpyDeviceToHost));
        return MIHEVC_OK;
    }
    const int per = 1024 >> (2 * log2n);
    hipLaunchKernelGGL(k_transform_blocks, dim3((unsigned)((n_blocks + per - 1) / per)), dim3(NT), 0, 0, dres.as<int16_t>(), dlvl.as<int16_t>(),
                       drec.as<int16_t>(), n_blocks, log2n, qp, bit_depth, intra);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(levels, dlvl.p, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(recon_residual, drec.p, bytes, hipMemcpyDeviceToHost));
    return MIHEVC_OK;
}

int mihevc_k_intra_frame(int device, const void *sy, const void *su, const void *sv, int w, int h, const mihevc_cost_params *prm, void *ry, void *ru,
                         void *rv, mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_, int16_t *cv, uint64_t *est)
{
    if (!sy || !su || !sv || !prm || !ry || !ru || !rv || !cu || !cy || !cu_ || !cv || !geometry_ok(w, h)) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    if (prm->bit_depth == 8) return stage_intra<uint8_t>(sy, su, sv, w, h, prm, ry, ru, rv, cu, cy, cu_, cv, est);
    if (prm->bit_depth == 10) return stage_intra<uint16_t>(sy, su, sv, w, h, prm, ry, ru, rv, cu, cy, cu_, cv, est);
    return MIHEVC_EINVAL;
}

int mihevc_k_inter_frame(int device, const void *sy, const void *su, const void *sv, const void *fy, const void *fu, const void *fv, int w, int h,
                         const mihevc_cost_params *prm, const int16_t *centers, void *ry, void *ru, void *rv, mihevc_cu_rec *cu, int16_t *cy,
                         int16_t *cu_, int16_t *cv, int32_t *me_dump, uint64_t *est)
{
    if (!sy || !su || !sv || !fy || !fu || !fv || !prm || !ry || !ru || !rv || !cu || !cy || !cu_ || !cv || !geometry_ok(w, h)) return MIHEVC_EINVAL;
    if (prm->me_range < 1 || prm->me_range > MAX_RANGE) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    if (prm->bit_depth == 8) return stage_inter<uint8_t>(sy, su, sv, fy, fu, fv, w, h, prm, centers, ry, ru, rv, cu, cy, cu_, cv, me_dump, est);
    if (prm->bit_depth == 10) return stage_inter<uint16_t>(sy, su, sv, fy, fu, fv, w, h, prm, centers, ry, ru, rv, cu, cy, cu_, cv, me_dump, est);
    return MIHEVC_EINVAL;
}

int mihevc_k_b_frame(int device, const void *sy, const void *su, const void *sv, const void *f0y, const void *f0u, const void *f0v, const void *f1y, const void *f1u,
                     const void *f1v, int w, int h, const mihevc_cost_params *prm, const int16_t *centers0, const int16_t *centers1, void *ry, void *ru, void *rv,
                     mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_, int16_t *cv, int32_t *me_dump0, int32_t *me_dump1, uint64_t *est)
{
    if (!sy || !su || !sv || !f0y || !f0u || !f0v || !f1y || !f1u || !f1v || !prm || !ry || !ru || !rv || !cu || !cy || !cu_ || !cv || !geometry_ok(w, h)) return MIHEVC_EINVAL;
    if (prm->me_range < 1 || prm->me_range > MAX_RANGE) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    if (prm->bit_depth == 8) return stage_b<uint8_t>(sy, su, sv, f0y, f0u, f0v, f1y, f1u, f1v, w, h, prm, centers0, centers1, ry, ru, rv, cu, cy, cu_, cv, me_dump0, me_dump1, est);
    if (prm->bit_depth == 10) return stage_b<uint16_t>(sy, su, sv, f0y, f0u, f0v, f1y, f1u, f1v, w, h, prm, centers0, centers1, ry, ru, rv, cu, cy, cu_, cv, me_dump0, me_dump1, est);
    return MIHEVC_EINVAL;
}

int mihevc_k_deblock(int device, void *ry, void *ru, void *rv, int w, int h, const mihevc_cu_rec *cu, int bit_depth)
{
    if (!ry || !ru || !rv || !cu || !geometry_ok(w, h)) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    if (bit_depth == 8) return stage_deblock<uint8_t>(ry, ru, rv, w, h, cu, bit_depth);
    if (bit_depth == 10) return stage_deblock<uint16_t>(ry, ru, rv, w, h, cu, bit_depth);
    return MIHEVC_EINVAL;
}

int mihevc_k_sao(int device, const void *sy, const void *su, const void *sv, const void *dy, const void *du, const void *dv, int w, int h,
                 const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *sao)
{
    if (!sy || !su || !sv || !dy || !du || !dv || !prm || !oy || !ou || !ov || !sao || !geometry_ok(w, h)) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    if (prm->bit_depth == 8) return stage_sao<uint8_t>(sy, su, sv, dy, du, dv, w, h, prm, oy, ou, ov, sao);
    if (prm->bit_depth == 10) return stage_sao<uint16_t>(sy, su, sv, dy, du, dv, w, h, prm, oy, ou, ov, sao);
    return MIHEVC_EINVAL;
}

int mihevc_k_loop_filter(int device, const void *sy, const void *su, const void *sv, const void *ry, const void *ru, const void *rv, int w, int h,
                         const mihevc_cu_rec *cu, const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *sao)
{
    if (!sy || !su || !sv || !ry || !ru || !rv || !cu || !prm || !oy || !ou || !ov || !sao || !geometry_ok(w, h)) return MIHEVC_EINVAL;
    if (int e = select_device(device)) return e;
    if (prm->bit_depth == 8) return stage_sao<uint8_t>(sy, su, sv, ry, ru, rv, w, h, prm, oy, ou, ov, sao, cu);
    if (prm->bit_depth == 10) return stage_sao<uint16_t>(sy, su, sv, ry, ru, rv, w, h, prm, oy, ou, ov, sao, cu);
    return MIHEVC_EINVAL;
}

#ifdef MIHEVC_PHASE_PROF
// diagnostic build only: read (and optionally clear) the per-call-site cycle table of GpuExec::phase
int mihevc_debug_phase_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mihevc::g_phase_prof), sizeof(unsigned long long) * 8 * 1024 * 3) != hipSuccess) return MIHEVC_EDEVICE;
    if (reset) {
        static unsigned long long zero[8 * 1024 * 3];
        if (hipMemcpyToSymbol(HIP_SYMBOL(mihevc::g_phase_prof), zero, sizeof zero) != hipSuccess) return MIHEVC_EDEVICE;
    }
    return MIHEVC_OK;
}
#endif

}  // extern "C"
