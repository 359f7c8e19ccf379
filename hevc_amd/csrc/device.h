// hevc_amd/csrc/device.h — launch layer over the gfx950 kernels (hevc_amd/csrc/kernels/*.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "kernels/common.h"
#include "kernels/inter.h"
#include "kernels/intra.h"
#include "kernels/loopfilter.h"

namespace mihevc {

// a device sample plane with an edge border; `pl.p` addresses sample (0,0)
template <typename T> struct DevPlane {
    T *base = nullptr;
    Plane<T> pl{nullptr, 0};
    int w = 0, h = 0, pad = 0;
    size_t bytes() const { return (size_t)pl.stride * (h + 2 * pad) * sizeof(T); }
};
template <typename T> hipError_t alloc_plane(DevPlane<T> &d, int w, int h, int pad);
template <typename T> void free_plane(DevPlane<T> &d);

// Per-picture argument blocks live in device memory (one per picture in flight); launches take an array of them
// and index it with blockIdx.y, so one launch covers every picture of a lock-step batch.
// list 0 / 1: against InterArgs::ref / ref1 (B pictures: the anchor after the picture)
template <typename T> hipError_t launch_me_search(hipStream_t st, const InterArgs<T> *d_args, int n_ctu, int batch, int me_range, int list = 0);
// B pictures: list-0 tree and refinement, list-1 refinement, bi-prediction trial, residual
template <typename T> hipError_t launch_inter_ctu_b(hipStream_t st, const InterArgs<T> *d_args, int n_ctu, int batch, int me_range);
template <typename T> hipError_t launch_inter_ctu(hipStream_t st, const InterArgs<T> *d_args, int n_ctu, int batch, int me_range);
// all anti-diagonals of an I picture batch; h_args is the host copy (geometry only), d_args the device array
// after_plan (optional): recorded between stage A (k_intra_plan, throughput-bound) and the anti-diagonal chain of stage B (latency-bound: other streams' work fits beside it)
// flow (optional, IntraFlow::order non-null): stage B as ONE launch in which every CTU program waits for the two CTUs its prediction depends on (k_intra_flow) instead of
// one launch per anti-diagonal
struct IntraFlowSlot { int cx, cy, dep0, dep1; };      // a CTU and the raster indices of the CTUs to wait for (-1: none)
struct IntraFlow {
    const IntraFlowSlot *order = nullptr;              // device: the picture's CTUs, anti-diagonal (x + 2y inside the tile) major: a CTU's dependencies come earlier
    int *flags = nullptr;                              // device: [batch][n_ctu], a CTU's word holds `gen` once it is coded (any older value: not yet)
    int *err = nullptr;                                // device: set when a wait gave up (never seen: the bound keeps a bug from hanging the device)
    int gen = 0;
};
// host: the slot table of a picture's tile grid
void build_intra_flow_order(int ctus_w, int ctus_h, int tile_cols, int tile_rows, IntraFlowSlot *out);
template <typename T> hipError_t launch_intra_picture(hipStream_t st, const IntraArgs<T> *d_args, int ctus_w, int ctus_h, int batch, int tile_cols, int tile_rows, hipEvent_t after_plan,
                                                      const IntraFlow &flow = IntraFlow());
template <typename T> hipError_t launch_intra_p(hipStream_t st, const IntraArgs<T> *d_args, int n_ctu, int batch);
// a whole chunk: d_args[i] = picture i in stream order; its 1/4-size SOURCE picture (lsrc) and its search centres from lsrc against lref (the predecessor's lsrc)
template <typename T> hipError_t launch_pre_search_chunk(hipStream_t st, const PreArgs<T> *d_args, int w, int h, int n_ctu, int n_pictures);
// with_lowres false: the 1/4-size pictures were made by launch_prep_p_step
template <typename T> hipError_t launch_pre_search(hipStream_t st, const PreArgs<T> *d_args, int w, int h, int n_ctu, int batch, bool with_lowres);
template <typename T> hipError_t launch_deblock(hipStream_t st, const DeblockArgs<T> *d_args_v, const DeblockArgs<T> *d_args_h, int w, int h, int batch);
template <typename T> hipError_t launch_sao(hipStream_t st, const SaoArgs<T> *d_args, int w, int h, int batch, bool decide);
template <typename T> hipError_t launch_pad(hipStream_t st, const SaoArgs<T> *d_args, int w, int h, int batch);
// per-picture sum of squared error into args.sse[0..2] (u64, accumulated with atomics: zero the targets first)
template <typename T> hipError_t launch_frame_sse(hipStream_t st, const SaoArgs<T> *d_args, int batch);
template <typename T> hipError_t launch_sse_fold(hipStream_t st, const SaoArgs<T> *d_args, int n_ctu, int batch);
constexpr int MAX_LANES = 16;
struct StepParams { CostParams prm[MAX_LANES]; int p_tile_cols, p_tile_rows; };      // one P step's cost parameters per lane, passed by value; the P pictures' tile grid (intra second pass: availability)
template <typename T> hipError_t launch_begin_p_step(hipStream_t st, IntraArgs<T> *ia, InterArgs<T> *ea, SaoArgs<T> *sa, const StepParams &p, int batch);
// head of a P step in one launch: border pad of the pictures `prev` describes (nullptr: none), 1/4-size pictures for `pre` (nullptr: none), then launch_begin_p_step's work
template <typename T> hipError_t launch_prep_p_step(hipStream_t st, const SaoArgs<T> *prev, const PreArgs<T> *pre, IntraArgs<T> *ia, InterArgs<T> *ea, SaoArgs<T> *sa,
                                                    const StepParams &p, int w, int h, int batch);

// scene-cut detector input: luma plane of one source picture
template <typename T> struct ScenePic { const T *p; int stride; };
template <typename T> hipError_t launch_scene_diff(hipStream_t st, const ScenePic<T> *pics, unsigned long long *out, int w, int h, int n);

template <typename T> hipError_t launch_extend_margin(hipStream_t st, Plane<T> p, int sw, int sh, int pw, int ph);

// rows between the slices of one picture (csrc/slice_group.h): jobs[i] for i < n_jobs, one grid row of `blocks_per_job` workgroups each
hipError_t launch_copy_rows(hipStream_t st, const RowCopy *d_jobs, int n_jobs, int blocks_per_job);

int gfx950_device_count();

}  // namespace mihevc
