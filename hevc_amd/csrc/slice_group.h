// hevc_amd/csrc/slice_group.h — the sessions that code the slices of ONE picture on several devices (BASELINE configs[4]) meet here.
//
// Every band (a full-width run of CTU rows) is its own session, driven by its own host thread on its own device; closed GOPs run in lock-step
// inside every session exactly as for whole pictures.  What the bands exchange per picture (a neighbour halo over xGMI, no
// collective): (X1) after the analysis, 8 rows of the pre-deblock reconstruction + one row of CU records either side of every seam, so that
// deblocking and SAO run ACROSS the seams; (X2) after SAO, the PAD_Y rows of the final reconstruction either side of every seam, which become the
// neighbour's border rows: motion vectors point across seams as if the picture were whole.  Both are pulls by the reader straight out of the
// neighbour's device memory (peer access over xGMI; an ordinary device pointer when two bands share a device), ordered by events.  The rate
// controller's inputs (estimates, CABAC sizes) are summed over the bands, so every band takes the same decisions: ONE rate plan per picture.
//
// This header holds the host-side meeting point only: what a band publishes for its neighbours, "has band k enqueued step n yet" (an event may
// only be waited on after its record call), a sum over all bands, and failure propagation (a band that fails wakes everyone).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace mihevc {

constexpr int kMaxBands = 16;
constexpr int kHaloLanes = 16;

struct BandPub {                       // what band k lets its neighbours see (device pointers are valid from every band's device: peer access)
    int device = -1;
    int w = 0, h = 0, is16 = 0;
    // per GOP lane: the padded reconstructions (ping-pong) and their pitches in samples
    void *rec_p[kHaloLanes][2][3] = {};
    int rec_stride[3] = {0, 0, 0};
    // X1 export block, two parities: per lane [top: 8 luma rows | 4 + 4 chroma rows | one row of CU records][bottom: the same]
    void *x1_export[2] = {nullptr, nullptr};
    size_t x1_lane_bytes = 0;
    hipEvent_t ev_x1[2] = {nullptr, nullptr}, ev_x2[2] = {nullptr, nullptr};
    long long x1_step = -1, x2_step = -1, pub_chunk = -1;      // guarded by SliceGroup::m
};

class SliceGroup {
public:
    explicit SliceGroup(int n) : n_(n), acc_(), res_() {}
    int size() const { return n_; }
    BandPub &pub(int k) { return pub_[k]; }

    void fail()
    {
        std::lock_guard<std::mutex> l(m_);
        failed_ = true;
        cv_.notify_all();
    }
    bool failed()
    {
        std::lock_guard<std::mutex> l(m_);
        return failed_;
    }
    // sum of v over all bands, element by element; every band calls it at the same point of its (identical) control flow.  false: the group failed.
    bool allreduce(std::vector<double> &v)
    {
        std::unique_lock<std::mutex> l(m_);
        if (failed_) return false;
        const int gen = gen_;
        if (arrived_ == 0) acc_.assign(v.size(), 0.0);
        if (acc_.size() != v.size()) { failed_ = true; cv_.notify_all(); return false; }      // bands out of step: a bug, never a hang
        for (size_t i = 0; i < v.size(); i++) acc_[i] += v[i];
        if (++arrived_ == n_) { res_ = acc_; arrived_ = 0; gen_++; cv_.notify_all(); }
        else cv_.wait(l, [&] { return gen_ != gen || failed_; });
        if (failed_) return false;
        v = res_;
        return true;
    }
    bool barrier() { std::vector<double> z(1, 0.0); return allreduce(z); }
    // band k announces that it has ENQUEUED (recorded the event of) exchange `which` of global step `step`
    void announce(int k, int which, long long step)
    {
        std::lock_guard<std::mutex> l(m_);
        (which == 1 ? pub_[k].x1_step : pub_[k].x2_step) = step;
        cv_.notify_all();
    }
    bool wait_for(int k, int which, long long step)
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return failed_ || (which == 1 ? pub_[k].x1_step : pub_[k].x2_step) >= step; });
        return !failed_;
    }

    // process-wide registry: the sessions of one picture find each other by the id their configuration carries
    static std::shared_ptr<SliceGroup> join(int64_t id, int n)
    {
        static std::mutex rm;
        static std::map<int64_t, std::weak_ptr<SliceGroup>> reg;
        std::lock_guard<std::mutex> l(rm);
        auto &w = reg[id];
        std::shared_ptr<SliceGroup> g = w.lock();
        if (!g || g->size() != n) { g = std::make_shared<SliceGroup>(n); w = g; }
        for (auto it = reg.begin(); it != reg.end();) it = it->second.expired() ? reg.erase(it) : std::next(it);
        return g;
    }

private:
    const int n_;
    std::mutex m_;
    std::condition_variable cv_;
    bool failed_ = false;
    int gen_ = 0, arrived_ = 0;
    std::vector<double> acc_, res_;
    BandPub pub_[kMaxBands];
};

}  // namespace mihevc
