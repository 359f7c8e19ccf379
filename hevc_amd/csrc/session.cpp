// hevc_amd/csrc/session.cpp — the encoder session behind mihevc_open / send_frame / receive_packet.
//
// Pipeline: source pictures are collected in HBM; every `gops_in_flight * keyint` pictures
// (or at flush) the chunk is encoded.  Closed GOPs are independent, so the chunk's GOPs run in LOCK-STEP: step t
// launches each stage once for picture t of every GOP (blockIdx.y = GOP lane).  Per step: intra anti-diagonals
// (t = 0) or ME + inter CTU (t > 0) -> deblock V/H -> SAO decide/apply -> border pad -> SSE, then one D2H copy of
// the step's symbols into pinned memory on a second stream, and one CABAC job per picture on the host pool.
// The device never waits for CABAC except when the 4-deep symbol ring wraps.
// Replaces, for the selected files, the `ffmpeg -c:v libx265` child of the reference (core/transcoder.py:506).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bitstream.h"
#include "device.h"
#include "slice_group.h"

using namespace mihevc;

namespace {

constexpr int kRing = 12;     // symbol slots per lane (capacity): slot 0 holds the IDR picture, the rest rotate over the P steps.
                              // A session uses s->ring of them: 8 up to 1080p-class levels, 12 from level 5 (2160p+), where the CABAC of one
                              // picture (12 ms at 2160p, 45 ms at 4320p) outlasts five device steps when few GOP lanes are busy
constexpr int kIdrStart = 5;  // first chunk of a session: IDR pictures under rate control are analysed at the CRF's IDR QP + 5, then re-analysed only
                              // where the rate model asks for a QP at least kIdrRedo away (round 1 analysed every IDR at three QPs: 23 % of device time)
constexpr int kIdrRedo = 2;
constexpr double kBudgetShare = 0.985;   // a GOP is planned to 98.5 % of vbv-maxrate x its duration: the estimate-to-CABAC ratio is known to ~1 %
constexpr double kCpbStart = 0.9;        // CPB fullness every closed GOP may assume at its IDR (= the buffering period SEI's initial delay)
constexpr double kIdrCpbShare = 0.85;    // an IDR picture may take at most this share of that fullness
constexpr double kCutAbs = 8.0;          // scene cut: mean absolute difference of consecutive pictures above this many grey levels (8-bit scale) ...
constexpr double kCutRatio = 1.8;        // ... and this many times the running mean over the ordinary pictures before it

// Host worker pool for the CABAC jobs.  ONE pool per process, shared by every session and grown to the largest size a session asks for: a batch
// codes many clips back to back, and starting / joining 16 threads per clip was 2 ms of every 80 ms 1080p clip (bench step_phases open + close).
// The threads live until the process exits (they are parked on a condition variable); a session waits for ITS jobs, never for the threads.
class ThreadPool {
public:
    static ThreadPool &shared(int n)
    {
        static ThreadPool *p = new ThreadPool();      // never destroyed: no join at process exit, the threads hold no session state
        p->grow(n);
        return *p;
    }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            q_.push_back(std::move(f));
        }
        cv_.notify_one();
    }

private:
    void grow(int n)
    {
        std::lock_guard<std::mutex> l(m_);
        while ((int)threads_.size() < n) { threads_.emplace_back([this] { run(); }); threads_.back().detach(); }
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return !q_.empty(); });
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_;
};

// Process-wide cache of device / pinned-host allocations keyed by (device, size): a batch transcodes many clips of
// the same geometry back to back (gui/mainwindow.py queue), and hipMalloc/hipHostMalloc/hipFree cost tens of ms
// per session otherwise (bench step_phases: close 51 ms).  Buffers return to the cache at mihevc_close.
class BufferCache {
public:
    static BufferCache &get() { static BufferCache c; return c; }
    hipError_t alloc(int dev, size_t n, bool pinned, void **out)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            auto &v = free_[key(dev, n, pinned)];
            if (!v.empty()) { *out = v.back(); v.pop_back(); bytes_ -= n; return hipSuccess; }
        }
        return pinned ? hipHostMalloc(out, n, hipHostMallocDefault) : hipMalloc(out, n);
    }
    void release(int dev, size_t n, bool pinned, void *p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> l(m_);
        if (bytes_ + n > kMaxBytes) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); return; }
        bytes_ += n;
        free_[key(dev, n, pinned)].push_back(p);
    }
private:
    static constexpr size_t kMaxBytes = (size_t)24 << 30;      // 24 GiB of 288: plenty for a few clip geometries
    static std::string key(int dev, size_t n, bool pinned) { return std::to_string(dev) + (pinned ? "h" : "d") + std::to_string(n); }
    std::mutex m_;
    std::map<std::string, std::vector<void *>> free_;
    size_t bytes_ = 0;
};

// the same for the sessions' two streams (create + destroy: about a millisecond per session)
class StreamCache {
public:
    static StreamCache &get() { static StreamCache c; return c; }
    hipError_t acquire(int dev, hipStream_t *out)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            auto &v = free_[dev];
            if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        }
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    void release(int dev, hipStream_t st)
    {
        if (!st) return;
        std::lock_guard<std::mutex> l(m_);
        auto &v = free_[dev];
        if (v.size() >= 16) { (void)hipStreamDestroy(st); return; }
        v.push_back(st);
    }
private:
    std::mutex m_;
    std::map<int, std::vector<hipStream_t>> free_;
};

struct Packet {
    std::vector<uint8_t> data;
    int64_t pts = 0, dts = 0;
    bool key = false, ready = false;
};

// symbol block of one picture: [cu | coef Y | coef U | coef V | sao | sse[3] | rate estimate]; the device twin carries the per-CTU squared errors behind it
// (SaoArgs::sse_ctu: never copied to the host, k_sse_fold turns them into sse[3])
struct SymLayout {
    size_t cu, cu_bytes, cy, cu_, cv, sao, sse, est, total, sse_ctu, dev_total;
    SymLayout(int w, int h)
    {
        size_t n8 = (size_t)(w / 8) * (h / 8), ny = (size_t)w * h, nctu = (size_t)((w + 31) / 32) * ((h + 31) / 32);
        const size_t row = (size_t)(w / 8) * sizeof(mihevc_cu_rec);
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        cu = al(row);                 // one row of records in front of the picture's and one behind: where the neighbour slices' rows go (slice_group.h)
        cu_bytes = n8 * sizeof(mihevc_cu_rec);
        cy = al(cu + cu_bytes + row);
        cu_ = al(cy + ny * 2);
        cv = al(cu_ + ny / 2);
        sao = al(cv + ny / 2);
        sse = al(sao + nctu * sizeof(mihevc_sao_ctu));
        est = sse + 3 * sizeof(unsigned long long);
        total = al(est + sizeof(unsigned long long));
        sse_ctu = total;
        dev_total = al(sse_ctu + nctu * 3 * sizeof(uint32_t));
    }
};

constexpr int kSeamRows = 8;      // rows of the pre-deblock reconstruction exchanged either side of a seam (deblocking reads 4 and writes 3; one 8x8 grid row)

}  // namespace

struct mihevc_session {
    mihevc_config cfg;
    int device = 0;
    int w = 0, h = 0, ctus_w = 0, ctus_h = 0, n_ctu = 0;      // coded size
    TileGrid tiles;                                            // IDR pictures (PPS 1); 1x1 when cfg.intra_tiles == 0
    TileGrid ptiles;                                           // P pictures (PPS 0, cfg.p_tiles); 1x1 when off
    int keyint = 90, lanes = 4, me_range = 16, qp_p = 22, qp_i = 19;
    bool is16 = false, keep_recon = false, flushed = false, failed = false, flushing = false;
    std::string err;
    hipStream_t st_compute = nullptr, st_copy = nullptr, st_pre = nullptr;      // st_pre: the chunk's pre-search, beside the IDR step
    // uploads of host frames (mihevc_send_frame / _async) go through st_pre (idle outside a chunk's IDR step; a FOURTH stream per session made two of them share a
    // hardware queue: the copy stream's SSE pass and symbol copies then queued behind the compute stream's kernels, +10 ms of bubbles per 300-frame clip); the
    // chunk's first launch waits for ev_up
    hipEvent_t ev_up = nullptr;
    bool up_pending = false;
    // source pictures of the current chunk (device), in display order
    struct Src { void *base[3]; void *p[3]; int stride[3]; int64_t pts; bool borrowed; };      // borrowed: the caller's device planes, not copied
    size_t plane_bytes[3][3] = {{0}};   // [kind: plain / padded reference / work][plane] allocation sizes (for the buffer cache)
    std::vector<Src> pending;
    std::vector<Src> free_src;
    // per lane
    struct Lane {
        void *rec_base[3][3], *rec_p[3][3]; int rec_stride[3];       // padded final reconstructions: the anchors ping-pong between 0 and 1; 2 = B pictures (cfg.bframes; never a reference)
        void *work_base[3], *work_p[3]; int work_stride[3];           // pre-deblock / deblocked picture (unpadded)
        int32_t *me = nullptr, *me1 = nullptr;      // integer-search tables (me1: list 1 of B pictures)
        IpInfo *ip = nullptr;      // per CTU: inter pass -> intra second pass of P pictures
        IntraPlan *plan = nullptr; // per CTU: k_intra_plan -> k_intra_diag (IDR pictures)
        uint8_t *sym_dev[kRing] = {nullptr}, *sym_host[kRing] = {nullptr};
    };
    std::vector<Lane> lane;
    void *d_args = nullptr;           // argument blocks of a whole chunk
    uint8_t *h_args = nullptr;        // pinned staging for the same
    size_t args_cap = 0;
    hipEvent_t ev_compute[kRing] = {}, ev_copy[kRing] = {};
    std::vector<hipEvent_t> ev_pool;   // profile_stages: start/stop pairs
    struct Mark { int stage, pictures; size_t ev; };
    std::vector<Mark> marks;
    // host side
    ThreadPool *pool = nullptr;
    std::mutex m;
    std::condition_variable cv;
    int jobs_open[kRing] = {0};
    int ring = 8;                 // slots in use (<= kRing)
    int host_threads = 2;         // CABAC worker threads this session asked the process-wide pool for
    std::map<int64_t, Packet> packets;     // by output index
    int64_t next_out = 0, frames_in = 0, frames_done = 0;
    std::vector<uint8_t> headers, cur_packet;
    std::map<int64_t, std::vector<uint16_t>> recon;   // keep_recon: final pictures by index (Y,U,V concatenated)
    mihevc_stats stats{};
    // rate control (VBV-capped constant quality, one controller per GOP lane): see DESIGN.md §Rate control
    bool rc_on = false;
    double ratio_i = 1.0, ratio_p = 1.0;      // learned (CABAC bits) / (device estimate); updated once per chunk (deterministic)
    bool rho_measured = false;                // the session's first chunk measures rho with a trial analysis of the GOPs' first P picture
    double rho_pi = 1.0 / 16.0;               // learned (P bits) / (IDR bits) at equal QP: the prior before a GOP's first P estimate lands
    void *d_flow = nullptr; size_t flow_bytes = 0; int flow_gen = 0;      // IDR pictures' stage B as one dataflow launch (device.h IntraFlow): [slot table | flags[MAX_LANES][n_ctu] | err]
    void *d_probe = nullptr; size_t probe_cap = 0;      // cfg.bframes = -1: the probe's low-resolution pictures, centres and costs
    double beta_bp = 0.45;                    // cfg.bframes: learned (B bits at QP + 2) / (P bits at QP): what a B picture takes of the GOP budget beside a P picture
    int64_t pts_step = 1, first_pts = 0;      // pts distance of the first two frames: with B pictures dts = (pts of the frame at the packet's place in decoding order) - pts_step
    int idr_qp_hint = -1;                     // mean IDR QP the last chunk settled on: where the next chunk's IDR analysis starts
    int last_gop_len = 0;                     // length of the stream's previous GOP (picture timing SEI at the next IDR)
    double scene_avg = 0;                     // running mean of the picture-to-picture difference over ordinary pictures (scene-cut detector)
    void *d_low = nullptr; size_t low_cap = 0;       // per chunk: 1/4-size SOURCE pictures of every picture, then the search centres of every picture (pre-search)
    hipEvent_t ev_pre = nullptr, ev_args = nullptr;  // the chunk's centres are ready (st_pre) / the IDR step's k_intra_plan is through (compute stream: the argument blocks are on the device too)
    void *d_scene = nullptr; size_t scene_cap = 0;   // per chunk: picture pointers / pitches in, difference sums out (k_scene_diff)
    struct FrameRec { int qp = 0, type = 0; long long bits = -1, bits_local = -1; unsigned long long est_q4 = 0, est_local = 0; bool est_known = false; };      // with slices that share one rate plan (group): bits / est_q4 are sums over the slices, bits_local this slice's
    std::vector<FrameRec> frames;             // by output index
    std::atomic<long long> entropy_ns{0};
    // ---- one slice of a picture whose slices exchange rows (cfg.slice_halo; csrc/slice_group.h)
    std::shared_ptr<SliceGroup> group;
    int band = 0, n_bands = 1;
    int band_h[kMaxBands] = {0};              // coded heights of all bands
    long long gstep = 0, chunk_no = 0;        // steps / chunks since the session was opened: the same in every band of the group
    hipEvent_t ev_x1[2] = {nullptr, nullptr}, ev_x2[2] = {nullptr, nullptr};
    void *x1_export[2] = {nullptr, nullptr};
    size_t x1_part_bytes = 0, x1_lane_bytes = 0, x1_bytes = 0;
    void *d_jobs = nullptr; uint8_t *h_jobs = nullptr; size_t jobs_cap = 0;
    std::vector<int> peers_enabled;
};

namespace {

#define HIPCK(s, expr)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (s)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            (s)->failed = true;                                                           \
            if ((s)->group) (s)->group->fail();                                           \
            return MIHEVC_EDEVICE;                                                        \
        }                                                                                 \
    } while (0)

size_t esize(const mihevc_session *s) { return s->is16 ? 2 : 1; }

// padded 0: a plain picture; 1: a reference picture with its PAD border all round; 2: a work picture with kSeamRows rows above and below (the rows the
// neighbour slices hand over for deblocking across seams; unused otherwise)
int alloc_planes(mihevc_session *s, void *base[3], void *p[3], int stride[3], int padded)
{
    for (int i = 0; i < 3; i++) {
        int w = i ? s->w / 2 : s->w, h = i ? s->h / 2 : s->h, pad = padded == 1 ? (i ? PAD_C : PAD_Y) : 0, vm = padded == 2 ? (i ? kSeamRows / 2 : kSeamRows) : pad;
        stride[i] = (w + 2 * pad + 63) & ~63;
        s->plane_bytes[padded][i] = (size_t)stride[i] * (h + 2 * vm) * esize(s);
        HIPCK(s, BufferCache::get().alloc(s->device, s->plane_bytes[padded][i], false, &base[i]));
        p[i] = (uint8_t *)base[i] + ((size_t)vm * stride[i] + pad) * esize(s);
    }
    return 0;
}

int get_src(mihevc_session *s, mihevc_session::Src &out)
{
    if (!s->free_src.empty()) { out = s->free_src.back(); s->free_src.pop_back(); return 0; }
    return alloc_planes(s, out.base, out.p, out.stride, 0);
}

int ensure_lanes(mihevc_session *s, int n)
{
    SymLayout sl(s->w, s->h);
    while ((int)s->lane.size() < n) {
        mihevc_session::Lane L;
        memset(&L, 0, sizeof L);
        for (int k = 0; k < (s->cfg.bframes != 0 ? 3 : 2); k++)
            if (int e = alloc_planes(s, L.rec_base[k], L.rec_p[k], L.rec_stride, 1)) return e;
        if (int e = alloc_planes(s, L.work_base, L.work_p, L.work_stride, 2)) return e;
        HIPCK(s, BufferCache::get().alloc(s->device, (size_t)s->n_ctu * 63 * sizeof(int32_t), false, (void **)&L.me));
        if (s->cfg.bframes != 0) HIPCK(s, BufferCache::get().alloc(s->device, (size_t)s->n_ctu * 63 * sizeof(int32_t), false, (void **)&L.me1));
        HIPCK(s, BufferCache::get().alloc(s->device, (size_t)s->n_ctu * sizeof(IpInfo), false, (void **)&L.ip));
        HIPCK(s, BufferCache::get().alloc(s->device, (size_t)s->n_ctu * sizeof(IntraPlan), false, (void **)&L.plan));
        for (int k = 0; k < s->ring; k++) {
            HIPCK(s, BufferCache::get().alloc(s->device, sl.dev_total, false, (void **)&L.sym_dev[k]));
            HIPCK(s, BufferCache::get().alloc(s->device, sl.total, true, (void **)&L.sym_host[k]));
        }
        s->lane.push_back(L);
    }
    return 0;
}

// the IDR pictures' dataflow launch (k_intra_flow), OPT-IN through MIHEVC_INTRA_FLOW: slot table and flag words, made once per session.  Buffers come from the process-wide
// cache with whatever an earlier session left in them, so the flags are zeroed here and generations count from 1.  Default: one launch per anti-diagonal (k_intra_diag).
// Why opt-in: the no-deadlock argument (a wait only points at lower workgroup ids, ids are dispatched in order) holds for ONE such kernel on the device.  Several at once
// (sessions sharing a device: a sliced picture's bands, a batch's workers) can fill an XCD's workgroup slots with each other's waiting workgroups while the one workgroup
// every chain waits for has no slot: seen once in four runs of the 4320p picture as 8 sessions on one device — the bounded wait turned it into an error instead of a hang.
// The gain (intra stage -7 %, 0.6 % of a clip) does not pay for that.
static int ensure_flow(mihevc_session *s)
{
    if (s->d_flow || !getenv("MIHEVC_INTRA_FLOW")) return 0;
    const size_t o_flags = (size_t)s->n_ctu * sizeof(IntraFlowSlot), bytes = o_flags + ((size_t)MAX_LANES * s->n_ctu + 1) * sizeof(int);
    HIPCK(s, BufferCache::get().alloc(s->device, bytes, false, &s->d_flow));
    s->flow_bytes = bytes;
    std::vector<IntraFlowSlot> order((size_t)s->n_ctu);
    build_intra_flow_order(s->ctus_w, s->ctus_h, s->tiles.cols, s->tiles.rows, order.data());
    HIPCK(s, hipMemcpyAsync(s->d_flow, order.data(), o_flags, hipMemcpyHostToDevice, s->st_compute));
    HIPCK(s, hipMemsetAsync((uint8_t *)s->d_flow + o_flags, 0, bytes - o_flags, s->st_compute));
    HIPCK(s, hipStreamSynchronize(s->st_compute));      // `order` leaves scope
    return 0;
}
static IntraFlow next_flow(mihevc_session *s)
{
    IntraFlow f;
    if (!s->d_flow) return f;
    f.order = (const IntraFlowSlot *)s->d_flow;
    f.flags = (int *)((uint8_t *)s->d_flow + (size_t)s->n_ctu * sizeof(IntraFlowSlot));
    f.err = f.flags + (size_t)MAX_LANES * s->n_ctu;
    f.gen = ++s->flow_gen;
    return f;
}

// argument blocks of one lock-step step: five arrays of `gops` entries each, so one launch per stage covers all lanes
template <typename T> struct StepLayout {
    size_t intra, inter, dbk_v, dbk_h, sao, pre, total;
    explicit StepLayout(int gops)
    {
        auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
        intra = 0;
        inter = al(intra + (size_t)2 * gops * sizeof(IntraArgs<T>));      // [gops, 2 gops): the IDR re-analysis pass, compacted
        dbk_v = al(inter + gops * sizeof(InterArgs<T>));
        dbk_h = al(dbk_v + gops * sizeof(DeblockArgs<T>));
        sao = al(dbk_h + gops * sizeof(DeblockArgs<T>));
        pre = al(sao + gops * sizeof(SaoArgs<T>));
        total = al(pre + gops * sizeof(PreArgs<T>));
    }
};
template <typename T> struct StepView {
    IntraArgs<T> *intra; InterArgs<T> *inter; DeblockArgs<T> *dbk_v, *dbk_h; SaoArgs<T> *sao; PreArgs<T> *pre;
    StepView(uint8_t *base, const StepLayout<T> &l, int t)
    {
        uint8_t *b = base + (size_t)t * l.total;
        intra = (IntraArgs<T> *)(b + l.intra); inter = (InterArgs<T> *)(b + l.inter);
        dbk_v = (DeblockArgs<T> *)(b + l.dbk_v); dbk_h = (DeblockArgs<T> *)(b + l.dbk_h); sao = (SaoArgs<T> *)(b + l.sao);
        pre = (PreArgs<T> *)(b + l.pre);
    }
};

template <typename T> Plane<T> mk(void *p, int stride) { return Plane<T>{(T *)p, stride}; }
template <typename T> Plane<const T> mkc(void *p, int stride) { return Plane<const T>{(const T *)p, stride}; }

// CABAC of one picture: `parts` host jobs share its tiles (cfg.p_tiles / the IDR grid: every tile is its own substream), the last one to finish
// puts the access unit together and publishes the packet.  One part = the whole picture in one job, as before round 3.
struct PictureJob {
    mihevc_session *s;
    int slot, lane_i, slice_type, poc, qp, prev_gop_len, parts, n_tiles, dec_pos;      // poc: place in the GOP in display order, dec_pos: in decoding order
    int64_t index, pts, dts, dec_index;      // index: display order (frame records, reconstructions); dec_index: decoding order (packets)
    bool first_of_stream, reorder;      // reorder: the stream announces B pictures (dts one frame early, output delay in the picture timing SEI)
    PictureSyms pic;
    std::vector<std::vector<uint8_t>> sub;
    std::atomic<int> left;
    std::atomic<long long> ns{0};
};

void picture_symbols(mihevc_session *s, int slot, int lane_i, PictureSyms &pic)
{
    SymLayout sl(s->w, s->h);
    const uint8_t *b = s->lane[lane_i].sym_host[slot];
    pic.cu = (const mihevc_cu_rec *)(b + sl.cu);
    pic.coef[0] = (const int16_t *)(b + sl.cy); pic.coef[1] = (const int16_t *)(b + sl.cu_); pic.coef[2] = (const int16_t *)(b + sl.cv);
    pic.sao = s->cfg.sao ? (const mihevc_sao_ctu *)(b + sl.sao) : nullptr;
}

// last part of a picture: access unit = AUD first (7.4.2.4.4), parameter sets (+ HDR10 SEI), buffering period at the IDR, picture timing, the slice
void publish_picture(PictureJob *j)
{
    mihevc_session *s = j->s;
    auto t0 = std::chrono::steady_clock::now();
    SymLayout sl(s->w, s->h);
    const uint8_t *b = s->lane[j->lane_i].sym_host[j->slot];
    Packet pk;
    pk.pts = j->pts; pk.dts = j->dts; pk.key = j->slice_type == 2;
    // a picture's later slices (sessions on other devices, cfg.slice_index > 0) contribute their slice NAL unit only: the access unit's
    // delimiter, parameter sets and SEI come with slice 0
    const bool au_head = s->cfg.slice_count <= 1 || s->cfg.slice_index == 0;
    if (s->cfg.aud && au_head) write_aud(j->slice_type, pk.data);
    if (au_head && j->slice_type == 2 && (j->first_of_stream || s->cfg.repeat_headers)) pk.data.insert(pk.data.end(), s->headers.begin(), s->headers.end());
    if (s->cfg.hrd && au_head) {
        if (j->slice_type == 2) write_sei_buffering_period(s->cfg, pk.data);
        // clock ticks since the previous buffering period: position in the GOP, or the previous GOP's length at an IDR
        // (cfg.bframes: removal happens in decoding order; a picture is shown one tick after the picture at its display place was removed)
        write_sei_pic_timing(s->cfg, (uint32_t)(j->dec_pos > 0 ? j->dec_pos - 1 : (j->dec_index > 0 ? j->prev_gop_len - 1 : 0)), pk.data,
                             j->reorder ? (uint32_t)(j->poc + 1 - j->dec_pos) : 0u);
    }
    assemble_picture(s->cfg, j->pic, j->sub, pk.data, false);
    const unsigned long long *sse = (const unsigned long long *)(b + sl.sse);
    pk.ready = true;
    auto t1 = std::chrono::steady_clock::now();
    s->entropy_ns += j->ns.load() + std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    const int slot = j->slot;
    const int64_t index = j->index;
    {
        std::lock_guard<std::mutex> l(s->m);
        s->stats.sse_y += (double)sse[0]; s->stats.sse_u += (double)sse[1]; s->stats.sse_v += (double)sse[2];
        s->stats.bytes_out += (int64_t)pk.data.size();
        if ((size_t)index < s->frames.size()) {
            auto &fr = s->frames[(size_t)index];
            fr.bits_local = (long long)pk.data.size() * 8;
            fr.est_local = *(const unsigned long long *)(b + sl.est);
            if (!s->group) {          // (slices with one rate plan: the step loop sums sizes and estimates over the slices at fixed points)
                fr.bits = fr.bits_local;
                fr.est_q4 = *(const unsigned long long *)(b + sl.est);
                fr.est_known = true;
            }
        }
        s->packets[j->dec_index] = std::move(pk);
        s->frames_done++;
        delete j;
        s->jobs_open[slot]--;
        s->cv.notify_all();      // under the lock: mihevc_close may delete the session as soon as its last job has let go of the mutex
    }
}

void entropy_part(PictureJob *j, int part)
{
    auto t0 = std::chrono::steady_clock::now();
    const int t_a = (int)((long long)j->n_tiles * part / j->parts), t_b = (int)((long long)j->n_tiles * (part + 1) / j->parts);
    encode_tiles(j->s->cfg, j->pic, t_a, t_b, j->sub);
    j->ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    if (j->left.fetch_sub(1) == 1) publish_picture(j);
}

// sum of |a - b| over every 4th sample of every 4th row of consecutive pending source pictures: out[i] for the pair (i - 1, i), out[0] = 0
template <typename T> int scene_differences(mihevc_session *s, int n, std::vector<unsigned long long> &out)
{
    const size_t in_bytes = (size_t)n * sizeof(ScenePic<T>), need = ((in_bytes + 255) & ~(size_t)255) + (size_t)n * sizeof(unsigned long long);
    if (need > s->scene_cap) {
        BufferCache &bc = BufferCache::get();
        bc.release(s->device, s->scene_cap, false, s->d_scene);
        s->d_scene = nullptr; s->scene_cap = 0;
        const size_t cap = (need + 0xfff) & ~(size_t)0xfff;
        HIPCK(s, bc.alloc(s->device, cap, false, &s->d_scene));
        s->scene_cap = cap;
    }
    std::vector<ScenePic<T>> pics((size_t)n);
    for (int i = 0; i < n; i++) pics[(size_t)i] = ScenePic<T>{(const T *)s->pending[(size_t)i].p[0], s->pending[(size_t)i].stride[0]};
    unsigned long long *d_out = (unsigned long long *)((uint8_t *)s->d_scene + ((in_bytes + 255) & ~(size_t)255));
    HIPCK(s, hipMemcpyAsync(s->d_scene, pics.data(), in_bytes, hipMemcpyHostToDevice, s->st_compute));
    HIPCK(s, hipMemsetAsync(d_out, 0, (size_t)n * sizeof(unsigned long long), s->st_compute));
    HIPCK(s, launch_scene_diff<T>(s->st_compute, (const ScenePic<T> *)s->d_scene, d_out, s->w, s->h, n));
    HIPCK(s, hipMemcpyAsync(out.data(), d_out, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->st_compute));
    HIPCK(s, hipStreamSynchronize(s->st_compute));
    return 0;
}

// ---- the slices of one picture exchange rows (cfg.slice_halo; csrc/slice_group.h) ------------------------------------------------------------
// job tables of one step: [export: lanes x 8][import: lanes x 8][pull: lanes x 3 x reach], RowCopy each
struct JobLayout {
    int lanes, reach;
    size_t exp, imp, pull, total;
    JobLayout(int lanes_, int reach_) : lanes(lanes_), reach(reach_)
    {
        exp = 0; imp = exp + (size_t)lanes * 8 * sizeof(RowCopy); pull = imp + (size_t)lanes * 8 * sizeof(RowCopy);
        total = (pull + (size_t)lanes * 3 * (size_t)std::max(1, reach) * sizeof(RowCopy) + 255) & ~(size_t)255;
    }
};

// bands whose rows lie within PAD_Y rows above (dir -1) / below (dir +1) this band, nearest first, with the rows each contributes
static void bands_in_reach(const mihevc_session *s, int dir, std::vector<std::pair<int, int>> &out)
{
    int left = PAD_Y;
    for (int b = s->band + dir; b >= 0 && b < s->n_bands && left > 0; b += dir) {
        const int rows = std::min(left, s->band_h[b]);
        out.push_back({b, rows});
        left -= rows;
    }
}
static int rows_in_reach(const mihevc_session *s, int dir)
{
    std::vector<std::pair<int, int>> v;
    bands_in_reach(s, dir, v);
    int n = 0;
    for (auto &x : v) n += x.second;
    return n;
}

static int group_sum(mihevc_session *s, std::vector<double> &v)
{
    if (!s->group) return 0;
    if (!s->group->allreduce(v)) { s->failed = true; s->err = "another slice of the picture failed"; return MIHEVC_EDEVICE; }
    return 0;
}

// cfg.bframes = -1: are B pictures worth it for this chunk?  With a B picture between every two anchors an anchor predicts from TWO pictures back; that pays when
// motion stays trackable over two pictures (translation, static content) and costs when it does not (zoom, fades: tools/rd_curve.py, profiles/r03: +7 % bits on
// the `stress` clip even with B pictures at the anchors' QP).  The probe asks the integer search itself: for up to 4 pictures spread over the chunk, k_me_search
// of the SOURCE picture against the source one place back and two places back (copied into lane 0's reference buffers, border padded); c1 / c2 = the 32x32 nodes'
// best costs (SAD << 4 + lambda * mvd bits) summed over all CTUs.  B pictures when c2 <= kProbeRatio x c1.  A 1/4-size search cannot tell (sub-sample
// motion dominates its SADs: it rated the translating clip WORSE than the zooming one).  ~0.4 ms per chunk at 1080p.  stats.reserved[0] / [1] keep the last
// c1 / c2 per CTU, [2] the decision (tests and tools read them).
constexpr double kProbeRatio = 1.4;
template <typename T> static int probe_bframes(mihevc_session *s, int n, bool &use_b)
{
    use_b = false;
    if (n < 3) return 0;
    if (int e = ensure_lanes(s, 1)) return e;
    mihevc_session::Lane &L = s->lane[0];
    const int K = std::min(4, (n - 2 + 15) / 16 + 1);
    std::vector<int> at;
    for (int k = 0; k < K; k++) { const int p = 2 + (int)((long long)(n - 3) * k / std::max(1, K - 1)); if (at.empty() || at.back() != p) at.push_back(p); }
    const size_t o_inter = (2 * sizeof(SaoArgs<T>) + 255) & ~(size_t)255, o_total = o_inter + 2 * at.size() * sizeof(InterArgs<T>);
    if (o_total > s->probe_cap) {
        BufferCache &bc = BufferCache::get();
        bc.release(s->device, s->probe_cap, false, s->d_probe);
        s->d_probe = nullptr; s->probe_cap = 0;
        const size_t cap = (o_total + 0xffff) & ~(size_t)0xffff;
        HIPCK(s, bc.alloc(s->device, cap, false, &s->d_probe));
        s->probe_cap = cap;
    }
    uint8_t *base = (uint8_t *)s->d_probe;
    mihevc_cost_params c;
    mihevc_cost_params_for_qp(s->qp_p, s->cfg.bit_depth, s->me_range, &c);
    const CostParams P{c.qp, c.qp_c, c.bit_depth, c.lambda_sad_q4, c.lambda_q4, c.me_range, 1, 1, false, false, false, false, false, false, false, 0};
    SaoArgs<T> pad[2];
    memset(pad, 0, sizeof pad);
    for (int k = 0; k < 2; k++) { for (int i = 0; i < 3; i++) pad[k].out[i] = mk<T>(L.rec_p[k][i], L.rec_stride[i]); pad[k].w = s->w; pad[k].h = s->h; }
    std::vector<InterArgs<T>> ia(2 * at.size());
    for (size_t k = 0; k < at.size(); k++)
        for (int d = 0; d < 2; d++) {
            InterArgs<T> &a = ia[2 * k + (size_t)d];
            memset((void *)&a, 0, sizeof a);
            const mihevc_session::Src &src = s->pending[(size_t)at[k]];
            for (int i = 0; i < 3; i++) { a.src[i] = mkc<T>(src.p[i], src.stride[i]); a.ref[i] = mkc<T>(L.rec_p[d][i], L.rec_stride[i]); }
            a.w = s->w; a.h = s->h; a.ctus_w = s->ctus_w; a.prm = P; a.me = d ? L.me1 : L.me;
        }
    HIPCK(s, hipMemcpyAsync(base, pad, sizeof pad, hipMemcpyHostToDevice, s->st_compute));
    HIPCK(s, hipMemcpyAsync(base + o_inter, ia.data(), ia.size() * sizeof(InterArgs<T>), hipMemcpyHostToDevice, s->st_compute));
    std::vector<int32_t> me((size_t)2 * at.size() * s->n_ctu * 63);
    const size_t es = esize(s), me_bytes = (size_t)s->n_ctu * 63 * sizeof(int32_t);
    for (size_t k = 0; k < at.size(); k++) {
        for (int d = 0; d < 2; d++) {          // the luma of the pictures one and two places back -> the reference buffers (the search reads luma only)
            const mihevc_session::Src &r = s->pending[(size_t)(at[k] - 1 - d)];
            HIPCK(s, hipMemcpy2DAsync(L.rec_p[d][0], L.rec_stride[0] * es, r.p[0], r.stride[0] * es, s->w * es, s->h, hipMemcpyDeviceToDevice, s->st_compute));
        }
        HIPCK(s, launch_pad<T>(s->st_compute, (const SaoArgs<T> *)base, s->w, s->h, 2));
        for (int d = 0; d < 2; d++) {
            HIPCK(s, launch_me_search<T>(s->st_compute, (const InterArgs<T> *)(base + o_inter) + 2 * k + (size_t)d, s->n_ctu, 1, s->me_range, 0));
            HIPCK(s, hipMemcpyAsync(me.data() + (2 * k + (size_t)d) * s->n_ctu * 63, d ? L.me1 : L.me, me_bytes, hipMemcpyDeviceToHost, s->st_compute));
        }
    }
    HIPCK(s, hipStreamSynchronize(s->st_compute));
    unsigned long long c1 = 0, c2 = 0;
    for (size_t k = 0; k < at.size(); k++)
        for (int ctu = 0; ctu < s->n_ctu; ctu++) {
            // node 0 (the 32x32 block); CTUs the picture cuts off have no such node: their four 16x16 / sixteen 8x8 nodes stand in
            for (int d = 0; d < 2; d++) {
                const int32_t *m = me.data() + ((2 * k + (size_t)d) * s->n_ctu + (size_t)ctu) * 63;
                unsigned long long v = 0;
                if (m[2] >= 0) v = (unsigned long long)m[2];
                else for (int nd = 1; nd < 21; nd++) { if (nd < 5 ? m[3 * nd + 2] >= 0 : m[3 * (1 + ((nd - 5) >> 2)) + 2] < 0 && m[3 * nd + 2] >= 0) v += (unsigned long long)m[3 * nd + 2]; }
                (d ? c2 : c1) += v;
            }
        }
    use_b = (double)c2 <= kProbeRatio * (double)c1;
    const unsigned long long per = (unsigned long long)s->n_ctu * at.size();
    s->stats.reserved[0] = (int32_t)std::min<unsigned long long>(0x7fffffff, c1 / per);
    s->stats.reserved[1] = (int32_t)std::min<unsigned long long>(0x7fffffff, c2 / per);
    s->stats.reserved[2] = use_b;
    return 0;
}

// cfg.bframes: a closed GOP of `len` pictures is coded I0 P2 b1 P4 b3 ...: step 0 the IDR picture, odd steps the anchors (P), even steps the B picture between the
// last two anchors; the GOP's last picture is always an anchor.  Display position / slice type (2 I, 1 P, 0 B) of step t; without B pictures step = position.
static inline int pos_of_step(bool bf, int t, int len) { return !bf || t == 0 ? t : (t & 1) ? std::min(t + 1, len - 1) : t - 1; }
static inline int type_of_step(bool bf, int t) { return t == 0 ? 2 : (bf && !(t & 1)) ? 0 : 1; }

template <typename T> int encode_chunk(mihevc_session *s)
{
    const int n = (int)s->pending.size();
    if (!n) return 0;
    const auto wall0 = std::chrono::steady_clock::now();      // stats.reserved[3..5]: host time of the chunk in front of its first launch / behind its last kernel / in all (us, summed)
    // ---- GOP layout of the chunk.  Scene cuts (x265 scenecut + min-keyint, reference core/transcoder.py:401) divide the chunk into segments; every
    //      segment is coded as the FEWEST closed GOPs keyint allows (the IDR count of an IDR-every-keyint layout), of near-equal length when
    //      cfg.gop_balance is set: the lanes of the lock-step pipeline then run out together instead of idling behind a short last GOP (a 300-picture
    //      clip at keyint 90 is 4 x 75 steps, not 90 steps of which 60 drive three lanes).  With gop_balance 0 a segment's IDRs sit every keyint pictures.
    //      The cut detector is the mean absolute difference of every 4th sample of every 4th row between consecutive source pictures (k_scene_diff, one
    //      launch for the chunk): a cut is a difference above kCutAbs grey levels that is also kCutRatio times the running mean over the ordinary
    //      pictures before it, taken when every GOP of the segment it closes keeps at least min-keyint pictures.  A session that codes one slice of
    //      the picture sees only its band, and the slices of a picture must agree on its type: no cut detection there.
    const int keyint = s->keyint;
    auto gops_of = [keyint](int len) { return (len + keyint - 1) / keyint; };
    std::vector<int> seg{0};                  // segment starts
    if (s->cfg.scenecut && s->cfg.slice_count <= 1 && n > 1 && s->cfg.min_keyint < keyint) {
        std::vector<unsigned long long> diff((size_t)n, 0);
        if (int e = scene_differences<T>(s, n, diff)) return e;
        const double per = (double)((s->w + 3) / 4) * ((s->h + 3) / 4) * (1 << (s->cfg.bit_depth - 8));
        int total = 0;                        // GOPs of the closed segments
        // Until the running mean has seen an ordinary picture (a session's first pictures) the chunk's MEDIAN difference stands in for it: a cut or a
        // flash at the session's second picture is then a jump like any other and never becomes the mean the next pictures are measured against
        double median = 0;
        if (s->scene_avg <= 0) {
            std::vector<unsigned long long> sorted(diff.begin() + 1, diff.end());
            std::nth_element(sorted.begin(), sorted.begin() + (ptrdiff_t)(sorted.size() / 2), sorted.end());
            median = (double)sorted[sorted.size() / 2] / per;
        }
        auto is_jump = [&](int i) {
            const double d = (double)diff[(size_t)i] / per, base = s->scene_avg > 0 ? s->scene_avg : median;
            return d > kCutAbs && d > kCutRatio * base;
        };
        for (int i = 1; i < n; i++) {
            const double d = (double)diff[(size_t)i] / per;
            const int len = i - seg.back(), g = gops_of(len);
            const bool jump = is_jump(i);
            // a run of jumps (a flash: into the odd picture and out of it again) is cut at its LAST picture: the GOP then starts on the scene that stays,
            // not on the flash it would have to predict everything from
            const bool last_of_run = !(i + 1 < n && is_jump(i + 1));
            const int shortest = s->cfg.gop_balance ? len / g : (len % keyint ? len % keyint : keyint);
            // the GOP the cut opens must keep min-keyint pictures too: the next chunk starts with an IDR picture of its own (the stream's last chunk may end short)
            const bool tail_ok = s->flushing || n - i >= std::max(1, s->cfg.min_keyint);
            if (jump && last_of_run && tail_ok && shortest >= std::max(1, s->cfg.min_keyint) && total + g + gops_of(n - i) <= MAX_LANES) { total += g; seg.push_back(i); }
            if (!jump) s->scene_avg = s->scene_avg > 0 ? 0.8 * s->scene_avg + 0.2 * d : d;      // ordinary pictures only: a jump says nothing about the new scene's motion
        }
    }
    std::vector<int> gstart_stream;
    for (size_t k = 0; k < seg.size(); k++) {
        const int a0 = seg[k], len = (k + 1 < seg.size() ? seg[k + 1] : n) - a0, g = gops_of(len);
        if (s->cfg.gop_balance)
            for (int j = 0, at = a0; j < g; at += len / g + (j < len % g), j++) gstart_stream.push_back(at);
        else
            for (int at = a0; at < a0 + len; at += keyint) gstart_stream.push_back(at);
    }
    const int gops = (int)gstart_stream.size();
    // lanes by GOP length, longest first: the lanes that still have a picture at step t are then a prefix [0, batch[t])
    std::vector<int> order((size_t)gops), gstart((size_t)gops), glen((size_t)gops), prev_len((size_t)gops);
    for (int g = 0; g < gops; g++) order[(size_t)g] = g;
    auto len_of = [&](int k) { return (k + 1 < gops ? gstart_stream[(size_t)k + 1] : n) - gstart_stream[(size_t)k]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len_of(a) > len_of(b); });
    for (int g = 0; g < gops; g++) {
        const int k = order[(size_t)g];
        gstart[(size_t)g] = gstart_stream[(size_t)k]; glen[(size_t)g] = len_of(k);
        prev_len[(size_t)g] = k > 0 ? len_of(k - 1) : s->last_gop_len;
    }
    s->last_gop_len = len_of(gops - 1);
    if (int e = ensure_lanes(s, gops)) return e;
    if (int e = ensure_flow(s)) return e;
    SymLayout sl(s->w, s->h);
    const int steps = glen[0];
    // ---- slices that exchange rows: what the neighbours need to know about this band's buffers, then everybody's (csrc/slice_group.h)
    const bool grp = (bool)s->group;
    const int up = grp && s->band > 0 ? 1 : 0, dn = grp && s->band + 1 < s->n_bands ? 1 : 0;
    std::vector<std::pair<int, int>> reach_up, reach_dn;
    if (grp) { bands_in_reach(s, -1, reach_up); bands_in_reach(s, +1, reach_dn); }
    const int halo_top = grp ? rows_in_reach(s, -1) : 0, halo_bottom = grp ? rows_in_reach(s, +1) : 0;
    const int reach = (int)(reach_up.size() + reach_dn.size());
    const JobLayout jl(gops, reach);
    uint8_t *hj = nullptr, *dj = nullptr;
    if (grp) {
        if (gops > kHaloLanes) { s->err = "too many GOP lanes for sliced pictures"; return MIHEVC_EINVAL; }
        BandPub &me = s->group->pub(s->band);
        me.device = s->device; me.w = s->w; me.h = s->h; me.is16 = s->is16;
        for (int g = 0; g < gops; g++)
            for (int k = 0; k < 2; k++)
                for (int i = 0; i < 3; i++) me.rec_p[g][k][i] = s->lane[g].rec_p[k][i];
        for (int i = 0; i < 3; i++) me.rec_stride[i] = s->lane[0].rec_stride[i];
        for (int k = 0; k < 2; k++) { me.x1_export[k] = s->x1_export[k]; me.ev_x1[k] = s->ev_x1[k]; me.ev_x2[k] = s->ev_x2[k]; }
        me.x1_lane_bytes = s->x1_lane_bytes;
        if (!s->group->barrier()) { s->failed = true; s->err = "another slice of the picture failed"; return MIHEVC_EDEVICE; }
        for (auto &v : {reach_up, reach_dn})
            for (auto &br : v) {
                const int dev = s->group->pub(br.first).device;
                if (dev == s->device || std::find(s->peers_enabled.begin(), s->peers_enabled.end(), dev) != s->peers_enabled.end()) continue;
                hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); s->err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); s->failed = true; s->group->fail(); return MIHEVC_EDEVICE; }
                (void)hipGetLastError();
                s->peers_enabled.push_back(dev);
            }
        const size_t jneed = (size_t)steps * jl.total;
        if (jneed > s->jobs_cap) {
            BufferCache &bc = BufferCache::get();
            bc.release(s->device, s->jobs_cap, false, s->d_jobs); bc.release(s->device, s->jobs_cap, true, s->h_jobs);
            s->d_jobs = nullptr; s->h_jobs = nullptr; s->jobs_cap = 0;
            const size_t cap = (jneed + 0xffff) & ~(size_t)0xffff;
            HIPCK(s, bc.alloc(s->device, cap, false, &s->d_jobs));
            HIPCK(s, bc.alloc(s->device, cap, true, (void **)&s->h_jobs));
            s->jobs_cap = cap;
        }
        hj = s->h_jobs; dj = (uint8_t *)s->d_jobs;
        memset(hj, 0, jneed);
    }
    const int ring = s->ring;
    auto slot_of = [ring](int t) { return t == 0 ? 0 : 1 + (t - 1) % (ring - 1); };
    // ---- build every step's argument blocks, upload once ----
    const StepLayout<T> lay(gops);
    const size_t flat_off = (size_t)(steps + 1) * lay.total;  // + one block for the rho trial (below)
    const int kQpB = s->cfg.b_qp_offset >= 0 ? std::min(8, s->cfg.b_qp_offset) : 2;      // a B picture takes the QP of the anchors around it + 2 (x265 pbratio 1.3): nothing predicts from it
    bool bf_decided = s->cfg.bframes > 0;
    if (s->cfg.bframes < 0) { if (int e = probe_bframes<T>(s, n, bf_decided)) return e; }
    const bool bf = bf_decided;
    const int n_pre = bf ? 2 * n : n;       // pre-search blocks: one per picture of the chunk (stream order); with B pictures a second one per picture for list 1
    const size_t need = flat_off + (size_t)n_pre * sizeof(PreArgs<T>);
    if (need > s->args_cap) {
        BufferCache &bc = BufferCache::get();
        bc.release(s->device, s->args_cap, false, s->d_args); bc.release(s->device, s->args_cap, true, s->h_args);
        s->d_args = nullptr; s->h_args = nullptr; s->args_cap = 0;
        const size_t cap = (need + 0xffff) & ~(size_t)0xffff;      // whole 64 KiB: the next session's chunk finds the block in the cache
        HIPCK(s, bc.alloc(s->device, cap, false, &s->d_args));
        HIPCK(s, bc.alloc(s->device, cap, true, (void **)&s->h_args));
        s->args_cap = cap;
    }
    uint8_t *ha = s->h_args, *da = (uint8_t *)s->d_args;
    // the chunk's 1/4-size source pictures and search centres (cfg.pre_search): [n pictures of (w/4)(h/4) bytes | n x n_ctu x 2 int16]
    const size_t low_pic = (size_t)(s->w >> 2) * (s->h >> 2), low_bytes = ((size_t)n_pre * low_pic + 255) & ~(size_t)255;
    if (s->cfg.pre_search) {
        const size_t want = low_bytes + (size_t)n_pre * s->n_ctu * 2 * sizeof(int16_t);
        if (want > s->low_cap) {
            BufferCache &bc = BufferCache::get();
            bc.release(s->device, s->low_cap, false, s->d_low);
            s->d_low = nullptr; s->low_cap = 0;
            const size_t cap = (want + 0xfffff) & ~(size_t)0xfffff;
            HIPCK(s, bc.alloc(s->device, cap, false, &s->d_low));
            s->low_cap = cap;
        }
    }
    uint8_t *const low = (uint8_t *)s->d_low;
    int16_t *const cen = (int16_t *)((uint8_t *)s->d_low + low_bytes);
    auto prm_for = [&](int qp) {
        mihevc_cost_params c;
        mihevc_cost_params_for_qp(qp, s->cfg.bit_depth, s->me_range, &c);
        return CostParams{c.qp, c.qp_c, c.bit_depth, c.lambda_sad_q4, c.lambda_q4, c.me_range, s->tiles.cols, s->tiles.rows, s->cfg.intra_nxn != 0, s->cfg.intra_in_p != 0, s->cfg.pre_search != 0, s->cfg.rdo_zero != 0, s->cfg.chroma_modes != 0,
                          s->cfg.slice_count > 1 && !s->cfg.slice_halo && s->cfg.slice_index > 0, s->cfg.slice_count > 1 && !s->cfg.slice_halo && s->cfg.slice_index < s->cfg.slice_count - 1, std::max(0, s->cfg.rdo_cg)};
    };
    const int64_t first_index = s->frames_in - n;
    {
        std::lock_guard<std::mutex> l(s->m);
        s->frames.resize((size_t)s->frames_in);
    }
    std::vector<int> batch(steps, 0);
    for (int t = 0; t < steps; t++)
        for (int g = 0; g < gops; g++) {
            if (t >= glen[(size_t)g]) continue;
            const int pos = pos_of_step(bf, t, glen[(size_t)g]), ptype = type_of_step(bf, t);
            int fi = gstart[(size_t)g] + pos;
            // lanes with a picture at step t are a prefix [0, batch): they are sorted by GOP length
            batch[t] = g + 1;
            mihevc_session::Lane &L = s->lane[g];
            mihevc_session::Src &src = s->pending[fi];
            StepView<T> hv(ha, lay, t);
            struct { IntraArgs<T> &intra; InterArgs<T> &inter; DeblockArgs<T> &dbk_v, &dbk_h; SaoArgs<T> &sao; } A{hv.intra[g], hv.inter[g], hv.dbk_v[g], hv.dbk_h[g], hv.sao[g]};
            uint8_t *sym = L.sym_dev[slot_of(t)];
            // reconstruction buffers: anchor number k (the IDR picture is 0) goes to buffer k & 1 and predicts from the other one; the B picture between
            // anchors k - 1 and k reads both and goes to buffer 2
            const int anchor = !bf ? t : (t + 1) / 2;
            const int cur = ptype == 0 ? 2 : anchor & 1, prev = ptype == 0 ? (t / 2 - 1) & 1 : (anchor & 1) ^ 1, nxt = ptype == 0 ? (t / 2) & 1 : 0;
            const int ref_pos = ptype == 0 ? pos - 1 : !bf ? pos - 1 : pos_of_step(bf, std::max(0, t - 2), glen[(size_t)g]) * (t > 1) ;      // display position of the list-0 reference
            const CostParams P = prm_for(t == 0 ? s->qp_i : s->qp_p);      // provisional; the controller patches it per step
            for (int i = 0; i < 3; i++) {
                A.intra.src[i] = mkc<T>(src.p[i], src.stride[i]); A.intra.rec[i] = mk<T>(L.work_p[i], L.work_stride[i]);
                A.inter.src[i] = mkc<T>(src.p[i], src.stride[i]); A.inter.ref[i] = mkc<T>(L.rec_p[prev][i], L.rec_stride[i]);
                A.inter.rec[i] = mk<T>(L.work_p[i], L.work_stride[i]);
                A.dbk_v.rec[i] = A.dbk_h.rec[i] = mk<T>(L.work_p[i], L.work_stride[i]);
                A.sao.src[i] = mkc<T>(src.p[i], src.stride[i]); A.sao.dbk[i] = mkc<T>(L.work_p[i], L.work_stride[i]);
                A.sao.out[i] = mk<T>(L.rec_p[cur][i], L.rec_stride[i]);
            }
            A.intra.w = A.inter.w = A.dbk_v.w = A.dbk_h.w = A.sao.w = s->w;
            A.intra.h = A.inter.h = A.dbk_v.h = A.dbk_h.h = A.sao.h = s->h;
            A.intra.ctus_w = A.inter.ctus_w = A.sao.ctus_w = s->ctus_w; A.intra.ctus_h = s->ctus_h;
            A.intra.prm = A.inter.prm = A.sao.prm = P;
            A.intra.cu = A.inter.cu = (mihevc_cu_rec *)(sym + sl.cu);
            A.dbk_v.cu = A.dbk_h.cu = (const mihevc_cu_rec *)(sym + sl.cu);
            A.sao.halo_top = halo_top; A.sao.halo_bottom = halo_bottom;
            // SAO on: the CTU programs of the SAO kernel deblock their own tile first (one launch for 8.7.2 + 8.7.3, the work picture stays as the analysis left it)
            A.sao.cu = s->cfg.sao ? (const mihevc_cu_rec *)(sym + sl.cu) : nullptr;
            if (grp) {
                // deblocking runs over the band EXTENDED by the rows the neighbours hand over (kSeamRows of their pre-deblock reconstruction + one row of CU
                // records either side): the seams are inner edges of that picture
                const int w8 = s->w >> 3;
                for (int i = 0; i < 3; i++) {
                    const int tr = up ? (i ? kSeamRows / 2 : kSeamRows) : 0;
                    A.dbk_v.rec[i] = A.dbk_h.rec[i] = mk<T>((T *)L.work_p[i] - (ptrdiff_t)tr * L.work_stride[i], L.work_stride[i]);
                }
                A.dbk_v.h = A.dbk_h.h = s->h + kSeamRows * (up + dn);
                A.dbk_v.cu = A.dbk_h.cu = (const mihevc_cu_rec *)(sym + sl.cu) - (up ? w8 : 0);
                // the step's row copies
                const long long G = s->gstep + t;
                const size_t es = sizeof(T), part = s->x1_part_bytes;
                RowCopy *je = (RowCopy *)(hj + (size_t)t * jl.total + jl.exp) + (size_t)g * 8, *ji = (RowCopy *)(hj + (size_t)t * jl.total + jl.imp) + (size_t)g * 8;
                RowCopy *jp = (RowCopy *)(hj + (size_t)t * jl.total + jl.pull) + (size_t)g * 3 * std::max(1, reach);
                uint8_t *xe = (uint8_t *)s->x1_export[G & 1] + (size_t)g * s->x1_lane_bytes;
                const size_t off_pl[3] = {0, (size_t)kSeamRows * s->w * es, (size_t)kSeamRows * s->w * es + (size_t)(kSeamRows / 2) * (s->w / 2) * es};
                const size_t off_cu = off_pl[2] + (size_t)(kSeamRows / 2) * (s->w / 2) * es;
                mihevc_cu_rec *cu0 = (mihevc_cu_rec *)(sym + sl.cu);
                for (int side = 0; side < 2; side++) {          // export: this band's first / last rows -> [top part | bottom part]
                    for (int i = 0; i < 3; i++) {
                        const int pw = i ? s->w / 2 : s->w, ph = i ? s->h / 2 : s->h, rows = i ? kSeamRows / 2 : kSeamRows;
                        je[side * 4 + i] = RowCopy{(const uint8_t *)L.work_p[i] + (size_t)(side ? ph - rows : 0) * L.work_stride[i] * es, xe + side * part + off_pl[i], (int)(pw * es), rows,
                                                   (int)(L.work_stride[i] * es), (int)(pw * es)};
                    }
                    je[side * 4 + 3] = RowCopy{cu0 + (size_t)(side ? (s->h >> 3) - 1 : 0) * w8, xe + side * part + off_cu, (int)(w8 * sizeof(mihevc_cu_rec)), 1, 0, 0};
                }
                for (int side = 0; side < 2; side++) {          // import: the upper neighbour's bottom part -> the rows above this band; the lower neighbour's top part -> below
                    if (!(side ? dn : up)) continue;
                    const BandPub &nb = s->group->pub(s->band + (side ? 1 : -1));
                    const uint8_t *xs = (const uint8_t *)nb.x1_export[G & 1] + (size_t)g * nb.x1_lane_bytes + (side ? 0 : part);
                    for (int i = 0; i < 3; i++) {
                        const int pw = i ? s->w / 2 : s->w, ph = i ? s->h / 2 : s->h, rows = i ? kSeamRows / 2 : kSeamRows;
                        ji[side * 4 + i] = RowCopy{xs + off_pl[i], (uint8_t *)L.work_p[i] + ((ptrdiff_t)(side ? ph : -rows) * L.work_stride[i]) * (ptrdiff_t)es, (int)(pw * es), rows,
                                                   (int)(pw * es), (int)(L.work_stride[i] * es)};
                    }
                    ji[side * 4 + 3] = RowCopy{xs + off_cu, side ? cu0 + (size_t)(s->h >> 3) * w8 : cu0 - w8, (int)(w8 * sizeof(mihevc_cu_rec)), 1, 0, 0};
                }
                if (t > 0) {                                    // pull: the final reconstruction either side of the seams -> the border rows of this band's reference
                    int k = 0;
                    for (int side = 0; side < 2; side++) {
                        int done = 0;
                        for (auto &br : side ? reach_dn : reach_up) {
                            const BandPub &nb = s->group->pub(br.first);
                            for (int i = 0; i < 3; i++) {
                                const int pw = i ? s->w / 2 : s->w, ph = i ? s->h / 2 : s->h, nh = i ? nb.h / 2 : nb.h, rows = i ? br.second / 2 : br.second, dn_ = i ? done / 2 : done;
                                const uint8_t *src = (const uint8_t *)nb.rec_p[g][prev][i] + (size_t)(side ? 0 : nh - rows) * nb.rec_stride[i] * es;
                                uint8_t *dst = (uint8_t *)L.rec_p[prev][i] + ((ptrdiff_t)(side ? ph + dn_ : -(dn_ + rows)) * L.rec_stride[i]) * (ptrdiff_t)es;
                                jp[k++] = RowCopy{src, dst, (int)(pw * es), rows, (int)(nb.rec_stride[i] * es), (int)(L.rec_stride[i] * es)};
                            }
                            done += br.second;
                        }
                    }
                }
            }
            // levels go straight to the pinned host block (device-mapped): only TUs with a non-zero level are stored, so the
            // 6 MB/picture coefficient planes never cross PCIe as a blit (profiles/r01: copyBuffer was 17 % of GPU time)
            uint8_t *symh = L.sym_host[slot_of(t)];
            int16_t *c3[3] = {(int16_t *)(symh + sl.cy), (int16_t *)(symh + sl.cu_), (int16_t *)(symh + sl.cv)};
            for (int i = 0; i < 3; i++) A.intra.coef[i] = A.inter.coef[i] = c3[i];
            A.intra.sparse_coef = A.inter.sparse_coef = 1;
            A.intra.diagonal = 0;
            A.inter.centers = nullptr; A.inter.me = L.me;
            for (int i = 0; i < 3; i++) A.inter.ref1[i] = ptype == 0 ? mkc<T>(L.rec_p[nxt][i], L.rec_stride[i]) : Plane<const T>{nullptr, 0};
            A.inter.centers1 = nullptr; A.inter.me1 = ptype == 0 ? L.me1 : nullptr;
            if (s->cfg.pre_search) {       // search centres: the chunk's pre-search fills them for every picture (below)
                const size_t idx = (size_t)fi, ridx = (size_t)(gstart[(size_t)g] + std::max(0, ref_pos));
                PreArgs<T> &P4 = ((PreArgs<T> *)(ha + flat_off))[idx];
                P4.src = A.inter.src[0]; P4.ref = A.inter.src[0];
                P4.lsrc = low + idx * low_pic; P4.lref = low + (t > 0 ? ridx : idx) * low_pic;      // against the SOURCE of the picture it will predict from; an IDR picture's centres are never read
                P4.w = s->w; P4.h = s->h; P4.bit_depth = s->cfg.bit_depth; P4.centers = cen + idx * (size_t)s->n_ctu * 2; P4.cost = nullptr;
                if (t > 0) A.inter.centers = P4.centers;
                if (bf) {                  // the second block: a B picture against the source of the anchor AFTER it (other pictures: a copy of the first, never read)
                    PreArgs<T> &P5 = ((PreArgs<T> *)(ha + flat_off))[(size_t)n + idx];
                    P5 = P4;
                    P5.lsrc = low + ((size_t)n + idx) * low_pic; P5.centers = cen + ((size_t)n + idx) * (size_t)s->n_ctu * 2;
                    if (ptype == 0) { P5.lref = low + (idx + 1) * low_pic; A.inter.centers1 = P5.centers; }
                }
            }
            // P pictures: the inter pass leaves per-CTU costs for the intra second pass, which runs on the same work picture,
            // records and levels with the one-tile PPS 0 geometry
            const bool ipass = ptype == 1 && s->cfg.intra_in_p;
            A.inter.ip = ipass ? L.ip : nullptr;
            A.intra.ip = ipass ? L.ip : nullptr;
            A.intra.plan = t == 0 ? L.plan : nullptr;      // P pictures' second pass plans and codes a CTU inside one workgroup
            if (t > 0) { A.intra.prm.tile_cols = s->ptiles.cols; A.intra.prm.tile_rows = s->ptiles.rows; }
            A.dbk_v.bit_depth = A.dbk_h.bit_depth = s->cfg.bit_depth; A.dbk_v.dir = 0; A.dbk_h.dir = 1;
            A.dbk_v.y_org = A.dbk_h.y_org = up ? kSeamRows : 0;
            A.sao.sao = s->cfg.sao ? (mihevc_sao_ctu *)(sym + sl.sao) : nullptr;
            A.sao.sse = (unsigned long long *)(sym + sl.sse);
            A.sao.sse_ctu = s->cfg.sao ? (uint32_t *)(sym + sl.sse_ctu) : nullptr;
            A.intra.est = A.inter.est = (unsigned long long *)(sym + sl.est);
        }
    HIPCK(s, hipMemcpyAsync(da, ha, need, hipMemcpyHostToDevice, s->st_compute));
    if (grp) HIPCK(s, hipMemcpyAsync(dj, hj, (size_t)steps * jl.total, hipMemcpyHostToDevice, s->st_compute));
    // cfg.pre_search: the search centres of EVERY picture of the chunk come from the 1/4-size SOURCE pictures (this picture against the one before it: nothing
    // in it waits for a reconstruction), in two launches on a stream of their own in the IDR step below: the work (7 % of a clip's device time when it ran
    // inside every step) sits beside the anti-diagonal chain, which leaves most of the device idle.  The first P step waits for ev_pre.
    // ---- per-lane rate controllers ----
    // CPB model (x265 nal-hrd=vbr + vbv-maxrate / vbv-bufsize, reference core/transcoder.py:399-400).  The GOPs of a chunk are coded in
    // lock-step, so a GOP cannot know the buffer level its predecessor leaves.  Every closed GOP is therefore planned to be buffer-neutral:
    // it may assume the fullness kCpbStart x bufsize at its IDR (what the buffering period SEI announces for the first one), its IDR takes at
    // most kIdrCpbShare of that, and its pictures together take at most kBudgetShare of what the channel delivers during the GOP — so the
    // level at the next IDR is at least the assumed one again (tests replay the produced sizes through the Annex C arrival / removal schedule).
    const double fps = (double)s->cfg.fps_num / s->cfg.fps_den;
    // a slice of a picture (one device of several) plans with its share of the picture's rate and buffer
    // (slices that share one rate plan — cfg.slice_halo — plan the whole picture's rate from inputs summed over the slices)
    const double share = grp ? 1.0 : (s->cfg.slice_count > 1 && s->cfg.rate_share_q16 > 0) ? s->cfg.rate_share_q16 / 65536.0 : 1.0;
    std::vector<int> qp_prev(gops, s->qp_p), gop_len(gops, 0);
    std::vector<double> budget(gops, 0.0);
    for (int g = 0; g < gops; g++) {
        gop_len[g] = glen[(size_t)g];
        budget[g] = kBudgetShare * share * s->cfg.vbv_maxrate_kbps * 1000.0 * gop_len[g] / fps;
    }
    const double cpb_idr_cap = s->cfg.vbv_bufsize_kbits > 0 ? kIdrCpbShare * kCpbStart * share * s->cfg.vbv_bufsize_kbits * 1000.0 : 1e30;
    const int p_slots = s->ring - 1;          // a P step's CABAC job is complete once its slot has been handed out again
    // output (display) index of the picture lane g codes at step j (cfg.bframes: steps are in decoding order)
    auto fidx = [&](int g, int j) { return (size_t)(first_index + gstart[(size_t)g] + pos_of_step(bf, j, glen[(size_t)g])); };
    // P-picture QP of lane g at step t.  Every input is deterministic: CABAC sizes only of pictures whose ring slot has been
    // reused (steps <= t - p_slots), device estimates of steps <= t - 2 (the step loop waits for that copy), a model for the
    // picture in flight.  The controller solves for the constant QP that spends the rest of the GOP budget and walks towards
    // it (+3 / -1 per picture, dead band 0.75): a constant QP is what the budget buys the most PSNR with.
    auto decide_p = [&](int g, int t) -> int {
        std::lock_guard<std::mutex> l(s->m);
        auto frame = [&](int j) -> mihevc_session::FrameRec & { return s->frames[fidx(g, j)]; };
        auto is_p = [&](int j) { return type_of_step(bf, j) == 1; };
        // CABAC / estimate ratio of this GOP's finished P / B pictures, seeded with two pictures' worth of the session ratio
        double sum_b = 0, sum_e = 0, seed = 0;
        for (int j = 1; j <= t - p_slots; j++)
            if (frame(j).bits >= 0 && frame(j).est_q4 > 0) { sum_b += (double)frame(j).bits; sum_e += (double)frame(j).est_q4 / 16.0; }
        for (int j = t - 2; j >= 1 && seed == 0; j--) if (frame(j).est_known) seed = 2.0 * (double)frame(j).est_q4 / 16.0;
        const double rp = (sum_e + seed) > 0 ? (sum_b + s->ratio_p * seed) / (sum_e + seed) : s->ratio_p;
        // reference point (q_ref, b_ref) of the rate model b(q) = b_ref * 2^((q_ref - q) / 6): the last two P estimates, or the
        // IDR picture scaled by the learned P/I ratio before any P estimate exists.  (cfg.bframes: P pictures only; a B picture is modelled as
        // beta_bp x a P picture at its QP - kQpB.)
        const auto &idr = frame(0);
        const double idr_bits = (double)idr.est_q4 / 16.0 * s->ratio_i;
        double b_ref = idr_bits * s->rho_pi, lg = 0;
        int q_ref = idr.qp, have = 0;
        for (int j = t - 2; j >= 1 && have < 2; j--) {
            if (!is_p(j) || !frame(j).est_known) continue;
            const double b = std::max(1.0, (double)frame(j).est_q4 / 16.0 * rp);
            if (!have) q_ref = frame(j).qp;
            lg += std::log2(b) + (frame(j).qp - q_ref) / 6.0;
            have++;
        }
        if (have) b_ref = std::exp2(lg / have);
        double spent = idr_bits;
        for (int j = 1; j < t; j++) {
            const auto &fr = frame(j);
            if (j <= t - p_slots && fr.bits >= 0) spent += (double)fr.bits;
            else if (j <= t - 2 && fr.est_known) spent += (double)fr.est_q4 / 16.0 * rp;
            else spent += (is_p(j) ? 1.0 : s->beta_bp) * b_ref * std::exp2((q_ref - (fr.qp - (is_p(j) ? 0 : kQpB))) / 6.0);
        }
        // what is left of the budget, shared by the pictures still to come in units of a P picture (a B picture counts beta_bp)
        double units = 0;
        for (int j = t; j < gop_len[g]; j++) units += is_p(j) ? 1.0 : s->beta_bp;
        double target = (budget[g] - spent) / std::max(0.5, units);
        target = std::max(target, 0.25 * budget[g] / gop_len[g]);
        const double q_ss = q_ref + 6.0 * std::log2(b_ref / target);
        int qp;
        if (t == 1) qp = std::max((int)std::lround(q_ss), idr.qp);    // first P: straight to the model, never finer than its IDR
        else {
            const double d = q_ss - qp_prev[g];
            qp = qp_prev[g] + (d >= 0.75 ? std::min(3, (int)std::lround(d)) : d <= -0.75 ? -1 : 0);
        }
        return std::min(std::max(qp, s->qp_p), 51);                   // the CRF is the quality ceiling, the VBV only raises QP
    };
    auto patch_qp = [&](int t, int g, int qp) {
        StepView<T> hv(ha, lay, t);
        hv.intra[g].prm = hv.inter[g].prm = hv.sao[g].prm = prm_for(qp);
        if (t > 0) { hv.intra[g].prm.tile_cols = s->ptiles.cols; hv.intra[g].prm.tile_rows = s->ptiles.rows; }
        std::lock_guard<std::mutex> l(s->m);
        auto &fr = s->frames[fidx(g, t)];
        fr.qp = qp; fr.type = type_of_step(bf, t);
    };
    // ---- lock-step over the GOPs ----
    hipEvent_t t_begin, t_end;
    HIPCK(s, hipEventCreate(&t_begin)); HIPCK(s, hipEventCreate(&t_end));
    HIPCK(s, hipEventRecord(t_begin, s->st_compute));
    const auto wall1 = std::chrono::steady_clock::now();
    auto mark = [&](int stage, int pictures, bool begin) -> int {       // bracket a stage with events when profiling
        if (!s->cfg.profile_stages || (s->cfg.profile_stages == 2 && stage != 2)) return 0;      // 2: the dominant stage (inter_ctu) only
        size_t need_ev = s->marks.size() * 2 + 2;
        while (s->ev_pool.size() < need_ev) { hipEvent_t e; HIPCK(s, hipEventCreate(&e)); s->ev_pool.push_back(e); }
        if (begin) { s->marks.push_back({stage, pictures, s->marks.size() * 2}); HIPCK(s, hipEventRecord(s->ev_pool[s->marks.back().ev], s->st_compute)); }
        else HIPCK(s, hipEventRecord(s->ev_pool[s->marks.back().ev + 1], s->st_compute));
        return 0;
    };
#define STAGE(idx, pics, call) do { if (int e_ = mark(idx, pics, true)) return e_; HIPCK(s, call); if (int e_ = mark(idx, pics, false)) return e_; } while (0)
    for (int t = 0; t < steps; t++) {
        const int B = batch[t];
        const int slot0 = slot_of(t);
        {   // the slot this step writes must have been drained by its previous CABAC jobs
            std::unique_lock<std::mutex> l(s->m);
            s->cv.wait(l, [&] { return s->jobs_open[slot0] == 0; });
        }
        StepView<T> dv(da, lay, t), hv(ha, lay, t);
        std::vector<int> qp_step(B), lane_slot(B, slot0);
        if (grp && s->rc_on && t - p_slots >= 1) {
            // the pictures of step t - p_slots have left the CABAC jobs of EVERY slice once all slices are here: their sizes summed over the slices
            const int j = t - p_slots;
            std::vector<double> v((size_t)batch[j]);
            {
                std::lock_guard<std::mutex> l(s->m);
                for (int g = 0; g < batch[j]; g++) v[(size_t)g] = (double)s->frames[fidx(g, j)].bits_local;
            }
            if (int e = group_sum(s, v)) return e;
            std::lock_guard<std::mutex> l(s->m);
            for (int g = 0; g < batch[j]; g++) s->frames[fidx(g, j)].bits = (long long)v[(size_t)g];
        }
        if (t >= 2) HIPCK(s, hipStreamWaitEvent(s->st_compute, s->ev_copy[slot_of(t - 2)], 0));      // the SSE pass of step t - 2 still reads the picture buffer this step reuses
        if (s->rc_on && t >= 3) {
            // rate feedback with a fixed lag of two steps: wait for the symbol copy of step t-2 (step t-1 is already queued behind
            // it, so the device never idles) and take its estimates.  A fixed lag makes the QP sequence reproducible.
            const int j = t - 2;
            HIPCK(s, hipEventSynchronize(s->ev_copy[slot_of(j)]));
            std::vector<double> v((size_t)batch[j]);
            for (int g = 0; g < batch[j]; g++) v[(size_t)g] = (double)*(const unsigned long long *)(s->lane[g].sym_host[slot_of(j)] + sl.est);
            if (int e = group_sum(s, v)) return e;          // slices with one rate plan: the picture's estimate is the sum over its slices
            {
                std::lock_guard<std::mutex> l(s->m);
                for (int g = 0; g < batch[j]; g++) {
                    auto &fr = s->frames[fidx(g, j)];
                    if (!fr.est_known) { fr.est_q4 = (unsigned long long)v[(size_t)g]; fr.est_known = true; }
                }
            }
        }
        for (int g = 0; g < B; g++) {
            // IDR pictures under rate control start where the last chunk's IDR pictures ended (first chunk: kIdrStart above the CRF's IDR QP)
            const int q_idr = !s->rc_on ? s->qp_i : std::min(51, std::max(s->qp_i, s->idr_qp_hint >= 0 ? s->idr_qp_hint : s->qp_i + kIdrStart));
            const int ptype = type_of_step(bf, t);
            // a B picture: the QP of the last anchor + kQpB (nothing predicts from it); it does not move the controller's walk
            qp_step[g] = t == 0 ? q_idr : ptype == 0 ? std::min(51, qp_prev[g] + kQpB) : (s->rc_on ? decide_p(g, t) : s->qp_p);
            patch_qp(t, g, qp_step[g]);
            if (ptype != 0) qp_prev[g] = qp_step[g];
        }
        if (t == 0) {
            HIPCK(s, hipMemcpyAsync(da + (size_t)t * lay.total, ha + (size_t)t * lay.total, lay.total, hipMemcpyHostToDevice, s->st_compute));
            for (int g = 0; g < B; g++) HIPCK(s, hipMemsetAsync(s->lane[g].sym_dev[0] + sl.sse, 0, 4 * sizeof(unsigned long long), s->st_compute));
            STAGE(0, B, launch_intra_picture<T>(s->st_compute, dv.intra, s->ctus_w, s->ctus_h, B, s->tiles.cols, s->tiles.rows, s->cfg.pre_search ? s->ev_args : nullptr, next_flow(s)));
            if (s->cfg.pre_search) {       // the chunk's search centres: beside the anti-diagonal chain, not beside k_intra_plan (both want the ALUs)
                HIPCK(s, hipStreamWaitEvent(s->st_pre, s->ev_args, 0));
                HIPCK(s, launch_pre_search_chunk<T>(s->st_pre, (const PreArgs<T> *)(da + flat_off), s->w, s->h, s->n_ctu, n_pre));
                HIPCK(s, hipEventRecord(s->ev_pre, s->st_pre));
            }
            if (s->rc_on) {
                // ONE analysis per IDR picture, then the rate model decides: IDR bits scale as 2^(-dQP/6) around the analysed point, P size
                // at the IDR's QP is rho x IDR size, the rest of the GOP budget is shared by the P pictures; wanted is the IDR QP whose
                // predicted steady P QP sits 3 above it (the usual I/P offset), never finer than the CRF asks, and whose picture fits the
                // share of the CPB an IDR may take.  Only lanes whose wanted QP is kIdrRedo or more away are analysed again, at that QP.
                HIPCK(s, hipStreamSynchronize(s->st_compute));
                std::vector<unsigned long long> ev((size_t)B);
                std::vector<int> qa(qp_step);                       // QP each lane's current analysis was made at
                for (int g = 0; g < B; g++) HIPCK(s, hipMemcpy(&ev[(size_t)g], s->lane[g].sym_dev[0] + sl.est, sizeof(unsigned long long), hipMemcpyDeviceToHost));
                if (grp) {
                    std::vector<double> v(ev.begin(), ev.end());
                    if (int e = group_sum(s, v)) return e;
                    for (int g = 0; g < B; g++) ev[(size_t)g] = (unsigned long long)v[(size_t)g];
                }
                auto want_for = [&](int g, double rho) {
                    const double ib_a = std::max(1.0, (double)ev[(size_t)g] / 16.0 * s->ratio_i);
                    int pick = 51;
                    double best_d = 1e30;
                    for (int q = s->qp_i; q <= 51; q++) {
                        const double ib = ib_a * std::exp2((qa[(size_t)g] - q) / 6.0), rest = budget[g] - ib;
                        double d;
                        if (ib > cpb_idr_cap && q < 51) continue;
                        if (gop_len[g] < 2) d = ib <= budget[g] ? -1e9 + q : 1e9 + ib;          // IDR-only GOP: finest that fits
                        else if (rest <= 0) d = 1e9 + ib;
                        else {
                            double units = 0;              // the GOP's other pictures in units of a P picture (cfg.bframes: a B picture counts beta_bp)
                            for (int j = 1; j < gop_len[g]; j++) units += type_of_step(bf, j) == 1 ? 1.0 : s->beta_bp;
                            const double q_ss = std::max((double)s->qp_p, q + 6.0 * std::log2(ib * rho / (rest / std::max(0.5, units))));
                            d = std::fabs(q_ss - (q + 3));
                        }
                        if (d < best_d) { best_d = d; pick = q; }
                    }
                    return pick;
                };
                std::vector<int> want((size_t)B);
                for (int g = 0; g < B; g++) want[(size_t)g] = want_for(g, s->rho_pi);
                if (!s->rho_measured && steps > 1 && batch[1] > 0) {
                    // First chunk of a session: rho is only a prior (1/16).  Measure it: analyse every GOP's first P picture once against
                    // the UNFILTERED reconstruction of the IDR analysis (copied + padded into the reference buffer the real step 0
                    // overwrites afterwards) at the wanted IDR QP + 3, read the estimate, and decide again.  Costs one P step per session.
                    const int B1 = batch[1], tb = steps;                 // trial block index
                    memcpy(ha + (size_t)tb * lay.total, ha + (size_t)1 * lay.total, lay.total);
                    StepView<T> tv(ha, lay, tb), dtv(da, lay, tb), h1(ha, lay, 1);
                    std::vector<int> qp_trial((size_t)B1);
                    for (int g = 0; g < B1; g++) {
                        mihevc_session::Lane &L = s->lane[g];
                        qp_trial[(size_t)g] = std::min(51, want[(size_t)g] + 3);
                        tv.sao[g] = hv.sao[g];
                        tv.sao[g].sao = nullptr; tv.sao[g].sse = nullptr; tv.sao[g].sse_ctu = nullptr; tv.sao[g].cu = nullptr;
                        tv.sao[g].halo_top = tv.sao[g].halo_bottom = 0;      // the trial predicts from this band's own unfiltered picture with a replicated border
                        tv.inter[g] = h1.inter[g];
                        for (int i = 0; i < 3; i++) tv.inter[g].rec[i] = mk<T>(L.rec_p[1][i], L.rec_stride[i]);
                        tv.inter[g].prm = prm_for(qp_trial[(size_t)g]);
                        tv.inter[g].ip = nullptr;
                        HIPCK(s, hipMemsetAsync(L.sym_dev[slot_of(1)] + sl.sse, 0, 4 * sizeof(unsigned long long), s->st_compute));
                    }
                    HIPCK(s, hipMemcpyAsync(da + (size_t)tb * lay.total, ha + (size_t)tb * lay.total, lay.total, hipMemcpyHostToDevice, s->st_compute));
                    HIPCK(s, launch_sao<T>(s->st_compute, dtv.sao, s->w, s->h, B1, false));
                    HIPCK(s, launch_pad<T>(s->st_compute, dtv.sao, s->w, s->h, B1));
                    if (s->cfg.pre_search) HIPCK(s, hipStreamWaitEvent(s->st_compute, s->ev_pre, 0));      // the chunk's search centres
                    HIPCK(s, launch_me_search<T>(s->st_compute, dtv.inter, s->n_ctu, B1, s->me_range));
                    HIPCK(s, launch_inter_ctu<T>(s->st_compute, dtv.inter, s->n_ctu, B1, s->me_range));
                    HIPCK(s, hipStreamSynchronize(s->st_compute));
                    double lg = 0;
                    int nl = 0;
                    std::vector<double> epv((size_t)B1, 0.0);
                    for (int g = 0; g < B1; g++) {
                        unsigned long long ep = 0;
                        HIPCK(s, hipMemcpy(&ep, s->lane[g].sym_dev[slot_of(1)] + sl.est, sizeof ep, hipMemcpyDeviceToHost));
                        epv[(size_t)g] = (double)ep;
                    }
                    if (int e = group_sum(s, epv)) return e;
                    for (int g = 0; g < B1; g++) {
                        const unsigned long long ep = (unsigned long long)epv[(size_t)g];
                        if (!ep || !ev[(size_t)g]) continue;
                        // both estimates brought to one QP: the P picture from its trial QP, the IDR picture from the QP it was analysed at
                        lg += std::log2((double)ep / (double)ev[(size_t)g]) + (qp_trial[(size_t)g] - qa[(size_t)g]) / 6.0;
                        nl++;
                    }
                    if (nl) s->rho_pi = std::min(1.0, std::max(1.0 / 256, std::exp2(lg / nl)));
                    s->rho_measured = true;
                    for (int g = 0; g < B; g++) want[(size_t)g] = want_for(g, s->rho_pi);
                }
                std::vector<int> redo;
                for (int g = 0; g < B; g++)
                    if (std::abs(want[(size_t)g] - qa[(size_t)g]) >= kIdrRedo) redo.push_back(g);
                if (!redo.empty()) {
                    // second analysis of those lanes at the wanted QP into the same buffers (every CTU, record and non-zero TU is rewritten;
                    // levels of TUs that are zero now are never read by the entropy coder): argument blocks compacted behind the first B
                    for (size_t k = 0; k < redo.size(); k++) {
                        const int g = redo[k];
                        qp_step[g] = want[(size_t)g];
                        patch_qp(t, g, qp_step[g]);
                        hv.intra[B + (int)k] = hv.intra[g];
                        HIPCK(s, hipMemsetAsync(s->lane[g].sym_dev[0] + sl.sse, 0, 4 * sizeof(unsigned long long), s->st_compute));
                    }
                    HIPCK(s, hipMemcpyAsync(da + (size_t)t * lay.total, ha + (size_t)t * lay.total, lay.total, hipMemcpyHostToDevice, s->st_compute));
                    STAGE(0, (int)redo.size(), launch_intra_picture<T>(s->st_compute, dv.intra + B, s->ctus_w, s->ctus_h, (int)redo.size(), s->tiles.cols, s->tiles.rows, nullptr, next_flow(s)));
                    HIPCK(s, hipStreamSynchronize(s->st_compute));
                    std::vector<double> v(redo.size(), 0.0);
                    for (size_t k = 0; k < redo.size(); k++) {
                        unsigned long long e2 = 0;
                        HIPCK(s, hipMemcpy(&e2, s->lane[redo[k]].sym_dev[0] + sl.est, sizeof e2, hipMemcpyDeviceToHost));
                        v[k] = (double)e2;
                    }
                    if (int e = group_sum(s, v)) return e;
                    for (size_t k = 0; k < redo.size(); k++) ev[(size_t)redo[k]] = (unsigned long long)v[k];
                }
                int sum_q = 0;
                for (int g = 0; g < B; g++) {
                    qp_prev[g] = qp_step[g];
                    sum_q += qp_step[g];
                    std::lock_guard<std::mutex> l(s->m);
                    auto &fr = s->frames[(size_t)(first_index + gstart[(size_t)g])];
                    fr.est_q4 = ev[(size_t)g]; fr.est_known = true;
                }
                s->idr_qp_hint = (sum_q + B / 2) / B;
            }
        } else {
            {   // the step's QPs reach the device inside one tiny launch that also zeroes the slot's SSE + estimate accumulators; everything
                // else in the step's argument block went up with the chunk
                // The same launch pads the border of the previous step's pictures (nothing before this step's searches reads it) and makes the 1/4-size
                // pictures: one launch boundary on the compute stream instead of three (~6 us each, profiles/r02_e: kernel time 733 of 797 us per step).
                StepParams sp{};
                sp.p_tile_cols = s->ptiles.cols; sp.p_tile_rows = s->ptiles.rows;
                for (int g = 0; g < B; g++) sp.prm[g] = hv.inter[g].prm;
                StepView<T> pv(da, lay, t - 1);
                if (grp && reach > 0) {
                    // X2: the final reconstruction either side of the seams, straight out of the neighbours' pictures of the previous step, into the border
                    // rows of this band's reference pictures (the pad below fills in their left / right ends and whatever lies beyond the whole picture)
                    const long long G = s->gstep + t;
                    for (auto *v : {&reach_up, &reach_dn})
                        for (auto &br : *v) {
                            if (!s->group->wait_for(br.first, 2, G - 1)) { s->failed = true; s->err = "another slice of the picture failed"; return MIHEVC_EDEVICE; }
                            HIPCK(s, hipStreamWaitEvent(s->st_compute, s->group->pub(br.first).ev_x2[(G - 1) & 1], 0));
                        }
                    HIPCK(s, launch_copy_rows(s->st_compute, (const RowCopy *)(dj + (size_t)t * jl.total + jl.pull), B * 3 * reach, 16));
                }
                HIPCK(s, launch_prep_p_step<T>(s->st_compute, pv.sao, (const PreArgs<T> *)nullptr, dv.intra, dv.inter, dv.sao, sp, s->w, s->h, B));
                if (t == 1 && s->cfg.pre_search) HIPCK(s, hipStreamWaitEvent(s->st_compute, s->ev_pre, 0));      // the chunk's search centres (st_pre, under the IDR step)
            }
            // stage 1 = the integer search around the chunk's search centres (a B picture: against both anchors)
            const bool bstep = type_of_step(bf, t) == 0;
            if (int e_ = mark(1, B, true)) return e_;
            HIPCK(s, launch_me_search<T>(s->st_compute, dv.inter, s->n_ctu, B, s->me_range, 0));
            if (bstep) HIPCK(s, launch_me_search<T>(s->st_compute, dv.inter, s->n_ctu, B, s->me_range, 1));
            if (int e_ = mark(1, B, false)) return e_;
            if (bstep) STAGE(2, B, launch_inter_ctu_b<T>(s->st_compute, dv.inter, s->n_ctu, B, s->me_range));
            else STAGE(2, B, launch_inter_ctu<T>(s->st_compute, dv.inter, s->n_ctu, B, s->me_range));
            if (s->cfg.intra_in_p && !bstep) STAGE(7, B, launch_intra_p<T>(s->st_compute, dv.intra, s->n_ctu, B));
        }
        if (grp) {
            // X1: kSeamRows rows of the pre-deblock reconstruction + one row of CU records either side of every seam.  Every band puts its own first and last
            // rows where its neighbours can read them (the band's picture is deblocked in place right after), then takes the neighbours'
            const long long G = s->gstep + t;
            HIPCK(s, launch_copy_rows(s->st_compute, (const RowCopy *)(dj + (size_t)t * jl.total + jl.exp), B * 8, 8));
            HIPCK(s, hipEventRecord(s->ev_x1[G & 1], s->st_compute));
            s->group->announce(s->band, 1, G);
            for (auto *v : {&reach_up, &reach_dn})
                for (auto &br : *v) {
                    if (!s->group->wait_for(br.first, 1, G)) { s->failed = true; s->err = "another slice of the picture failed"; return MIHEVC_EDEVICE; }
                    HIPCK(s, hipStreamWaitEvent(s->st_compute, s->group->pub(br.first).ev_x1[G & 1], 0));
                }
            if (up + dn) HIPCK(s, launch_copy_rows(s->st_compute, (const RowCopy *)(dj + (size_t)t * jl.total + jl.imp), B * 8, 8));
        }
        if (!s->cfg.sao) STAGE(3, B, launch_deblock<T>(s->st_compute, dv.dbk_v, dv.dbk_h, s->w, s->h + (grp ? kSeamRows * (up + dn) : 0), B));
        STAGE(4, B, launch_sao<T>(s->st_compute, dv.sao, s->w, s->h, B, s->cfg.sao != 0));
        if (grp) {
            const long long G = s->gstep + t;
            HIPCK(s, hipEventRecord(s->ev_x2[G & 1], s->st_compute));
            s->group->announce(s->band, 2, G);
        }
        HIPCK(s, hipEventRecord(s->ev_compute[slot0], s->st_compute));      // (the border pad of these pictures is part of the next step's first launch)
        HIPCK(s, hipStreamWaitEvent(s->st_copy, s->ev_compute[slot0], 0));
        // SSE (statistics only): the SAO programs left every CTU's squared error in the symbol block's device tail; one small launch on the copy stream, in
        // front of the symbol copies that carry its sums, adds them up.  (Until round 3 a pass of its own re-read source and reconstruction here: 7 MB per
        // picture and 25 us per step beside the compute stream.)  Without SAO that pass still runs: k_sao_apply is a plain copy and has no source.
        if (s->cfg.sao) HIPCK(s, launch_sse_fold<T>(s->st_copy, dv.sao, s->n_ctu, B));
        else HIPCK(s, launch_frame_sse<T>(s->st_copy, dv.sao, B));
        for (int g = 0; g < B; g++)
        {   // CU records, then SAO parameters + SSE + rate estimate (the level planes were written to the host block directly)
            uint8_t *hd = s->lane[g].sym_host[lane_slot[g]], *dd = s->lane[g].sym_dev[lane_slot[g]];
            HIPCK(s, hipMemcpyAsync(hd + sl.cu, dd + sl.cu, sl.cu_bytes, hipMemcpyDeviceToHost, s->st_copy));
            HIPCK(s, hipMemcpyAsync(hd + sl.sao, dd + sl.sao, sl.total - sl.sao, hipMemcpyDeviceToHost, s->st_copy));
        }
        if (s->keep_recon) {
            for (int g = 0; g < B; g++) {
                std::vector<uint16_t> &dst = s->recon[(int64_t)fidx(g, t)];
                const int rbuf = !bf ? (t & 1) : type_of_step(bf, t) == 0 ? 2 : ((t + 1) / 2) & 1;
                dst.assign((size_t)s->w * s->h * 3 / 2, 0);
                std::vector<uint8_t> tmp((size_t)s->w * s->h * 3 / 2 * esize(s));
                size_t off = 0;
                HIPCK(s, hipStreamSynchronize(s->st_compute));
                for (int i = 0; i < 3; i++) {
                    int pw = i ? s->w / 2 : s->w, ph = i ? s->h / 2 : s->h;
                    HIPCK(s, hipMemcpy2D(tmp.data() + off * esize(s), pw * esize(s), s->lane[g].rec_p[rbuf][i], s->lane[g].rec_stride[i] * esize(s),
                                         pw * esize(s), ph, hipMemcpyDeviceToHost));
                    off += (size_t)pw * ph;
                }
                for (size_t k = 0; k < dst.size(); k++) dst[k] = s->is16 ? ((uint16_t *)tmp.data())[k] : tmp[k];
            }
        }
        HIPCK(s, hipEventRecord(s->ev_copy[slot0], s->st_copy));
        {
            std::lock_guard<std::mutex> l(s->m);
            for (int g = 0; g < B; g++) s->jobs_open[lane_slot[g]]++;
        }
        // host jobs per picture: the pool's threads shared by the pictures of this step (their tiles, when the picture has several: cfg.p_tiles / IDR grid)
        const int parts_wanted = std::max(1, s->host_threads / std::max(1, B));
        for (int g = 0; g < B; g++) {
            const int pos = pos_of_step(bf, t, glen[(size_t)g]);
            PictureJob *j = new PictureJob();
            j->s = s; j->slot = lane_slot[g]; j->lane_i = g; j->index = (int64_t)fidx(g, t); j->pts = s->pending[gstart[(size_t)g] + pos].pts;
            // packets leave in DECODING order: place t of the GOP; dts = the pts of the frame at that place in display order, one frame earlier when
            // B pictures reorder (an anchor is decoded one picture before the B picture in front of it is shown)
            j->dec_index = first_index + gstart[(size_t)g] + t;
            j->reorder = s->cfg.bframes != 0;
            j->dts = s->pending[gstart[(size_t)g] + t].pts - (j->reorder ? s->pts_step : 0);
            j->pic.ref_dist = type_of_step(bf, t) == 1 ? pos - std::max(0, pos_of_step(bf, bf ? std::max(0, t - 2) * (t > 1) : t - 1, glen[(size_t)g])) : 0;
            j->dec_pos = t;
            j->prev_gop_len = prev_len[(size_t)g];
            j->slice_type = type_of_step(bf, t); j->poc = pos; j->qp = qp_step[g]; j->first_of_stream = j->dec_index == 0;
            j->pic.slice_type = j->slice_type; j->pic.poc = pos; j->pic.qp = j->qp;
            picture_symbols(s, j->slot, g, j->pic);
            j->n_tiles = picture_tiles(s->cfg, j->pic);
            j->parts = std::min(j->n_tiles, parts_wanted);
            j->sub.resize((size_t)j->n_tiles);
            j->left.store(j->parts);
            hipEvent_t ev = s->ev_copy[j->slot];
            for (int part = 0; part < j->parts; part++)
                s->pool->submit([s, j, part, ev] {
                    (void)hipSetDevice(s->device);          // worker threads start on device 0: wait on the event in its own device's context
                    (void)hipEventSynchronize(ev);
                    entropy_part(j, part);
                });
        }
    }
#undef STAGE
    HIPCK(s, hipEventRecord(t_end, s->st_compute));
    HIPCK(s, hipStreamSynchronize(s->st_compute));
    const auto wall2 = std::chrono::steady_clock::now();
    HIPCK(s, hipStreamSynchronize(s->st_copy));
    HIPCK(s, hipStreamSynchronize(s->st_pre));        // a chunk without P steps never waited for its pre-search: its buffers are reused by the next chunk
    if (s->d_flow) {      // a dataflow wait that gave up (a bug: the pictures are garbage, never hand them out)
        int bad = 0;
        HIPCK(s, hipMemcpy(&bad, (uint8_t *)s->d_flow + s->flow_bytes - sizeof(int), sizeof bad, hipMemcpyDeviceToHost));
        if (bad) { s->failed = true; s->err = "k_intra_flow: a CTU waited for a neighbour that never finished"; return MIHEVC_EDEVICE; }
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, t_begin, t_end);
    s->stats.device_ms += ms;
    for (auto &mk : s->marks) {
        float e = 0;
        if (hipEventElapsedTime(&e, s->ev_pool[mk.ev], s->ev_pool[mk.ev + 1]) == hipSuccess) {
            s->stats.stage_ms[mk.stage] += e; s->stats.stage_launches[mk.stage]++; s->stats.stage_pictures[mk.stage] += mk.pictures;
        }
    }
    s->marks.clear();
    (void)hipEventDestroy(t_begin); (void)hipEventDestroy(t_end);
    {   // all CABAC jobs of the chunk
        std::unique_lock<std::mutex> l(s->m);
        s->cv.wait(l, [&] { int n = 0; for (int k = 0; k < kRing; k++) n += s->jobs_open[k]; return n == 0; });
    }
    s->gstep += steps;
    s->chunk_no++;
    if (grp && s->rc_on) {        // the sizes of every picture of the chunk, summed over the slices (estimates were summed step by step)
        std::vector<double> v((size_t)2 * n);
        {
            std::lock_guard<std::mutex> l(s->m);
            for (int i = 0; i < n; i++) { const auto &fr = s->frames[(size_t)(first_index + i)]; v[(size_t)i] = (double)fr.bits_local; v[(size_t)(n + i)] = (double)fr.est_local; }
        }
        if (int e = group_sum(s, v)) return e;
        std::lock_guard<std::mutex> l(s->m);
        for (int i = 0; i < n; i++) { auto &fr = s->frames[(size_t)(first_index + i)]; fr.bits = (long long)v[(size_t)i]; fr.est_q4 = (unsigned long long)v[(size_t)(n + i)]; fr.est_known = true; }
    } else if (grp) {
        if (!s->group->barrier()) { s->failed = true; return MIHEVC_EDEVICE; }      // no band leaves a chunk (and reuses its buffers) while another still reads them
    }
    if (s->rc_on) {
        // learn from the finished chunk (all CABAC sizes are known now, so this is deterministic): CABAC bits per estimated
        // bit for I and P pictures, and the P/I size ratio at equal QP
        std::lock_guard<std::mutex> l(s->m);
        double bi = 0, ei = 0, bp = 0, ep = 0, lg = 0, lgb = 0;
        int np = 0, nb = 0;
        for (int g = 0; g < gops; g++) {
            const auto &idr = s->frames[(size_t)(first_index + gstart[(size_t)g])];
            if (idr.bits < 0 || !idr.est_q4) continue;
            bi += (double)idr.bits; ei += (double)idr.est_q4 / 16.0;
            for (int j = 1; j < gop_len[g]; j++) {          // steps: decoding order
                const auto &fr = s->frames[fidx(g, j)];
                if (fr.bits <= 0 || !fr.est_q4) continue;
                bp += (double)fr.bits; ep += (double)fr.est_q4 / 16.0;
                if (type_of_step(bf, j) == 1) { lg += std::log2((double)fr.bits / (double)idr.bits) + (fr.qp - idr.qp) / 6.0; np++; }
                else { lgb += std::log2((double)fr.bits / (double)idr.bits) + (fr.qp - kQpB - idr.qp) / 6.0; nb++; }
            }
        }
        if (ei > 0) s->ratio_i = 0.5 * s->ratio_i + 0.5 * bi / ei;
        if (ep > 0) s->ratio_p = 0.5 * s->ratio_p + 0.5 * bp / ep;
        if (np) s->rho_pi = std::min(1.0, std::max(1.0 / 256, 0.5 * s->rho_pi + 0.5 * std::exp2(lg / np)));
        // a B picture at QP + kQpB against a P picture at QP (both brought to the IDR picture's QP through the 2^(-dQP/6) rule)
        if (np && nb) s->beta_bp = std::min(1.5, std::max(0.05, 0.5 * s->beta_bp + 0.5 * std::exp2(lgb / nb - lg / np)));
    }
    for (auto &src : s->pending) if (!src.borrowed) s->free_src.push_back(src);
    s->pending.clear();
    {
        const auto wall3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (int32_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        s->stats.reserved[3] += us(wall0, wall1); s->stats.reserved[4] += us(wall2, wall3); s->stats.reserved[5] += us(wall0, wall3);
    }
    return 0;
}

int run_chunk(mihevc_session *s)
{
    if (hipSetDevice(s->device) != hipSuccess) return MIHEVC_EDEVICE;
    if (s->up_pending) {       // frames still on their way up (mihevc_send_frame_async, device-to-device copies): everything the chunk launches comes behind them
        HIPCK(s, hipEventRecord(s->ev_up, s->st_pre));
        HIPCK(s, hipStreamWaitEvent(s->st_compute, s->ev_up, 0));
        HIPCK(s, hipStreamWaitEvent(s->st_pre, s->ev_up, 0));
        s->up_pending = false;
    }
    return s->is16 ? encode_chunk<uint16_t>(s) : encode_chunk<uint8_t>(s);
}

}  // namespace

extern "C" {

int mihevc_open(const mihevc_config *cfg, int device, mihevc_session **out)
{
    if (!cfg || !out) return MIHEVC_EINVAL;
    *out = nullptr;
    if (cfg->width < 16 || cfg->height < 16 || (cfg->width & 1) || (cfg->height & 1) || cfg->width > 8192 || cfg->height > 4352) return MIHEVC_EINVAL;
    if (cfg->bit_depth != 8 && cfg->bit_depth != 10) return MIHEVC_EINVAL;
    if (cfg->fps_num <= 0 || cfg->fps_den <= 0 || cfg->keyint < 1 || cfg->keyint > 240) return MIHEVC_EINVAL;
    if (cfg->bframes < -1 || cfg->bframes > 1 || (cfg->bframes && cfg->slice_count > 1)) return MIHEVC_EINVAL;      // B pictures: whole pictures only (for now)
    if (cfg->slice_count > 1) {        // one slice of a picture: a band of whole CTU rows (the last band takes the picture's remainder)
        if (cfg->slice_count > 16 || cfg->slice_index < 0 || cfg->slice_index >= cfg->slice_count || cfg->pic_height < cfg->height) return MIHEVC_EINVAL;
        int rows = 0;
        for (int k = 0; k < cfg->slice_count; k++) { if (cfg->slice_ctu_rows[k] < 1) return MIHEVC_EINVAL; rows += cfg->slice_ctu_rows[k]; }
        if (rows != (cfg->pic_height + 31) / 32) return MIHEVC_EINVAL;
        const int y0 = 32 * slice_first_row(*cfg, cfg->slice_index), y1 = std::min(cfg->pic_height, y0 + 32 * cfg->slice_ctu_rows[cfg->slice_index]);
        if (cfg->height != y1 - y0) return MIHEVC_EINVAL;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MIHEVC_ENODEV;
    if (device < 0 || device >= n) return MIHEVC_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MIHEVC_EDEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MIHEVC_ENODEV;      // the code objects are gfx950 only
    if (hipSetDevice(device) != hipSuccess) return MIHEVC_EDEVICE;
    mihevc_session *s = new (std::nothrow) mihevc_session();
    if (!s) return MIHEVC_ENOMEM;
    s->cfg = *cfg;
    if (s->cfg.sao < 0) s->cfg.sao = 1;
    s->device = device;
    CodedSize cs = coded_size(cfg->width, cfg->height);
    s->w = cs.w; s->h = cs.h;
    s->ctus_w = (s->w + CTU - 1) / CTU; s->ctus_h = (s->h + CTU - 1) / CTU; s->n_ctu = s->ctus_w * s->ctus_h;
    s->tiles = tile_grid(s->cfg);
    s->ptiles = p_tile_grid(s->cfg);
    s->ring = cfg->level_idc >= 150 ? kRing : 8;
    s->is16 = cfg->bit_depth > 8;
    s->keyint = cfg->keyint;
    s->lanes = cfg->gops_in_flight > 0 ? std::min(cfg->gops_in_flight, MAX_LANES) : 4;
    s->me_range = cfg->me_range > 0 ? std::min(cfg->me_range, MAX_RANGE) : 15;   // 8 quads x 31 rows = 248 items: one pass of the 256-thread search
    // constant-quality operating point: P pictures at crf + 2, IDR pictures 3 below (x265's ipratio 1.4 ~ 3 QP)
    s->qp_p = cfg->qp >= 0 ? cfg->qp : std::min(51, std::max(0, cfg->crf + 2));
    s->qp_i = std::max(0, s->qp_p - 3);
    s->stats.last_qp = s->qp_p;
    s->rc_on = cfg->qp < 0 && cfg->vbv_maxrate_kbps > 0;
    write_parameter_sets(s->cfg, s->headers);
    bool ok = StreamCache::get().acquire(s->device, &s->st_compute) == hipSuccess && StreamCache::get().acquire(s->device, &s->st_copy) == hipSuccess &&
              StreamCache::get().acquire(s->device, &s->st_pre) == hipSuccess;
    for (int i = 0; ok && i < kRing; i++)
        ok = hipEventCreateWithFlags(&s->ev_compute[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s->ev_copy[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s->ev_pre, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s->ev_args, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s->ev_up, hipEventDisableTiming) == hipSuccess;
    if (!ok) {               // give back what was acquired (event handles of the slots never reached stay null)
        for (int i = 0; i < kRing; i++) { if (s->ev_compute[i]) (void)hipEventDestroy(s->ev_compute[i]); if (s->ev_copy[i]) (void)hipEventDestroy(s->ev_copy[i]); }
        if (s->ev_pre) (void)hipEventDestroy(s->ev_pre);
        if (s->ev_args) (void)hipEventDestroy(s->ev_args);
        if (s->ev_up) (void)hipEventDestroy(s->ev_up);
        StreamCache::get().release(s->device, s->st_compute);
        StreamCache::get().release(s->device, s->st_copy);
        StreamCache::get().release(s->device, s->st_pre);
        delete s;
        return MIHEVC_EDEVICE;
    }
    if (cfg->slice_count > 1 && cfg->slice_halo) {
        // one slice of a picture whose slices exchange rows: meet the others (csrc/slice_group.h), and get the buffers the neighbours read
        s->n_bands = cfg->slice_count; s->band = cfg->slice_index;
        for (int k = 0, y0 = 0; k < cfg->slice_count; k++) {
            const int y1 = std::min(cfg->pic_height, y0 + 32 * cfg->slice_ctu_rows[k]);
            s->band_h[k] = coded_size(cfg->width, y1 - y0).h;
            y0 = y1;
        }
        const size_t es = s->is16 ? 2 : 1;
        s->x1_part_bytes = (((size_t)kSeamRows * s->w + (size_t)kSeamRows * (s->w / 2)) * es + (size_t)(s->w >> 3) * sizeof(mihevc_cu_rec) + 255) & ~(size_t)255;
        s->x1_lane_bytes = 2 * s->x1_part_bytes;
        s->x1_bytes = s->x1_lane_bytes * kHaloLanes;
        for (int k = 0; ok && k < 2; k++)
            ok = BufferCache::get().alloc(s->device, s->x1_bytes, false, &s->x1_export[k]) == hipSuccess &&
                 hipEventCreateWithFlags(&s->ev_x1[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s->ev_x2[k], hipEventDisableTiming) == hipSuccess;
        if (!ok || cfg->slice_group == 0) { mihevc_close(s); return cfg->slice_group == 0 ? MIHEVC_EINVAL : MIHEVC_EDEVICE; }
        s->group = SliceGroup::join(cfg->slice_group, cfg->slice_count);
    }
    int threads = cfg->host_threads > 0 ? cfg->host_threads : (int)std::min(16u, std::max(2u, std::thread::hardware_concurrency()));
    s->pool = &ThreadPool::shared(threads);
    s->host_threads = threads;
    *out = s;
    return MIHEVC_OK;
}

static int ingest(mihevc_session *s, const void *y, const void *u, const void *v, int pitch_y, int pitch_c, int64_t pts, bool device_src, bool async)
{
    if (!s || !y || !u || !v) return MIHEVC_EINVAL;
    if (s->failed) return MIHEVC_EDEVICE;
    if (s->flushed) return MIHEVC_ESTATE;
    if (hipSetDevice(s->device) != hipSuccess) return MIHEVC_EDEVICE;
    mihevc_session::Src src;
    const void *in[3] = {y, u, v};
    const size_t es = esize(s);
    // Device planes whose size already is the coded size (no margin to fill) are used where they are: the header's contract keeps them valid and
    // unmodified until the picture's packet is out.  Saves three 2-D copies per frame (4 % of the device time of a 1080p clip, and most of the
    // wall time of handing 300 frames over).
    if (device_src && s->cfg.width == s->w && s->cfg.height == s->h && pitch_y >= s->w && pitch_c >= s->w / 2 &&
        ((uintptr_t)y & 3) == 0 && ((uintptr_t)u & 3) == 0 && ((uintptr_t)v & 3) == 0 && (pitch_y * es) % 4 == 0 && (pitch_c * es) % 4 == 0) {
        for (int i = 0; i < 3; i++) { src.base[i] = nullptr; src.p[i] = const_cast<void *>(in[i]); src.stride[i] = i ? pitch_c : pitch_y; }
        src.pts = pts; src.borrowed = true;
        if (s->frames_in == 0) s->first_pts = pts; else if (s->frames_in == 1) s->pts_step = std::max<int64_t>(1, pts - s->first_pts);
        s->pending.push_back(src);
        s->frames_in++;
        s->stats.frames_in = s->frames_in;
        if ((int)s->pending.size() >= s->lanes * s->keyint) return run_chunk(s);
        return MIHEVC_OK;
    }
    if (int e = get_src(s, src)) return e;
    src.pts = pts; src.borrowed = false;
    // uploads run on a stream of their own; the chunk's first launch waits for the event behind the last one.  The synchronous entry point waits
    // here (the caller may reuse its buffers on return), the asynchronous one returns with the copies in flight
    for (int i = 0; i < 3; i++) {
        int pw = i ? s->w / 2 : s->w, ph = i ? s->h / 2 : s->h;               // coded plane size
        int sw = i ? s->cfg.width / 2 : s->cfg.width, sh = i ? s->cfg.height / 2 : s->cfg.height, pitch = i ? pitch_c : pitch_y;
        if (pitch < sw) return MIHEVC_EINVAL;
        HIPCK(s, hipMemcpy2DAsync(src.p[i], src.stride[i] * es, in[i], pitch * es, sw * es, sh, device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st_pre));
        // replicate the last column/row into the coded-size margin (the conformance window crops it again)
        if (pw > sw || ph > sh) {
            if (s->is16) HIPCK(s, launch_extend_margin<uint16_t>(s->st_pre, Plane<uint16_t>{(uint16_t *)src.p[i], src.stride[i]}, sw, sh, pw, ph));
            else HIPCK(s, launch_extend_margin<uint8_t>(s->st_pre, Plane<uint8_t>{(uint8_t *)src.p[i], src.stride[i]}, sw, sh, pw, ph));
        }
    }
    if (!device_src && !async) HIPCK(s, hipStreamSynchronize(s->st_pre));     // caller's buffers may be reused on return
    else s->up_pending = true;
    if (s->frames_in == 0) s->first_pts = pts; else if (s->frames_in == 1) s->pts_step = std::max<int64_t>(1, pts - s->first_pts);
    s->pending.push_back(src);
    s->frames_in++;
    s->stats.frames_in = s->frames_in;
    if ((int)s->pending.size() >= s->lanes * s->keyint) return run_chunk(s);
    return MIHEVC_OK;
}

int mihevc_send_frame(mihevc_session *s, const void *y, const void *u, const void *v, int pitch_y, int pitch_c, int64_t pts)
{
    return ingest(s, y, u, v, pitch_y, pitch_c, pts, false, false);
}
int mihevc_send_frame_async(mihevc_session *s, const void *y, const void *u, const void *v, int pitch_y, int pitch_c, int64_t pts)
{
    return ingest(s, y, u, v, pitch_y, pitch_c, pts, false, true);
}
int mihevc_send_frame_device(mihevc_session *s, const void *y, const void *u, const void *v, int pitch_y, int pitch_c, int64_t pts)
{
    return ingest(s, y, u, v, pitch_y, pitch_c, pts, true, false);
}
int mihevc_send_frames_device(mihevc_session *s, int n, const void *const *y, const void *const *u, const void *const *v, int pitch_y, int pitch_c, int64_t first_pts)
{
    if (!s || n < 0 || (n && (!y || !u || !v))) return MIHEVC_EINVAL;
    for (int i = 0; i < n; i++)
        if (int e = ingest(s, y[i], u[i], v[i], pitch_y, pitch_c, first_pts + i, true, false)) return e;
    return MIHEVC_OK;
}
int mihevc_sync_uploads(mihevc_session *s)
{
    if (!s) return MIHEVC_EINVAL;
    if (s->failed) return MIHEVC_EDEVICE;
    if (hipSetDevice(s->device) != hipSuccess) return MIHEVC_EDEVICE;
    HIPCK(s, hipStreamSynchronize(s->st_pre));
    s->up_pending = false;
    return MIHEVC_OK;
}

int mihevc_flush(mihevc_session *s)
{
    if (!s) return MIHEVC_EINVAL;
    if (s->failed) return MIHEVC_EDEVICE;
    if (s->flushed) return MIHEVC_OK;
    s->flushing = true;
    int e = run_chunk(s);
    s->flushed = true;
    return e;
}

int mihevc_abort(mihevc_session *s)
{
    if (!s) return MIHEVC_EINVAL;
    s->failed = true;
    if (s->err.empty()) s->err = "aborted by the caller";
    if (s->group) s->group->fail();
    return MIHEVC_OK;
}

int mihevc_receive_packet(mihevc_session *s, const uint8_t **data, size_t *size, int64_t *pts, int64_t *dts, int *keyframe)
{
    if (!s || !data || !size) return MIHEVC_EINVAL;
    std::lock_guard<std::mutex> l(s->m);
    auto it = s->packets.find(s->next_out);
    if (it == s->packets.end() || !it->second.ready) return (s->flushed && s->next_out >= s->frames_in) ? MIHEVC_EOF : MIHEVC_EAGAIN;
    s->cur_packet = std::move(it->second.data);
    if (pts) *pts = it->second.pts;
    if (dts) *dts = it->second.dts;        // packets come in decoding order; dts < pts only with B pictures (cfg.bframes)
    if (keyframe) *keyframe = it->second.key;
    s->packets.erase(it);
    s->next_out++;
    s->stats.frames_out = s->next_out;
    *data = s->cur_packet.data();
    *size = s->cur_packet.size();
    return MIHEVC_OK;
}

int mihevc_get_headers(mihevc_session *s, const uint8_t **data, size_t *size)
{
    if (!s || !data || !size) return MIHEVC_EINVAL;
    *data = s->headers.data();
    *size = s->headers.size();
    return MIHEVC_OK;
}

int mihevc_get_stats(const mihevc_session *s, mihevc_stats *out)
{
    if (!s || !out) return MIHEVC_EINVAL;
    std::lock_guard<std::mutex> l(const_cast<mihevc_session *>(s)->m);
    *out = s->stats;
    out->entropy_ms = (double)s->entropy_ns.load() / 1e6;
    return MIHEVC_OK;
}

int mihevc_set_keep_recon(mihevc_session *s, int keep)
{
    if (!s) return MIHEVC_EINVAL;
    s->keep_recon = keep != 0;
    return MIHEVC_OK;
}

int mihevc_get_recon(mihevc_session *s, int64_t index, uint16_t *y, uint16_t *u, uint16_t *v)
{
    if (!s || !y || !u || !v) return MIHEVC_EINVAL;
    auto it = s->recon.find(index);
    if (it == s->recon.end()) return MIHEVC_ESTATE;
    size_t ny = (size_t)s->w * s->h;
    memcpy(y, it->second.data(), ny * 2);
    memcpy(u, it->second.data() + ny, ny / 2);
    memcpy(v, it->second.data() + ny + ny / 4, ny / 2);
    return MIHEVC_OK;
}

int mihevc_get_frame_info(mihevc_session *s, int64_t index, int *qp, int *slice_type, int64_t *bits)
{
    if (!s) return MIHEVC_EINVAL;
    std::lock_guard<std::mutex> l(s->m);
    if (index < 0 || (size_t)index >= s->frames.size()) return MIHEVC_ESTATE;
    const auto &fr = s->frames[(size_t)index];
    if (qp) *qp = fr.qp;
    if (slice_type) *slice_type = fr.type;
    if (bits) *bits = fr.bits;
    return MIHEVC_OK;
}

int mihevc_coded_size(const mihevc_session *s, int *w, int *h)
{
    if (!s || !w || !h) return MIHEVC_EINVAL;
    *w = s->w; *h = s->h;
    return MIHEVC_OK;
}

const char *mihevc_last_error(const mihevc_session *s) { return s ? s->err.c_str() : "null session"; }

void mihevc_close(mihevc_session *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    {   // the session's CABAC jobs still in flight (an abandoned session): they hold pointers into it
        std::unique_lock<std::mutex> l(s->m);
        s->cv.wait(l, [&] { int n = 0; for (int k = 0; k < kRing; k++) n += s->jobs_open[k]; return n == 0; });
    }
    if (s->st_compute) (void)hipStreamSynchronize(s->st_compute);
    if (s->st_copy) (void)hipStreamSynchronize(s->st_copy);
    if (s->st_pre) (void)hipStreamSynchronize(s->st_pre);
    BufferCache &bc = BufferCache::get();
    SymLayout sl(s->w, s->h);
    auto free3 = [&](void *b[3], int padded) { for (int i = 0; i < 3; i++) bc.release(s->device, s->plane_bytes[padded][i], false, b[i]); };
    for (auto &x : s->pending) if (!x.borrowed) free3(x.base, 0);
    for (auto &x : s->free_src) free3(x.base, 0);
    for (auto &L : s->lane) {
        free3(L.rec_base[0], 1); free3(L.rec_base[1], 1); free3(L.work_base, 2);
        if (L.rec_base[2][0]) free3(L.rec_base[2], 1);
        bc.release(s->device, (size_t)s->n_ctu * 63 * sizeof(int32_t), false, L.me);
        bc.release(s->device, (size_t)s->n_ctu * 63 * sizeof(int32_t), false, L.me1);
        bc.release(s->device, (size_t)s->n_ctu * sizeof(IpInfo), false, L.ip);
        bc.release(s->device, (size_t)s->n_ctu * sizeof(IntraPlan), false, L.plan);
        for (int k = 0; k < s->ring; k++) { bc.release(s->device, sl.dev_total, false, L.sym_dev[k]); bc.release(s->device, sl.total, true, L.sym_host[k]); }
    }
    bc.release(s->device, s->args_cap, false, s->d_args);
    bc.release(s->device, s->scene_cap, false, s->d_scene);
    bc.release(s->device, s->args_cap, true, s->h_args);
    for (int i = 0; i < kRing; i++) { if (s->ev_compute[i]) (void)hipEventDestroy(s->ev_compute[i]); if (s->ev_copy[i]) (void)hipEventDestroy(s->ev_copy[i]); }
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->ev_pre) (void)hipEventDestroy(s->ev_pre);
    if (s->ev_args) (void)hipEventDestroy(s->ev_args);
    if (s->ev_up) (void)hipEventDestroy(s->ev_up);
    bc.release(s->device, s->low_cap, false, s->d_low);
    bc.release(s->device, s->jobs_cap, false, s->d_jobs); bc.release(s->device, s->jobs_cap, true, s->h_jobs);
    bc.release(s->device, s->probe_cap, false, s->d_probe);
    bc.release(s->device, s->flow_bytes, false, s->d_flow);
    for (int k = 0; k < 2; k++) {
        bc.release(s->device, s->x1_bytes, false, s->x1_export[k]);
        if (s->ev_x1[k]) (void)hipEventDestroy(s->ev_x1[k]);
        if (s->ev_x2[k]) (void)hipEventDestroy(s->ev_x2[k]);
    }
    StreamCache::get().release(s->device, s->st_compute);       // both idle: synchronised above
    StreamCache::get().release(s->device, s->st_copy);
    StreamCache::get().release(s->device, s->st_pre);
    delete s;
}

}  // extern "C"
