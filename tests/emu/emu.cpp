// tests/emu/emu.cpp — TEST HARNESS, NOT PRODUCT.  Steps the phase programs of hevc_amd/csrc/kernels/*.h on the
// CPU with the sequential executor (one "thread" at a time, barrier = end of loop), so the kernel SOURCE can be
// checked against the oracle on machines without a GPU.  It proves the kernels' logic, not the GPU execution:
// the -m gpu tests run the real gfx950 binaries through the C ABI.  hevc_amd/ never loads this library.
#include <chrono>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../hevc_amd/csrc/kernels/common.h"
#include "../../hevc_amd/csrc/kernels/inter.h"
#include "../../hevc_amd/csrc/kernels/intra.h"
#include "../../hevc_amd/csrc/kernels/loopfilter.h"

using namespace mihevc;

// LDS is not zeroed on the device: with EMU_SHARED_FILL=<seed> the shared state of every workgroup starts as pseudo-random bytes, so a
// read of a field the program never initialised shows up as a mismatch against the oracle (default: zeros)
static int emu_order() { const char *e = getenv("EMU_ORDER"); return e ? atoi(e) : 0; }
template <class S> static S *fresh_shared()
{
    S *p = (S *)malloc(sizeof(S));
    const char *e = getenv("EMU_SHARED_FILL");
    if (!e) { memset((void *)p, 0, sizeof(S)); return p; }
    static unsigned long long x = 0;
    if (!x) x = 0x9E3779B97F4A7C15ull ^ (unsigned long long)atoll(e);
    unsigned char *b = (unsigned char *)p;
    for (size_t i = 0; i < sizeof(S); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b[i] = (unsigned char)(x >> 32); }
    return p;
}

template <typename T> struct Padded {
    std::vector<T> buf;
    int stride, pad;
    Plane<T> plane;
    Padded(int w, int h, int pad_) : pad(pad_)
    {
        stride = w + 2 * pad;
        buf.assign((size_t)stride * (h + 2 * pad), 0);
        plane.p = buf.data() + (size_t)pad * stride + pad;
        plane.stride = stride;
    }
    void load(const T *src, int w, int h)
    {
        for (int y = 0; y < h; y++) memcpy(plane.p + (ptrdiff_t)y * stride, src + (size_t)y * w, w * sizeof(T));
        for (int i = 0; i < pad_border_count(w, h, pad); i++) pad_border_sample<T>(plane, w, h, pad, i);
    }
    void store(T *dst, int w, int h) const
    {
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, plane.p + (ptrdiff_t)y * stride, w * sizeof(T));
    }
};

static CostParams to_prm(const mihevc_cost_params *p)
{
    return CostParams{p->qp, p->qp_c, p->bit_depth, p->lambda_sad_q4, p->lambda_q4, p->me_range, p->tile_cols, p->tile_rows, p->intra_nxn, p->intra_in_p, p->pre_search, p->rdo_zero, p->chroma_modes, p->mc_top, p->mc_bottom, p->rdo_cg};
}

// ---- EMU_WAVES=<seed>: the four waves of a workgroup as four host threads -------------------------------------------------------------
// The sequential executor runs the code BETWEEN two phases once, so it cannot see a wave that reads shared state for a uniform branch
// after another wave has already entered the next phase and rewritten it (the NxN race of round 1).  Here every wave runs the whole CTU
// program on its own thread, phases end in a real barrier, and a random wave is delayed after each barrier to provoke such skew.  A wave
// that takes a different branch misses a barrier: the barrier times out and the frame call reports -2.
struct WaveBarrier {       // spinning barrier on C++ atomics (sanitizers follow their acquire / release edges)
    std::atomic<int> waiting{0}, generation{0};
    std::atomic<bool> broken{false};
    bool wait()
    {
        if (broken.load()) return false;
        const int gen = generation.load(std::memory_order_acquire);
        if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == NT / 64) {
            waiting.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (generation.load(std::memory_order_acquire) == gen) {
            if (broken.load()) return false;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { broken.store(true); return false; }
            std::this_thread::yield();
        }
        return true;
    }
};
struct WaveAbort {};
struct WaveExec {
    int wave;
    WaveBarrier *bar;
    unsigned long long rnd;
    template <class F> void lanes(F &&f) { for (int l = 0; l < 64; l++) f(wave * 64 + l); }
    template <class F> void phase(F &&f)
    {
        lanes(f);
        if (!bar->wait()) throw WaveAbort{};
        rnd ^= rnd << 13; rnd ^= rnd >> 7; rnd ^= rnd << 17;
        if ((rnd >> 40) % 16 == 0) std::this_thread::sleep_for(std::chrono::microseconds(30));      // this wave falls behind
    }
    template <class F> void wave_step(F &&f) { lanes(f); }
    void atomic_add(int *p, int v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
    void atomic_add(unsigned *p, unsigned v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
    void atomic_add(unsigned long long *p, unsigned long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
    void atomic_or(unsigned *p, unsigned v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
    void atomic_and(unsigned *p, unsigned v) { __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
    void atomic_min(unsigned long long *p, unsigned long long v)
    {
        unsigned long long cur = __atomic_load_n(p, __ATOMIC_RELAXED);
        while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
    unsigned long long peek(const unsigned long long *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
    void atomic_add_global(unsigned long long *p, unsigned long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
};
// run one workgroup program on four wave threads; false when a barrier broke
template <class Program> static bool run_waves(unsigned long long seed, Program &&program)
{
    static WaveBarrier bar;        // one long-lived object (workgroups run one after the other): sanitizers track a mutex by its address
    bar.waiting.store(0); bar.broken.store(false);
    std::vector<std::thread> th;
    for (int wv = 0; wv < NT / 64; wv++)
        th.emplace_back([&, wv] {
            WaveExec ex{wv, &bar, (seed + 1) * 0x9E3779B97F4A7C15ull + (unsigned long long)wv * 0xD1B54A32D192ED03ull};
            try { program(ex); } catch (const WaveAbort &) {}
        });
    for (auto &t : th) t.join();
    return !bar.broken.load();
}

template <typename T>
static int inter_frame(const T *sy, const T *su, const T *sv, const T *ry, const T *ru, const T *rv, int w, int h,
                       const mihevc_cost_params *prm, const int16_t *centers, T *oy, T *ou, T *ov, mihevc_cu_rec *cu,
                       int16_t *cy, int16_t *cu_, int16_t *cv, int32_t *me_dump, unsigned long long *est)
{
    Padded<T> ref0(w, h, PAD_Y), ref1(w / 2, h / 2, PAD_C), ref2(w / 2, h / 2, PAD_C);
    ref0.load(ry, w, h); ref1.load(ru, w / 2, h / 2); ref2.load(rv, w / 2, h / 2);
    InterArgs<T> a;
    a.src[0] = {sy, w}; a.src[1] = {su, w / 2}; a.src[2] = {sv, w / 2};
    a.ref[0] = {ref0.plane.p, ref0.stride}; a.ref[1] = {ref1.plane.p, ref1.stride}; a.ref[2] = {ref2.plane.p, ref2.stride};
    a.rec[0] = {oy, w}; a.rec[1] = {ou, w / 2}; a.rec[2] = {ov, w / 2};
    a.w = w; a.h = h; a.ctus_w = (w + CTU - 1) / CTU;
    a.prm = to_prm(prm); a.centers = centers; a.cu = cu; a.coef[0] = cy; a.coef[1] = cu_; a.coef[2] = cv; a.est = est; a.sparse_coef = 0;
    for (int i = 0; i < 3; i++) a.ref1[i] = {nullptr, 0};
    a.centers1 = nullptr; a.me1 = nullptr;
    if (est) *est = 0;
    std::vector<IpInfo> ipv;
    a.ip = nullptr;
    int n_ctu = a.ctus_w * ((h + CTU - 1) / CTU), R = a.prm.me_range;
    std::vector<int32_t> me((size_t)n_ctu * 63);
    a.me = me.data();
    if (a.prm.intra_in_p) { ipv.assign((size_t)n_ctu, IpInfo{0, 0, 0}); a.ip = ipv.data(); }
    SeqExec ex; ex.order = emu_order();
    std::vector<uint8_t> ls, lr;
    std::vector<int16_t> cen;
    if (a.prm.pre_search && !centers) {
        PreArgs<T> pa;
        ls.assign((size_t)(w / 4) * (h / 4), 0); lr = ls; cen.assign((size_t)n_ctu * 2, 0);
        pa.src = a.src[0]; pa.ref = a.ref[0]; pa.lsrc = ls.data(); pa.lref = lr.data(); pa.w = w; pa.h = h; pa.bit_depth = a.prm.bit_depth; pa.centers = cen.data(); pa.cost = nullptr;
        for (int i = 0; i < 2 * (w / 4) * (h / 4); i++) lowres_sample<T>(pa, i);
        for (int c = 0; c < n_ctu; c++) {
            PreShared ps;
            if (const char *e = getenv("EMU_WAVES")) { if (!run_waves((unsigned long long)atoll(e) + 31u * (unsigned)c, [&](WaveExec &wx) { pre_search_program<T>(wx, ps, pa, c); })) return -2; }
            else pre_search_program<T>(ex, ps, pa, c);
        }
        a.centers = cen.data();
    }
    const char *waves = getenv("EMU_WAVES");
    std::vector<uint8_t> win((size_t)me_win_elems(R) + 8);
    std::vector<T> wy((size_t)mc_win_y(R) * mc_win_y_stride(R) + 16), wu((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16), wv(wu.size());
    for (int c = 0; c < n_ctu; c++) {
        MeShared<T> *ms = fresh_shared<MeShared<T>>();
        bool ok = true;
        if (waves) ok = run_waves((unsigned long long)atoll(waves) + (unsigned)c, [&](WaveExec &wx) { me_search_program<T>(wx, *ms, win.data(), a, c); });
        else me_search_program<T>(ex, *ms, win.data(), a, c);
        free(ms);
        if (!ok) return -2;
    }
    if (me_dump) memcpy(me_dump, me.data(), me.size() * sizeof(int32_t));
    for (int c = 0; c < n_ctu; c++) {
        InterShared<T> *is = fresh_shared<InterShared<T>>();
        bool ok = true;
        if (waves) ok = run_waves((unsigned long long)atoll(waves) + 7919u * (unsigned)c, [&](WaveExec &wx) { inter_ctu_program<T>(wx, *is, wy.data(), wu.data(), wv.data(), a, c); });
        else inter_ctu_program<T>(ex, *is, wy.data(), wu.data(), wv.data(), a, c);
        free(is);
        if (!ok) return -2;
    }
    if (a.ip) {       // intra second pass, two rounds like the device's two launches
        IntraArgs<T> ia;
        for (int i = 0; i < 3; i++) { ia.src[i] = a.src[i]; ia.rec[i] = a.rec[i]; ia.coef[i] = a.coef[i]; }
        ia.w = w; ia.h = h; ia.ctus_w = a.ctus_w; ia.ctus_h = (h + CTU - 1) / CTU;
        ia.prm = a.prm;        // tile_cols / tile_rows: the P pictures' own grid
        ia.cu = cu; ia.diagonal = 0; ia.est = est; ia.sparse_coef = 0; ia.ip = a.ip; ia.plan = nullptr;
        for (int round = 0; round < 2; round++)
            for (int c = 0; c < n_ctu; c++) {
                if (!ip_eligible(ia.ip, ia.ctus_w, ia.ctus_h, c % ia.ctus_w, c / ia.ctus_w, round)) continue;
                IntraShared<T> *is = fresh_shared<IntraShared<T>>();
                bool ok = true;
                if (waves) ok = run_waves((unsigned long long)atoll(waves) + 104729u * (unsigned)c, [&](WaveExec &wx) { intra_ctu_program<T>(wx, *is, ia, c % ia.ctus_w, c / ia.ctus_w); });
                else intra_ctu_program<T>(ex, *is, ia, c % ia.ctus_w, c / ia.ctus_w);
                free(is);
                if (!ok) return -2;
            }
    }
    return 0;
}

// B picture between two anchors: integer search against both (list 0: r0*, list 1: r1*), then the B form of the CTU program
template <typename T>
static int b_frame(const T *sy, const T *su, const T *sv, const T *r0y, const T *r0u, const T *r0v, const T *r1y, const T *r1u, const T *r1v, int w, int h,
                   const mihevc_cost_params *prm, const int16_t *centers0, const int16_t *centers1, T *oy, T *ou, T *ov, mihevc_cu_rec *cu,
                   int16_t *cy, int16_t *cu_, int16_t *cv, int32_t *me_dump0, int32_t *me_dump1, unsigned long long *est)
{
    Padded<T> p00(w, h, PAD_Y), p01(w / 2, h / 2, PAD_C), p02(w / 2, h / 2, PAD_C), p10(w, h, PAD_Y), p11(w / 2, h / 2, PAD_C), p12(w / 2, h / 2, PAD_C);
    p00.load(r0y, w, h); p01.load(r0u, w / 2, h / 2); p02.load(r0v, w / 2, h / 2);
    p10.load(r1y, w, h); p11.load(r1u, w / 2, h / 2); p12.load(r1v, w / 2, h / 2);
    InterArgs<T> a;
    a.src[0] = {sy, w}; a.src[1] = {su, w / 2}; a.src[2] = {sv, w / 2};
    a.ref[0] = {p00.plane.p, p00.stride}; a.ref[1] = {p01.plane.p, p01.stride}; a.ref[2] = {p02.plane.p, p02.stride};
    a.ref1[0] = {p10.plane.p, p10.stride}; a.ref1[1] = {p11.plane.p, p11.stride}; a.ref1[2] = {p12.plane.p, p12.stride};
    a.rec[0] = {oy, w}; a.rec[1] = {ou, w / 2}; a.rec[2] = {ov, w / 2};
    a.w = w; a.h = h; a.ctus_w = (w + CTU - 1) / CTU;
    a.prm = to_prm(prm); a.centers = centers0; a.centers1 = centers1; a.cu = cu; a.coef[0] = cy; a.coef[1] = cu_; a.coef[2] = cv; a.est = est; a.sparse_coef = 0; a.ip = nullptr;
    if (est) *est = 0;
    const int n_ctu = a.ctus_w * ((h + CTU - 1) / CTU), R = a.prm.me_range;
    std::vector<int32_t> me0((size_t)n_ctu * 63), me1((size_t)n_ctu * 63);
    a.me = me0.data(); a.me1 = me1.data();
    SeqExec ex; ex.order = emu_order();
    const char *waves = getenv("EMU_WAVES");
    std::vector<uint8_t> win((size_t)me_win_elems(R) + 8);
    std::vector<T> wy((size_t)mc_win_y(R) * mc_win_y_stride(R) + 16), wu((size_t)mc_win_c(R) * mc_win_c_stride(R) + 16), wv(wu.size());
    for (int list = 0; list < 2; list++) {
        const InterArgs<T> al = list ? list1_view(a) : a;
        for (int c = 0; c < n_ctu; c++) {
            MeShared<T> *ms = fresh_shared<MeShared<T>>();
            bool ok = true;
            if (waves) ok = run_waves((unsigned long long)atoll(waves) + (unsigned)c + 1000u * list, [&](WaveExec &wx) { me_search_program<T>(wx, *ms, win.data(), al, c); });
            else me_search_program<T>(ex, *ms, win.data(), al, c);
            free(ms);
            if (!ok) return -2;
        }
    }
    if (me_dump0) memcpy(me_dump0, me0.data(), me0.size() * sizeof(int32_t));
    if (me_dump1) memcpy(me_dump1, me1.data(), me1.size() * sizeof(int32_t));
    for (int c = 0; c < n_ctu; c++) {
        InterShared<T> *is = fresh_shared<InterShared<T>>();
        BiShared *bs = fresh_shared<BiShared>();
        bool ok = true;
        if (waves) ok = run_waves((unsigned long long)atoll(waves) + 7919u * (unsigned)c, [&](WaveExec &wx) { inter_ctu_program<T, WaveExec, true>(wx, *is, wy.data(), wu.data(), wv.data(), a, c, bs); });
        else inter_ctu_program<T, SeqExec, true>(ex, *is, wy.data(), wu.data(), wv.data(), a, c, bs);
        free(is); free(bs);
        if (!ok) return -2;
    }
    return 0;
}

template <typename T>
static int intra_frame(const T *sy, const T *su, const T *sv, int w, int h, const mihevc_cost_params *prm, T *oy, T *ou, T *ov,
                       mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_, int16_t *cv, unsigned long long *est)
{
    IntraArgs<T> a;
    a.src[0] = {sy, w}; a.src[1] = {su, w / 2}; a.src[2] = {sv, w / 2};
    a.rec[0] = {oy, w}; a.rec[1] = {ou, w / 2}; a.rec[2] = {ov, w / 2};
    a.w = w; a.h = h; a.ctus_w = (w + CTU - 1) / CTU; a.ctus_h = (h + CTU - 1) / CTU;
    a.prm = to_prm(prm); a.cu = cu; a.coef[0] = cy; a.coef[1] = cu_; a.coef[2] = cv; a.est = est; a.sparse_coef = 0; a.ip = nullptr; a.plan = nullptr;
    if (est) *est = 0;
    SeqExec ex; ex.order = emu_order();
    // same launch order as the device: per tile, one anti-diagonal (cx + 2 cy inside the tile) at a time
    const int tcn = a.prm.tile_cols > 1 ? a.prm.tile_cols : 1, trn = a.prm.tile_rows > 1 ? a.prm.tile_rows : 1;
    const int colw = (a.ctus_w + tcn - 1) / tcn, rowh = (a.ctus_h + trn - 1) / trn;
    for (int d = 0; d <= (colw - 1) + 2 * (rowh - 1); d++) {
        a.diagonal = d;
        for (int b = 0; b < tcn * trn * rowh; b++) {
            const int tile = b / rowh, r = b % rowh, tx = tile % tcn, ty = tile / tcn;
            const int cx0 = tile_bd(tx, tcn, a.ctus_w), cx1 = tile_bd(tx + 1, tcn, a.ctus_w), cy0 = tile_bd(ty, trn, a.ctus_h), cy1 = tile_bd(ty + 1, trn, a.ctus_h);
            const int cyi = cy0 + r, cxi = cx0 + d - 2 * r;
            if (cyi >= cy1 || cxi < cx0 || cxi >= cx1) continue;
            IntraShared<T> *is = fresh_shared<IntraShared<T>>();
            bool ok = true;
            if (const char *e = getenv("EMU_WAVES")) ok = run_waves((unsigned long long)atoll(e) + (unsigned)(cyi * 4096 + cxi), [&](WaveExec &wx) { intra_ctu_program<T>(wx, *is, a, cxi, cyi); });
            else intra_ctu_program<T>(ex, *is, a, cxi, cyi);
            free(is);
            if (!ok) return -2;
        }
    }
    return 0;
}

// row0 / y_org: deblock rows [row0, row0 + h) of a whole picture's planes as a picture of its own whose first y_org rows belong to the slice above
// (DeblockArgs::y_org; the band extended by the rows its neighbours hand over, csrc/slice_group.h)
template <typename T> static int deblock(T *y, T *u, T *v, int w, int h, const mihevc_cu_rec *cu, int bit_depth, int row0 = 0, int y_org = 0)
{
    DeblockArgs<T> a;
    a.rec[0] = {y + (ptrdiff_t)row0 * w, w}; a.rec[1] = {u + (ptrdiff_t)(row0 / 2) * (w / 2), w / 2}; a.rec[2] = {v + (ptrdiff_t)(row0 / 2) * (w / 2), w / 2};
    a.w = w; a.h = h; a.cu = cu + (ptrdiff_t)(row0 / 8) * (w / 8); a.bit_depth = bit_depth; a.y_org = y_org;
    for (int dir = 0; dir < 2; dir++) {
        a.dir = dir;
        for (int i = 0; i < (w / 8) * (h / 8) * 2; i++) deblock_segment<T>(a, i);
    }
    return 0;
}

// y0 / halo: the planes are those of a whole picture of which rows [y0, y0 + h) are coded here as one slice whose filters run across the seams
// (SaoArgs::halo; csrc/slice_group.h): the kernel then sees its band as the picture and finds the neighbour rows where the exchange puts them
template <typename T>
static int sao(const T *sy, const T *su, const T *sv, const T *dy, const T *du, const T *dv, int w, int h, const mihevc_cost_params *prm,
               T *oy, T *ou, T *ov, mihevc_sao_ctu *out, int y0 = 0, int halo = 0, uint32_t *sse_ctu = nullptr, const mihevc_cu_rec *cu = nullptr)
{
    SaoArgs<T> a;
    const ptrdiff_t oy_ = (ptrdiff_t)y0 * w, oc_ = (ptrdiff_t)(y0 / 2) * (w / 2);
    a.src[0] = {sy + oy_, w}; a.src[1] = {su + oc_, w / 2}; a.src[2] = {sv + oc_, w / 2};
    a.dbk[0] = {dy + oy_, w}; a.dbk[1] = {du + oc_, w / 2}; a.dbk[2] = {dv + oc_, w / 2};
    a.out[0] = {oy + oy_, w}; a.out[1] = {ou + oc_, w / 2}; a.out[2] = {ov + oc_, w / 2};
    a.w = w; a.h = h; a.ctus_w = (w + CTU - 1) / CTU; a.prm = to_prm(prm); a.sao = out; a.sse = nullptr; a.sse_ctu = sse_ctu; a.cu = cu ? cu + (size_t)(y0 / 8) * (w / 8) : nullptr; a.halo_top = (halo & 1) ? 1 : 0; a.halo_bottom = (halo & 2) ? 1 : 0;
    SeqExec ex; ex.order = emu_order();
    int n_ctu = a.ctus_w * ((h + CTU - 1) / CTU);
    for (int c = 0; c < n_ctu; c++) {
        SaoShared<T> s;
        if (const char *e = getenv("EMU_WAVES")) { if (!run_waves((unsigned long long)atoll(e) + 131u * (unsigned)c, [&](WaveExec &wx) { sao_ctu_program<T>(wx, s, a, c); })) return -2; }
        else sao_ctu_program<T>(ex, s, a, c);
    }
    return 0;          // the CTU programs applied the offsets themselves (as k_sao_decide does)
}

extern "C" {
int emu_inter_frame(const void *sy, const void *su, const void *sv, const void *ry, const void *ru, const void *rv, int w, int h,
                    const mihevc_cost_params *prm, const int16_t *centers, void *oy, void *ou, void *ov, mihevc_cu_rec *cu,
                    int16_t *cy, int16_t *cu_, int16_t *cv, int32_t *me_dump, unsigned long long *est)
{
    if (prm->bit_depth == 8)
        return inter_frame<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (const uint8_t *)ry, (const uint8_t *)ru,
                                    (const uint8_t *)rv, w, h, prm, centers, (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov, cu, cy, cu_, cv, me_dump, est);
    return inter_frame<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (const uint16_t *)ry, (const uint16_t *)ru,
                                 (const uint16_t *)rv, w, h, prm, centers, (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov, cu, cy, cu_, cv, me_dump, est);
}
int emu_b_frame(const void *sy, const void *su, const void *sv, const void *r0y, const void *r0u, const void *r0v, const void *r1y, const void *r1u, const void *r1v,
                int w, int h, const mihevc_cost_params *prm, const int16_t *centers0, const int16_t *centers1, void *oy, void *ou, void *ov, mihevc_cu_rec *cu,
                int16_t *cy, int16_t *cu_, int16_t *cv, int32_t *me_dump0, int32_t *me_dump1, unsigned long long *est)
{
    if (prm->bit_depth == 8)
        return b_frame<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (const uint8_t *)r0y, (const uint8_t *)r0u, (const uint8_t *)r0v,
                                (const uint8_t *)r1y, (const uint8_t *)r1u, (const uint8_t *)r1v, w, h, prm, centers0, centers1, (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov,
                                cu, cy, cu_, cv, me_dump0, me_dump1, est);
    return b_frame<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (const uint16_t *)r0y, (const uint16_t *)r0u, (const uint16_t *)r0v,
                             (const uint16_t *)r1y, (const uint16_t *)r1u, (const uint16_t *)r1v, w, h, prm, centers0, centers1, (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov,
                             cu, cy, cu_, cv, me_dump0, me_dump1, est);
}
int emu_intra_frame(const void *sy, const void *su, const void *sv, int w, int h, const mihevc_cost_params *prm, void *oy, void *ou, void *ov,
                    mihevc_cu_rec *cu, int16_t *cy, int16_t *cu_, int16_t *cv, unsigned long long *est)
{
    if (prm->bit_depth == 8)
        return intra_frame<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, w, h, prm, (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov, cu, cy, cu_, cv, est);
    return intra_frame<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, w, h, prm, (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov, cu, cy, cu_, cv, est);
}
int emu_deblock(void *y, void *u, void *v, int w, int h, const mihevc_cu_rec *cu, int bit_depth)
{
    if (bit_depth == 8) return deblock<uint8_t>((uint8_t *)y, (uint8_t *)u, (uint8_t *)v, w, h, cu, bit_depth);
    return deblock<uint16_t>((uint16_t *)y, (uint16_t *)u, (uint16_t *)v, w, h, cu, bit_depth);
}
int emu_deblock_band(void *y, void *u, void *v, int w, int row0, int h, int y_org, const mihevc_cu_rec *cu, int bit_depth)
{
    if (bit_depth == 8) return deblock<uint8_t>((uint8_t *)y, (uint8_t *)u, (uint8_t *)v, w, h, cu, bit_depth, row0, y_org);
    return deblock<uint16_t>((uint16_t *)y, (uint16_t *)u, (uint16_t *)v, w, h, cu, bit_depth, row0, y_org);
}
int emu_sao_band(const void *sy, const void *su, const void *sv, const void *dy, const void *du, const void *dv, int w, int y0, int h, int halo,
                 const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *out)
{
    if (prm->bit_depth == 8)
        return sao<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (const uint8_t *)dy, (const uint8_t *)du, (const uint8_t *)dv, w, h, prm,
                            (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov, out, y0, halo);
    return sao<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (const uint16_t *)dy, (const uint16_t *)du, (const uint16_t *)dv, w, h, prm,
                         (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov, out, y0, halo);
}
// the fused loop filter (SaoArgs::cu): r* = the PRE-deblock reconstruction; y0 / h / halo as emu_sao_band (cu: the whole picture's records)
int emu_loop_filter(const void *sy, const void *su, const void *sv, const void *ry, const void *ru, const void *rv, int w, int y0, int h, int halo, const mihevc_cu_rec *cu,
                    const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *out)
{
    if (prm->bit_depth == 8)
        return sao<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (const uint8_t *)ry, (const uint8_t *)ru, (const uint8_t *)rv, w, h, prm,
                            (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov, out, y0, halo, nullptr, cu);
    return sao<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (const uint16_t *)ry, (const uint16_t *)ru, (const uint16_t *)rv, w, h, prm,
                         (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov, out, y0, halo, nullptr, cu);
}
int emu_sao_sse(const void *sy, const void *su, const void *sv, const void *dy, const void *du, const void *dv, int w, int h,
                const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *out, uint32_t *sse_ctu)
{
    if (prm->bit_depth == 8)
        return sao<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (const uint8_t *)dy, (const uint8_t *)du, (const uint8_t *)dv, w, h, prm,
                            (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov, out, 0, 0, sse_ctu);
    return sao<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (const uint16_t *)dy, (const uint16_t *)du, (const uint16_t *)dv, w, h, prm,
                         (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov, out, 0, 0, sse_ctu);
}
int emu_sao(const void *sy, const void *su, const void *sv, const void *dy, const void *du, const void *dv, int w, int h,
            const mihevc_cost_params *prm, void *oy, void *ou, void *ov, mihevc_sao_ctu *out)
{
    if (prm->bit_depth == 8)
        return sao<uint8_t>((const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (const uint8_t *)dy, (const uint8_t *)du, (const uint8_t *)dv, w, h, prm,
                            (uint8_t *)oy, (uint8_t *)ou, (uint8_t *)ov, out);
    return sao<uint16_t>((const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (const uint16_t *)dy, (const uint16_t *)du, (const uint16_t *)dv, w, h, prm,
                         (uint16_t *)oy, (uint16_t *)ou, (uint16_t *)ov, out);
}
}
