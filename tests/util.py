"""Shared helpers of the test-suite: deterministic synthetic pictures and array plumbing between the oracle
(uint16 containers) and the product C ABI (uint8 planes at 8 bit, uint16 at 10 bit)."""
import ctypes as C

import numpy as np

from oracle import oracle as O


def synth_frame(h, w, seed=0, shift=(0, 0), bit_depth=8, detail=True) -> O.Frame:
    """Band-limited texture + sine + moving rectangles + per-frame grain; `shift` translates the texture."""
    rng = np.random.default_rng(seed)
    big = rng.normal(0, 1, (h + 384, w + 384))          # room for shifts up to +-190 samples
    for _ in range(3):
        big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(big, -1, 0) + np.roll(big, -1, 1)) / 5
    big = (big - big.min()) / (big.max() - big.min())
    oy, ox = 192 + shift[1], 192 + shift[0]
    y = big[oy:oy + h, ox:ox + w] * 180 + 30
    yy, xx = np.mgrid[0:h, 0:w]
    y = y + 15 * np.sin((xx + shift[0]) / 7.0)
    if detail:
        r2 = np.random.default_rng(seed + 7)
        for _ in range(max(2, (h * w) // 4096)):          # sharp-edged rectangles: force small CUs / angular modes
            rw, rh = int(r2.integers(4, 40)), int(r2.integers(4, 40))
            rx, ry = int(r2.integers(0, max(1, w - rw))), int(r2.integers(0, max(1, h - rh)))
            rx = (rx + shift[0] * 2) % max(1, w - rw)
            y[ry:ry + rh, rx:rx + rw] = r2.integers(16, 235)
        # a high-frequency checker patch
        y[h // 2:h // 2 + 16, 8:40] = np.where(((xx[:16, :32] // 2) + (yy[:16, :32] // 2)) % 2 == 0, 40, 210)
    g = np.random.default_rng((seed * 1000 + shift[0] * 31 + shift[1] + 1) % (1 << 32))      # negative shifts with seed 0: numpy takes non-negative seeds only
    y = np.clip(y + g.normal(0, 1.5, (h, w)), 0, 255)
    u = np.clip(128 + 30 * np.sin(yy[::2, ::2] / 9.0) + 20 * big[oy:oy + h:2, ox:ox + w:2] + g.normal(0, 1, (h // 2, w // 2)), 0, 255)
    v = np.clip(128 + 30 * np.cos(xx[::2, ::2] / 11.0) - 20 * big[oy:oy + h:2, ox:ox + w:2] + g.normal(0, 1, (h // 2, w // 2)), 0, 255)
    sc = 1 << (bit_depth - 8)
    return O.Frame((y * sc).astype(np.uint16), (u * sc).astype(np.uint16), (v * sc).astype(np.uint16))


def dtype_for(bit_depth):
    return np.uint8 if bit_depth == 8 else np.uint16


def planes(frame: O.Frame, bit_depth):
    dt = dtype_for(bit_depth)
    return [np.ascontiguousarray(p.astype(dt)) for p in (frame.y, frame.u, frame.v)]


def to_frame(ps) -> O.Frame:
    return O.Frame(*[p.astype(np.uint16) for p in ps])


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def n_ctus(w, h):
    return ((w + 31) // 32) * ((h + 31) // 32)


def psnr(a, b, peak=255.0):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(peak * peak / mse)


class StageApi:
    """Thin wrapper over the per-stage C-ABI entry points (mihevc_k_* or the emu_* twins with the same shape)."""

    def __init__(self, lib, prefix, device=None):
        self.lib, self.prefix, self.device = lib, prefix, device

    def _call(self, name, *args):
        f = getattr(self.lib, self.prefix + name)
        rc = f(*(([self.device] if self.device is not None else []) + list(args)))
        assert rc == 0, f"{self.prefix}{name} -> {rc}"

    def intra(self, src: O.Frame, prm):
        bd = prm.bit_depth
        h, w = src.shape
        s = planes(src, bd)
        o = [np.zeros_like(p) for p in s]
        a = O.Analysis(h, w)
        est = C.c_uint64(0)
        self._call("intra_frame", ptr(s[0]), ptr(s[1]), ptr(s[2]), w, h, C.byref(prm), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(a.cu), ptr(a.coef_y),
                   ptr(a.coef_u), ptr(a.coef_v), C.byref(est))
        a.rec, a.est = to_frame(o), est.value
        return a

    def inter(self, src: O.Frame, ref: O.Frame, prm, centers=None):
        bd = prm.bit_depth
        h, w = src.shape
        s, r = planes(src, bd), planes(ref, bd)
        o = [np.zeros_like(p) for p in s]
        a = O.Analysis(h, w)
        me = np.zeros((n_ctus(w, h), 21, 3), np.int32)
        est = C.c_uint64(0)
        cen = np.ascontiguousarray(centers, dtype=np.int16) if centers is not None else None
        self._call("inter_frame", ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(r[0]), ptr(r[1]), ptr(r[2]), w, h, C.byref(prm),
                   ptr(cen) if cen is not None else None, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(a.cu), ptr(a.coef_y), ptr(a.coef_u), ptr(a.coef_v), ptr(me), C.byref(est))
        a.rec, a.me, a.est = to_frame(o), me, est.value
        return a

    def b(self, src: O.Frame, ref0: O.Frame, ref1: O.Frame, prm, centers0=None, centers1=None):
        """B picture between two anchors (mihevc_k_b_frame / emu_b_frame): a.me = (list-0 dump, list-1 dump)"""
        bd = prm.bit_depth
        h, w = src.shape
        s, r0, r1 = planes(src, bd), planes(ref0, bd), planes(ref1, bd)
        o = [np.zeros_like(p) for p in s]
        a = O.Analysis(h, w)
        me = [np.zeros((n_ctus(w, h), 21, 3), np.int32) for _ in range(2)]
        est = C.c_uint64(0)
        cen = [np.ascontiguousarray(c, dtype=np.int16) if c is not None else None for c in (centers0, centers1)]
        self._call("b_frame", ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(r0[0]), ptr(r0[1]), ptr(r0[2]), ptr(r1[0]), ptr(r1[1]), ptr(r1[2]), w, h, C.byref(prm),
                   ptr(cen[0]) if cen[0] is not None else None, ptr(cen[1]) if cen[1] is not None else None, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(a.cu), ptr(a.coef_y),
                   ptr(a.coef_u), ptr(a.coef_v), ptr(me[0]), ptr(me[1]), C.byref(est))
        a.rec, a.me, a.est = to_frame(o), tuple(me), est.value
        return a

    def deblock(self, rec: O.Frame, cu, bd):
        r = planes(rec, bd)
        h, w = rec.shape
        self._call("deblock", ptr(r[0]), ptr(r[1]), ptr(r[2]), w, h, ptr(np.ascontiguousarray(cu)), bd)
        return to_frame(r)

    def sao(self, src: O.Frame, dbk: O.Frame, prm):
        bd = prm.bit_depth
        h, w = src.shape
        s, d = planes(src, bd), planes(dbk, bd)
        o = [np.zeros_like(p) for p in s]
        sp = np.zeros(n_ctus(w, h), O.SAO_DTYPE)
        self._call("sao", ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(d[0]), ptr(d[1]), ptr(d[2]), w, h, C.byref(prm), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(sp))
        return to_frame(o), sp

    def loop_filter(self, src: O.Frame, rec: O.Frame, cu, prm, band=None):
        """deblocking + SAO in one pass over the PRE-deblock reconstruction (the fused CTU program a session runs).  band = (y0, h, halo): emulator only, rows
        [y0, y0 + h) of the picture as one slice whose filters run across the seams (halo bit 0: a slice above, bit 1: below)"""
        bd = prm.bit_depth
        h, w = src.shape
        s, d = planes(src, bd), planes(rec, bd)
        o = [np.zeros_like(p) for p in s]
        cu = np.ascontiguousarray(cu)
        if band is None and self.prefix != "emu_":
            sp = np.zeros(n_ctus(w, h), O.SAO_DTYPE)
            self._call("loop_filter", ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(d[0]), ptr(d[1]), ptr(d[2]), w, h, ptr(cu), C.byref(prm), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(sp))
            return to_frame(o), sp
        y0, bh, halo = band if band is not None else (0, h, 0)
        sp = np.zeros(n_ctus(w, bh), O.SAO_DTYPE)
        self._call("loop_filter", ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(d[0]), ptr(d[1]), ptr(d[2]), w, y0, bh, halo, ptr(cu), C.byref(prm), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(sp))
        return to_frame(o), sp

    def sao_sse(self, src: O.Frame, dbk: O.Frame, prm):
        """emulator only: SAO with the per-CTU squared-error table the CTU programs leave (SaoArgs::sse_ctu)"""
        bd = prm.bit_depth
        h, w = src.shape
        s, d = planes(src, bd), planes(dbk, bd)
        o = [np.zeros_like(p) for p in s]
        sp = np.zeros(n_ctus(w, h), O.SAO_DTYPE)
        sse = np.full((n_ctus(w, h), 3), 0xffffffff, np.uint32)
        self._call("sao_sse", ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(d[0]), ptr(d[1]), ptr(d[2]), w, h, C.byref(prm), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(sp), ptr(sse))
        return to_frame(o), sp, sse


def same_analysis(a, b):
    return (a.rec.same(b.rec) and np.array_equal(a.cu, b.cu) and np.array_equal(a.coef_y, b.coef_y) and
            np.array_equal(a.coef_u, b.coef_u) and np.array_equal(a.coef_v, b.coef_v) and a.est == b.est)


def describe_diff(a, b):
    out = []
    if not np.array_equal(a.cu, b.cu):
        i = np.argwhere(a.cu != b.cu)[0]
        out.append(f"cu[{tuple(i)}]: {a.cu[tuple(i)]} vs {b.cu[tuple(i)]}")
    for n in ("y", "u", "v"):
        pa, pb = getattr(a.rec, n), getattr(b.rec, n)
        if not np.array_equal(pa, pb):
            ys, xs = np.nonzero(pa != pb)
            out.append(f"rec.{n} {len(ys)} diffs, first at x={xs[0]} y={ys[0]}: {pa[ys[0], xs[0]]} vs {pb[ys[0], xs[0]]}")
    for n in ("coef_y", "coef_u", "coef_v"):
        if not np.array_equal(getattr(a, n), getattr(b, n)):
            out.append(f"{n} differs")
    if a.est != b.est:
        out.append(f"rate estimate {a.est} vs {b.est}")
    return "; ".join(out) or "identical"


def run_pipeline(api_or_oracle, srcs, prm_i, prm_p, bd=8):
    """I then P pictures through analysis -> deblock -> SAO with either the oracle module or a StageApi.
    Returns list of (analysis, deblocked, final, sao_params)."""
    out, ref = [], None
    for i, src in enumerate(srcs):
        prm = prm_i if i == 0 else prm_p
        if api_or_oracle is O:
            a = O.analyze_intra(src, prm) if i == 0 else O.analyze_inter(src, ref, prm, dump_me=True)
            d = O.deblock(a.rec, a.cu, bd)
            f, sp = O.sao(src, d, prm)
        else:
            a = api_or_oracle.intra(src, prm) if i == 0 else api_or_oracle.inter(src, ref, prm)
            d = api_or_oracle.deblock(a.rec, a.cu, bd)
            f, sp = api_or_oracle.sao(src, d, prm)
        out.append((a, d, f, sp))
        ref = f
    return out


def idr_positions(n, keyint, lanes=4, balance=True):
    """IDR pictures of a session without scene cuts (hevc_amd/csrc/session.cpp encode_chunk): chunks of lanes x keyint pictures, every chunk coded as the
    fewest GOPs keyint allows, of near-equal length (cfg.gop_balance, the default) or with an IDR every keyint pictures."""
    out, pos = [], 0
    while pos < n:
        m = min(lanes * keyint, n - pos)
        g = (m + keyint - 1) // keyint
        at = pos
        for j in range(g):
            out.append(at)
            at += (m // g + (j < m % g)) if balance else keyint
        pos += m
    return out
