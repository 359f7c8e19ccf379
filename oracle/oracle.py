"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
hevc_amd/ never does (tests/test_layout.py greps for that).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
CTU = 32
PAD = 80

_lib = None


def build(force: bool = False) -> Path:
    so = HERE / "liboracle.so"
    srcs = [HERE / "hevc_oracle.c", HERE / "hevc_dec.c", HERE / "hevc_dec_recon.c", HERE / "hevc_oracle.h", HERE / "hevc_dec_recon.h"]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs if s.exists()):
        subprocess.run(["make", "-C", str(HERE), "-s"], check=True)
    return so


class Params(C.Structure):
    _fields_ = [("qp", C.c_int), ("qp_c", C.c_int), ("bit_depth", C.c_int), ("lambda_sad_q4", C.c_int),
                ("lambda_q4", C.c_int), ("me_range", C.c_int), ("tile_cols", C.c_int), ("tile_rows", C.c_int), ("intra_nxn", C.c_int), ("intra_in_p", C.c_int), ("pre_search", C.c_int), ("rdo_zero", C.c_int), ("chroma_modes", C.c_int), ("mc_top", C.c_int), ("mc_bottom", C.c_int), ("rdo_cg", C.c_int)]


CU_DTYPE = np.dtype([("log2_size", "u1"), ("flags", "u1"), ("chroma_mode", "u1"), ("qp", "u1"), ("intra_mode", "u1", (4,)),
                     ("mvx", "<i2"), ("mvy", "<i2"), ("cbf_y4", "u1"), ("pad", "u1", (3,))])
SAO_DTYPE = np.dtype([("type", "u1", (2,)), ("eo_class", "u1", (2,)), ("band_pos", "u1", (3,)), ("offset", "i1", (3, 4)), ("pad", "u1")])
assert CU_DTYPE.itemsize == 16 and SAO_DTYPE.itemsize == 20


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.orc_dec_open.restype = C.c_void_p
        _lib.orc_dec_error.restype = C.c_char_p
        _lib.orc_dec_error.argtypes = [C.c_void_p]
        _lib.orc_dec_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        _lib.orc_dec_close.argtypes = [C.c_void_p]
        _lib.orc_dec_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
        _lib.orc_dec_get_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_dec_query.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(qp: int, bit_depth: int = 8, me_range: int = 16) -> Params:
    """Same integer cost parameters the product derives (mihevc_cost_params): lambda = 0.57 * 2^((qp-12)/3)."""
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    sh = bit_depth - 8          # distortions grow 4x (SAD) / 16x (SSE) from 8 to 10 bit: the multipliers follow
    return Params(qp, int(lib().orc_chroma_qp(qp)), bit_depth, int(round(16 * lam ** 0.5)) << sh, int(round(16 * lam)) << (2 * sh), me_range, 1, 1, 0, 0, 0, 0, 0)


# ---------------------------------------------------------------- primitives
def fwd_transform(res: np.ndarray, dst=False, bit_depth=8) -> np.ndarray:
    n = res.shape[0]
    res = np.ascontiguousarray(res, dtype=np.int16)
    out = np.empty((n, n), np.int16)
    lib().orc_fwd_transform(_p(res), n, _p(out), int(np.log2(n)), int(dst), bit_depth)
    return out


def inv_transform(coef: np.ndarray, dst=False, bit_depth=8) -> np.ndarray:
    n = coef.shape[0]
    coef = np.ascontiguousarray(coef, dtype=np.int16)
    out = np.empty((n, n), np.int16)
    lib().orc_inv_transform(_p(coef), _p(out), n, int(np.log2(n)), int(dst), bit_depth)
    return out


def quant(coef: np.ndarray, qp: int, bit_depth=8, intra=True) -> np.ndarray:
    n = coef.shape[0]
    coef = np.ascontiguousarray(coef, dtype=np.int16)
    out = np.empty((n, n), np.int16)
    lib().orc_quant(_p(coef), _p(out), int(np.log2(n)), qp, bit_depth, int(intra))
    return out


def dequant(lvl: np.ndarray, qp: int, bit_depth=8) -> np.ndarray:
    n = lvl.shape[0]
    lvl = np.ascontiguousarray(lvl, dtype=np.int16)
    out = np.empty((n, n), np.int16)
    lib().orc_dequant(_p(lvl), _p(out), int(np.log2(n)), qp, bit_depth)
    return out


def transform_matrix() -> np.ndarray:
    m = np.empty((32, 32), np.int16)
    lib().orc_transform_matrix(_p(m))
    return m


def intra_pred(ref: np.ndarray, log2n: int, mode: int, c_idx=0, bit_depth=8, filtered=True, strong=True) -> np.ndarray:
    """ref: 4N+1 samples (bottom-left ... corner ... top-right). Applies 8.4.4.2.3 filtering when `filtered`."""
    n = 1 << log2n
    ref = np.ascontiguousarray(ref, dtype=np.uint16)
    assert ref.size == 4 * n + 1
    use = ref
    if filtered:
        use = np.empty_like(ref)
        lib().orc_intra_filter_ref(_p(ref), _p(use), log2n, mode, c_idx, bit_depth, int(strong))
    out = np.empty((n, n), np.uint16)
    lib().orc_intra_pred(_p(use), _p(out), n, log2n, mode, c_idx, bit_depth)
    return out


def intra_build_ref(rec: np.ndarray, x0: int, y0: int, log2n: int, c_idx=0, bit_depth=8) -> np.ndarray:
    rec = np.ascontiguousarray(rec, dtype=np.uint16)
    h, w = rec.shape
    out = np.empty(4 * (1 << log2n) + 1, np.uint16)
    lib().orc_intra_build_ref(_p(rec), w, x0, y0, log2n, w, h, None, 0, c_idx, bit_depth, _p(out))
    return out


def pad_plane(plane: np.ndarray, pad: int) -> np.ndarray:
    """returns a padded copy (edge replication); data origin at [pad, pad]"""
    return np.pad(np.asarray(plane, dtype=np.uint16), pad, mode="edge")


def interp_luma(ref_padded: np.ndarray, pad: int, x, y, mvx, mvy, w, h, bit_depth=8) -> np.ndarray:
    ref_padded = np.ascontiguousarray(ref_padded, dtype=np.uint16)
    stride = ref_padded.shape[1]
    out = np.empty((h, w), np.uint16)
    base = ref_padded.ctypes.data + 2 * (pad * stride + pad)
    lib().orc_interp_luma(C.c_void_p(base), stride, x, y, mvx, mvy, w, h, bit_depth, _p(out), w)
    return out


def interp_chroma(ref_padded: np.ndarray, pad: int, xc, yc, mvx, mvy, w, h, bit_depth=8) -> np.ndarray:
    ref_padded = np.ascontiguousarray(ref_padded, dtype=np.uint16)
    stride = ref_padded.shape[1]
    out = np.empty((h, w), np.uint16)
    base = ref_padded.ctypes.data + 2 * (pad * stride + pad)
    lib().orc_interp_chroma(C.c_void_p(base), stride, xc, yc, mvx, mvy, w, h, bit_depth, _p(out), w)
    return out


def satd(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint16)
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return int(lib().orc_satd(_p(a), a.shape[1], _p(b), b.shape[1], a.shape[1], a.shape[0]))


def sad(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint16)
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return int(lib().orc_sad(_p(a), a.shape[1], _p(b), b.shape[1], a.shape[1], a.shape[0]))


# ---------------------------------------------------------------- frame stages
class Frame:
    """planar 4:2:0 picture in uint16 containers"""

    def __init__(self, y, u, v):
        self.y = np.ascontiguousarray(y, dtype=np.uint16)
        self.u = np.ascontiguousarray(u, dtype=np.uint16)
        self.v = np.ascontiguousarray(v, dtype=np.uint16)

    @property
    def shape(self):
        return self.y.shape

    def padded(self):
        return Frame(pad_plane(self.y, PAD), pad_plane(self.u, PAD // 2), pad_plane(self.v, PAD // 2))

    def copy(self):
        return Frame(self.y.copy(), self.u.copy(), self.v.copy())

    def same(self, o) -> bool:
        return np.array_equal(self.y, o.y) and np.array_equal(self.u, o.u) and np.array_equal(self.v, o.v)


class Analysis:
    def __init__(self, h, w):
        self.rec = Frame(np.zeros((h, w), np.uint16), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint16))
        self.cu = np.zeros((h // 8, w // 8), CU_DTYPE)
        self.coef_y = np.zeros((h, w), np.int16)
        self.coef_u = np.zeros((h // 2, w // 2), np.int16)
        self.coef_v = np.zeros((h // 2, w // 2), np.int16)
        self.me = None
        self.est = 0


def analyze_intra(src: Frame, prm: Params) -> Analysis:
    h, w = src.shape
    a = Analysis(h, w)
    est = C.c_uint64(0)
    lib().orc_analyze_intra_frame(_p(src.y), _p(src.u), _p(src.v), w, w // 2, w, h, C.byref(prm),
                                  _p(a.rec.y), _p(a.rec.u), _p(a.rec.v), w, w // 2, _p(a.cu), _p(a.coef_y), _p(a.coef_u), _p(a.coef_v), C.byref(est))
    a.est = est.value
    return a


def analyze_inter(src: Frame, ref: Frame, prm: Params, centers=None, dump_me=False) -> Analysis:
    """ref: the UNPADDED reference reconstruction; padding happens here"""
    h, w = src.shape
    a = Analysis(h, w)
    rp = ref.padded()
    sy, sc = rp.y.shape[1], rp.u.shape[1]
    by = rp.y.ctypes.data + 2 * (PAD * sy + PAD)
    bu = rp.u.ctypes.data + 2 * (PAD // 2 * sc + PAD // 2)
    bv = rp.v.ctypes.data + 2 * (PAD // 2 * sc + PAD // 2)
    n_ctu = ((w + CTU - 1) // CTU) * ((h + CTU - 1) // CTU)
    cen = None
    if centers is not None:
        cen = np.ascontiguousarray(centers, dtype=np.int16).reshape(n_ctu, 2)
    me = np.zeros((n_ctu, 21, 3), np.int32) if dump_me else None
    est = C.c_uint64(0)
    lib().orc_analyze_inter_frame(_p(src.y), _p(src.u), _p(src.v), w, w // 2,
                                  C.c_void_p(by), C.c_void_p(bu), C.c_void_p(bv), sy, sc, w, h, C.byref(prm),
                                  _p(cen) if cen is not None else None,
                                  _p(a.rec.y), _p(a.rec.u), _p(a.rec.v), w, w // 2, _p(a.cu), _p(a.coef_y), _p(a.coef_u), _p(a.coef_v),
                                  _p(me) if me is not None else None, C.byref(est))
    a.me = me
    a.est = est.value
    return a


def _padded_bases(ref: Frame):
    rp = ref.padded()
    sy, sc = rp.y.shape[1], rp.u.shape[1]
    return rp, sy, sc, (rp.y.ctypes.data + 2 * (PAD * sy + PAD), rp.u.ctypes.data + 2 * (PAD // 2 * sc + PAD // 2), rp.v.ctypes.data + 2 * (PAD // 2 * sc + PAD // 2))


def analyze_b(src: Frame, ref0: Frame, ref1: Frame, prm: Params, centers0=None, centers1=None, dump_me=False) -> Analysis:
    """B picture between two anchors (orc_analyze_b_frame): ref0 = the anchor before it in display order (list 0), ref1 = the one after it (list 1), both
    UNPADDED final reconstructions.  a.me = (list-0 dump, list-1 dump) when dump_me.  CU records of inter CUs: flags & 32 = list 1 used (vector in the
    intra_mode bytes), flags & 64 = list 0 not used."""
    h, w = src.shape
    a = Analysis(h, w)
    keep0, sy, sc, b0 = _padded_bases(ref0)
    keep1, _, _, b1 = _padded_bases(ref1)
    n_ctu = ((w + CTU - 1) // CTU) * ((h + CTU - 1) // CTU)
    cen = [np.ascontiguousarray(c, dtype=np.int16).reshape(n_ctu, 2) if c is not None else None for c in (centers0, centers1)]
    me = [np.zeros((n_ctu, 21, 3), np.int32) if dump_me else None for _ in range(2)]
    est = C.c_uint64(0)
    lib().orc_analyze_b_frame(_p(src.y), _p(src.u), _p(src.v), w, w // 2, *[C.c_void_p(x) for x in b0], *[C.c_void_p(x) for x in b1], sy, sc, w, h, C.byref(prm),
                              _p(cen[0]) if cen[0] is not None else None, _p(cen[1]) if cen[1] is not None else None,
                              _p(a.rec.y), _p(a.rec.u), _p(a.rec.v), w, w // 2, _p(a.cu), _p(a.coef_y), _p(a.coef_u), _p(a.coef_v),
                              _p(me[0]) if dump_me else None, _p(me[1]) if dump_me else None, C.byref(est))
    del keep0, keep1
    a.me = tuple(me) if dump_me else None
    a.est = est.value
    return a


def search_centres(src: Frame, prev: Frame, bit_depth=8) -> np.ndarray:
    """Per-CTU search centres of picture `src` from the 1/4-size pictures of `src` and `prev` (orc_lowres + orc_pre_search).  A session feeds its
    integer search with the centres of the SOURCE picture against the SOURCE picture before it (the whole chunk at once, before any reconstruction
    exists): replaying a session means passing these as `centers` to analyze_inter."""
    h, w = src.shape
    lw, lh = w >> 2, h >> 2
    ls, lr = np.empty((lh, lw), np.uint16), np.empty((lh, lw), np.uint16)
    lib().orc_lowres(_p(src.y), w, w, h, bit_depth, _p(ls))
    lib().orc_lowres(_p(prev.y), w, w, h, bit_depth, _p(lr))
    n_ctu = ((w + CTU - 1) // CTU) * ((h + CTU - 1) // CTU)
    cen = np.zeros((n_ctu, 2), np.int16)
    lib().orc_pre_search(_p(ls), _p(lr), lw, lh, _p(cen))
    return cen


def search_cost(src: Frame, other: Frame, bit_depth=8) -> int:
    """Sum over the CTUs of the smallest SAD of their 8x8 low-resolution block of `src` over the +-14 window in `other` (orc_pre_search_cost): what the
    session's B-picture probe compares between the picture one place and two places before (mihevc_config.bframes = -1)."""
    h, w = src.shape
    lw, lh = w >> 2, h >> 2
    ls, lr = np.empty((lh, lw), np.uint16), np.empty((lh, lw), np.uint16)
    lib().orc_lowres(_p(src.y), w, w, h, bit_depth, _p(ls))
    lib().orc_lowres(_p(other.y), w, w, h, bit_depth, _p(lr))
    n_ctu = ((w + CTU - 1) // CTU) * ((h + CTU - 1) // CTU)
    cen, cost = np.zeros((n_ctu, 2), np.int16), np.zeros(n_ctu, np.uint32)
    lib().orc_pre_search_cost(_p(ls), _p(lr), lw, lh, _p(cen), _p(cost))
    return int(cost.sum())


def deblock(rec: Frame, cu: np.ndarray, bit_depth=8) -> Frame:
    out = rec.copy()
    h, w = out.shape
    cu = np.ascontiguousarray(cu)
    lib().orc_deblock_frame(_p(out.y), _p(out.u), _p(out.v), w, w // 2, w, h, _p(cu), bit_depth, 0)
    return out


def sao(src: Frame, dbk: Frame, prm: Params):
    h, w = src.shape
    out = Frame(np.zeros_like(dbk.y), np.zeros_like(dbk.u), np.zeros_like(dbk.v))
    n_ctu = ((w + CTU - 1) // CTU) * ((h + CTU - 1) // CTU)
    params = np.zeros(n_ctu, SAO_DTYPE)
    lib().orc_sao_frame(_p(src.y), _p(src.u), _p(src.v), w, w // 2, _p(dbk.y), _p(dbk.u), _p(dbk.v), w, w // 2,
                        _p(out.y), _p(out.u), _p(out.v), w, w // 2, w, h, C.byref(prm), _p(params))
    return out, params


def sao_apply(dbk: Frame, params: np.ndarray, bit_depth=8) -> Frame:
    h, w = dbk.shape
    out = Frame(np.zeros_like(dbk.y), np.zeros_like(dbk.u), np.zeros_like(dbk.v))
    params = np.ascontiguousarray(params)
    lib().orc_sao_apply_frame(_p(dbk.y), _p(dbk.u), _p(dbk.v), w, w // 2, _p(out.y), _p(out.u), _p(out.v), w, w // 2,
                              w, h, bit_depth, _p(params))
    return out


# ---------------------------------------------------------------- decoder
class DecodeError(RuntimeError):
    pass


def decode(stream: bytes):
    """-> (frames[list of Frame at coded size], info dict).  Raises DecodeError with the decoder's message."""
    L = lib()
    d = L.orc_dec_open()
    try:
        n = L.orc_dec_decode(d, stream, len(stream))
        if n < 0:
            raise DecodeError(L.orc_dec_error(d).decode())
        w, h, bd, cw, ch = (C.c_int() for _ in range(5))
        L.orc_dec_info(d, C.byref(w), C.byref(h), C.byref(bd), C.byref(cw), C.byref(ch))
        frames = []
        for i in range(n):
            f = Frame(np.zeros((h.value, w.value), np.uint16), np.zeros((h.value // 2, w.value // 2), np.uint16),
                      np.zeros((h.value // 2, w.value // 2), np.uint16))
            L.orc_dec_get_frame(d, i, _p(f.y), _p(f.u), _p(f.v))
            frames.append(f)
        info = {"width": w.value, "height": h.value, "bit_depth": bd.value, "conf_width": cw.value, "conf_height": ch.value}

        def q(name):
            v = C.c_longlong()
            return v.value if L.orc_dec_query(d, name.encode(), C.byref(v)) else None
        info["query"] = q
        # materialise the common fields now (decoder is closed on return)
        for k in ("sps.profile_idc", "sps.level_idc", "sps.tier_flag", "sps.compat", "vui.colour_primaries", "vui.transfer", "vui.matrix",
                  "vui.full_range", "vui.chroma_loc_present", "vui.chroma_loc_top", "vui.num_units_in_tick", "vui.time_scale",
                  "vui.hrd_present", "vps.num_units_in_tick", "vps.time_scale", "count.aud", "count.slices", "sei.137.size", "sei.144.size",
                  "sei.mdcv.gx", "sei.mdcv.gy", "sei.mdcv.bx", "sei.mdcv.by", "sei.mdcv.rx", "sei.mdcv.ry", "sei.mdcv.wpx", "sei.mdcv.wpy",
                  "sei.mdcv.max_lum", "sei.mdcv.min_lum", "sei.cll.max_cll", "sei.cll.max_fall", "pps.init_qp", "pps.tile_cols", "pps.tile_rows", "vui.hrd_present", "hrd.bit_rate_value_minus1", "hrd.cpb_size_value_minus1", "hrd.cbr_flag",
                  "sei.bp.initial_delay", "sei.bp.initial_offset", "count.sei_bp", "count.sei_pt", "sei.pt.au_cpb_removal_delay_minus1", "slice.last_qp",
                  "slice.max_merge", "sps.conf_right", "sps.conf_bottom", "sps.sao", "sps.amp", "sps.strong_intra", "sps.poc_bits",
                  "vps.level_idc", "sps.max_num_reorder", "sps.max_dec_pic_buffering_minus1", "hrd.bit_rate_value_minus1", "hrd.cpb_size_value_minus1", "hrd.bit_rate_scale", "hrd.cpb_size_scale"):
            info[k] = q(k)
        del info["query"]
        return frames, info
    finally:
        L.orc_dec_close(d)
