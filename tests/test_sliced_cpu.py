"""CPU: one picture as several slices (BASELINE configs[4]): full-width bands of CTU rows, each coded as a picture of its own
by its own session / device — motion vectors never reach across a band's edge, in-loop filters stop there, the parameter sets describe
the whole picture and the slice headers carry the bands' addresses.  Checked here without a GPU: the kernel sources (stepped) against the
oracle under the motion constraint, and product CABAC -> merged access units -> the oracle decoder (a normal picture-level decode that
knows nothing about bands) == the bands' reconstructions stacked."""
import ctypes as C

import numpy as np
import pytest

from hevc_amd import _lib
from oracle import oracle as O
from tests import util
from tests.test_kernel_source_stepped import emu  # noqa: F401  (fixture)


def band_frames(h, w, rows, n, bd=8, seed=5):
    """n pictures with strong vertical motion, cut into bands of `rows` CTU rows"""
    full = [util.synth_frame(h, w, seed=seed, shift=(2 * i, 7 * i), bit_depth=bd) for i in range(n)]
    bands, y0 = [], 0
    for r in rows:
        y1 = min(h, y0 + 32 * r)
        bands.append([O.Frame(f.y[y0:y1], f.u[y0 // 2:y1 // 2], f.v[y0 // 2:y1 // 2]) for f in full])
        y0 = y1
    return full, bands


def mv_rows_ok(y, n, my, h, top, bottom):
    ly0, ly1 = y + (my >> 2) - (3 if my & 3 else 0), y + n - 1 + (my >> 2) + (4 if my & 3 else 0)
    cy0, cy1 = (y >> 1) + (my >> 3) - (1 if my & 7 else 0), (y >> 1) + (n >> 1) - 1 + (my >> 3) + (2 if my & 7 else 0)
    return not (top and (ly0 < 0 or cy0 < 0)) and not (bottom and (ly1 > h - 1 or cy1 > (h >> 1) - 1))


@pytest.mark.parametrize("w,h,bd,top,bottom,pre", [(160, 96, 8, 1, 1, 1), (136, 72, 8, 1, 0, 0), (96, 128, 10, 0, 1, 1)])
def test_motion_constrained_analysis_kernels_equal_oracle(emu, w, h, bd, top, bottom, pre):
    prm = O.default_params(27, bd, 12)
    prm.mc_top, prm.mc_bottom, prm.pre_search, prm.rdo_zero = top, bottom, pre, 1
    a, b = util.synth_frame(h, w, 5, bit_depth=bd), util.synth_frame(h, w, 5, shift=(3, 21), bit_depth=bd)      # 21 rows of vertical motion
    ref = O.sao(a, O.deblock(O.analyze_intra(a, prm).rec, O.analyze_intra(a, prm).cu, bd), prm)[0]
    want = O.analyze_inter(b, ref, prm, dump_me=True)
    got = emu.inter(b, ref, prm)
    assert np.array_equal(want.me, got.me) and util.same_analysis(want, got), util.describe_diff(want, got)
    # every coded vector keeps its block's filter taps inside the picture on the constrained sides, and the constraint binds somewhere
    free = O.default_params(27, bd, 12)
    free.pre_search, free.rdo_zero = pre, 1
    assert not np.array_equal(O.analyze_inter(b, ref, free).cu["mvy"], want.cu["mvy"])
    for (by, bx), r in np.ndenumerate(want.cu):
        n = 1 << int(r["log2_size"])
        y = (by * 8) & ~(n - 1)
        assert mv_rows_ok(y, n, int(r["mvy"]), h, top, bottom), (bx, by, r)


def sliced_cfg(w, h, bd, rows, k, level=120, **kw):
    cfg = _lib.default_config()
    cfg.width, cfg.bit_depth, cfg.level_idc = w, bd, level
    cfg.pic_height, cfg.slice_count, cfg.slice_index = h, len(rows), k
    for i, r in enumerate(rows):
        cfg.slice_ctu_rows[i] = r
    y0 = 32 * sum(rows[:k])
    cfg.height = min(h, y0 + 32 * rows[k]) - y0
    for key, v in kw.items():
        setattr(cfg, key, v)
    return cfg


@pytest.mark.parametrize("w,h,bd,rows,level,keyint", [(160, 96, 8, (2, 1), 63, 3), (544, 320, 8, (4, 3, 3), 120, 2), (320, 200, 10, (3, 4), 93, 4)])
def test_sliced_pictures_decode_to_the_stacked_band_reconstructions(w, h, bd, rows, level, keyint):
    lib = _lib.load()
    n = 5
    full, bands = band_frames(h, w, rows, n, bd)
    buf = (C.c_uint8 * (4 << 20))()
    cfgs = [sliced_cfg(w, h, bd, rows, k, level, aud=1) for k in range(len(rows))]
    heads = []
    for cfg in cfgs:
        m = lib.mihevc_write_parameter_sets(C.byref(cfg), buf, len(buf))
        assert m > 0
        heads.append(bytes(buf[:m]))
    assert len(set(heads)) == 1, "every slice's session must write the same parameter sets"
    prm_i, prm_p = O.default_params(24, bd, 12), O.default_params(27, bd, 12)
    refs = [None] * len(rows)
    recs = [[None] * len(rows) for _ in range(n)]
    stream = b""
    for i in range(n):
        intra = i % keyint == 0
        au = b""
        for k, cfg in enumerate(cfgs):
            prm = O.Params.from_buffer_copy(bytes(prm_i if intra else prm_p))
            prm.mc_top, prm.mc_bottom = int(k > 0), int(k < len(rows) - 1)
            if intra:
                prm.tile_cols, prm.tile_rows = _lib.tile_grid(cfg)
            src = bands[k][i]
            a = O.analyze_intra(src, prm) if intra else O.analyze_inter(src, refs[k], prm)
            refs[k], sao = O.sao(src, O.deblock(a.rec, a.cu, bd), prm)
            recs[i][k] = refs[k]
            m = lib.mihevc_encode_picture_host(C.byref(cfg), 2 if intra else 1, i % keyint, prm.qp, util.ptr(a.cu), util.ptr(a.coef_y), util.ptr(a.coef_u),
                                               util.ptr(a.coef_v), util.ptr(sao), buf, len(buf))
            assert m > 0, m
            pkt = bytes(buf[:m])
            if k == 0:       # the first slice's packet opens the access unit: AUD, then the parameter sets of the stream's first picture
                cut = pkt.index(b"\0\0\0\1", 4)
                pkt = pkt[:cut] + (heads[0] if i == 0 else b"") + pkt[cut:]
            else:            # later slices contribute their slice NAL only
                pkt = pkt[pkt.index(b"\0\0\0\1", 4):]
            au += pkt
        stream += au
    dec, info = O.decode(stream)
    assert len(dec) == n and info["count.slices"] == n * len(rows) and info["count.aud"] == n
    assert (info["width"], info["conf_height"]) == (w, h)
    for i in range(n):
        want = O.Frame(np.vstack([r.y for r in recs[i]]), np.vstack([r.u for r in recs[i]]), np.vstack([r.v for r in recs[i]]))
        ch = dec[i].y.shape[0]
        assert ch >= want.y.shape[0]
        got = O.Frame(dec[i].y[:want.y.shape[0]], dec[i].u[:want.u.shape[0]], dec[i].v[:want.v.shape[0]])
        assert got.same(want), f"picture {i}: decoded picture != the slices' reconstructions"
    for f, d in zip(full, dec):
        assert util.psnr(d.y[:h], f.y, peak=(1 << bd) - 1.0) > 28.0


@pytest.mark.parametrize("w,h,bd,rows", [(96, 128, 8, (2, 1, 1)), (72, 96, 10, (1, 2))])
def test_deblocking_of_a_band_extended_by_its_neighbours_rows_equals_the_whole_picture(emu, w, h, bd, rows):
    """DeblockArgs::y_org: a band deblocked as the picture [8 rows of the slice above | band | 8 rows of the slice below] (what the X1 exchange of
    csrc/slice_group.h builds) gets exactly the whole picture's deblocked rows, and so do the rows next to the seams that SAO will look at."""
    prm = O.default_params(32, bd, 8)
    a = O.analyze_intra(util.synth_frame(h, w, 4, bit_depth=bd), prm)
    b = O.analyze_inter(util.synth_frame(h, w, 4, shift=(3, 5), bit_depth=bd), O.deblock(a.rec, a.cu, bd), prm)
    for an in (a, b):
        want = O.deblock(an.rec, an.cu, bd)
        r0 = 0
        for k, r in enumerate(rows):
            y0, y1 = 32 * r0, min(h, 32 * (r0 + r))
            up, dn = int(k > 0), int(k < len(rows) - 1)
            pl = util.planes(an.rec, bd)                      # a fresh copy of the pre-deblock picture: the band filters its own extended picture in place
            rc = emu.lib.emu_deblock_band(util.ptr(pl[0]), util.ptr(pl[1]), util.ptr(pl[2]), w, y0 - 8 * up, (y1 - y0) + 8 * (up + dn), 8 * up, util.ptr(np.ascontiguousarray(an.cu)), bd)
            assert rc == 0
            got = util.to_frame(pl)
            lo, hi = y0 - up, y1 + dn                         # the band's rows and the one row either side SAO reads
            assert np.array_equal(got.y[lo:hi], want.y[lo:hi]) and np.array_equal(got.u[(lo + 1) // 2:hi // 2], want.u[(lo + 1) // 2:hi // 2]) and \
                np.array_equal(got.v[(lo + 1) // 2:hi // 2], want.v[(lo + 1) // 2:hi // 2]), (k, "inter" if an is b else "intra")
            r0 += r


@pytest.mark.parametrize("w,h,bd,rows", [(96, 128, 8, (2, 1, 1)), (72, 96, 10, (1, 2))])
def test_sao_of_a_band_with_halo_rows_equals_the_whole_picture(emu, w, h, bd, rows):
    """SaoArgs::halo (csrc/slice_group.h): a band that finds its neighbours' deblocked rows above / below decides and applies SAO exactly as the
    whole picture would for the band's CTUs — statistics and edge classes look across the seam."""
    prm = O.default_params(30, bd, 8)
    src = util.synth_frame(h, w, 9, bit_depth=bd)
    a = O.analyze_intra(src, prm)
    dbk = O.deblock(a.rec, a.cu, bd)
    want, want_sp = O.sao(src, dbk, prm)
    s, d = util.planes(src, bd), util.planes(dbk, bd)
    out = [np.zeros_like(x) for x in s]
    wc = (w + 31) // 32
    got_sp = np.zeros(wc * ((h + 31) // 32), O.SAO_DTYPE)
    r0 = 0
    for k, r in enumerate(rows):
        y0, bh = 32 * r0, min(h, 32 * (r0 + r)) - 32 * r0
        halo = (1 if k > 0 else 0) | (2 if k < len(rows) - 1 else 0)
        sp = np.zeros(wc * r, O.SAO_DTYPE)
        rc = emu.lib.emu_sao_band(util.ptr(s[0]), util.ptr(s[1]), util.ptr(s[2]), util.ptr(d[0]), util.ptr(d[1]), util.ptr(d[2]), w, y0, bh, halo, C.byref(prm),
                                  util.ptr(out[0]), util.ptr(out[1]), util.ptr(out[2]), util.ptr(sp))
        assert rc == 0
        got_sp[wc * r0:wc * (r0 + r)] = sp
        r0 += r
    assert np.array_equal(got_sp, want_sp)
    assert util.to_frame(out).same(want)
    assert (want_sp["type"] != 0).any()


def halo_pipeline(full, rows, cfgs, prm_i, prm_p, keyint, bd, prm_of=None, idr_at=None):
    """The definition of what slices that exchange rows (cfg.slice_halo) must produce, from the oracle's whole-picture stages: IDR pictures are analysed band by
    band (a slice boundary ends intra prediction, and every slice has its own tile rows), everything else is the WHOLE picture's pipeline — motion search and
    compensation, deblocking and SAO across the seams — with the search centres found band by band (each band searches its own 1/4-size pictures).
    Yields (intra, analysis of the whole picture, sao parameters, final reconstruction) per picture.  prm_of(i, intra) -> Params and idr_at (picture
    indices) replace prm_i / prm_p / keyint when a session's own per-picture QPs and GOP layout are replayed."""
    h, w = full[0].shape
    ys = [0]
    for r in rows:
        ys.append(min(h, ys[-1] + 32 * r))
    band = lambda f, k: O.Frame(f.y[ys[k]:ys[k + 1]], f.u[ys[k] // 2:ys[k + 1] // 2], f.v[ys[k] // 2:ys[k + 1] // 2])      # noqa: E731
    ref = None
    for i, src in enumerate(full):
        intra = (i in idr_at) if idr_at is not None else i % keyint == 0
        if prm_of is not None:
            prm_i = prm_p = prm_of(i, intra)
        if intra:
            parts = []
            for k, cfg in enumerate(cfgs):
                prm = O.Params.from_buffer_copy(bytes(prm_i))
                prm.tile_cols, prm.tile_rows = _lib.tile_grid(cfg)
                parts.append(O.analyze_intra(band(src, k), prm))
            a = O.Analysis(h, w)
            a.rec = O.Frame(np.vstack([p.rec.y for p in parts]), np.vstack([p.rec.u for p in parts]), np.vstack([p.rec.v for p in parts]))
            a.cu = np.vstack([p.cu for p in parts])
            a.coef_y, a.coef_u, a.coef_v = (np.vstack([getattr(p, n) for p in parts]) for n in ("coef_y", "coef_u", "coef_v"))
            a.est = sum(p.est for p in parts)
            prm = prm_i
        else:
            cen = np.vstack([O.search_centres(band(src, k), band(full[i - 1], k), bd) for k in range(len(rows))]) if prm_p.pre_search else None
            a = O.analyze_inter(src, ref, prm_p, centers=cen)
            prm = prm_p
        ref, sao = O.sao(src, O.deblock(a.rec, a.cu, bd), prm)
        yield intra, a, sao, ref


@pytest.mark.parametrize("w,h,bd,rows,level,keyint", [(160, 96, 8, (2, 1), 63, 3), (544, 320, 8, (4, 3, 3), 120, 2), (320, 200, 10, (3, 4), 93, 4), (96, 160, 8, (1, 1, 1, 2), 63, 5)])
def test_slices_with_filters_and_motion_across_the_seams(w, h, bd, rows, level, keyint):
    """cfg.slice_halo on the host side: pps / slice_loop_filter_across_slices_enabled_flag = 1, every band's coder still sees only its band (merge / AMVP /
    MPM / SAO merge end at the slice boundary).  The merged stream, decoded as ordinary pictures, equals the whole-picture pipeline above — whose motion
    vectors cross the seams freely (asserted) and whose in-loop filters run across them (the decoder's own arithmetic has to do the same)."""
    lib = _lib.load()
    n = 5
    full, _ = band_frames(h, w, rows, n, bd)
    buf = (C.c_uint8 * (4 << 20))()
    cfgs = [sliced_cfg(w, h, bd, rows, k, level, aud=1, slice_halo=1, slice_group=7) for k in range(len(rows))]
    m = lib.mihevc_write_parameter_sets(C.byref(cfgs[0]), buf, len(buf))
    head = bytes(buf[:m])
    prm_i, prm_p = O.default_params(24, bd, 12), O.default_params(27, bd, 12)
    prm_p.pre_search = prm_p.rdo_zero = 1
    ys = [0]
    for r in rows:
        ys.append(min(h, ys[-1] + 32 * r))
    wc = (w + 31) // 32
    stream, recs, crossing = b"", [], 0
    for i, (intra, a, sao, ref) in enumerate(halo_pipeline(full, rows, cfgs, prm_i, prm_p, keyint, bd)):
        au = b""
        for k, cfg in enumerate(cfgs):
            y0, y1 = ys[k], ys[k + 1]
            cu = np.ascontiguousarray(a.cu[y0 // 8:(y1 + 7) // 8])
            cy, cu_, cv = (np.ascontiguousarray(x) for x in (a.coef_y[y0:y1], a.coef_u[y0 // 2:y1 // 2], a.coef_v[y0 // 2:y1 // 2]))
            sp = np.ascontiguousarray(sao[wc * sum(rows[:k]):wc * sum(rows[:k + 1])])
            m = lib.mihevc_encode_picture_host(C.byref(cfg), 2 if intra else 1, i % keyint, (prm_i if intra else prm_p).qp, util.ptr(cu), util.ptr(cy), util.ptr(cu_),
                                               util.ptr(cv), util.ptr(sp), buf, len(buf))
            assert m > 0, m
            pkt = bytes(buf[:m])
            if k == 0:
                cut = pkt.index(b"\0\0\0\1", 4)
                pkt = pkt[:cut] + (head if i == 0 else b"") + pkt[cut:]
            else:
                pkt = pkt[pkt.index(b"\0\0\0\1", 4):]
            au += pkt
            if not intra:       # vectors that reach across the band's edges
                for (by, bx), r in np.ndenumerate(cu):
                    nn = 1 << int(r["log2_size"])
                    yy = (by * 8) & ~(nn - 1)
                    crossing += not mv_rows_ok(yy, nn, int(r["mvy"]), y1 - y0, k > 0, k < len(rows) - 1)
        stream += au
        recs.append(ref)
    assert crossing > 0, "no motion vector crosses a seam: the case is not exercised"
    dec, info = O.decode(stream)
    assert len(dec) == n and info["count.slices"] == n * len(rows)
    for i in range(n):
        got = O.Frame(dec[i].y[:h], dec[i].u[:h // 2], dec[i].v[:h // 2])
        assert got.same(recs[i]), f"picture {i}: decoded picture != the whole-picture pipeline"
