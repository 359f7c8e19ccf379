"""CPU: the product's host entropy coder / parameter-set writer (libmihevc.so, no device needed) fed with symbols
from the oracle analysis, decoded by the oracle decoder: the decoded pictures must equal the oracle's
reconstruction bit for bit.  This is the conformance gate available without a third-party decoder."""
import ctypes as C

import numpy as np
import pytest

from hevc_amd import _lib, mp4
from oracle import oracle as O
from tests import util


def encode_pictures(cfg, srcs, qp, bd, me_range=8, keyint=1000, nxn=0, intra_in_p=0):
    lib = _lib.load()
    buf = (C.c_uint8 * (4 << 20))()
    n = lib.mihevc_write_parameter_sets(C.byref(cfg), buf, len(buf))
    assert n > 0
    headers = bytes(buf[:n])
    stream, recs, packets, ref = b"", [], [], None
    prm_i, prm_p = O.default_params(max(0, qp - 3), bd, me_range), O.default_params(qp, bd, me_range)
    prm_i.tile_cols, prm_i.tile_rows = _lib.tile_grid(cfg)       # IDR pictures are analysed for the grid PPS 1 signals
    prm_i.intra_nxn = prm_p.intra_nxn = nxn
    prm_p.intra_in_p = intra_in_p
    prm_p.tile_cols, prm_p.tile_rows = _lib.p_tile_grid(cfg)   # P pictures: PPS 0's own grid (cfg.p_tiles), which the intra second pass must respect
    cus = []
    for i, src in enumerate(srcs):
        intra = i % keyint == 0
        prm = prm_i if intra else prm_p
        a = O.analyze_intra(src, prm) if intra else O.analyze_inter(src, ref, prm)
        dbk = O.deblock(a.rec, a.cu, bd)
        ref, sao = O.sao(src, dbk, prm) if cfg.sao else (dbk, None)
        n = lib.mihevc_encode_picture_host(C.byref(cfg), 2 if intra else 1, i % keyint, prm.qp, util.ptr(a.cu), util.ptr(a.coef_y), util.ptr(a.coef_u),
                                           util.ptr(a.coef_v), util.ptr(sao) if cfg.sao else None, buf, len(buf))
        assert n > 0, n
        packets.append((bytes(buf[:n]), i, intra))
        cus.append(a.cu)
        pkt = packets[-1][0]
        if i == 0:          # parameter sets belong to the first access unit, after its AUD when there is one (7.4.2.4.4)
            cut = pkt.index(b"\0\0\0\1", 4) if cfg.aud else 0
            pkt = pkt[:cut] + headers + pkt[cut:]
        stream += pkt
        recs.append(ref)
    encode_pictures.last_cus = cus
    return headers, stream, recs, packets


def make_cfg(w, h, bd=8, **kw):
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth = w, h, bd
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("w,h,qp,bd,n,keyint", [(64, 64, 30, 8, 3, 100), (96, 80, 20, 8, 4, 2), (136, 72, 36, 8, 3, 100), (72, 104, 26, 10, 3, 100),
                                                 (160, 96, 12, 8, 2, 100)])
def test_stream_decodes_to_the_oracle_reconstruction(w, h, qp, bd, n, keyint):
    cfg = make_cfg(w, h, bd, aud=1)
    srcs = [util.synth_frame(h, w, seed=11, shift=(3 * i, i), bit_depth=bd) for i in range(n)]
    _, stream, recs, _ = encode_pictures(cfg, srcs, qp, bd, keyint=keyint)
    frames, info = O.decode(stream)
    assert len(frames) == n and info["count.aud"] == n and info["bit_depth"] == bd
    for i, (f, r) in enumerate(zip(frames, recs)):
        assert f.same(r), f"picture {i} differs after decode"


@pytest.mark.parametrize("w,h,level,grid,qp,bd", [(512, 128, 120, (2, 2), 30, 8), (544, 160, 120, (2, 2), 24, 8), (800, 224, 93, (3, 3), 34, 8),
                                                  (512, 192, 150, (2, 3), 28, 10), (512, 128, 63, (1, 1), 30, 8)])
def test_idr_tiles_decode_to_the_oracle_reconstruction(w, h, level, grid, qp, bd):
    # IDR pictures use PPS 1 (uniform tile grid, as many tiles as Table A.8 / A.4.1 allow): prediction, MPM, CABAC contexts and
    # SAO merge stop at tile borders, every tile is its own CABAC substream behind an entry point.  P pictures keep PPS 0.
    cfg = make_cfg(w, h, bd, level_idc=level)
    assert _lib.tile_grid(cfg) == grid
    srcs = [util.synth_frame(h, w, seed=21, shift=(2 * i, i), bit_depth=bd) for i in range(3)]
    _, stream, recs, packets = encode_pictures(cfg, srcs, qp, bd, keyint=2)
    frames, info = O.decode(stream)
    assert len(frames) == 3
    for i, (f, r) in enumerate(zip(frames, recs)):
        assert f.same(r), f"picture {i} differs after decode"
    if grid != (1, 1):
        assert (info["pps.tile_cols"], info["pps.tile_rows"]) == grid
    # the same pictures without tiles: different bits (prediction crosses the borders), same machinery
    cfg0 = make_cfg(w, h, bd, level_idc=level, intra_tiles=0)
    assert _lib.tile_grid(cfg0) == (1, 1)
    _, stream0, recs0, _ = encode_pictures(cfg0, srcs, qp, bd, keyint=2)
    frames0, _ = O.decode(stream0)
    assert all(f.same(r) for f, r in zip(frames0, recs0))
    if grid != (1, 1):
        assert stream0 != stream


def test_tile_entry_points_are_checked():
    # flipping a bit inside the entry-point table must be caught by the decoder's substream-size check
    cfg = make_cfg(512, 128, level_idc=120)
    _, stream, _, packets = encode_pictures(cfg, [util.synth_frame(128, 512, seed=3)], 30, 8)
    frames, _ = O.decode(stream)
    assert len(frames) == 1
    headers_len = len(stream) - len(packets[0][0])
    bad = bytearray(stream)
    bad[headers_len + 4 + 2 + 2] ^= 0x04         # start code (4) + NAL header (2) + 2 bytes into the slice header: offset bits
    with pytest.raises(Exception):
        O.decode(bytes(bad))


@pytest.mark.parametrize("w,h,qp,bd", [(64, 64, 22, 8), (136, 72, 30, 8), (72, 104, 18, 10), (544, 160, 26, 8)])
def test_intra_nxn_with_dst_decodes_to_the_oracle_reconstruction(w, h, qp, bd):
    # 8x8 intra CUs may split into four 4x4 PUs (part_mode NxN): four modes with PU-level MPM derivation, DST-VII 4x4 luma
    # TUs with mode-dependent scans, 4x4 chroma TUs at the CU level, cbf per 4x4 block; also inside the IDR tile grid
    cfg = make_cfg(w, h, bd)
    srcs = [util.synth_frame(h, w, seed=31, shift=(i, 2 * i), bit_depth=bd) for i in range(2)]
    _, stream, recs, _ = encode_pictures(cfg, srcs, qp, bd, keyint=1, nxn=1)
    cus = encode_pictures.last_cus
    assert any((cu["flags"] & 16).any() for cu in cus), "the content must make the analysis choose NxN somewhere"
    nx = cus[0][(cus[0]["flags"] & 16) != 0]
    assert (nx["log2_size"] == 3).all() and len({tuple(m) for m in nx["intra_mode"]}) > 1
    frames, _ = O.decode(stream)
    for i, (f, r) in enumerate(zip(frames, recs)):
        assert f.same(r), f"picture {i} differs after decode"
    _, stream0, _, _ = encode_pictures(cfg, srcs, qp, bd, keyint=1, nxn=0)
    assert len(stream) != len(stream0)


def occluded_clip(w, h, bd, n=3):
    """Frame i = the texture shifted by (2i, i) with a fresh, unrelated patch pasted in: nothing in the reference predicts the patch."""
    out = []
    for i in range(n):
        f = util.synth_frame(h, w, seed=41, shift=(2 * i, i), bit_depth=bd)
        if i:
            g = util.synth_frame(h, w, seed=900 + i, bit_depth=bd)
            x0, y0, pw, ph = 32 * (i % 2), 32, min(w, 96), min(h - 32, 64)
            f.y[y0:y0 + ph, x0:x0 + pw] = g.y[y0:y0 + ph, x0:x0 + pw]
            f.u[y0 // 2:(y0 + ph) // 2, x0 // 2:(x0 + pw) // 2] = g.u[y0 // 2:(y0 + ph) // 2, x0 // 2:(x0 + pw) // 2]
            f.v[y0 // 2:(y0 + ph) // 2, x0 // 2:(x0 + pw) // 2] = g.v[y0 // 2:(y0 + ph) // 2, x0 // 2:(x0 + pw) // 2]
        out.append(f)
    return out


@pytest.mark.parametrize("w,h,qp,bd,nxn", [(160, 128, 28, 8, 0), (136, 104, 24, 8, 1), (128, 96, 30, 10, 0)])
def test_intra_ctus_in_p_pictures_decode_to_the_oracle_reconstruction(w, h, qp, bd, nxn):
    # second pass of P pictures: CTUs the reference cannot predict are re-coded as intra (pred_mode_flag = 1 inside a P slice:
    # MPM from inter neighbours = DC, no merge/AMVP candidates from intra neighbours, Bs 2 at their edges)
    cfg = make_cfg(w, h, bd)
    srcs = occluded_clip(w, h, bd)
    _, stream, recs, _ = encode_pictures(cfg, srcs, qp, bd, nxn=nxn, intra_in_p=1)
    cus = encode_pictures.last_cus
    n_intra = [int(((cu["flags"] & 1) == 0).sum()) for cu in cus[1:]]
    assert all(n > 0 for n in n_intra), n_intra                      # the pasted patch went intra in both P pictures
    assert all(((cu["flags"] & 1) != 0).any() for cu in cus[1:])     # ... and the rest stayed inter
    frames, _ = O.decode(stream)
    for i, (f, r) in enumerate(zip(frames, recs)):
        assert f.same(r), f"picture {i} differs after decode"
    _, stream0, recs0, _ = encode_pictures(cfg, srcs, qp, bd, nxn=nxn, intra_in_p=0)
    assert len(stream) < 1.02 * len(stream0)                         # chosen by J = D + lambda R: never clearly more bits ...
    assert util.psnr(recs[2].y, srcs[2].y) >= util.psnr(recs0[2].y, srcs[2].y) - 0.3       # ... and never clearly worse pictures


def test_sao_off_and_skip_heavy_static_content():
    w, h = 96, 64
    cfg = make_cfg(w, h, sao=0)
    still = util.synth_frame(h, w, seed=2, detail=False)
    srcs = [still, still.copy(), still.copy()]            # identical pictures -> P pictures are all skip/merge
    _, stream, recs, packets = encode_pictures(cfg, srcs, 32, 8)
    frames, info = O.decode(stream)
    assert info["sps.sao"] == 0 and all(f.same(r) for f, r in zip(frames, recs))
    assert len(packets[1][0]) < 40 and len(packets[2][0]) < 40       # a skipped 96x64 picture is a handful of bytes


def test_parameter_sets_carry_the_operating_point():
    # SDR 1080p: Main, level 4 (idc 120), main tier, bt709, limited range, 30 fps.  1080 = 135 * 8, so the coded size IS
    # 1920x1080 (MinCb 8) with a partial last CTU row and no conformance window.
    cfg = make_cfg(1920, 1080)
    srcs = [util.synth_frame(1080, 1920, seed=1, detail=False)]
    headers, stream, recs, _ = encode_pictures(cfg, srcs, 38, 8)
    frames, info = O.decode(stream)
    assert frames[0].same(recs[0])
    assert (info["width"], info["height"], info["conf_width"], info["conf_height"]) == (1920, 1080, 1920, 1080)
    assert (info["sps.profile_idc"], info["sps.level_idc"], info["sps.tier_flag"], info["vps.level_idc"]) == (1, 120, 0, 120)
    assert info["sps.compat"] == 0x60000000
    assert (info["vui.colour_primaries"], info["vui.transfer"], info["vui.matrix"], info["vui.full_range"]) == (1, 1, 1, 0)
    assert (info["vui.num_units_in_tick"], info["vui.time_scale"], info["vps.time_scale"]) == (1, 30, 30)
    assert info["vui.chroma_loc_present"] == 0 and info["sei.137.size"] is None and info["count.aud"] == 0
    assert info["slice.max_merge"] is None or info["slice.max_merge"] == 5


def test_conformance_window_for_sizes_off_the_8_grid():
    w, h = 100, 60                        # coded 104 x 64, cropped by 4 luma samples right and bottom
    cfg = make_cfg(w, h)
    srcs = [util.synth_frame(64, 104, seed=6), util.synth_frame(64, 104, seed=6, shift=(2, 1))]
    _, stream, recs, _ = encode_pictures(cfg, srcs, 30, 8)
    frames, info = O.decode(stream)
    assert (info["width"], info["height"], info["conf_width"], info["conf_height"]) == (104, 64, 100, 60)
    assert (info["sps.conf_right"], info["sps.conf_bottom"]) == (2, 2)          # in chroma units
    assert all(f.same(r) for f, r in zip(frames, recs))


def test_hdr10_signalling():
    # the HDR set of core/utils.py:58-69: bt2020 / smpte2084 / bt2020nc, chromaloc 0, aud, SEI 137 + 144 with the defaults
    cfg = make_cfg(64, 64, 10, hdr10=1, colour_primaries=9, transfer=16, matrix=9, chroma_loc=0, aud=1, repeat_headers=1, level_idc=150, hrd=1,
                   vbv_maxrate_kbps=11760, vbv_bufsize_kbits=14112)
    cfg.fps_num, cfg.fps_den = 30000, 1001
    srcs = [util.synth_frame(64, 64, seed=4, bit_depth=10)]
    _, stream, recs, _ = encode_pictures(cfg, srcs, 24, 10)
    frames, info = O.decode(stream)
    assert frames[0].same(recs[0])
    assert (info["sps.profile_idc"], info["sps.level_idc"], info["sps.compat"]) == (2, 150, 0x20000000)
    assert (info["vui.colour_primaries"], info["vui.transfer"], info["vui.matrix"]) == (9, 16, 9)
    assert info["vui.chroma_loc_present"] == 1 and info["vui.chroma_loc_top"] == 0
    assert (info["vui.num_units_in_tick"], info["vui.time_scale"]) == (1001, 30000)
    assert (info["sei.137.size"], info["sei.144.size"]) == (24, 4)
    assert (info["sei.mdcv.gx"], info["sei.mdcv.gy"], info["sei.mdcv.bx"], info["sei.mdcv.by"], info["sei.mdcv.rx"], info["sei.mdcv.ry"]) == \
           (13250, 34500, 7500, 3000, 34000, 16000)
    assert (info["sei.mdcv.wpx"], info["sei.mdcv.wpy"], info["sei.mdcv.max_lum"], info["sei.mdcv.min_lum"]) == (15635, 16450, 10000000, 50)
    assert (info["sei.cll.max_cll"], info["sei.cll.max_fall"]) == (1000, 400)
    # hrd=1: NAL HRD parameters in the VUI, bit rate in 64 bit/s units and CPB size in 16 bit units (scales 0), VBR
    assert info["vui.hrd_present"] == 1 and info["hrd.cbr_flag"] == 0
    assert (info["hrd.bit_rate_value_minus1"] + 1) * 64 == 11760000 and (info["hrd.cpb_size_value_minus1"] + 1) * 16 == 14112000


def test_decoder_rejects_corruption():
    cfg = make_cfg(64, 64)
    _, stream, _, _ = encode_pictures(cfg, [util.synth_frame(64, 64, seed=1)], 30, 8)
    bad = bytearray(stream)
    bad[len(bad) // 2] ^= 0x55
    try:
        frames, _ = O.decode(bytes(bad))
        ok = False          # may still parse; then the picture must differ or the decoder must have complained
    except O.DecodeError:
        ok = True
    assert ok or frames is not None


def test_mp4_hvc1_container(tmp_path):
    w, h, n = 96, 80, 5
    cfg = make_cfg(w, h, aud=1)
    srcs = [util.synth_frame(h, w, seed=8, shift=(i, 0)) for i in range(n)]
    headers, stream, recs, packets = encode_pictures(cfg, srcs, 28, 8, keyint=3)
    out = tmp_path / "clip.mp4"
    mux = mp4.Mp4Writer(out, cfg)
    for i, (data, pts, key) in enumerate(packets):
        mux.add_sample((headers if key else b"") + data, pts, key)      # keyframe packets may repeat the headers: the muxer drops them
    mux.finish(headers)
    data = out.read_bytes()
    top = mp4.parse_boxes(data)
    assert [b[0] for b in top] == ["ftyp", "moov", "mdat"]               # faststart: moov first
    assert data[top[0][1]:top[0][1] + 4] == b"mp42"                      # -brand mp42
    def find(path, start, end):
        for name in path:
            nxt = [b for b in mp4.parse_boxes(data, start, end) if b[0] == name]
            assert nxt, name
            _, start, end = nxt[0]
        return start, end
    s, e = find(["moov", "trak", "mdia", "minf", "stbl", "stsd"], 0, len(data))
    entries = mp4.parse_boxes(data, s + 8, e)
    assert entries[0][0] == "hvc1"                                       # -tag:v hvc1
    inner = {b[0]: b for b in mp4.parse_boxes(data, entries[0][1] + 78, entries[0][2])}
    assert {"hvcC", "colr"} <= set(inner)
    assert data[inner["colr"][1]:inner["colr"][1] + 4] == b"nclx"        # +write_colr
    hs, he = find(["moov", "trak", "mdia", "hdlr"], 0, len(data))
    assert b"VideoHandler" in data[hs:he]
    # samples: length-prefixed NALs without parameter sets; rebuild Annex-B and decode
    ss, se = find(["moov", "trak", "mdia", "minf", "stbl", "stsz"], 0, len(data))
    count = int.from_bytes(data[ss + 8:ss + 12], "big")
    sizes = [int.from_bytes(data[ss + 12 + 4 * i:ss + 16 + 4 * i], "big") for i in range(count)]
    cs, ce = find(["moov", "trak", "mdia", "minf", "stbl", "stco"], 0, len(data))
    off = int.from_bytes(data[cs + 8:cs + 12], "big")
    assert off == top[2][1] and sum(sizes) == top[2][2] - top[2][1] and count == n
    ks, ke = find(["moov", "trak", "mdia", "minf", "stbl", "stss"], 0, len(data))
    assert [int.from_bytes(data[ks + 8 + 4 * i:ks + 12 + 4 * i], "big") for i in range(int.from_bytes(data[ks + 4:ks + 8], "big"))] == [1, 4]
    annexb = b""
    p = off
    for sz in sizes:
        q = p
        while q < p + sz:
            ln = int.from_bytes(data[q:q + 4], "big")
            nal = data[q + 4:q + 4 + ln]
            assert mp4.nal_type(nal) not in (32, 33, 34)
            annexb += b"\0\0\0\1" + nal
            if q == off:            # the out-of-band parameter sets go after the first sample's AUD (it must lead its access unit)
                assert mp4.nal_type(nal) == 35
                annexb += headers
            q += 4 + ln
        p += sz
    frames, info = O.decode(annexb)
    assert len(frames) == n and all(f.same(r) for f, r in zip(frames, recs))
    # hvcC carries exactly our VPS/SPS/PPS
    hv = data[inner["hvcC"][1]:inner["hvcC"][2]]
    assert hv[0] == 1 and hv[1] == 1 and hv[12] == 120 and hv[22] == 3     # version, Main, level 4.0, three arrays
    for nal in mp4.split_annexb(headers):
        assert nal in hv


@pytest.mark.parametrize("w,h,level,grid,qp,bd,ipass", [(544, 320, 120, (2, 2), 30, 8, 0), (544, 320, 120, (2, 2), 34, 8, 1), (800, 224, 93, (2, 2), 26, 10, 0), (1056, 160, 150, (2, 2), 30, 8, 0)])
def test_p_picture_tiles_decode_to_the_oracle_reconstruction(w, h, level, grid, qp, bd, ipass):
    """cfg.p_tiles: P pictures carry a tile grid of their own in PPS 0 (one CABAC substream and one host job per tile).  Merge / AMVP candidates, the
    skip / split contexts and SAO merge stop at tile borders, motion compensation and the in-loop filters do not; IDR pictures keep PPS 1.  The
    pieces (encode_tiles per range + assemble_picture) are what the session's host jobs run; here the whole-picture entry point codes them."""
    cfg = make_cfg(w, h, bd, level_idc=level, p_tiles=1, aud=1)
    assert _lib.p_tile_grid(cfg) == grid
    off = make_cfg(w, h, bd, level_idc=level, p_tiles=0, aud=1)
    assert _lib.p_tile_grid(off) == (1, 1)
    srcs = [util.synth_frame(h, w, seed=31, shift=(5 * i, 2 * i), bit_depth=bd) for i in range(4)]
    _, stream, recs, packets = encode_pictures(cfg, srcs, qp, bd, keyint=3, intra_in_p=ipass)
    frames, info = O.decode(stream)
    assert len(frames) == 4 and info["count.aud"] == 4
    for i, (f, r) in enumerate(zip(frames, recs)):
        assert f.same(r), f"picture {i} differs after decode"
    if ipass:
        assert any((~cu["flags"][1:] & 1).any() for cu in encode_pictures.last_cus[1:3]), "no intra CU in a P picture: the case is not exercised"
    # the same pictures without P tiles: same reconstruction when no intra CU depends on availability, a few more bytes with tiles (contexts restart)
    _, stream0, recs0, packets0 = encode_pictures(off, srcs, qp, bd, keyint=3, intra_in_p=0)
    if not ipass:
        assert all(a.same(b) for a, b in zip(recs, recs0))
        p_bytes, p_bytes0 = sum(len(p[0]) for p in packets if not p[2]), sum(len(p[0]) for p in packets0 if not p[2])
        assert p_bytes0 <= p_bytes <= 1.08 * p_bytes0 + 64, (p_bytes, p_bytes0)


def test_default_p_tile_grid_is_one_tile_per_1080p_area():
    for w, h, level, want in [(1920, 1080, 120, (1, 1)), (2560, 1440, 150, (1, 1)), (3840, 2160, 150, (2, 2)), (7680, 4320, 180, (4, 4)), (4096, 2160, 153, (2, 2))]:
        assert _lib.p_tile_grid(make_cfg(w, h, level_idc=level)) == want, (w, h)
    sl = make_cfg(7680, 544, 10, level_idc=180, pic_height=4320, slice_count=8, slice_index=0)
    assert _lib.p_tile_grid(sl) == (1, 1)              # a sliced picture's slices already are one host job each
    # A.4.1 bounds every tile once tiles are on, a single column or row too: pictures narrower than 256 or lower than 64 luma samples stay untiled even when
    # cfg.p_tiles asks for tiles (found by tests/fuzz_sessions.py in round 3: 146x126 with p_tiles = 1 came out as 1 x 2 tiles with a 146-sample column)
    for w, h in [(146, 126), (128, 216), (502, 24), (248, 512)]:
        assert _lib.p_tile_grid(make_cfg(w, h, level_idc=150, p_tiles=1)) == (1, 1), (w, h)
    assert _lib.p_tile_grid(make_cfg(256, 128, level_idc=150, p_tiles=1)) == (1, 2)


def coding_order(n):
    """cfg.bframes = 1: display positions of a closed GOP of n pictures in DECODING order with their slice types: I0 P2 b1 P4 b3 ...; the last picture is
    always an anchor (an even GOP ends ... P(n-2) b(n-3) P(n-1))"""
    anchors = list(range(0, n, 2)) + ([n - 1] if n > 1 and (n - 1) % 2 else [])
    out = [(0, 2)]
    for prev, cur in zip(anchors, anchors[1:]):
        out.append((cur, 1))
        if cur - prev == 2:
            out.append((prev + 1, 0))
    return out


def encode_gop_with_b(cfg, srcs, qp, bd, me_range=8):
    """oracle analysis -> product host coder for ONE closed GOP with a B picture between every two anchors; returns (stream in decoding order,
    reconstructions by display position, slice types by display position)"""
    lib = _lib.load()
    buf = (C.c_uint8 * (4 << 20))()
    n = lib.mihevc_write_parameter_sets(C.byref(cfg), buf, len(buf))
    stream = bytes(buf[:n]) if not cfg.aud else b""
    headers = bytes(buf[:n])
    prm_i, prm_p, prm_b = O.default_params(max(0, qp - 3), bd, me_range), O.default_params(qp, bd, me_range), O.default_params(qp + 2, bd, me_range)
    prm_i.tile_cols, prm_i.tile_rows = _lib.tile_grid(cfg)
    for prm in (prm_p, prm_b):
        prm.rdo_zero = 1
    recs, types, cus = {}, {}, {}
    last_anchor = None
    for k, (pos, st) in enumerate(coding_order(len(srcs))):
        src = srcs[pos]
        if st == 2:
            a, prm = O.analyze_intra(src, prm_i), prm_i
        elif st == 1:
            a, prm = O.analyze_inter(src, recs[last_anchor], prm_p), prm_p
        else:
            a, prm = O.analyze_b(src, recs[pos - 1], recs[pos + 1], prm_b), prm_b
        dbk = O.deblock(a.rec, a.cu, bd)
        rec, sao = O.sao(src, dbk, prm) if cfg.sao else (dbk, None)
        recs[pos], types[pos], cus[pos] = rec, st, a.cu
        if st != 0:
            last_anchor = pos
        m = lib.mihevc_encode_picture_host(C.byref(cfg), st, pos, prm.qp, util.ptr(a.cu), util.ptr(a.coef_y), util.ptr(a.coef_u), util.ptr(a.coef_v),
                                           util.ptr(sao) if cfg.sao else None, buf, len(buf))
        assert m > 0, (m, pos, st)
        pkt = bytes(buf[:m])
        if k == 0 and cfg.aud:
            cut = pkt.index(b"\0\0\0\1", 4)
            pkt = pkt[:cut] + headers + pkt[cut:]
        stream += pkt
    return stream, [recs[i] for i in range(len(srcs))], [types[i] for i in range(len(srcs))], [cus[i] for i in range(len(srcs))]


@pytest.mark.parametrize("w,h,qp,bd,n,aud", [(96, 80, 26, 8, 7, 1), (136, 72, 32, 8, 6, 0), (72, 104, 24, 10, 5, 1), (160, 96, 20, 8, 9, 1)])
def test_b_pictures_decode_to_the_oracle_reconstruction(w, h, qp, bd, n, aud):
    """cfg.bframes: I0 P2 b1 P4 b3 ... in decoding order; B slices (TRAIL_N) with one picture per list, merge candidates with the combined bi-predictive
    ones, AMVP with the vector of the other list's picture scaled (x -1), inter_pred_idc, bi-prediction by the default weighted average.  The decoder
    returns pictures in output order: they must equal the oracle's reconstructions by display position, and all three prediction kinds must occur."""
    cfg = make_cfg(w, h, bd, aud=aud, bframes=1)
    srcs = [util.synth_frame(h, w, seed=17, shift=(3 * i, 2 * i), bit_depth=bd) for i in range(n)]
    stream, recs, types, cus = encode_gop_with_b(cfg, srcs, qp, bd)
    assert types[0] == 2 and types[n - 1] != 0 and all(t == 0 for t in types[1:n - 1:2])
    frames, info = O.decode(stream)
    assert len(frames) == n
    for i, (f, r) in enumerate(zip(frames, recs)):
        assert f.same(r), f"display picture {i} (slice type {types[i]}) differs after decode"
    kinds = set()
    for cu, t in zip(cus, types):
        if t == 0:
            fl = cu["flags"][(cu["flags"] & 1) == 1]
            kinds |= {int(x) & 96 for x in np.unique(fl)}
    assert kinds == {0, 32, 96}, kinds          # list 0 only, both lists, list 1 only
    assert info["sps.max_num_reorder"] == 1 and info["sps.max_dec_pic_buffering_minus1"] == 2
