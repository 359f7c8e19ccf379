"""GPU (-m gpu): the gfx950 kernels, called through the C ABI, against the oracle — bit exact.
Also the whole session: bitstream -> oracle decoder == encoder reconstruction == oracle pipeline."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hevc_amd import _lib
    L = _lib.load()
    assert L.mihevc_device_count() >= 1, "no gfx950 device visible: the GPU tests need an MI355X"
    return L


@pytest.fixture(scope="module")
def api(lib):
    return util.StageApi(lib, "mihevc_k_", device=0)


def lib_params(lib, qp, bd, rng):
    from hevc_amd import _lib
    p = _lib.cost_params(qp, bd, rng)
    return O.Params(p.qp, p.qp_c, p.bit_depth, p.lambda_sad_q4, p.lambda_q4, p.me_range), p


@pytest.mark.parametrize("log2n,dst", [(2, 0), (2, 1), (3, 0), (4, 0), (5, 0)])
@pytest.mark.parametrize("qp,bd,intra", [(22, 8, 1), (37, 8, 0), (27, 10, 1)])
def test_k3_transform_quant_roundtrip(lib, log2n, dst, qp, bd, intra):
    n = 1 << log2n
    rng = np.random.default_rng(log2n * 100 + qp)
    nb = 70
    amp = (1 << bd) - 1
    res = rng.integers(-amp, amp + 1, (nb, n, n)).astype(np.int16)
    res[0] = 0
    res[1] = 7                       # DC only
    res[2] = rng.integers(-3, 4, (n, n))
    lv = np.zeros_like(res)
    rc = np.zeros_like(res)
    assert lib.mihevc_k_transform(0, util.ptr(res), util.ptr(lv), util.ptr(rc), nb, log2n, qp, bd, intra, dst) == 0
    for b in range(nb):
        coef = O.fwd_transform(res[b], dst=bool(dst), bit_depth=bd)
        want_l = O.quant(coef, qp, bit_depth=bd, intra=bool(intra))
        want_r = O.inv_transform(O.dequant(want_l, qp, bit_depth=bd), dst=bool(dst), bit_depth=bd) if want_l.any() else np.zeros_like(want_l)
        assert np.array_equal(lv[b], want_l), (b, log2n)
        assert np.array_equal(rc[b], want_r), (b, log2n)
    assert not lv[0].any() and not rc[0].any()


CASES = [
    (64, 64, 30, 8, 8),
    (96, 80, 22, 8, 8),
    (136, 72, 35, 8, 16),
    (72, 104, 26, 10, 8),
    (160, 96, 14, 8, 12),
    (320, 192, 24, 8, 32),
]


@pytest.mark.parametrize("w,h,qp,bd,rng", CASES)
def test_stage_parity_i_p_p(lib, api, w, h, qp, bd, rng):
    """K2+K3 (intra), K1+K3 (inter), K4a (deblock), K4b (SAO) each against the oracle on the same inputs."""
    prm_i, cp_i = lib_params(lib, max(0, qp - 3), bd, rng)
    prm_p, cp_p = lib_params(lib, qp, bd, rng)
    prm_p.rdo_zero = cp_p.rdo_zero = int(qp >= 24)       # RD zero-out of inter TUs on for the higher QPs, off for the rest
    prm_p.rdo_cg = cp_p.rdo_cg = 5 if 22 <= qp <= 30 else 0   # RD zero-out of 4x4 coefficient groups at the session's default strength / off
    prm_i.chroma_modes = cp_i.chroma_modes = int(qp < 35)  # chroma intra mode decision
    srcs = [util.synth_frame(h, w, seed=3, shift=(2 * i, i), bit_depth=bd) for i in range(3)]
    want = util.run_pipeline(O, srcs, prm_i, prm_p, bd)
    ref = None
    for i, (src, (a, d, f, sp)) in enumerate(zip(srcs, want)):
        cp = cp_i if i == 0 else cp_p
        got = api.intra(src, cp) if i == 0 else api.inter(src, ref, cp)
        if i:
            assert np.array_equal(a.me, got.me), f"picture {i}: integer search differs"
        assert util.same_analysis(a, got), f"picture {i}: " + util.describe_diff(a, got)
        assert api.deblock(a.rec, a.cu, bd).same(d), f"deblock picture {i}"
        gf, gsp = api.sao(src, d, cp)
        assert np.array_equal(gsp, sp), f"sao params picture {i}"
        assert gf.same(f), f"sao picture {i}"
        lf, lsp = api.loop_filter(src, a.rec, a.cu, cp)       # what a session runs: both filters in one CTU program over the pre-deblock picture
        assert np.array_equal(lsp, sp) and lf.same(f), f"fused loop filter picture {i}"
        ref = f


@pytest.mark.parametrize("w,h,grid,qp,bd", [(544, 160, (2, 2), 26, 8), (800, 224, (3, 3), 33, 8), (512, 192, (2, 3), 22, 10), (1920, 1080, (5, 5), 24, 8)])
def test_intra_tile_grid_parity(lib, api, w, h, grid, qp, bd):
    """K2 with the IDR tile grid (every tile its own CTU wavefront) against the oracle, up to the full 1080p 5x5 grid."""
    prm, cp = lib_params(lib, qp, bd, 8)
    prm.tile_cols, prm.tile_rows = grid
    cp.tile_cols, cp.tile_rows = grid
    prm.chroma_modes = cp.chroma_modes = 1
    src = util.synth_frame(h, w, seed=9, bit_depth=bd)
    want, got = O.analyze_intra(src, prm), api.intra(src, cp)
    assert util.same_analysis(want, got), util.describe_diff(want, got)


@pytest.mark.parametrize("w,h,grid,qp,bd", [(64, 64, (1, 1), 22, 8), (136, 72, (1, 1), 30, 8), (72, 104, (1, 1), 18, 10), (544, 160, (2, 2), 26, 8),
                                            (1920, 1080, (5, 5), 27, 8)])
def test_intra_nxn_parity(lib, api, w, h, grid, qp, bd):
    """K2/K3 with part_mode NxN: four 4x4 PUs per 8x8 CU, DST-VII 4x4 luma TUs, against the oracle."""
    prm, cp = lib_params(lib, qp, bd, 8)
    prm.tile_cols, prm.tile_rows = grid
    cp.tile_cols, cp.tile_rows = grid
    prm.intra_nxn = cp.intra_nxn = 1
    src = util.synth_frame(h, w, seed=31, bit_depth=bd)
    want, got = O.analyze_intra(src, prm), api.intra(src, cp)
    assert (want.cu["flags"] & 16).any()
    assert util.same_analysis(want, got), util.describe_diff(want, got)


@pytest.mark.parametrize("w,h,qp,bd,nxn", [(160, 128, 28, 8, 0), (136, 104, 24, 8, 1), (128, 96, 30, 10, 0), (640, 352, 26, 8, 0)])
def test_intra_second_pass_of_p_pictures_parity(lib, api, w, h, qp, bd, nxn):
    """K2 in P pictures: the inter pass hands per-CTU costs to k_intra_p, which re-codes badly predicted CTUs as intra in two
    independent-set rounds; decisions, records, levels, reconstruction and rate estimate equal the oracle."""
    from tests.test_bitstream_cpu import occluded_clip
    prm, cp = lib_params(lib, qp, bd, 8)
    prm.intra_in_p = cp.intra_in_p = 1
    prm.intra_nxn = cp.intra_nxn = nxn
    srcs = occluded_clip(w, h, bd)
    a0 = O.analyze_intra(srcs[0], prm)
    ref = O.sao(srcs[0], O.deblock(a0.rec, a0.cu, bd), prm)[0]
    for i in (1, 2):
        want = O.analyze_inter(srcs[i], ref, prm, dump_me=True)
        got = api.inter(srcs[i], ref, cp)
        assert ((want.cu["flags"] & 1) == 0).any(), "the occluded patch must go intra"
        assert util.same_analysis(want, got), f"picture {i}: " + util.describe_diff(want, got)
        ref = O.sao(srcs[i], O.deblock(want.rec, want.cu, bd), prm)[0]


@pytest.mark.parametrize("w,h,bd,shift", [(192, 128, 8, (38, -22)), (136, 104, 10, (-50, 17)), (640, 352, 8, (-56, 56))])
def test_pre_search_parity(lib, api, w, h, bd, shift):
    """k_lowres + k_pre_search: centres from the 1/4-size pictures, identical to the oracle; the integer search then finds the shift."""
    prm, cp = lib_params(lib, 27, bd, 8)
    prm.pre_search = cp.pre_search = 1
    base = util.synth_frame(h + 128, w + 128, 5, bit_depth=bd)          # a pure translation: two crops of one larger picture
    def crop(ox, oy):
        return O.Frame(base.y[64 + oy:64 + oy + h, 64 + ox:64 + ox + w].copy(), base.u[32 + oy // 2:32 + (oy + h) // 2, 32 + ox // 2:32 + (ox + w) // 2].copy(),
                       base.v[32 + oy // 2:32 + (oy + h) // 2, 32 + ox // 2:32 + (ox + w) // 2].copy())
    a, b = crop(0, 0), crop(shift[0], shift[1])                        # b(x) = a(x + shift): every block of b sits at +shift in a
    want = O.analyze_inter(b, a, prm, dump_me=True)
    got = api.inter(b, a, cp)
    assert np.array_equal(want.me, got.me) and util.same_analysis(want, got), util.describe_diff(want, got)
    vals, counts = np.unique(want.cu["mvx"].astype(np.int32) * 4096 + want.cu["mvy"], return_counts=True)
    assert vals[np.argmax(counts)] == 4 * shift[0] * 4096 + 4 * shift[1] and counts.max() > 0.3 * want.cu.size     # the true shift dominates (part of the picture has no counterpart in the reference)


def test_search_centres(lib, api):
    w, h = 96, 64
    prm, cp = lib_params(lib, 26, 8, 8)
    a, b = util.synth_frame(h, w, 5), util.synth_frame(h, w, 5, shift=(11, -6))
    cen = np.zeros((util.n_ctus(w, h), 2), np.int16)
    cen[:, 0], cen[:, 1] = 10, -5
    want = O.analyze_inter(b, a, prm, centers=cen, dump_me=True)
    got = api.inter(b, a, cp, centers=cen)
    assert np.array_equal(want.me, got.me) and util.same_analysis(want, got)


def _encode(cfg, frames, keep=True):
    from hevc_amd.encoder import Encoder
    out = b""
    with Encoder(cfg, device=0, keep_recon=keep) as enc:
        for f in frames:
            y, u, v = util.planes(f, cfg.bit_depth)
            enc.send(y, u, v)
            for data, pts, key in enc.packets():
                out += data
        enc.flush()
        n = 0
        pk = []
        for data, pts, key in enc.packets():
            out += data
            pk.append((len(data), pts, key))
        recs = [enc.recon(i) for i in range(len(frames))] if keep else []
        st = enc.stats()
        return out, recs, st, enc.coded_size()


@pytest.mark.parametrize("w,h,bd,keyint,n,nxn,ipass", [(96, 80, 8, 4, 10, 0, 0), (132, 76, 8, 5, 7, 1, 1), (64, 64, 10, 3, 7, 1, 0), (544, 160, 8, 3, 7, 0, 1),
                                                         (544, 160, 8, 3, 4, 1, 0)])
def test_session_stream_decodes_to_encoder_reconstruction(lib, w, h, bd, keyint, n, nxn, ipass):
    """End to end: session -> Annex-B -> oracle decoder must equal the encoder's own reconstruction AND the oracle
    pipeline run with the same QPs; includes non-multiple-of-8 sizes (conformance window) and a short last GOP."""
    from hevc_amd import _lib
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint = w, h, bd, keyint, 2
    cfg.qp, cfg.me_range, cfg.gops_in_flight, cfg.aud, cfg.intra_nxn, cfg.intra_in_p = 27, 8, 2, 1, nxn, ipass
    cfg.scenecut = 0                         # IDR pictures by keyint alone: on 64x64 the detector's 256 samples see a cut in this clip's fourth step
    if bd == 10:
        cfg.hdr10, cfg.colour_primaries, cfg.transfer, cfg.matrix, cfg.chroma_loc, cfg.repeat_headers, cfg.hrd = 1, 9, 16, 9, 0, 1, 1
        cfg.level_idc = 150                  # level 5: the session uses the deeper (12-slot) symbol ring
    frames = [util.synth_frame(h, w, seed=9, shift=(i, i // 2), bit_depth=bd) for i in range(n)]
    if ipass:          # paste an unrelated patch into every second picture so the intra second pass has work
        for i in range(1, n, 2):
            g = util.synth_frame(h, w, seed=500 + i, bit_depth=bd)
            frames[i].y[32:64, 32:96] = g.y[32:64, 32:96]
            frames[i].u[16:32, 16:48] = g.u[16:32, 16:48]
            frames[i].v[16:32, 16:48] = g.v[16:32, 16:48]
    stream, recs, st, (cw, ch) = _encode(cfg, frames)
    assert st.frames_out == n and cw % 8 == 0 and ch % 8 == 0
    dec, info = O.decode(stream)
    assert len(dec) == n and info["width"] == cw and info["conf_width"] == w and info["conf_height"] == h
    assert info["count.aud"] == n
    # oracle pipeline on the padded source with the session's QPs
    qp_p, qp_i = st.last_qp, max(0, st.last_qp - 3)
    prm_i, _ = lib_params(lib, qp_i, bd, 8)
    prm_p, _ = lib_params(lib, qp_p, bd, 8)
    prm_i.tile_cols, prm_i.tile_rows = _lib.tile_grid(cfg)       # 544x160: IDR pictures carry a 2x2 tile grid (PPS 1)
    prm_i.intra_nxn = prm_p.intra_nxn = cfg.intra_nxn            # NxN trial when the session asks for it
    prm_i.chroma_modes = prm_p.chroma_modes = cfg.chroma_modes   # default 1
    prm_p.intra_in_p, prm_p.pre_search, prm_p.rdo_zero, prm_p.rdo_cg = cfg.intra_in_p, cfg.pre_search, cfg.rdo_zero, cfg.rdo_cg   # session defaults: pre-search and both RD zero-outs on
    assert _lib.tile_grid(cfg) == ((2, 2) if w >= 256 else (1, 1))
    idr = util.idr_positions(n, keyint, cfg.gops_in_flight)       # (132, 76, keyint 5, 7 pictures): GOPs of 4 + 3, not 5 + 2
    ref = prev_pad = None
    for i, f in enumerate(frames):
        pad = O.Frame(np.pad(f.y, ((0, ch - h), (0, cw - w)), mode="edge"), np.pad(f.u, ((0, (ch - h) // 2), (0, (cw - w) // 2)), mode="edge"),
                      np.pad(f.v, ((0, (ch - h) // 2), (0, (cw - w) // 2)), mode="edge"))
        if i in idr:
            a = O.analyze_intra(pad, prm_i)
        else:       # the session's search centres come from the source pictures (this one against the one before it)
            a = O.analyze_inter(pad, ref, prm_p, centers=O.search_centres(pad, prev_pad, bd) if cfg.pre_search else None)
        prev_pad = pad
        prm = prm_i if i in idr else prm_p
        ref, _ = O.sao(pad, O.deblock(a.rec, a.cu, bd), prm)
        enc_rec = O.Frame(*recs[i])
        assert enc_rec.same(ref), f"frame {i}: session reconstruction != oracle pipeline"
        assert dec[i].same(enc_rec), f"frame {i}: decoded picture != encoder reconstruction"
    if bd == 10:
        assert info["sei.mdcv.gx"] == 13250 and info["sei.cll.max_cll"] == 1000 and info["sps.profile_idc"] == 2
        # hrd=1: a buffering period SEI at every IDR, a picture timing SEI in every access unit, AUD leading each of them
        assert info["vui.hrd_present"] == 1 and info["count.sei_bp"] == len(idr) and info["count.sei_pt"] == n
        full = 90000 * cfg.vbv_bufsize_kbits // cfg.vbv_maxrate_kbps
        assert abs(info["sei.bp.initial_delay"] - full * 9 // 10) <= 1 and abs(info["sei.bp.initial_delay"] + info["sei.bp.initial_offset"] - full) <= 1
        assert info["sei.pt.au_cpb_removal_delay_minus1"] == (n - 1 - idr[-1]) - 1 if n - 1 > idr[-1] else True


def test_rate_control_caps_the_gop_bitrate_and_stays_bit_exact(lib):
    """VBV-capped constant quality: per-GOP bits stay near vbv-maxrate, QP never drops below the CRF floor, and the
    stream still decodes to the encoder reconstruction == oracle pipeline replayed with the per-picture QPs."""
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    w, h, n, keyint, bd = 192, 128, 40, 20, 8
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.keyint, cfg.min_keyint, cfg.me_range, cfg.gops_in_flight = w, h, keyint, 2, 8, 2
    cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits = 19, -1, 200, 240
    frames = [util.synth_frame(h, w, seed=21, shift=(2 * i, i), bit_depth=bd) for i in range(n)]
    stream = b""
    with Encoder(cfg, device=0, keep_recon=True) as enc:
        for f in frames:
            enc.send(*util.planes(f, bd))
        enc.flush()
        sizes = []
        for data, pts, key in enc.packets():
            stream += data
            sizes.append(len(data) * 8)
        infos = [enc.frame_info(i) for i in range(n)]
        recs = [O.Frame(*enc.recon(i)) for i in range(n)]
    dec, _ = O.decode(stream)
    assert len(dec) == n and all(d.same(r) for d, r in zip(dec, recs))
    qps = [q for q, _, _ in infos]
    assert [i for i, (_, t, _) in enumerate(infos) if t == 2] == util.idr_positions(n, keyint, cfg.gops_in_flight) == [0, 20]
    assert min(qps[1:keyint]) >= cfg.crf + 2 and max(qps) > cfg.crf + 2          # the cap had to raise QP on this clip
    budget = cfg.vbv_maxrate_kbps * 1000 * keyint / 30.0
    for g in range(n // keyint):
        assert sum(sizes[g * keyint:(g + 1) * keyint]) <= 1.5 * budget, (g, sum(sizes[g * keyint:(g + 1) * keyint]), budget)
    ref = None
    for i, f in enumerate(frames):
        prm, _ = lib_params(lib, qps[i], bd, 8)
        prm.intra_nxn, prm.intra_in_p, prm.pre_search, prm.rdo_zero, prm.rdo_cg = cfg.intra_nxn, cfg.intra_in_p, cfg.pre_search, cfg.rdo_zero, cfg.rdo_cg
        prm.chroma_modes = cfg.chroma_modes
        a = O.analyze_intra(f, prm) if i % keyint == 0 else O.analyze_inter(f, ref, prm, centers=O.search_centres(f, frames[i - 1], bd) if cfg.pre_search else None)
        ref, _ = O.sao(f, O.deblock(a.rec, a.cu, bd), prm)
        assert recs[i].same(ref), f"picture {i} (qp {qps[i]})"


def test_convert_video_end_to_end_on_the_gpu(tmp_path):
    """The boundary itself: convert_video on a .y4m clip -> method MI355X, an hvc1 MP4 whose samples decode (oracle
    decoder) to pictures close to the source; progress and the six-key result dict as in the reference."""
    import threading
    from hevc_amd import mp4, transcoder as T, yuvio
    w, h, n = 160, 96, 12
    clip = yuvio.SyntheticClip("motion", 1, w, h, n)
    src = list(clip.frames())
    inp = tmp_path / "clip.y4m"
    yuvio.write_y4m(inp, src, w, h, 30)
    seen = []
    res = T.convert_video(inp, tmp_path, progress_callback=lambda name, f, t: seen.append((name, f, t)), skip_validator=True)
    assert res == {"file": "clip.y4m", "status": "SUCCESS", "quality": res["quality"], "retries": 0, "method": "MI355X", "hdr": False}
    assert seen and seen[-1][1] == seen[-1][2]
    data = (tmp_path / "clip.mp4").read_bytes()
    top = mp4.parse_boxes(data)
    assert [b[0] for b in top] == ["ftyp", "moov", "mdat"]

    def find(path, start, end):
        for name in path:
            _, start, end = [b for b in mp4.parse_boxes(data, start, end) if b[0] == name][0]
        return start, end
    s0, e0 = find(["moov", "trak", "mdia", "minf", "stbl", "stsd"], 0, len(data))
    entry = mp4.parse_boxes(data, s0 + 8, e0)[0]
    hv = [b for b in mp4.parse_boxes(data, entry[1] + 78, entry[2]) if b[0] == "hvcC"][0]
    rec = data[hv[1]:hv[2]]
    annexb, p = b"", 23
    for _ in range(rec[22]):                       # parameter-set arrays of the hvcC record
        cnt = int.from_bytes(rec[p + 1:p + 3], "big")
        p += 3
        for _ in range(cnt):
            ln = int.from_bytes(rec[p:p + 2], "big")
            annexb += b"\0\0\0\1" + rec[p + 2:p + 2 + ln]
            p += 2 + ln
    q = top[2][1]
    while q < top[2][2]:                           # length-prefixed NAL units of the mdat
        ln = int.from_bytes(data[q:q + 4], "big")
        annexb += b"\0\0\0\1" + data[q + 4:q + 4 + ln]
        q += 4 + ln
    frames, info = O.decode(annexb)
    assert len(frames) == n and info["conf_width"] == w and info["conf_height"] == h
    for f, (y, u, v) in zip(frames, src):
        assert util.psnr(f.y[:h, :w], y) > 24.0      # 160x96 lands in level 1/2: the reference policy caps it near 150 kb/s
    ev = threading.Event()
    ev.set()                                       # cancelled before the first frame
    (tmp_path / "x").mkdir()
    res = T.convert_video(inp, tmp_path / "x", skip_validator=True, stop_event=ev)
    assert res["status"] == "CANCELLED"


@pytest.mark.parametrize("w,h,bd,keyint,lanes,n", [(16, 16, 8, 3, 4, 5), (24, 40, 8, 1, 2, 3), (48, 32, 10, 90, 1, 4), (322, 182, 8, 2, 3, 7), (64, 64, 8, 4, 4, 1)])
def test_session_edge_geometries(lib, w, h, bd, keyint, lanes, n):
    """Smallest picture, sizes off the 8 grid in both directions, all-IDR (keyint 1), a GOP longer than the clip, a single frame:
    the stream decodes to the encoder reconstruction and the conformance window restores the display size."""
    from hevc_amd import _lib
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight = w, h, bd, keyint, 1, lanes
    cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits, cfg.me_range = 22, -1, 300, 360, 8
    cw, ch = (w + 7) & ~7, (h + 7) & ~7
    frames = [util.synth_frame(ch, cw, seed=77, shift=(i, i), bit_depth=bd, detail=min(cw, ch) >= 64) for i in range(n)]
    for f in frames:                       # hand over the display-size part only
        f.y, f.u, f.v = f.y[:h, :w].copy(), f.u[:h // 2, :w // 2].copy(), f.v[:h // 2, :w // 2].copy()
    stream, recs, st, (gw, gh) = _encode(cfg, frames)
    assert (gw, gh) == (cw, ch) and st.frames_out == n
    dec, info = O.decode(stream)
    assert len(dec) == n and (info["conf_width"], info["conf_height"]) == (w, h)
    for i in range(n):
        assert dec[i].same(O.Frame(*recs[i])), f"frame {i}"


def test_flush_without_frames_and_reuse_after_flush(lib):
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.keyint = 64, 64, 4
    with Encoder(cfg, device=0) as enc:
        enc.flush()
        assert list(enc.packets()) == []
        assert enc.stats().frames_out == 0


@pytest.mark.parametrize("w,h,keyint,lanes,n", [(96, 80, 4, 2, 19), (256, 64, 5, 3, 47)])
def test_several_chunks_with_rate_control(lib, w, h, keyint, lanes, n):
    """More pictures than one lock-step chunk (lanes x keyint): the session runs chunk after chunk, the rate controller carries its
    learned ratios across them, buffers are reused; every picture still decodes to the encoder's reconstruction."""
    from hevc_amd import _lib
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight, cfg.me_range = w, h, keyint, 2, lanes, 8
    cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits = 20, -1, 150, 180
    frames = [util.synth_frame(h, w, seed=13, shift=(2 * i, i), bit_depth=8) for i in range(n)]
    stream, recs, st, _ = _encode(cfg, frames)
    assert st.frames_out == n
    dec, info = O.decode(stream)
    assert len(dec) == n
    for i in range(n):
        assert dec[i].same(O.Frame(*recs[i])), f"frame {i}"
    assert info["pps.tile_cols"] in (None, 1) or w >= 256


def test_one_clip_sharded_by_gop_chunks_over_two_sessions(lib):
    """SURVEY §8e finer unit: chunks of gops_in_flight x keyint pictures round-robin over devices, no exchange.  With a fixed QP the
    merged stream is the single-session stream picture for picture (here both sessions share GPU 0; on a node they are different GPUs)."""
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder, ShardedEncoder
    w, h, n, bd = 96, 80, 37, 8
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight, cfg.me_range, cfg.qp = w, h, 3, 2, 2, 8, 28      # chunk = 6 pictures
    frames = [util.planes(util.synth_frame(h, w, seed=17, shift=(i, i // 2), bit_depth=bd), bd) for i in range(n)]
    with Encoder(cfg, device=0) as enc:
        single = {}
        for i, (y, u, v) in enumerate(frames):
            enc.send(y, u, v, pts=i)
        enc.flush()
        for data, pts, key in enc.packets():
            single[pts] = (data, key)
        headers = enc.headers()
    sh = ShardedEncoder(cfg, [0, 0])
    try:
        got = []
        for y, u, v in frames:
            sh.send(y, u, v)
            got += sh.ready()
        got += sh.finish()
    finally:
        sh.close()
    assert [p for _, p, _ in got] == list(range(n))                        # presentation order, nothing missing
    for data, pts, key in got:
        want, wkey = single[pts]
        assert key == wkey
        assert data == want or (key and data.endswith(want[-64:]))        # a session's first IDR carries the parameter sets in-band
    stream = b"".join(d for d, _, _ in got)
    dec, _ = O.decode(stream if stream.startswith(headers[:8]) else headers + stream)
    assert len(dec) == n


def test_random_sessions_decode_to_their_reconstruction(lib):
    """24 random geometry / GOP / knob / rate-control combinations (tests/fuzz_sessions.py runs the same generator for longer)"""
    from tests import fuzz_sessions
    assert fuzz_sessions.run(24, seed=3, verbose=False, large=False) == []


def test_random_sessions_on_large_pictures_with_nxn(lib, monkeypatch):
    """Pictures up to 2160p, several lanes and IDR QP variants per launch, the NxN trial forced on: the load under which a race on the
    trial's condition (waves skipping its barriers) used to corrupt streams; small pictures never showed it."""
    from tests import fuzz_sessions
    monkeypatch.setenv("FUZZ_SET", "intra_nxn=1")
    assert fuzz_sessions.run(8, seed=5, verbose=False, large=True) == []


def test_concurrent_sessions_from_several_threads(lib):
    """The reference runs one convert_video per QThread (gui/mainwindow.py:289-301): sessions opened, fed and closed from four threads at
    once on one device must not disturb each other (shared buffer cache, per-session streams, host pools)."""
    import threading
    import numpy as np
    from tests import fuzz_sessions
    fuzz_sessions.LARGE = False
    failed, lock = [], threading.Lock()

    def worker(seed):
        rng = np.random.default_rng(seed)
        for it in range(6):
            desc, ok = fuzz_sessions.one_case(rng, it)
            if not ok:
                with lock:
                    failed.append(f"thread {seed}: {desc}")

    threads = [threading.Thread(target=worker, args=(100 + i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert failed == []


@pytest.mark.parametrize("w,h,qp,bd,rng,pre", [(96, 80, 26, 8, 8, 0), (136, 72, 32, 8, 12, 1), (200, 104, 24, 10, 15, 1), (544, 320, 22, 8, 15, 1)])
def test_b_picture_stage_equals_oracle(api, w, h, qp, bd, rng, pre):
    """cfg.bframes on the device (mihevc_k_b_frame: k_me_search against both anchors + k_inter_ctu_b) against orc_analyze_b_frame, bit for bit: both
    integer-search dumps, records incl. which lists and the list-1 vector, levels, reconstruction, estimate; then the deblocking kernel with the
    two-list boundary strength."""
    from hevc_amd import _lib
    cp_i, cp_p, cp_b = _lib.cost_params(max(0, qp - 3), bd, rng), _lib.cost_params(qp, bd, rng), _lib.cost_params(qp + 2, bd, rng)
    cp_p.rdo_zero = cp_b.rdo_zero = 1
    to_prm = lambda cp: O.Params(cp.qp, cp.qp_c, cp.bit_depth, cp.lambda_sad_q4, cp.lambda_q4, cp.me_range, 1, 1, 0, 0, 0, cp.rdo_zero, 0)      # noqa: E731
    f = [util.synth_frame(h, w, seed=23, shift=(3 * i, 2 * i), bit_depth=bd) for i in range(3)]
    a0 = O.analyze_intra(f[0], to_prm(cp_i))
    r0, _ = O.sao(f[0], O.deblock(a0.rec, a0.cu, bd), to_prm(cp_i))
    a2 = O.analyze_inter(f[2], r0, to_prm(cp_p))
    r2, _ = O.sao(f[2], O.deblock(a2.rec, a2.cu, bd), to_prm(cp_p))
    c0 = O.search_centres(f[1], f[0], bd) if pre else None
    c1 = O.search_centres(f[1], f[2], bd) if pre else None
    want = O.analyze_b(f[1], r0, r2, to_prm(cp_b), c0, c1, dump_me=True)
    got = api.b(f[1], r0, r2, cp_b, c0, c1)
    assert np.array_equal(want.me[0], got.me[0]) and np.array_equal(want.me[1], got.me[1]), "integer searches differ"
    assert util.same_analysis(want, got), util.describe_diff(want, got)
    assert len({int(x) & 96 for x in np.unique(want.cu["flags"])}) >= 2
    assert api.deblock(want.rec, want.cu, bd).same(O.deblock(want.rec, want.cu, bd))


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("kind", ["flat_extremes", "checkerboard", "noise_extremes"])
def test_fractional_search_at_the_limits_of_the_16_bit_packed_transform(lib, api, kind, bd):
    """the 8-bit fractional search's Hadamard stages on 16-bit pairs (v_pk_add / sub / max_i16) with differences of +-255 everywhere: the device's packed arithmetic
    against the oracle's 32-bit one (the CPU twin of this test steps the kernel source with emulated packed operations)"""
    w, h = 96, 64
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    top, dt = (1 << bd) - 1, np.uint8 if bd == 8 else np.uint16        # (10 bit: +-1023, the transform's 16-bit lanes reach 1023 x 32 = 32736 of 32767)
    if kind == "flat_extremes":
        a, b = np.zeros((h, w), dt), np.full((h, w), top, dt)
    elif kind == "checkerboard":
        a = (((xx + yy) & 1) * top).astype(dt)
        b = (top - a).astype(dt)
    else:
        a = (rng.integers(0, 2, (h, w)) * top).astype(dt)
        b = (rng.integers(0, 2, (h, w)) * top).astype(dt)

    def frame(y):
        c = np.ascontiguousarray(y[::2, ::2])
        return O.Frame(y.copy(), c.copy(), (top - c).astype(dt))
    prm, cp = lib_params(lib, 30, bd, 8)
    prm.rdo_zero = cp.rdo_zero = 1
    for src, ref in ((frame(a), frame(b)), (frame(b), frame(a))):
        want, got = O.analyze_inter(src, ref, prm, dump_me=True), api.inter(src, ref, cp)
        assert np.array_equal(want.me, got.me)
        assert util.same_analysis(want, got), util.describe_diff(want, got)


def test_idr_dataflow_launch_is_opt_in_and_equals_the_diagonal_chain(lib, api, monkeypatch):
    """MIHEVC_INTRA_FLOW: stage B of an I picture as ONE launch in which CTUs wait for their neighbours (k_intra_flow) instead of one launch per anti-diagonal.  Off by
    default (several such kernels at once on one device can starve each other: csrc/session.cpp ensure_flow); with the switch on, one picture through the stage entry
    and a session on its own must give what the default gives — the oracle's picture, the same stream."""
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    w, h, bd, qp = 544, 320, 8, 27
    prm, cp = lib_params(lib, qp, bd, 8)
    prm.tile_cols = cp.tile_cols = 2
    prm.tile_rows = cp.tile_rows = 2
    src = util.synth_frame(h, w, seed=17, bit_depth=bd)
    want = O.analyze_intra(src, prm)
    streams = []
    for on in (False, True):
        if on:
            monkeypatch.setenv("MIHEVC_INTRA_FLOW", "1")
        else:
            monkeypatch.delenv("MIHEVC_INTRA_FLOW", raising=False)
        got = api.intra(src, cp)
        assert util.same_analysis(want, got), ("flow" if on else "chain") + ": " + util.describe_diff(want, got)
        cfg = _lib.default_config()
        cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight, cfg.qp, cfg.level_idc = w, h, bd, 4, 2, 2, qp, 120
        frames = [util.synth_frame(h, w, seed=17, shift=(2 * i, i), bit_depth=bd) for i in range(9)]
        with Encoder(cfg, device=0) as enc:
            for f in frames:
                enc.send(*util.planes(f, bd))
            enc.flush()
            streams.append(b"".join(p[0] for p in enc.packets()))
    assert streams[0] == streams[1] and len(streams[0]) > 1000
