"""GPU (-m gpu): the BASELINE.json configurations themselves, deterministically.

config 2 — 1920x1080 8-bit Main at the reference's libx265 operating point (crf 19, vbv-maxrate 2940, vbv-bufsize 3528, keyint 90:
/root/reference core/transcoder.py:398-411), and config 3 — 3840x2160 Main10 HDR10 (level 5, 10x11 IDR tile grid, 12-slot symbol ring,
the HDR10 set of core/utils.py:58-69).  Each: per-stage HIP vs oracle on I+P(+P) of the bench clip (`SyntheticClip("motion")`) including
the integer-search dump, then a session at the operating point whose stream must decode (oracle decoder) to the encoder's reconstruction.
The fuzz tests draw their sizes at random and never reached these two geometries."""
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


def operating_point(w, h, hdr, n):
    """mihevc_config exactly as encode_file / bench.py derive it from the reference's policy functions"""
    from hevc_amd.encoder import config_for
    from hevc_amd.probe import VideoInfo
    from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values
    if hdr:
        info = VideoInfo(w, h, 30.0, "bt2020", "smpte2084", "bt2020nc", "yuv420p10le", "", "", 0, True, "eng", n, n / 30.0)
    else:
        info = VideoInfo(w, h, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", n, n / 30.0)
    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info, use_nvenc=False)
    level, tier = calculate_apple_hevc_level(info)
    return config_for(info, crf, maxrate, bufsize, gop, level, tier), (crf, maxrate, bufsize, gop, level)


def clip_frames(w, h, bd, n, seed=0):
    """first n pictures of the bench clip, padded to the coded size by edge replication (what the session's ingest does)"""
    from hevc_amd.yuvio import SyntheticClip
    cw, ch = (w + 7) & ~7, (h + 7) & ~7
    out = []
    for y, u, v in SyntheticClip("motion", seed, w, h, n, bit_depth=bd).frames():
        out.append(((y, u, v), O.Frame(np.pad(y, ((0, ch - h), (0, cw - w)), mode="edge"), np.pad(u, ((0, (ch - h) // 2), (0, (cw - w) // 2)), mode="edge"),
                                       np.pad(v, ((0, (ch - h) // 2), (0, (cw - w) // 2)), mode="edge"))))
    return out


def session_params(lib, cfg, qp, idr):
    """oracle + device cost parameters of one picture of a session opened with cfg (the knobs the session passes to its kernels)"""
    from hevc_amd import _lib
    cp = _lib.cost_params(qp, cfg.bit_depth, cfg.me_range if cfg.me_range > 0 else 15)
    cp.tile_cols, cp.tile_rows = _lib.tile_grid(cfg) if idr else _lib.p_tile_grid(cfg)
    cp.intra_nxn, cp.intra_in_p, cp.pre_search, cp.rdo_zero, cp.chroma_modes = cfg.intra_nxn, cfg.intra_in_p, cfg.pre_search, cfg.rdo_zero, cfg.chroma_modes
    prm = O.Params(cp.qp, cp.qp_c, cp.bit_depth, cp.lambda_sad_q4, cp.lambda_q4, cp.me_range, cp.tile_cols, cp.tile_rows, cp.intra_nxn, cp.intra_in_p,
                   cp.pre_search, cp.rdo_zero, cp.chroma_modes)
    prm.rdo_cg = cp.rdo_cg = max(0, cfg.rdo_cg)
    return prm, cp


def stage_parity(lib, api, cfg, frames, qp_p):
    bd = cfg.bit_depth
    ref = None
    for i, (_, src) in enumerate(frames):
        prm, cp = session_params(lib, cfg, max(0, qp_p - 3) if i == 0 else qp_p, i == 0)
        want = O.analyze_intra(src, prm) if i == 0 else O.analyze_inter(src, ref, prm, dump_me=True)
        got = api.intra(src, cp) if i == 0 else api.inter(src, ref, cp)
        if i:
            assert np.array_equal(want.me, got.me), f"picture {i}: integer search differs"
            assert (want.cu["flags"] & 1).all() or cfg.intra_in_p
        assert util.same_analysis(want, got), f"picture {i}: " + util.describe_diff(want, got)
        d = O.deblock(want.rec, want.cu, bd)
        assert api.deblock(want.rec, want.cu, bd).same(d), f"deblock picture {i}"
        f, sp = O.sao(src, d, prm)
        gf, gsp = api.sao(src, d, cp)
        assert np.array_equal(gsp, sp), f"sao params picture {i}"
        assert gf.same(f), f"sao picture {i}"
        lf, lsp = api.loop_filter(src, want.rec, want.cu, cp)
        assert np.array_equal(lsp, sp) and lf.same(f), f"fused loop filter picture {i}"
        ref = f


def run_session(cfg, frames):
    from hevc_amd.encoder import Encoder
    stream, sizes = b"", []
    with Encoder(cfg, device=0, keep_recon=True) as enc:
        for (y, u, v), _ in frames:
            enc.send(y, u, v)
        enc.flush()
        for data, pts, key in enc.packets():
            stream += data
            sizes.append(len(data) * 8)
        infos = [enc.frame_info(i) for i in range(len(frames))]
        recs = [O.Frame(*enc.recon(i)) for i in range(len(frames))]
        st = enc.stats()
    return stream, sizes, infos, recs, st


def replay(lib, cfg, frames, infos, recs, upto):
    """the oracle pipeline with the session's per-picture QPs must give the session's reconstruction"""
    ref = None
    for i in range(upto):
        qp, st, _ = infos[i]
        prm, _ = session_params(lib, cfg, qp, st == 2)
        src = frames[i][1]
        a = O.analyze_intra(src, prm) if st == 2 else O.analyze_inter(src, ref, prm, centers=O.search_centres(src, frames[i - 1][1], cfg.bit_depth) if cfg.pre_search else None)
        ref, _ = O.sao(src, O.deblock(a.rec, a.cu, cfg.bit_depth), prm)
        assert recs[i].same(ref), f"picture {i} (qp {qp}): session reconstruction != oracle pipeline"


def test_config2_1080p_stage_parity_i_p_p(lib, api):
    cfg, (crf, maxrate, bufsize, gop, level) = operating_point(1920, 1080, False, 300)
    assert (crf, maxrate, bufsize, gop, level) == (19, 2940, 3528, 90, "4")          # SURVEY App. A golden
    from hevc_amd import _lib
    assert _lib.tile_grid(cfg) == (5, 5)
    stage_parity(lib, api, cfg, clip_frames(1920, 1080, 8, 3), crf + 2)


def test_config2_1080p_session_at_the_reference_operating_point(lib):
    n = 12
    cfg, (crf, maxrate, bufsize, gop, _) = operating_point(1920, 1080, False, 300)
    frames = clip_frames(1920, 1080, 8, n)
    stream, sizes, infos, recs, st = run_session(cfg, frames)
    assert st.frames_out == n
    dec, info = O.decode(stream)
    assert len(dec) == n and (info["width"], info["height"], info["conf_width"], info["conf_height"]) == (1920, 1080, 1920, 1080)
    assert info["sps.profile_idc"] == 1 and info["sps.level_idc"] == 120 and (info["pps.tile_cols"], info["pps.tile_rows"]) == (5, 5)
    for i in range(n):
        assert dec[i].same(recs[i]), f"frame {i}: decoded picture != encoder reconstruction"
    assert [t for _, t, _ in infos] == [2] + [1] * (n - 1)
    assert all(q >= crf + 2 for q, _, _ in infos[1:]) and infos[0][0] >= crf - 1        # CRF is the quality ceiling; the VBV cap only raises QP
    assert [b for _, _, b in infos] == sizes
    replay(lib, cfg, frames, infos, recs, 3)
    y = np.stack([f[0][0] for f in frames]).astype(np.float64)
    r = np.stack([x.y[:1080] for x in recs]).astype(np.float64)
    assert 10 * np.log10(255.0 ** 2 / np.mean((y - r) ** 2)) > 36.0


def test_config3_2160p_main10_hdr10_stage_parity_i_p(lib, api):
    cfg, (crf, maxrate, bufsize, gop, level) = operating_point(3840, 2160, True, 300)
    assert (crf, maxrate, bufsize, gop, level) == (19, 11760, 14112, 60, "5") and cfg.bit_depth == 10 and cfg.hrd == 1
    from hevc_amd import _lib
    assert _lib.tile_grid(cfg) == (10, 11)
    stage_parity(lib, api, cfg, clip_frames(3840, 2160, 10, 2), crf + 2)


def test_config3_2160p_main10_hdr10_session(lib):
    n = 6
    cfg, (crf, maxrate, bufsize, gop, _) = operating_point(3840, 2160, True, 300)
    frames = clip_frames(3840, 2160, 10, n)
    stream, sizes, infos, recs, st = run_session(cfg, frames)
    assert st.frames_out == n
    dec, info = O.decode(stream)
    assert len(dec) == n and (info["width"], info["height"], info["bit_depth"]) == (3840, 2160, 10)
    for i in range(n):
        assert dec[i].same(recs[i]), f"frame {i}: decoded picture != encoder reconstruction"
    # Main10, level 5, 10x11 IDR tiles, HDR10 signalling (core/utils.py:58-69 defaults), HRD + buffering period / picture timing SEI
    assert info["sps.profile_idc"] == 2 and info["sps.level_idc"] == 150 and (info["pps.tile_cols"], info["pps.tile_rows"]) == (10, 11)
    assert (info["vui.colour_primaries"], info["vui.transfer"], info["vui.matrix"], info["vui.full_range"]) == (9, 16, 9, 0)
    assert (info["sei.mdcv.gx"], info["sei.mdcv.gy"], info["sei.mdcv.bx"], info["sei.mdcv.by"], info["sei.mdcv.rx"], info["sei.mdcv.ry"]) == (13250, 34500, 7500, 3000, 34000, 16000)
    assert (info["sei.mdcv.wpx"], info["sei.mdcv.wpy"], info["sei.mdcv.max_lum"], info["sei.mdcv.min_lum"]) == (15635, 16450, 10000000, 50)
    assert (info["sei.cll.max_cll"], info["sei.cll.max_fall"]) == (1000, 400)
    assert info["count.aud"] == n and info["vui.hrd_present"] == 1 and info["count.sei_bp"] == 1 and info["count.sei_pt"] == n
    rate = (info["hrd.bit_rate_value_minus1"] + 1) << (6 + info["hrd.bit_rate_scale"])
    cpb = (info["hrd.cpb_size_value_minus1"] + 1) << (4 + info["hrd.cpb_size_scale"])
    assert abs(rate - maxrate * 1000) <= maxrate * 10 and abs(cpb - bufsize * 1000) <= bufsize * 10
    assert all(q >= crf + 2 for q, _, _ in infos[1:])
    replay(lib, cfg, frames, infos, recs, 2)


def hrd_arrival_schedule(sizes_bits, keys, bit_rate, init_delays_90k, fps):
    """H.265 C.2.2 (cbr_flag = 0) + C.2.3 for a stream without reordering: picture n is removed at n / fps (the picture timing SEI counts
    clock ticks since the last buffering period), its first bit may enter the CPB no earlier than its removal time minus the initial delay
    announced by the LAST buffering period, and never before the previous picture has fully arrived.  Returns the smallest slack
    (removal time - final arrival time) in seconds: negative = the CPB underflows."""
    t_af, slack, delay = 0.0, 1e9, init_delays_90k[0] / 90000.0
    k = 0
    for n, (bits, key) in enumerate(zip(sizes_bits, keys)):
        if key:
            delay = init_delays_90k[min(k, len(init_delays_90k) - 1)] / 90000.0
            k += 1
        t_r = init_delays_90k[0] / 90000.0 + n / fps
        t_ai = max(t_af, t_r - delay)
        t_af = t_ai + bits / bit_rate
        slack = min(slack, t_r - t_af)
    return slack


@pytest.mark.parametrize("w,h,bd,keyint,n,maxrate,bufsize", [(640, 352, 8, 30, 150, 600, 720), (416, 240, 10, 20, 130, 400, 480)])
def test_cpb_schedule_of_the_produced_sizes_never_underflows(lib, w, h, bd, keyint, n, maxrate, bufsize):
    """nal-hrd=vbr:vbv-maxrate:vbv-bufsize (reference core/transcoder.py:399-400): the sizes the session produces, fed through the
    Annex C arrival / removal schedule that its own VUI HRD parameters + buffering period / picture timing SEI describe, never
    underflow the CPB; the average rate stays under vbv-maxrate; the QP never drops below the CRF's."""
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    from hevc_amd.yuvio import SyntheticClip
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint, cfg.me_range = w, h, bd, keyint, 2, 8
    cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits, cfg.hrd, cfg.aud, cfg.level_idc = 19, -1, maxrate, bufsize, 1, 1, 93
    sizes, keys, stream = [], [], b""
    with Encoder(cfg, device=0) as enc:
        for y, u, v in SyntheticClip("motion", 4, w, h, n, bit_depth=bd).frames():
            enc.send(y, u, v)
            for data, pts, key in enc.packets():
                sizes.append(len(data) * 8); keys.append(key); stream += data
        enc.flush()
        for data, pts, key in enc.packets():
            sizes.append(len(data) * 8); keys.append(key); stream += data
        qps = [enc.frame_info(i)[0] for i in range(n)]
    idr = [i for i, k in enumerate(keys) if k]
    # 4 GOPs per chunk; the last chunk's GOPs are equalised (130 pictures at keyint 20: 4 x 20, then 17 + 17 + 16); this clip has no scene cut
    assert len(sizes) == n and idr == util.idr_positions(n, keyint)
    dec, info = O.decode(stream[:sum(sizes[:idr[1] + 2]) // 8])          # two buffering periods' worth of SEI parsed back
    assert info["vui.hrd_present"] == 1 and info["count.sei_bp"] == 2
    rate = (info["hrd.bit_rate_value_minus1"] + 1) << (6 + info["hrd.bit_rate_scale"])
    cpb = (info["hrd.cpb_size_value_minus1"] + 1) << (4 + info["hrd.cpb_size_scale"])
    assert abs(rate - maxrate * 1000) <= 64 and abs(cpb - bufsize * 1000) <= 16
    d0 = info["sei.bp.initial_delay"]
    assert abs(d0 - 90000 * 0.9 * cpb / rate) <= 2
    slack = hrd_arrival_schedule(sizes, keys, rate, [d0], 30.0)
    assert slack >= 0.0, f"CPB underflow: a picture finishes arriving {-slack * 1e3:.1f} ms after its removal time"
    assert sum(sizes) / (n / 30.0) <= maxrate * 1000, (sum(sizes) / (n / 30.0), maxrate * 1000)
    assert max(sizes) <= 0.9 * cpb
    assert min(qps[1:]) >= cfg.crf + 2 and min(qps) >= cfg.crf - 1 and max(qps) > cfg.crf + 2       # the cap had to bind on this clip


@pytest.mark.parametrize("pattern,w,h,n", [("bars", 1280, 720, 12), ("flat", 320, 192, 9), ("bars", 64, 64, 5)])
def test_config1_content_and_degenerate_pictures(lib, pattern, w, h, n):
    """BASELINE configs[0] names the reference's own test clip: 720p30 lavfi testsrc-like colour bars (tests/generate_test_videos.py:26-31) through the
    libx265 CPU path — ffmpeg does not exist on this pool, so the same KIND of content (static bars + sweeping gradient + counter block, `SyntheticClip
    ("bars")`) goes through the native path instead.  Plus pictures with nothing in them (one grey level: every residual quantises to zero, every CU
    skips): the stream must still decode to the encoder's reconstruction and the rate controller must not divide by an empty estimate."""
    from hevc_amd.yuvio import SyntheticClip
    cfg, _ = operating_point(w, h, False, n)
    cfg.keyint, cfg.min_keyint, cfg.gops_in_flight = 6, 2, 2
    if pattern == "flat":
        frames = [((np.full((h, w), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8)), None)] * n
    else:
        frames = [(f, None) for f in SyntheticClip("bars", 0, w, h, n).frames()]
    stream, sizes, infos, recs, st = run_session(cfg, frames)
    dec, info = O.decode(stream)
    assert len(dec) == n and st.frames_out == n
    for i in range(n):
        assert dec[i].same(recs[i]), f"frame {i}: decoded picture != encoder reconstruction"
        assert util.psnr(recs[i].y[:h, :w], frames[i][0][0]) > (50.0 if pattern == "flat" else 30.0)
    if pattern == "flat":
        assert max(sizes[1:]) < 4000                      # all-skip P pictures: a few hundred bits


def test_stress_clip_zoom_fades_flash_and_cut(lib):
    """Content a translational search and a per-GOP rate plan do not like (`SyntheticClip("stress")`: zoom, fade to dark and back, twelve occluders, a
    one-picture white flash at n/4, a hard cut at n/2) at the reference's operating point for its size: the stream still decodes to the encoder's
    reconstruction, the bitrate stays under vbv-maxrate, the CPB schedule its own SEI describe never underflows, and the IDR pictures sit where the
    content changes for good — at the cut, and AFTER the flash (a run of jumps is cut at its last picture), not on it."""
    from hevc_amd.encoder import Encoder
    from hevc_amd.yuvio import SyntheticClip
    w, h, n = 640, 352, 120
    cfg, (crf, maxrate, bufsize, gop, _level) = operating_point(w, h, False, n)
    cfg.hrd, cfg.aud, cfg.min_keyint = 1, 1, 8
    clip = SyntheticClip("stress", 3, w, h, n)
    src = [clip.frame(i) for i in range(n)]
    sizes, keys, stream = [], [], b""
    with Encoder(cfg, device=0, keep_recon=True) as enc:
        for y, u, v in src:
            enc.send(y, u, v)
        enc.flush()
        for data, pts, key in enc.packets():
            sizes.append(len(data) * 8); keys.append(key); stream += data
        recs = [O.Frame(*enc.recon(i)) for i in range(n)]
        qps = [enc.frame_info(i)[0] for i in range(n)]
    idr = [i for i, k in enumerate(keys) if k]
    assert idr[0] == 0 and n // 2 in idr and n // 4 + 1 in idr and n // 4 not in idr, idr
    assert all(b - a <= gop for a, b in zip(idr, idr[1:] + [n]))
    dec, info = O.decode(stream)
    assert len(dec) == n and all(d.same(r) for d, r in zip(dec, recs))
    rate = (info["hrd.bit_rate_value_minus1"] + 1) << (6 + info["hrd.bit_rate_scale"])
    assert hrd_arrival_schedule(sizes, keys, rate, [info["sei.bp.initial_delay"]], 30.0) >= 0.0
    # four seconds with two forced IDR pictures on top of the periodic ones: the CPB (checked above) is the hard limit, it starts 0.9 full and lends its
    # content once; the mean rate of so short a clip may sit a little above vbv-maxrate (here 301 of 294 kb/s)
    assert sum(sizes) / (n / 30.0) <= 1.05 * maxrate * 1000, (sum(sizes) / (n / 30.0), maxrate * 1000)
    psnr = [util.psnr(r.y[:h, :w], s[0]) for r, s in zip(recs, src)]
    assert min(psnr[:n // 4] + psnr[n // 4 + 1:]) > 22.0 and np.mean(psnr) > 27.0, (min(psnr), float(np.mean(psnr)))      # 294 kb/s for zooming 640x352: sanity, not a quality claim (measured 25.9 / 29.5 dB)
    assert min(qps) >= crf - 1
