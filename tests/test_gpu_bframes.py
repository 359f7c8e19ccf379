"""GPU (-m gpu): sessions with B pictures (cfg.bframes; x265 bframes, the reference's preset=slow codes them: core/transcoder.py:399).

Every closed GOP is coded I0 P2 b1 P4 b3 ...: the packets come in DECODING order with dts <= pts, the stream decodes (the oracle decoder reorders
by picture order count) to the encoder's reconstructions by display position, and the reconstructions equal the oracle pipeline replayed in coding
order with the session's QPs — anchors from the anchor before them, B pictures from both neighbours (orc_analyze_b_frame)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util
from tests.test_bitstream_cpu import coding_order
from tests.test_gpu_configs import session_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hevc_amd import _lib
    L = _lib.load()
    assert L.mihevc_device_count() >= 1, "no gfx950 device visible: the GPU tests need an MI355X"
    return L


def run_b_session(cfg, frames, bd):
    from hevc_amd.encoder import Encoder
    n = len(frames)
    pk = []
    with Encoder(cfg, device=0, keep_recon=True) as enc:
        for f in frames:
            enc.send(*util.planes(f, bd))
            pk += list(enc.packets_dts())
        enc.flush()
        pk += list(enc.packets_dts())
        infos = [enc.frame_info(i) for i in range(n)]
        recs = [O.Frame(*enc.recon(i)) for i in range(n)]
        st = enc.stats()
    return pk, infos, recs, st


def gops_of(infos):
    idr = [i for i, (_, t, _) in enumerate(infos) if t == 2]
    return [(a, b) for a, b in zip(idr, idr[1:] + [len(infos)])]


def replay(lib, cfg, frames, infos, recs, with_b=True):
    """the oracle pipeline in coding order with the session's per-picture QPs"""
    bd = cfg.bit_depth
    for g0, g1 in gops_of(infos):
        rec, last = {}, None
        for pos, st in (coding_order(g1 - g0) if with_b else [(p, 1 if p else 2) for p in range(g1 - g0)]):
            i = g0 + pos
            assert infos[i][1] == st, (i, infos[i], st)
            prm, _ = session_params(lib, cfg, infos[i][0], st == 2)
            src = frames[i]
            if st == 2:
                a = O.analyze_intra(src, prm)
            elif st == 1:
                a = O.analyze_inter(src, rec[last], prm, centers=O.search_centres(src, frames[g0 + last], bd) if cfg.pre_search else None)
            else:
                a = O.analyze_b(src, rec[pos - 1], rec[pos + 1], prm, O.search_centres(src, frames[i - 1], bd) if cfg.pre_search else None,
                                O.search_centres(src, frames[i + 1], bd) if cfg.pre_search else None)
            rec[pos], _ = O.sao(src, O.deblock(a.rec, a.cu, bd), prm)
            if st != 0:
                last = pos
            assert recs[i].same(rec[pos]), f"display picture {i} (slice type {st}, qp {infos[i][0]}): session reconstruction != oracle pipeline"


@pytest.mark.parametrize("w,h,bd,keyint,n,lanes,rc", [(192, 128, 8, 7, 20, 2, 0), (320, 200, 10, 6, 17, 3, 0), (416, 240, 8, 10, 30, 2, 1)])
def test_sessions_with_b_pictures(lib, w, h, bd, keyint, n, lanes, rc):
    from hevc_amd import _lib
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint, cfg.me_range, cfg.gops_in_flight, cfg.aud, cfg.bframes, cfg.scenecut = w, h, bd, keyint, 2, 12, lanes, 1, 1, 0
    cfg.level_idc = 93
    if rc:
        cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits, cfg.hrd = 20, -1, 500, 600, 1
    else:
        cfg.qp = 28
    frames = [util.synth_frame(h, w, seed=9, shift=(3 * i, i), bit_depth=bd) for i in range(n)]
    pk, infos, recs, st = run_b_session(cfg, frames, bd)
    assert st.frames_out == n and len(pk) == n
    # GOP layout and picture types: B pictures at the odd places between two anchors, never the GOP's last picture
    for g0, g1 in gops_of(infos):
        want = {pos: t for pos, t in coding_order(g1 - g0)}
        assert [infos[g0 + p][1] for p in range(g1 - g0)] == [want[p] for p in range(g1 - g0)]
        # packets of the GOP in decoding order, pts = display place, dts non-decreasing and never after pts
        got = [p for p in pk if g0 <= p[1] < g1]
        assert [p[1] - g0 for p in got] == [pos for pos, _ in coding_order(g1 - g0)]
        assert got[0][2] and not any(p[2] for p in got[1:])
    dts = [p[3] for p in pk]
    assert dts == sorted(dts) and len(set(dts)) == n and all(p[3] <= p[1] for p in pk)
    assert sum(1 for _, t, _ in infos if t == 0) >= n // 3
    if not rc:
        assert {q for q, t, _ in infos if t == 1} == {28} and {q for q, t, _ in infos if t == 0} == {30} and {q for q, t, _ in infos if t == 2} == {25}
    dec, info = O.decode(b"".join(p[0] for p in pk))
    assert len(dec) == n and info["count.aud"] == n and info["sps.max_num_reorder"] == 1
    for i in range(n):
        assert dec[i].same(recs[i]), f"display picture {i}: decoded picture != encoder reconstruction"
    if rc:
        assert info["vui.hrd_present"] == 1 and info["count.sei_pt"] == n
        assert sum(len(p[0]) for p in pk) * 8 / (n / 30.0) <= 1.05 * 500e3
    replay(lib, cfg, frames, infos, recs)
    psnr = np.mean([util.psnr(r.y, f.y, peak=(1 << bd) - 1.0) for r, f in zip(recs, frames)])
    assert psnr > 26.0          # sanity only (the capped case: 500 kb/s for 416x240 of moving texture, measured 29.3 dB)


def test_b_pictures_through_convert_video_and_the_muxer(lib, tmp_path, monkeypatch):
    """encode_file with a configuration that asks for B pictures: the MP4 carries samples in decoding order with a ctts box (composition offsets) and an
    edit list; the elementary stream taken back out of the file decodes to as many pictures as went in"""
    from hevc_amd import encoder, mp4, transcoder as T, yuvio
    w, h, n = 192, 128, 13
    frames = list(yuvio.SyntheticClip("motion", 2, w, h, n).frames())
    p = tmp_path / "b.y4m"
    yuvio.write_y4m(p, frames, w, h, 30)
    real = encoder.config_for

    def with_b(*a, **k):
        cfg = real(*a, **k)
        cfg.bframes = 1
        return cfg
    monkeypatch.setattr(encoder, "config_for", with_b)
    res = T.convert_video(p, tmp_path, skip_validator=True)
    assert res["status"] == "SUCCESS" and res["method"] == "MI355X"
    data = (tmp_path / "b.mp4").read_bytes()

    def find(path, start, end):
        for name in path:
            _, start, end = [b for b in mp4.parse_boxes(data, start, end) if b[0] == name][0]
        return start, end
    t0, t1 = find(["moov", "trak"], 0, len(data))
    assert "edts" in [b[0] for b in mp4.parse_boxes(data, t0, t1)]
    s0, e0 = find(["moov", "trak", "mdia", "minf", "stbl"], 0, len(data))
    kinds = {b[0]: b for b in mp4.parse_boxes(data, s0, e0)}
    assert "ctts" in kinds
    c = data[kinds["ctts"][1]:kinds["ctts"][2]]
    runs = int.from_bytes(c[4:8], "big")
    offs = []
    for k in range(runs):
        cnt, off = int.from_bytes(c[8 + 8 * k:12 + 8 * k], "big"), int.from_bytes(c[12 + 8 * k:16 + 8 * k], "big")
        offs += [off] * cnt
    assert len(offs) == n and offs[0] == 1 and set(offs) == {0, 1, 2}      # I: one frame, anchors: two, B pictures: none (fps_den = 1)


def probe_positions(n):
    """pictures the session's probe looks at (hevc_amd/csrc/session.cpp probe_bframes)"""
    K = min(4, (n - 2 + 15) // 16 + 1)
    at = []
    for k in range(K):
        p = 2 + (n - 3) * k // max(1, K - 1)
        if not at or at[-1] != p:
            at.append(p)
    return at


def probe_cost(src, ref, qp, bd, me_range):
    """sum over the CTUs of the 32x32 node's best integer-search cost against `ref` (a SOURCE picture, border replicated); CTUs the picture cuts off: their
    16x16 nodes, and the 8x8 nodes of cut-off 16x16 ones"""
    me = O.analyze_inter(src, ref, O.default_params(qp, bd, me_range), dump_me=True).me
    total = 0
    for m in me:
        if m[0, 2] >= 0:
            total += int(m[0, 2])
            continue
        for nd in range(1, 21):
            if (m[nd, 2] >= 0) if nd < 5 else (m[1 + ((nd - 5) >> 2), 2] < 0 and m[nd, 2] >= 0):
                total += int(m[nd, 2])
    return total


@pytest.mark.parametrize("pattern,w,h,n", [("bars", 640, 352, 24), ("motion", 640, 352, 24), ("stress", 640, 352, 40), ("motion", 200, 104, 9)])
def test_adaptive_b_decision_follows_the_probe(lib, pattern, w, h, n):
    """cfg.bframes = -1: the session decides per chunk from the integer search of sample SOURCE pictures against the sources one and two places back (the
    oracle's search is the same arithmetic): the costs it reports equal the oracle's sums, the decision is c2 <= 1.4 c1, the picture types follow it, and
    the stream decodes and replays either way."""
    from hevc_amd import _lib
    from hevc_amd.yuvio import SyntheticClip
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.keyint, cfg.min_keyint, cfg.me_range, cfg.gops_in_flight, cfg.bframes, cfg.scenecut, cfg.qp, cfg.level_idc = w, h, 12, 2, 12, 4, -1, 0, 30, 93
    clip = SyntheticClip(pattern, 1, w, h, n)
    frames = [O.Frame(*clip.frame(i)) for i in range(n)]
    pk, infos, recs, st = run_b_session(cfg, frames, 8)
    at = probe_positions(n)
    c1 = sum(probe_cost(frames[p], frames[p - 1], 30, 8, 12) for p in at)
    c2 = sum(probe_cost(frames[p], frames[p - 2], 30, 8, 12) for p in at)
    per = util.n_ctus(w, h) * len(at)
    print(pattern, "probe", st.reserved[0], st.reserved[1], "oracle", c1 // per, c2 // per, "ratio", round(c2 / max(1, c1), 3))
    assert (st.reserved[0], st.reserved[1]) == (c1 // per, c2 // per)
    want = c2 <= 1.4 * c1
    assert bool(st.reserved[2]) == want
    assert any(t == 0 for _, t, _ in infos) == want
    dec, info = O.decode(b"".join(p[0] for p in pk))
    assert len(dec) == n and all(d.same(r) for d, r in zip(dec, recs))
    replay(lib, cfg, frames, infos, recs, with_b=want)
