"""GPU (-m gpu): the boundary's failure and multi-worker paths on the real device."""
import csv
import threading
from pathlib import Path

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


def _clip(tmp_path, name, w, h, n, seed=1, bit_depth=8):
    from hevc_amd import yuvio
    frames = list(yuvio.SyntheticClip("motion", seed, w, h, n, bit_depth=bit_depth).frames())
    p = tmp_path / name
    yuvio.write_y4m(p, frames, w, h, 30, bit_depth=bit_depth)
    return p, frames


def test_batch_runner_over_three_clips_on_device_0(lib, tmp_path):
    """f2: the headless queue drives real encodes — worker PROCESSES pinned to the device, CSV with method / device columns"""
    from hevc_amd import batch, mp4
    files = [_clip(tmp_path, f"c{i}.y4m", 160, 96, 8, seed=i)[0] for i in range(3)]
    out = tmp_path / "out"
    r = batch.BatchRunner(files, out, max_workers=2, skip_validator=True).start()
    assert r.use_processes                                   # the default with a device present: one process per worker
    res = r.wait()
    assert len(res) == 3 and all(x["status"] == "SUCCESS" and x["method"] == "MI355X" for x in res), res
    rows = list(csv.DictReader(open(out / "transcode_log.csv")))
    assert len(rows) == 3 and list(rows[0])[:6] == ["file", "status", "quality", "retries", "method", "hdr"]
    assert {x["method"] for x in rows} == {"MI355X"} and {x["device"] for x in rows} == {"0"}
    for f in files:
        top = mp4.parse_boxes((out / (f.stem + ".mp4")).read_bytes())
        assert [b[0] for b in top] == ["ftyp", "moov", "mdat"]


def test_ten_bit_sdr_input_is_coded_main10_not_wrapped(lib, tmp_path):
    """a y4m tagged C420p10 that probes as SDR: bit depth follows the sample format (round 1 wrapped the samples modulo 256)"""
    from hevc_amd import mp4, transcoder as T
    w, h, n = 160, 96, 5
    p, frames = _clip(tmp_path, "sdr10.y4m", w, h, n, bit_depth=10)
    res = T.convert_video(p, tmp_path, skip_validator=True)
    assert res["status"] == "SUCCESS" and res["method"] == "MI355X" and res["hdr"] is False
    data = (tmp_path / "sdr10.mp4").read_bytes()
    top = mp4.parse_boxes(data)

    def find(path, start, end):
        for name in path:
            _, start, end = [b for b in mp4.parse_boxes(data, start, end) if b[0] == name][0]
        return start, end
    s0, e0 = find(["moov", "trak", "mdia", "minf", "stbl", "stsd"], 0, len(data))
    entry = mp4.parse_boxes(data, s0 + 8, e0)[0]
    boxes = {b[0]: b for b in mp4.parse_boxes(data, entry[1] + 78, entry[2])}
    assert "mdcv" not in boxes and "clli" not in boxes       # SDR: no HDR10 boxes
    rec = data[boxes["hvcC"][1]:boxes["hvcC"][2]]
    annexb, q = b"", 23
    for _ in range(rec[22]):
        cnt = int.from_bytes(rec[q + 1:q + 3], "big")
        q += 3
        for _ in range(cnt):
            ln = int.from_bytes(rec[q:q + 2], "big")
            annexb += b"\0\0\0\1" + rec[q + 2:q + 2 + ln]
            q += 2 + ln
    q = top[2][1]
    while q < top[2][2]:
        ln = int.from_bytes(data[q:q + 4], "big")
        annexb += b"\0\0\0\1" + data[q + 4:q + 4 + ln]
        q += 4 + ln
    dec, info = O.decode(annexb)
    assert len(dec) == n and info["bit_depth"] == 10 and info["sps.profile_idc"] == 2
    for f, (y, u, v) in zip(dec, frames):
        assert util.psnr(f.y[:h, :w], y, peak=1023.0) > 24.0


def test_sharded_encode_cancel_and_worker_failure_leave_nothing_behind(lib, tmp_path, monkeypatch):
    """one clip over two sessions (devices [0, 0]): a cancel in the middle and a worker that dies must end with the workers joined, the
    sessions closed afterwards and no .mdat.tmp left; convert_video reports CANCELLED / falls down the ladder, never crashes or hangs"""
    from hevc_amd import encoder, transcoder as T
    p, _ = _clip(tmp_path, "long.y4m", 96, 80, 40)
    ev = threading.Event()
    before = threading.active_count()
    calls = []        # the callback ticks once per frame handed over (packets only appear when a chunk has been coded): cancel at the fifth
    res = T.convert_video(p, tmp_path, skip_validator=True, devices=[0, 0], stop_event=ev,
                          progress_callback=lambda n, f, t: calls.append(f) or (len(calls) >= 5 and ev.set()))
    assert res["status"] == "CANCELLED"
    assert not list(tmp_path.glob("*.mdat.tmp")) and threading.active_count() <= before
    # a worker that fails on its third frame: the error reaches the caller's thread, nothing blocks on the full queue
    real_send, seen = encoder.Encoder.send, [0]

    def flaky(self, y, u, v, pts=None):
        seen[0] += 1
        if seen[0] == 3:
            raise RuntimeError("injected worker failure")
        return real_send(self, y, u, v, pts=pts)
    monkeypatch.setattr(encoder.Encoder, "send", flaky)
    monkeypatch.setenv("PATH", "/nonexistent")               # no ffmpeg: the ladder ends in FAILED / CPU, as the reference does
    res = T.convert_video(p, tmp_path, skip_validator=True, devices=[0, 0])
    assert res["status"] == "FAILED" and res["method"] == "CPU"
    assert not list(tmp_path.glob("*.mdat.tmp")) and threading.active_count() <= before
    monkeypatch.undo()
    res = T.convert_video(p, tmp_path, skip_validator=True, devices=[0, 0])     # and the device is fine afterwards
    assert res["status"] == "SUCCESS" and res["method"] == "MI355X"


@pytest.mark.parametrize("w,h,bd,n_slices,level,rc", [(544, 320, 8, 3, 120, 0), (320, 200, 10, 2, 93, 1), (160, 96, 8, 2, 63, 1)])
def test_one_picture_as_slices_over_several_sessions(lib, w, h, bd, n_slices, level, rc):
    """BASELINE configs[4] with the devices [0, 0(, 0)]: every picture is cut into bands of CTU rows, one session per band, one slice each; the
    merged stream is decoded by the oracle decoder as ordinary pictures and must equal the bands' reconstructions stacked — which also proves
    that no motion vector reached across a band's edge (the decoder would have read the neighbour's samples, the encoder its own border)."""
    from hevc_amd import _lib
    from hevc_amd.encoder import SlicedEncoder, slice_rows
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.level_idc, cfg.keyint, cfg.min_keyint, cfg.me_range, cfg.gops_in_flight, cfg.aud = w, h, bd, level, 4, 2, 12, 2, 1
    if rc:
        cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits = 20, -1, 400, 480
    else:
        cfg.qp = 27
    n = 10
    frames = [util.synth_frame(h, w, seed=5, shift=(2 * i, 7 * i), bit_depth=bd) for i in range(n)]      # 7 rows of vertical motion per picture
    sl = SlicedEncoder(cfg, [0] * n_slices, keep_recon=True, halo=False)        # nothing is exchanged: motion-constrained slices (round 2's form)
    try:
        assert sl.rows == slice_rows(h, n_slices) and len(sl.rows) == n_slices
        got = []
        for f in frames:
            sl.send(*util.planes(f, bd))
            got += sl.ready()
        got += sl.finish()
        recs = [O.Frame(*sl.recon(i)) for i in range(n)]
        stats = sl.stats()
    finally:
        sl.close()
    assert [p for _, p, _ in got] == list(range(n)) and [k for _, _, k in got] == [i % 4 == 0 for i in range(n)]
    dec, info = O.decode(b"".join(d for d, _, _ in got))
    assert len(dec) == n and info["count.slices"] == n * n_slices and info["count.aud"] == n and (info["width"], info["conf_height"]) == (w, h)
    for i in range(n):
        d = O.Frame(dec[i].y[:recs[i].y.shape[0]], dec[i].u[:recs[i].u.shape[0]], dec[i].v[:recs[i].v.shape[0]])
        assert d.same(recs[i]), f"picture {i}: decoded picture != the slices' reconstructions"
        assert util.psnr(dec[i].y[:h], frames[i].y, peak=(1 << bd) - 1.0) > 24.0          # sanity only: 400 kb/s caps
    assert all(st.frames_out == n for st in stats)
    if w >= 512:
        assert info["pps.tile_cols"] == 2 and info["pps.tile_rows"] >= n_slices        # IDR pictures: tiles inside every slice


@pytest.mark.parametrize("w,h,bd,n_slices,level,rc", [(544, 320, 8, 3, 120, 0), (320, 200, 10, 2, 93, 1), (160, 160, 8, 4, 63, 1)])
def test_slices_that_exchange_rows_equal_the_whole_picture_pipeline(lib, w, h, bd, n_slices, level, rc):
    """BASELINE configs[4] as SURVEY §8e describes it, with the devices [0, 0(, 0 ...)]: the bands' sessions pull rows out of each other's pictures per
    step (cfg.slice_halo: reference rows for motion across the seams, pre-deblock rows + CU records for deblocking / SAO across them; on one device the
    peer pointers are ordinary device pointers) and plan ONE rate per picture.  The stacked reconstructions must equal, bit for bit, the oracle's
    WHOLE-PICTURE pipeline replayed with the session's QPs (tests/test_sliced_cpu.py halo_pipeline: only IDR analysis and search centres are per band),
    and the merged stream must decode to them.  (160x160 over 4: bands of 2 + 1 + 1 + 1 CTU rows, shorter than the 80 reference rows a band needs:
    the rows come from two and three bands away.)"""
    from hevc_amd import _lib
    from hevc_amd.encoder import SlicedEncoder, slice_rows
    from tests.test_gpu_configs import session_params
    from tests.test_sliced_cpu import halo_pipeline, mv_rows_ok
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.level_idc, cfg.keyint, cfg.min_keyint, cfg.me_range, cfg.gops_in_flight, cfg.aud = w, h, bd, level, 4, 2, 12, 2, 1
    if rc:
        cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits = 20, -1, 400, 480
    else:
        cfg.qp = 27
    n = 10
    frames = [util.synth_frame(h, w, seed=5, shift=(2 * i, 7 * i), bit_depth=bd) for i in range(n)]      # 7 rows of vertical motion per picture
    sl = SlicedEncoder(cfg, [0] * n_slices, keep_recon=True)
    try:
        assert sl.halo and sl.rows == slice_rows(h, n_slices)
        got = []
        for f in frames:
            sl.send(*util.planes(f, bd))
            got += sl.ready()
        got += sl.finish()
        recs = [O.Frame(*sl.recon(i)) for i in range(n)]
        infos = [[e.frame_info(i) for i in range(n)] for e in sl._encs]
        cfgs = sl._cfgs
    finally:
        sl.close()
    assert all([(q, t) for q, t, _ in inf] == [(q, t) for q, t, _ in infos[0]] for inf in infos), "the bands of a picture must agree on its type and QP (one rate plan)"
    if rc:
        assert len({q for q, t, _ in infos[0] if t == 1}) > 1 or max(q for q, _, _ in infos[0]) > cfg.crf + 2       # the cap had to bind
    idr_at = {i for i, (_, t, _) in enumerate(infos[0]) if t == 2}
    assert idr_at == {0, 4, 8}
    crossing, k = 0, 0
    ys = np.cumsum([0] + [32 * r for r in sl.rows])
    for i, (intra, a, sao, ref) in enumerate(halo_pipeline(frames, sl.rows, cfgs, None, None, None, bd, prm_of=lambda i, intra: session_params(lib, cfgs[0], infos[0][i][0], False)[0], idr_at=idr_at)):
        assert recs[i].same(ref), f"picture {i}: the bands' reconstructions != the whole-picture pipeline"
        if not intra:
            for (by, bx), r in np.ndenumerate(a.cu):
                nn, k = 1 << int(r["log2_size"]), int(np.searchsorted(ys, by * 8, side="right")) - 1
                crossing += not mv_rows_ok(((by * 8) & ~(nn - 1)) - ys[k], nn, int(r["mvy"]), min(h, ys[k + 1]) - ys[k], k > 0, k < n_slices - 1)
    assert crossing > 0, "no motion vector crosses a seam: the exchange is not exercised"
    dec, info = O.decode(b"".join(d for d, _, _ in got))
    assert len(dec) == n and info["count.slices"] == n * n_slices and info["count.aud"] == n
    for i in range(n):
        assert O.Frame(dec[i].y[:h], dec[i].u[:h // 2], dec[i].v[:h // 2]).same(recs[i]), f"picture {i}: decoded picture != the slices' reconstructions"


def test_convert_video_with_row_split(lib, tmp_path):
    from hevc_amd import mp4, transcoder as T
    p, frames = _clip(tmp_path, "rows.y4m", 320, 192, 9)
    res = T.convert_video(p, tmp_path, skip_validator=True, devices=[0, 0], row_split=True)
    assert res["status"] == "SUCCESS" and res["method"] == "MI355X"
    top = mp4.parse_boxes((tmp_path / "rows.mp4").read_bytes())
    assert [b[0] for b in top] == ["ftyp", "moov", "mdat"]


def test_scene_cuts_start_a_gop_and_min_keyint_holds(lib):
    """x265 scenecut + min-keyint (reference core/transcoder.py:401: keyint / min-keyint reach the encoder): IDR pictures where the content changes,
    never closer than min_keyint to the last one, and every keyint pictures otherwise; the stream still decodes to the encoder reconstruction, and
    the cut costs far fewer bits as an IDR than as a P picture predicted from the wrong scene."""
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    w, h, n = 192, 128, 40
    cuts = [7, 9, 22]                       # 9 is 2 pictures after 7: inside min_keyint, must stay a P picture
    frames, scene = [], 0
    for i in range(n):
        if i in cuts:
            scene += 1
        frames.append(util.synth_frame(h, w, seed=40 + scene, shift=(i, i // 2)))

    def encode(scenecut):
        cfg = _lib.default_config()
        cfg.width, cfg.height, cfg.keyint, cfg.min_keyint, cfg.me_range, cfg.gops_in_flight, cfg.qp, cfg.scenecut = w, h, 16, 4, 8, 2, 28, scenecut
        stream = b""
        with Encoder(cfg, device=0, keep_recon=True) as enc:
            for f in frames:
                enc.send(*util.planes(f, 8))
            enc.flush()
            sizes = []
            for data, pts, key in enc.packets():
                stream += data
                sizes.append(len(data))
            types = [enc.frame_info(i)[1] for i in range(n)]
            recs = [O.Frame(*enc.recon(i)) for i in range(n)]
        dec, _ = O.decode(stream)
        assert len(dec) == n and all(d.same(r) for d, r in zip(dec, recs))
        return [i for i, t in enumerate(types) if t == 2], sum(sizes), recs

    idr_on, bytes_on, recs_on = encode(1)
    idr_off, bytes_off, recs_off = encode(0)
    # chunks are gops_in_flight x keyint = 32 pictures: IDR at 0 (chunk), 7 (cut), 22 (cut: 9 is too close to 7), 23 = 7 + 16 is not needed
    # because the cut at 22 restarted the count; 32 opens the next chunk
    assert idr_off == [0, 16, 32]
    assert idr_on == [0, 7, 22, 32], idr_on
    assert bytes_on < 0.95 * bytes_off
    psnr = lambda recs: float(np.mean([util.psnr(r.y, f.y) for r, f in zip(recs, frames)]))     # noqa: E731
    assert psnr(recs_on) > psnr(recs_off) - 0.1


def test_resident_frames_handed_over_in_one_call_give_the_same_stream(lib):
    """mihevc_send_frames_device (n pictures already in HBM, one call) against mihevc_send_frame_device n times: the same bytes, packet by packet, with the clip spanning
    several chunks (chunks that fill up are coded inside the call).  The frames are put into device memory with the HIP runtime libmihevc itself links (hipMalloc /
    hipMemcpy through ctypes): no second runtime in the process, whatever test ran before."""
    import ctypes as C
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    from hevc_amd.yuvio import SyntheticClip
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    w, h, n = 320, 192, 23
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight, cfg.crf, cfg.qp, cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits, cfg.level_idc = w, h, 5, 2, 2, 22, -1, 800, 960, 93
    assert hip.hipSetDevice(0) == 0
    ptrs = []
    for f in SyntheticClip("motion", 3, w, h, n).frames():
        row = []
        for pl in f:
            pl = np.ascontiguousarray(pl)
            d = C.c_void_p()
            assert hip.hipMalloc(C.byref(d), pl.nbytes) == 0
            assert hip.hipMemcpy(d, pl.ctypes.data_as(C.c_void_p), pl.nbytes, 1) == 0      # hipMemcpyHostToDevice
            row.append(d.value)
        ptrs.append(row)
    try:
        out = []
        for batch in (False, True):
            with Encoder(cfg, device=0) as enc:
                if batch:
                    enc.send_device_batch([f[0] for f in ptrs], [f[1] for f in ptrs], [f[2] for f in ptrs], w, w // 2)
                else:
                    for i, f in enumerate(ptrs):
                        enc.send_device(f[0], f[1], f[2], w, w // 2, pts=i)
                enc.flush()
                out.append(list(enc.packets()))
                assert enc.stats().frames_out == n
        assert len(out[0]) == n and out[0] == out[1]
    finally:
        for row in ptrs:
            for d in row:
                hip.hipFree(d)
