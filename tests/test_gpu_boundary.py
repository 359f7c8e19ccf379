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
