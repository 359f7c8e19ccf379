"""CPU: the host-side failure paths the round-1 review flagged: MP4 chunk-offset width, the ffmpeg pipe front end's exit
status / frame count, the audio remux argv, sample-format handling of Encoder.send and config_for."""
import struct
import sys
from pathlib import Path

import numpy as np
import pytest

from hevc_amd import _lib, encoder, mp4, probe, yuvio

FAKE_FFMPEG = """#!/bin/sh
# stand-in for ffmpeg: records its argv; as a decoder (-f rawvideo ... -) it emits FAKE_FRAMES frames of FAKE_FB bytes
echo "$@" >> "$FAKE_LOG"
case "$*" in
  *"-f rawvideo"*" -") if [ -n "$FAKE_STDERR_KB" ]; then head -c $((FAKE_STDERR_KB*1024)) /dev/zero | tr '\\0' 'e' >&2; fi; i=0; while [ $i -lt ${FAKE_FRAMES:-0} ]; do head -c ${FAKE_FB:-0} /dev/zero; i=$((i+1)); done; exit ${FAKE_RC:-0};;
esac
for a in "$@"; do last="$a"; done
: > "$last"
exit ${FAKE_RC:-0}
"""


@pytest.fixture
def fake_ffmpeg(tmp_path, monkeypatch):
    b = tmp_path / "bin"
    b.mkdir()
    f = b / "ffmpeg"
    f.write_text(FAKE_FFMPEG)
    f.chmod(0o755)
    monkeypatch.setenv("PATH", f"{b}:/usr/bin:/bin")
    monkeypatch.setenv("FAKE_LOG", str(tmp_path / "ffmpeg.log"))
    return tmp_path / "ffmpeg.log"


def _info(w=64, h=48, n=4, audio=0, pix="yuv420p", hdr=False):
    return probe.VideoInfo(w, h, 30.0, "bt709", "bt709", "bt709", pix, "", "", audio, hdr, "eng", n, n / 30.0)


def test_pipe_clip_checks_exit_status_and_frame_count(fake_ffmpeg, tmp_path, monkeypatch):
    fb = 64 * 48 * 3 // 2
    monkeypatch.setenv("FAKE_FB", str(fb))
    monkeypatch.setenv("FAKE_FRAMES", "4")
    clip = yuvio.open_any(tmp_path / "a.mkv", _info(n=4))
    assert len(list(clip.frames())) == 4 and clip.bit_depth == 8
    monkeypatch.setenv("FAKE_RC", "1")                      # decode error after the frames: must not look like a complete clip
    with pytest.raises(RuntimeError, match="ffmpeg decode failed"):
        list(yuvio.open_any(tmp_path / "a.mkv", _info(n=4)).frames())
    monkeypatch.delenv("FAKE_RC")
    monkeypatch.setenv("FAKE_FRAMES", "40")                 # a stream that ends early
    with pytest.raises(RuntimeError, match="decoded 40 of 300"):
        list(yuvio.open_any(tmp_path / "a.mkv", _info(n=300)).frames())
    clip10 = yuvio.open_any(tmp_path / "a.mkv", _info(pix="yuv420p10le"))      # 10-bit SDR: depth follows the sample format
    assert clip10.bit_depth == 10
    clip10.close()


def test_pipe_clip_survives_a_decoder_that_floods_stderr(fake_ffmpeg, tmp_path, monkeypatch):
    """a damaged input can make `ffmpeg -v error` write more than a pipe buffer (64 KiB) to stderr before its first frame; with stderr
    on a pipe that nobody reads the child blocked there while frames() blocked on stdout.  Now: the frames arrive, the exit status decides."""
    import threading
    fb = 64 * 48 * 3 // 2
    monkeypatch.setenv("FAKE_FB", str(fb))
    monkeypatch.setenv("FAKE_FRAMES", "4")
    monkeypatch.setenv("FAKE_STDERR_KB", "300")
    got = []
    t = threading.Thread(target=lambda: got.append(len(list(yuvio.open_any(tmp_path / "a.mkv", _info(n=4)).frames()))), daemon=True)
    t.start()
    t.join(30)
    assert not t.is_alive(), "frames() hangs behind a full stderr pipe"
    assert got == [4]
    monkeypatch.setenv("FAKE_RC", "1")
    with pytest.raises(RuntimeError, match="ffmpeg decode failed.*eeee"):
        list(yuvio.open_any(tmp_path / "a.mkv", _info(n=4)).frames())


def test_audio_remux_uses_the_reference_audio_flags(fake_ffmpeg, tmp_path):
    from hevc_amd.transcoder import get_audio_flags
    info = _info(audio=6)
    assert encoder.remux_audio(tmp_path / "v.mp4", tmp_path / "src.mkv", tmp_path / "out.mp4", info)
    argv = fake_ffmpeg.read_text().split()
    for tok in get_audio_flags(6) + ["-c:v", "copy", "-map", "0:v:0", "1:a:0?", "hvc1"]:
        assert tok in argv, tok
    assert argv[-1] == str(tmp_path / "out.mp4")


def test_audio_remux_without_ffmpeg_reports_failure(tmp_path, monkeypatch):
    monkeypatch.setenv("PATH", "/nonexistent")
    assert not encoder.remux_audio(tmp_path / "v.mp4", tmp_path / "src.mkv", tmp_path / "out.mp4", _info(audio=2))


def test_bit_depth_follows_the_sample_format():
    assert encoder.bit_depth_of(_info()) == 8
    assert encoder.bit_depth_of(_info(pix="yuv420p10le")) == 10 and encoder.bit_depth_of(_info(pix="p010le")) == 10
    assert encoder.bit_depth_of(_info(hdr=True)) == 10
    # the depth is read from the END of the name: 8-bit formats whose chroma layout spells a "10" or "11" stay 8 bit
    assert [encoder.bit_depth_of(_info(pix=f)) for f in ("yuv410p", "yuvj411p", "yuv411p", "nv12", "yuv444p")] == [8] * 5
    assert [encoder.bit_depth_of(_info(pix=f)) for f in ("p016le", "yuv420p12le", "yuv422p10be", "p010")] == [10] * 4
    cfg = encoder.config_for(_info(pix="yuv420p10le"), 19, 600, 720, 90, "3", "main")
    assert cfg.bit_depth == 10 and cfg.hdr10 == 0 and cfg.colour_primaries == 1       # Main10 without the HDR10 SEI set


def test_send_rejects_wrong_shape_and_too_wide_samples():
    class Dummy(encoder.Encoder):
        def __init__(self, cfg):
            self.cfg, self._pts = cfg, 0
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth = 64, 48, 8
    e = Dummy(cfg)
    y, u, v = np.zeros((48, 64), np.uint8), np.zeros((24, 32), np.uint8), np.zeros((24, 32), np.uint8)
    with pytest.raises(ValueError, match="do not match"):
        e.send(y[:40], u, v)
    with pytest.raises(ValueError, match="do not match"):
        e.send(y, u[:, :30], v)
    with pytest.raises(ValueError, match="8-bit session"):
        e.send(y.astype(np.uint16) + 700, u, v)                 # would wrap modulo 256


def test_mp4_chunk_offset_width_is_decided_once(tmp_path):
    cfg = _lib.default_config()
    cfg.width, cfg.height = 64, 48
    buf = (__import__("ctypes").c_uint8 * 4096)()
    n = _lib.load().mihevc_write_parameter_sets(__import__("ctypes").byref(cfg), buf, len(buf))
    headers = bytes(buf[:n])
    for sizes, want64 in (([1000] * 10, False), ([1 << 30] * 4, True), ([(1 << 32) - 4], True), ([(1 << 32) - 200], False)):
        w = mp4.Mp4Writer(tmp_path / "x.mp4", cfg)
        w._sizes, w._sync, w.dry_run = list(sizes), [1], True
        w.finish(headers)
        lay = w.layout
        assert lay["co64"] == want64 and lay["big_mdat"] == want64
        top = mp4.parse_boxes(w._moov)
        assert top[0][0] == "moov"
        data = w._moov

        def find(path, start, end):
            for name in path:
                _, start, end = [b for b in mp4.parse_boxes(data, start, end) if b[0] == name][0]
            return start, end
        s0, e0 = find(["moov", "trak", "mdia", "minf", "stbl"], 0, len(data))
        kinds = {b[0]: b for b in mp4.parse_boxes(data, s0, e0)}
        assert ("co64" in kinds) == want64 and ("stco" in kinds) != want64
        b = kinds["co64" if want64 else "stco"]
        off = struct.unpack(">Q" if want64 else ">I", data[b[1] + 8:b[2]])[0]
        assert off == lay["chunk_offset"] == 28 + len(data) + (16 if want64 else 8)      # ftyp (28 bytes) + moov + mdat header
        w.abort()


def test_cpulist_parsing_and_numa_binding_is_optional():
    """hevc_amd.utils.bind_to_device_node: one process per GPU pins itself to the device's NUMA node; without a device (this container) or without
    sysfs information it returns None and leaves the affinity alone."""
    import os
    from hevc_amd.utils import bind_to_device_node, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and parse_cpulist("") == set() and parse_cpulist("5") == {5}
    before = os.sched_getaffinity(0)
    assert bind_to_device_node(0) is None or len(os.sched_getaffinity(0)) >= 8
    if bind_to_device_node(99) is None:
        pass
    assert os.sched_getaffinity(0) <= before
