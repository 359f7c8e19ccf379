"""Raw clip IO and the deterministic synthetic clips of the measurement plan.

The reference generates its test inputs with `ffmpeg -f lavfi -i testsrc=...` (tests/generate_test_videos.py:26-31)
and decodes arbitrary containers inside ffmpeg (core/transcoder.py:461).  Neither is possible without ffmpeg, so
the native path reads planar 4:2:0 directly (`.y4m`, or `.yuv` named `<stem>_<W>x<H>_<fps>[_10bit][_hdr].yuv`) and,
when ffmpeg IS present, any other container through an `ffmpeg -f rawvideo` pipe.
"""
from __future__ import annotations

import re
import shutil
import subprocess
import tempfile
from fractions import Fraction
from pathlib import Path
from typing import Iterator, Optional, Tuple

import numpy as np

Planes = Tuple[np.ndarray, np.ndarray, np.ndarray]
_YUV_NAME = re.compile(r'_(\d+)x(\d+)_(\d+(?:\.\d+)?)(?:fps)?(?:_(8|10)bit)?(?:_(hdr|sdr))?$', re.I)


class Clip:
    """A planar 4:2:0 clip on disk."""

    def __init__(self, path: Path, width: int, height: int, fps: float, bit_depth: int, n_frames: int, data_offset: int,
                 frame_header: int = 0, hdr: bool = False):
        self.path, self.width, self.height, self.fps, self.bit_depth = Path(path), width, height, fps, bit_depth
        self.n_frames, self.data_offset, self.frame_header, self.hdr = n_frames, data_offset, frame_header, hdr
        self._f = open(self.path, 'rb')

    @property
    def frame_bytes(self) -> int:
        return self.width * self.height * 3 // 2 * (2 if self.bit_depth > 8 else 1)

    def frames(self) -> Iterator[Planes]:
        dt = np.dtype('<u2') if self.bit_depth > 8 else np.uint8
        w, h = self.width, self.height
        self._f.seek(self.data_offset)
        for _ in range(self.n_frames):
            if self.frame_header:
                line = self._f.readline()
                if not line.startswith(b'FRAME'):
                    return
            buf = self._f.read(self.frame_bytes)
            if len(buf) < self.frame_bytes:
                return
            a = np.frombuffer(buf, dt)
            yield a[:w * h].reshape(h, w), a[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), a[w * h * 5 // 4:].reshape(h // 2, w // 2)

    def close(self):
        self._f.close()


def open_clip(path: Path) -> Clip:
    path = Path(path)
    if path.suffix.lower() == '.y4m':
        with open(path, 'rb') as f:
            header = f.readline()
        if not header.startswith(b'YUV4MPEG2'):
            raise ValueError('not a YUV4MPEG2 file')
        w = h = 0
        fps, depth = 30.0, 8
        for tok in header.split()[1:]:
            t = tok.decode()
            if t[0] == 'W':
                w = int(t[1:])
            elif t[0] == 'H':
                h = int(t[1:])
            elif t[0] == 'F':
                n, d = t[1:].split(':')
                fps = int(n) / int(d) if int(d) else 30.0
            elif t[0] == 'C':
                if not t[1:].startswith('420'):
                    raise ValueError(f'unsupported chroma format {t}')
                depth = 10 if 'p10' in t else 8
        fb = w * h * 3 // 2 * (2 if depth > 8 else 1)
        n_frames = (path.stat().st_size - len(header)) // (fb + 6)
        return Clip(path, w, h, fps, depth, n_frames, len(header), frame_header=6, hdr=depth > 8 and 'hdr' in path.stem.lower())
    m = _YUV_NAME.search(path.stem)
    if not m:
        raise ValueError('raw .yuv needs <stem>_<W>x<H>_<fps>[_10bit][_hdr].yuv')
    w, h, fps = int(m.group(1)), int(m.group(2)), float(m.group(3))
    depth = int(m.group(4) or 8)
    fb = w * h * 3 // 2 * (2 if depth > 8 else 1)
    return Clip(path, w, h, fps, depth, path.stat().st_size // fb, 0, hdr=(m.group(5) or '').lower() == 'hdr')


class _PipeClip:
    """Any container decoded by an `ffmpeg -f rawvideo` child (only when ffmpeg exists on this host).  A decode error or a short
    stream raises: a truncated clip must not come out as SUCCESS."""

    def __init__(self, path: Path, info):
        self.width, self.height, self.fps = info.width, info.height, info.fps
        from .encoder import bit_depth_of
        self.bit_depth = bit_depth_of(info)
        self.n_frames = info.nb_frames or (int(info.duration * info.fps) if info.duration and info.fps else 0)
        pix = 'yuv420p10le' if self.bit_depth > 8 else 'yuv420p'
        self._name = Path(path).name
        # stderr goes to a file, not a pipe: a damaged input can make `-v error` write more than a pipe buffer holds before the first frame, and a
        # child blocked on stderr while frames() blocks on stdout would hang the encode for ever instead of failing it
        self._err = tempfile.TemporaryFile()
        self._p = subprocess.Popen(['ffmpeg', '-v', 'error', '-i', str(path), '-f', 'rawvideo', '-pix_fmt', pix, '-'],
                                   stdout=subprocess.PIPE, stderr=self._err)

    def frames(self) -> Iterator[Planes]:
        w, h = self.width, self.height
        fb = w * h * 3 // 2 * (2 if self.bit_depth > 8 else 1)
        dt = np.dtype('<u2') if self.bit_depth > 8 else np.uint8
        n = 0
        while True:
            buf = self._p.stdout.read(fb)
            if len(buf) < fb:
                break
            a = np.frombuffer(buf, dt)
            n += 1
            yield a[:w * h].reshape(h, w), a[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), a[w * h * 5 // 4:].reshape(h // 2, w // 2)
        rc = self._p.wait()
        self._err.seek(0, 2)
        self._err.seek(max(0, self._err.tell() - 2000))
        err = self._err.read().decode('utf-8', 'replace')[-500:]
        if rc != 0:
            raise RuntimeError(f'{self._name}: ffmpeg decode failed (exit {rc}): {err}')
        if self.n_frames and n < self.n_frames - max(2, self.n_frames // 100):      # container frame counts can be off by a frame or two
            raise RuntimeError(f'{self._name}: decoded {n} of {self.n_frames} frames')

    def close(self):
        try:
            self._p.kill()
            self._p.wait()
        except Exception:
            pass
        try:
            self._p.stdout.close()
            self._err.close()
        except Exception:
            pass


def open_any(path: Path, info=None):
    path = Path(path)
    if path.suffix.lower() in ('.y4m', '.yuv'):
        return open_clip(path)
    if shutil.which('ffmpeg') is None:
        raise RuntimeError(f'{path.name}: only .y4m/.yuv can be read without ffmpeg on this host')
    return _PipeClip(path, info)


def write_y4m(path: Path, frames, width: int, height: int, fps=30, bit_depth: int = 8):
    fr = Fraction(str(fps)).limit_denominator(1001)
    tag = 'C420p10' if bit_depth > 8 else 'C420jpeg'
    with open(path, 'wb') as f:
        f.write(f'YUV4MPEG2 W{width} H{height} F{fr.numerator}:{fr.denominator} Ip A1:1 {tag}\n'.encode())
        dt = np.dtype('<u2') if bit_depth > 8 else np.uint8
        for y, u, v in frames:
            f.write(b'FRAME\n')
            for p in (y, u, v):
                f.write(np.ascontiguousarray(p, dtype=dt).tobytes())


def write_yuv(path: Path, frames, bit_depth: int = 8):
    dt = np.dtype('<u2') if bit_depth > 8 else np.uint8
    with open(path, 'wb') as f:
        for y, u, v in frames:
            for p in (y, u, v):
                f.write(np.ascontiguousarray(p, dtype=dt).tobytes())


# ---------------------------------------------------------------------------------------------
# synthetic clips: deterministic from (pattern, seed, W, H, n, bit_depth)
# ---------------------------------------------------------------------------------------------
class SyntheticClip:
    """`bars`: static colour bars + sweeping gradient + frame counter block (mirrors lavfi testsrc's intent).
    `motion`: band-limited Gaussian texture translating (+3,+1) px/frame, two rectangles on independent paths,
    i.i.d. grain (sigma 2 LSB at 8 bit, 8 LSB at 10 bit) re-seeded per frame — the headline workload.
    `stress`: what a translational search and a per-GOP rate plan do not like — the texture ZOOMS (scale 1 .. 1.5 and back), the whole picture FADES to
    dark and back, twelve small occluders move on their own paths, one hard cut (frame n/2: another texture) and a one-frame white flash (frame n/4):
    for the rate-control / scene-cut / conformance tests, not a benchmark."""

    def __init__(self, pattern: str, seed: int, width: int, height: int, n_frames: int, bit_depth: int = 8, fps: float = 30.0):
        assert pattern in ('bars', 'motion', 'stress')
        self.pattern, self.seed, self.width, self.height, self.n_frames, self.bit_depth, self.fps = pattern, seed, width, height, n_frames, bit_depth, fps
        self.hdr = bit_depth > 8
        self._tex = None

    def _texture(self):
        if self._tex is None:
            rng = np.random.default_rng(self.seed)
            th, tw = self.height + 64, self.width + 64
            t = rng.standard_normal((th, tw)).astype(np.float32)
            for _ in range(3):                       # box blur x3 ~ Gaussian, with wrap-around so the field tiles
                for ax in (0, 1):
                    t = sum(np.roll(t, k, ax) for k in range(-3, 4)) / 7.0
            t = (t - t.mean()) / (t.std() + 1e-6)
            self._tex = np.tile(t, (2, 2))          # tiled twice so a wrapped window is one contiguous slice
        return self._tex

    def frame(self, i: int) -> Planes:
        w, h, bd = self.width, self.height, self.bit_depth
        lo, hi = (16, 235) if bd == 8 else (64, 940)
        sc = 1 << (bd - 8)
        if self.pattern == 'bars':
            xs = np.arange(w)
            bar = (xs * 8 // w)
            lum = np.array([235, 210, 170, 145, 106, 81, 41, 16], np.float32)[bar]
            y = np.tile(lum, (h, 1))
            y[h * 3 // 4:] = ((xs + 4 * i) % w) * (219.0 / w) + 16          # sweeping gradient
            cb = np.array([128, 16, 166, 54, 202, 90, 240, 128], np.float32)[bar[::2]]
            cr = np.array([128, 146, 16, 34, 222, 240, 110, 128], np.float32)[bar[::2]]
            u, v = np.tile(cb, (h // 2, 1)), np.tile(cr, (h // 2, 1))
            blk = 32
            bits = [(i >> k) & 1 for k in range(8)]                            # frame counter block
            for k, b in enumerate(bits):
                y[8:8 + blk, 8 + k * blk:8 + (k + 1) * blk] = 235 if b else 16
            y, u, v = y * sc, u * sc, v * sc
        elif self.pattern == 'stress':
            n = max(2, self.n_frames)
            t = self._texture()
            if i >= n // 2:
                t = t[::-1, ::-1]                                  # the scene after the cut
            th, tw = t.shape[0] // 2, t.shape[1] // 2
            zoom = 1.0 + 0.5 * abs(np.sin(np.pi * i / 40.0))
            ys = (np.arange(h, dtype=np.float32) - h / 2) / zoom + h / 2 + 0.5 * i
            xs = (np.arange(w, dtype=np.float32) - w / 2) / zoom + w / 2 + 1.5 * i
            win = t[np.mod(np.rint(ys).astype(np.int64), th)[:, None], np.mod(np.rint(xs).astype(np.int64), tw)[None, :]]
            fade = 0.35 + 0.65 * abs(np.cos(np.pi * i / 55.0))
            y = 126 + 44 * win
            g = np.random.default_rng((self.seed + 7) * 100003 + i)
            for k in range(12):                                    # occluders: position from a per-object linear path
                rw, rh = w // (12 + k), h // (9 + k % 5)
                x0 = (37 * k * k + (3 + k % 7 - 3) * 2 * i) % max(1, w - rw)
                y0 = (91 * k + (k % 5 - 2) * 3 * i) % max(1, h - rh)
                y[y0:y0 + rh, x0:x0 + rw] = 40 + 17 * k
            y = 16 + (y - 16) * fade + 1.5 * g.standard_normal((h, w), dtype=np.float32)
            u = 128 + 30 * win[::2, ::2] * fade + g.standard_normal((h // 2, w // 2), dtype=np.float32)
            v = 128 - 22 * win[::2, ::2] * fade + g.standard_normal((h // 2, w // 2), dtype=np.float32)
            if i == n // 4:
                y = y * 0 + 225
            y, u, v = y * sc, u * sc, v * sc
        else:
            t = self._texture()
            th, tw = t.shape[0] // 2, t.shape[1] // 2
            r0, c0 = (1 * i) % th, (3 * i) % tw
            win = t[r0:r0 + h, c0:c0 + w]
            y = 126 + 38 * win
            # two rectangles on independent paths
            for k, (rw, rh, vx, vy, lum) in enumerate(((w // 8, h // 6, 5, 2, 200), (w // 10, h // 5, -4, 3, 60))):
                x0 = (w // 4 * (k + 1) + vx * i) % max(1, w - rw)
                y0 = (h // 3 * (k + 1) + vy * i) % max(1, h - rh)
                y[y0:y0 + rh, x0:x0 + rw] = lum
            g = np.random.default_rng((self.seed + 1) * 100003 + i)
            y = y + 2.0 * g.standard_normal((h, w), dtype=np.float32)
            yy, xx = np.mgrid[0:h // 2, 0:w // 2]
            u = 128 + 24 * win[::2, ::2] + g.standard_normal((h // 2, w // 2), dtype=np.float32)
            v = 128 - 24 * win[::2, ::2] + 10 * np.sin((xx + 3 * i / 2) / 40.0).astype(np.float32) + g.standard_normal((h // 2, w // 2), dtype=np.float32)
            y, u, v = y * sc, u * sc, v * sc
        dt = np.uint16 if bd > 8 else np.uint8
        return (np.clip(np.rint(y), lo, hi).astype(dt), np.clip(np.rint(u), lo, hi + 5 * sc).astype(dt), np.clip(np.rint(v), lo, hi + 5 * sc).astype(dt))

    def frames(self) -> Iterator[Planes]:
        for i in range(self.n_frames):
            yield self.frame(i)

    def close(self):
        pass
