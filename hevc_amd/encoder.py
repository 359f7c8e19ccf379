"""Python driver of the native MI355X encoder (C ABI in include/mihevc.h, loaded by hevc_amd._lib).

`Encoder` is a thin object wrapper over one `mihevc_session`; `encode_file` is what `convert_video` calls for the
'MI355X' backend: probe -> operating point (the same numbers the reference hands to libx265,
core/transcoder.py:357-412) -> read frames -> session -> Annex-B packets -> MP4 (hvc1).  ctypes releases the GIL
during every call, so N worker threads can each drive their own session like the reference's N QThreads each
block on their own ffmpeg child (gui/worker.py:30-52).
"""
from __future__ import annotations

import ctypes as C
import logging
import re
import threading
from fractions import Fraction
from pathlib import Path
from typing import Callable, Iterator, Optional, Tuple

import numpy as np

from . import _lib
from .probe import VideoInfo

logger = logging.getLogger(__name__)

_COLOUR_CODES = {  # ffprobe names -> H.265 Table E.3/E.4/E.5 code points
    "primaries": {"bt709": 1, "bt470bg": 5, "smpte170m": 6, "bt2020": 9},
    "transfer": {"bt709": 1, "smpte170m": 6, "smpte2084": 16, "pq": 16, "arib-std-b67": 18, "bt2020-10": 14},
    "matrix": {"bt709": 1, "bt470bg": 5, "smpte170m": 6, "bt2020nc": 9, "bt2020-ncl": 9, "bt2020": 9},
}


_DEPTH_IN_FMT = re.compile(r'(?:p|gray|gbrp|yuva?\d{3}p?)(9|10|12|14|16)(?:le|be)?$|^(?:p0|p2|p4|y2)(10|12|16)(?:le|be)?$')


def bit_depth_of(info: VideoInfo) -> int:
    """10 for sample formats deeper than 8 bit (yuv420p10le, p010le, yuv420p12le ... — this path codes Main or Main10, so 12 / 16-bit sources come
    in as 10) and for anything probed as HDR, else 8.  The depth is parsed from the END of the format name: yuv410p / yuvj411p are 8-bit formats
    whose chroma layout happens to spell a '10' or '11'.  Deviation from the reference, documented in INTEGRATION.md §4: its libx265 branch codes
    every non-HDR input as 8-bit Main (core/transcoder.py:363-364); the native path keeps a 10-bit SDR source's samples (Main10, no HDR10 SEI)."""
    if info.hdr:
        return 10
    m = _DEPTH_IN_FMT.search((info.pix_fmt or '').lower())
    return 10 if m and int(m.group(1) or m.group(2)) > 8 else 8


def lanes_for(total_frames: int, keyint: int) -> int:
    """GOP lanes of the session's lock-step pipeline (`gops_in_flight`): a chunk is lanes x keyint pictures and its GOPs run side by side, so a clip
    of up to 8 GOPs is one chunk with every GOP its own lane, longer clips run 8 at a time (a step's fixed part — small kernels, launch ramps, the IDR chain —
    is shared by more pictures: 4609 -> 5150 fps from 4 to 8 lanes on a 1440-frame 1080p clip, no more at 12 or 16); 4 is the floor and what an unknown length gets."""
    if total_frames <= 0 or keyint <= 0:
        return 4
    return max(4, min(8, -(-total_frames // keyint)))


def config_for(info: VideoInfo, crf: int, vbv_maxrate: int, vbv_bufsize: int, gop: int, level: str, tier: str,
               master_display: str = "", max_cll: str = "") -> _lib.Config:
    """Map the reference's libx265 operating point (build_ffmpeg_params CPU branch) onto a mihevc_config."""
    from .utils import parse_master_display, parse_max_cll
    cfg = _lib.default_config()
    cfg.width, cfg.height = int(info.width), int(info.height)
    fr = Fraction(str(info.fps or 30.0)).limit_denominator(1001)
    cfg.fps_num, cfg.fps_den = fr.numerator, fr.denominator
    hdr = bool(info.hdr)
    # bit depth follows the SAMPLE format, not the HDR vote: a 10-bit SDR source is coded Main10 without the HDR10 SEI set
    cfg.bit_depth = bit_depth_of(info)
    cfg.level_idc = int(round(float(level) * 30))
    cfg.tier = 1 if tier == "high" else 0
    cfg.crf, cfg.qp = int(crf), -1
    cfg.vbv_maxrate_kbps, cfg.vbv_bufsize_kbits = int(vbv_maxrate), int(vbv_bufsize)
    cfg.keyint, cfg.min_keyint = int(gop), max(2, int(gop) // 2)
    cfg.gops_in_flight = lanes_for(int(getattr(info, 'nb_frames', 0) or 0), int(gop))
    if hdr:   # the HDR10 set of core/utils.py:58-69
        cfg.colour_primaries, cfg.transfer, cfg.matrix = 9, 16, 9
        cfg.chroma_loc, cfg.aud, cfg.repeat_headers, cfg.hdr10, cfg.hrd = 0, 1, 1, 1, 1       # ... hrd=1:aud=1:chromaloc=0:repeat-headers=1
        md = parse_master_display(master_display or info.master_display)
        for i, (x, y) in enumerate((md.g, md.b, md.r)):
            cfg.md_primaries[i][0], cfg.md_primaries[i][1] = x, y
        cfg.md_white[0], cfg.md_white[1] = md.wp
        cfg.md_max_lum, cfg.md_min_lum = md.lum
        cfg.max_cll, cfg.max_fall = parse_max_cll(max_cll or info.max_cll)
    else:
        cfg.colour_primaries = _COLOUR_CODES["primaries"].get(info.color_primaries, 1)
        cfg.transfer = _COLOUR_CODES["transfer"].get(info.color_transfer, 1)
        cfg.matrix = _COLOUR_CODES["matrix"].get(info.color_space, 1)
    return cfg


class Encoder:
    """One encode session on one MI355X.  Raises _lib.MihevcError on any native failure (never falls back)."""

    def __init__(self, cfg: _lib.Config, device: int = 0, keep_recon: bool = False):
        self._lib = _lib.load()
        self.cfg = cfg
        self._s = C.c_void_p()
        rc = self._lib.mihevc_open(C.byref(cfg), device, C.byref(self._s))
        if rc != 0:
            raise _lib.MihevcError(rc, "mihevc_open")
        if keep_recon:
            self._lib.mihevc_set_keep_recon(self._s, 1)
        self._pts = 0

    # -- lifecycle
    def close(self):
        if self._s:
            self._lib.mihevc_close(self._s)
            self._s = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            detail = self._lib.mihevc_last_error(self._s).decode() if self._s else ""
            raise _lib.MihevcError(rc, f"{what} {detail}".strip())

    # -- data path
    def send(self, y: np.ndarray, u: np.ndarray, v: np.ndarray, pts: Optional[int] = None):
        """Planes as uint8 (8 bit) or uint16 (10 bit) arrays of the DISPLAY size.  Shape and sample width are checked here: the C ABI takes
        plain pointers, so a short plane would be read past its end and 10-bit samples handed to an 8-bit session would wrap modulo 256."""
        c = self.cfg
        dt = np.uint8 if c.bit_depth == 8 else np.uint16
        if y.shape != (c.height, c.width) or u.shape != (c.height // 2, c.width // 2) or v.shape != u.shape:
            raise ValueError(f"plane shapes {y.shape}/{u.shape}/{v.shape} do not match the session's {c.width}x{c.height} 4:2:0")
        for p in (y, u, v):            # a wider container is fine as long as the VALUES fit (tests keep 8-bit pictures in uint16 arrays)
            if p.dtype.itemsize > np.dtype(dt).itemsize and p.size and int(p.max()) >> c.bit_depth:
                raise ValueError(f"{p.dtype} samples up to {int(p.max())} handed to a {c.bit_depth}-bit session (open it with bit_depth 10 or down-convert first)")
        y, u, v = (np.ascontiguousarray(p, dtype=dt) for p in (y, u, v))
        pts = self._pts if pts is None else pts
        self._pts = pts + 1
        self._check(self._lib.mihevc_send_frame(self._s, y.ctypes.data, u.ctypes.data, v.ctypes.data, y.shape[1], u.shape[1], pts), "send_frame")

    def send_async(self, y: np.ndarray, u: np.ndarray, v: np.ndarray, pts: Optional[int] = None):
        """mihevc_send_frame_async: returns with the upload in flight.  The arrays must already have the session's sample type and be C-contiguous (no
        copy is made here, that is the point) and must stay alive and unmodified until sync_uploads() or flush(); page-locked memory makes the copies DMA."""
        c = self.cfg
        dt = np.uint8 if c.bit_depth == 8 else np.uint16
        for p in (y, u, v):
            if p.dtype != dt or not p.flags.c_contiguous:
                raise ValueError("send_async needs C-contiguous planes of the session's sample type")
        if y.shape != (c.height, c.width) or u.shape != (c.height // 2, c.width // 2) or v.shape != u.shape:
            raise ValueError(f"plane shapes {y.shape}/{u.shape}/{v.shape} do not match the session's {c.width}x{c.height} 4:2:0")
        pts = self._pts if pts is None else pts
        self._pts = pts + 1
        self._check(self._lib.mihevc_send_frame_async(self._s, y.ctypes.data, u.ctypes.data, v.ctypes.data, y.shape[1], u.shape[1], pts), "send_frame_async")

    def sync_uploads(self):
        self._check(self._lib.mihevc_sync_uploads(self._s), "sync_uploads")

    def send_device(self, y_ptr: int, u_ptr: int, v_ptr: int, pitch_y: int, pitch_c: int, pts: Optional[int] = None):
        pts = self._pts if pts is None else pts
        self._pts = pts + 1
        self._check(self._lib.mihevc_send_frame_device(self._s, y_ptr, u_ptr, v_ptr, pitch_y, pitch_c, pts), "send_frame_device")

    def send_device_batch(self, y_ptrs, u_ptrs, v_ptrs, pitch_y: int, pitch_c: int, first_pts: Optional[int] = None):
        """mihevc_send_frames_device: len(y_ptrs) pictures already in device memory in ONE call.  The pointer lists may be ctypes arrays of c_void_p made once
        for frames that stay where they are (device_pointer_arrays)."""
        n = len(y_ptrs)
        arrs = [p if isinstance(p, C.Array) else (C.c_void_p * n)(*p) for p in (y_ptrs, u_ptrs, v_ptrs)]
        pts = self._pts if first_pts is None else first_pts
        self._pts = pts + n
        self._check(self._lib.mihevc_send_frames_device(self._s, n, arrs[0], arrs[1], arrs[2], pitch_y, pitch_c, pts), "send_frames_device")

    def flush(self):
        self._check(self._lib.mihevc_flush(self._s), "flush")

    def abort(self):
        """mihevc_abort: give the session up (every later call fails; sessions of the same picture's other slices stop waiting for it)"""
        if self._s:
            self._lib.mihevc_abort(self._s)

    def packets(self) -> Iterator[Tuple[bytes, int, bool]]:
        """Drain every packet that is ready: (annexb bytes, pts, keyframe)."""
        data, size = C.POINTER(C.c_uint8)(), C.c_size_t()
        pts, dts, key = C.c_int64(), C.c_int64(), C.c_int()
        while True:
            rc = self._lib.mihevc_receive_packet(self._s, C.byref(data), C.byref(size), C.byref(pts), C.byref(dts), C.byref(key))
            if rc in (_lib.EAGAIN, _lib.EOF):
                return
            self._check(rc, "receive_packet")
            yield C.string_at(data, size.value), pts.value, bool(key.value)

    def packets_dts(self) -> Iterator[Tuple[bytes, int, bool, int]]:
        """The same with the decoding time stamp: (annexb bytes, pts, keyframe, dts).  Packets always come in DECODING order; dts differs from pts only
        when the session codes B pictures (cfg.bframes)."""
        data, size = C.POINTER(C.c_uint8)(), C.c_size_t()
        pts, dts, key = C.c_int64(), C.c_int64(), C.c_int()
        while True:
            rc = self._lib.mihevc_receive_packet(self._s, C.byref(data), C.byref(size), C.byref(pts), C.byref(dts), C.byref(key))
            if rc in (_lib.EAGAIN, _lib.EOF):
                return
            self._check(rc, "receive_packet")
            yield C.string_at(data, size.value), pts.value, bool(key.value), dts.value

    def headers(self) -> bytes:
        data, size = C.POINTER(C.c_uint8)(), C.c_size_t()
        self._check(self._lib.mihevc_get_headers(self._s, C.byref(data), C.byref(size)), "get_headers")
        return C.string_at(data, size.value)

    def stats(self) -> _lib.Stats:
        st = _lib.Stats()
        self._check(self._lib.mihevc_get_stats(self._s, C.byref(st)), "get_stats")
        return st

    def coded_size(self) -> Tuple[int, int]:
        w, h = C.c_int(), C.c_int()
        self._lib.mihevc_coded_size(self._s, C.byref(w), C.byref(h))
        return w.value, h.value

    def recon(self, index: int):
        w, h = self.coded_size()
        y, u, v = np.zeros((h, w), np.uint16), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint16)
        self._check(self._lib.mihevc_get_recon(self._s, index, y.ctypes.data, u.ctypes.data, v.ctypes.data), "get_recon")
        return y, u, v

    def frame_info(self, index: int):
        """(qp, slice_type, bits) of output picture `index`"""
        qp, st, bits = C.c_int(), C.c_int(), C.c_int64()
        self._check(self._lib.mihevc_get_frame_info(self._s, index, C.byref(qp), C.byref(st), C.byref(bits)), "get_frame_info")
        return qp.value, st.value, bits.value

    def psnr_y(self) -> float:
        st = self.stats()
        n = max(1, st.frames_out) * self.coded_size()[0] * self.coded_size()[1]
        peak = (1 << self.cfg.bit_depth) - 1
        mse = st.sse_y / n
        return 99.0 if mse <= 0 else float(10 * np.log10(peak * peak / mse))


class ShardedEncoder:
    """One clip over several MI355X: closed GOPs are independent, so consecutive chunks of `gops_in_flight x keyint` pictures go
    round-robin to one session per device (SURVEY.md §8e "GOP g -> device g mod n"); nothing is exchanged between the devices
    (no RCCL) and the packets come back in presentation order.  `devices` may name a device twice (two sessions on one GPU)."""

    def __init__(self, cfg: _lib.Config, devices):
        import queue
        self.cfg, self.devices = cfg, list(devices)
        lanes = cfg.gops_in_flight if cfg.gops_in_flight > 0 else 4
        self.chunk = max(1, lanes * cfg.keyint)
        self._encs = [Encoder(cfg, device=d) for d in self.devices]
        self._q = [queue.Queue(maxsize=2 * self.chunk) for _ in self.devices]
        self._out = {}                      # pts -> (data, key)
        self._lock = threading.Lock()
        self._err = []
        self._threads = [threading.Thread(target=self._worker, args=(k,), daemon=True) for k in range(len(self.devices))]
        for t in self._threads:
            t.start()
        self._n_in, self._next_out = 0, 0
        self._aborted = False

    def _worker(self, k):
        enc, q = self._encs[k], self._q[k]
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                y, u, v, pts = item
                enc.send(y, u, v, pts=pts)          # ctypes releases the GIL: the sessions run concurrently
                self._collect(enc)
            if not self._aborted:
                enc.flush()
                self._collect(enc)
        except Exception as exc:                    # surfaced by send()/finish() on the caller's thread
            self._err.append(exc)

    def _collect(self, enc):
        got = list(enc.packets())
        if got:
            with self._lock:
                for data, pts, key in got:
                    self._out[pts] = (data, key)

    def send(self, y, u, v):
        import queue
        k = (self._n_in // self.chunk) % len(self.devices)
        while True:                                 # a bounded queue whose worker has died must not block the caller for ever
            if self._err:
                raise self._err[0]
            try:
                self._q[k].put((y, u, v, self._n_in), timeout=0.2)
                break
            except queue.Full:
                continue
        self._n_in += 1

    def ready(self):
        """Packets that are next in presentation order: (data, pts, key)."""
        out = []
        with self._lock:
            while self._next_out in self._out:
                data, key = self._out.pop(self._next_out)
                out.append((data, self._next_out, key))
                self._next_out += 1
        return out

    def _stop_workers(self, drain: bool):
        """Hand every worker its sentinel and join it; drain=True first throws away what is still queued (abort)."""
        import queue
        for q, t in zip(self._q, self._threads):
            while t.is_alive():
                if drain:
                    try:
                        while True:
                            q.get_nowait()
                    except queue.Empty:
                        pass
                try:
                    q.put(None, timeout=0.2)
                    break
                except queue.Full:
                    continue
        for t in self._threads:
            t.join()

    def finish(self):
        self._stop_workers(drain=False)
        if self._err:
            raise self._err[0]
        return self.ready()

    def abort(self):
        """Cancel / failure path: no session may be closed while its worker can still be inside a native call."""
        self._aborted = True
        self._stop_workers(drain=True)

    def headers(self) -> bytes:
        return self._encs[0].headers()

    def stats(self):
        return [e.stats() for e in self._encs]

    def close(self):
        if any(t.is_alive() for t in self._threads):
            self.abort()
        for e in self._encs:
            e.close()


def slice_rows(height: int, n: int):
    """CTU rows of n full-width slices of a picture `height` high: as even as whole rows allow, the taller ones first."""
    rows = (height + 31) // 32
    n = max(1, min(n, rows, 16))
    return [rows // n + (1 if k < rows % n else 0) for k in range(n)]


class SlicedEncoder:
    """One PICTURE over several MI355X (BASELINE configs[4]): the picture is cut into full-width bands of CTU rows, band k is coded by its own
    session on devices[k] as one slice (own CABAC stream, slice_segment_address in its header), and the slices of a picture are put together
    into one access unit.  halo=True (default): the sessions exchange rows per picture (include/mihevc.h slice_halo; SURVEY §8e's neighbour halo:
    point-to-point pulls over xGMI, no collective) — the reference rows either side of every seam, so motion vectors cross seams as in a whole
    picture, and the pre-deblock rows + CU records, so deblocking and SAO run across the seams; the rate controller plans the whole picture from
    inputs summed over the bands.  halo=False: nothing is exchanged (motion-constrained slices, filters stop at the seams, one rate controller
    per band: round 2's form, -1.5 dB at 4320p over 8).  `devices` may name a device several times (tests: all bands on one GPU)."""

    _group_ids = iter(range(1, 1 << 30))

    def __init__(self, cfg: _lib.Config, devices, keep_recon: bool = False, halo: bool = True):
        import queue
        self.cfg, self.devices = cfg, list(devices)
        self.rows = slice_rows(cfg.height, len(self.devices))
        self.devices = self.devices[:len(self.rows)]
        self.halo = bool(halo) and len(self.rows) > 1
        group = next(SlicedEncoder._group_ids) if self.halo else 0
        self._bands, self._cfgs, y0 = [], [], 0
        for k, r in enumerate(self.rows):
            c = type(cfg).from_buffer_copy(bytes(cfg))
            y1 = min(cfg.height, y0 + 32 * r)
            c.pic_height, c.slice_count, c.slice_index, c.height = cfg.height, len(self.rows), k, y1 - y0
            for i, rr in enumerate(self.rows):
                c.slice_ctu_rows[i] = rr
            c.rate_share_q16 = max(1, round(65536 * r / sum(self.rows)))
            c.slice_halo, c.slice_group = int(self.halo), group
            self._cfgs.append(c)
            self._bands.append((y0, y1))
            y0 = y1
        self._encs = []
        try:
            for c, d in zip(self._cfgs, self.devices):
                self._encs.append(Encoder(c, device=d, keep_recon=keep_recon))
        except Exception:
            for e in self._encs:
                e.close()
            raise
        self._q = [queue.Queue(maxsize=8) for _ in self._encs]
        self._out = [dict() for _ in self._encs]            # per slice: pts -> (data, key)
        self._lock = threading.Lock()
        self._err = []
        self._aborted = False
        self._threads = [threading.Thread(target=self._worker, args=(k,), daemon=True) for k in range(len(self._encs))]
        for t in self._threads:
            t.start()
        self._n_in, self._next_out = 0, 0

    def _worker(self, k):
        enc, q = self._encs[k], self._q[k]
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                y, u, v, pts = item
                enc.send(y, u, v, pts=pts)
                self._collect(k)
            if not self._aborted:
                enc.flush()
                self._collect(k)
        except Exception as exc:
            self._err.append(exc)
            for e in self._encs:                    # the other bands may be waiting for this one inside a native call: let them go
                e.abort()

    def _collect(self, k):
        got = list(self._encs[k].packets())
        if got:
            with self._lock:
                for data, pts, key in got:
                    self._out[k][pts] = (data, key)

    def send(self, y, u, v):
        import queue
        for k, (y0, y1) in enumerate(self._bands):
            item = (y[y0:y1], u[y0 // 2:y1 // 2], v[y0 // 2:y1 // 2], self._n_in)
            while True:
                if self._err:
                    raise self._err[0]
                try:
                    self._q[k].put(item, timeout=0.2)
                    break
                except queue.Full:
                    continue
        self._n_in += 1

    def ready(self):
        """Access units that are complete (every slice present) and next in order: (data, pts, key)."""
        out = []
        with self._lock:
            while all(self._next_out in o for o in self._out):
                parts = [o.pop(self._next_out) for o in self._out]
                out.append((b"".join(p[0] for p in parts), self._next_out, parts[0][1]))
                self._next_out += 1
        return out

    _stop_workers = ShardedEncoder._stop_workers

    def finish(self):
        self._stop_workers(drain=False)
        if self._err:
            raise self._err[0]
        return self.ready()

    def abort(self):
        self._aborted = True
        for e in self._encs:                        # a band blocked in a native call waiting for its neighbours returns with an error
            e.abort()
        self._stop_workers(drain=True)

    def headers(self) -> bytes:
        return self._encs[0].headers()

    def stats(self):
        return [e.stats() for e in self._encs]

    def recon(self, index: int):
        """the picture's reconstruction: the bands' reconstructions stacked (sessions opened with keep_recon)"""
        parts = [e.recon(index) for e in self._encs]
        return tuple(np.vstack([p[i] for p in parts]) for i in range(3))

    def close(self):
        if any(t.is_alive() for t in self._threads):
            self.abort()
        for e in self._encs:
            e.close()


def remux_audio(video_mp4: Path, source: Path, out_path: Path, info: VideoInfo) -> bool:
    """The native path writes video only; the reference always carries the source's audio as AAC (core/transcoder.py:423-450,480-489).
    When the source has audio (only container inputs can, and those need ffmpeg to be decoded at all) the video track is copied and the
    audio encoded with the reference's own flags.  False when ffmpeg is missing or fails: the caller then falls down the ladder."""
    import shutil
    import subprocess
    from .transcoder import VIDEO_METADATA_FLAGS, get_audio_flags
    if shutil.which('ffmpeg') is None:
        return False
    cmd = ['ffmpeg', '-hide_banner', '-y', '-i', str(video_mp4), '-i', str(source), '-map', '0:v:0', '-map', '1:a:0?', '-map_metadata', '1',
           '-c:v', 'copy', '-tag:v', 'hvc1'] + VIDEO_METADATA_FLAGS
    for kv in ('handler_name=SoundHandler', f'language={info.audio_language or "eng"}', 'title="Main Audio"'):
        cmd += ['-metadata:s:a:0', kv]
    cmd += get_audio_flags(info.audio_channels) + ['-brand', 'mp42', '-movflags', '+write_colr+use_metadata_tags+faststart', str(out_path)]
    try:
        return subprocess.run(cmd, capture_output=True).returncode == 0 and Path(out_path).exists()
    except Exception:
        return False


def encode_file(file_path: Path, out_path: Path, info: VideoInfo, progress_callback: Optional[Callable[[str, int, int], None]] = None,
                total_frames: int = 1, stop_event: Optional[threading.Event] = None, device: Optional[int] = None, debug: bool = False,
                devices=None, row_split: bool = False) -> int:
    """Encode `file_path` to `out_path` (MP4/hvc1) on an MI355X.  Returns 0 on success, 1 on failure/cancel —
    the same (returncode) shape `run_ffmpeg` gives `convert_video` (core/transcoder.py:497-535)."""
    from . import mp4, yuvio
    from .transcoder import calculate_apple_hevc_level, calculate_dynamic_values

    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
    level, tier = calculate_apple_hevc_level(info)
    clip = yuvio.open_any(Path(file_path), info)
    mux = None
    ok = False
    try:
        if clip.bit_depth > 8 and bit_depth_of(info) == 8:      # the file's own header outranks the probe (a y4m tagged C420p10 probed as SDR)
            info.pix_fmt = 'yuv420p10le'
        cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
        total = clip.n_frames or total_frames
        wants_audio = bool(info.audio_channels and info.audio_channels > 0) and Path(file_path).suffix.lower() not in ('.y4m', '.yuv')
        video_path = Path(out_path).with_suffix('.video.mp4') if wants_audio else Path(out_path)
        mux = mp4.Mp4Writer(video_path, cfg)
        n_out = 0

        def progress():
            if progress_callback:
                try:
                    progress_callback(Path(file_path).name, n_out, total)
                except Exception:
                    logger.debug("progress_callback raised", exc_info=True)

        if devices and len(devices) > 1:            # one clip over several GPUs: GOP chunks round-robin, or every picture split by CTU rows
            sh = SlicedEncoder(cfg, devices) if row_split else ShardedEncoder(cfg, devices)
            try:
                for y, u, v in clip.frames():
                    if stop_event is not None and stop_event.is_set():
                        return 1
                    sh.send(y, u, v)
                    for data, pts, key in sh.ready():
                        mux.add_sample(data, pts, key)
                        n_out += 1
                    progress()
                for data, pts, key in sh.finish():
                    mux.add_sample(data, pts, key)
                    n_out += 1
                progress()
                headers = sh.headers()
            finally:
                sh.close()                          # joins the workers (abort) before any session is closed
        else:
            with Encoder(cfg, device=device or 0) as enc:
                for i, (y, u, v) in enumerate(clip.frames()):
                    if stop_event is not None and stop_event.is_set():
                        return 1
                    enc.send(y, u, v, pts=i)
                    for data, pts, key, dts in enc.packets_dts():
                        mux.add_sample(data, pts, key, dts)
                        n_out += 1
                    progress()
                enc.flush()
                for data, pts, key, dts in enc.packets_dts():
                    mux.add_sample(data, pts, key, dts)
                    n_out += 1
                progress()
                headers = enc.headers()
        if n_out == 0:
            return 1
        mux.finish(headers)
        mux = None
        if wants_audio:
            good = remux_audio(video_path, Path(file_path), Path(out_path), info)
            try:
                video_path.unlink()
            except OSError:
                pass
            if not good:
                logger.warning("%s: audio could not be carried over (ffmpeg remux failed); falling back to the ffmpeg path", Path(file_path).name)
                return 1
        ok = True
        return 0
    finally:
        if mux is not None and not ok:
            mux.abort()                             # cancel, failure or exception: no .mdat.tmp and no open handle stay behind
        clip.close()
