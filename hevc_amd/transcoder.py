"""Per-file transcode policy and the `convert_video` boundary.

This is the host-side mirror of the reference's L2 layer (reference: core/transcoder.py).  Every
public name of that module that lies on the hot path keeps its name, argument meaning and error
behaviour here:

    FFmpegParams            core/transcoder.py:25-33
    decide_encoder          core/transcoder.py:70-75      (+ `choose_backend`: the MI355X outcome)
    calculate_apple_hevc_level / HEVC_LEVEL_LIMITS   core/transcoder.py:158-187
    compute_aligned_gop     core/transcoder.py:211-260
    calculate_dynamic_values core/transcoder.py:263-354
    build_ffmpeg_params     core/transcoder.py:357-412
    get_audio_flags         core/transcoder.py:423-450
    build_ffmpeg_command    core/transcoder.py:452-495
    run_ffmpeg              core/transcoder.py:497-535
    convert_video           core/transcoder.py:537-638

The arithmetic is re-derived, not transcribed: tests/test_policy.py checks it against
tests/golden/params.json (values captured by importing the reference).  Known reference quirks
are reproduced on purpose because they define the operating point: the tier test
compares samples/s with a kbps column, and the level bit-rate column is about a quarter of the
spec's MaxBR.

What is new is the third encoder outcome: the native MI355X encoder (`hevc_amd.encoder`), reached
through the C ABI in include/mihevc.h.  `convert_video` tries it first when selected, and falls
back to libx265 exactly as the reference falls back from NVENC (core/transcoder.py:575-599).
"""
from __future__ import annotations

import logging
import math
import subprocess
import threading
from dataclasses import dataclass
from fractions import Fraction
from pathlib import Path
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Tuple

from .probe import VideoInfo, probe_media
from .utils import build_hdr_metadata, has_mi355x

logger = logging.getLogger(__name__)

ProgressCb = Optional[Callable[[str, int, int], None]]

# serialises validator runs across worker threads (the reference's core/ module forgot to define
# it — core/transcoder.py:55 — the monolith has it at apple_hevc_batch.py:45)
validator_lock = threading.Lock()


@dataclass
class FFmpegParams:
    vcodec: str
    pix_fmt: str
    profile: str
    level: str
    color_flags: List[str]
    vparams: List[str]
    hdr_metadata: List[str]


# ---------------------------------------------------------------------------------------------
# level / tier
# ---------------------------------------------------------------------------------------------
class LevelRow(NamedTuple):
    max_luma_ps: int       # samples per picture
    max_luma_sr: int       # samples per second
    max_bitrate_bps: int   # reference's conservative cap (≈ spec MaxBR / 4)
    max_cpb_bits: int
    main_tier_kbps: int
    high_tier_kbps: int


def _row(ps, sr, br, cpb_bytes, mt, ht):
    return LevelRow(ps, sr, br, cpb_bytes * 8, mt, ht)


# same key order and 6-tuples as core/transcoder.py:158-172 (kept as plain tuples for callers that index)
HEVC_LEVEL_LIMITS: Dict[str, LevelRow] = {
    '1':   _row(36864, 552960, 64_000, 4608, 128, 128),
    '2':   _row(122880, 3686400, 150_000, 18432, 1500, 3000),
    '2.1': _row(245760, 7372800, 300_000, 36864, 3000, 6000),
    '3':   _row(552960, 16588800, 600_000, 61440, 6000, 12000),
    '3.1': _row(983040, 33177600, 1_200_000, 122880, 10000, 20000),
    '4':   _row(2228224, 66846720, 3_000_000, 245760, 12000, 30000),
    '4.1': _row(2228224, 133693440, 6_000_000, 491520, 20000, 50000),
    '5':   _row(8912896, 267386880, 12_000_000, 983040, 25000, 100000),
    '5.1': _row(8912896, 534773760, 24_000_000, 1966080, 40000, 160000),
    '5.2': _row(8912896, 1069547520, 48_000_000, 3932160, 60000, 240000),
    '6':   _row(35651584, 1069547520, 48_000_000, 3932160, 60000, 240000),
    '6.1': _row(35651584, 2139095040, 96_000_000, 7864320, 120000, 480000),
    '6.2': _row(35651584, 4278190080, 192_000_000, 15728640, 240000, 800000),
}


def calculate_apple_hevc_level(info: VideoInfo) -> Tuple[str, str]:
    """Lowest level whose picture size and sample rate fit -> (level, tier).

    The tier rule compares samples/s against the *kbps* column (reference behaviour,
    core/transcoder.py:183), so for any real video the answer is 'main'."""
    ps = info.width * info.height
    sr = round(ps * info.fps)
    wants_high = bool(info.hdr) or max(info.width, info.height) >= 3840 or info.fps > 60
    for name, row in HEVC_LEVEL_LIMITS.items():
        if ps <= row.max_luma_ps and sr <= row.max_luma_sr:
            tier = 'high' if (wants_high and sr <= row.high_tier_kbps) else 'main'
            return name, tier
    return '6.2', 'main'


def level_idc(level: str) -> int:
    """'4' -> 120, '3.1' -> 93 (general_level_idc = 30 * level)."""
    return int(round(float(level) * 30))


# ---------------------------------------------------------------------------------------------
# GOP / CRF / VBV
# ---------------------------------------------------------------------------------------------
def compute_aligned_gop(fps: float, preferred_gop_sec: float, max_gop_frames: int = 240) -> int:
    """GOP length in frames, snapped to a whole number of seconds (core/transcoder.py:211-260).

    Among 1..8 s candidates (frame counts from the rational frame rate) take the one closest to
    preferred_gop_sec*fps that fits [2, max]; then re-snap: integer rates to fps*n, fractional
    (NTSC) rates to round(fps * whole_seconds)."""
    fps = max(1.0, fps)
    want = max(2, min(preferred_gop_sec * fps, max_gop_frames))
    try:
        q = Fraction(str(fps)).limit_denominator(1001)
        num, den = q.numerator, q.denominator
    except Exception:
        num, den = int(round(fps)), 1

    pick, pick_err = None, math.inf
    for seconds in range(1, 9):
        frames = round(num * seconds / den)
        if 2 <= frames <= max_gop_frames and abs(frames - want) < pick_err:
            pick, pick_err = frames, abs(frames - want)
    if pick is None:
        pick = max(2, min(int(round(want)), max_gop_frames))

    if abs(round(fps) - fps) < 1e-6:
        whole = int(round(fps))
        pick = max(2, min(whole * max(1, round(pick / whole)), max_gop_frames))
    else:
        secs = max(1, round(pick / fps))
        pick = min(max_gop_frames, max(2, round(fps * secs)))
    return pick


_CRF_BY_HEIGHT = ((480, 17), (720, 18), (1080, 19), (1440, 20), (2160, 21), (4320, 22))
# (min longest side, SDR kbps, HDR kbps)
_TARGET_KBPS = ((7680, 140000, 140000), (3840, 50000, 65000), (2560, 26000, 30000), (1920, 16000, 19000), (0, 8000, 10000))
_DENSE, _SPARSE = 0.00025, 0.00006      # "motion density" = frames / pixels thresholds


def calculate_dynamic_values(info: VideoInfo, use_nvenc: bool = True, gpu_name: str = '') -> Tuple[int, int, int, int, int]:
    """-> (crf, cq, vbv_maxrate_kbps, vbv_bufsize_kbits, gop_frames)  (core/transcoder.py:263-354)."""
    longest = max(info.width, info.height)
    fps = float(info.fps) if info.fps else 30.0
    hdr = bool(info.hdr)

    crf = next((c for h, c in _CRF_BY_HEIGHT if info.height <= h), _CRF_BY_HEIGHT[-1][1])
    if hdr:
        crf = max(8, crf - 1)

    if info.nb_frames:
        frames = info.nb_frames
    elif info.duration:
        frames = int(round(info.duration * fps))
    else:
        frames = int(round(60 * fps))
    density = frames / (info.width * info.height + 1)
    if density > _DENSE:
        crf += 1
    elif density < _SPARSE:
        crf = max(8, crf - 1)
    crf = max(16, min(crf, 24))

    kbps = next((h if hdr else s) for edge, s, h in _TARGET_KBPS if longest >= edge)
    if density > _DENSE:
        kbps = int(kbps * 1.15)
    elif density < _SPARSE:
        kbps = int(kbps * 0.92)
    maxrate = int(kbps)
    bufsize = int(maxrate * 1.5)

    level, _tier = calculate_apple_hevc_level(info)
    row = HEVC_LEVEL_LIMITS.get(str(level))
    if row is not None:
        cap_kbps = int(row.max_bitrate_bps / 1000)
        cap_kbits = int(row.max_cpb_bits / 1000)
        maxrate = min(maxrate, int(cap_kbps * 0.98))
        bufsize = min(bufsize, max(int(maxrate * 1.2), int(cap_kbits * 0.9)))

    seconds = (2.0 if longest >= 3840 else 2.5) if hdr else (2.5 if longest >= 3840 else 3.0)
    if fps > 60:
        seconds *= 1.05
    gop = compute_aligned_gop(fps, seconds, max_gop_frames=240)
    if abs(round(fps) - fps) < 1e-6:
        whole = int(round(fps))
        gop = max(2, min(240, whole * max(1, round(gop / whole))))
    return crf, crf + 1, maxrate, bufsize, gop


def build_ffmpeg_params(info: VideoInfo, use_nvenc: bool = False, gpu_name: str = '') -> FFmpegParams:
    """libx265 operating point for one file (core/transcoder.py:357-412, CPU branch).  The reference's hevc_nvenc branch is out of
    scope here: `use_nvenc` / `gpu_name` keep the reference's positional signature and are ignored."""
    hdr = bool(info.hdr)
    level, tier = calculate_apple_hevc_level(info)
    profile, pix_fmt = ('main10', 'p010le') if hdr else ('main', 'yuv420p')
    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
    fields = [f'crf={crf}', 'preset=slow', 'log-level=error', 'nal-hrd=vbr', f'vbv-maxrate={maxrate}',
              f'vbv-bufsize={bufsize}', f'tier={tier}', f'keyint={gop}', f'min-keyint={max(2, gop // 2)}',
              f'profile={profile}', f'level-idc={level}']
    if hdr:
        fields += build_hdr_metadata(info.master_display, info.max_cll, use_nvenc=False, fps=info.fps)[1].split(':')
    return FFmpegParams('libx265', pix_fmt, profile, level, [], ['-x265-params', ':'.join(fields), '-threads', '0'], [])


# ---------------------------------------------------------------------------------------------
# argv assembly
# ---------------------------------------------------------------------------------------------
VIDEO_METADATA_FLAGS = ['-metadata:s:v:0', 'handler_name=VideoHandler']
_LAYOUTS = {1: 'mono', 2: 'stereo', 6: '5.1', 8: '7.1'}


def get_audio_flags(audio_channels: int) -> List[str]:
    """AAC flags: 64 kb/s per channel, floor 128k, ceiling 512k, >=256k beyond stereo."""
    if not audio_channels or audio_channels < 1:
        return []
    kbps = min(512, max(128, 64 * audio_channels))
    if audio_channels > 2:
        kbps = max(kbps, 256)
    flags = ['-c:a', 'aac', '-b:a', f'{kbps}k', '-ar', '48000', '-ac', str(max(1, audio_channels))]
    if audio_channels in _LAYOUTS:
        flags += ['-channel_layout', _LAYOUTS[audio_channels]]
    return flags


def build_ffmpeg_command(file_path: Path, out_path: Path, ff_params: FFmpegParams, audio_channels: int,
                         audio_language: Optional[str] = 'eng', extra_vparams: Optional[List[str]] = None) -> List[str]:
    cmd = ['ffmpeg', '-hide_banner', '-y', '-i', str(file_path), '-map_metadata', '0',
           '-c:v', ff_params.vcodec, '-pix_fmt', ff_params.pix_fmt, '-profile:v', ff_params.profile, '-tag:v', 'hvc1']
    cmd += ff_params.hdr_metadata or []
    cmd += extra_vparams if extra_vparams else ff_params.vparams
    cmd += VIDEO_METADATA_FLAGS
    if audio_channels and audio_channels > 0:
        for kv in ('handler_name=SoundHandler', f'language={audio_language or "eng"}', 'title="Main Audio"'):
            cmd += ['-metadata:s:a:0', kv]
        cmd += get_audio_flags(audio_channels)
    cmd += ['-color_range', 'tv', '-brand', 'mp42', '-movflags', '+write_colr+use_metadata_tags+faststart', str(out_path)]
    return cmd


# ---------------------------------------------------------------------------------------------
# process boundary
# ---------------------------------------------------------------------------------------------
def run_ffmpeg(cmd: List[str], progress_callback: ProgressCb, file_name: str, total_frames: int,
               stop_event: Optional[threading.Event] = None, debug: bool = False) -> Tuple[int, str]:
    """Run `cmd`, forward `frame=` progress, honour stop_event; -> (returncode, combined output).
    Spawn failures return (1, str(error)); callback exceptions are swallowed (core/transcoder.py:497-535)."""
    if debug:
        logger.debug('ffmpeg: %s', ' '.join(cmd))
    captured: List[str] = []
    try:
        with subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              encoding='utf-8', errors='replace') as proc:
            frame = 0
            for line in proc.stdout:
                captured.append(line)
                if stop_event is not None and stop_event.is_set():
                    try:
                        proc.terminate()
                    except Exception:
                        pass
                    return 1, ''.join(captured)
                if 'frame=' not in line:
                    continue
                try:
                    frame = int(line.strip().split('frame=')[-1].split()[0])
                except Exception:
                    pass                      # keep the last good counter, still tick the callback
                if progress_callback:
                    _safe_progress(progress_callback, file_name, frame, total_frames)
            return proc.wait(), ''.join(captured)
    except Exception as exc:
        logger.error('ffmpeg failed to run: %s — %s', cmd[:3], exc)
        return 1, str(exc)


def _safe_progress(cb, name, frame, total):
    try:
        cb(name, frame, total)
    except Exception:
        logger.debug('progress_callback raised', exc_info=True)


def detect_validator_path() -> Optional[Path]:
    for p in ('/Applications/Apple Video Tools/AppleHEVCValidator', '/usr/local/bin/AppleHEVCValidator',
              '/usr/bin/AppleHEVCValidator', '/opt/homebrew/bin/AppleHEVCValidator',
              'C:/Program Files/Apple/AppleHEVCValidator.exe'):
        if Path(p).exists():
            return Path(p)
    return None


def run_apple_validator(file_path: Path, refresh_cache=False) -> bool:
    """True = passed or validator absent (core/transcoder.py:46-68)."""
    validator = detect_validator_path()
    if validator is None:
        logger.warning('AppleHEVCValidator not installed; output compatibility unverified')
        return True
    with validator_lock:
        try:
            subprocess.run([str(validator), str(file_path)], check=True, capture_output=True, text=True, encoding='utf-8')
            return True
        except subprocess.CalledProcessError as exc:
            logger.warning('AppleHEVCValidator rejected %s: %s', file_path.name, (exc.stderr or '')[:2000])
            return False
        except Exception as exc:
            logger.error('AppleHEVCValidator error: %s', exc)
            return False


# ---------------------------------------------------------------------------------------------
# encoder choice
# ---------------------------------------------------------------------------------------------
def choose_backend(info: Optional[VideoInfo], force_cpu: bool, force_gpu: bool) -> str:
    """'CPU' | 'MI355X'.  force_cpu wins; otherwise a present MI355X is used, else the CPU (libx265 through ffmpeg).  force_gpu with
    no GPU encoder silently yields 'CPU', as in the reference (core/transcoder.py:70-75)."""
    if force_cpu:
        return 'CPU'
    return 'MI355X' if has_mi355x() else 'CPU'


def decide_encoder(info: Optional[VideoInfo], force_cpu: bool, force_gpu: bool) -> bool:
    """Reference signature: True = a GPU encoder will be used."""
    return choose_backend(info, force_cpu, force_gpu) != 'CPU'


# ---------------------------------------------------------------------------------------------
# the boundary
# ---------------------------------------------------------------------------------------------
def convert_video(file_path: Path, out_dir: Path, progress_callback: ProgressCb = None, debug: bool = False,
                  skip_validator: bool = False, force_cpu: bool = False, force_gpu: bool = False,
                  stop_event: Optional[threading.Event] = None, device: Optional[int] = None,
                  devices: Optional[list] = None, row_split: bool = False) -> Dict[str, Any]:
    """Transcode one file to `out_dir/<stem>.mp4`; never raises for encode failures.

    Returns {"file","status","quality","retries","method","hdr"} — the six CSV columns of the
    reference (gui/mainwindow.py:351).  `method` is the path that produced the file: 'MI355X'
    or 'CPU'.  `device` (new, optional) pins the MI355X ordinal for the batch scheduler; `devices`
    (new, optional) spreads ONE clip over several MI355X: GOP chunks round-robin (hevc_amd.encoder.ShardedEncoder) or, with
    `row_split`, every picture as one slice of CTU rows per device (hevc_amd.encoder.SlicedEncoder; BASELINE configs[4])."""
    file_path, out_dir = Path(file_path), Path(out_dir)
    info = probe_media(file_path)
    out_path = out_dir / (file_path.stem + '.mp4')
    backend = choose_backend(info, force_cpu, force_gpu)
    result: Dict[str, Any] = {'file': file_path.name, 'status': 'FAILED', 'quality': None, 'retries': 0,
                              'method': backend, 'hdr': info.hdr}
    crf, _cq, _, _, _ = calculate_dynamic_values(info)
    total_frames = max(1, int(info.duration * info.fps)) if info.duration and info.fps else 1

    if backend == 'MI355X':
        from . import encoder as native
        try:
            rc = native.encode_file(file_path, out_path, info, progress_callback=progress_callback,
                                    total_frames=total_frames, stop_event=stop_event, device=device, debug=debug, devices=devices, row_split=row_split)
        except Exception as exc:            # the native path must never take the caller down
            logger.warning('MI355X encode failed for %s: %s', file_path.name, exc, exc_info=debug)
            rc = 1
        if rc == 0:
            result.update(status='SUCCESS', quality=crf, retries=0, method='MI355X')
        elif not (stop_event is not None and stop_event.is_set()):
            backend = 'CPU'                                   # fall down the ladder: libx265 through ffmpeg, as the reference's NVENC ladder does

    if backend == 'CPU' and result['status'] != 'SUCCESS':
        ff = build_ffmpeg_params(info)
        cmd = build_ffmpeg_command(file_path, out_path, ff, info.audio_channels, info.audio_language)
        rc, text = run_ffmpeg(cmd, progress_callback, file_path.name, total_frames, stop_event=stop_event, debug=debug)
        if rc == 0:
            result.update(status='SUCCESS', quality=crf, retries=0, method='CPU')
        else:
            result['method'] = 'CPU'
            logger.error('CPU transcode failed: %s\n%s', file_path.name, text[:2000])

    if result['status'] == 'SUCCESS' and not skip_validator:
        try:
            run_apple_validator(out_path)
        except Exception:
            logger.debug('validator raised', exc_info=True)
    if stop_event is not None and stop_event.is_set() and result['status'] != 'SUCCESS':
        result['status'] = 'CANCELLED'
    if progress_callback:
        _safe_progress(progress_callback, file_path.name, total_frames, total_frames)
    return result
