"""Source probing: `VideoInfo` + `probe_media`.

Host-side mirror of the reference's L1 probe layer (reference: core/probe.py:9-24 `VideoInfo`,
core/probe.py:47-116 `probe_media`, core/probe.py:31-38 `parse_fps`).  Same field order, same
defaults, same HDR vote and the same "any failure -> 1920x1080@30 SDR stereo" fallback, so callers
written against the reference keep working.  On top of that this module can read the headers of
raw containers natively (`.y4m`, and `.yuv` files whose name carries `_WxH_FPS[_10bit]`), because
the MI355X path and its benchmarks must run on hosts with no ffprobe installed.
"""
from __future__ import annotations

import json
import logging
import re
import subprocess
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

logger = logging.getLogger(__name__)


@dataclass
class VideoInfo:
    # field order is part of the contract (positional construction is used by callers/tests)
    width: int
    height: int
    fps: float
    color_primaries: str
    color_transfer: str
    color_space: str
    pix_fmt: str
    master_display: str
    max_cll: str
    audio_channels: int
    hdr: bool = False
    audio_language: Optional[str] = 'eng'
    nb_frames: Optional[int] = None
    duration: Optional[float] = None


# the four HDR "votes" (core/probe.py:26-29); two or more => HDR (core/probe.py:76-82)
HDR_PIXFMTS = {'yuv420p10le', 'p010le', 'yuv444p10le'}
HDR_COLOR_SPACES = {'bt2020', 'bt2020-ncl', 'bt2020nc'}
HDR_TRANSFERS = {'smpte2084', 'pq'}
HDR_PRIMARIES = {'bt2020', 'bt2020-ncl'}

_FALLBACK = (1920, 1080, 30.0, 'bt709', 'bt709', 'bt709', 'yuv420p', '', '', 2, False, 'eng', None, None)


def parse_fps(rate_str) -> float:
    """'num/den' -> float; anything malformed -> 30.0 (core/probe.py:31-38)."""
    try:
        if not rate_str or '/' not in rate_str:
            return 30.0
        num, den = (int(x) for x in rate_str.split('/'))
        return num / den if den else 30.0
    except Exception:
        return 30.0


def _rate_to_fps(rate: str) -> float:
    rate = (rate or '').strip()
    if not rate or rate == '0/0':
        return 30.0
    try:
        num, den = (int(x) for x in rate.split('/'))
        return num / den if den else 30.0
    except Exception:
        return 30.0


def _first_tag(tags: dict, *names, default=''):
    for n in names:
        if tags.get(n):
            return tags[n]
    return default


def hdr_vote(color_primaries: str, color_transfer: str, color_space: str, pix_fmt: str) -> bool:
    votes = (color_primaries in HDR_PRIMARIES) + (color_transfer in HDR_TRANSFERS) + \
            (color_space in HDR_COLOR_SPACES) + (pix_fmt in HDR_PIXFMTS)
    return votes >= 2


def info_from_ffprobe_json(doc: dict) -> VideoInfo:
    """Build a VideoInfo from `ffprobe -print_format json -show_streams -show_format` output."""
    streams = doc.get('streams', [])
    v = next((s for s in streams if s.get('codec_type') == 'video'), None)
    if v is None:
        raise ValueError('no video stream')
    fmt = doc.get('format', {}) or {}
    tags = fmt.get('tags', {}) or {}
    width = int(v.get('width') or 1920)
    height = int(v.get('height') or 1080)
    fps = _rate_to_fps(v.get('avg_frame_rate') or v.get('r_frame_rate') or '30/1')

    def colour(key, tag_upper):
        return (v.get(key) or tags.get(tag_upper) or tags.get(tag_upper.lower()) or 'bt709').lower()

    prim = colour('color_primaries', 'COLOR_PRIMARIES')
    trc = colour('color_transfer', 'COLOR_TRANSFER')
    spc = colour('color_space', 'COLOR_SPACE')
    pix = (v.get('pix_fmt') or '').lower()
    md = _first_tag(tags, 'master-display', 'MASTER_DISPLAY', 'master_display', 'mastering_display')
    cll = _first_tag(tags, 'max-cll', 'MAX_CLL', 'max_cll')

    a = next((s for s in streams if s.get('codec_type') == 'audio'), None)
    if a is not None:
        atags = a.get('tags', {}) or {}
        lang = atags.get('language') or atags.get('LANGUAGE') or 'eng'
        channels = int(a.get('channels', a.get('CHANNELS', 2)))
    else:
        lang, channels = None, 0

    try:
        nb = int(v['nb_frames']) if v.get('nb_frames') else None
    except Exception:
        nb = None
    try:
        dur = float(fmt['duration']) if fmt.get('duration') else None
    except Exception:
        dur = None
    return VideoInfo(width, height, fps, prim, trc, spc, pix, md, cll, channels,
                     hdr_vote(prim, trc, spc, pix), lang, nb, dur)


# ---------------------------------------------------------------------------------------------
# native raw-container headers (no ffprobe needed) — build-only extension, see module docstring
# ---------------------------------------------------------------------------------------------
_YUV_NAME = re.compile(r'_(\d+)x(\d+)_(\d+(?:\.\d+)?)(?:fps)?(?:_(8|10)bit)?(?:_(hdr|sdr))?', re.I)


def probe_raw(file_path: Path) -> Optional[VideoInfo]:
    """Header-only probe of .y4m / named .yuv; returns None when the file is not a raw container."""
    from .yuvio import open_clip   # local import: yuvio imports VideoInfo lazily too
    suffix = file_path.suffix.lower()
    if suffix not in ('.y4m', '.yuv'):
        return None
    clip = open_clip(file_path)
    try:
        hdr = clip.hdr
        if hdr:
            tags = ('bt2020', 'smpte2084', 'bt2020nc')
        else:
            tags = ('bt709', 'bt709', 'bt709')
        pix = 'yuv420p10le' if clip.bit_depth > 8 else 'yuv420p'
        dur = clip.n_frames / clip.fps if clip.fps else None
        return VideoInfo(clip.width, clip.height, float(clip.fps), *tags, pix, '', '', 0, hdr, None,
                         clip.n_frames, dur)
    finally:
        clip.close()


def probe_media(file_path: Path) -> VideoInfo:
    """ffprobe the file; raw .y4m/.yuv are read natively; any error -> the reference's fallback."""
    file_path = Path(file_path)
    try:
        raw = probe_raw(file_path) if file_path.exists() else None
        if raw is not None:
            return raw
    except Exception as exc:  # fall through to ffprobe / fallback
        logger.debug('native probe failed for %s: %s', file_path.name, exc)
    try:
        out = subprocess.run(
            ['ffprobe', '-v', 'quiet', '-print_format', 'json', '-show_streams', '-show_format', str(file_path)],
            capture_output=True, text=True, check=True, encoding='utf-8')
        return info_from_ffprobe_json(json.loads(out.stdout))
    except Exception as exc:
        logger.error('probe failed: %s, %s', file_path.name, exc)
        return VideoInfo(*_FALLBACK)
