"""Encoder / GPU detection and HDR10 signalling strings.

Mirror of the reference's core/utils.py for this path: `build_hdr_metadata` (core/utils.py:29-70, libx265 branch) and `has_mi355x`,
the selection hook that stands where the reference asks `has_nvenc` (core/utils.py:9-15).  The reference's
hevc_nvenc helpers (`has_nvenc`, `detect_gpu_type`, the NVENC argv) are out of scope and not built.
"""
from __future__ import annotations

import os
import logging
import shutil
import subprocess
from dataclasses import dataclass
from functools import lru_cache
from typing import List, Optional, Tuple

logger = logging.getLogger(__name__)

DEFAULT_MASTER_DISPLAY = 'G(13250,34500)B(7500,3000)R(34000,16000)WP(15635,16450)L(10000000,50)'
DEFAULT_MAX_CLL = '1000,400'


def has_ffmpeg() -> bool:
    return shutil.which('ffmpeg') is not None


def has_libx265() -> bool:
    try:
        out = subprocess.run(['ffmpeg', '-hide_banner', '-encoders'],
                             capture_output=True, text=True, check=True, encoding='utf-8')
    except Exception:
        return False
    return 'libx265' in out.stdout


@lru_cache(maxsize=1)
def mi355x_device_count() -> int:
    """Number of gfx950 devices the native library can open; 0 when the library or the GPU is absent."""
    try:
        from . import _lib
        return max(0, int(_lib.load().mihevc_device_count()))
    except Exception as exc:
        logger.debug('libmihevc unavailable: %s', exc)
        return 0


def has_mi355x() -> bool:
    return mi355x_device_count() > 0


def parse_cpulist(text: str) -> set:
    """'0-3,8,10-11' (the kernel's cpulist format) -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_device_node(device: int, min_cpus: int = 8):
    """One process per GPU: run this process (and every thread it starts from now on: the library's CABAC workers, the pinned-buffer
    allocations' first touch) on the CPUs of the NUMA node next to `device`.  Returns the node, or None when the platform does not name one,
    the node has fewer than `min_cpus` usable CPUs, or anything about it fails — binding is an optimisation, never an error."""
    try:
        from . import _lib
        node = int(_lib.load().mihevc_device_numa_node(device))
        if node < 0:
            return None
        with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
            cpus = parse_cpulist(f.read()) & os.sched_getaffinity(0)
        if len(cpus) < min_cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:       # noqa: BLE001
        return None


def build_hdr_metadata(master_display: str, max_cll: str, use_nvenc: bool = False, fps: float = 30.0) -> List[str]:
    """HDR10 argv fragment for libx265: one `-x265-params` string (core/utils.py:58-69).  Empty inputs fall back to the reference's
    P3-D65 1000-nit defaults.  `use_nvenc` keeps the reference's signature; its hevc_nvenc branch is out of scope and not built."""
    md = (master_display or '').strip() or DEFAULT_MASTER_DISPLAY
    cll = (max_cll or '').strip() or DEFAULT_MAX_CLL
    fields = ['hdr10=1', 'colorprim=bt2020', 'transfer=smpte2084', 'colormatrix=bt2020nc',
              f'master-display={md}', f'max-cll={cll}', 'hrd=1', 'aud=1', 'chromaloc=0', 'repeat-headers=1']
    return ['-x265-params', ':'.join(fields)]


@dataclass
class MasteringDisplay:
    """Parsed `G(x,y)B(x,y)R(x,y)WP(x,y)L(max,min)` in SEI units (0.00002 chroma, 0.0001 cd/m2)."""
    g: Tuple[int, int]
    b: Tuple[int, int]
    r: Tuple[int, int]
    wp: Tuple[int, int]
    lum: Tuple[int, int]


def parse_master_display(text: Optional[str]) -> MasteringDisplay:
    import re
    text = (text or '').strip() or DEFAULT_MASTER_DISPLAY
    m = re.fullmatch(r'G\((\d+),(\d+)\)B\((\d+),(\d+)\)R\((\d+),(\d+)\)WP\((\d+),(\d+)\)L\((\d+),(\d+)\)', text)
    if not m:
        return parse_master_display(DEFAULT_MASTER_DISPLAY)
    v = [int(x) for x in m.groups()]
    return MasteringDisplay((v[0], v[1]), (v[2], v[3]), (v[4], v[5]), (v[6], v[7]), (v[8], v[9]))


def parse_max_cll(text: Optional[str]) -> Tuple[int, int]:
    text = (text or '').strip() or DEFAULT_MAX_CLL
    try:
        a, b = text.split(',')
        return int(a), int(b)
    except Exception:
        return 1000, 400
