#!/usr/bin/env python3
"""Per-phase cycle table of the CTU programs (diagnostic build libmihevc_prof.so: `make -C hevc_amd/csrc prof`).

  MIHEVC_LIBRARY=hevc_amd/libmihevc_prof.so python tools/phase_profile.py [frames keyint]

Runs the short 1080p session of tests/prof_clip.py, then prints for every ex.phase() call site (header, line): calls, mean cycles of
the whole phase as wave 0 sees it (work + barrier wait), mean cycles of wave 0's own work, and the share of the sum.  The stamps cost
time and forbid overlaps the real kernels have: read the SHARES, never the length (cdna_hip_programming.md §7, In-kernel stamps)."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("MIHEVC_LIBRARY", str(ROOT / "hevc_amd" / "libmihevc_prof.so"))
from hevc_amd import _lib                       # noqa: E402
from hevc_amd.encoder import Encoder            # noqa: E402
from hevc_amd.yuvio import SyntheticClip        # noqa: E402

FILES = {6: "inter.h", 7: "intra.h", 2: "residual.h", 0: "loopfilter.h", 5: "device.hip"}

n, keyint = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
lib = _lib.load()
buf = (C.c_ulonglong * (8 * 1024 * 3))()
lib.mihevc_debug_phase_profile.argtypes = [C.c_void_p, C.c_int]
cfg = _lib.default_config()
cfg.keyint, cfg.min_keyint, cfg.gops_in_flight, cfg.me_range = keyint, 2, 4, 15
for k, v in (a.split("=") for a in sys.argv[3:]):
    setattr(cfg, k, int(v))
clip = SyntheticClip("motion", 0, 1920, 1080, n)
frames = list(clip.frames())
for rep in range(2):            # the first pass warms caches and code objects; the table is cleared before the second
    assert lib.mihevc_debug_phase_profile(None, 1) == 0
    with Encoder(cfg) as enc:
        for y, u, v in frames:
            enc.send(y, u, v)
        enc.flush()
        list(enc.packets())
assert lib.mihevc_debug_phase_profile(buf, 0) == 0
rows = []
for fid, name in FILES.items():
    src = (ROOT / "hevc_amd" / "csrc" / ("kernels/" + name if name.endswith(".h") else name)).read_text().splitlines()
    for line in range(1024):
        tot, work, calls = buf[(fid * 1024 + line) * 3], buf[(fid * 1024 + line) * 3 + 1], buf[(fid * 1024 + line) * 3 + 2]
        if calls:
            rows.append((tot, work, calls, name, line, src[line - 1].strip()[:70] if 0 < line <= len(src) else ""))
total = sum(r[0] for r in rows)
print(f"{'header:line':18s} {'calls':>9s} {'cyc/phase':>10s} {'wave0 work':>10s} {'share':>7s}  source")
for tot, work, calls, name, line, text in sorted(rows, reverse=True):
    print(f"{name + ':' + str(line):18s} {calls:9d} {tot / calls:10.0f} {work / calls:10.0f} {100.0 * tot / total:6.2f}%  {text}")
