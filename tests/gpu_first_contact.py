"""First-contact script for a GPU box: environment probe + per-stage timings at 1080p (not a test)."""
import ctypes as C
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hevc_amd import _lib          # noqa: E402
from tests import util             # noqa: E402

print("cpus", os.cpu_count(), "ffmpeg", shutil.which("ffmpeg"), "ffprobe", shutil.which("ffprobe"), "x265", shutil.which("x265"))
try:
    print(subprocess.run(["bash", "-c", "lscpu | grep -E 'Model name|^CPU\\(s\\)' ; free -g | head -2"], capture_output=True, text=True).stdout)
except Exception as e:
    print(e)
L = _lib.load()
print("gfx950 devices:", L.mihevc_device_count())
w, h = 1920, 1088
api = util.StageApi(L, "mihevc_k_", device=0)
cp = _lib.cost_params(24, 8, 16)
f0 = util.synth_frame(h, w, 1, detail=True)
f1 = util.synth_frame(h, w, 1, shift=(3, 1))
for name, fn in (("intra", lambda: api.intra(f0, cp)),):
    t = time.time(); a = fn(); print(name, "1080p incl. copies: %.1f ms" % ((time.time() - t) * 1e3))
t = time.time(); a = api.intra(f0, cp); print("intra 2nd: %.1f ms" % ((time.time() - t) * 1e3))
d = api.deblock(a.rec, a.cu, 8)
fin, sp = api.sao(f0, d, cp)
for R in (16, 32):
    cpr = _lib.cost_params(24, 8, R)
    t = time.time(); p = api.inter(f1, fin, cpr); print("inter R=%d incl. copies: %.1f ms" % (R, (time.time() - t) * 1e3))
print("P cu sizes", np.bincount(p.cu["log2_size"].ravel(), minlength=6)[3:], "median mv", np.median(p.cu["mvx"]), np.median(p.cu["mvy"]))
print("psnr I %.2f" % util.psnr(fin.y, f0.y))
