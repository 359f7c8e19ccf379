"""BASELINE configs[4] geometry with every band's session on device 0: `python tests/measure_sliced.py W H N n_slices halo(0|1) [hdr=1]` -> one JSON line
(fps with host buffers, bitrate, PSNR-Y).  A functional / quality measurement on ONE GPU, not a scaling number."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch                                                 # noqa: E402
torch.cuda.init()
from hevc_amd.encoder import SlicedEncoder, config_for       # noqa: E402
from hevc_amd.probe import VideoInfo                         # noqa: E402
from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values   # noqa: E402
from hevc_amd.yuvio import SyntheticClip                     # noqa: E402

w, h, n, n_slices, halo = (int(x) for x in sys.argv[1:6])
hdr = True
tags = ("bt2020", "smpte2084", "bt2020nc", "yuv420p10le") if hdr else ("bt709", "bt709", "bt709", "yuv420p")
info = VideoInfo(w, h, 30.0, *tags, "", "", 0, hdr, "eng", n, n / 30.0)
crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
level, tier = calculate_apple_hevc_level(info)
cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
frames = list(SyntheticClip("motion", 0, w, h, n, bit_depth=10).frames())
t0 = time.perf_counter()
sl = SlicedEncoder(cfg, [0] * n_slices, halo=bool(halo))
try:
    nbytes = 0
    for y, u, v in frames:
        sl.send(y, u, v)
        nbytes += sum(len(d) for d, _p, _k in sl.ready())
    nbytes += sum(len(d) for d, _p, _k in sl.finish())
    sse = sum(st.sse_y for st in sl.stats())
    qps = [sl._encs[0].frame_info(i)[0] for i in range(n)]
finally:
    sl.close()
dt = time.perf_counter() - t0
psnr = 10 * np.log10(1023.0 ** 2 / (sse / (n * w * h)))
print(json.dumps({"size": f"{w}x{h}", "slices": sl.rows, "halo": bool(halo), "frames": n, "fps_host_buffers_all_bands_on_one_gpu": round(n / dt, 1),
                  "bitrate_kbps": round(nbytes * 8 / (n / 30.0) / 1e3, 1), "target_kbps": maxrate, "psnr_y_db": round(float(psnr), 3), "qp_first_last": [qps[0], qps[1], qps[-1]]}))
