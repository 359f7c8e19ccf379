"""One configuration, frames resident in HBM, with mihevc_config overrides: `python tests/measure_one.py W H N hdr [field=int ...]` -> one JSON line
(fps, bitrate, PSNR, device / CABAC time).  For A/B runs of a knob on a GPU box (not a test, not the bench line)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch                                                 # noqa: E402
torch.cuda.init()                                            # torch's HIP runtime before libmihevc's
from hevc_amd.encoder import Encoder, config_for             # noqa: E402
from hevc_amd.probe import VideoInfo                         # noqa: E402
from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values   # noqa: E402
from hevc_amd.yuvio import SyntheticClip                     # noqa: E402

w, h, n, hdr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sets = dict(kv.split("=") for kv in sys.argv[5:])
pattern = sets.pop("pattern", "motion")
repeats = int(sets.pop("repeats", 3))
tags = ("bt2020", "smpte2084", "bt2020nc", "yuv420p10le") if hdr else ("bt709", "bt709", "bt709", "yuv420p")
info = VideoInfo(w, h, 30.0, *tags, "", "", 0, bool(hdr), "eng", n, n / 30.0)
crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
level, tier = calculate_apple_hevc_level(info)
cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
for k, v in sets.items():
    setattr(cfg, k, int(v))
dev = [[torch.from_numpy(p).cuda() for p in f] for f in SyntheticClip(pattern, 0, w, h, n, bit_depth=10 if hdr else 8).frames()]
torch.cuda.synchronize()
best = None
for _ in range(repeats):
    t0 = time.perf_counter()
    nbytes = 0
    with Encoder(cfg) as enc:
        for i, (y, u, v) in enumerate(dev):
            enc.send_device(y.data_ptr(), u.data_ptr(), v.data_ptr(), w, w // 2, pts=i)
            nbytes += sum(len(d[0]) for d in enc.packets())
        enc.flush()
        nbytes += sum(len(d[0]) for d in enc.packets())
        dt = time.perf_counter() - t0
        st, psnr = enc.stats(), enc.psnr_y()
    rec = {"size": f"{w}x{h}", "frames": n, "set": sets, "fps_hbm_resident": round(n / dt, 1), "device_ms_per_frame": round(st.device_ms / n, 3),
           "entropy_ms_per_frame_sum_over_threads": round(st.entropy_ms / n, 3), "bitrate_kbps": round(nbytes * 8 / (n / 30.0) / 1e3, 1), "target_kbps": maxrate, "psnr_y_db": round(psnr, 3)}
    if best is None or rec["fps_hbm_resident"] > best["fps_hbm_resident"]:
        best = rec
print(json.dumps(best))
