"""One clip, frames resident in HBM, coded by K sessions at once on ONE device: session k takes the k-th run of whole GOPs (`python tests/measure_split.py W H N K [field=int ...]`).
Closed GOPs are independent, so the clip's stream is the sessions' streams one after the other.  For A/B runs on a GPU box (not a test, not the bench line)."""
import json
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch                                                 # noqa: E402
torch.cuda.init()
from hevc_amd.encoder import Encoder, config_for             # noqa: E402
from hevc_amd.probe import VideoInfo                         # noqa: E402
from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values   # noqa: E402
from hevc_amd.yuvio import SyntheticClip                     # noqa: E402

w, h, n, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sets = dict(kv.split("=") for kv in sys.argv[5:])
repeats = int(sets.pop("repeats", 4))
info = VideoInfo(w, h, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", n, n / 30.0)
crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
level, tier = calculate_apple_hevc_level(info)
dev = [[torch.from_numpy(p).cuda() for p in f] for f in SyntheticClip("motion", 0, w, h, n).frames()]
torch.cuda.synchronize()
n_gops = -(-n // gop)
per = -(-n_gops // K)                      # GOPs per session
glen = -(-n // n_gops)                     # pictures per GOP with equal GOPs (cfg.gop_balance)
cuts = [min(n, k * per * glen) for k in range(K + 1)]
best = None
for _ in range(repeats):
    res = [None] * K

    def run(k):
        a, b = cuts[k], cuts[k + 1]
        sub = VideoInfo(w, h, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", b - a, (b - a) / 30.0)
        cfg = config_for(sub, crf, maxrate, bufsize, gop, level, tier)
        for kk, v in sets.items():
            setattr(cfg, kk, int(v))
        nbytes = 0
        with Encoder(cfg) as enc:
            for i in range(a, b):
                y, u, v = dev[i]
                enc.send_device(y.data_ptr(), u.data_ptr(), v.data_ptr(), w, w // 2, pts=i)
                nbytes += sum(len(d[0]) for d in enc.packets())
            enc.flush()
            nbytes += sum(len(d[0]) for d in enc.packets())
            st = enc.stats()
            res[k] = (nbytes, st.sse_y, st.device_ms, cfg.gops_in_flight)
    th = [threading.Thread(target=run, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    import math
    sse = sum(r[1] for r in res)
    rec = {"size": f"{w}x{h}", "frames": n, "sessions": K, "cuts": cuts, "lanes": [r[3] for r in res], "fps_hbm_resident": round(n / dt, 1),
           "bitrate_kbps": round(sum(r[0] for r in res) * 8 / (n / 30.0) / 1e3, 1), "psnr_y_db": round(10 * math.log10(255.0 ** 2 / (sse / (n * w * ((h + 7) & ~7)))), 3),
           "device_ms": [round(r[2], 2) for r in res]}
    if best is None or rec["fps_hbm_resident"] > best["fps_hbm_resident"]:
        best = rec
print(json.dumps(best))
