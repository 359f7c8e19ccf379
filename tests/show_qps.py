"""Print per-picture QP / type / bits of the bench clip (rate-control behaviour at the reference's 1080p operating point)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np                                # noqa: E402
from hevc_amd import _lib                         # noqa: E402
from hevc_amd.encoder import Encoder, config_for  # noqa: E402
from hevc_amd.probe import VideoInfo              # noqa: E402
from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values  # noqa: E402
from hevc_amd.yuvio import SyntheticClip          # noqa: E402

W, H, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 300
pattern = sys.argv[2] if len(sys.argv) > 2 else "motion"
info = VideoInfo(W, H, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", N, N / 30.0)
crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info, use_nvenc=False)
level, tier = calculate_apple_hevc_level(info)
cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
clip = SyntheticClip(pattern, 0, W, H, N)
with Encoder(cfg) as enc:
    for y, u, v in clip.frames():
        enc.send(y, u, v)
    enc.flush()
    nb = sum(len(d) for d, _, _ in enc.packets())
    infos = [enc.frame_info(i) for i in range(N)]
    print("kbps", round(nb * 8 / (N / 30.0) / 1e3, 1), "psnr", round(enc.psnr_y(), 3))
for g in range(0, N, gop):
    seg = infos[g:g + gop]
    print(f"GOP@{g}: I qp {seg[0][0]} bits {seg[0][2]}  P qp min/mean/max {min(q for q, _, _ in seg[1:])}/{np.mean([q for q, _, _ in seg[1:]]):.1f}/{max(q for q, _, _ in seg[1:])} "
          f"P bits mean {np.mean([b for _, _, b in seg[1:]]):.0f} gop kbps {sum(b for _, _, b in seg) / (len(seg) / 30.0) / 1e3:.0f}")
print("GOP0 qp:", " ".join(str(q) for q, _, _ in infos[:gop]))
print("GOP0 kbit:", " ".join(str(b // 1000) for _, _, b in infos[:gop]))
