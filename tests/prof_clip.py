"""Short 1080p encode for profiling runs (rocprofv3): 2 GOPs x 12 pictures through the full session."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hevc_amd import _lib                       # noqa: E402
from hevc_amd.encoder import Encoder            # noqa: E402
from hevc_amd.yuvio import SyntheticClip        # noqa: E402

n, keyint = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = _lib.default_config()
cfg.keyint, cfg.min_keyint, cfg.gops_in_flight, cfg.me_range, cfg.profile_stages = keyint, 2, 4, 15, 1
clip = SyntheticClip("motion", 0, 1920, 1080, n)
with Encoder(cfg) as enc:
    nb = 0
    for y, u, v in clip.frames():
        enc.send(y, u, v)
    enc.flush()
    for data, pts, key in enc.packets():
        nb += len(data)
    st = enc.stats()
    print("frames", st.frames_out, "bytes", nb, "psnr", round(enc.psnr_y(), 2), "device_ms", round(st.device_ms, 2))
    for i, name in enumerate(_lib.STAGE_NAMES[:7]):
        if st.stage_launches[i]:
            print(f"  {name:10s} {st.stage_ms[i] / st.stage_pictures[i]:8.4f} ms/picture  ({st.stage_launches[i]} launches)")
