"""One 1080p clip over several sessions (hevc_amd.encoder.ShardedEncoder): GOP chunks round-robin, packets merged in order.
Usage: python tests/shard_bench.py <frames> <dev,dev,...>   e.g. 1080 0,0 (two sessions on GPU 0) or 1440 0,1,2,3 on a node."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hevc_amd.encoder import Encoder, ShardedEncoder, config_for          # noqa: E402
from hevc_amd.probe import VideoInfo                                      # noqa: E402
from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values   # noqa: E402
from hevc_amd.yuvio import SyntheticClip                                  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
devs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,0").split(",")]
W, H = 1920, 1080
info = VideoInfo(W, H, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", N, N / 30.0)
crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info, use_nvenc=False)
level, tier = calculate_apple_hevc_level(info)
cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
frames = list(SyntheticClip("motion", 0, W, H, N).frames())
for label, make in (("single session", lambda: None), (f"sharded over {devs}", lambda: ShardedEncoder(cfg, devs))):
    for rep in range(2):
        t0 = time.perf_counter()
        sh = make()
        nb = 0
        if sh is None:
            with Encoder(cfg, device=devs[0]) as enc:
                for i, (y, u, v) in enumerate(frames):
                    enc.send(y, u, v, pts=i)
                    nb += sum(len(d) for d, _, _ in enc.packets())
                enc.flush()
                nb += sum(len(d) for d, _, _ in enc.packets())
        else:
            try:
                for y, u, v in frames:
                    sh.send(y, u, v)
                    nb += sum(len(d) for d, _, _ in sh.ready())
                nb += sum(len(d) for d, _, _ in sh.finish())
            finally:
                sh.close()
        dt = time.perf_counter() - t0
    print(f"{label}: {N / dt:.1f} fps (host buffers), {nb * 8 / (N / 30.0) / 1e3:.1f} kb/s")
