"""Measurements for DESIGN.md that are NOT the bench line: BASELINE configs[2] (2160p30 HDR10 10-bit) and the
PCIe-inclusive 1080p rate (frames handed over as host buffers).  Run on a GPU box; prints one JSON object."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hevc_amd import _lib                                   # noqa: E402
from hevc_amd.encoder import Encoder, config_for             # noqa: E402
from hevc_amd.probe import VideoInfo                         # noqa: E402
from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values   # noqa: E402
from hevc_amd.yuvio import SyntheticClip                     # noqa: E402


def run(w, h, n, hdr, repeats=2):
    tags = ("bt2020", "smpte2084", "bt2020nc", "yuv420p10le") if hdr else ("bt709", "bt709", "bt709", "yuv420p")
    info = VideoInfo(w, h, 30.0, *tags, "", "", 0, hdr, "eng", n, n / 30.0)
    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
    level, tier = calculate_apple_hevc_level(info)
    cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
    cfg.profile_stages = 1
    frames = list(SyntheticClip("motion", 0, w, h, n, bit_depth=10 if hdr else 8).frames())
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        with Encoder(cfg) as enc:
            nbytes = 0
            for y, u, v in frames:
                enc.send(y, u, v)
                for d, _p, _k in enc.packets():
                    nbytes += len(d)
            enc.flush()
            for d, _p, _k in enc.packets():
                nbytes += len(d)
            st = enc.stats()
            psnr = enc.psnr_y()
        dt = time.perf_counter() - t0
        rec = {"size": f"{w}x{h}", "bit_depth": cfg.bit_depth, "frames": n, "keyint": gop, "fps_pcie_inclusive": round(n / dt, 1),
               "device_ms_per_frame": round(st.device_ms / n, 3), "bitrate_kbps": round(nbytes * 8 / (n / 30.0) / 1e3, 1), "target_kbps": maxrate,
               "psnr_y_db": round(psnr, 2), "stages_ms_per_picture": {_lib.STAGE_NAMES[i]: round(st.stage_ms[i] / max(1, st.stage_pictures[i]), 4) for i in range(7)}}
        if best is None or rec["fps_pcie_inclusive"] > best["fps_pcie_inclusive"]:
            best = rec
    return best


def run_resident(w, h, n, hdr):
    """the same session with the frames already in HBM (mihevc_send_frame_device, as bench.py's headline does): what the device sustains without the upload"""
    import torch
    tags = ("bt2020", "smpte2084", "bt2020nc", "yuv420p10le") if hdr else ("bt709", "bt709", "bt709", "yuv420p")
    info = VideoInfo(w, h, 30.0, *tags, "", "", 0, hdr, "eng", n, n / 30.0)
    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
    level, tier = calculate_apple_hevc_level(info)
    cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
    dev = [[torch.from_numpy(p).cuda() for p in f] for f in SyntheticClip("motion", 0, w, h, n, bit_depth=10 if hdr else 8).frames()]
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        with Encoder(cfg) as enc:
            for i, (y, u, v) in enumerate(dev):
                enc.send_device(y.data_ptr(), u.data_ptr(), v.data_ptr(), w, w // 2, pts=i)
                for _d in enc.packets():
                    pass
            enc.flush()
            for _d in enc.packets():
                pass
        best = max(best, n / (time.perf_counter() - t0))
    return round(best, 1)


def run_sliced(w, h, n, n_slices):
    """BASELINE configs[4] geometry with every slice's session on device 0: functional and a one-GPU time, NOT a scaling number"""
    from hevc_amd.encoder import SlicedEncoder
    info = VideoInfo(w, h, 30.0, "bt2020", "smpte2084", "bt2020nc", "yuv420p10le", "", "", 0, True, "eng", n, n / 30.0)
    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info)
    level, tier = calculate_apple_hevc_level(info)
    cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
    frames = list(SyntheticClip("motion", 0, w, h, n, bit_depth=10).frames())
    t0 = time.perf_counter()
    sl = SlicedEncoder(cfg, [0] * n_slices)
    try:
        nbytes = 0
        for y, u, v in frames:
            sl.send(y, u, v)
            nbytes += sum(len(d) for d, _p, _k in sl.ready())
        nbytes += sum(len(d) for d, _p, _k in sl.finish())
        sse = sum(st.sse_y for st in sl.stats())
    finally:
        sl.close()
    dt = time.perf_counter() - t0
    psnr = 10 * np.log10(1023.0 ** 2 / (sse / (n * w * h)))
    return {"size": f"{w}x{h}", "slices": sl.rows, "frames": n, "fps_pcie_inclusive_all_slices_on_one_gpu": round(n / dt, 1),
            "bitrate_kbps": round(nbytes * 8 / (n / 30.0) / 1e3, 1), "target_kbps": maxrate, "psnr_y_db": round(float(psnr), 2)}


try:                       # torch's HIP runtime has to come up before libmihevc's
    import torch
    torch.cuda.init()
except Exception:          # noqa: BLE001
    torch = None
out = {"720p8_host_buffers": run(1280, 720, 300, False), "1080p8_host_buffers": run(1920, 1080, 300, False),
       "2160p10_hdr10_host_buffers": run(3840, 2160, 120, True)}
if torch is not None:
    out["720p8_host_buffers"]["fps_hbm_resident"] = run_resident(1280, 720, 300, False)
    out["2160p10_hdr10_host_buffers"]["fps_hbm_resident"] = run_resident(3840, 2160, 120, True)
if len(sys.argv) > 1 and sys.argv[1] == "8k":
    out["4320p10_hdr10_host_buffers_one_gpu"] = run(7680, 4320, 60, True, repeats=1)
    if torch is not None:
        out["4320p10_hdr10_host_buffers_one_gpu"]["fps_hbm_resident"] = run_resident(7680, 4320, 60, True)
    out["4320p10_hdr10_8_slices_on_ONE_gpu"] = run_sliced(7680, 4320, 30, 8)
print(json.dumps(out))
