#!/usr/bin/env python3
"""bench.py — encoded 1080p30 frames/s on MI355X (BASELINE.json metric), one process per GPU.

A step = one complete encode of the workload clip: BASELINE configs[1], "1080p30 SDR 8-bit Main profile, CQ mode":
300 frames (10 s) of the synthetic `motion` clip, seed = rank, already resident in HBM when the
timed region starts.  Timed: every device stage + D2H of the symbols + host CABAC until the last NAL byte exists.
N > 1 shards one clip per GPU with no data-path collective (BASELINE configs[3]); scaling is weak.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(width, height, qp, me_range, budget_frames=3):
    """The oracle (scalar C port of the same path) on ONE host core, on the first frames of the same clip:
    1 I + (budget_frames-1) P pictures through analysis + deblock + SAO.  Reported baseline, not the target."""
    from hevc_amd.yuvio import SyntheticClip
    from oracle import oracle as O
    clip = SyntheticClip("motion", 0, width, height, budget_frames)
    ch = (height + 7) & ~7
    prm_i, prm_p = O.default_params(max(0, qp - 3), me_range=me_range), O.default_params(qp, me_range=me_range)
    from hevc_amd import _lib
    cfg = _lib.default_config()
    cfg.width, cfg.height = width, height
    prm_i.tile_cols, prm_i.tile_rows = _lib.tile_grid(cfg)        # the same IDR tile grid and NxN trial the device path runs
    prm_i.intra_nxn, prm_i.chroma_modes = cfg.intra_nxn, cfg.chroma_modes
    prm_p.intra_in_p, prm_p.pre_search, prm_p.rdo_zero = cfg.intra_in_p, cfg.pre_search, cfg.rdo_zero
    t0 = time.perf_counter()
    ref = None
    for i, (y, u, v) in enumerate(clip.frames()):
        f = O.Frame(np.pad(y, ((0, ch - height), (0, 0)), mode="edge"), np.pad(u, ((0, (ch - height) // 2), (0, 0)), mode="edge"),
                    np.pad(v, ((0, (ch - height) // 2), (0, 0)), mode="edge"))
        a = O.analyze_intra(f, prm_i) if i == 0 else O.analyze_inter(f, ref, prm_p)
        ref, _ = O.sao(f, O.deblock(a.rec, a.cu, 8), prm_i if i == 0 else prm_p)
    dt = time.perf_counter() - t0
    return {"value": round(budget_frames / dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"oracle/hevc_oracle.c, first {budget_frames} pictures (1 I + {budget_frames - 1} P) of the same {width}x{height} clip, "
                      f"analysis+deblock+SAO, no CABAC, {dt:.1f} s; libx265 itself is unavailable (no ffmpeg on this host)"}


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (profiles/r*/traffic.json, written by
    tools/profile_bench.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  PMC passes cannot
    run inside the timed bench, so the figure is only reported for the default workload it was collected on."""
    if workload != (1920, 1080, 300, 15):
        return None, None
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*", "traffic.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None, None
    return (k["hbm_bytes_per_launch"], os.path.relpath(files[-1], here)) if k else (None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--me-range", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0, help="CABAC worker threads (0 = library default)")
    ap.add_argument("--qp", type=int, default=-1, help="experiments only: force the P-picture QP instead of deriving it from the CRF")
    ap.add_argument("--intra-nxn", type=int, default=None, help="experiments only: override cfg.intra_nxn (4x4 PUs + DST in IDR pictures)")
    ap.add_argument("--intra-tiles", type=int, default=None, help="experiments only: override cfg.intra_tiles (IDR tile grid)")
    ap.add_argument("--pre-search", type=int, default=None, help="experiments only: override cfg.pre_search")
    ap.add_argument("--rdo-zero", type=int, default=None, help="experiments only: override cfg.rdo_zero")
    ap.add_argument("--intra-in-p", type=int, default=None, help="experiments only: override cfg.intra_in_p")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder, config_for
    from hevc_amd.probe import VideoInfo
    from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values
    from hevc_amd.yuvio import SyntheticClip

    W, H, N = args.width, args.height, args.frames
    info = VideoInfo(W, H, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", N, N / 30.0)
    crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info, use_nvenc=False)
    level, tier = calculate_apple_hevc_level(info)
    cfg = config_for(info, crf, maxrate, bufsize, gop, level, tier)
    cfg.me_range, cfg.profile_stages = args.me_range, 1
    cfg.qp = args.qp
    cfg.host_threads = args.host_threads
    if args.intra_nxn is not None:
        cfg.intra_nxn = args.intra_nxn
    if args.intra_tiles is not None:
        cfg.intra_tiles = args.intra_tiles
    for name in ("pre_search", "rdo_zero", "intra_in_p"):
        if getattr(args, name) is not None:
            setattr(cfg, name, getattr(args, name))

    # synthetic clip -> HBM (untimed).  torch is plumbing for device memory only.
    clip = SyntheticClip("motion", rank, W, H, N)
    ys, us, vs = [], [], []
    for y, u, v in clip.frames():
        ys.append(torch.from_numpy(y).cuda(non_blocking=False))
        us.append(torch.from_numpy(u).cuda())
        vs.append(torch.from_numpy(v).cuda())
    torch.cuda.synchronize()

    phase_s = np.zeros(4)       # open, send, flush+drain, close

    def step():
        t0 = time.perf_counter()
        enc = Encoder(cfg, device=local)
        t1 = time.perf_counter()
        try:
            nbytes = 0
            for i in range(N):
                enc.send_device(ys[i].data_ptr(), us[i].data_ptr(), vs[i].data_ptr(), W, W // 2, pts=i)
                for data, _pts, _key in enc.packets():
                    nbytes += len(data)
            t2 = time.perf_counter()
            enc.flush()
            for data, _pts, _key in enc.packets():
                nbytes += len(data)
            t3 = time.perf_counter()
            st = enc.stats()
            psnr = enc.psnr_y()
        finally:
            enc.close()
        t4 = time.perf_counter()
        phase_s[:] += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        return st, nbytes, psnr

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phase_s[:] = 0
    barrier()
    t0 = time.perf_counter()
    stage_ms = np.zeros(8)
    stage_launch = np.zeros(8)
    stage_pics = np.zeros(8)
    last = None
    for _ in range(args.steps):
        st, nbytes, psnr = step()
        stage_ms += np.array(st.stage_ms[:])
        stage_launch += np.array(st.stage_launches[:])
        stage_pics += np.array(st.stage_pictures[:])
        last = (st, nbytes, psnr)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        st, nbytes, psnr = last
        fps = world * args.steps * N / dt
        # roofline of the dominant kernel: algorithmic bytes per launch / mean launch time (HIP events on the session's
        # compute stream).  Per picture: inter_ctu reads source + reference and writes the reconstruction = 3*S;
        # me_search reads source + reference luma = 2*W*H; intra = S read + S write; loop filters: see DESIGN.md.
        S = W * ((H + 7) & ~7) * 3 // 2
        per_pic = {0: 2 * S, 1: 2 * W * ((H + 7) & ~7), 2: 3 * S, 3: 4 * S, 4: 4 * S, 5: S, 6: 2 * S}
        dom = int(np.argmax(stage_ms[:7]))
        launches = max(1.0, stage_launch[dom])
        avg_ms = stage_ms[dom] / launches
        bytes_per_launch = per_pic[dom] * stage_pics[dom] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic("k_" + _lib.STAGE_NAMES[dom] if dom else "k_intra_diag", (W, H, N, args.me_range))
        out = {
            "metric": "encoded 1080p30 frames/sec/GPU; PSNR-Y parity vs libx265 at matched bitrate",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H}@30 SDR 8-bit Main, crf {crf} capped by VBV maxrate {maxrate} kbps / bufsize {bufsize} kbit "
                                   f"(the reference's libx265 operating point), {N}-frame synthetic 'motion' clip per GPU, "
                                   f"keyint {gop}, IPPP, full-search +-{args.me_range}, one clip per GPU"},
            "quality": {"psnr_y_db": round(psnr, 3), "bitrate_kbps": round(nbytes * 8 / (N / 30.0) / 1e3, 1),
                        "libx265_parity": "unavailable: no ffmpeg/libx265 on this host"},
            "stages_ms_per_picture": {_lib.STAGE_NAMES[i]: round(stage_ms[i] / max(1.0, stage_pics[i]), 4) for i in range(8)},
            "host": {"entropy_ms_per_frame_sum_over_threads": round(st.entropy_ms / max(1, st.frames_out), 4), "cpus": os.cpu_count(),
                     "step_phases_ms": dict(zip(("open", "send", "flush_drain", "close"), [round(x / args.steps * 1e3, 2) for x in phase_s])),
                     "device_ms_per_step": round(st.device_ms, 2)},
            "roofline": {"bound": "hbm", "kernel": _lib.STAGE_NAMES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "avg_launch_ms": round(avg_ms, 4), "pictures_per_launch": round(stage_pics[dom] / launches, 2),
                         "note": "integer-VALU/LDS bound path: the HBM fraction is small by construction (SURVEY.md §0.5)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, st.last_qp, args.me_range)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
