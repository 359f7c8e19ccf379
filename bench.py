#!/usr/bin/env python3
"""bench.py — encoded 1080p30 frames/s on MI355X (BASELINE.json metric), one process per GPU, no collective on the data path.

A step = one complete encode of the workload clip: BASELINE configs[1], "1080p30 SDR 8-bit Main profile, CQ mode":
300 frames (10 s) of the synthetic `motion` clip, seed = rank, already resident in HBM when the
timed region starts.  Timed: every device stage + D2H of the symbols + host CABAC until the last NAL byte exists.

N > 1 (BASELINE configs[3], one clip per GPU): either the driver starts the ranks (`python -m torch.distributed.run ... bench.py --gpus N`,
RANK / LOCAL_RANK / WORLD_SIZE in the environment) or `python bench.py --gpus N` spawns N child processes itself BEFORE any GPU call.
Every rank binds device LOCAL_RANK and encodes its own clip; barrier and MAX-over-ranks time go over gloo (host TCP on 127.0.0.1) —
nothing is exchanged between GPUs, so there is no RCCL anywhere (north_star).  Scaling is weak.

Outside the timed region rank 0 adds: `stream_ok` (the last step's stream decoded by the oracle decoder: picture count and per-plane SSE
against the source must equal the encoder's own statistics), `value_pcie_inclusive` (the same clip handed over as host buffers), a
`configs` block with the 2160p30 Main10 HDR10 run (BASELINE configs[2]), `cpu_baseline` (the oracle on one host core) and `libx265`
(the reference's own ffmpeg command when ffmpeg + libx265 exist on the host; probed, never assumed).
"""
import argparse
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

# A process that drives several sessions (a batch's workers in threads, this file's two-session leg) has 6+ HIP streams; the runtime maps them onto 4 hardware queues by
# default and streams that share one serialise.  Has to be in the environment before the HIP runtime starts: hence here, in front of torch.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "encoded 1080p30 frames/sec/GPU; PSNR-Y parity vs libx265 at matched bitrate"
# MIHEVC_BENCH_SHARE_GPU=1: rehearsal only — lets `--gpus N` run on fewer than N devices (ranks share them; the JSON line carries "shared_gpu_rehearsal": true and is no scaling number)
SHARE_GPU = os.environ.get("MIHEVC_BENCH_SHARE_GPU") == "1"


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def cpu_baseline(width, height, qp, me_range, gop_pictures=3):
    """The oracle (scalar C port of the same path: analysis + deblocking + SAO) PLUS the host CABAC coder, on ALL the host cores this process is bound to,
    the way a CPU encoder of this design would use them: one closed GOP per core (closed GOPs are independent), every core codes `gop_pictures` pictures
    (1 I + P) of the same clip from its own start picture.  value = pictures of all cores / wall time.  A bounded sample (a full 75-picture GOP per core
    would take minutes on this scalar port); a reported baseline, not the target — and not libx265: see the `libx265` block for that."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor as Pool
    from hevc_amd import _lib
    from hevc_amd.yuvio import SyntheticClip
    from oracle import oracle as O
    cores = max(1, min(32, len(os.sched_getaffinity(0))))      # GOP-parallel over the bound cores, at most 32 (the sample has to stay a few tens of seconds)
    ch = (height + 7) & ~7
    cfg = _lib.default_config()
    cfg.width, cfg.height = width, height
    lib = _lib.load()
    O.lib()
    clip = SyntheticClip("motion", 0, width, height, cores * gop_pictures + 1)
    clip.frame(0)
    with Pool(max_workers=8) as ex:              # the pictures are made before the clock starts (numpy holds the interpreter lock: 32 threads would queue on it)
        pictures = list(ex.map(clip.frame, range(cores * gop_pictures)))

    def one_gop(k):
        prm_i, prm_p = O.default_params(max(0, qp - 3), me_range=me_range), O.default_params(qp, me_range=me_range)
        prm_i.tile_cols, prm_i.tile_rows = _lib.tile_grid(cfg)        # the same IDR tile grid and knobs the device path runs
        prm_i.intra_nxn, prm_i.chroma_modes = cfg.intra_nxn, cfg.chroma_modes
        prm_p.intra_in_p, prm_p.pre_search, prm_p.rdo_zero, prm_p.rdo_cg = cfg.intra_in_p, cfg.pre_search, cfg.rdo_zero, cfg.rdo_cg
        buf = (C.c_uint8 * (8 << 20))()
        ref = prev = None
        nbytes = 0
        for i in range(gop_pictures):
            y, u, v = pictures[k * gop_pictures + i]
            f = O.Frame(np.pad(y, ((0, ch - height), (0, 0)), mode="edge"), np.pad(u, ((0, (ch - height) // 2), (0, 0)), mode="edge"),
                        np.pad(v, ((0, (ch - height) // 2), (0, 0)), mode="edge"))
            cen = O.search_centres(f, prev, 8) if i and cfg.pre_search else None        # as the session: centres from the source pictures
            a = O.analyze_intra(f, prm_i) if i == 0 else O.analyze_inter(f, ref, prm_p, centers=cen)
            prev = f
            prm = prm_i if i == 0 else prm_p
            ref, sao = O.sao(f, O.deblock(a.rec, a.cu, 8), prm)
            n = lib.mihevc_encode_picture_host(C.byref(cfg), 2 if i == 0 else 1, i, prm.qp, a.cu.ctypes.data, a.coef_y.ctypes.data, a.coef_u.ctypes.data, a.coef_v.ctypes.data,
                                               sao.ctypes.data, buf, len(buf))
            nbytes += max(0, n)
        return nbytes
    t0 = time.perf_counter()
    with Pool(max_workers=cores) as ex:          # ctypes releases the GIL inside the oracle and the coder: the GOPs run in parallel
        sizes = list(ex.map(one_gop, range(cores)))
    dt = time.perf_counter() - t0
    n = cores * gop_pictures
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/hevc_oracle.c analysis + deblocking + SAO and the host CABAC coder, {cores} closed GOPs of {gop_pictures} pictures (1 I + {gop_pictures - 1} P) of the same "
                      f"{width}x{height} clip, one GOP per bound core in parallel, {dt:.1f} s wall, {sum(sizes) * 8 * 30 / n / 1e3:.0f} kb/s at 30 fps; a bounded sample, "
                      f"not a full {75}-picture GOP per core; for libx265 itself see the `libx265` block"}


# ------------------------------------------------------------------------------------------------ libx265 (the reference's own command)
def libx265_probe():
    """(available, reason): ffmpeg on PATH and libx265 among its encoders — probed at run time."""
    exe = shutil.which("ffmpeg")
    if exe is None:
        return False, "unavailable: shutil.which('ffmpeg') found no ffmpeg on this host"
    try:
        txt = subprocess.run([exe, "-hide_banner", "-encoders"], capture_output=True, text=True, timeout=30).stdout
    except Exception as exc:      # noqa: BLE001 — a broken ffmpeg is reported, not raised
        return False, f"unavailable: `ffmpeg -encoders` failed ({exc})"
    if "libx265" not in txt:
        return False, "unavailable: ffmpeg is present but lists no libx265 encoder"
    return True, exe


def libx265_baseline(info, frames, bit_depth, budget_s=120.0):
    """The reference's CPU command (core/transcoder.py:398-411,460-493 via hevc_amd.transcoder.build_ffmpeg_command) on a raw-video
    wrapper of the same synthetic pictures, timed on this host's cores; PSNR-Y of its output (decoded by the same ffmpeg) by numpy."""
    ok, why = libx265_probe()
    if not ok:
        return {"available": False, "reason": why}
    from hevc_amd.transcoder import build_ffmpeg_command, build_ffmpeg_params
    W, H, n = info.width, info.height, len(frames)
    pix = "yuv420p10le" if bit_depth > 8 else "yuv420p"
    with tempfile.TemporaryDirectory(prefix="mihevc_x265_") as td:
        raw, out = Path(td) / "clip.yuv", Path(td) / "clip.mp4"
        with open(raw, "wb") as f:
            for y, u, v in frames:
                f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
        ff = build_ffmpeg_params(info, False, "unknown")
        cmd = build_ffmpeg_command(raw, out, ff, 0, "eng")
        i = cmd.index("-i")
        cmd[i:i] = ["-f", "rawvideo", "-pix_fmt", pix, "-s", f"{W}x{H}", "-r", "30"]      # the lossless wrapper of the same YUV
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s * 10)
        except subprocess.TimeoutExpired:
            return {"available": True, "error": "libx265 run exceeded its time budget"}
        dt = time.perf_counter() - t0
        if p.returncode != 0 or not out.exists():
            return {"available": True, "error": "ffmpeg/libx265 failed: " + (p.stderr or "")[-400:]}
        kbps = out.stat().st_size * 8 / (n / 30.0) / 1e3
        dec = subprocess.run(["ffmpeg", "-v", "error", "-i", str(out), "-f", "rawvideo", "-pix_fmt", pix, "-"], capture_output=True)
        psnr = None
        fb = W * H * 3 // 2 * (2 if bit_depth > 8 else 1)
        if dec.returncode == 0 and len(dec.stdout) >= n * fb:
            dt_ = np.dtype("<u2") if bit_depth > 8 else np.uint8
            se = 0.0
            for k, (y, _, _) in enumerate(frames):
                d = np.frombuffer(dec.stdout, dt_, W * H, k * fb).reshape(H, W).astype(np.float64) - y
                se += float((d * d).sum())
            mse = se / (n * W * H)
            psnr = 99.0 if mse == 0 else 10 * np.log10(((1 << bit_depth) - 1) ** 2 / mse)
        return {"available": True, "fps": round(n / dt, 3), "frames": n, "bitrate_kbps": round(kbps, 1), "psnr_y_db": None if psnr is None else round(float(psnr), 3),
                "cores": os.cpu_count(), "cmd": " ".join(cmd[:cmd.index(str(out))][-12:]), "note": "x265 sizes its own pool (-threads 0)"}


# ------------------------------------------------------------------------------------------------ committed profile data
def newest_profile_dir():
    """the newest profiles/r*/ that holds a traffic.json (tools/profile_bench.sh); pmc.json (tools/pmc_kernels.sh) is only used from the SAME directory:
    counters of one build say nothing about another (round 2 reported a VALU share from two builds before its traffic figures)"""
    dirs = sorted(os.path.dirname(f) for f in glob.glob(str(ROOT / "profiles" / "r*" / "traffic.json")))
    return dirs[-1] if dirs else None


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (profiles/r*/traffic.json, written by
    tools/profile_bench.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  PMC passes cannot
    run inside the timed bench, so the figure is only reported for the default workload it was collected on."""
    d = newest_profile_dir()
    if workload != (1920, 1080, 300, 15) or not d:
        return None, None, None
    try:
        ks = json.load(open(os.path.join(d, "traffic.json")))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None, None
    k = ks.get(kernel)
    return (k["hbm_bytes_per_launch"] if k else None, os.path.relpath(os.path.join(d, "traffic.json"), ROOT), ks)


def measured_valu(kernel):
    """VALU issue share of `kernel` from the SQ counter pass that sits beside that traffic.json (pmc.json, tools/pmc_kernels.sh on the bench command):
    SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in quad-cycles over the same sampled waves) = share of a wave's resident time in which it
    issues vector ALU work; times the resident waves per SIMD (the kernel's launch bounds, read from csrc/device.hip) = share of a SIMD's issue slots.
    None when that directory has no pmc.json."""
    d = newest_profile_dir()
    f = os.path.join(d, "pmc.json") if d else None
    if not f or not os.path.exists(f):
        return None
    try:
        k = json.load(open(f))["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None
    if not k:
        return None
    return {"valu_issue_frac": k["valu_issue_frac"], "valu_active_per_wave": k["valu_active_per_wave"], "waves_per_simd": k["waves_per_simd"],
            "wait_frac": k.get("wait_frac"), "source": os.path.relpath(f, ROOT)}


# ------------------------------------------------------------------------------------------------ stream check (oracle decoder)
def verify_stream(packets, headers, cfg, clip, stats, limit=None):
    """Decode the stream with the oracle decoder (one thread per closed GOP; ctypes releases the GIL) and compare with the source:
    the number of pictures and the per-plane SSE must equal what the encoder itself reported (its k_frame_sse sums), i.e. the decoder
    reconstructs exactly the pictures the encoder reconstructed.  limit = only the first `limit` pictures (then counts only)."""
    from oracle import oracle as O
    gops, cur = [], []
    for data, _pts, key in packets:
        if key and cur:
            gops.append(cur)
            cur = []
        cur.append(data)
    if cur:
        gops.append(cur)
    if limit is not None:
        gops, packets = [gops[0][:limit]], packets[:limit]
    W, H = cfg.width, cfg.height
    cw, ch = (W + 7) & ~7, (H + 7) & ~7
    starts = np.cumsum([0] + [len(g) for g in gops])

    def one(k):
        stream = b"".join(gops[k])
        if k > 0 and not cfg.repeat_headers:
            stream = headers + stream            # only the session's first IDR carries the parameter sets unless repeat_headers
        dec, info = O.decode(stream)
        sse = np.zeros(3)
        for j, f in enumerate(dec):
            y, u, v = clip.frame(int(starts[k]) + j)
            src = (np.pad(y, ((0, ch - H), (0, cw - W)), mode="edge"), np.pad(u, ((0, (ch - H) // 2), (0, (cw - W) // 2)), mode="edge"),
                   np.pad(v, ((0, (ch - H) // 2), (0, (cw - W) // 2)), mode="edge"))
            for p, (a, b) in enumerate(zip(src, (f.y, f.u, f.v))):
                d = a.astype(np.int64) - b.astype(np.int64)
                sse[p] += float((d * d).sum())
        return len(dec), sse

    try:
        with ThreadPoolExecutor(max_workers=min(8, len(gops))) as ex:
            res = list(ex.map(one, range(len(gops))))
    except Exception as exc:      # noqa: BLE001 — a stream the decoder rejects is a failed check, with the reason
        return {"stream_ok": False, "error": str(exc)[:200]}
    n = sum(r[0] for r in res)
    sse = sum(r[1] for r in res)
    out = {"pictures_decoded": n, "decoder": "oracle/hevc_dec.c (own decoder; no third-party decoder exists on this pool)"}
    if limit is None:
        same = [float(sse[0]) == float(stats.sse_y), float(sse[1]) == float(stats.sse_u), float(sse[2]) == float(stats.sse_v)]
        out["stream_ok"] = bool(n == len(packets) and all(same))
        out["sse_equal_to_encoder_stats"] = same
    else:
        out["stream_ok"] = bool(n == len(packets))
    return out


# ------------------------------------------------------------------------------------------------ rank launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: N children, one per GPU, started before this process touches a GPU."""
    if args.stub:
        n_dev = args.gpus
    else:
        import torch
        n_dev = torch.cuda.device_count()          # counts devices without initialising the GPU runtime in this process
    if n_dev < args.gpus and not (SHARE_GPU and n_dev >= 1):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} MI355X visible on this host — one process per GPU needs {args.gpus} devices "
                         "(there is no CPU fallback and ranks never share a device)")
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                r = p.poll()
                if r is None:
                    continue
                procs.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in procs:             # a failed rank leaves the others waiting at the barrier: stop them (exact PIDs)
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    sys.exit(rc)


class Ranks:
    """barrier / max / gather over gloo (host TCP): timing plumbing only, nothing from the data path goes through it"""

    def __init__(self, rank, world):
        self.rank, self.world, self.dist = rank, world, None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."): the bench's stdout is ONE JSON line, so file
            # descriptor 1 points at stderr while the group comes up (the first collective opens the full mesh)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def gather(self, value):
        if self.dist is None:
            return [value]
        import torch
        t = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(t, torch.tensor([value], dtype=torch.float64))
        return [float(x.item()) for x in t]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ one rank
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
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs (stream check, PCIe-inclusive run, 2160p config, libx265 probe run)")
    ap.add_argument("--host-threads", type=int, default=0, help="CABAC worker threads (0 = library default)")
    ap.add_argument("--qp", type=int, default=-1, help="experiments only: force the P-picture QP instead of deriving it from the CRF")
    ap.add_argument("--profile-stages", type=int, default=2, help="HIP events in the timed steps: 2 = around the dominant kernel (inter_ctu) only, 1 = every stage, 0 = none")
    ap.add_argument("--stub", action="store_true", help="tests only: no GPU, no encoder — exercises the rank spawn / rendezvous / JSON path (tests/test_bench_ranks.py)")
    for name in ("intra-nxn", "intra-tiles", "pre-search", "rdo-zero", "intra-in-p", "chroma-modes", "gop-balance", "scenecut", "rdo-cg", "gops-in-flight"):
        ap.add_argument("--" + name, type=int, default=None, help="experiments only: override cfg." + name.replace("-", "_"))
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                        # never returns
    # stdout carries ONE JSON line and nothing else: from here on file descriptor 1 is stderr for everything in this process (gloo's connection
    # notes, runtime warnings, stray prints of a library), and the line goes out through a private duplicate of the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    W, H, N = args.width, args.height, args.frames

    if args.stub:
        ranks = Ranks(rank, world)
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.05 * (1 + rank))
        own = ranks.gather(time.perf_counter() - t0)         # every rank's own K steps
        ranks.barrier()
        dts = ranks.gather(time.perf_counter() - t0)         # barrier to barrier; the job takes the MAX over ranks
        if rank == 0:
            dt = max(dts)
            emit({"metric": METRIC, "value": round(world * args.steps * N / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "u8", "data": "stub", "config": {"workload": "stub"}, "per_rank_fps": [round(args.steps * N / d, 2) for d in own], "stub": True})
        ranks.close()
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    if local >= torch.cuda.device_count():
        if not SHARE_GPU:
            raise SystemExit(f"bench.py: rank {rank} wants device {local} but only {torch.cuda.device_count()} are visible")
        local %= torch.cuda.device_count()      # rehearsal of the multi-rank path on a box with fewer devices: ranks share them, the line says so
    torch.cuda.set_device(local)
    torch.cuda.init()                                 # torch's HIP runtime comes up before libmihevc's
    from hevc_amd.utils import bind_to_device_node
    numa_node = bind_to_device_node(local)            # one process per GPU: CABAC workers and pinned buffers on the device's NUMA node
    ranks = Ranks(rank, world)

    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder, config_for
    from hevc_amd.probe import VideoInfo
    from hevc_amd.transcoder import calculate_apple_hevc_level, calculate_dynamic_values
    from hevc_amd.yuvio import SyntheticClip

    def operating_point(w, h, n, hdr):
        if hdr:
            info = VideoInfo(w, h, 30.0, "bt2020", "smpte2084", "bt2020nc", "yuv420p10le", "", "", 0, True, "eng", n, n / 30.0)
        else:
            info = VideoInfo(w, h, 30.0, "bt709", "bt709", "bt709", "yuv420p", "", "", 0, False, "eng", n, n / 30.0)
        crf, _cq, maxrate, bufsize, gop = calculate_dynamic_values(info, use_nvenc=False)
        level, tier = calculate_apple_hevc_level(info)
        return info, config_for(info, crf, maxrate, bufsize, gop, level, tier), (crf, maxrate, bufsize, gop)

    info, cfg, (crf, maxrate, bufsize, gop) = operating_point(W, H, N, False)
    cfg.me_range, cfg.profile_stages = args.me_range, args.profile_stages
    cfg.qp = args.qp
    cfg.host_threads = args.host_threads
    for name in ("intra_nxn", "intra_tiles", "pre_search", "rdo_zero", "intra_in_p", "chroma_modes", "gop_balance", "scenecut", "rdo_cg", "gops_in_flight"):
        if getattr(args, name) is not None:
            setattr(cfg, name, getattr(args, name))

    # synthetic clip -> HBM (untimed).  torch is plumbing for device memory only.
    clip = SyntheticClip("motion", rank, W, H, N)
    clip.frame(0)                                    # builds the texture once, before the threads
    with ThreadPoolExecutor(max_workers=8) as ex:
        host_frames = list(ex.map(clip.frame, range(N)))
    ys = [torch.from_numpy(f[0]).cuda() for f in host_frames]
    us = [torch.from_numpy(f[1]).cuda() for f in host_frames]
    vs = [torch.from_numpy(f[2]).cuda() for f in host_frames]
    torch.cuda.synchronize()

    import ctypes as C

    def pointer_arrays(planes):      # the frames stay where they are: their device pointers as C arrays, made once
        return tuple((C.c_void_p * len(p))(*[t.data_ptr() for t in p]) for p in planes)
    clip_ptrs = pointer_arrays((ys, us, vs))
    phase_s = np.zeros(4)       # open, send, flush+drain, close

    def step(keep=None, host=False, c=cfg, frames=None, n=N, dev=None, use_async=False):
        """one pass over the clip; dev = (ys, us, vs, luma pitch in samples) picks other device-resident planes than the bench clip's"""
        dy, du, dv, dw = dev if dev is not None else (ys, us, vs, W)
        t0 = time.perf_counter()
        enc = Encoder(c, device=local)
        t1 = time.perf_counter()
        try:
            nbytes = 0
            if not host:          # frames in HBM: ONE call hands all of them over (mihevc_send_frames_device)
                py, pu, pv = clip_ptrs if dev is None and n == N else pointer_arrays((dy[:n], du[:n], dv[:n]))
                enc.send_device_batch(py, pu, pv, dw, dw // 2, first_pts=0)
                for pk in enc.packets():
                    nbytes += len(pk[0])
                    if keep is not None:
                        keep.append(pk)
            for i in range(n if host else 0):
                if host and use_async:
                    enc.send_async(*frames[i], pts=i)      # mihevc_send_frame_async: the planes stay valid until flush (they are the clip's own pinned arrays)
                elif host:
                    enc.send(*frames[i], pts=i)
                else:
                    enc.send_device(dy[i].data_ptr(), du[i].data_ptr(), dv[i].data_ptr(), dw, dw // 2, pts=i)
                for pk in enc.packets():
                    nbytes += len(pk[0])
                    if keep is not None:
                        keep.append(pk)
            t2 = time.perf_counter()
            enc.flush()
            for pk in enc.packets():
                nbytes += len(pk[0])
                if keep is not None:
                    keep.append(pk)
            t3 = time.perf_counter()
            st = enc.stats()
            psnr = enc.psnr_y()
            hdrs = enc.headers()
        finally:
            enc.close()
        t4 = time.perf_counter()
        phase_s[:] += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        return st, nbytes, psnr, hdrs

    def barrier():
        ranks.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phase_s[:] = 0
    barrier()
    t0 = time.perf_counter()
    stage_ms, stage_launch, stage_pics = np.zeros(8), np.zeros(8), np.zeros(8)
    last, kept = None, []
    for k in range(args.steps):
        st, nbytes, psnr, hdrs = step(keep=kept if k == args.steps - 1 else None)
        stage_ms += np.array(st.stage_ms[:])
        stage_launch += np.array(st.stage_launches[:])
        stage_pics += np.array(st.stage_pictures[:])
        last = (st, nbytes, psnr, hdrs)
    own = time.perf_counter() - t0                   # this rank's own K steps (every step ends with the last NAL byte on the host)
    roof_ms, roof_launch, roof_pics = stage_ms[2], stage_launch[2], stage_pics[2]      # inter_ctu, measured live in the timed steps
    barrier()
    dts = ranks.gather(time.perf_counter() - t0)     # barrier to barrier; the job takes the MAX over ranks
    dt = max(dts)
    own = ranks.gather(own)
    st, nbytes, psnr, hdrs = last

    # ---- untimed legs ----
    if rank == 0 and args.profile_stages != 1:       # one more step with every stage bracketed: the per-stage table (events cost ~5 % fps)
        cfg.profile_stages = 1
        stp = step()[0]
        stage_ms, stage_launch, stage_pics = np.array(stp.stage_ms[:]), np.array(stp.stage_launches[:], dtype=float), np.array(stp.stage_pictures[:], dtype=float)
        cfg.profile_stages = args.profile_stages
    check = {"stream_ok": None}
    if not args.no_extras:
        check = verify_stream(kept, hdrs, cfg, clip, st)            # every rank checks its own stream
    oks = ranks.gather(1.0 if check.get("stream_ok") else 0.0)

    if rank == 0:
        fps = world * args.steps * N / dt
        # roofline of the dominant kernel: algorithmic bytes per launch / mean launch time (HIP events on the session's
        # compute stream).  Per picture: inter_ctu reads source + reference and writes the reconstruction = 3*S;
        # me_search reads source + reference luma = 2*W*H; intra = S read + S write; loop filters: see DESIGN.md.
        S = W * ((H + 7) & ~7) * 3 // 2
        per_pic = {0: 2 * S, 1: 2 * W * ((H + 7) & ~7), 2: 3 * S, 3: 4 * S, 4: 4 * S, 5: S, 6: 2 * S}
        dom = int(np.argmax(stage_ms[:7]))               # by the fully bracketed step; the timed steps bracket inter_ctu, the dominant kernel
        if args.profile_stages == 2 or dom == 2:
            dom, launches, avg_ms = 2, max(1.0, roof_launch), roof_ms / max(1.0, roof_launch)
            bytes_per_launch = per_pic[2] * roof_pics / launches
        else:
            launches = max(1.0, stage_launch[dom])
            avg_ms = stage_ms[dom] / launches
            bytes_per_launch = per_pic[dom] * stage_pics[dom] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kname = "k_" + _lib.STAGE_NAMES[dom] if dom else "k_intra_diag"
        traffic, traffic_src, traffic_all = measured_traffic(kname, (W, H, N, args.me_range))
        valu = measured_valu(kname)
        ok265, why265 = libx265_probe()
        out = {
            "metric": METRIC,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H}@30 SDR 8-bit Main, crf {crf} capped by VBV maxrate {maxrate} kbps / bufsize {bufsize} kbit "
                                   f"(the reference's libx265 operating point), {N}-frame synthetic 'motion' clip per GPU, "
                                   f"keyint {gop} ({(N + gop - 1) // gop} closed GOPs" + (" of equal length" if cfg.gop_balance else "") + f"), IPPP, full-search +-{args.me_range}, one clip per GPU, input resident in HBM"},
            "per_rank_fps": [round(args.steps * N / d, 2) for d in own], **({"shared_gpu_rehearsal": True} if SHARE_GPU else {}),
            "stream_ok": bool(all(o == 1.0 for o in oks)) if not args.no_extras else None,
            "stream_check": check,
            "quality": {"psnr_y_db": round(psnr, 3), "bitrate_kbps": round(nbytes * 8 / (N / 30.0) / 1e3, 1), "vbv_maxrate_kbps": maxrate,
                        "libx265_parity": "see `libx265`" if ok265 else why265},
            "stages_ms_per_picture": {_lib.STAGE_NAMES[i]: round(stage_ms[i] / max(1.0, stage_pics[i]), 4) for i in range(8)},
            "host": {"entropy_ms_per_frame_sum_over_threads": round(st.entropy_ms / max(1, st.frames_out), 4), "cpus": os.cpu_count(), "numa_node": numa_node, "cpus_bound": len(os.sched_getaffinity(0)),
                     "step_phases_ms": dict(zip(("open", "send", "flush_drain", "close"), [round(x / args.steps * 1e3, 2) for x in phase_s])),
                     "device_ms_per_step": round(st.device_ms, 2),
                     "chunk_host_ms": {"before_first_launch": round(st.reserved[3] / 1e3, 2), "after_last_kernel": round(st.reserved[4] / 1e3, 2), "chunk_wall": round(st.reserved[5] / 1e3, 2)}},
            "roofline": {"bound": "hbm", "kernel": _lib.STAGE_NAMES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "avg_launch_ms": round(avg_ms, 4), "pictures_per_launch": round(bytes_per_launch / per_pic[dom], 2),
                         "dominant_by_full_profile": _lib.STAGE_NAMES[int(np.argmax(stage_ms[:7]))],
                         "valu_issue_frac": valu["valu_issue_frac"] if valu else None, "valu": valu,
                         "note": "integer-VALU/LDS bound path: the HBM fraction is small by construction (SURVEY.md §0.5); valu_issue_frac is the "
                                 "share of SIMD issue slots the kernel fills, from the committed SQ counter pass"},
        }
        # the whole pipeline beside its dominant kernel: algorithmic bytes of a picture over ALL stages (an IDR picture reads its source and writes its
        # reconstruction, a P picture also reads its reference once: SURVEY.md §8d) x pictures / step time, and the HBM bytes the counters saw for all kernels
        n_idr = (N + gop - 1) // gop
        alg_clip = n_idr * 2 * S + (N - n_idr) * 3 * S
        pipe = {"algorithmic_bytes_per_picture": int(alg_clip / N), "achieved": round(world * args.steps * alg_clip / dt / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(world * args.steps * alg_clip / dt / 1e9 / HBM_PEAK_GBS, 6), "traffic_bytes_per_picture": None}
        if traffic_all:
            tot = sum(k["hbm_bytes_per_launch"] * k["dispatches"] for k in traffic_all.values())
            pipe["traffic_bytes_per_picture"] = int(tot / N)          # the PMC passes run ONE step of this clip
            pipe["traffic_over_algorithmic"] = round(tot / alg_clip, 2)
            pipe["traffic_source"] = traffic_src
        out["pipeline"] = pipe
        if not args.no_extras and world == 1:     # the untimed legs describe ONE device: at N > 1 the other ranks would only wait for rank 0
            # the boundary hands over HOST buffers (mihevc_send_frame): the same clip, upload inside the clock
            n_p = max(1, min(args.steps, 3))
            step(host=True, frames=host_frames)
            t1 = time.perf_counter()
            for _ in range(n_p):
                step(host=True, frames=host_frames)
            out["value_pcie_inclusive"] = round(n_p * N / (time.perf_counter() - t1), 2)
            # the same with the caller's planes in PINNED host memory (what a decoder feeding this library would use): the upload is a DMA, no staging copy
            pinned = [tuple(torch.from_numpy(p).pin_memory().numpy() for p in f) for f in host_frames]
            step(host=True, frames=pinned, use_async=True)
            t1 = time.perf_counter()
            for _ in range(n_p):
                step(host=True, frames=pinned, use_async=True)
            out["value_pcie_inclusive_pinned"] = round(n_p * N / (time.perf_counter() - t1), 2)
            # ... and as a batch runs it (hevc_amd/batch.py: two sessions per device): two host threads, each feeding its own session clip after clip from the pinned
            # planes — one session's uploads (DMA) run under the other's kernels.  A single session cannot hide them: its four GOP lanes step together, so the first
            # step needs frame 225 of 300 on the device.  Whole-device throughput, frames of both sessions / wall time.
            import threading
            reps, errs = max(2, n_p), []

            def feed():
                try:
                    for _ in range(reps):
                        step(host=True, frames=pinned, use_async=True)
                except Exception as e:       # noqa: BLE001 -- reported below
                    errs.append(repr(e))
            step(host=True, frames=pinned, use_async=True)
            th = [threading.Thread(target=feed) for _ in range(2)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            out["value_pcie_inclusive_pinned_two_sessions"] = None if errs else round(2 * reps * N / (time.perf_counter() - t1), 2)
            # the same two threads with the frames resident in HBM: what the device delivers when a second session's launches fill the first one's IDR steps and kernel tails
            errs2 = []

            def feed_dev():
                try:
                    for _ in range(reps):
                        step()
                except Exception as e:       # noqa: BLE001
                    errs2.append(repr(e))
            th = [threading.Thread(target=feed_dev) for _ in range(2)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            out["value_two_sessions"] = None if errs2 else round(2 * reps * N / (time.perf_counter() - t1), 2)
            # ONE clip, its two halves (whole GOPs each) coded by two sessions at once and joined: closed GOPs are independent, the IDR pictures sit where one session puts them
            errs3, half = [], ((N + gop - 1) // gop + 1) // 2 * (-(-N // ((N + gop - 1) // gop)))

            def code_half(a, b):
                try:
                    sub_info, sub_cfg, _ = operating_point(W, H, b - a, False)
                    sub_cfg.me_range, sub_cfg.host_threads = args.me_range, args.host_threads
                    with Encoder(sub_cfg, device=local) as enc:
                        for i in range(a, b):
                            enc.send_device(ys[i].data_ptr(), us[i].data_ptr(), vs[i].data_ptr(), W, W // 2, pts=i)
                            for _pk in enc.packets():
                                pass
                        enc.flush()
                        for _pk in enc.packets():
                            pass
                except Exception as e:       # noqa: BLE001
                    errs3.append(repr(e))
            best = None
            if 0 < half < N:
                for _ in range(1 + reps):
                    th = [threading.Thread(target=code_half, args=(0, half)), threading.Thread(target=code_half, args=(half, N))]
                    t1 = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    dt1 = time.perf_counter() - t1
                    best = dt1 if best is None else min(best, dt1)
            out["value_one_clip_two_sessions"] = None if errs3 or best is None else round(N / best, 2)
            out["pcie_inclusive_note"] = ("value_one_clip_two_sessions: frames in HBM, the clip's two halves (whole GOPs) coded by two sessions at once (best of the repeats; tests/measure_split.py); "
                                          "value_two_sessions: frames in HBM, two host threads with a session each, clip after clip (not the headline: `value` is one clip, one session); "
                                          "value_pcie_inclusive: mihevc_send_frame (synchronous copy per frame, pageable planes); value_pcie_inclusive_pinned: "
                                          "mihevc_send_frame_async from page-locked planes (uploads run as DMA beside the caller, the chunk waits for the last one); "
                                          "value_pcie_inclusive_pinned_two_sessions: the same from two host threads with a session each (a batch's two workers per device): "
                                          "one session's uploads run under the other's kernels")
            del pinned
            out["configs"] = {"1080p30_sdr_8bit": {"fps_hbm_resident": round(fps / world, 2), "fps_pcie_inclusive": out["value_pcie_inclusive"],
                                                    "fps_pcie_inclusive_pinned": out["value_pcie_inclusive_pinned"], "bitrate_kbps": out["quality"]["bitrate_kbps"], "psnr_y_db": out["quality"]["psnr_y_db"]}}
            del ys[:], us[:], vs[:]
            out["configs"]["2160p30_hdr10_main10"] = run_2160p(local, operating_point, step, SyntheticClip, verify_stream, torch)
            out["libx265"] = libx265_baseline(info, host_frames, 8) if ok265 else {"available": False, "reason": why265}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, st.last_qp, args.me_range)
        emit(out)
    ranks.barrier()
    ranks.close()


def run_2160p(local, operating_point, step, SyntheticClip, verify, torch, n=300):
    """BASELINE configs[2]: 3840x2160 Main10 HDR10 at the reference's operating point, the 300-frame clip SURVEY §8d names (5 closed GOPs of
    keyint 60, one lane each): frames resident in HBM (7.5 GB) as for the headline value, and handed over as pageable host buffers (PCIe-inclusive)."""
    w, h = 3840, 2160
    info, cfg, (crf, maxrate, bufsize, gop) = operating_point(w, h, n, True)
    clip = SyntheticClip("motion", 0, w, h, n, bit_depth=10)
    clip.frame(0)
    with ThreadPoolExecutor(max_workers=8) as ex:
        frames = list(ex.map(clip.frame, range(n)))
    kept = []
    step(host=True, frames=frames, c=cfg, n=n)
    t0 = time.perf_counter()
    st, nbytes, psnr, hdrs = step(keep=kept, host=True, frames=frames, c=cfg, n=n)
    dt = time.perf_counter() - t0
    chk = verify(kept, hdrs, cfg, clip, st, limit=4)
    dev = tuple([torch.from_numpy(f[k]).cuda() for f in frames] for k in range(3)) + (w,)      # 3 GB of HBM
    torch.cuda.synchronize()
    step(c=cfg, n=n, dev=dev)
    t0 = time.perf_counter()
    for _ in range(2):
        step(c=cfg, n=n, dev=dev)
    dt_dev = (time.perf_counter() - t0) / 2
    return {"fps_hbm_resident": round(n / dt_dev, 2), "fps_pcie_inclusive": round(n / dt, 2), "frames": n, "bitrate_kbps": round(nbytes * 8 / (n / 30.0) / 1e3, 1), "vbv_maxrate_kbps": maxrate,
            "psnr_y_db": round(psnr, 3), "crf": crf, "keyint": gop, "first_pictures_decode": chk.get("stream_ok"), "pictures_decoded": chk.get("pictures_decoded")}


if __name__ == "__main__":
    main()
