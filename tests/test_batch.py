"""CPU: headless batch queue (gui/mainwindow.py queue semantics) + raw clip IO + the N>1 bench path on gloo."""
import csv
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

from hevc_amd import batch, probe, yuvio

ROOT = Path(__file__).resolve().parents[1]


def test_queue_refill_affinity_and_csv(tmp_path):
    files = [tmp_path / f"clip{i}.mp4" for i in range(7)]
    for f in files:
        f.write_bytes(b"x")
    seen, lock, running, peak = [], threading.Lock(), [0], [0]

    def fake_convert(f, out_dir, progress_callback=None, stop_event=None, device=None, **kw):
        with lock:
            running[0] += 1
            peak[0] = max(peak[0], running[0])
        time.sleep(0.02)
        with lock:
            running[0] -= 1
            seen.append((f.name, device))
        return {"file": f.name, "status": "SUCCESS", "quality": 19, "retries": 0, "method": "MI355X", "hdr": False}

    r = batch.BatchRunner(files, tmp_path / "out", max_workers=3, convert=fake_convert, n_devices=2).start()
    res = r.wait()
    assert len(res) == 7 and peak[0] <= 3 and {d for _, d in seen} == {0, 1}      # workers 0,1,2 -> devices 0,1,0
    rows = list(csv.DictReader(open(tmp_path / "out" / "transcode_log.csv")))
    assert len(rows) == 7 and list(rows[0])[:6] == ["file", "status", "quality", "retries", "method", "hdr"]
    assert sorted(x["file"] for x in rows) == sorted(f.name for f in files)


def test_stop_all_signals_only_active_workers(tmp_path):
    files = [tmp_path / f"c{i}.mov" for i in range(4)]
    started = threading.Event()

    def slow(f, out_dir, progress_callback=None, stop_event=None, **kw):
        started.set()
        cancelled = stop_event.wait(2.0)
        return {"file": f.name, "status": "CANCELLED" if cancelled else "SUCCESS", "quality": None, "retries": 0, "method": "CPU", "hdr": False}

    r = batch.BatchRunner(files, tmp_path / "o", max_workers=2, convert=slow, n_devices=0).start()
    started.wait(2)
    time.sleep(0.05)
    r.stop_all()                     # reference semantics: the two active ones stop, the queue keeps feeding
    time.sleep(0.1)
    r.stop_all(cancel_queued=True)
    res = r.wait()
    assert sum(x["status"] == "CANCELLED" for x in res) >= 2 and len(res) <= 4


def test_worker_exception_becomes_failed_unknown(tmp_path):
    def boom(*a, **k):
        raise RuntimeError("encoder died")
    res = batch.BatchRunner([tmp_path / "a.mkv"], tmp_path / "o", convert=boom, n_devices=0).start().wait()
    assert res[0]["status"] == "FAILED" and res[0]["method"] == "UNKNOWN"       # gui/worker.py:43-52


def test_scan_inputs(tmp_path):
    for n in ("a.mp4", "b.txt", "sub/c.MKV", "d_64x64_30.yuv"):
        p = tmp_path / n
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"")
    assert [p.name for p in batch.scan_inputs(tmp_path)] == ["a.mp4", "d_64x64_30.yuv", "c.MKV"]


def test_raw_clip_io_and_native_probe(tmp_path):
    clip = yuvio.SyntheticClip("motion", 3, 64, 48, 4)
    frames = list(clip.frames())
    again = list(yuvio.SyntheticClip("motion", 3, 64, 48, 4).frames())
    assert all(np.array_equal(a, b) for fa, fb in zip(frames, again) for a, b in zip(fa, fb))        # deterministic
    assert not np.array_equal(frames[0][0], frames[1][0])
    y4m = tmp_path / "m.y4m"
    yuvio.write_y4m(y4m, frames, 64, 48, 30)
    c = yuvio.open_clip(y4m)
    assert (c.width, c.height, c.n_frames, c.bit_depth) == (64, 48, 4, 8)
    back = list(c.frames())
    c.close()
    assert all(np.array_equal(a, b) for fa, fb in zip(frames, back) for a, b in zip(fa, fb))
    raw = tmp_path / "clip_64x48_30_10bit_hdr.yuv"
    f10 = list(yuvio.SyntheticClip("bars", 0, 64, 48, 2, bit_depth=10).frames())
    yuvio.write_yuv(raw, f10, 10)
    info = probe.probe_media(raw)
    assert (info.width, info.height, info.hdr, info.pix_fmt, info.nb_frames, info.audio_channels) == (64, 48, True, "yuv420p10le", 2, 0)
    assert f10[0][0].max() <= 940 and f10[0][0].min() >= 64


GLOO_WORKER = r'''
import os, sys, json, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from hevc_amd.yuvio import SyntheticClip
clip = SyntheticClip("motion", rank, 64, 48, 2)        # one clip per rank, seed = rank: no data-path collective
y = next(clip.frames())[0]
dist.barrier()
dt = 0.1 * (rank + 1)                                   # pretend rank 1 is slower
t = torch.tensor([dt]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
frames = torch.tensor([2.0]); dist.all_reduce(frames)   # only for the test's bookkeeping
if rank == 0:
    print(json.dumps({"max_dt": float(t.item()), "frames": float(frames.item()), "value": world * 2 / float(t.item()), "sum0": int(y.sum())}))
else:
    print(json.dumps({"sum1": int(y.sum())}))
dist.destroy_process_group()
'''


def test_two_rank_sharding_and_max_over_ranks_timing_on_gloo(tmp_path):
    """The N>1 contract of bench.py (one clip per rank, barrier, MAX over ranks, value = all frames / that time)
    exercised with world_size 2 on the gloo backend."""
    import json
    w = tmp_path / "w.py"
    w.write_text(GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(w), str(ROOT)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert abs(outs[0]["max_dt"] - 0.2) < 1e-6 and outs[0]["frames"] == 4.0 and abs(outs[0]["value"] - 20.0) < 1e-3
    assert outs[0]["sum0"] != outs[1]["sum1"]            # different seeds -> different clips


def convert_for_process_test(f, out_dir, progress_callback=None, stop_event=None, device=None, **kw):
    """stand-in for convert_video in the worker PROCESSES (top level so a spawned child can import it)"""
    import os
    for i in range(3):
        if stop_event.is_set():
            return {"file": f.name, "status": "CANCELLED", "quality": None, "retries": 0, "method": "MI355X", "hdr": False}
        progress_callback(f.name, i + 1, 3)
        time.sleep(0.02)
    (Path(out_dir) / (f.stem + ".pid")).write_text(f"{os.getpid()} {device}")
    return {"file": f.name, "status": "SUCCESS", "quality": 19, "retries": 0, "method": "MI355X", "hdr": False}


def test_one_worker_process_per_device(tmp_path):
    """process mode (what a real MI355X batch uses): two spawned workers pinned to devices 0 / 1 take files from the parent's FIFO,
    progress and results come back over the queue, the CSV carries method and device"""
    import os
    files = [tmp_path / f"clip{i}.mp4" for i in range(5)]
    for f in files:
        f.write_bytes(b"x")
    out = tmp_path / "out"
    prog, done = [], []
    r = batch.BatchRunner(files, out, max_workers=2, convert=convert_for_process_test, n_devices=2, use_processes=True,
                          on_progress=lambda n, a, b: prog.append((n, a, b)), on_finished=done.append).start()
    res = r.wait()
    assert len(res) == 5 and all(x["status"] == "SUCCESS" for x in res) and len(done) == 5
    pids = {}
    for f in files:
        pid, dev = (out / (f.stem + ".pid")).read_text().split()
        pids.setdefault(pid, set()).add(dev)
    assert len(pids) == 2 and os.getpid() not in {int(p) for p in pids}           # two worker processes, neither is the parent
    assert sorted(d for v in pids.values() for d in v) == ["0", "1"]              # each pinned to ONE device
    assert {n for n, _, _ in prog} == {f.name for f in files} and all(b == 3 for _, _, b in prog)
    rows = list(csv.DictReader(open(out / "transcode_log.csv")))
    assert len(rows) == 5 and {x["device"] for x in rows} == {"0", "1"} and {x["method"] for x in rows} == {"MI355X"}


def convert_that_kills_its_process(f, out_dir, progress_callback=None, stop_event=None, device=None, **kw):
    """stand-in for convert_video whose worker PROCESS dies on one file (a device runtime fault takes the process with it) while the others
    keep sending progress messages ten times a second"""
    import os
    if f.stem == "clip1":
        os._exit(17)
    for i in range(10):
        progress_callback(f.name, i + 1, 10)
        time.sleep(0.1)
    (Path(out_dir) / (f.stem + ".pid")).write_text(str(os.getpid()))
    return {"file": f.name, "status": "SUCCESS", "quality": 19, "retries": 0, "method": "MI355X", "hdr": False}


def test_a_dead_worker_process_is_named_and_replaced(tmp_path):
    """the lost file is logged under its real name although the other worker's progress messages never let the queue run dry, the
    slot gets a fresh process, and the files still queued are coded"""
    files = [tmp_path / f"clip{i}.mp4" for i in range(5)]
    for f in files:
        f.write_bytes(b"x")
    out = tmp_path / "out"
    res = batch.BatchRunner(files, out, max_workers=2, convert=convert_that_kills_its_process, n_devices=2, use_processes=True).start().wait()
    by = {x["file"]: x for x in res}
    assert set(by) == {f.name for f in files}, res
    assert by["clip1.mp4"]["status"] == "FAILED" and by["clip1.mp4"]["method"] == "UNKNOWN"
    assert all(by[f.name]["status"] == "SUCCESS" for f in files if f.stem != "clip1")
    rows = list(csv.DictReader(open(out / "transcode_log.csv")))
    assert sorted(x["file"] for x in rows) == sorted(f.name for f in files)
    assert len({(out / (f.stem + ".pid")).read_text() for f in files if f.stem != "clip1"}) >= 2       # the replacement process took files too
