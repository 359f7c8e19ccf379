"""Headless batch queue: the reference's MainWindow queue semantics without Qt, one clip per GPU.

Reference behaviour reproduced (gui/mainwindow.py:254-355): recursive scan for INPUT_EXTS under the input
directory (config.py:9-12), a FIFO of files, N workers each running `convert_video` on one file at a time, refill
on completion, a CSV log rewritten after every finished file with the six reference columns
`file,status,quality,retries,method,hdr` (gui/mainwindow.py:347-355), and `stop_all` that signals only the ACTIVE
workers (gui/mainwindow.py:303-308) — queued files keep starting, as in the reference.
New: GPU affinity.  Worker k is pinned to MI355X ordinal `k % n_devices`, so a batch shards one clip per GPU across
the node with no collective (BASELINE.json configs[3]).  Extra CSV columns follow the six.

One PROCESS per worker when the real `convert_video` drives MI355X devices (one process per GPU): at ~3000 fps per device
the per-frame Python of eight clips would queue on one interpreter lock, and a fault in one device's runtime must not take the other
seven clips down.  The workers are spawned (never forked: the parent may already hold a GPU context), take files from the parent one at
a time and send progress / results back over a queue; the parent keeps the FIFO, the CSV and the callbacks.  With an injected `convert`
(tests) or no device the workers are threads, as in round 1.
"""
from __future__ import annotations

import csv
import importlib
import logging
import multiprocessing as mp
import threading
import time
from collections import deque
from pathlib import Path
from typing import Callable, Dict, List, Optional

from .transcoder import convert_video
from .utils import mi355x_device_count

logger = logging.getLogger(__name__)

INPUT_EXTS = ('.mp4', '.mov', '.mkv', '.avi', '.wmv', '.flv', '.ts', '.m2ts', '.mts', '.m4v', '.webm', '.3gp', '.f4v', '.ogv', '.vob',
              '.mpg', '.mpeg', '.y4m', '.yuv')          # config.py:9-12 plus the raw containers the native path reads
CSV_FIELDS = ['file', 'status', 'quality', 'retries', 'method', 'hdr']          # gui/mainwindow.py:351
EXTRA_FIELDS = ['seconds', 'device']


def scan_inputs(input_dir: Path) -> List[Path]:
    return sorted(p for p in Path(input_dir).rglob('*') if p.is_file() and p.suffix.lower() in INPUT_EXTS)


def _proc_worker(slot, device, task_q, result_q, stop_ev, out_dir, kw, convert_ref):
    """Body of one worker process: files arrive one at a time from the parent; progress is throttled to ~10 messages a second."""
    mod, name = convert_ref
    convert = getattr(importlib.import_module(mod), name)
    if device is not None and device >= 0:
        from .utils import bind_to_device_node
        bind_to_device_node(device)          # this process drives one device: its threads and pinned buffers stay on that device's NUMA node
    while True:
        f = task_q.get()
        if f is None:
            return
        last = [0.0]                         # (the parent clears stop_ev before it hands out a file: a stop_all() that lands after the hand-over stays set)

        def progress(fname, frame, total):
            now = time.time()
            if frame >= total or now - last[0] > 0.1:
                last[0] = now
                result_q.put(('progress', slot, fname, frame, total))
        t0 = time.time()
        try:
            k = dict(kw)
            if device is not None:
                k['device'] = device
            res = dict(convert(Path(f), Path(out_dir), progress_callback=progress, stop_event=stop_ev, **k))
        except Exception as exc:             # gui/worker.py:43-52: last-resort FAILED/UNKNOWN result
            res = {'file': Path(f).name, 'status': 'FAILED', 'quality': None, 'retries': 0, 'method': 'UNKNOWN', 'hdr': False, 'error': str(exc)}
        res['seconds'] = round(time.time() - t0, 3)
        res['device'] = device if res.get('method') == 'MI355X' else ''
        result_q.put(('done', slot, res))


class BatchRunner:
    def __init__(self, files: List[Path], out_dir: Path, max_workers: Optional[int] = None, debug=False, skip_validator=False,
                 force_cpu=False, force_gpu=False, csv_path: Optional[Path] = None,
                 on_progress: Optional[Callable[[str, int, int], None]] = None, on_finished: Optional[Callable[[Dict], None]] = None,
                 convert=convert_video, n_devices: Optional[int] = None, use_processes: Optional[bool] = None):
        self.queue = deque(Path(f) for f in files)
        self.out_dir = Path(out_dir)
        self.n_devices = mi355x_device_count() if n_devices is None else n_devices
        # default: TWO worker processes per device.  One session leaves a good part of an MI355X idle while it walks an IDR picture's anti-diagonal chain or
        # waits for its CABAC jobs (profiles/r02: k_intra_diag keeps < 20 % of the CUs busy); two ranks sharing one GPU reached 5674 fps against ~5000 for
        # one (gpurun_out/r2_ranks2b.json), and short clips (<= 4 GOPs: one chunk) gain the most.  Worker k still drives device k % n_devices.
        self.max_workers = max_workers or max(1, 2 * self.n_devices) or 1
        self.kw = dict(debug=debug, skip_validator=skip_validator, force_cpu=force_cpu, force_gpu=force_gpu)
        self.csv_path = Path(csv_path) if csv_path else self.out_dir / 'transcode_log.csv'
        self.on_progress, self.on_finished, self.convert = on_progress, on_finished, convert
        self.use_processes = (convert is convert_video and self.n_devices > 0) if use_processes is None else bool(use_processes)
        self.results: List[Dict] = []
        self._lock = threading.Lock()
        self._active: Dict[int, threading.Event] = {}
        self._threads: List[threading.Thread] = []
        self._procs: list = []

    # -- process mode: one spawned process per worker slot, the parent dispatches
    def _spawn(self, k: int):
        device = k % self.n_devices if self.n_devices else None
        ref = (self.convert.__module__, self.convert.__name__)
        p = self._ctx.Process(target=_proc_worker, args=(k, device, self._task_q[k], self._result_q, self._stop_ev[k], str(self.out_dir), self.kw, ref), daemon=True)
        p.start()
        return p

    def _start_processes(self, n: int):
        self._ctx = ctx = mp.get_context('spawn')
        self._result_q = ctx.Queue()
        self._task_q = [ctx.Queue() for _ in range(n)]
        self._stop_ev = [ctx.Event() for _ in range(n)]
        self._slot_file: Dict[int, str] = {}          # what every busy slot is working on (a worker that dies takes its file with it: the log names it)
        for k in range(n):
            self._procs.append(self._spawn(k))
        t = threading.Thread(target=self._dispatch, args=(n,), daemon=True)
        self._threads.append(t)
        t.start()

    def _hand_out(self, k: int) -> bool:
        """next queued file -> slot k (lock held); False when the queue is empty (the slot's worker is told to leave)"""
        if not self.queue:
            self._active.pop(k, None)
            self._slot_file.pop(k, None)
            self._task_q[k].put(None)
            return False
        f = str(self.queue.popleft())
        self._stop_ev[k].clear()                      # a stop that concerned the slot's previous file; cleared HERE, before the hand-over
        self._slot_file[k] = f
        self._active[k] = self._stop_ev[k]
        self._task_q[k].put(f)
        return True

    def _dispatch(self, n: int):
        busy = 0
        with self._lock:
            for k in range(n):
                busy += self._hand_out(k)
        last_check = time.time()
        while busy:
            msg = None
            try:
                msg = self._result_q.get(timeout=0.5)
            except Exception:
                pass
            if time.time() - last_check >= 0.5 or msg is None:
                # dead workers are looked for on a timer, whatever the message traffic of the others.  A process that died takes its file with it:
                # FAILED / UNKNOWN under the file's real name (like a raised exception, gui/worker.py:43-52), and the slot gets a fresh process
                # so the rest of the queue is still coded
                last_check = time.time()
                for k, p in enumerate(self._procs):
                    if p.is_alive() or k not in self._active:
                        continue
                    with self._lock:
                        lost = self._slot_file.pop(k, '?')
                        self._active.pop(k, None)
                        logger.error('[ERROR] worker %d died (exit %s) while coding %s', k, p.exitcode, lost)
                        self.results.append({'file': Path(lost).name, 'status': 'FAILED', 'quality': None, 'retries': 0, 'method': 'UNKNOWN', 'hdr': False,
                                             'seconds': '', 'device': ''})
                        self.save_csv()
                        busy -= 1
                        if self.queue:
                            self._task_q[k] = self._ctx.Queue()          # the dead process may have left the old one half-read
                            self._procs[k] = self._spawn(k)
                            busy += self._hand_out(k)
            if msg is None:
                continue
            if msg[0] == 'progress':
                if self.on_progress:
                    try:
                        self.on_progress(msg[2], msg[3], msg[4])
                    except Exception:
                        logger.debug('on_progress raised', exc_info=True)
                continue
            _, k, res = msg
            with self._lock:
                self.results.append(res)
                self.save_csv()
                if not self._hand_out(k):
                    busy -= 1
            if self.on_finished:
                try:
                    self.on_finished(res)
                except Exception:
                    logger.debug('on_finished raised', exc_info=True)

    # -- reference: MainWindow.start_batch / _start_next_worker / on_finished
    def _worker(self, slot: int):
        while True:
            with self._lock:
                if not self.queue:
                    self._active.pop(slot, None)
                    return
                f = self.queue.popleft()
                ev = threading.Event()
                self._active[slot] = ev
            device = slot % self.n_devices if self.n_devices else None
            t0 = time.time()
            try:
                kw = dict(self.kw)
                if device is not None:
                    kw['device'] = device
                res = self.convert(f, self.out_dir, progress_callback=self.on_progress, stop_event=ev, **kw)
            except Exception as exc:            # gui/worker.py:43-52: last-resort FAILED/UNKNOWN result
                logger.error('[ERROR] %s: %s', f.name, exc)
                res = {'file': f.name, 'status': 'FAILED', 'quality': None, 'retries': 0, 'method': 'UNKNOWN', 'hdr': False}
            res = dict(res)
            res['seconds'] = round(time.time() - t0, 3)
            res['device'] = device if res.get('method') == 'MI355X' else ''
            with self._lock:
                self.results.append(res)
                self.save_csv()
            if self.on_finished:
                try:
                    self.on_finished(res)
                except Exception:
                    logger.debug('on_finished raised', exc_info=True)

    def start(self):
        self.out_dir.mkdir(parents=True, exist_ok=True)
        n = min(self.max_workers, len(self.queue))
        if self.use_processes and n:
            self._start_processes(n)
            return self
        for k in range(n):
            t = threading.Thread(target=self._worker, args=(k,), daemon=True)
            self._threads.append(t)
            t.start()
        return self

    def stop_all(self, cancel_queued: bool = False):
        """Signal the active workers (reference semantics); cancel_queued=True also empties the queue."""
        with self._lock:
            if cancel_queued:
                self.queue.clear()
            for ev in self._active.values():
                ev.set()

    def wait(self) -> List[Dict]:
        for t in self._threads:
            t.join()
        for p in self._procs:
            p.join(timeout=10)
        return self.results

    def save_csv(self):
        with open(self.csv_path, 'w', newline='', encoding='utf-8') as f:
            w = csv.DictWriter(f, fieldnames=CSV_FIELDS + EXTRA_FIELDS, extrasaction='ignore')
            w.writeheader()
            for r in self.results:
                w.writerow(r)


def batch_convert(input_dir: Path, output_dir: Path, max_workers: Optional[int] = None, **kw) -> List[Dict]:
    """One-call form (the monolith's batch_convert, apple_hevc_batch.py:861-882)."""
    return BatchRunner(scan_inputs(input_dir), output_dir, max_workers=max_workers, **kw).start().wait()


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description='Headless Apple-HEVC batch transcode (MI355X / libx265)')
    ap.add_argument('-i', '--input', dest='input_dir', required=True)
    ap.add_argument('-o', '--output', dest='output_dir', required=True)
    ap.add_argument('--max-workers', type=int, default=None)
    ap.add_argument('--force-cpu', action='store_true')
    ap.add_argument('--force-gpu', action='store_true')
    ap.add_argument('--skip-validator', action='store_true')
    ap.add_argument('--debug', action='store_true')
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if a.debug else logging.INFO)
    res = batch_convert(Path(a.input_dir), Path(a.output_dir), a.max_workers, debug=a.debug, skip_validator=a.skip_validator,
                        force_cpu=a.force_cpu, force_gpu=a.force_gpu)
    ok = sum(r['status'] == 'SUCCESS' for r in res)
    print(f'{ok}/{len(res)} files converted; log: {Path(a.output_dir) / "transcode_log.csv"}')
    return 0 if ok == len(res) else 1


if __name__ == '__main__':
    raise SystemExit(main())
