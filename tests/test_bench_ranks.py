"""CPU: bench.py's own multi-rank path (`--gpus N` was ignored and the N > 1 branch had never executed).

`--stub` replaces the encoder by a sleep so that the parts that matter here run without a GPU: the parent spawning one child per
rank before any GPU call, the gloo rendezvous on 127.0.0.1 (no RCCL: nothing is exchanged on the data path), barrier + MAX-over-ranks
timing, rank 0 printing ONE JSON line with n_gpus = N and the per-rank rates — both when bench.py spawns the ranks itself and when
the driver starts them with torch.distributed.run."""
import json
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]         # nothing else may reach stdout (gloo's connection notes go to stderr)
    assert len(lines) == 1 and lines[0].startswith("{"), stdout
    return json.loads(lines[0])


def test_bench_spawns_one_process_per_rank():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--stub"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _one_json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and len(out["per_rank_fps"]) == 2
    # the stub's rank 1 sleeps twice as long as rank 0: the job time is the MAX over ranks, the aggregate counts both ranks' frames.  Structural checks
    # only (a wall-clock ratio failed once on a loaded box): the line's own fields must agree with each other
    assert out["per_rank_fps"][0] > out["per_rank_fps"][1]
    frames = 2 * 3 * 300                                              # ranks x steps x the default 300 frames
    assert abs(out["value"] - frames / (3 * out["ms_per_step"] / 1e3)) <= 0.01 * out["value"]      # value = all ranks' frames / the job's time
    assert out["value"] <= 2 * out["per_rank_fps"][1] * 1.001         # the job is never faster than its slowest rank allows


def test_bench_under_the_drivers_launcher():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--stub"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _one_json_line(p.stdout)
    assert out["n_gpus"] == 2 and len(out["per_rank_fps"]) == 2


def test_bench_refuses_more_ranks_than_devices():
    """no MI355X in this container: --gpus 2 must fail loudly before anything is spawned (and N = 1 must refuse to run without a GPU)"""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "MI355X visible" in p.stderr
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in p.stderr


def test_world_size_must_match_gpus():
    import os
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--stub"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
