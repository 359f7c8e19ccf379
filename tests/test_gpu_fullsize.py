"""GPU (-m gpu): BASELINE configs[3] and configs[4] at FULL size on the one device a test box has.

configs[4] — 7680x4320 Main10 HDR10, level 6 (reference operating point: core/transcoder.py:263-354 -> crf 20, vbv 47040 / 56448, keyint 60;
HDR10 set of core/utils.py:58-69): (a) every picture as 8 slices of CTU rows, all eight band sessions on device 0 — what the 8-GPU split runs
per device, with the row exchange between the bands going through ordinary device pointers instead of xGMI peer mappings; (b) per-stage HIP vs oracle on the unsliced 4320p pictures (I + P, incl. the integer-search dump).
configs[3] — one clip per GPU: `bench.py --gpus 2` with both ranks sharing device 0 (MIHEVC_BENCH_SHARE_GPU=1: the rank / rendezvous / MAX-over-ranks
path with REAL encoders; the number it prints is no scaling figure and says so), and the headless batch queue over eight 1080p clips with two
worker processes.  A node with 8 devices is not in reach of the test box: "unmeasured on more than one GPU" stays true."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from tests import util
from tests.test_gpu_configs import clip_frames, operating_point, stage_parity

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from hevc_amd import _lib
    L = _lib.load()
    assert L.mihevc_device_count() >= 1, "no gfx950 device visible: the GPU tests need an MI355X"
    return L


@pytest.fixture(scope="module")
def api(lib):
    return util.StageApi(lib, "mihevc_k_", device=0)


@pytest.fixture(scope="module")
def frames_4320p():
    return clip_frames(7680, 4320, 10, 3)


def test_config5_4320p_main10_hdr10_as_8_slices_on_device_0(lib, frames_4320p):
    from hevc_amd.encoder import SlicedEncoder, slice_rows
    n = 3
    cfg, (crf, maxrate, bufsize, gop, level) = operating_point(7680, 4320, True, 120)
    assert (crf, maxrate, bufsize, gop, level) == (20, 47040, 56448, 60, "6") and cfg.bit_depth == 10          # SURVEY App. A golden
    sl = SlicedEncoder(cfg, [0] * 8, keep_recon=True)
    try:
        assert sl.halo and sl.rows == slice_rows(4320, 8) == [17] * 7 + [16]
        got = []
        for (y, u, v), _ in frames_4320p:
            sl.send(y, u, v)
            got += sl.ready()
        got += sl.finish()
        recs = [O.Frame(*sl.recon(i)) for i in range(n)]
        stats = sl.stats()
        infos = [[e.frame_info(i) for i in range(n)] for e in sl._encs]
        cfgs = sl._cfgs
    finally:
        sl.close()
    # the eight bands exchange rows (cfg.slice_halo) and share one rate plan: same type and QP in every band, and the stacked reconstructions of the first
    # two pictures equal the oracle's WHOLE-PICTURE pipeline replayed with those QPs (tests/test_sliced_cpu.py halo_pipeline), bit for bit
    from tests.test_gpu_configs import session_params
    from tests.test_sliced_cpu import halo_pipeline
    assert all([(q, t) for q, t, _ in inf] == [(q, t) for q, t, _ in infos[0]] for inf in infos)
    srcs = [f for _, f in frames_4320p[:2]]
    for i, (intra, a, sao, ref) in enumerate(halo_pipeline(srcs, sl.rows, cfgs, None, None, None, 10, prm_of=lambda i, intra: session_params(lib, cfgs[0], infos[0][i][0], False)[0], idr_at={0})):
        assert recs[i].same(ref), f"picture {i}: the bands' reconstructions != the whole-picture pipeline"
    assert [p for _, p, _ in got] == list(range(n)) and [k for _, _, k in got] == [True, False, False]
    dec, info = O.decode(b"".join(d for d, _, _ in got))
    assert len(dec) == n and info["count.slices"] == 8 * n and info["count.aud"] == n
    assert (info["width"], info["conf_width"], info["conf_height"], info["bit_depth"]) == (7680, 7680, 4320, 10)
    assert info["sps.profile_idc"] == 2 and info["sps.level_idc"] == 180 and info["sps.tier_flag"] == 0
    # HDR10 signalling as the reference asks libx265 for it (core/utils.py:58-69)
    assert (info["vui.colour_primaries"], info["vui.transfer"], info["vui.matrix"], info["vui.full_range"]) == (9, 16, 9, 0)
    assert (info["sei.mdcv.gx"], info["sei.mdcv.gy"], info["sei.mdcv.bx"], info["sei.mdcv.by"], info["sei.mdcv.rx"], info["sei.mdcv.ry"]) == (13250, 34500, 7500, 3000, 34000, 16000)
    assert (info["sei.mdcv.wpx"], info["sei.mdcv.wpy"], info["sei.mdcv.max_lum"], info["sei.mdcv.min_lum"]) == (15635, 16450, 10000000, 50)
    assert (info["sei.cll.max_cll"], info["sei.cll.max_fall"]) == (1000, 400)
    assert info["vui.hrd_present"] == 1 and info["count.sei_bp"] == 1 and info["count.sei_pt"] == n
    rate = (info["hrd.bit_rate_value_minus1"] + 1) << (6 + info["hrd.bit_rate_scale"])
    cpb = (info["hrd.cpb_size_value_minus1"] + 1) << (4 + info["hrd.cpb_size_scale"])
    assert abs(rate - maxrate * 1000) <= maxrate * 10 and abs(cpb - bufsize * 1000) <= bufsize * 10
    for i in range(n):
        d = O.Frame(dec[i].y[:recs[i].y.shape[0]], dec[i].u[:recs[i].u.shape[0]], dec[i].v[:recs[i].v.shape[0]])
        assert d.same(recs[i]), f"picture {i}: decoded picture != the bands' reconstructions stacked"
        assert util.psnr(dec[i].y[:4320], frames_4320p[i][0][0], peak=1023.0) > 30.0          # sanity only: three pictures under a rate cap, grain of sigma 8 LSB in the source (measured 32.7 dB)
    assert all(st.frames_out == n for st in stats)


def test_config5_4320p_unsliced_stage_parity_i_p(lib, api, frames_4320p):
    from hevc_amd import _lib
    cfg, (crf, *_rest) = operating_point(7680, 4320, True, 120)
    assert _lib.tile_grid(cfg) == (20, 22)                   # Table A.8 at level 6
    stage_parity(lib, api, cfg, frames_4320p[:2], crf + 2)


def test_config4_bench_two_ranks_share_device_0():
    """the real bench.py at world size 2 with real encoders: each rank its own clip (seed = rank), gloo barrier, MAX over ranks, one JSON line"""
    env = dict(os.environ, MIHEVC_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["shared_gpu_rehearsal"] is True
    assert len(out["per_rank_fps"]) == 2 and all(f > 100 for f in out["per_rank_fps"])
    assert out["stream_ok"] is True                           # both ranks decoded their own 300-picture stream with the oracle decoder
    assert abs(out["value"] - 2 * 300 / (out["ms_per_step"] / 1e3)) / out["value"] < 0.01          # whole-job rate: both clips over the slowest rank's time
    assert "1920x1080" in out["config"]["workload"] and out["dtype"] == "u8"


def test_config4_batch_of_eight_1080p_clips_two_worker_processes(lib, tmp_path):
    """BASELINE configs[3] as the reference runs it (gui/mainwindow.py:289-301: a queue of files, N workers): eight 1080p clips through the headless
    queue with two worker processes on the device; every file SUCCESS by the native path, the CSV carries the reference's six columns"""
    import csv
    from hevc_amd import batch, mp4, yuvio
    files = []
    for i in range(8):
        p = tmp_path / f"clip{i}.y4m"
        yuvio.write_y4m(p, yuvio.SyntheticClip("motion", i, 1920, 1080, 6).frames(), 1920, 1080, 30)
        files.append(p)
    out = tmp_path / "out"
    r = batch.BatchRunner(files, out, max_workers=2, skip_validator=True).start()
    assert r.use_processes
    res = r.wait()
    from hevc_amd.probe import probe_media
    from hevc_amd.transcoder import calculate_dynamic_values
    crf = calculate_dynamic_values(probe_media(files[0]))[0]          # the reference's CRF for a clip this short (its "motion density" term sees 6 frames)
    assert len(res) == 8 and all(x["status"] == "SUCCESS" and x["method"] == "MI355X" and x["quality"] == crf for x in res), res
    rows = list(csv.DictReader(open(out / "transcode_log.csv")))
    assert len(rows) == 8 and list(rows[0])[:6] == ["file", "status", "quality", "retries", "method", "hdr"]
    assert sorted(x["file"] for x in rows) == sorted(f.name for f in files)
    for f in files:
        top = mp4.parse_boxes((out / (f.stem + ".mp4")).read_bytes())
        assert [b[0] for b in top] == ["ftyp", "moov", "mdat"]


@pytest.mark.parametrize("w,h,hdr,qp", [(1920, 1080, False, 27), (3840, 2160, True, 29)])
def test_b_picture_stage_parity_at_full_size(lib, api, w, h, hdr, qp):
    """cfg.bframes at the reference's picture sizes: I0, P2 (predicted over two pictures) and b1 between their reconstructions on the bench clip — k_inter_ctu_b with both
    integer searches, the list-1 refinement and the bi-prediction trial against orc_analyze_b_frame, at 1920x1080 8 bit and 3840x2160 Main10 (the small-size B tests:
    tests/test_gpu_parity.py, tests/test_gpu_bframes.py)"""
    from tests.test_gpu_configs import session_params
    cfg, _ = operating_point(w, h, hdr, 60)
    bd = cfg.bit_depth
    frames = [f for _, f in clip_frames(w, h, bd, 3)]
    prm_i, cp_i = session_params(lib, cfg, qp - 3, True)
    prm_p, cp_p = session_params(lib, cfg, qp, False)
    prm_b, cp_b = session_params(lib, cfg, qp + 2, False)
    a0 = O.analyze_intra(frames[0], prm_i)
    r0, _ = O.sao(frames[0], O.deblock(a0.rec, a0.cu, bd), prm_i)
    c2 = O.search_centres(frames[2], frames[0], bd) if cfg.pre_search else None
    a2 = O.analyze_inter(frames[2], r0, prm_p, centers=c2)
    g2 = api.inter(frames[2], r0, cp_p, centers=c2)
    assert util.same_analysis(a2, g2), "anchor over two pictures: " + util.describe_diff(a2, g2)
    r2, _ = O.sao(frames[2], O.deblock(a2.rec, a2.cu, bd), prm_p)
    c0, c1 = (O.search_centres(frames[1], frames[0], bd), O.search_centres(frames[1], frames[2], bd)) if cfg.pre_search else (None, None)
    want = O.analyze_b(frames[1], r0, r2, prm_b, c0, c1, dump_me=True)
    got = api.b(frames[1], r0, r2, cp_b, c0, c1)
    assert np.array_equal(want.me[0], got.me[0]) and np.array_equal(want.me[1], got.me[1]), "integer searches differ"
    assert util.same_analysis(want, got), "B picture: " + util.describe_diff(want, got)
    assert (want.cu["flags"] & 32).any() and ((want.cu["flags"] & 96) == 32).any(), "no bi-predicted CU: the case is not exercised"
