#!/usr/bin/env python3
"""tools/rd_curve.py — rate-distortion points of the native encoder and Bjontegaard deltas between two builds / settings.

The quality half of BASELINE.json's metric asks for PSNR-Y parity with libx265 at matched bitrate (reference operating point:
core/transcoder.py:398-411, `preset=slow`).  libx265 does not exist on this pool, so the instrument that CAN be kept is a reproducible
RD curve of this encoder: 4 fixed QPs x {motion, stress, bars} -> (kb/s, PSNR-Y/U/V) and the BD-rate of any change against a stored
run.  Every coding tool added from round 3 on states its BD-rate from this script (JSON under profiles/).

    python tools/rd_curve.py --out profiles/r03_rd_base.json                       # needs an MI355X (no CPU fallback)
    python tools/rd_curve.py --out profiles/r03_rd_x.json --against profiles/r03_rd_base.json --set bframes=1
    python tools/rd_curve.py --compare profiles/a.json profiles/b.json             # no GPU: BD-rate of b against a

Fixed-QP runs (cfg.qp = QP for P pictures, IDR 3 below, the session's rule), no VBV: the curve is the encoder's, not the rate controller's.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

QPS = (22, 27, 32, 37)
CLIPS = ("motion", "stress", "bars")


def bd_rate(base, test):
    """Bjontegaard delta rate (%) of `test` against `base`: lists of (kbps, psnr) with >= 4 points each.  Cubic fit of log-rate over PSNR,
    integrated over the common PSNR interval; negative = `test` needs fewer bits for the same PSNR."""
    def fit(pts):
        pts = sorted(pts, key=lambda p: p[1])
        r, d = np.log(np.array([p[0] for p in pts])), np.array([p[1] for p in pts])
        return np.polyfit(d, r, min(3, len(pts) - 1)), d.min(), d.max()
    pa, lo_a, hi_a = fit(base)
    pb, lo_b, hi_b = fit(test)
    lo, hi = max(lo_a, lo_b), min(hi_a, hi_b)
    if hi <= lo:
        return None
    ia, ib = np.polyint(pa), np.polyint(pb)
    avg = ((np.polyval(ib, hi) - np.polyval(ib, lo)) - (np.polyval(ia, hi) - np.polyval(ia, lo))) / (hi - lo)
    return float((np.exp(avg) - 1) * 100)


def bd_psnr(base, test):
    """Bjontegaard delta PSNR (dB) at equal rate: cubic fit of PSNR over log-rate"""
    def fit(pts):
        pts = sorted(pts)
        r, d = np.log(np.array([p[0] for p in pts])), np.array([p[1] for p in pts])
        return np.polyfit(r, d, min(3, len(pts) - 1)), r.min(), r.max()
    pa, lo_a, hi_a = fit(base)
    pb, lo_b, hi_b = fit(test)
    lo, hi = max(lo_a, lo_b), min(hi_a, hi_b)
    if hi <= lo:
        return None
    ia, ib = np.polyint(pa), np.polyint(pb)
    return float(((np.polyval(ib, hi) - np.polyval(ib, lo)) - (np.polyval(ia, hi) - np.polyval(ia, lo))) / (hi - lo))


def compare(a, b):
    out = {}
    for clip in a["clips"]:
        if clip not in b["clips"]:
            continue
        pa = [(p["kbps"], p["psnr_y"]) for p in a["clips"][clip]["points"]]
        pb = [(p["kbps"], p["psnr_y"]) for p in b["clips"][clip]["points"]]
        out[clip] = {"bd_rate_pct": None if bd_rate(pa, pb) is None else round(bd_rate(pa, pb), 2),
                     "bd_psnr_db": None if bd_psnr(pa, pb) is None else round(bd_psnr(pa, pb), 3)}
    vals = [v["bd_rate_pct"] for v in out.values() if v["bd_rate_pct"] is not None]
    out["mean_bd_rate_pct"] = round(float(np.mean(vals)), 2) if vals else None
    return out


def run(args):
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    from hevc_amd.yuvio import SyntheticClip
    lib = _lib.load()
    if lib.mihevc_device_count() < 1:
        raise SystemExit("rd_curve.py needs an MI355X (the native path has no CPU fallback); --compare works without one")
    W, H, N, bd = args.width, args.height, args.frames, args.bit_depth
    peak = float((1 << bd) - 1)
    res = {"width": W, "height": H, "frames": N, "bit_depth": bd, "qps": list(QPS), "settings": dict(kv.split("=") for kv in args.set), "clips": {}}
    for name in args.clips:
        clip = SyntheticClip(name, 0, W, H, N, bit_depth=bd)
        frames = [clip.frame(i) for i in range(N)]
        pts = []
        for qp in QPS:
            cfg = _lib.default_config()
            cfg.width, cfg.height, cfg.bit_depth, cfg.qp, cfg.keyint, cfg.min_keyint = W, H, bd, qp, args.keyint, max(2, args.keyint // 2)
            cfg.level_idc = 120 if W * H <= 1920 * 1088 else 150 if W * H <= 3840 * 2176 else 180
            cfg.gops_in_flight = max(1, min(8, -(-N // args.keyint)))
            for kv in args.set:
                k, v = kv.split("=")
                setattr(cfg, k, int(v))
            nbytes, t0 = 0, time.perf_counter()
            with Encoder(cfg, device=args.device) as enc:
                for i, (y, u, v) in enumerate(frames):
                    enc.send(y, u, v, pts=i)
                    nbytes += sum(len(p[0]) for p in enc.packets())
                enc.flush()
                nbytes += sum(len(p[0]) for p in enc.packets())
                st = enc.stats()
                cw, ch = enc.coded_size()
            dt = time.perf_counter() - t0
            assert st.frames_out == N
            npx = float(N * cw * ch)
            ps = [99.0 if s <= 0 else 10 * np.log10(peak * peak / (s / d)) for s, d in ((st.sse_y, npx), (st.sse_u, npx / 4), (st.sse_v, npx / 4))]
            pts.append({"qp": qp, "kbps": round(nbytes * 8 / (N / 30.0) / 1e3, 2), "psnr_y": round(float(ps[0]), 4), "psnr_u": round(float(ps[1]), 4),
                        "psnr_v": round(float(ps[2]), 4), "fps_host_buffers": round(N / dt, 1)})
            if cfg.bframes < 0:       # the adaptive decision of the last chunk: cost per CTU one / two pictures back, B pictures chosen
                pts[-1]["b_probe"] = [int(st.reserved[0]), int(st.reserved[1]), int(st.reserved[2])]
            print(f"{name:7s} qp {qp}: {pts[-1]['kbps']:10.1f} kb/s  {pts[-1]['psnr_y']:.3f} dB  ({pts[-1]['fps_host_buffers']:.0f} fps) {pts[-1].get('b_probe', '')}", file=sys.stderr)
        res["clips"][name] = {"points": pts}
    if args.against:
        base = json.load(open(args.against))
        res["against"] = {"file": str(args.against), **compare(base, res)}
        print(json.dumps(res["against"], indent=1), file=sys.stderr)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "clips"} | {"clips": {c: [(p["kbps"], p["psnr_y"]) for p in v["points"]] for c, v in res["clips"].items()}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--against", help="stored run to compute BD-rate against")
    ap.add_argument("--compare", nargs=2, metavar=("BASE", "TEST"), help="no encode: BD-rate of TEST against BASE")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--keyint", type=int, default=30)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--clips", nargs="+", default=list(CLIPS))
    ap.add_argument("--set", nargs="*", default=[], metavar="FIELD=INT", help="mihevc_config fields to override, e.g. rdo_cg=3 bframes=1")
    args = ap.parse_args()
    if args.compare:
        print(json.dumps(compare(json.load(open(args.compare[0])), json.load(open(args.compare[1]))), indent=1))
        return
    run(args)


if __name__ == "__main__":
    main()
