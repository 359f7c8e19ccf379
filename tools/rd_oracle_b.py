#!/usr/bin/env python3
"""CPU-only look at what B pictures buy before (and independent of) the device path: the oracle's analysis + the product's host coder on one closed GOP,
IPPP against I P b P b ..., four QPs, BD-rate by tools/rd_curve.py's routine.  Small pictures (the oracle is a scalar port): a direction, not the number
that goes into profiles/ (that comes from tools/rd_curve.py on the GPU).    python tools/rd_oracle_b.py [W H N]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from hevc_amd import _lib                     # noqa: E402
from hevc_amd.yuvio import SyntheticClip      # noqa: E402
from oracle import oracle as O                # noqa: E402
import rd_curve                               # noqa: E402
from tests import util                        # noqa: E402
from tests.test_bitstream_cpu import coding_order, make_cfg     # noqa: E402

W, H, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (416, 240, 17)
lib = _lib.load()
buf = (C.c_uint8 * (8 << 20))()


def run(srcs, qp, bframes, b_off=2):
    cfg = make_cfg(W, H, 8, bframes=bframes)
    order = coding_order(len(srcs)) if bframes else [(0, 2)] + [(i, 1) for i in range(1, len(srcs))]
    recs, last, bits, prev_src = {}, None, 0, {}
    for pos, st in order:
        src = srcs[pos]
        q = max(0, qp - 3) if st == 2 else qp + b_off if st == 0 else qp
        prm = O.default_params(q, 8, 15)
        prm.rdo_zero = prm.chroma_modes = 1
        if st == 2:
            prm.tile_cols, prm.tile_rows = _lib.tile_grid(cfg)
            a = O.analyze_intra(src, prm)
        elif st == 1:
            a = O.analyze_inter(src, recs[last], prm, centers=O.search_centres(src, srcs[last]))
        else:
            a = O.analyze_b(src, recs[pos - 1], recs[pos + 1], prm, O.search_centres(src, srcs[pos - 1]), O.search_centres(src, srcs[pos + 1]))
        rec, sao = O.sao(src, O.deblock(a.rec, a.cu, 8), prm)
        recs[pos] = rec
        if st != 0:
            last = pos
        m = lib.mihevc_encode_picture_host(C.byref(cfg), st, pos, prm.qp, util.ptr(a.cu), util.ptr(a.coef_y), util.ptr(a.coef_u), util.ptr(a.coef_v), util.ptr(sao), buf, len(buf))
        assert m > 0
        bits += 8 * m
    mse = np.mean([np.mean((recs[i].y.astype(np.float64) - srcs[i].y) ** 2) for i in range(len(srcs))])
    return bits / (len(srcs) / 30.0) / 1e3, 10 * np.log10(255.0 ** 2 / mse)


for pattern in ("motion", "stress"):
    clip = SyntheticClip(pattern, 0, W, H, N)
    srcs = [O.Frame(*clip.frame(i)) for i in range(N)]
    base = [run(srcs, qp, 0) for qp in rd_curve.QPS]
    for b_off in (1, 2, 3):
        test = [run(srcs, qp, 1, b_off) for qp in rd_curve.QPS]
        print(f"{pattern}: IPPP {[(round(k), round(p, 2)) for k, p in base]}  IbP(b +{b_off}) {[(round(k), round(p, 2)) for k, p in test]}  "
              f"BD-rate {rd_curve.bd_rate(base, test):+.2f} %  BD-PSNR {rd_curve.bd_psnr(base, test):+.3f} dB", flush=True)
