#!/usr/bin/env python3
"""Golden coded pictures of this build's own encoder decisions: `python tests/golden/make_stream_goldens.py` rewrites tests/golden/streams.json.

For a few small clips the ORACLE analyses every picture (oracle/hevc_oracle.c: the decisions the kernels must reproduce), the product's host coder
(libmihevc.so, no device needed) turns the symbols into slice NAL units, and the SHA-256 of every coded picture and of every reconstruction goes into
the fixture.  tests/test_golden_streams.py recomputes them on the CPU (any change of an encoder decision, of the rate-free bitstream or of the filters
shows up as a changed hash and has to be re-blessed here, in the same commit) and, on a GPU box, encodes the same clips with a real session: the
MI355X path must produce byte-identical pictures.  These pin the build against ITSELF over time; parity with libx265 stays unpinned."""
import ctypes as C
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hevc_amd import _lib                 # noqa: E402
from oracle import oracle as O            # noqa: E402
from tests import util                    # noqa: E402

# name: width, height, bit depth, pictures, keyint, lanes, qp, me range, level, extra config knobs
CASES = {
    "p96x80_8bit": (96, 80, 8, 6, 4, 2, 27, 8, 93, {}),
    "p72x104_10bit_hdr": (72, 104, 10, 5, 3, 2, 24, 8, 150, {"hdr10": 1, "colour_primaries": 9, "transfer": 16, "matrix": 9}),
    "p544x160_tiles_nxn_cg": (544, 160, 8, 5, 3, 2, 30, 12, 120, {"intra_nxn": 1, "rdo_cg": 5}),
    "p130x70_window": (130, 70, 8, 4, 2, 3, 33, 15, 93, {"chroma_modes": 0, "rdo_zero": 0}),
}


def config(name):
    w, h, bd, n, keyint, lanes, qp, rng, level, extra = CASES[name]
    cfg = _lib.default_config()
    cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight = w, h, bd, keyint, 1, lanes
    cfg.qp, cfg.me_range, cfg.level_idc, cfg.scenecut, cfg.aud, cfg.hrd = qp, rng, level, 0, 0, 0
    for k, v in extra.items():
        setattr(cfg, k, v)
    return cfg


def frames(name):
    w, h, bd, n = CASES[name][:4]
    return [util.synth_frame(h, w, seed=77, shift=(2 * i, i // 2), bit_depth=bd) for i in range(n)]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def frame_hash(f):
    return sha(f.y.tobytes() + f.u.tobytes() + f.v.tobytes())


def oracle_pictures(name):
    """[(coded picture bytes, reconstruction)] in stream order, from the oracle analysis + the host coder"""
    w, h, bd, n, keyint, lanes, qp, rng, level, extra = CASES[name]
    cfg = config(name)
    lib = _lib.load()
    cw, ch = (w + 7) & ~7, (h + 7) & ~7
    idr = util.idr_positions(n, keyint, lanes)
    prm_i, prm_p = O.default_params(max(0, qp - 3), bd, rng), O.default_params(qp, bd, rng)
    prm_i.tile_cols, prm_i.tile_rows = _lib.tile_grid(cfg)
    for prm in (prm_i, prm_p):
        prm.intra_nxn, prm.chroma_modes = cfg.intra_nxn, cfg.chroma_modes
    prm_p.intra_in_p, prm_p.pre_search, prm_p.rdo_zero, prm_p.rdo_cg = cfg.intra_in_p, cfg.pre_search, cfg.rdo_zero, cfg.rdo_cg
    buf = (C.c_uint8 * (4 << 20))()
    out, ref, poc, prev_pad = [], None, 0, None
    import numpy as np
    for i, f in enumerate(frames(name)):
        pad = O.Frame(np.pad(f.y, ((0, ch - h), (0, cw - w)), mode="edge"), np.pad(f.u, ((0, (ch - h) // 2), (0, (cw - w) // 2)), mode="edge"),
                      np.pad(f.v, ((0, (ch - h) // 2), (0, (cw - w) // 2)), mode="edge"))
        intra = i in idr
        poc = 0 if intra else poc + 1
        prm = prm_i if intra else prm_p
        a = O.analyze_intra(pad, prm) if intra else O.analyze_inter(pad, ref, prm, centers=O.search_centres(pad, prev_pad, bd) if cfg.pre_search else None)
        prev_pad = pad
        ref, sao = O.sao(pad, O.deblock(a.rec, a.cu, bd), prm)
        nb = lib.mihevc_encode_picture_host(C.byref(cfg), 2 if intra else 1, poc, prm.qp, util.ptr(a.cu), util.ptr(a.coef_y), util.ptr(a.coef_u),
                                            util.ptr(a.coef_v), util.ptr(sao), buf, len(buf))
        assert nb > 0, nb
        out.append((bytes(buf[:nb]), ref))
    return out


def compute(name):
    pics = oracle_pictures(name)
    return {"pictures": [sha(p) for p, _ in pics], "bytes": [len(p) for p, _ in pics], "recon": [frame_hash(r) for _, r in pics]}


if __name__ == "__main__":
    out = {name: compute(name) for name in CASES}
    (Path(__file__).parent / "streams.json").write_text(json.dumps(out, indent=1) + "\n")
    for k, v in out.items():
        print(k, v["bytes"])
