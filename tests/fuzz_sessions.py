"""Randomised end-to-end check on a GPU box: random geometry / bit depth / GOP structure / analysis knobs / rate control, every stream must
decode (oracle decoder) to the encoder's own reconstruction.  Usage: python tests/fuzz_sessions.py [iterations] [seed]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np                                   # noqa: E402
from hevc_amd import _lib                            # noqa: E402
from hevc_amd.encoder import Encoder                 # noqa: E402
from oracle import oracle as O                       # noqa: E402
from tests import util                               # noqa: E402



try:                       # the device-frame cases hold their planes in torch tensors: bring torch's HIP runtime up before libmihevc's
    import torch
    torch.cuda.init()
except Exception:          # noqa: BLE001
    torch = None
LARGE = False     # third argument "large": pictures up to 2160p, longer clips (several chunks)


def run(iters, seed, verbose=True, large=None):
    """returns the descriptions of the failing cases"""
    global LARGE
    if large is not None:
        LARGE = large
    rng = np.random.default_rng(seed)
    failed = []
    for it in range(iters):
        desc, ok = one_case(rng, it)
        if not ok:
            failed.append(desc)
            print("FAIL", desc, flush=True)
        elif verbose and (LARGE or it % 10 == 0):
            print("ok  ", desc, flush=True)
    return failed


def one_case(rng, it):
    if True:
        w, h = int(rng.integers(8, 200)) * 2, int(rng.integers(8, 120)) * 2
        if rng.random() < 0.25:
            w = int(rng.integers(128, 330)) * 2          # wide enough for IDR tiles
        if LARGE:
            w, h = int(rng.integers(150, 1000)) * 2, int(rng.integers(100, 560)) * 2
            if rng.random() < 0.15:
                w, h = 3840, 2160
        bd = 10 if rng.random() < 0.3 else 8
        keyint, lanes, n = int(rng.integers(1, 8)), int(rng.integers(1, 5)), int(rng.integers(1, 16))
        if LARGE:
            n = int(rng.integers(1, 40)) if w < 3000 else int(rng.integers(1, 8))
        cfg = _lib.default_config()
        cfg.width, cfg.height, cfg.bit_depth, cfg.keyint, cfg.min_keyint, cfg.gops_in_flight = w, h, bd, keyint, int(rng.integers(1, 4)), lanes
        cfg.me_range = int(rng.choice([4, 8, 15, 24]))
        cfg.level_idc = int(rng.choice([93, 120, 150, 180]))
        for name in ("intra_nxn", "intra_in_p", "chroma_modes", "rdo_zero", "pre_search", "intra_tiles", "sao", "aud", "hrd", "repeat_headers", "gop_balance", "scenecut"):
            setattr(cfg, name, int(rng.random() < 0.5))
        cfg.rdo_cg = int(rng.choice([0, 2, 5, 8]))
        cfg.bframes = int(rng.choice([0, 0, 1, -1]))          # round 3: B pictures (fixed / decided by the probe), P pictures as tiles (forced: at least 2 x 2 where the level allows)
        cfg.p_tiles = int(rng.choice([-1, 0, 1]))
        if rng.random() < 0.5:
            cfg.qp = int(rng.integers(10, 45))
        else:
            cfg.crf, cfg.qp = int(rng.integers(14, 30)), -1
            cfg.vbv_maxrate_kbps = int(rng.integers(50, 4000)); cfg.vbv_bufsize_kbits = int(cfg.vbv_maxrate_kbps * 1.2)
        cw, ch = (w + 7) & ~7, (h + 7) & ~7
        detail = min(cw, ch) >= 64
        frames = []
        for i in range(n):
            f = util.synth_frame(ch, cw, seed=int(rng.integers(0, 1000)) if rng.random() < 0.15 else 7, shift=(int(rng.integers(-3, 4)) * i, i), bit_depth=bd, detail=detail)
            frames.append((f.y[:h, :w].copy(), f.u[:h // 2, :w // 2].copy(), f.v[:h // 2, :w // 2].copy()))
        dev_path, extra_y, extra_c = bool(rng.random() < 0.3) and torch is not None, int(rng.integers(0, 9)), int(rng.integers(0, 5))
        desc = f"#{it} {'dev ' if dev_path else ''}{w}x{h} bd{bd} keyint{keyint} lanes{lanes} n{n} qp{cfg.qp} crf{cfg.crf} vbv{cfg.vbv_maxrate_kbps} R{cfg.me_range} lvl{cfg.level_idc} " \
               f"nxn{cfg.intra_nxn} ip{cfg.intra_in_p} cm{cfg.chroma_modes} rz{cfg.rdo_zero} cg{cfg.rdo_cg} ps{cfg.pre_search} tiles{cfg.intra_tiles} sao{cfg.sao} aud{cfg.aud} hrd{cfg.hrd} b{cfg.bframes} pt{cfg.p_tiles}"
        only = os.environ.get("FUZZ_ONLY")               # "23,27": run just these cases (the generator still draws every case)
        if only and it not in [int(x) for x in only.split(",")]:
            return desc, True
        for kv in filter(None, os.environ.get("FUZZ_SET", "").split(",")):      # "pre_search=0,intra_nxn=1": override knobs
            setattr(cfg, kv.split("=")[0], int(kv.split("=")[1]))
        try:
            stream = b""
            with Encoder(cfg, device=0, keep_recon=True) as enc:
                if dev_path:        # frames already in HBM, rows padded to an arbitrary pitch (in samples)
                    keep = []
                    for y, u, v in frames:
                        t = []
                        for pl, ex in ((y, extra_y), (u, extra_c), (v, extra_c)):
                            host = np.zeros((pl.shape[0], pl.shape[1] + ex), pl.dtype)
                            host[:, :pl.shape[1]] = pl
                            t.append(torch.from_numpy(host).cuda())
                        keep.append(t)
                    torch.cuda.synchronize()
                    for i, t in enumerate(keep):
                        enc.send_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), w + extra_y, w // 2 + extra_c, pts=i)
                else:
                    for y, u, v in frames:
                        enc.send(y, u, v)
                enc.flush()
                for data, pts, key in enc.packets():
                    stream += data
                recs = [O.Frame(*enc.recon(i)) for i in range(n)]
                out = enc.stats().frames_out
            dec, info = O.decode(stream)
            ok = out == n and len(dec) == n and all(d.same(r) for d, r in zip(dec, recs)) and (info["conf_width"], info["conf_height"]) == (w, h)
            if not ok:
                bad = [i for i, (d, r) in enumerate(zip(dec, recs)) if not d.same(r)]
                desc += f" out={out} decoded={len(dec)} differing pictures {bad[:8]}"
                if bad:
                    ys, xs = np.nonzero(dec[bad[0]].y != recs[bad[0]].y)
                    desc += f" luma diffs {len(ys)} first (x={xs[0]}, y={ys[0]})" if len(ys) else " chroma only"
        except Exception as exc:            # noqa: BLE001
            ok = False
            desc += f" EXC {type(exc).__name__}: {exc}"
    return desc, ok


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"
    bad = run(iters, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{iters - len(bad)}/{iters} passed")
    sys.exit(1 if bad else 0)
