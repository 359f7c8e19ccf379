#!/usr/bin/env python3
"""Driver of tests/sanitize_cpu.sh: steps the kernel sources (tests/emu) against the oracle on a few pictures with the emulator library given in
EMU_LIB (an ASAN/UBSAN or TSAN build).  EMU_WAVES set: four wave threads with real barriers.  Test infrastructure, not product."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np                      # noqa: E402
from oracle import oracle as O          # noqa: E402
from tests import util                  # noqa: E402

emu = util.StageApi(C.CDLL(os.environ["EMU_LIB"]), "emu_")
cases = [(96, 96, 24, 8, 8), (72, 104, 26, 10, 8)] + ([] if os.environ.get("EMU_WAVES") else [(384, 320, 30, 8, 15)])
for (w, h, qp, bd, rng) in cases:
    prm_i = O.default_params(max(0, qp - 3), bit_depth=bd, me_range=rng)
    prm_p = O.default_params(qp, bit_depth=bd, me_range=rng)
    prm_p.rdo_zero, prm_p.rdo_cg, prm_p.pre_search = 1, 5, 1
    prm_i.chroma_modes = prm_p.chroma_modes = 1
    prm_i.intra_nxn = 1
    srcs = [util.synth_frame(h, w, seed=3, shift=(2 * i, i), bit_depth=bd) for i in range(3)]
    want = util.run_pipeline(O, srcs, prm_i, prm_p, bd)
    ref = None
    for i, (src, (a, d, f, sp)) in enumerate(zip(srcs, want)):
        prm = prm_i if i == 0 else prm_p
        got = emu.intra(src, prm) if i == 0 else emu.inter(src, ref, prm)
        assert util.same_analysis(a, got), (w, h, i)
        assert emu.deblock(a.rec, a.cu, bd).same(d)
        gf, gsp = emu.sao(src, d, prm)
        assert np.array_equal(gsp, sp) and gf.same(f)
        lf, lsp = emu.loop_filter(src, a.rec, a.cu, prm)                 # round 3: the fused loop filter a session runs
        assert np.array_equal(lsp, sp) and lf.same(f)
        if h % 32 == 0 and h >= 96:                                       # ... and a band of it with seams on both sides
            bf, bsp = emu.loop_filter(src, a.rec, a.cu, prm, band=(32, 32, 3))
            assert np.array_equal(bf.y[32:64], f.y[32:64]) and np.array_equal(bf.u[16:32], f.u[16:32])
        ref = f
    # round 3: a B picture between the first and the third picture's reconstructions
    recs = [x[2] for x in want]
    prm_b = O.default_params(min(51, qp + 2), bit_depth=bd, me_range=rng)
    prm_b.rdo_zero, prm_b.chroma_modes = 1, 1
    wb = O.analyze_b(srcs[1], recs[0], recs[2], prm_b, None, None)
    gb = emu.b(srcs[1], recs[0], recs[2], prm_b)
    assert util.same_analysis(wb, gb), (w, h, "B")
    print("ok", w, h, bd, flush=True)
