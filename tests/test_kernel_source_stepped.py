"""CPU: the HIP kernel SOURCES (hevc_amd/csrc/kernels/*.h) stepped with the sequential executor must reproduce the
oracle bit for bit.  This checks the kernels' logic without a GPU; the -m gpu tests check the gfx950 binaries.
tests/emu/libkernel_emu.so is a test harness only — hevc_amd/ never loads it (tests/test_layout.py)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from tests import util

EMU_DIR = Path(__file__).resolve().parent / "emu"


@pytest.fixture(scope="module")
def emu():
    so = EMU_DIR / "libkernel_emu.so"
    srcs = [EMU_DIR / "emu.cpp"] + list((EMU_DIR.parents[1] / "hevc_amd" / "csrc" / "kernels").glob("*.h"))
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", "-pthread", "-o", str(so), str(EMU_DIR / "emu.cpp")], check=True)
    return util.StageApi(C.CDLL(str(so)), "emu_")


CASES = [
    # w, h, qp, bit depth, me range
    (64, 64, 30, 8, 8),
    (96, 80, 22, 8, 8),        # partial CTU row
    (136, 72, 35, 8, 16),      # partial CTU row and column
    (72, 104, 26, 10, 8),      # Main10
    (160, 96, 14, 8, 12),      # low QP: many coefficients, small CUs
]


@pytest.mark.parametrize("w,h,qp,bd,rng", CASES)
def test_stepped_kernels_equal_oracle(emu, w, h, qp, bd, rng):
    prm_i = O.default_params(max(0, qp - 3), bit_depth=bd, me_range=rng)
    prm_p = O.default_params(qp, bit_depth=bd, me_range=rng)
    prm_p.rdo_zero = int(qp >= 26)               # RD zero-out of inter TUs on for the higher QPs (where it bites), off for the rest
    prm_p.rdo_cg = 5 if 22 <= qp <= 30 else 0    # RD zero-out of 4x4 coefficient groups: the session's default strength / off
    prm_i.chroma_modes = int(qp < 35)            # chroma intra mode decision on for most cases (4x4, 8x8 and 16x16 chroma blocks)
    srcs = [util.synth_frame(h, w, seed=3, shift=(2 * i, i), bit_depth=bd) for i in range(3)]
    want = util.run_pipeline(O, srcs, prm_i, prm_p, bd)
    ref = None
    for i, (src, (a, d, f, sp)) in enumerate(zip(srcs, want)):
        prm = prm_i if i == 0 else prm_p
        got = emu.intra(src, prm) if i == 0 else emu.inter(src, ref, prm)
        assert util.same_analysis(a, got), f"picture {i}: " + util.describe_diff(a, got)
        if i:
            assert np.array_equal(a.me, got.me)
        gd = emu.deblock(a.rec, a.cu, bd)
        assert gd.same(d), f"deblock picture {i}"
        gf, gsp = emu.sao(src, d, prm)
        assert np.array_equal(gsp, sp) and gf.same(f), f"sao picture {i}"
        ref = f
    # the content must exercise more than one CU size somewhere in the run
    sizes = np.unique(np.concatenate([x[0].cu["log2_size"].ravel() for x in want]))
    assert len(sizes) >= 2 or qp >= 30


@pytest.mark.parametrize("w,h,grid,qp,bd", [(544, 160, (2, 2), 26, 8), (800, 224, (3, 3), 33, 8), (512, 192, (2, 3), 22, 10)])
def test_stepped_intra_with_tile_grid(emu, w, h, grid, qp, bd):
    """IDR pictures are analysed per tile: neighbours across a tile border are unavailable (prediction and MPM)."""
    prm = O.default_params(qp, bit_depth=bd)
    prm.tile_cols, prm.tile_rows = grid
    src = util.synth_frame(h, w, seed=9, bit_depth=bd)
    want, got = O.analyze_intra(src, prm), emu.intra(src, prm)
    assert util.same_analysis(want, got), util.describe_diff(want, got)
    flat = O.default_params(qp, bit_depth=bd)
    assert not util.same_analysis(O.analyze_intra(src, flat), want)       # the grid changes the prediction at the borders


@pytest.mark.parametrize("w,h,grid,qp,bd", [(64, 64, (1, 1), 22, 8), (136, 72, (1, 1), 30, 8), (72, 104, (1, 1), 18, 10), (544, 160, (2, 2), 26, 8)])
def test_stepped_intra_nxn(emu, w, h, grid, qp, bd):
    """8x8 intra CUs tried as four 4x4 PUs (NxN, DST-VII): same decisions, levels, records and reconstruction as the oracle."""
    prm = O.default_params(qp, bit_depth=bd)
    prm.tile_cols, prm.tile_rows = grid
    prm.intra_nxn = 1
    prm.chroma_modes = int(bd == 8)
    src = util.synth_frame(h, w, seed=31, bit_depth=bd)
    want, got = O.analyze_intra(src, prm), emu.intra(src, prm)
    assert (want.cu["flags"] & 16).any(), "content must make NxN win somewhere"
    assert util.same_analysis(want, got), util.describe_diff(want, got)


@pytest.mark.parametrize("w,h,qp,bd,nxn", [(160, 128, 28, 8, 0), (136, 104, 24, 8, 1), (128, 96, 30, 10, 0)])
def test_stepped_intra_second_pass_of_p_pictures(emu, w, h, qp, bd, nxn):
    """P pictures: CTUs the reference cannot predict are re-coded as intra in two independent-set rounds, exactly as the oracle."""
    from tests.test_bitstream_cpu import occluded_clip
    prm = O.default_params(qp, bit_depth=bd, me_range=8)
    prm.intra_in_p, prm.intra_nxn, prm.chroma_modes = 1, nxn, 1
    srcs = occluded_clip(w, h, bd)
    ref = O.sao(srcs[0], O.deblock(*(lambda a: (a.rec, a.cu))(O.analyze_intra(srcs[0], prm)), bd), prm)[0]
    for i in (1, 2):
        want = O.analyze_inter(srcs[i], ref, prm, dump_me=True)
        got = emu.inter(srcs[i], ref, prm)
        assert ((want.cu["flags"] & 1) == 0).any(), "the occluded patch must go intra"
        assert util.same_analysis(want, got), f"picture {i}: " + util.describe_diff(want, got)
        ref = O.sao(srcs[i], O.deblock(want.rec, want.cu, bd), prm)[0]


@pytest.mark.parametrize("w,h,bd,shift", [(192, 128, 8, (38, -22)), (136, 104, 10, (-50, 17)), (160, 96, 8, (3, 1)), (384, 320, 8, (22, -13))])   # the last: CTUs whose window is fetched as dwords
def test_stepped_pre_search_finds_fast_motion(emu, w, h, bd, shift):
    """Search centres from the 1/4-size pictures: +-8 integer search around them follows a global shift far outside +-8."""
    prm = O.default_params(27, bit_depth=bd, me_range=8)
    prm.pre_search = 1
    base = util.synth_frame(h + 128, w + 128, 5, bit_depth=bd)          # a pure translation: two crops of one larger picture
    def crop(ox, oy):
        return O.Frame(base.y[64 + oy:64 + oy + h, 64 + ox:64 + ox + w].copy(), base.u[32 + oy // 2:32 + (oy + h) // 2, 32 + ox // 2:32 + (ox + w) // 2].copy(),
                       base.v[32 + oy // 2:32 + (oy + h) // 2, 32 + ox // 2:32 + (ox + w) // 2].copy())
    a, b = crop(0, 0), crop(shift[0], shift[1])                        # b(x) = a(x + shift): every block of b sits at +shift in a
    want = O.analyze_inter(b, a, prm, dump_me=True)
    got = emu.inter(b, a, prm)
    assert util.same_analysis(want, got) and np.array_equal(want.me, got.me), util.describe_diff(want, got)
    vals, counts = np.unique(want.cu["mvx"].astype(np.int32) * 4096 + want.cu["mvy"], return_counts=True)
    assert vals[np.argmax(counts)] == 4 * shift[0] * 4096 + 4 * shift[1] and counts.max() > 0.3 * want.cu.size     # the true shift dominates (part of the picture has no counterpart in the reference)
    if max(abs(shift[0]), abs(shift[1])) > 8:
        prm.pre_search = 0
        assert O.analyze_inter(b, a, prm).est > 1.5 * want.est        # without the centres the motion is out of reach


def test_search_centres_are_honoured(emu):
    w, h, bd = 96, 64, 8
    prm = O.default_params(26, me_range=8)
    a, b = util.synth_frame(h, w, 5), util.synth_frame(h, w, 5, shift=(11, -6))
    cen = np.zeros((util.n_ctus(w, h), 2), np.int16)
    cen[:, 0], cen[:, 1] = 10, -5          # true motion (11,-6) is only reachable around this centre with range 8
    want = O.analyze_inter(b, a, prm, centers=cen, dump_me=True)
    got = emu.inter(b, a, prm, centers=cen)
    assert util.same_analysis(want, got) and np.array_equal(want.me, got.me)
    assert np.median(want.cu["mvx"]) == 44 and np.median(want.cu["mvy"]) == -24


@pytest.mark.parametrize("order,fill", [("1", None), ("2", "11")])
def test_phase_programs_do_not_depend_on_thread_order_or_initial_lds(emu, monkeypatch, order, fill):
    """Inside a phase the threads run in any order on the device, and LDS starts with whatever the previous workgroup left: stepping the
    kernels with reversed / shuffled thread order and pseudo-random initial shared state must still reproduce the oracle."""
    monkeypatch.setenv("EMU_ORDER", order)
    if fill:
        monkeypatch.setenv("EMU_SHARED_FILL", fill)
    prm = O.default_params(24, bit_depth=8, me_range=8)
    prm.intra_nxn, prm.chroma_modes, prm.rdo_zero, prm.pre_search, prm.rdo_cg = 1, 1, 1, 1, 5
    srcs = [util.synth_frame(104, 136, seed=31, shift=(3 * i, i), bit_depth=8) for i in range(2)]
    want, got = O.analyze_intra(srcs[0], prm), emu.intra(srcs[0], prm)
    assert util.same_analysis(want, got), util.describe_diff(want, got)
    ref = O.sao(srcs[0], O.deblock(want.rec, want.cu, 8), prm)[0]
    want, got = O.analyze_inter(srcs[1], ref, prm, dump_me=True), emu.inter(srcs[1], ref, prm)
    assert util.same_analysis(want, got), util.describe_diff(want, got)


def test_waves_as_threads_with_real_barriers(emu, monkeypatch):
    """EMU_WAVES: the four waves of every workgroup run as four host threads, phases end in a real barrier and random waves are delayed
    after it.  Code between two phases (uniform branches on shared state) then runs once per wave, as on the device: a wave that reads a
    word another wave has already rewritten takes the other branch, misses a barrier and the call fails (this is how the NxN race of
    round 1 shows on a CPU).  NxN, chroma modes, RD zero-out, pre-search and the intra second pass are all on."""
    from tests.test_bitstream_cpu import occluded_clip
    monkeypatch.setenv("EMU_WAVES", "5")
    prm = O.default_params(24, bit_depth=8, me_range=8)
    prm.intra_nxn, prm.chroma_modes, prm.rdo_zero, prm.pre_search, prm.intra_in_p, prm.rdo_cg = 1, 1, 1, 1, 1, 5
    srcs = occluded_clip(136, 104, 8)
    want, got = O.analyze_intra(srcs[0], prm), emu.intra(srcs[0], prm)
    assert (want.cu["flags"] & 16).any()
    assert util.same_analysis(want, got), util.describe_diff(want, got)
    dbk = O.deblock(want.rec, want.cu, 8)
    (ref, sp), (gref, gsp) = O.sao(srcs[0], dbk, prm), emu.sao(srcs[0], dbk, prm)
    assert ref.same(gref) and np.array_equal(sp, gsp)
    want, got = O.analyze_inter(srcs[1], ref, prm, dump_me=True), emu.inter(srcs[1], ref, prm)
    assert ((want.cu["flags"] & 1) == 0).any()
    assert util.same_analysis(want, got), util.describe_diff(want, got)


@pytest.mark.parametrize("w,h,qp,bd,rng,pre", [(96, 80, 26, 8, 8, 0), (136, 72, 32, 8, 12, 1), (72, 104, 24, 10, 8, 1), (160, 96, 20, 8, 12, 0)])
def test_stepped_b_picture_kernels_equal_oracle(emu, w, h, qp, bd, rng, pre):
    """cfg.bframes: the B form of the CTU program (list-0 tree + refinement, list-1 refinement from its own integer search, bi-prediction trial by the
    default weighted average of the 14-bit predictions, cheapest of three) against orc_analyze_b_frame: both integer-search dumps, records (incl. which
    lists and the list-1 vector), levels, reconstruction, estimate; then deblocking with the two-list boundary strength and SAO."""
    prm_i, prm_p, prm_b = O.default_params(max(0, qp - 3), bd, rng), O.default_params(qp, bd, rng), O.default_params(qp + 2, bd, rng)
    prm_p.rdo_zero = prm_b.rdo_zero = 1
    f = [util.synth_frame(h, w, seed=23, shift=(3 * i, 2 * i), bit_depth=bd) for i in range(3)]
    a0 = O.analyze_intra(f[0], prm_i)
    r0, _ = O.sao(f[0], O.deblock(a0.rec, a0.cu, bd), prm_i)
    a2 = O.analyze_inter(f[2], r0, prm_p)
    r2, _ = O.sao(f[2], O.deblock(a2.rec, a2.cu, bd), prm_p)
    c0 = O.search_centres(f[1], f[0], bd) if pre else None
    c1 = O.search_centres(f[1], f[2], bd) if pre else None
    want = O.analyze_b(f[1], r0, r2, prm_b, c0, c1, dump_me=True)
    got = emu.b(f[1], r0, r2, prm_b, c0, c1)
    assert np.array_equal(want.me[0], got.me[0]) and np.array_equal(want.me[1], got.me[1]), "integer searches differ"
    assert util.same_analysis(want, got), util.describe_diff(want, got)
    kinds = {int(x) & 96 for x in np.unique(want.cu["flags"])}
    assert len(kinds) >= 2, kinds
    d = O.deblock(want.rec, want.cu, bd)
    assert emu.deblock(want.rec, want.cu, bd).same(d)
    gf, gsp = emu.sao(f[1], d, prm_b)
    wf, wsp = O.sao(f[1], d, prm_b)
    assert np.array_equal(gsp, wsp) and gf.same(wf)


@pytest.mark.parametrize("w,h,bd", [(136, 104, 8), (96, 72, 10)])
def test_sao_programs_leave_the_squared_error_of_their_ctu(emu, w, h, bd):
    """SaoArgs::sse_ctu: every CTU program writes the squared error (source vs final reconstruction) of its own samples, per plane; k_sse_fold adds the table up
    into the picture's statistics.  Against numpy, CTU by CTU, partial CTUs included."""
    prm = O.default_params(30, bit_depth=bd, me_range=8)
    src = util.synth_frame(h, w, seed=21, bit_depth=bd)
    a = O.analyze_intra(src, prm)
    dbk = O.deblock(a.rec, a.cu, bd)
    ref, sp = O.sao(src, dbk, prm)
    got, gsp, sse = emu.sao_sse(src, dbk, prm)
    assert ref.same(got) and np.array_equal(sp, gsp)
    cw = (w + 31) // 32
    for c in range(sse.shape[0]):
        x0, y0 = (c % cw) * 32, (c // cw) * 32
        for pl, (s_, r_, sh) in enumerate(((src.y, ref.y, 0), (src.u, ref.u, 1), (src.v, ref.v, 1))):
            d = s_[y0 >> sh:(y0 + 32) >> sh, x0 >> sh:(x0 + 32) >> sh].astype(np.int64) - r_[y0 >> sh:(y0 + 32) >> sh, x0 >> sh:(x0 + 32) >> sh].astype(np.int64)
            assert int(sse[c, pl]) == int((d * d).sum()), (c, pl)
    assert sse.sum() > 0


@pytest.mark.parametrize("w,h,bd,qp", [(136, 104, 8, 30), (96, 72, 10, 26), (200, 136, 8, 38)])
def test_fused_loop_filter_equals_deblocking_then_sao(emu, w, h, bd, qp):
    """SaoArgs::cu: the CTU programs deblock their own tile of the PRE-deblock reconstruction (halo of 4 luma / 2 chroma samples, both edge passes) and then decide and
    apply SAO.  The result must be the two picture passes' (oracle deblock, then oracle SAO) bit for bit: intra pictures (Bs 2, chroma edges), P pictures with
    skipped / coded CUs of all sizes (Bs 0 / 1 by motion and residual), partial CTUs on the right and at the bottom."""
    from tests.test_bitstream_cpu import occluded_clip
    prm = O.default_params(qp, bit_depth=bd, me_range=8)
    srcs = occluded_clip(w, h, bd) if bd == 8 else [util.synth_frame(h, w, seed=5, shift=(3 * i, 2 * i), bit_depth=bd) for i in range(3)]
    a = O.analyze_intra(srcs[0], prm)
    want, wsp = O.sao(srcs[0], O.deblock(a.rec, a.cu, bd), prm)
    got, gsp = emu.loop_filter(srcs[0], a.rec, a.cu, prm)
    assert np.array_equal(wsp, gsp) and want.same(got), "intra picture"
    assert not O.deblock(a.rec, a.cu, bd).same(a.rec)
    ref = want
    for i in (1, 2):
        a = O.analyze_inter(srcs[i], ref, prm)
        dbk = O.deblock(a.rec, a.cu, bd)
        want, wsp = O.sao(srcs[i], dbk, prm)
        got, gsp = emu.loop_filter(srcs[i], a.rec, a.cu, prm)
        assert np.array_equal(wsp, gsp) and want.same(got), f"P picture {i}"
        ref = want


def test_fused_loop_filter_in_wave_threads_and_any_lane_order(emu, monkeypatch):
    prm = O.default_params(32, bit_depth=8, me_range=8)
    src = util.synth_frame(72, 104, seed=8, bit_depth=8)
    a = O.analyze_intra(src, prm)
    want, wsp = O.sao(src, O.deblock(a.rec, a.cu, 8), prm)
    for order in ("1", "2"):
        monkeypatch.setenv("EMU_ORDER", order)
        got, gsp = emu.loop_filter(src, a.rec, a.cu, prm)
        assert np.array_equal(wsp, gsp) and want.same(got), order
    monkeypatch.delenv("EMU_ORDER")
    monkeypatch.setenv("EMU_WAVES", "3")
    got, gsp = emu.loop_filter(src, a.rec, a.cu, prm)
    assert np.array_equal(wsp, gsp) and want.same(got)


def test_fused_loop_filter_of_a_band_filters_across_its_seams(emu):
    """one slice (a band of CTU rows) of a picture whose other slices are coded elsewhere, filters across the seams (cfg.slice_halo): the band's programs find the
    neighbours' pre-deblock rows and CU records beyond their first and last row (where the exchange of csrc/slice_group.h puts them) and produce exactly the band's
    rows of the whole picture's loop filter"""
    w, h, bd = 136, 160, 8
    prm = O.default_params(34, bit_depth=bd, me_range=8)
    src = util.synth_frame(h, w, seed=12, bit_depth=bd)
    a = O.analyze_intra(src, prm)
    want, wsp = O.sao(src, O.deblock(a.rec, a.cu, bd), prm)
    cw = (w + 31) // 32
    for y0, bh, halo in ((0, 64, 2), (64, 32, 3), (96, 64, 1)):
        got, gsp = emu.loop_filter(src, a.rec, a.cu, prm, band=(y0, bh, halo))
        assert np.array_equal(got.y[y0:y0 + bh], want.y[y0:y0 + bh]) and np.array_equal(got.u[y0 // 2:(y0 + bh) // 2], want.u[y0 // 2:(y0 + bh) // 2])
        assert np.array_equal(got.v[y0 // 2:(y0 + bh) // 2], want.v[y0 // 2:(y0 + bh) // 2])
        assert np.array_equal(gsp, wsp[(y0 // 32) * cw:((y0 + bh + 31) // 32) * cw])


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("kind", ["flat_extremes", "checkerboard", "noise_extremes"])
def test_fractional_search_at_the_limits_of_the_16_bit_packed_transform(emu, kind, bd):
    """8 bit: the fractional search's Hadamard stages run on 16-bit pairs (v_pk_add / sub / max_i16).  Differences of +-255 in every sample drive the transform to its
    largest coefficients (255 x 32 before the last stage, 255 x 64 after it) and the packed sums to 65280 of 65535: black against white, a checkerboard of extremes,
    and random extremes must still equal the oracle's 32-bit arithmetic."""
    w, h = 96, 64
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    top, dt = (1 << bd) - 1, np.uint8 if bd == 8 else np.uint16        # (10 bit: +-1023, the transform's 16-bit lanes reach 1023 x 32 = 32736 of 32767)
    if kind == "flat_extremes":
        a, b = np.zeros((h, w), dt), np.full((h, w), top, dt)
    elif kind == "checkerboard":
        a = (((xx + yy) & 1) * top).astype(dt)
        b = (top - a).astype(dt)
    else:
        a = (rng.integers(0, 2, (h, w)) * top).astype(dt)
        b = (rng.integers(0, 2, (h, w)) * top).astype(dt)

    def frame(y):
        c = np.ascontiguousarray(y[::2, ::2])
        return O.Frame(y.copy(), c.copy(), (top - c).astype(dt))
    prm = O.default_params(30, bit_depth=bd, me_range=8)
    prm.rdo_zero = 1
    for src, ref in ((frame(a), frame(b)), (frame(b), frame(a))):
        want, got = O.analyze_inter(src, ref, prm, dump_me=True), emu.inter(src, ref, prm)
        assert np.array_equal(want.me, got.me)
        assert util.same_analysis(want, got), util.describe_diff(want, got)
