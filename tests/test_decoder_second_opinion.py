"""CPU: the decoder's OWN normative arithmetic (oracle/hevc_dec_recon.c) against the oracle's (oracle/hevc_oracle.c) and against closed forms.

VERDICT r02: hevc_dec.c used to call the oracle's dequant / inverse transform / intra prediction / deblocking / SAO, so "the stream decodes to the
encoder's reconstruction" could not see an error in those five.  They are now written twice, clause by clause, sharing nothing; here both are run on
random inputs (they must agree bit for bit) and the second one is checked against values worked out BY HAND from the clauses of H.265 (diagonal
modes as pure copies, a negative-angle mode with its invAngle projection, strong / normal / chroma deblocking on a step, tc / beta at 10 bit).
Every stream test (tests/test_bitstream_cpu.py, test_sliced_cpu.py, test_golden_streams.py, the GPU session tests) now goes through the second
implementation, since the decoder uses nothing else.  Parity with libx265 stays unpinned (no third-party codec exists on this pool)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

L = O.lib()


def p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------------ both implementations, random inputs
def test_chroma_qp_table_both():
    want = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37] + [q - 6 for q in range(44, 58)]       # Table 8-10
    assert [L.orc_dec2_chroma_qp(q) for q in range(58)] == want
    assert [L.orc_chroma_qp(q) for q in range(52)] == want[:52]


@pytest.mark.parametrize("log2n,dst", [(2, 0), (2, 1), (3, 0), (4, 0), (5, 0)])
@pytest.mark.parametrize("bd", [8, 10])
def test_scaling_and_inverse_transform_both(log2n, dst, bd):
    rng = np.random.default_rng(100 * log2n + 10 * dst + bd)
    n = 1 << log2n
    for trial in range(60):
        qp = int(rng.integers(0, 52))
        lvl = np.zeros((n, n), np.int16)
        k = int(rng.integers(1, n * n + 1)) if trial % 3 else 1
        idx = rng.choice(n * n, size=k, replace=False)
        mag = 32767 if trial % 7 == 0 else (300 if trial % 2 else 6)       # incl. levels that drive the clips of 8.6.3 / 8.6.4.1
        lvl.flat[idx] = rng.integers(-mag, mag + 1, size=k)
        want = O.inv_transform(O.dequant(lvl, qp, bd), dst=bool(dst), bit_depth=bd)
        got = np.empty((n, n), np.int32)
        L.orc_dec2_residual(p(lvl), p(got), log2n, qp, bd, dst)
        # residual samples are not clipped by 8.6.4 (a conforming stream keeps them small); the oracle stores them in 16 bits, so the extreme
        # trials are compared modulo 2^16
        assert np.array_equal(want, got.astype(np.int16)), (qp, trial)


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_prediction_35_modes_both(bd):
    rng = np.random.default_rng(bd)
    peak = (1 << bd) - 1
    for log2n in (2, 3, 4, 5):
        n = 1 << log2n
        for c_idx in (0, 1):
            if c_idx and log2n == 5:
                continue
            for kind in range(4):
                if kind == 0:
                    ref = rng.integers(0, peak + 1, 4 * n + 1)
                elif kind == 1:      # smooth ramp + small noise: the bi-linear (strong) filter condition of 8.4.4.2.3 holds at 32x32
                    ref = np.clip(np.linspace(peak * 0.3, peak * 0.6, 4 * n + 1) + rng.integers(-1, 2, 4 * n + 1), 0, peak)
                elif kind == 2:
                    ref = np.full(4 * n + 1, peak)
                else:
                    ref = rng.integers(0, 2, 4 * n + 1) * peak
                ref = ref.astype(np.uint16)
                for strong in (0, 1):
                    for mode in range(35):
                        want = O.intra_pred(ref, log2n, mode, c_idx, bd, filtered=True, strong=bool(strong))
                        got = np.empty((n, n), np.uint16)
                        L.orc_dec2_intra_pred(p(ref), p(got), log2n, mode, c_idx, bd, strong)
                        assert np.array_equal(want, got), (log2n, c_idx, kind, strong, mode)


def random_cu_map(rng, w, h, inter_share=0.6):
    cu = np.zeros((h // 8, w // 8), O.CU_DTYPE)

    def fill(x, y, l2):
        n = 1 << l2
        if l2 > 3 and (x + n > w or y + n > h or rng.random() < 0.5):
            for k in range(4):
                xx, yy = x + (k & 1) * n // 2, y + (k >> 1) * n // 2
                if xx < w and yy < h:
                    fill(xx, yy, l2 - 1)
            return
        r = np.zeros((), O.CU_DTYPE)
        r["log2_size"] = l2
        inter = rng.random() < inter_share
        r["flags"] = (1 if inter else 0) | (2 if rng.random() < 0.5 else 0) | (4 if rng.random() < 0.3 else 0) | (8 if rng.random() < 0.3 else 0)
        if not inter and l2 == 3 and rng.random() < 0.3:
            r["flags"] |= 16
            r["cbf_y4"] = int(rng.integers(0, 16))
            r["flags"] = (int(r["flags"]) & ~2) | (2 if r["cbf_y4"] else 0)
        r["qp"] = int(rng.integers(20, 45))
        r["mvx"], r["mvy"] = (int(rng.integers(-6, 7)), int(rng.integers(-6, 7))) if inter else (0, 0)
        cu[y // 8:(y + n) // 8, x // 8:(x + n) // 8] = r
    for y in range(0, h, 32):
        for x in range(0, w, 32):
            fill(x, y, 5)
    return cu


@pytest.mark.parametrize("w,h,bd", [(96, 64, 8), (72, 104, 10), (160, 96, 8)])
def test_deblocking_both(w, h, bd):
    rng = np.random.default_rng(w + h + bd)
    peak = (1 << bd) - 1
    for trial in range(6):
        cu = random_cu_map(rng, w, h)
        # blocky content: every 8x8 block flat + a little noise, so that all of dE = 0 / 1 / 2 and the dEp / dEq branches occur
        base = rng.integers(peak // 4, 3 * peak // 4, (h // 8, w // 8))
        amp = (2, 6, 20)[trial % 3] << (bd - 8)
        y = np.kron(base, np.ones((8, 8), np.int64)) + rng.integers(-amp, amp + 1, (h, w))
        u = np.kron(base[:, :], np.ones((4, 4), np.int64)) + rng.integers(-amp, amp + 1, (h // 2, w // 2))
        v = np.kron(base[::-1, :], np.ones((4, 4), np.int64)) + rng.integers(-amp, amp + 1, (h // 2, w // 2))
        f = O.Frame(np.clip(y, 0, peak), np.clip(u, 0, peak), np.clip(v, 0, peak))
        want = O.deblock(f, cu, bd)
        got = f.copy()
        L.orc_dec2_deblock(p(got.y), p(got.u), p(got.v), w, w // 2, w, h, p(np.ascontiguousarray(cu)), bd)
        assert not want.same(f)
        assert want.same(got), trial


@pytest.mark.parametrize("w,h,bd", [(96, 64, 8), (72, 104, 10)])
def test_sao_apply_both(w, h, bd):
    rng = np.random.default_rng(w * h + bd)
    peak = (1 << bd) - 1
    n_ctu = ((w + 31) // 32) * ((h + 31) // 32)
    for trial in range(6):
        f = O.Frame(rng.integers(0, peak + 1, (h, w)), rng.integers(0, peak + 1, (h // 2, w // 2)), rng.integers(0, peak + 1, (h // 2, w // 2)))
        if trial % 2:      # smooth content: equal neighbours (sign 0) occur
            f = O.Frame(f.y >> 3 << 3, f.u >> 4 << 4, f.v >> 4 << 4)
        sp = np.zeros(n_ctu, O.SAO_DTYPE)
        omax = 7 if bd == 8 else 31
        for i in range(n_ctu):
            sp[i]["type"] = rng.integers(0, 3, 2)
            sp[i]["eo_class"] = rng.integers(0, 4, 2)
            sp[i]["band_pos"] = rng.integers(0, 32, 3)
            for c in range(3):
                t = sp[i]["type"][1 if c else 0]
                o = rng.integers(0, omax + 1, 4)
                sp[i]["offset"][c] = o * rng.choice([-1, 1], 4) if t == 1 else [o[0], o[1], -o[2], -o[3]]
        want = O.sao_apply(f, sp, bd)
        got = O.Frame(np.zeros_like(f.y), np.zeros_like(f.u), np.zeros_like(f.v))
        L.orc_dec2_sao(p(f.y), p(f.u), p(f.v), w, w // 2, p(got.y), p(got.u), p(got.v), w, h, bd, p(sp))
        assert want.same(got), trial


# ------------------------------------------------------------------------------------------------ closed forms, worked out by hand from the clauses
def d2_pred(ref, log2n, mode, c_idx=1, bd=8, strong=0):
    n = 1 << log2n
    out = np.empty((n, n), np.uint16)
    L.orc_dec2_intra_pred(p(np.ascontiguousarray(ref, np.uint16)), p(out), log2n, mode, c_idx, bd, strong)
    return out


@pytest.mark.parametrize("log2n", [2, 3, 4])
def test_diagonal_modes_are_pure_copies(log2n):
    """intraPredAngle = +-32: iFact = 0 for every row, so 8.4.4.2.6 degenerates into a copy along the 45-degree diagonal (chroma: no smoothing filter)"""
    n = 1 << log2n
    ref = (np.arange(4 * n + 1) * 3 + 7).astype(np.uint16)
    left = lambda y: int(ref[2 * n - 1 - y])        # noqa: E731  p[-1][y], y = -1 is the corner
    top = lambda x: int(ref[2 * n + 1 + x])         # noqa: E731  p[x][-1]
    m34, m2, m18 = d2_pred(ref, log2n, 34), d2_pred(ref, log2n, 2), d2_pred(ref, log2n, 18)
    for y in range(n):
        for x in range(n):
            assert m34[y, x] == top(x + y + 1)                                   # up-right
            assert m2[y, x] == left(x + y + 1)                                   # down-left
            assert m18[y, x] == (top(x - y - 1) if x > y else left(y - x - 1) if y > x else int(ref[2 * n]))      # up-left, through the corner


def test_negative_angle_with_inverse_angle_projection_by_hand():
    """mode 13 (intraPredAngle -9, invAngle -910), 8x8, horizontal class: the main reference is the left column, and since (8 * -9) >> 5 = -3 < -1 the
    entries ref[-1], ref[-2], ref[-3] come from the top row at -1 + ((x * -910 + 128) >> 8) = 3, 6, 10 (worked out by hand); per column j the
    offset and weight are ((j + 1) * -9) >> 5 and & 31."""
    n = 8
    rng = np.random.default_rng(13)
    ref = rng.integers(0, 256, 4 * n + 1).astype(np.uint16)
    left = lambda y: int(ref[2 * n - 1 - y])        # noqa: E731
    top = lambda x: int(ref[2 * n + 1 + x])         # noqa: E731
    r = {x: left(x - 1) for x in range(0, n + 1)}   # ref[x] = p[-1][-1 + x]
    r[-1], r[-2], r[-3] = top(3), top(6), top(10)
    idx = [-1, -1, -1, -2, -2, -2, -2, -3]
    fact = [23, 14, 5, 28, 19, 10, 1, 24]
    got = d2_pred(ref, 3, 13)
    for j in range(n):          # j = x
        for i in range(n):      # i = y
            want = ((32 - fact[j]) * r[i + idx[j] + 1] + fact[j] * r[i + idx[j] + 2] + 16) >> 5
            assert got[i, j] == want, (i, j)
    # mode 11 (angle -2): (8 * -2) >> 5 = -1, no projection; every column interpolates between p[-1][y - 1] and p[-1][y]
    got = d2_pred(ref, 3, 11)
    for j in range(n):
        for i in range(n):
            assert got[i, j] == (2 * (j + 1) * left(i - 1) + (32 - 2 * (j + 1)) * left(i) + 16) >> 5


def test_planar_dc_and_edge_filters_by_hand():
    n = 4
    ref = np.array([10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120, 130, 140, 150, 160, 170], np.uint16)      # p[-1][7..0], corner, p[0..7][-1]
    left = lambda y: int(ref[2 * n - 1 - y])        # noqa: E731
    top = lambda x: int(ref[2 * n + 1 + x])         # noqa: E731
    got = d2_pred(ref, 2, 0, c_idx=0)
    for y in range(n):
        for x in range(n):
            assert got[y, x] == ((3 - x) * left(y) + (x + 1) * top(4) + (3 - y) * top(x) + (y + 1) * left(4) + 4) >> 3
    dc = (sum(top(i) + left(i) for i in range(4)) + 4) >> 3
    got = d2_pred(ref, 2, 1, c_idx=0)
    assert got[0, 0] == (left(0) + 2 * dc + top(0) + 2) >> 2 and got[0, 2] == (top(2) + 3 * dc + 2) >> 2 and got[3, 0] == (left(3) + 3 * dc + 2) >> 2 and got[2, 2] == dc
    assert (d2_pred(ref, 2, 1, c_idx=1) == dc).all()                                # chroma: no edge filter
    got = d2_pred(ref, 2, 26, c_idx=0)                                              # vertical + boundary smoothing of column 0
    assert all(got[y, x] == top(x) for y in range(4) for x in range(1, 4))
    assert [int(got[y, 0]) for y in range(4)] == [min(255, max(0, top(0) + ((left(y) - int(ref[8])) >> 1))) for y in range(4)]


def intra_cus(w, h, log2, qp):
    cu = np.zeros((h // 8, w // 8), O.CU_DTYPE)
    cu["log2_size"], cu["flags"], cu["qp"] = log2, 2, qp
    return cu


def step_picture(w, h, x0, a, b):
    y = np.full((h, w), a, np.uint16)
    y[:, x0:] = b
    c = np.full((h // 2, w // 2), a, np.uint16)
    c[:, x0 // 2:] = b
    return O.Frame(y, c.copy(), c.copy())


def dbk2(f, cu, bd):
    g = f.copy()
    h, w = g.shape
    L.orc_dec2_deblock(p(g.y), p(g.u), p(g.v), w, w // 2, w, h, p(np.ascontiguousarray(cu)), bd)
    return g


def test_deblocking_strong_normal_and_chroma_by_hand():
    """Bs = 2 (intra), QP 37: beta' = 36, tc' = tc'[37 + 2] = 5.  A step of 10 between flat halves passes every test of 8.7.2.5.6 (|p0 - q0| = 10 <
    (5 tc + 1) >> 1 = 13) -> strong filter; a step of 20 does not -> normal filter with delta = (9 * 20 - 3 * 20 + 8) >> 4 = 8 clipped to tc = 5,
    and both sides flat (dp = dq = 0 < (beta + (beta >> 1)) >> 3 = 6) -> p1 / q1 move by clip(+-tc >> 1)."""
    g = dbk2(step_picture(16, 16, 8, 100, 110), intra_cus(16, 16, 3, 37), 8)
    assert [int(v) for v in g.y[5, 4:12]] == [100, 101, 103, 104, 106, 108, 109, 110]
    assert (g.y[:, 4:12] == g.y[5, 4:12]).all() and (g.u == step_picture(16, 16, 8, 100, 110).u).all()      # chroma edges lie on the 8-sample CHROMA grid only
    g = dbk2(step_picture(16, 16, 8, 100, 120), intra_cus(16, 16, 3, 37), 8)
    assert [int(v) for v in g.y[9, 4:12]] == [100, 100, 102, 105, 115, 118, 120, 120]
    # chroma: 16x16 coding blocks put the edge x = 16 on the chroma grid; QpC(37) = 34, tc' = tc'[34 + 2] = 4, delta = ((10 << 2) + 100 - 110 + 4) >> 3 = 4
    g = dbk2(step_picture(32, 16, 16, 100, 110), intra_cus(32, 16, 4, 37), 8)
    assert [int(v) for v in g.u[3, 6:10]] == [100, 104, 106, 110] and (g.v == g.u).all()
    # inter blocks without residual and with equal motion: no edge is filtered; a vector difference of 4 quarter samples makes Bs = 1
    cu = intra_cus(16, 16, 3, 37)
    cu["flags"] = 1
    f = step_picture(16, 16, 8, 100, 110)
    assert dbk2(f, cu, 8).same(f)
    cu[:, 1]["mvx"] = 4
    g = dbk2(f, cu, 8)
    assert not g.same(f) and (g.u == f.u).all()          # Bs = 1: luma only; tc' = tc'[37] = 4
    cu[:, 1]["mvx"] = 3
    assert dbk2(f, cu, 8).same(f)


def test_deblocking_tc_and_beta_scale_with_bit_depth():
    """10 bit: beta = beta' * 4 = 144, tc = tc' * 4 = 20 (8.7.2.5.3).  A step of 40: |p0 - q0| = 40 < (5 * 20 + 1) >> 1 = 50 -> strong;
    p0' = (400 + 800 + 800 + 880 + 440 + 4) >> 3 = 415.  A step of 52 is not strong: delta = (9 * 52 - 3 * 52 + 8) >> 4 = 20 = tc."""
    g = dbk2(step_picture(16, 16, 8, 400, 440), intra_cus(16, 16, 3, 37), 10)
    assert [int(v) for v in g.y[2, 4:12]] == [400, 405, 410, 415, 425, 430, 435, 440]
    g = dbk2(step_picture(16, 16, 8, 400, 452), intra_cus(16, 16, 3, 37), 10)
    assert [int(v) for v in g.y[2, 6:10]] == [410, 420, 432, 442]
    # below the table's first non-zero tc (Q = qp + 2 < 18) nothing moves
    f = step_picture(16, 16, 8, 400, 440)
    assert dbk2(f, intra_cus(16, 16, 3, 15), 10).same(f)


def test_sao_band_and_edge_by_hand():
    w = h = 32
    y = np.full((h, w), 64, np.uint16)
    y[10, 10] = 60                                   # a local minimum: edgeIdx 1 in every class
    y[20, 20] = 70                                   # a local maximum: edgeIdx 4
    f = O.Frame(y, np.full((16, 16), 100, np.uint16), np.full((16, 16), 200, np.uint16))
    sp = np.zeros(1, O.SAO_DTYPE)
    sp[0]["type"] = (2, 1)
    sp[0]["eo_class"] = (0, 0)
    sp[0]["offset"][0] = (3, 1, -1, -2)
    sp[0]["band_pos"] = (0, 12, 24)                  # 100 >> 3 = 12 -> band k = 0; 200 >> 3 = 25 -> k = 1
    sp[0]["offset"][1] = (5, 0, 0, 0)
    sp[0]["offset"][2] = (0, -6, 0, 0)
    got = O.Frame(np.zeros_like(f.y), np.zeros_like(f.u), np.zeros_like(f.v))
    L.orc_dec2_sao(p(f.y), p(f.u), p(f.v), w, w // 2, p(got.y), p(got.u), p(got.v), w, h, 8, p(sp))
    want = y.copy()
    want[10, 10] = 63                                # minimum + offset[0]
    want[10, 9] = want[10, 11] = 64 - 1              # its horizontal neighbours see one smaller neighbour: edgeIdx 3 (concave corner) -> offset[2] = -1
    want[20, 20] = 68                                # maximum + offset[3]
    want[20, 19] = want[20, 21] = 64 + 1             # one larger neighbour: edgeIdx 2 -> offset[1] = +1
    assert np.array_equal(got.y, want)
    assert (got.u == 105).all() and (got.v == 194).all()
