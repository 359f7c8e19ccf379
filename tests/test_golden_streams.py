"""Golden coded pictures (tests/golden/streams.json, made by tests/golden/make_stream_goldens.py from the oracle analysis + the host coder):
the CPU path must still produce them, and on a GPU box a real MI355X session must produce the SAME BYTES — kernels, session pipeline and host
coder together against a committed fixture, not only against an oracle run of the same commit."""
import importlib.util
import json
from pathlib import Path

import pytest

from oracle import oracle as O
from tests import util

GOLDEN = Path(__file__).parent / "golden"
_spec = importlib.util.spec_from_file_location("make_stream_goldens", GOLDEN / "make_stream_goldens.py")
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
WANT = json.loads((GOLDEN / "streams.json").read_text())


def test_fixture_covers_every_case():
    assert set(WANT) == set(G.CASES)


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_and_host_coder_still_produce_the_golden_pictures(name):
    got = G.compute(name)
    assert got["bytes"] == WANT[name]["bytes"] and got["pictures"] == WANT[name]["pictures"], "coded pictures changed: re-bless tests/golden/streams.json in the same commit if intended"
    assert got["recon"] == WANT[name]["recon"]
    # and the fixture is a decodable stream: parameter sets + pictures -> the oracle decoder gives back the hashed reconstructions
    import ctypes as C
    from hevc_amd import _lib
    cfg, buf = G.config(name), (C.c_uint8 * (1 << 16))()
    n = _lib.load().mihevc_write_parameter_sets(C.byref(cfg), buf, len(buf))
    dec, _ = O.decode(bytes(buf[:n]) + b"".join(p for p, _ in G.oracle_pictures(name)))
    assert [G.frame_hash(d) for d in dec] == WANT[name]["recon"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_mi355x_session_produces_the_golden_pictures(name):
    from hevc_amd import _lib
    from hevc_amd.encoder import Encoder
    assert _lib.load().mihevc_device_count() >= 1, "no gfx950 device visible"
    cfg = G.config(name)
    bd = cfg.bit_depth
    with Encoder(cfg, device=0, keep_recon=True) as enc:
        for f in G.frames(name):
            enc.send(*util.planes(f, bd))
        enc.flush()
        packets = [d for d, _pts, _key in enc.packets()]
        headers = enc.headers()
        recs = [O.Frame(*enc.recon(i)) for i in range(len(packets))]
    assert packets[0].startswith(headers)
    packets[0] = packets[0][len(headers):]           # the session puts the parameter sets in front of the first IDR
    assert [len(p) for p in packets] == WANT[name]["bytes"]
    assert [G.sha(p) for p in packets] == WANT[name]["pictures"]
    assert [G.frame_hash(r) for r in recs] == WANT[name]["recon"]
