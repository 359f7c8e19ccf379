"""CPU: the C-ABI library loads, exports every symbol include/mihevc.h declares, struct layouts agree with the
ctypes mirrors, and — with no GPU in this container — every device entry point fails loudly (no fallback)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from hevc_amd import _lib

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "mihevc.h").read_text()


def test_every_declared_symbol_is_exported():
    declared = set(re.findall(r"\b(mihevc_[a-z0-9_]+)\s*\(", HEADER))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mihevc_abi_version() == int(re.search(r"#define MIHEVC_ABI_VERSION (\d+)", HEADER).group(1))


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mihevc.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(mihevc_config),sizeof(mihevc_stats),sizeof(mihevc_cost_params),sizeof(mihevc_cu_rec),sizeof(mihevc_sao_ctu),"
                   "offsetof(mihevc_config,me_range),offsetof(mihevc_stats,stage_ms));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    from oracle import oracle as O
    assert sizes[:5] == [C.sizeof(_lib.Config), C.sizeof(_lib.Stats), C.sizeof(_lib.CostParams), O.CU_DTYPE.itemsize, O.SAO_DTYPE.itemsize]
    assert sizes[5] == _lib.Config.me_range.offset and sizes[6] == _lib.Stats.stage_ms.offset


def test_defaults_are_the_reference_operating_point():
    c = _lib.default_config()      # core/transcoder.py:398-411 for 1080p30 SDR
    assert (c.width, c.height, c.crf, c.vbv_maxrate_kbps, c.vbv_bufsize_kbits, c.keyint, c.min_keyint, c.level_idc, c.tier) == \
           (1920, 1080, 19, 2940, 3528, 90, 45, 120, 0)
    assert (c.md_primaries[0][0], c.md_primaries[2][1], c.md_max_lum, c.max_cll, c.max_fall) == (13250, 16000, 10000000, 1000, 400)


def test_cost_params_match_the_oracle_helper():
    from oracle import oracle as O
    for qp in range(0, 52):
        for bd in (8, 10):
            p, o = _lib.cost_params(qp, bd, 16), O.default_params(qp, bd, 16)
            assert (p.qp, p.qp_c, p.lambda_sad_q4, p.lambda_q4) == (o.qp, o.qp_c, o.lambda_sad_q4, o.lambda_q4), qp


@pytest.mark.skipif(_lib.load().mihevc_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    lib = _lib.load()
    assert lib.mihevc_device_count() == 0
    cfg = _lib.default_config()
    s = C.c_void_p()
    assert lib.mihevc_open(C.byref(cfg), 0, C.byref(s)) == _lib.ENODEV and not s.value
    z = np.zeros((64, 64), np.int16)
    assert lib.mihevc_k_transform(0, z.ctypes.data, z.ctypes.data, z.ctypes.data, 1, 3, 22, 8, 1, 0) == _lib.ENODEV
    from hevc_amd.encoder import Encoder
    with pytest.raises(_lib.MihevcError) as e:
        Encoder(cfg)
    assert e.value.code == _lib.ENODEV
    from hevc_amd import utils
    utils.mi355x_device_count.cache_clear()
    assert utils.has_mi355x() is False
    assert lib.mihevc_strerror(_lib.ENODEV).decode().startswith("no gfx950")


def test_bad_arguments_are_rejected():
    lib = _lib.load()
    cfg = _lib.default_config()
    buf = (C.c_uint8 * 16)()
    assert lib.mihevc_write_parameter_sets(C.byref(cfg), buf, 16) == _lib.ENOMEM      # too small, not a crash
    cfg.bit_depth = 9
    assert lib.mihevc_write_parameter_sets(C.byref(cfg), buf, 16) == _lib.EINVAL
    assert lib.mihevc_open(None, 0, None) == _lib.EINVAL
