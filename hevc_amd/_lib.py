"""ctypes binding of hevc_amd/libmihevc.so (C ABI: include/mihevc.h).

The library holds the gfx950 code objects, the host CABAC/bitstream writer and the session pipeline.  There is no
Python or CPU fallback for any of it: `load()` raises if the shared object is missing or an exported symbol the
header declares is absent, and every encode entry point returns MIHEVC_ENODEV on a host without an MI355X.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MIHEVC_LIBRARY", HERE / "libmihevc.so"))      # override: A/B runs of two builds in one GPU call

OK, EAGAIN, EOF, EINVAL, ENODEV, ENOMEM, EDEVICE, ESTATE = 0, -1, -2, -3, -4, -5, -6, -7


class Config(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in (
        "width", "height", "fps_num", "fps_den", "bit_depth", "level_idc", "tier", "crf", "qp", "vbv_maxrate_kbps",
        "vbv_bufsize_kbits", "keyint", "min_keyint", "colour_primaries", "transfer", "matrix", "full_range", "chroma_loc",
        "aud", "repeat_headers", "hdr10")] +
        [("md_primaries", (C.c_uint16 * 2) * 3), ("md_white", C.c_uint16 * 2), ("md_max_lum", C.c_uint32), ("md_min_lum", C.c_uint32),
         ("max_cll", C.c_uint16), ("max_fall", C.c_uint16)] +
        [(n, C.c_int32) for n in ("me_range", "gops_in_flight", "host_threads", "sao", "profile_stages", "intra_tiles", "intra_nxn", "intra_in_p", "hrd", "pre_search", "rdo_zero", "chroma_modes", "pic_height", "slice_count", "slice_index")] +
        [("slice_ctu_rows", C.c_int32 * 16), ("rate_share_q16", C.c_int32), ("scenecut", C.c_int32), ("gop_balance", C.c_int32), ("rdo_cg", C.c_int32), ("p_tiles", C.c_int32), ("bframes", C.c_int32), ("b_qp_offset", C.c_int32), ("slice_halo", C.c_int32), ("slice_group", C.c_int32)])


class Stats(C.Structure):
    _fields_ = [("frames_in", C.c_int64), ("frames_out", C.c_int64), ("bytes_out", C.c_int64), ("sse_y", C.c_double), ("sse_u", C.c_double),
                ("sse_v", C.c_double), ("device_ms", C.c_double), ("entropy_ms", C.c_double), ("last_qp", C.c_int32), ("reserved", C.c_int32 * 7),
                ("stage_ms", C.c_double * 8), ("stage_launches", C.c_int64 * 8), ("stage_pictures", C.c_int64 * 8)]


STAGE_NAMES = ("intra", "me_search", "inter_ctu", "deblock", "sao", "pad", "sse", "intra_p")


class CostParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("qp", "qp_c", "bit_depth", "lambda_sad_q4", "lambda_q4", "me_range", "tile_cols", "tile_rows", "intra_nxn", "intra_in_p", "pre_search", "rdo_zero", "chroma_modes", "mc_top", "mc_bottom", "rdo_cg")]


# every symbol include/mihevc.h declares; tests/test_abi.py checks the header against this list and the .so
EXPORTS = (
    "mihevc_abi_version", "mihevc_device_count", "mihevc_device_numa_node", "mihevc_config_default", "mihevc_open", "mihevc_send_frame", "mihevc_send_frame_async", "mihevc_sync_uploads", "mihevc_send_frame_device", "mihevc_send_frames_device",
    "mihevc_receive_packet", "mihevc_flush", "mihevc_abort", "mihevc_close", "mihevc_get_stats", "mihevc_get_headers", "mihevc_set_keep_recon",
    "mihevc_get_recon", "mihevc_coded_size", "mihevc_get_frame_info", "mihevc_strerror", "mihevc_last_error", "mihevc_cost_params_for_qp", "mihevc_tile_grid", "mihevc_p_tile_grid", "mihevc_k_transform",
    "mihevc_k_intra_frame", "mihevc_k_inter_frame", "mihevc_k_b_frame", "mihevc_k_deblock", "mihevc_k_sao", "mihevc_k_loop_filter", "mihevc_write_parameter_sets",
    "mihevc_encode_picture_host",
)

_lib = None


class MihevcError(RuntimeError):
    def __init__(self, code: int, what: str = ""):
        self.code = code
        msg = strerror(code)
        super().__init__(f"{what}: {msg} ({code})" if what else f"{msg} ({code})")


def load() -> C.CDLL:
    """Load libmihevc.so; raises OSError/AttributeError loudly when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise OSError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)")
    # several sessions in one process (threads of a batch, SlicedEncoder's bands) are 3 HIP streams each; by default the runtime maps all of a process's streams onto 4
    # hardware queues, and streams sharing one serialise (one session's uploads then wait behind the other's kernels: DESIGN.md §5).  Only read when the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(str(LIB_PATH))
    for name in EXPORTS:
        getattr(lib, name)          # AttributeError if the ABI is incomplete
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.mihevc_strerror.restype = C.c_char_p
    lib.mihevc_last_error.restype = C.c_char_p
    lib.mihevc_last_error.argtypes = [vp]
    lib.mihevc_config_default.argtypes = [C.POINTER(Config)]
    lib.mihevc_config_default.restype = None
    lib.mihevc_open.argtypes = [C.POINTER(Config), i32, C.POINTER(vp)]
    lib.mihevc_send_frame.argtypes = [vp, vp, vp, vp, i32, i32, i64]
    lib.mihevc_send_frame_device.argtypes = [vp, vp, vp, vp, i32, i32, i64]
    lib.mihevc_send_frame_async.argtypes = [vp, vp, vp, vp, i32, i32, i64]
    lib.mihevc_send_frames_device.argtypes = [vp, i32, vp, vp, vp, i32, i32, i64]
    lib.mihevc_sync_uploads.argtypes = [vp]
    lib.mihevc_receive_packet.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    lib.mihevc_flush.argtypes = [vp]
    lib.mihevc_abort.argtypes = [vp]
    lib.mihevc_close.argtypes = [vp]
    lib.mihevc_close.restype = None
    lib.mihevc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.mihevc_get_headers.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    lib.mihevc_set_keep_recon.argtypes = [vp, i32]
    lib.mihevc_get_recon.argtypes = [vp, i64, vp, vp, vp]
    lib.mihevc_coded_size.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.mihevc_get_frame_info.argtypes = [vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    lib.mihevc_cost_params_for_qp.argtypes = [i32, i32, i32, C.POINTER(CostParams)]
    lib.mihevc_cost_params_for_qp.restype = None
    lib.mihevc_k_transform.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32]
    lib.mihevc_k_intra_frame.argtypes = [i32, vp, vp, vp, i32, i32, C.POINTER(CostParams), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mihevc_k_inter_frame.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(CostParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mihevc_k_b_frame.argtypes = [i32] + [vp] * 9 + [i32, i32, C.POINTER(CostParams)] + [vp] * 12
    lib.mihevc_k_deblock.argtypes = [i32, vp, vp, vp, i32, i32, vp, i32]
    lib.mihevc_k_sao.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(CostParams), vp, vp, vp, vp]
    lib.mihevc_k_loop_filter.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, C.POINTER(CostParams), vp, vp, vp, vp]
    lib.mihevc_write_parameter_sets.argtypes = [C.POINTER(Config), vp, C.c_size_t]
    lib.mihevc_encode_picture_host.argtypes = [C.POINTER(Config), i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_size_t]
    _lib = lib
    return lib


def strerror(code: int) -> str:
    try:
        return load().mihevc_strerror(code).decode()
    except Exception:
        return f"mihevc error {code}"


def default_config() -> Config:
    cfg = Config()
    load().mihevc_config_default(C.byref(cfg))
    return cfg


def cost_params(qp: int, bit_depth: int = 8, me_range: int = 16) -> CostParams:
    p = CostParams()
    load().mihevc_cost_params_for_qp(qp, bit_depth, me_range, C.byref(p))
    return p


def p_tile_grid(cfg: Config):
    """(columns, rows) of the tile grid P pictures of this configuration are coded with (mihevc_p_tile_grid); (1, 1) when off."""
    c, r = C.c_int(), C.c_int()
    rc = load().mihevc_p_tile_grid(C.byref(cfg), C.byref(c), C.byref(r))
    if rc != 0:
        raise MihevcError(rc, "mihevc_p_tile_grid")
    return c.value, r.value


def tile_grid(cfg: Config):
    """(columns, rows) of the tile grid IDR pictures of this configuration are coded with (mihevc_tile_grid)."""
    c, r = C.c_int(), C.c_int()
    rc = load().mihevc_tile_grid(C.byref(cfg), C.byref(c), C.byref(r))
    if rc != 0:
        raise MihevcError(rc, "mihevc_tile_grid")
    return c.value, r.value
