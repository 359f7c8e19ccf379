"""hevc_amd — MI355X-native HEVC encode path behind the reference's `convert_video` contract.

Layout (only what the hot path needs):
    transcoder.py / probe.py / utils.py   host mirror of the reference's core/ package (the boundary)
    encoder.py                            ctypes driver of the C ABI in include/mihevc.h
    mp4.py, yuvio.py, batch.py            mux, raw-clip IO + synthetic clips, headless batch queue
    csrc/                                 HIP kernels (gfx950), host CABAC/bitstream, the C ABI
"""
__version__ = "0.1.0"
