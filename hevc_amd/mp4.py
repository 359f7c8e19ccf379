"""Minimal ISO-BMFF writer for one HEVC video track, Apple style.

Produces what the reference asks ffmpeg's mov muxer for (core/transcoder.py:466,490-492): sample entry `hvc1`
(parameter sets only in `hvcC`), major brand `mp42`, a `colr` nclx box (+write_colr), `moov` before `mdat`
(+faststart), handler name `VideoHandler` (core/transcoder.py:421), plus `mdcv`/`clli` for HDR10.
MP4 mux stays on host cores by design (BASELINE.json north_star).
"""
from __future__ import annotations

import os
import struct
from pathlib import Path
from typing import List, Tuple

from . import _lib


def box(kind: bytes, *payload: bytes) -> bytes:
    body = b''.join(payload)
    return struct.pack('>I4s', 8 + len(body), kind) + body


def full_box(kind: bytes, version: int, flags: int, *payload: bytes) -> bytes:
    return box(kind, struct.pack('>I', (version << 24) | flags), *payload)


def split_annexb(data: bytes) -> List[bytes]:
    """Annex-B byte stream -> list of NAL units (without start codes)."""
    out, i, n = [], 0, len(data)
    starts = []
    while i + 3 <= n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    for k, s in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else n
        while e > s and data[e - 1] == 0 and k + 1 < len(starts):      # zero_byte of a 4-byte start code
            e -= 1
        out.append(data[s:e])
    return out


def unescape(nal: bytes) -> bytes:
    out, zeros = bytearray(), 0
    for b in nal:
        if zeros >= 2 and b == 3:
            zeros = 0
            continue
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def nal_type(nal: bytes) -> int:
    return (nal[0] >> 1) & 63


def hvcc_box(parameter_sets: bytes, bit_depth: int) -> bytes:
    """ISO/IEC 14496-15 8.3.3.1 HEVCDecoderConfigurationRecord from Annex-B VPS/SPS/PPS."""
    nals = split_annexb(parameter_sets)
    by_type = {t: [n for n in nals if nal_type(n) == t] for t in (32, 33, 34)}
    sps = unescape(by_type[33][0])
    ptl = sps[3:15]                       # after 2-byte NAL header + 1 byte (vps id, sub-layers, nesting): 12 bytes of PTL
    rec = bytearray([1]) + ptl[:1] + ptl[1:5] + ptl[5:11] + ptl[11:12]
    rec += struct.pack('>H', 0xF000)      # reserved 1111 + min_spatial_segmentation_idc 0
    rec += bytes([0xFC, 0xFC | 1, 0xF8 | (bit_depth - 8), 0xF8 | (bit_depth - 8)])   # parallelism 0, chroma 4:2:0, depths
    rec += struct.pack('>H', 0)           # avgFrameRate
    rec += bytes([(0 << 6) | (1 << 3) | (1 << 2) | 3])   # constantFrameRate 0, numTemporalLayers 1, temporalIdNested 1, 4-byte lengths
    rec += bytes([3])
    for t in (32, 33, 34):
        rec += bytes([0x80 | t]) + struct.pack('>H', len(by_type[t]))
        for n in by_type[t]:
            rec += struct.pack('>H', len(n)) + n
    return box(b'hvcC', bytes(rec))


class Mp4Writer:
    def __init__(self, path: Path, cfg: _lib.Config):
        self.path, self.cfg = Path(path), cfg
        self._tmp = self.path.with_name(self.path.name + '.mdat.tmp')
        self._f = open(self._tmp, 'wb')
        self._sizes: List[int] = []
        self._sync: List[int] = []
        self._pts: List[int] = []
        self._cts: List[int] = []          # composition offset of every sample in frames (pts - dts): non-zero only with B pictures (cfg.bframes)

    def add_sample(self, annexb: bytes, pts: int, keyframe: bool, dts: int = None):
        """One access unit, in DECODING order; parameter sets are dropped (hvc1 keeps them in hvcC only), NALs get 4-byte lengths.  dts (same unit as
        pts: frames) only differs from pts when B pictures reorder: the track then gets a ctts box and an edit list that takes the first delay out."""
        size = 0
        for nal in split_annexb(annexb):
            if nal_type(nal) in (32, 33, 34):
                continue
            self._f.write(struct.pack('>I', len(nal)))
            self._f.write(nal)
            size += 4 + len(nal)
        self._sizes.append(size)
        self._pts.append(pts)
        self._cts.append(0 if dts is None else max(0, pts - dts))
        if keyframe:
            self._sync.append(len(self._sizes))

    def abort(self):
        self._f.close()
        try:
            os.remove(self._tmp)
        except OSError:
            pass

    def _stsd(self, headers: bytes) -> bytes:
        c = self.cfg
        colr = box(b'colr', b'nclx', struct.pack('>HHHB', c.colour_primaries, c.transfer, c.matrix, 0x80 if c.full_range else 0))
        extra = b''
        if c.hdr10:
            prim = b''.join(struct.pack('>HH', c.md_primaries[i][0], c.md_primaries[i][1]) for i in range(3))
            extra += box(b'mdcv', prim, struct.pack('>HHII', c.md_white[0], c.md_white[1], c.md_max_lum, c.md_min_lum))
            extra += box(b'clli', struct.pack('>HH', c.max_cll, c.max_fall))
        entry = (b'\0' * 6 + struct.pack('>H', 1) + b'\0' * 16 + struct.pack('>HH', c.width, c.height) +
                 struct.pack('>II', 0x00480000, 0x00480000) + b'\0' * 4 + struct.pack('>H', 1) + b'\0' * 32 + struct.pack('>Hh', 0x18, -1))
        return full_box(b'stsd', 0, 0, struct.pack('>I', 1), box(b'hvc1', entry, hvcc_box(headers, c.bit_depth), colr, extra))

    def finish(self, headers: bytes):
        self._f.close()
        c, n = self.cfg, len(self._sizes)
        ts, delta = c.fps_num, c.fps_den
        dur = n * delta
        ftyp = box(b'ftyp', b'mp42', struct.pack('>I', 0), b'mp42', b'isom', b'hvc1')
        mat = struct.pack('>9I', 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
        stts = full_box(b'stts', 0, 0, struct.pack('>III', 1, n, delta))
        stss = full_box(b'stss', 0, 0, struct.pack('>I', len(self._sync)), b''.join(struct.pack('>I', s) for s in self._sync))
        stsc = full_box(b'stsc', 0, 0, struct.pack('>IIII', 1, 1, n, 1))
        stsz = full_box(b'stsz', 0, 0, struct.pack('>II', 0, n), b''.join(struct.pack('>I', s) for s in self._sizes))

        payload = sum(self._sizes)
        big_mdat = payload + 8 >= 1 << 32
        head = 16 if big_mdat else 8
        # the chunk offset is ftyp + moov + mdat header: 64-bit (co64) exactly when the file needs a 64-bit mdat or moov alone could reach
        # 4 GiB — decided ONCE, before moov is sized (sizing moov with stco and then switching to co64 left the offset 4 bytes short)
        big = big_mdat or 4 * len(self._sizes) + (1 << 16) >= 1 << 32

        # composition offsets (ctts, version 0: pts - dts >= 0 by construction) as runs, and the edit that starts the presentation at the first picture's time
        ctts, edts = b'', b''
        if any(self._cts):
            runs = []
            for v in self._cts:
                if runs and runs[-1][1] == v:
                    runs[-1][0] += 1
                else:
                    runs.append([1, v])
            ctts = full_box(b'ctts', 0, 0, struct.pack('>I', len(runs)), b''.join(struct.pack('>II', k, v * delta) for k, v in runs))
            edts = box(b'edts', full_box(b'elst', 0, 0, struct.pack('>I', 1), struct.pack('>IIHH', dur, min(self._cts[:1] or [0]) * delta, 1, 0)))

        def moov_with(chunk_offset: int) -> bytes:
            stco = full_box(b'co64' if big else b'stco', 0, 0, struct.pack('>I', 1), struct.pack('>Q' if big else '>I', chunk_offset))
            stbl = box(b'stbl', self._stsd(headers), stts, ctts, stss, stsc, stsz, stco)
            minf = box(b'minf', full_box(b'vmhd', 0, 1, b'\0' * 8), box(b'dinf', full_box(b'dref', 0, 0, struct.pack('>I', 1), full_box(b'url ', 0, 1))), stbl)
            mdia = box(b'mdia', full_box(b'mdhd', 0, 0, struct.pack('>IIIIHH', 0, 0, ts, dur, 0x55C4, 0)),
                       full_box(b'hdlr', 0, 0, b'\0' * 4, b'vide', b'\0' * 12, b'VideoHandler\0'), minf)
            tkhd = full_box(b'tkhd', 0, 3, struct.pack('>IIIII', 0, 0, 1, 0, dur), b'\0' * 8, struct.pack('>HHHH', 0, 0, 0, 0), mat,
                            struct.pack('>II', c.width << 16, c.height << 16))
            mvhd = full_box(b'mvhd', 0, 0, struct.pack('>IIII', 0, 0, ts, dur), struct.pack('>IH', 0x10000, 0x100), b'\0' * 10, mat, b'\0' * 24,
                            struct.pack('>I', 2))
            return box(b'moov', mvhd, box(b'trak', tkhd, edts, mdia))

        moov = moov_with(0)
        moov = moov_with(len(ftyp) + len(moov) + head)        # box sizes do not depend on the offset VALUE once its width is fixed
        self.layout = {'co64': big, 'big_mdat': big_mdat, 'chunk_offset': len(ftyp) + len(moov) + head}
        if getattr(self, 'dry_run', False):                   # tests: layout decisions for sizes that cannot be written for real
            self._moov = moov
            return
        with open(self.path, 'wb') as out, open(self._tmp, 'rb') as src:
            out.write(ftyp)
            out.write(moov)
            out.write(struct.pack('>I4sQ', 1, b'mdat', payload + 16) if big_mdat else struct.pack('>I4s', payload + 8, b'mdat'))
            while True:
                chunk = src.read(1 << 20)
                if not chunk:
                    break
                out.write(chunk)
        os.remove(self._tmp)


def parse_boxes(data: bytes, start: int = 0, end: int = None) -> List[Tuple[str, int, int]]:
    """[(type, payload_start, payload_end)] of the boxes in data[start:end] (tests / tooling)."""
    end = len(data) if end is None else end
    out, i = [], start
    while i + 8 <= end:
        size, kind = struct.unpack('>I4s', data[i:i + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack('>Q', data[i + 8:i + 16])[0]
            hdr = 16
        if size < hdr:
            break
        out.append((kind.decode('latin1'), i + hdr, i + size))
        i += size
    return out
