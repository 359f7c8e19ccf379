#!/usr/bin/env python3
"""Generate tests/golden/params.json by importing the reference's pure-Python policy code.

Run ONLY in the build container (needs /root/reference on disk):
    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_param_goldens.py
The output is data (inputs + the values the reference returned); no reference source is copied.
Covers SURVEY.md §8 rows a1-a8, a10 (core/transcoder.py, core/probe.py, core/utils.py).
"""
import json, sys, itertools, tempfile
from pathlib import Path

sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
import core.transcoder as T          # noqa: E402
import core.utils as U               # noqa: E402
import core.probe as P               # noqa: E402

OUT = Path(__file__).with_name("params.json")


def vi(w, h, fps, hdr, nb, dur, ach=0, md="", cll=""):
    if hdr:
        return P.VideoInfo(w, h, fps, 'bt2020', 'smpte2084', 'bt2020nc', 'yuv420p10le', md, cll, ach, True, 'eng', nb, dur)
    return P.VideoInfo(w, h, fps, 'bt709', 'bt709', 'bt709', 'yuv420p', md, cll, ach, False, 'eng', nb, dur)


def vi_dict(i):
    return dict(width=i.width, height=i.height, fps=i.fps, color_primaries=i.color_primaries,
                color_transfer=i.color_transfer, color_space=i.color_space, pix_fmt=i.pix_fmt,
                master_display=i.master_display, max_cll=i.max_cll, audio_channels=i.audio_channels,
                hdr=i.hdr, audio_language=i.audio_language, nb_frames=i.nb_frames, duration=i.duration)


RES = [(640, 480), (1280, 720), (1920, 1080), (1080, 1920), (2560, 1440), (3840, 2160), (7680, 4320), (320, 240), (4096, 2160)]
FPS = [23.976, 24.0, 25.0, 29.97, 30.0, 50.0, 59.94, 60.0, 120.0]
DUR = [(None, 10.0), (None, None), (None, 600.0), (9000, None)]

cases = []
for (w, h), fps, hdr, (nb, dur) in itertools.product(RES, FPS, [False, True], DUR):
    if nb is None and dur is not None and dur == 10.0:
        nbx = None
    else:
        nbx = nb
    info = vi(w, h, fps, hdr, nbx, dur)
    rec = {"info": [w, h, fps, hdr, nbx, dur]}   # compact: tests rebuild the VideoInfo via the same vi() recipe
    rec["level"] = list(T.calculate_apple_hevc_level(info))
    rec["nvenc_level"] = list(T.calculate_nvenc_hevc_level(info))
    rec["dynamic"] = list(T.calculate_dynamic_values(info))
    rec["nvenc_preset"] = T.select_nvenc_preset(info, "unknown")
    for use_nvenc in ((False, True) if dur == 10.0 else (False,)):   # NVENC branch is shape-only: sample it
        p = T.build_ffmpeg_params(info, use_nvenc, "unknown")
        key = "params_nvenc" if use_nvenc else "params_cpu"
        rec[key] = [p.vcodec, p.pix_fmt, p.profile, p.level, p.color_flags,
                    "\x1f".join(str(x) for x in p.vparams), "\x1f".join(p.hdr_metadata)]
    cases.append(rec)

gop = []
for fps in FPS + [15.0, 12.5, 47.952, 100.0, 119.88, 240.0, 1.0, 0.5, 7.0, 29.5]:
    for sec in (2.0, 2.5, 3.0, 2.1, 2.625, 3.15, 1.0, 8.0):
        gop.append({"fps": fps, "sec": sec, "gop": T.compute_aligned_gop(fps, sec)})
    gop.append({"fps": fps, "sec": 3.0, "max": 120, "gop": T.compute_aligned_gop(fps, 3.0, 120)})

audio = [{"channels": c, "flags": T.get_audio_flags(c)} for c in [None, 0, -1, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16]]

cmds = []
for (w, h, fps, hdr, ach, lang) in [(1920, 1080, 30.0, False, 0, 'eng'), (1280, 720, 30.0, False, 2, 'jpn'),
                                    (3840, 2160, 30.0, True, 6, None), (7680, 4320, 30.0, True, 0, 'eng'),
                                    (1920, 1080, 29.97, False, 1, 'und'), (3840, 2160, 60.0, True, 8, 'fra')]:
    info = vi(w, h, fps, hdr, None, 10.0, ach)
    for use_nvenc in (False, True):
        p = T.build_ffmpeg_params(info, use_nvenc, "unknown")
        cmd = T.build_ffmpeg_command(Path('in.mov'), Path('out') / 'in.mp4', p, ach, lang)
        cmds.append({"info": vi_dict(info), "use_nvenc": use_nvenc, "lang": lang, "cmd": cmd})
        if use_nvenc:
            for a in range(0, 7):
                cmds.append({"info": vi_dict(info), "use_nvenc": True, "lang": lang, "attempt": a,
                             "adjusted": T.adjust_nvenc_params(p.vparams, a)})

hdrmeta = []
for md, cll in [("", ""), ("G(1,2)B(3,4)R(5,6)WP(7,8)L(9,10)", "600,300"), ("  ", " 1000,200 "), (None, None)]:
    for nv in (False, True):
        hdrmeta.append({"master_display": md, "max_cll": cll, "use_nvenc": nv,
                        "out": U.build_hdr_metadata(md, cll, nv)})

ens = []
for vp in [[], ['-aud', '1'], ['-x265-params', 'aud=1'], ['-rc', 'vbr'], ['-chromaloc', '0', '-rc', 'vbr']]:
    for enc in ('x265', 'nvenc'):
        ens.append({"vparams": vp, "encoder": enc, "out": T.ensure_bitstream_headers(vp, encoder=enc)})

# contract of the boundary with ffmpeg/ffprobe absent (this container): SURVEY §8b
calls = []
with tempfile.TemporaryDirectory() as td:
    res = T.convert_video(Path(td) / "clip.mp4", Path(td), progress_callback=lambda *a: calls.append(list(a)),
                          skip_validator=True, force_cpu=True)
fallback = vi_dict(P.probe_media(Path("/nonexistent/clip.mp4")))
decide = [{"force_cpu": a, "force_gpu": b, "out": T.decide_encoder(None, a, b)} for a in (False, True) for b in (False, True)]

doc = {
    "_generator": "tests/golden/make_param_goldens.py (imports /root/reference/core/*.py; data only)",
    "cases": cases, "gop": gop, "audio": audio, "commands": cmds, "hdr_metadata": hdrmeta,
    "ensure_headers": ens,
    "no_ffmpeg_contract": {"result": res, "callbacks": calls, "probe_fallback": fallback, "decide_encoder": decide,
                           "has_nvenc": U.has_nvenc(), "detect_gpu_type": U.detect_gpu_type()},
    "level_limits": {k: list(v) for k, v in T.HEVC_LEVEL_LIMITS.items()},
    "parse_fps": [{"in": s, "out": P.parse_fps(s)} for s in ["30/1", "30000/1001", "0/0", "", "abc", "25", "24/0", None]],
}
OUT.write_text(json.dumps(doc, separators=(",", ":"), sort_keys=True))
print("wrote", OUT, OUT.stat().st_size, "bytes;", len(cases), "cases")
