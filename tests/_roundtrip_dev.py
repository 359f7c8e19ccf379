import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O

L = C.CDLL('/tmp/libmihevc_host.so')
class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width","height","fps_num","fps_den","bit_depth","level_idc","tier","crf","qp","vbv_maxrate_kbps","vbv_bufsize_kbits","keyint","min_keyint","colour_primaries","transfer","matrix","full_range","chroma_loc","aud","repeat_headers","hdr10")] + \
        [("md_primaries", C.c_uint16*6), ("md_white", C.c_uint16*2), ("md_max_lum", C.c_uint32), ("md_min_lum", C.c_uint32), ("max_cll", C.c_uint16), ("max_fall", C.c_uint16)] + \
        [(n, C.c_int32) for n in ("me_range","gops_in_flight","host_threads","sao")] + [("reserved", C.c_int32*8)]

def synth(h, w, seed, shift=(0,0)):
    rng = np.random.default_rng(seed)
    base = rng.normal(0, 1, (h+64, w+64))
    for _ in range(3):
        base = (base + np.roll(base,1,0) + np.roll(base,1,1) + np.roll(base,-1,0) + np.roll(base,-1,1))/5
    base = (base - base.min())/(base.max()-base.min())
    y = (base[32+shift[1]:32+shift[1]+h, 32+shift[0]:32+shift[0]+w]*200+20)
    yy, xx = np.mgrid[0:h,0:w]
    y = y + 20*np.sin(xx/7.0) 
    y = np.clip(y + rng.normal(0,1.5,(h,w)),0,255).astype(np.uint16)
    u = np.clip(128 + 30*np.sin(yy[::2,::2]/9.0) + rng.normal(0,1,(h//2,w//2)),0,255).astype(np.uint16)
    v = np.clip(128 + 30*np.cos(xx[::2,::2]/11.0)+ rng.normal(0,1,(h//2,w//2)),0,255).astype(np.uint16)
    return O.Frame(y,u,v)

def encode(cfg, st, poc, qp, a, sao):
    buf = (C.c_uint8 * (8<<20))()
    n = L.mihevc_encode_picture_host(C.byref(cfg), st, poc, qp, O._p(a.cu), O._p(a.coef_y), O._p(a.coef_u), O._p(a.coef_v), O._p(sao) if sao is not None else None, buf, len(buf))
    assert n > 0, n
    return bytes(buf[:n])

w, h = int(sys.argv[1]), int(sys.argv[2]); qp = int(sys.argv[3]); nfr = int(sys.argv[4])
cfg = Cfg(); L.mihevc_config_default(C.byref(cfg)); cfg.width, cfg.height = w, h; cfg.aud = 1
buf = (C.c_uint8 * 4096)()
n = L.mihevc_write_parameter_sets(C.byref(cfg), buf, 4096)
stream = bytes(buf[:n])
print("param sets", n, "bytes")
recs = []
ref = None
prm = O.default_params(qp, me_range=8)
for i in range(nfr):
    src = synth(h, w, 1, shift=(i*2, i))
    if i == 0:
        a = O.analyze_intra(src, O.default_params(qp-3) if False else prm)
    else:
        a = O.analyze_inter(src, ref, prm)
    dbk = O.deblock(a.rec, a.cu)
    out, sao = O.sao(src, dbk, prm)
    bs = encode(cfg, 2 if i == 0 else 1, i, qp, a, sao)
    print("frame", i, len(bs), "bytes", "psnr %.2f" % (10*np.log10(255**2/np.mean((out.y.astype(float)-src.y)**2))),
          "cu sizes", np.bincount(a.cu["log2_size"].ravel(), minlength=6)[3:], "sao types", np.bincount(sao["type"][:,0], minlength=3))
    stream += bs
    recs.append(out); ref = out
open('/tmp/rt.hevc','wb').write(stream)
try:
    frames, info = O.decode(stream)
except O.DecodeError as e:
    print("DECODE ERROR:", e); sys.exit(1)
print(len(frames), "decoded", info["width"], info["height"])
for i,(f,r) in enumerate(zip(frames, recs)):
    print(i, "match" if f.same(r) else "MISMATCH y=%d u=%d v=%d" % ((f.y!=r.y).sum(), (f.u!=r.u).sum(), (f.v!=r.v).sum()))
    if not f.same(r):
        ys, xs = np.nonzero(f.y != r.y)
        if len(ys): print(" first y diff at", xs[0], ys[0], "ctu", xs[0]//32, ys[0]//32)
