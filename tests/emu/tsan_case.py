import sys, ctypes as C
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from tests import util
from oracle import oracle as O
from tests.test_bitstream_cpu import occluded_clip
emu = util.StageApi(C.CDLL(sys.argv[1]), "emu_")
prm = O.default_params(24, bit_depth=8, me_range=8)
prm.intra_nxn, prm.chroma_modes, prm.rdo_zero, prm.pre_search, prm.intra_in_p = 1, 1, 1, 1, 1
srcs = occluded_clip(72, 40, 8)
want, got = O.analyze_intra(srcs[0], prm), emu.intra(srcs[0], prm)
print("intra same", util.same_analysis(want, got))
dbk = O.deblock(want.rec, want.cu, 8)
ref, sp = O.sao(srcs[0], dbk, prm)
gref, gsp = emu.sao(srcs[0], dbk, prm)
print("sao same", ref.same(gref), bool((sp == gsp).all()) if hasattr(sp, "all") else sp == gsp)
want, got = O.analyze_inter(srcs[1], ref, prm, dump_me=True), emu.inter(srcs[1], ref, prm)
print("inter same", util.same_analysis(want, got))
