"""CPU: repository contract — the product never touches the oracle / emu harness / a CPU fallback."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "hevc_amd"


def _sources():
    for p in PKG.rglob("*"):
        if p.suffix in (".py", ".cpp", ".hip", ".h"):
            yield p


def test_product_never_references_the_oracle_or_the_emulator():
    bad = []
    for p in _sources():
        txt = re.sub(r"//[^\n]*", "", p.read_text())       # code only: comments may NAME the harness, code may not use it
        for pat in (r"\bimport\s+oracle\b", r"\bfrom\s+oracle\b", r"liboracle", r"libkernel_emu", r"tests[/.]emu", r"orc_[a-z_]+\("):
            if re.search(pat, txt):
                bad.append((p.name, pat))
    assert not bad, bad


def test_no_cuda_or_dual_backend_code():
    for p in _sources():
        txt = p.read_text()
        assert "__HIP_PLATFORM_AMD__" not in txt and "cuda_runtime" not in txt and "triton" not in txt.lower(), p
    # torch is plumbing for bench.py only; the package itself does not need it
    for p in PKG.rglob("*.py"):
        assert not re.search(r"^\s*(import|from)\s+torch\b", p.read_text(), re.M), p


def test_required_files_exist():
    for rel in ("include/mihevc.h", "oracle/hevc_oracle.c", "oracle/hevc_dec.c", "bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md",
                "tests/golden/params.json", "tests/golden/make_param_goldens.py"):
        assert (ROOT / rel).exists(), rel
    hdr = (ROOT / "oracle" / "hevc_oracle.h").read_text()
    assert "TEST INFRASTRUCTURE" in hdr and "PARITY UNPINNED" in hdr
