"""CPU: every gfx950 kernel of the product builds without scratch (register spills) — hipcc cross-compiles without a GPU and reports
per-kernel resources with -Rpass-analysis=kernel-resource-usage.  Round 1 shipped k_inter_ctu with 88 bytes of scratch per lane (4x the
algorithmic HBM traffic); a later edit of round 2 silently pushed k_me_search back into spills — this keeps it from recurring."""
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "hevc_amd" / "csrc"


def test_no_kernel_uses_scratch(tmp_path):
    flags = re.search(r"CXXFLAGS \?= (.*)", (CSRC / "Makefile").read_text()).group(1).split()
    p = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", str(CSRC / "device.hip"),
                        "-o", str(tmp_path / "device.o")], capture_output=True, text=True, cwd=CSRC, timeout=1200)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    assert len(blocks) >= 30                       # 8-bit and 10-bit instances of every kernel
    seen = {}
    for b in blocks:
        name = b.split(" [")[0]
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgpr = int(re.search(r" VGPRs: (\d+)", b).group(1))
        seen[name] = (vgpr, scratch)
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane ({vgpr} VGPRs)"
    inter = [v for k, v in seen.items() if "k_inter_ctu" in k and "k_inter_ctu_b" not in k]
    assert inter and all(v[0] <= 128 for v in inter)            # 4 workgroups per CU
    inter_b = [v for k, v in seen.items() if "k_inter_ctu_b" in k]
    assert inter_b and all(v[0] <= 168 for v in inter_b)        # the B form (list-1 refinement + bi-prediction trial): 3 workgroups per CU
