"""Regression guard on the generated ISA of the LDS-DMA kernels that read transposed fragments (no GPU needed: hipcc
cross-compiles gfx950).  Round 5 found `s_waitcnt vmcnt(0)` put by the compiler in front of every stage's first
`__builtin_amdgcn_ds_read_tr16_b64` -- it drained the LDS-DMA ring, so rings of any depth ran as one slot (wgrad16, vocab_ce).
The reads go through inline asm since (csrc/dma_core.h::lds_tr16); this test fails if a builtin transposed read, or a
compiler-added vmcnt wait in front of one, comes back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _device_asm(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "--cuda-device-only", "-S", src, "-o", str(out)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return out.read_text()


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name", ["wgrad16.hip", "vocab_ce.hip"])
def test_transposed_fragment_reads_do_not_drain_the_dma_ring(name, tmp_path):
    asm = _device_asm(os.path.join(ROOT, "ark_amd", "csrc", name), tmp_path)
    kernels = re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", asm, flags=re.M | re.S)
    checked = 0
    for kname, body in kernels:
        if "load_lds" not in body or "ds_read_b64_tr_b16" not in body:
            continue
        checked += 1
        in_asm, prev = False, ""
        for line in body.splitlines():
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if t.startswith("ds_read_b64_tr_b16"):
                assert in_asm, f"{kname}: a transposed read outside inline asm (the builtin form is back)"
                assert not (prev.startswith("s_waitcnt") and "vmcnt" in prev and not prev_in_asm), \
                    f"{kname}: compiler-added `{prev}` in front of a transposed read"
            prev, prev_in_asm = t, in_asm
    assert checked >= 8, checked   # (every instantiation of the two kernel families)
