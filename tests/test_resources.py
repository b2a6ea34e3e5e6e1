"""CPU-side check of the gfx950 build's register allocation: no kernel may spill registers or use scratch memory.

The compiler's own per-kernel report (-Rpass-analysis=kernel-resource-usage, written by csrc/Makefile to csrc/build/*.res) is
parsed by tools/check_resources.py.  Round 3 shipped five kernels with VGPR spills (one of them inside a latency-bound
per-time-step loop) and four with SGPR spills without anyone noticing: a spill never shows up as a failure, only as time."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def test_no_kernel_spills_registers_or_uses_scratch():
    import check_resources
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("no hipcc on this host: the resource reports are produced by the gfx950 build")
    kernels = check_resources.parse(build=True)
    assert len(kernels) > 100, "resource report is missing (csrc/build/*.res): the Makefile writes it on every compile"
    bad = check_resources.offenders(kernels)
    assert not bad, "\n".join(f"{k['file']} {k['pretty']}: VGPR spill {k.get('vgpr_spill')}, SGPR spill {k.get('sgpr_spill')}, "
                              f"scratch {k.get('scratch')} B/lane" for k in bad)
    # launch-bound sanity of the kernels whose occupancy the design relies on
    by = {k["pretty"]: k for k in kernels}
    for name, occ in (("void k_stft_mel_p<double>", 3), ("void k_stft_mel_p<float>", 3)):
        assert by[name]["occ"] >= occ, (name, by[name])
