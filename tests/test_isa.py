"""Static checks on the compiled gfx950 ISA of the kernels (hipcc cross-compiles without a GPU): no
spills/scratch, no partial-register (SDWA dst_sel) write immediately followed by a dependent VALU read,
no DPP, and no redefinition of a wide store's data VGPRs within two wait states (tools/check_isa.py).
The hand-written SDWA statements are opaque to the compiler's hazard recognizer, so these are checked
on the ISA that ships -- and on a second schedule of the same source (the 32-row tile build)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.timeout(600)
def test_fused_kernels_isa_is_hazard_free(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import check_isa
    # {dec x 2 interp x 2 seeded} and {enc x 2 interp x 2 ident x 2 seeded}: one translation unit per direction
    for tu, kernels, sdwa in (("hgi_fused_dec.hip", 4, 100), ("hgi_fused_enc.hip", 8, 400)):
        out = str(tmp_path / (tu + ".s"))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                               os.path.join(ROOT, "rustyhgi_amd", "csrc", tu), "-o", out], stderr=subprocess.DEVNULL)
        r = check_isa.check(out)
        assert r["kernels"] == kernels, (tu, r)
        assert r["partial_writes"] > sdwa, (tu, r)         # the SDWA paths are really there
        assert r["adjacent_dependent"] == 0, (tu, r["examples"])
        assert r["scratch_bytes"] == 0 and r["vgpr_spills"] == 0, (tu, r)
        assert r["store_data_overwritten"] == 0, (tu, "data VGPR of a wide store redefined too early", r)
        assert r["dpp"] == 0, (tu, "DPP next to opaque SDWA asm is not allowed", r)
    # the 128 x 32 tile build that ships next to it (a second schedule of the same source), and the level-wise kernels
    for tu, extra in (("hgi_fused_dec32.hip", []), ("hgi_fused_enc32.hip", []), ("hgi_kernels.hip", [])):
        out = str(tmp_path / (tu + ".alt.s"))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S"] + extra +
                              [os.path.join(ROOT, "rustyhgi_amd", "csrc", tu), "-o", out], stderr=subprocess.DEVNULL)
        r = check_isa.check(out)
        assert r["adjacent_dependent"] == 0 and r["store_data_overwritten"] == 0 and r["dpp"] == 0, (tu, extra, r)
        assert r["scratch_bytes"] == 0 and r["vgpr_spills"] == 0, (tu, extra, r)
