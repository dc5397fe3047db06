"""Static checks on the compiled gfx950 ISA of the kernels (hipcc cross-compiles without a GPU): no
spills/scratch, no partial-register (SDWA dst_sel) write immediately followed by a dependent VALU read,
no DPP, and no redefinition of a wide store's data VGPRs within two wait states (tools/check_isa.py).
The hand-written SDWA statements are opaque to the compiler's hazard recognizer, so these are checked
on the ISA that ships -- and on a second schedule of the same source (the 32-row tile build)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


CSRC = os.path.join(ROOT, "rustyhgi_amd", "csrc")


def _isa(tmp_path, tu):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (tu + ".s"))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           os.path.join(CSRC, tu), "-o", out], stderr=subprocess.DEVNULL)
    return out


@pytest.mark.timeout(900)
def test_every_translation_unit_is_hazard_free(tmp_path):
    """Every .hip file of the library (enumerated, not listed): both tile geometries of both directions -- two
    schedules of the same source -- the level-wise / harness kernels and the C ABI's own kernels."""
    import check_isa
    units = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(units) >= 6, units
    for path in units:
        tu = os.path.basename(path)
        r = check_isa.check(_isa(tmp_path, tu))
        assert r["adjacent_dependent"] == 0, (tu, r["examples"])
        assert r["store_data_overwritten"] == 0, (tu, "data VGPR of a buffer store redefined too early", r["examples"])
        # DPP is ruled out where inline SDWA asm hides register writes from the compiler's hazard bookkeeping (the codec
        # kernels); a unit without partial-register writes (the entropy stage's scans) leaves DPP hazards to the compiler
        if r["partial_writes"]:
            assert r["dpp"] == 0, (tu, "DPP next to opaque SDWA asm is not allowed", r)
        assert r["scratch_bytes"] == 0 and r["vgpr_spills"] == 0, (tu, r)
        assert r["traps"] == 0, (tu, "s_trap in shipped ISA: the device must never abort", r)


@pytest.mark.timeout(600)
def test_fused_kernels_are_the_sdwa_builds(tmp_path):
    import check_isa
    # {dec x 2 interp x (unseeded | seed planes | seeds rebuilt in the kernel)} and {enc x 2 interp x 2 ident x
    # (unseeded | seed planes | seeds rebuilt in the kernel)}: one translation unit per direction
    for tu, kernels, sdwa in (("hgi_fused_dec.hip", 6, 100), ("hgi_fused_enc.hip", 12, 400)):
        r = check_isa.check(_isa(tmp_path, tu))
        assert r["kernels"] == kernels, (tu, r)
        assert r["partial_writes"] > sdwa, (tu, r)         # the SDWA paths are really there


def test_checker_sees_the_hazards_it_guards_against(tmp_path):
    """The rules fire on hand-made ISA: wide store + VALU write next to it, a b64 store with SGPR soffset, a trap."""
    import check_isa
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join([
        "buffer_store_dwordx4 v[12:15], v21, s[28:31], s40 offen nt", "v_lerp_u8 v12, v1, v2, v3",
        "buffer_store_dwordx2 v[4:5], v21, s[28:31], s40 offen", "v_mov_b32_e32 v5, 0",
        "buffer_store_dwordx2 v[6:7], v21, s[28:31], 0 offen", "v_mov_b32_e32 v6, 0",      # literal soffset: LLVM pads
        "s_trap 2", ""]))
    r = check_isa.check(str(bad))
    assert r["store_data_overwritten"] == 2 and r["traps"] == 1
    good = tmp_path / "good.s"
    good.write_text("\n".join(["buffer_store_dwordx4 v[12:15], v21, s[28:31], s40 offen nt", "s_nop 1",
                               "v_lerp_u8 v12, v1, v2, v3", ""]))
    r = check_isa.check(str(good))
    assert r["store_data_overwritten"] == 0 and r["traps"] == 0
