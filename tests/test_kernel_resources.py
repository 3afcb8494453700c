"""The register / LDS contract of the hot kernels, read from the compiler's resource report the build keeps next
to the objects (field_interpolation_amd/csrc/*.usage.txt, -Rpass-analysis=kernel-resource-usage).  Zero spills is
a performance requirement here, not a nicety: a scratch reload waits for every older global load and so drains
the software pipeline of the marching kernel (profiles/r1_ablation.md)."""
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "field_interpolation_amd", "csrc")


def _report(name):
    path = os.path.join(CSRC, name)
    if not os.path.exists(path):
        pytest.skip("%s not there (library built without the report)" % name)
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([\w /\[\]]+?):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _variants(report, kernel):
    return {k: v for k, v in report.items() if kernel in k}


def test_marching_kernel_budget():
    rep = _variants(_report("fi_stencil.usage.txt"), "k_apply_march3d")
    # {fp32, fp64} x {model_1, model_2, both} x {plain, fused} x {128 x 8, 64 x 16 tiles}, the fused ones once more
    # for contexts that keep their multi-row cells as packed blocks, the plain ones once more with the Chebyshev
    # epilogue of the polynomial preconditioner (template flag EPI, the last but one of the mangled name) and again with
    # the operand formed on load (PRO, the last flag), and the fp32 fused ones once more with the epilogue of the
    # V-cycle's smoother (register-allocated for 2 workgroups per CU)
    assert len(rep) == 72
    for name, r in rep.items():
        epi = "ELb1ELb0EEEv" in name or "ELb1ELb1EEEv" in name
        # (SGPR spills go to VGPR lanes, not to memory: the both-models variants keep 11-19 lane masks and bounds there,
        # the fused ones with the smoother's epilogue up to 40; the plain ones with the epilogue carry the storage format of
        # the polynomial's iterates as uniform state since round 4: 34 with both models on, 50 since round 5 -- the format is
        # looked at where a loaded vector is first used as well as where it is loaded)
        assert r["SGPRs Spill"] <= ((40 if r["LDS Size [bytes/block]"] > 30000 else 56) if epi else 24) and r["AGPRs"] == 0, name
        assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, name
        if r["LDS Size [bytes/block]"] > 30000:           # fused variants: 3 workgroups per CU (2 with both models)
            assert r["LDS Size [bytes/block]"] * 3 <= 160 * 1024, name
            assert r["VGPRs"] <= 256 and r["Occupancy [waves/SIMD]"] >= 2, name
        else:                                             # plain variants: 4 waves per SIMD, 4 workgroups per CU
            both32 = epi and "march3dIfLb1ELb1E" in name  # the epilogue with both models on in fp32: 3 waves per SIMD
            assert r["VGPRs"] <= (168 if both32 else 128) and r["Occupancy [waves/SIMD]"] >= (3 if both32 else 4), name
            assert r["LDS Size [bytes/block]"] * 4 <= 160 * 1024, name
    # the bench variant: fp32, model_2 only, fused -- three workgroups per CU
    bench = [r for n, r in rep.items() if "march3dIfLb0ELb1ELb1ELi32ELb0ELb0ELb0E" in n]
    assert len(bench) == 1 and bench[0]["VGPRs"] <= 168 and bench[0]["Occupancy [waves/SIMD]"] == 3
    # the fp64 fused variants without the epilogue (the CG's operator apply): THREE workgroups per CU by registers and by
    # LDS -- march_setup sizes their grids for three (fi_stencil.hip)
    f64 = [r for n, r in rep.items() if "march3dId" in n and r["LDS Size [bytes/block]"] > 30000 and "ELb0ELb0EEEv" in n]
    assert len(f64) == 12
    for r in f64:
        assert r["Occupancy [waves/SIMD]"] == 3 and r["LDS Size [bytes/block]"] * 3 <= 160 * 1024
    # the Chebyshev step of the bench: fp32, model_2 only, plain + epilogue -- four workgroups per CU, no spills
    for pro in "01":   # ... and its first step, which forms the operand while it loads r and the scaling
        cheb = [r for n, r in rep.items() if "march3dIfLb0ELb1ELb0ELi32ELb0ELb1ELb%sE" % pro in n]
        assert len(cheb) == 1 and cheb[0]["VGPRs"] <= 128 and cheb[0]["Occupancy [waves/SIMD]"] >= 4


def test_tile2d_kernel_budget():
    rep = _variants(_report("fi_stencil2d.usage.txt"), "k_apply_tile2d")
    assert len(rep) == 24          # the 12 of round 1, and again with the smoother's epilogue
    for name, r in rep.items():
        assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, name
        assert r["VGPRs"] <= 128 and r["LDS Size [bytes/block]"] <= 40 * 1024, name


def test_strip_kernel_budget():
    """fi_strip.hip (FI_STRIP=1, not the default): a wave per 128 x 4 strip with the z ring in registers -- one wave per SIMD by
    design (the plain variants of one model term fit three), no scratch in the variants with one model term (config 4 / 5);
    the two-role variants keep 27.5 KB of LDS, four workgroups per CU with the launch's 10 KB pad."""
    rep = _variants(_report("fi_strip.usage.txt"), "k_apply_strip3d")
    assert len(rep) == 9          # {model_1, model_2, both} x {plain, cells, cells + packed blocks}
    for name, r in rep.items():
        both = "strip3dIdLb1ELb1E" in name
        cells = r["LDS Size [bytes/block]"] > 0
        assert r["AGPRs"] == 0 and r["VGPRs"] <= 256 and r["Occupancy [waves/SIMD]"] >= 2, name
        if both and cells:   # 13 + 13 taps and the cell pipeline: 12 registers of the prologue spill (52 bytes per lane)
            assert r["VGPRs Spill"] <= 12 and r["ScratchSize [bytes/lane]"] <= 52, name
        else:
            assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, name
        if cells:
            assert (r["LDS Size [bytes/block]"] + 10 * 1024) * 4 <= 160 * 1024, name
