"""ISA hazard lint over the hand-written gfx950 kernels (CPU: hipcc cross-compiles to ISA without a GPU).

The kernels carry multi-instruction inline-asm blocks whose wait states the compiler's hazard recognizer cannot see and the
hardware does not interlock (MFMA result -> VALU read, VALU write -> DPP / permlane read, ...).  tools/isa_hazard_lint.py
walks the final instruction stream of every kernel and checks them; the MFMA wait counts are calibrated from what the
compiler itself inserts for a dependent pair.  Green at HEAD; red when the `s_nop 15; s_nop 3` that covers the block-4
epilogue's reads of the accumulators is compiled out (the bug of round 2, kept as a negative test), and red when the
`s_nop 1` that ends the f16-pair split (VALU write -> matrix operand read, rule R6) is compiled out -- round 3 removed it
once, every GPU test stayed green, and only this rule, added afterwards, says why it has to be there."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_hazard_lint as lint  # noqa: E402

CSRC = os.path.join(REPO, "keyword-spotting_amd", "csrc")
pytestmark = pytest.mark.skipif(not os.path.exists(lint.HIPCC), reason="hipcc not installed")


def test_calibration_matches_the_documented_wait_states():
    waits = lint.calibrate_mfma_waits()
    # XDL write -> VALU read on gfx950: passes + 3 (8-pass bf16 32x32x16 -> 11..12, 16-pass f32 32x32x2 -> 18..19)
    assert 10 <= waits["v_mfma_f32_32x32x16_bf16"] <= 13, waits
    assert 17 <= waits["v_mfma_f32_32x32x2_f32"] <= 20, waits


@pytest.mark.parametrize("unit", ["kws_dscnn.hip", "kws_cnntrad.hip", "kws_mfcc.hip", "kws_mfcc_f64.hip"])
def test_no_unprotected_hazard_in_the_product_kernels(unit):
    findings, waits, isa = lint.lint_file(os.path.join(CSRC, unit))
    flat = [(fn[:60], line, rule, msg) for fn, fs in findings.items() for line, rule, msg in fs]
    assert not flat, f"{unit} ({isa}): {flat[:5]}"
    # the walk saw the instructions it is there for
    body = open(isa).read()
    if unit in ("kws_dscnn.hip", "kws_cnntrad.hip"):
        assert body.count("v_mfma_f32_32x32x16_bf16") > 100
    assert "_dpp" in body or "row_shr" in body


def test_the_lint_fails_when_the_epilogue_wait_is_removed():
    findings, _, _ = lint.lint_file(os.path.join(CSRC, "kws_dscnn.hip"), ("-DKWS_X_NO_MFMA_EPILOGUE_NOP",))
    r1 = [(fn, f) for fn, fs in findings.items() for f in fs if f[1] == "R1"]
    assert r1, "removing `s_nop 15; s_nop 3` before the asm relu of the MFMA accumulators must be reported"
    assert any("v_max_f32" in f[2] for _, f in r1), r1[:3]   # the asm relu is the reader


def test_the_lint_fails_when_the_split_wait_is_removed():
    """Rule R6: the f16-pair split is an asm block whose last instructions write the registers the next matrix instruction
    reads as its B operand; the compiler pads one wait state after an asm block, two are required."""
    findings, _, _ = lint.lint_file(os.path.join(CSRC, "kws_dscnn.hip"), ("-DKWS_X_NO_SPLIT_NOP",))
    r6 = [(fn, f) for fn, fs in findings.items() for f in fs if f[1] == "R6"]
    assert r6, "removing the `s_nop 1` that ends split_pair8 must be reported"
    assert all("v_mfma_f32_32x32x16_f16" in f[2] for _, f in r6), r6[:3]
