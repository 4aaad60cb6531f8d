"""CPU test of the division identity the sphere test relies on (csrc/rtk_trace.hip divide_by): a reciprocal computed
once per segment plus one fused-multiply-add correction gives the correctly rounded quotient, i.e. the same bits as the
reference's `/` (sphere.h:43,46)."""
import os
import subprocess
import tempfile

from tests.conftest import ROOT


def test_reciprocal_plus_fma_correction_is_the_correctly_rounded_quotient():
    src = os.path.join(ROOT, "tests", "helpers", "division_identity.c")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "division_identity")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe, "30000000"]).decode()
    assert "mismatches: 0" in out, out
    differs = int(out.split("uncorrected n*y differs:")[1].split()[0])
    assert differs > 1_000_000   # the correction is doing real work: a bare n * (1/a) is wrong in ~27 % of the cases
