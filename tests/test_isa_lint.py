"""Build-time fence for the packed-fp32 operand form that goes wrong on MI355X (round 5): no translation unit of the
library may contain a v_pk_{mul,add,fma}_f32 whose low half selects the HIGH register of a VGPR pair in the src1 / src2
position (tools/pk_opsel_lint.py has the measurement and the rule).  CPU only: hipcc cross-compiles to assembly."""
import os
import shutil
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_lint_recognises_the_bad_and_the_good_forms():
    import pk_opsel_lint as lint
    asm = "\n".join([
        "_Z6kernelPf:",
        "\tv_pk_mul_f32 v[78:79], v[78:79], v[66:67] op_sel:[0,1]",                      # src1 hi -> lo: bad
        "\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,0,1]",  # src2 hi -> lo: bad
        "\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel:[1,0,0]",                   # src0: fine
        "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]",                          # lo -> hi: fine
        "\tv_pk_fma_f32 v[2:3], v[4:5], s[6:7], v[2:3] op_sel:[0,1,0]",                   # scalar pair: fine
        "\tv_pk_add_f32 v[6:7], v[6:7], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]",             # bad
    ])
    hits = lint.offending(asm)
    assert [h[1] for h in hits] == [2, 3, 7] and all(h[0] == "_Z6kernelPf" for h in hits)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_translation_unit_contains_the_bad_packed_operand_form():
    import pk_opsel_lint as lint
    assert lint.main([]) == 0
