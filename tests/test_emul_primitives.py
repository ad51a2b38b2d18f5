"""CPU: the reference's field / group vectors through the product's arithmetic SOURCES (field29.h, fp2.h, curve.h,
field.h) compiled under the thread emulator -- the same wsnark_selftest_* entry points the GPU run uses
(tests/test_gpu_primitives.py).  Catches formula and bound mistakes here; the device code itself is judged on the GPU."""
import pytest

import primitives_common as pc
from emul_util import emul_bn128


@pytest.fixture(scope="module")
def bn():
    return emul_bn128()


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("fname,which", [("fq", 0), ("fr", 1)])
def test_base_field_vectors(bn, fname, which, impl):
    pc.check_base_field(bn, fname, which, impl)


@pytest.mark.parametrize("impl", [0, 1, 2])
def test_fq2_vectors(bn, impl):
    pc.check_fq2(bn, impl)


@pytest.mark.parametrize("g,impl", [(g, i) for g in (1, 2) for i in pc.CURVE_IMPLS[g]])
def test_group_vectors(bn, orc, g, impl):
    pc.check_group(bn, orc, g, impl)


@pytest.mark.parametrize("g,impl", [(1, 5), (2, 6)])
def test_lane_split_tail_curves(bn, orc, g, impl):
    pc.check_group_pair_g1(bn, orc, g, impl)


def test_unsupported_ops_are_errors(bn):
    with pytest.raises(Exception):
        pc.st_field(bn, 2, 0, pc.INVERSE, [bytes(64)], [bytes(64)])     # the extension field is inverted on the host only
    with pytest.raises(Exception):
        pc.st_field(bn, 2, 0, pc.TOMONT, [bytes(64)], [bytes(64)])      # not defined on Fq2
    with pytest.raises(Exception):
        pc.st_curve(bn, 1, 0, 9, [bytes(96)], [bytes(96)])
