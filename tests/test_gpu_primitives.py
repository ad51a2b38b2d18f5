"""-m gpu: the reference's own primitive vectors on the DEVICE field and curve code (SURVEY.md section 8 rows a1-a3,
a6-a8): /root/reference test/f1.js:296-400 (Fq / Fr edge grid, Montgomery maps), test/bn128.js:84-185 (group law:
P+P, P-P, infinity operands, same point with different z), as harvested into tests/golden/fields.json and groups.json.
One lane per vector through wsnark_selftest_field / wsnark_selftest_curve of the hipcc-built libwsnark.so."""
import pytest

import primitives_common as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bn():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__
    __graft_entry__.ensure_built()
    import wasmsnark_amd
    b = wasmsnark_amd.build(device=0)
    assert b.lib.path.endswith("wasmsnark_amd/libwsnark.so")
    return b


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("fname,which", [("fq", 0), ("fr", 1)])
def test_base_field_vectors_on_device(bn, fname, which, impl):
    pc.check_base_field(bn, fname, which, impl)


@pytest.mark.parametrize("impl", [0, 1, 2])
def test_fq2_vectors_on_device(bn, impl):
    pc.check_fq2(bn, impl)


@pytest.mark.parametrize("g,impl", [(g, i) for g in (1, 2) for i in pc.CURVE_IMPLS[g]])
def test_group_vectors_on_device(bn, orc, g, impl):
    pc.check_group(bn, orc, g, impl)


@pytest.mark.parametrize("g,impl", [(1, 5), (2, 6)])
def test_lane_split_tail_curves_on_device(bn, orc, g, impl):
    pc.check_group_pair_g1(bn, orc, g, impl)
