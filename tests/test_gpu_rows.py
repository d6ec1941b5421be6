"""gsx_copy_column_groups (csrc/rows.hip): pack / unpack of row messages against torch slicing, bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_copy_column_groups_pack_unpack():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    from gsplat_amd import _cabi

    R = 100_003
    g = torch.Generator().manual_seed(1)
    a, b, c = (torch.randn(R, w, generator=g).cuda() for w in (2, 1, 3))
    ints = torch.randint(-5, 10_000, (R, 2), generator=g, dtype=torch.int32).cuda()
    aos = torch.randn(R, 11, generator=g).cuda()  # a strided source: columns 4..6 of wider rows
    msg = torch.full((R, 11), float("nan"), device="cuda")
    srcs = [a, b, c, ints.view(torch.float32), aos[:, 4:7]]
    offs = [0, 2, 3, 6, 8]
    _cabi.copy_column_groups([(s.data_ptr(), s.stride(0), msg[:, o:o + s.shape[1]].data_ptr(), 11, s.shape[1])
                              for s, o in zip(srcs, offs)], R)
    assert torch.equal(msg[:, 0:2], a) and torch.equal(msg[:, 2:3], b) and torch.equal(msg[:, 3:6], c)
    assert torch.equal(msg[:, 6:8].contiguous().view(torch.int32), ints) and torch.equal(msg[:, 8:11], aos[:, 4:7])
    # unpack two groups of the message into contiguous tensors
    o1, o2 = torch.empty(R, 3, device="cuda"), torch.empty(R, 2, dtype=torch.int32, device="cuda")
    _cabi.copy_column_groups([(msg[:, 3:6].data_ptr(), 11, o1.data_ptr(), 3, 3),
                              (msg[:, 6:8].data_ptr(), 11, o2.data_ptr(), 2, 2)], R)
    assert torch.equal(o1, c) and torch.equal(o2, ints)
    _cabi.copy_column_groups([(a.data_ptr(), 2, o1.data_ptr(), 3, 2)], 0)  # empty: no launch
    with pytest.raises(ValueError):
        _cabi.copy_column_groups([(a.data_ptr(), 1, o1.data_ptr(), 3, 2)], R)  # source stride below the width
