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


@pytest.mark.parametrize("c_local,n_per_rank", [(1, [1000]), (2, [700, 0, 513]), (4, [257, 300, 255, 1]), (3, [5])])
def test_mapped_and_message_copies_match_torch(c_local, n_per_rank):
    """gsx_copy_column_groups_mapped and gsx_copy_message_columns (csrc/rows.hip) against torch indexing: the exchange-order
    <-> [C_local][sum N] row map (sources of unequal and zero size), both directions, contiguous fields and column views,
    tiles that straddle source blocks, a last partial tile. Bit-exact."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    from gsplat_amd import _cabi
    from gsplat_amd.distributed import _RowMap, _copy_groups, _copy_message

    rm = _RowMap(c_local, n_per_rank)
    R = rm.rows
    idx = rm.index("cuda")
    assert sorted(idx.tolist()) == list(range(R))
    g = torch.Generator().manual_seed(R)
    msg = torch.randn(R, 9, generator=g).cuda()
    # message (exchange order) -> five field arrays in [C_local][sum N] order
    fields = [torch.full((R, w), float("nan"), device="cuda") for w in (2, 1, 3, 1, 2)]
    cols = [0, 2, 3, 6, 7]
    for use_map in (True, False):
        m = rm if use_map else None
        _copy_message(msg, cols, fields, to_msg=False, row_map=m)
        for f, c in zip(fields, cols):
            want = torch.empty_like(f)
            want[idx if use_map else torch.arange(R, device="cuda")] = msg[:, c:c + f.shape[1]]
            assert torch.equal(f, want)
        # and back, with two of the fields read through column views of wider rows
        wide = torch.randn(R, 9, generator=g).cuda()
        srcs = [wide[:, 0:2], fields[1], wide[:, 2:5], fields[3], fields[4]]
        back = torch.full((R, 9), float("nan"), device="cuda")
        _copy_message(back, cols, srcs, to_msg=True, row_map=m)
        for s_, c in zip(srcs, cols):
            pick = s_[idx] if use_map else s_
            assert torch.equal(back[:, c:c + s_.shape[1]], pick)
        # the per-word kernel with the same map (a message whose rows are not fully covered takes this path)
        part = torch.zeros(R, 9, device="cuda")
        _copy_groups([srcs[0], srcs[2]], [part[:, 0:2], part[:, 3:6]], R, m, map_dst=False)
        assert torch.equal(part[:, 0:2], srcs[0][idx] if use_map else srcs[0])
        assert torch.equal(part[:, 3:6], srcs[2][idx] if use_map else srcs[2]) and bool((part[:, 2] == 0).all())
    with pytest.raises(ValueError):  # a packed message must be covered by its groups
        _cabi.copy_message_columns(msg.data_ptr(), 9, R, [(0, 2, fields[0].data_ptr(), 2)], True)
    with pytest.raises(ValueError):  # segments that do not add up to the message
        _cabi.copy_message_columns(msg.data_ptr(), 9, R, [(0, 2, fields[0].data_ptr(), 2)], False, [c_local * 3], [3])
