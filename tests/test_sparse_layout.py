"""Host logic of the sparse pixel-set path, on CPU tensors: the layout builder of gsplat_amd (torch index ops, device
agnostic) against the loop restatement of the reference's build_sparse_tile_layout in oracle/."""
import numpy as np
import pytest
import torch


def _impl():
    import gsplat_amd._ops as ops

    return ops.impl("build_sparse_tile_layout")


def _random_pixels(P, I, W, H, seed):
    g = torch.Generator().manual_seed(seed)
    flat = torch.randperm(I * H * W, generator=g)[:P]  # unique (image, row, col)
    img, rem = flat // (H * W), flat % (H * W)
    return torch.stack([rem // W, rem % W], -1).to(torch.int32), img.to(torch.int32)


@pytest.mark.parametrize("tile_size,W,H,I,P", [(16, 70, 50, 2, 500), (16, 33, 17, 1, 33 * 17), (4, 21, 9, 3, 100),
                                               (8, 64, 64, 1, 1), (16, 40, 40, 2, 3000)])
def test_layout_matches_oracle(tile_size, W, H, I, P):
    from oracle import oracle

    pixels, image_ids = _random_pixels(P, I, W, H, seed=P + tile_size)
    tw, th = -(-W // tile_size), -(-H // tile_size)
    act, tmask, pmask, cum, pmap = _impl()(pixels, image_ids, I, tile_size, tw, th)
    act_o, tmask_o, pmask_o, cum_o, pmap_o = oracle.sparse_tile_layout(pixels, image_ids, I, tile_size, tw, th)
    assert act.dtype == torch.int32 and tmask.dtype == torch.bool and pmask.dtype == torch.uint64
    assert cum.dtype == torch.int64 and pmap.dtype == torch.int64
    assert np.array_equal(act.numpy(), act_o)
    assert np.array_equal(tmask.numpy(), tmask_o)
    assert np.array_equal(pmask.view(torch.int64).numpy().view(np.uint64), pmask_o)
    assert np.array_equal(cum.numpy(), cum_o)
    assert np.array_equal(pmap.numpy(), pmap_o)
    # contract: active_tiles == nonzero(mask), cumsum[-1] == P, pixel_map is a permutation
    assert np.array_equal(act.numpy(), np.nonzero(tmask.numpy().reshape(-1))[0])
    assert int(cum[-1]) == P and sorted(pmap.tolist()) == list(range(P))


def test_layout_full_tile_sets_bit_63():
    # a fully requested 8x8 tile: one word with all 64 bits (bit 63 = the sign bit of the int64 accumulator)
    rows, cols = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    pixels = torch.stack([rows.reshape(-1), cols.reshape(-1)], -1).to(torch.int32)
    act, tmask, pmask, cum, pmap = _impl()(pixels, torch.zeros(64, dtype=torch.int32), 1, 8, 2, 2)
    assert act.tolist() == [0] and int(cum[0]) == 64
    assert pmask.view(torch.int64).numpy().view(np.uint64).tolist() == [[0xFFFFFFFFFFFFFFFF]]
    assert pmap.tolist() == list(range(64))


def test_layout_empty():
    act, tmask, pmask, cum, pmap = _impl()(torch.zeros((0, 2), dtype=torch.int32), torch.zeros(0, dtype=torch.int32), 2,
                                           16, 3, 2)
    assert act.shape == (0,) and tmask.shape == (2, 2, 3) and not tmask.any()
    assert pmask.shape == (0, 4) and cum.tolist() == [0] and pmap.shape == (0,)


# ---- golden vectors: outputs of the REFERENCE's looped torch references (oracle/pin_sparse_against_reference.py) ----
def _gold():
    import os

    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sparse_ref.npz")))


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_oracle_and_layout_builder_match_reference_golden(case):
    from oracle import oracle

    g = _gold()
    C, N, W, H, ts, P = g[f"{case}_dims"].tolist()
    tw, th = -(-W // ts), -(-H // ts)
    pixels, image_ids = torch.from_numpy(g[f"{case}_pixels"]), torch.from_numpy(g[f"{case}_image_ids"])
    want = [g[f"{case}_{k}"] for k in ("active_tiles", "tile_mask", "pixel_mask", "pixel_cumsum", "pixel_map")]
    for got, w in zip(oracle.sparse_tile_layout(pixels, image_ids, C, ts, tw, th), want):
        assert np.array_equal(got, w)
    for got, w in zip(_impl()(pixels, image_ids, C, ts, tw, th), want):
        got = got.view(torch.int64).numpy().view(np.uint64) if got.dtype == torch.uint64 else got.numpy()
        assert np.array_equal(got, w)
    m2, rad, d = (torch.from_numpy(g[f"{case}_{k}"]) for k in ("means2d", "radii", "depths"))
    tmask, act = torch.from_numpy(g[f"{case}_tile_mask"]), torch.from_numpy(g[f"{case}_active_tiles"])
    off, fl = oracle.isect_tiles_sparse(m2, rad, d, tmask, act, C, ts, tw, th)
    assert np.array_equal(off.numpy(), g[f"{case}_tile_offsets"]) and np.array_equal(fl.numpy(), g[f"{case}_flatten_ids"])
    vis = (rad > 0).all(-1)
    ci, _ = torch.where(vis)
    off, fl = oracle.isect_tiles_sparse(m2[vis], rad[vis], d[vis], tmask, act, C, ts, tw, th, image_ids=ci)
    assert np.array_equal(off.numpy(), g[f"{case}_tile_offsets_packed"])
    assert np.array_equal(fl.numpy(), g[f"{case}_flatten_ids_packed"])
