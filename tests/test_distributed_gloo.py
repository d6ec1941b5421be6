"""CPU, world_size 2, gloo: the collective seams of rasterization(distributed=True) (gsplat_amd/distributed.py) —
all-gather of cameras (seam A) and the personalised all-to-all of projected rows (seam B), dense and packed, forward
routing and the autograd-generated reverse exchange. Mirrors what the reference checks for its collectives
(gsplat/distributed.py doctests / tests/test_rasterization.py:819-868) without needing the kernels: rows are synthetic
functions of (global camera, global gaussian) so every element's destination is known."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
N_PER_RANK = [5, 7]
D = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _row(cam, gid, width):
    """Synthetic payload row for (global camera, global gaussian)."""
    base = 1000.0 * cam + gid
    return base + 0.01 * torch.arange(width, dtype=torch.float32)


def _worker(rank, port, results, C_LOCAL):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from gsplat_amd import distributed as gd

        dev = torch.device("cpu")
        n_local = N_PER_RANK[rank]
        offset = sum(N_PER_RANK[:rank])
        total = sum(N_PER_RANK)

        # ---- helper collectives ------------------------------------------------------------------
        assert gd.all_gather_int32(WORLD, 10 + rank, device=dev) == [10, 11]
        assert gd.all_to_all_int32(WORLD, [rank * 10 + j for j in range(WORLD)], device=dev) == [rank, 10 + rank]
        a = torch.full((3, 2), float(rank), requires_grad=True)
        b = torch.full((3,), 5.0 + rank)
        ga, gb = gd.all_gather_tensor_list(WORLD, [a, b])
        assert ga.shape == (6, 2) and gb.shape == (6,) and ga[3:].eq(1).all() and gb[:3].eq(5).all()
        ga.sum().backward()
        assert a.grad.eq(WORLD).all()  # every rank's loss touches every rank's rows
        t = torch.arange(4.0)[:, None] + 10 * rank
        (o,) = gd.all_to_all_tensor_list(WORLD, [t], [1, 3] if rank == 0 else [2, 2])
        exp = torch.tensor([0.0, 10.0, 11.0]) if rank == 0 else torch.tensor([1.0, 2.0, 3.0, 12.0, 13.0])
        assert torch.equal(o[:, 0], exp), (rank, o)

        ctx = gd.DistributedRasterContext.create(batch_dims=(), sparse_grad=False, absgrad=False,
                                                 camera_model="pinhole", colors=torch.zeros(n_local, D), sh_degree=None,
                                                 n_cameras=C_LOCAL, device=dev, n_local=n_local)
        assert ctx.n_per_rank == N_PER_RANK and ctx.total_gaussians == total and ctx.gaussian_offset == offset

        # ---- seam A ------------------------------------------------------------------------------
        vm = torch.eye(4).repeat(C_LOCAL, 1, 1) * (rank + 1)
        Ks = torch.eye(3).repeat(C_LOCAL, 1, 1) * (rank + 1)
        vm.requires_grad_(True)
        V, K = ctx.gather_cameras(vm, Ks)
        assert V.shape == (WORLD * C_LOCAL, 4, 4) and K.shape == (WORLD * C_LOCAL, 3, 3)
        assert V[:C_LOCAL, 0, 0].eq(1).all() and V[C_LOCAL:, 0, 0].eq(2).all()
        (V * (rank + 1)).sum().backward()
        assert vm.grad.eq(1 + 2).all()  # sum over ranks of their weights

        # ---- seam B, dense -----------------------------------------------------------------------
        C_all = WORLD * C_LOCAL
        pay = torch.stack([torch.stack([_row(c, offset + g, 7 + D) for g in range(n_local)]) for c in range(C_all)])
        pay.requires_grad_(True)
        radii = torch.stack([torch.stack([torch.tensor([c, offset + g], dtype=torch.int32) for g in range(n_local)])
                             for c in range(C_all)])
        m2, dp, cn, op, ft = pay[..., 0:2], pay[..., 2], pay[..., 3:6], pay[..., 6], pay[..., 7:]
        r_, m2_, dp_, cn_, op_, ft_, img_, gid_ = ctx.scatter_projection(False, radii, m2, dp, cn, op, ft, None, None, None)
        assert r_.shape == (C_LOCAL, total, 2) and m2_.shape == (C_LOCAL, total, 2) and ft_.shape == (C_LOCAL, total, D)
        got = torch.cat([m2_, dp_[..., None], cn_, op_[..., None], ft_], -1)
        exp = torch.stack([torch.stack([_row(rank * C_LOCAL + c, g, 7 + D) for g in range(total)]) for c in range(C_LOCAL)])
        assert torch.equal(got.detach(), exp)
        assert torch.equal(r_[..., 0], torch.arange(C_LOCAL, dtype=torch.int32)[:, None].expand(-1, total) + rank * C_LOCAL)
        assert torch.equal(r_[..., 1], torch.arange(total, dtype=torch.int32)[None].expand(C_LOCAL, -1))
        # backward = reverse all-to-all: rank d's weight (d+1) lands on the rows that went to d
        (got * (rank + 1)).sum().backward()
        wexp = torch.cat([torch.full((C_LOCAL, n_local, 7 + D), float(d + 1)) for d in range(WORLD)], 0)
        assert torch.equal(pay.grad, wexp)

        # ---- seam B, dense, as two overlapped messages (geometry first, features in flight) ------------
        assert ctx.overlaps(False) and not ctx.overlaps(True)
        pay2 = pay.detach().clone().requires_grad_(True)
        geo = ctx.geometry_payload(radii, pay2[..., 0:2], pay2[..., 2], pay2[..., 3:6], pay2[..., 6])
        feats2 = pay2[..., 7:] * 1.0  # stands for the SH evaluation: created AFTER the geometry payload
        r2, m2b, dpb, cnb, opb, features = ctx.scatter_dense_begin(geo, feats2)
        assert m2b.is_contiguous() and cnb.is_contiguous() and r2.dtype == torch.int32
        got2 = torch.cat([m2b, dpb[..., None], cnb, opb[..., None], features()], -1)
        assert torch.equal(got2.detach(), exp) and torch.equal(r2, r_)
        (got2 * (rank + 1)).sum().backward()
        assert torch.equal(pay2.grad, wexp)
        assert ctx._geo_bwd.work is None  # the reverse geometry exchange was waited for by the _WaitGrad node
        # without feature rows (depth-only render modes)
        geo = ctx.geometry_payload(radii, pay2[..., 0:2], pay2[..., 2], pay2[..., 3:6], pay2[..., 6])
        r3, m3, _d3, _c3, _o3, features = ctx.scatter_dense_begin(geo, None)
        assert features() is None and torch.equal(m3.detach(), exp[..., 0:2]) and torch.equal(r3, r_)

        # ---- seam B, packed ----------------------------------------------------------------------
        g = torch.Generator().manual_seed(100 + rank)
        vis = torch.rand(C_all, n_local, generator=g) > 0.4
        cam_ids, g_ids = torch.where(vis)
        rows = torch.stack([_row(int(c), offset + int(gg), 7 + D) for c, gg in zip(cam_ids, g_ids)])
        rows.requires_grad_(True)
        radii_p = torch.stack([cam_ids.int(), (g_ids + offset).int()], -1)
        out = ctx.scatter_projection(True, radii_p, rows[:, 0:2], rows[:, 2], rows[:, 3:6], rows[:, 6], rows[:, 7:],
                                     torch.zeros_like(cam_ids), cam_ids, g_ids)
        r_, m2_, dp_, cn_, op_, ft_, img_, gid_ = out
        got = torch.cat([m2_, dp_[:, None], cn_, op_[:, None], ft_], -1)
        # expectation: every rank's visible rows whose camera belongs to me, in source-rank order
        exp_rows, exp_img, exp_gid = [], [], []
        for src in range(WORLD):
            gs = torch.Generator().manual_seed(100 + src)
            v = torch.rand(C_all, N_PER_RANK[src], generator=gs) > 0.4
            cc, gg = torch.where(v)
            for c, k in zip(cc.tolist(), gg.tolist()):
                if c // C_LOCAL == rank:
                    gidg = sum(N_PER_RANK[:src]) + k
                    exp_rows.append(_row(c, gidg, 7 + D)); exp_img.append(c - rank * C_LOCAL); exp_gid.append(gidg)
        assert torch.equal(got.detach(), torch.stack(exp_rows))
        assert img_.tolist() == exp_img and gid_.tolist() == exp_gid
        assert torch.equal(r_[:, 1].long(), gid_) and torch.equal(r_[:, 0].long() - rank * C_LOCAL, img_)
        (got * (rank + 1)).sum().backward()
        dest = torch.div(cam_ids, C_LOCAL, rounding_mode="floor")
        assert torch.equal(rows.grad, (dest + 1).float()[:, None].expand(-1, 7 + D))

        # ---- validation errors (Rendering.cpp:190-233) ---------------------------------------------
        for kw in (dict(batch_dims=(1,)), dict(sparse_grad=True), dict(absgrad=True), dict(camera_model="fisheye")):
            args = dict(batch_dims=(), sparse_grad=False, absgrad=False, camera_model="pinhole", colors=None,
                        sh_degree=None, n_cameras=C_LOCAL, device=dev, n_local=n_local)
            args.update(kw)
            with pytest.raises(RuntimeError, match="distributed=True"):
                gd.DistributedRasterContext.create(**args)
        results[rank] = "ok"
    except Exception as e:  # surface the failure to the parent
        import traceback

        results[rank] = traceback.format_exc()
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("c_local", [2, 1])  # 1 = the bench configuration (received buffer used without a concat)
def test_distributed_seams_world_size_2_gloo(c_local):
    mgr = mp.Manager()
    results = mgr.dict()
    port = _free_port()
    ctx = mp.spawn(_worker, args=(port, results, c_local), nprocs=WORLD, join=False)
    ok = ctx.join(timeout=240)
    while not ok:
        ok = ctx.join(timeout=240)
    assert dict(results) == {0: "ok", 1: "ok"}, dict(results)


def test_unequal_camera_counts_are_rejected():
    """World size 1 sanity of the validation path that needs no peers."""
    from gsplat_amd import distributed as gd

    with pytest.raises(ValueError, match="initialized default torch.distributed process group"):
        gd.DistributedRasterContext.create(batch_dims=(), sparse_grad=False, absgrad=False, camera_model="pinhole",
                                           colors=None, sh_degree=None, n_cameras=1, device=torch.device("cpu"),
                                           n_local=4)  # no process group initialised
