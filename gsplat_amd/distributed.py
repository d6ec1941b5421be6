"""Gaussian-sharded multi-GPU rasterization over RCCL (torch.distributed backend "nccl" on ROCm).

One process per GPU. Each rank holds a disjoint shard of the Gaussians and an equal number of
cameras. The data path restates the reference's scheme (``gsplat/cuda/csrc/
DistributedCollectives.cpp:57-86, 299-453``; python helpers ``gsplat/distributed.py:25-375``):

    seam A  all-gather cameras        -> every rank projects its shard against ALL cameras
    seam B  all-to-all (personalised) -> each projected (camera, gaussian) row travels to the rank
                                          that owns the camera; that rank composites its own
                                          cameras over all Gaussians
    backward: autograd replays seam B as the reverse all-to-all of the gradients.

(A sum all-reduce of partial images is NOT equivalent: front-to-back compositing needs one depth
order across all shards.) On MI355X the personalised exchange maps onto the 7 point-to-point xGMI
links of every GPU concurrently, so there is no ring bottleneck; payload per row is
8 B (radii) + 4*(7+D) B (means2d, depth, conic, opacity, D features) [+16 B ids when packed].

The collective seams are plain ``torch.distributed`` code and run on any backend (``gloo`` in the
CPU tests); the kernels around them require a ROCm device.
"""
from __future__ import annotations

import os
import socket
from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Tuple, Union

import torch
import torch.distributed as dist
import torch.distributed.nn.functional as distF
from torch import Tensor


# ----------------------------------------------------------------------------------------------
# helper collectives with the reference's names and semantics (gsplat/distributed.py:25-272)
# ----------------------------------------------------------------------------------------------
def all_gather_int32(world_size: int, value: Union[int, Tensor], device: Optional[torch.device] = None) -> List:
    """Gather one 32-bit integer from every rank (not differentiable)."""
    if world_size == 1:
        return [value]
    if isinstance(value, int):
        assert device is not None, "device is required for scalar input"
        value_tensor = torch.tensor(value, dtype=torch.int, device=device)
    else:
        value_tensor = value
    collected = torch.empty(world_size, dtype=value_tensor.dtype, device=value_tensor.device)
    dist.all_gather_into_tensor(collected, value_tensor.reshape(1))
    return collected.tolist() if isinstance(value, int) else list(collected.unbind())


def all_to_all_int32(world_size: int, values: List[Union[int, Tensor]], device: Optional[torch.device] = None) -> List:
    """Many-to-many exchange of one 32-bit integer per peer (not differentiable)."""
    if world_size == 1:
        return values
    assert len(values) == world_size
    if any(isinstance(v, int) for v in values):
        assert device is not None, "device is required for scalar input"
        send = torch.tensor([int(v) for v in values], dtype=torch.int, device=device)
        scalar = True
    else:
        send = torch.stack([v.reshape(()) for v in values]).to(torch.int)
        scalar = False
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    return recv.tolist() if scalar else list(recv.unbind())


def all_gather_tensor_list(world_size: int, tensor_list: List[Tensor]) -> List[Tensor]:
    """Differentiable all-gather of a list of tensors whose first dim is equal on all ranks; the
    tensors are packed into one message. Returns the list with first dim multiplied by world_size."""
    if world_size == 1:
        return tensor_list
    N = len(tensor_list[0])
    for t in tensor_list:
        assert len(t) == N, "All tensors should have the same first dimension size"
    data = torch.cat([t.reshape(N, -1) for t in tensor_list], dim=-1)
    sizes = [t.numel() // N for t in tensor_list]
    collected = torch.cat(distF.all_gather(data), dim=0)  # [W*N, sum sizes]
    outs = []
    for piece, t in zip(collected.split(sizes, dim=-1), tensor_list):
        outs.append(piece.reshape((-1,) + tuple(t.shape[1:])))
    return outs


def all_to_all_tensor_list(world_size: int, tensor_list: List[Tensor], splits: List[Union[int, Tensor]],
                           output_splits: Optional[List[Union[int, Tensor]]] = None) -> List[Tensor]:
    """Differentiable personalised exchange of a list of tensors split along the first dim."""
    if world_size == 1:
        return tensor_list
    N = len(tensor_list[0])
    for t in tensor_list:
        assert len(t) == N, "All tensors should have the same first dimension size"
    assert len(splits) == world_size
    data = torch.cat([t.reshape(N, -1) for t in tensor_list], dim=-1)
    sizes = [t.numel() // N if N > 0 else int(torch.tensor(t.shape[1:]).prod()) for t in tensor_list]
    if output_splits is None:
        output_splits = all_to_all_int32(world_size, splits, device=data.device)
    in_s = [int(s) for s in splits]
    out_s = [int(s) for s in output_splits]
    recv = _all_to_all_rows(data, in_s, out_s)
    outs = []
    for piece, t in zip(recv.split(sizes, dim=-1), tensor_list):
        outs.append(piece.reshape((-1,) + tuple(t.shape[1:])))
    return outs


def _all_to_all_rows(data: Tensor, in_splits: List[int], out_splits: List[int]) -> Tensor:
    """Row-wise all-to-all: autograd-aware for floating tensors that require grad."""
    out = torch.empty((sum(out_splits),) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device)
    if data.is_floating_point() and data.requires_grad:
        return distF.all_to_all_single(out, data.contiguous(), out_splits, in_splits)
    dist.all_to_all_single(out, data.contiguous(), out_splits, in_splits)
    return out


class _ExchangeRows(torch.autograd.Function):
    """Seam B as ONE collective each way: the float payload and the int32 radii (bit-cast to float32 columns) cross the
    all-to-all in a single message, so a step pays one collective launch / rendezvous in forward instead of two; the
    backward sends only the payload gradient (the radii carry none). Same result as two ``all_to_all_single`` calls
    (reference DistributedCollectives.cpp:368-453 sends radii, geometry and ids separately)."""

    @staticmethod
    def forward(ctx, payload: Tensor, radii: Tensor, in_splits: List[int], out_splits: List[int]):
        F_ = payload.shape[1]
        buf = torch.cat([payload, radii.contiguous().view(torch.float32)], dim=1)  # [rows, F + 2]
        out = torch.empty((sum(out_splits), F_ + 2), dtype=payload.dtype, device=payload.device)
        dist.all_to_all_single(out, buf, out_splits, in_splits)
        ctx.splits = (in_splits, out_splits)
        recv_r = out[:, F_:].contiguous().view(torch.int32)
        ctx.mark_non_differentiable(recv_r)
        return out[:, :F_], recv_r

    @staticmethod
    def backward(ctx, v_payload: Tensor, _v_radii):
        in_splits, out_splits = ctx.splits
        if v_payload is None:
            return None, None, None, None
        v_payload = v_payload.contiguous()
        back = torch.empty((sum(in_splits), v_payload.shape[1]), dtype=v_payload.dtype, device=v_payload.device)
        dist.all_to_all_single(back, v_payload, in_splits, out_splits)
        return back, None, None, None


# ----------------------------------------------------------------------------------------------
# control plane: per-call exchange of (Gaussians, cameras) per rank
# ----------------------------------------------------------------------------------------------
_meta_group_cache = {}


def _meta_group():
    """A CPU (gloo) process group for the two integers every rank publishes per call. The reference gathers them on
    the device and reads them back (DistributedCollectives.cpp:299-317), which makes the host wait for ALL queued GPU
    work at the top of every rasterization() call; over gloo the host never touches the stream, so the previous
    step's backward keeps the GPU busy while the next forward is being enqueued. Created collectively on first use
    (every rank enters rasterization(distributed=True) together); None if gloo is unavailable."""
    key = id(dist.group.WORLD)
    if key not in _meta_group_cache:
        grp = None
        if dist.get_backend() == "gloo":
            grp = dist.group.WORLD
        else:
            # single-node rendezvous over loopback (the launch contract of bench.py / cli()): pin gloo to `lo` instead of
            # letting it resolve the container's hostname, which need not resolve
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                grp = dist.new_group(backend="gloo")
            except Exception:  # no usable network interface for gloo: fall back to the device collective
                grp = None
        _meta_group_cache[key] = grp
    return _meta_group_cache[key]


def _gather_counts(W: int, n_local: int, n_cameras: int, device) -> List[List[int]]:
    """[[N_i, C_i] for every rank i]."""
    if W == 1:
        return [[int(n_local), int(n_cameras)]]
    grp = _meta_group()
    if grp is not None:
        mine = torch.tensor([n_local, n_cameras], dtype=torch.int64)
        out = [torch.empty(2, dtype=torch.int64) for _ in range(W)]
        dist.all_gather(out, mine, group=grp)
        return [t.tolist() for t in out]
    counts = torch.tensor([n_local, n_cameras], dtype=torch.int32, device=device)
    gathered = torch.empty((W, 2), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(gathered, counts.reshape(1, 2))
    return gathered.tolist()


# ----------------------------------------------------------------------------------------------
# the two seams of rasterization(distributed=True)
# ----------------------------------------------------------------------------------------------
@dataclass
class DistributedRasterContext:
    world_size: int
    rank: int
    n_local: int                 # Gaussians held by this rank
    n_per_rank: List[int]        # Gaussians of every rank
    c_local: int                 # cameras per rank (equal on all ranks)

    @property
    def total_gaussians(self) -> int:
        return sum(self.n_per_rank)

    @property
    def gaussian_offset(self) -> int:
        return sum(self.n_per_rank[: self.rank])

    @staticmethod
    def create(batch_dims, sparse_grad, absgrad, camera_model, colors, sh_degree, n_cameras, device,
               n_local: Optional[int] = None) -> "DistributedRasterContext":
        """Validation follows Rendering.cpp:190-233 (what distributed mode rejects)."""
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("distributed=True requires an initialized torch.distributed process group")
        if device.type == "cuda" and dist.get_backend() != "nccl":
            raise RuntimeError("distributed rasterization on GPUs requires the 'nccl' backend (RCCL on ROCm)")
        if len(batch_dims) != 0:
            raise ValueError("distributed=True does not support batch dimensions")
        if sparse_grad:
            raise ValueError("distributed=True does not support sparse_grad")
        if absgrad:
            raise ValueError("distributed=True does not support absgrad")
        if camera_model != "pinhole":
            raise ValueError("distributed=True only supports the pinhole camera model")
        if colors is not None and sh_degree is None and colors.dim() == 3:
            raise ValueError("distributed=True does not support per-view colors [C, N, D]")
        W, r = dist.get_world_size(), dist.get_rank()
        if n_local is None:
            raise ValueError("n_local is required")
        gathered = _gather_counts(W, n_local, n_cameras, device)
        cams = [g[1] for g in gathered]
        if any(c != cams[0] for c in cams):
            raise ValueError(f"distributed=True requires the same number of cameras on every rank, got {cams}")
        return DistributedRasterContext(W, r, n_local, [g[0] for g in gathered], n_cameras)

    # -- seam A ------------------------------------------------------------------------------
    def gather_cameras(self, viewmats: Tensor, Ks: Tensor) -> Tuple[Tensor, Tensor]:
        """[C_local,4,4],[C_local,3,3] -> [W*C_local,4,4],[W*C_local,3,3] (differentiable)."""
        if self.world_size == 1:
            return viewmats, Ks
        v, k = all_gather_tensor_list(self.world_size, [viewmats.contiguous(), Ks.contiguous()])
        return v, k

    # -- seam B ------------------------------------------------------------------------------
    def scatter_projection(self, packed: bool, radii, means2d, depths, conics, opacities, feats, batch_ids,
                           camera_ids, gaussian_ids):
        """Move every projected row to the rank owning its camera.

        dense in : [W*C_local, N_local, *]  -> out [C_local, sum_i N_i, *]
        packed in: [nnz, *] sorted by camera -> out [nnz', *] rows of this rank's cameras, with
                   image_ids = camera id local to this rank and gaussian_ids made global.
        Returns (radii, means2d, depths, conics, opacities, feats, image_ids, gaussian_ids)."""
        W, Cl = self.world_size, self.c_local
        if W == 1:  # a single rank owns every camera and every Gaussian: nothing moves
            if not packed:
                return radii, means2d, depths, conics, opacities.contiguous(), feats, None, None
            return radii, means2d, depths, conics, opacities, feats, camera_ids, gaussian_ids
        floats = [means2d, depths[..., None], conics, opacities[..., None]]
        if feats is not None:
            floats.append(feats)
        widths = [t.shape[-1] for t in floats]
        payload = torch.cat(floats, dim=-1)  # [..., 7 + D]
        if not packed:
            Nl = self.n_local
            in_s = [Cl * Nl] * W
            out_s = [Cl * n for n in self.n_per_rank]
            recv_f, recv_r = _ExchangeRows.apply(payload.reshape(W * Cl * Nl, -1), radii.reshape(W * Cl * Nl, 2), in_s,
                                                 out_s)
            # source rank i contributed [C_local, N_i, F]; concatenate along the Gaussian axis (with one camera per
            # rank the received buffer already IS [1, sum N_i, F])
            if Cl == 1:
                out_f, out_r = recv_f.reshape(1, -1, recv_f.shape[-1]), recv_r.reshape(1, -1, 2)
            else:
                out_f = torch.cat([p.reshape(Cl, n, -1) for p, n in zip(recv_f.split(out_s), self.n_per_rank)], dim=1)
                out_r = torch.cat([p.reshape(Cl, n, 2) for p, n in zip(recv_r.split(out_s), self.n_per_rank)], dim=1)
            pieces = out_f.split(widths, dim=-1)
            m2, dp, cn, op = pieces[0], pieces[1][..., 0], pieces[2], pieces[3][..., 0]
            ft = pieces[4] if feats is not None else None
            return out_r.contiguous(), m2.contiguous(), dp.contiguous(), cn.contiguous(), op.contiguous(), ft, None, None
        # packed: rows are sorted by (camera, gaussian); destination = camera // C_local
        dest = torch.div(camera_ids, Cl, rounding_mode="floor")
        in_s = torch.bincount(dest, minlength=W).tolist()
        ids = torch.stack([camera_ids - dest * Cl, gaussian_ids + self.gaussian_offset], dim=-1)  # int64 [nnz,2]
        out_s = all_to_all_int32(W, in_s, device=payload.device)
        recv_f, recv_r = _ExchangeRows.apply(payload, radii, in_s, out_s)
        recv_i = _all_to_all_rows(ids, in_s, out_s)
        pieces = recv_f.split(widths, dim=-1)
        m2, dp, cn, op = pieces[0], pieces[1][..., 0], pieces[2], pieces[3][..., 0]
        ft = pieces[4] if feats is not None else None
        return (recv_r.contiguous(), m2.contiguous(), dp.contiguous(), cn.contiguous(), op.contiguous(), ft,
                recv_i[:, 0].contiguous(), recv_i[:, 1].contiguous())


# ----------------------------------------------------------------------------------------------
# process launcher (gsplat/distributed.py:275-375)
# ----------------------------------------------------------------------------------------------
def _find_free_port() -> int:
    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def _distributed_worker(world_rank: int, world_size: int, fn: Callable, args: Any, local_rank: Optional[int] = None,
                        verbose: bool = False, backend: str = "nccl") -> bool:
    if local_rank is None:
        local_rank = world_rank
    if verbose:
        print("Distributed worker: %d / %d" % (world_rank + 1, world_size))
    distributed = world_size > 1
    if distributed:
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, world_size=world_size, rank=world_rank)
        _ = [None for _ in range(world_size)]
        dist.all_gather_object(_, 0)  # initialise the communicator on every rank
    fn(local_rank, world_rank, world_size, args)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if verbose:
        print("Job Done for worker: %d / %d" % (world_rank + 1, world_size))
    return True


def cli(fn: Callable, args: Any, verbose: bool = False) -> bool:
    """Run ``fn(local_rank, world_rank, world_size, args)`` once per visible GPU (one process per GPU,
    RCCL), or once in-process when a single GPU is visible. OpenMPI launches are honoured through
    ``OMPI_COMM_WORLD_*`` like the reference."""
    assert torch.cuda.is_available(), "a ROCm device is required!"
    if "OMPI_COMM_WORLD_SIZE" in os.environ:
        local_rank = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        world_size = int(os.environ["OMPI_COMM_WORLD_SIZE"])
        world_rank = int(os.environ["OMPI_COMM_WORLD_RANK"])
        return _distributed_worker(world_rank, world_size, fn, args, local_rank, verbose)
    world_size = torch.cuda.device_count()
    if world_size > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_find_free_port())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ctx = torch.multiprocessing.spawn(_distributed_worker, args=(world_size, fn, args, None, verbose),
                                          nprocs=world_size, join=False)
        try:
            ctx.join()
        except KeyboardInterrupt:
            for process in ctx.processes:
                if process.is_alive():
                    process.terminate()
                process.join()
        return True
    return _distributed_worker(0, 1, fn=fn, args=args)
