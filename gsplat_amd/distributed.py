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


# Bytes this rank hands to OTHER ranks in row exchanges since the last reset (forward messages and the reverse exchange of the
# gradients; the rows a rank keeps for its own cameras do not cross a link). bench.py reports it per step next to the N > 1 line
# so that a scaling run can be checked against SURVEY.md section 8(e)'s volume (c4 dense, 8 ranks: 0.67 GB per direction).
EXCHANGE_STATS = {"bytes_to_peers": 0, "collectives": 0}


def reset_exchange_stats() -> None:
    EXCHANGE_STATS["bytes_to_peers"] = 0
    EXCHANGE_STATS["collectives"] = 0


def _count_exchange(buf: Tensor, send_splits: List[int]) -> None:
    rows_out = sum(int(x) for x in send_splits)
    if dist.is_initialized():
        r = dist.get_rank()
        if r < len(send_splits):
            rows_out -= int(send_splits[r])
    row_bytes = buf.element_size() * (buf.numel() // max(buf.shape[0], 1)) if buf.dim() else buf.element_size()
    EXCHANGE_STATS["bytes_to_peers"] += rows_out * row_bytes
    EXCHANGE_STATS["collectives"] += 1


def _all_to_all_rows(data: Tensor, in_splits: List[int], out_splits: List[int]) -> Tensor:
    """Row-wise all-to-all: autograd-aware for floating tensors that require grad."""
    _count_exchange(data, in_splits)
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
        _count_exchange(buf, in_splits)
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
        _count_exchange(v_payload, out_splits)
        dist.all_to_all_single(back, v_payload, in_splits, out_splits)
        return back, None, None, None


class _Pending:
    """Handle of an exchange launched with async_op=True. wait() orders the CURRENT stream after the collective (RCCL:
    a stream-side wait, the host does not block; gloo: the host blocks) and is idempotent."""
    __slots__ = ("work",)

    def __init__(self):
        self.work = None

    def wait(self) -> None:
        if self.work is not None:
            self.work.wait()
            self.work = None


class _AsyncExchange(torch.autograd.Function):
    """A personalised row exchange that is LAUNCHED here and waited for by the caller (``fwd.wait()`` before the
    returned views are read), so that independent kernels run while the rows cross xGMI. Optional int32 ``radii`` ride
    along as two bit-cast columns. The backward launches the reverse exchange of the payload gradient; with
    ``bwd`` given it returns without waiting and the ``_WaitGrad`` node upstream waits (the autograd engine runs nodes
    created later in the forward first, so everything created between the two — the SH backward — overlaps it)."""

    @staticmethod
    def forward(ctx, payload: Tensor, radii: Optional[Tensor], in_splits: List[int], out_splits: List[int],
                fwd: _Pending, bwd: Optional[_Pending], message: Optional[Tensor] = None):
        F_ = payload.shape[1]
        if message is not None:  # payload is message[:, :F_] and the radii already sit behind it (_PackGeometry)
            buf = message
        else:
            buf = payload.contiguous() if radii is None else torch.cat([payload, radii.contiguous().view(torch.float32)], dim=1)
        out = torch.empty((sum(out_splits), buf.shape[1]), dtype=payload.dtype, device=payload.device)
        # The collective writes `out` AFTER the views below exist. A backend that bumps the tensor's version counter when
        # the work completes (gloo does; RCCL bumps it at launch) would make autograd reject every later view of those
        # views ("its base has been modified inplace"). `.data` aliases the storage with a version counter of its own.
        _count_exchange(buf, in_splits)
        fwd.work = dist.all_to_all_single(out.data, buf, out_splits, in_splits, async_op=True)
        ctx.splits, ctx.bwd = (in_splits, out_splits), bwd
        recv_r = out[:, F_:]  # float32 VIEW of the radii bits (no data is touched before the caller waits)
        ctx.mark_non_differentiable(recv_r)
        return out[:, :F_], recv_r

    @staticmethod
    def backward(ctx, v_payload: Tensor, _v_radii):
        if v_payload is None:
            return None, None, None, None, None, None, None
        in_splits, out_splits = ctx.splits
        v_payload = v_payload.contiguous()
        back = torch.empty((sum(in_splits), v_payload.shape[1]), dtype=v_payload.dtype, device=v_payload.device)
        _count_exchange(v_payload, out_splits)
        work = dist.all_to_all_single(back, v_payload, in_splits, out_splits, async_op=True)
        if ctx.bwd is None:
            work.wait()
        else:
            ctx.bwd.work = work
        return back, None, None, None, None, None, None


class _WaitGrad(torch.autograd.Function):
    """Identity whose backward first waits for a pending reverse exchange (see _AsyncExchange)."""

    @staticmethod
    def forward(ctx, x: Tensor, pending: _Pending):
        ctx.pending = pending
        return x.view_as(x)

    @staticmethod
    def backward(ctx, v):
        ctx.pending.wait()
        return v, None


def _row_source(t: Tensor, width: int):
    """(tensor to keep alive, data pointer, row stride in words) of a [rows, width] 4-byte tensor; column views with a
    uniform row stride (e.g. slices of the AoS gradient rows) are read in place, anything else through a contiguous copy."""
    if not (t.dim() == 2 and t.shape[1] == width and (width == 1 or t.stride(1) == 1) and t.stride(0) >= width):
        t = t.contiguous()
    return t, t.data_ptr(), t.stride(0)


class _RowMap:
    """Rows of a rank that owns several cameras: the exchange delivers them source rank by source rank, each source's block
    camera-major ([C_local][N_k] rows from source k); everything downstream wants [C_local][sum N_k]. `index` (built
    lazily, CPU path only) maps an exchange-order row to its [C_local][sum N] row."""

    def __init__(self, c_local: int, n_per_rank):
        self.c_local, self.seg_n = int(c_local), [int(n) for n in n_per_rank]
        self.seg_rows = [self.c_local * n for n in self.seg_n]
        self.total_n, self.rows = sum(self.seg_n), self.c_local * sum(self.seg_n)
        self._index = None

    def index(self, device) -> Tensor:
        if self._index is None:
            parts, off = [], 0
            for n in self.seg_n:
                c = torch.arange(self.c_local).repeat_interleave(n)
                parts.append(c * self.total_n + off + torch.arange(n).repeat(self.c_local))
                off += n
            self._index = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)
        return self._index.to(device)


_MAX_KERNEL_SEGMENTS = 64  # csrc/rows.hip kMaxSegments: source ranks the mapped copy kernels take in their argument block


def _copy_groups(srcs, dsts, rows: int, row_map: Optional[_RowMap] = None, map_dst: bool = True) -> None:
    """srcs / dsts: lists of [rows, w] 4-byte tensors (any uniform row stride). HIP kernel on the GPU (csrc/rows.hip); on
    CPU tensors — the gloo tests of the seams — plain torch copies. With `row_map`, one side is in exchange order and the
    other in [C_local][sum N] order (map_dst: the destination is the latter)."""
    if rows == 0:
        return
    if srcs[0].is_cuda and not (row_map is not None and len(row_map.seg_n) > _MAX_KERNEL_SEGMENTS):
        from . import _cabi

        keep, groups = [], []
        for s_, d_ in zip(srcs, dsts):
            w = d_.shape[1]
            s_, sp, ss = _row_source(s_, w)
            keep.append(s_)
            groups.append((sp, ss, d_.data_ptr(), d_.stride(0), w))
        if row_map is None:
            _cabi.copy_column_groups(groups, rows)
        else:
            _cabi.copy_column_groups_mapped(groups, row_map.seg_rows, row_map.seg_n, map_dst)
    elif row_map is None:
        for s_, d_ in zip(srcs, dsts):
            d_.copy_(s_)
    else:
        idx = row_map.index(srcs[0].device)
        for s_, d_ in zip(srcs, dsts):
            if map_dst:
                d_[idx] = s_
            else:
                d_.copy_(s_[idx])


def _copy_message(msg: Tensor, cols, fields, to_msg: bool, row_map: Optional[_RowMap] = None) -> None:
    """`msg`: [R, S] 4-byte rows at a uniform row stride (a whole message, or the leading columns of one); group k = its
    columns [cols[k], cols[k] + w_k) <-> fields[k] ([R, w_k], any uniform row stride). On the GPU one LDS-staged kernel
    with both sides coalesced (csrc/rows.hip: gsx_copy_message_columns) when the message rows are whole and short enough,
    else the per-word kernel; CPU tensors: torch copies. The row map applies to the field side."""
    R = msg.shape[0]
    if R == 0:
        return
    views = [msg[:, c:c + f.shape[1]] for c, f in zip(cols, fields)]
    stride = msg.stride(0)
    whole = msg.is_cuda and msg.stride(1) == 1 and stride <= 16 and (not to_msg or sum(f.shape[1] for f in fields) == stride)
    if row_map is not None and len(row_map.seg_n) > _MAX_KERNEL_SEGMENTS:
        whole = False  # the mapped kernels take the segment table by value (64 source ranks); beyond: index-based copies
    if not whole:
        if to_msg:
            _copy_groups(fields, views, R, row_map, map_dst=False)
        else:
            _copy_groups(views, fields, R, row_map, map_dst=True)
        return
    from . import _cabi

    keep, groups = [], []
    for c, f in zip(cols, fields):
        w = f.shape[1]
        f, fp, fs = _row_source(f, w) if to_msg else (f, f.data_ptr(), f.stride(0))
        keep.append(f)
        groups.append((c, w, fp, fs))
    _cabi.copy_message_columns(msg.data_ptr(), stride, R, groups, to_msg,
                               None if row_map is None else row_map.seg_rows, None if row_map is None else row_map.seg_n)


class _PackGeometry(torch.autograd.Function):
    """(means2d [R,2], depths [R], conics [R,3], opacities [R], radii int32 [R,2]) -> the seam-B message [R, 9] in ONE
    kernel, returned as (message, its differentiable payload view message[:, :7]). Backward: one kernel back."""

    @staticmethod
    def forward(ctx, means2d, depths, conics, opacities, radii):
        R = means2d.shape[0]
        msg = torch.empty((R, 9), dtype=means2d.dtype, device=means2d.device)
        srcs = [means2d, depths.reshape(R, 1), conics, opacities.reshape(R, 1), radii.view(torch.float32)]
        _copy_message(msg, [0, 2, 3, 6, 7], srcs, to_msg=True)
        ctx.mark_non_differentiable(msg)
        ctx.shapes = (depths.shape, opacities.shape)
        return msg, msg[:, :7]

    @staticmethod
    def backward(ctx, _v_msg, v):
        if v is None:
            return None, None, None, None, None
        # v_means2d / v_conics: no copy - the projection backward reads them through a row stride (they usually are column
        # views of the compositing kernel's gradient rows; here they are column views of the returned gradient message).
        # The two one-column fields are consumed by torch ops (a sum over the cameras, a contiguous() in the projection
        # op) that would each re-read the whole 28-byte rows: one kernel makes both contiguous.
        R = v.shape[0]
        v_dp, v_op = v.new_empty((R, 1)), v.new_empty((R, 1))
        _copy_message(v, [2, 6], [v_dp, v_op], to_msg=False)
        return v[:, 0:2], v_dp.reshape(ctx.shapes[0]), v[:, 3:6], v_op.reshape(ctx.shapes[1]), None


class _UnpackGeometry(torch.autograd.Function):
    """Received message views (payload [R,7], radii bits [R,2], both with the message's row stride) -> contiguous
    means2d, depths, conics, opacities, radii in ONE kernel; backward packs the four gradients (column views of the
    compositing backward's gradient rows are read in place) into the [R,7] buffer the reverse exchange sends."""

    @staticmethod
    def forward(ctx, payload, radii_bits, row_map: Optional[_RowMap] = None):
        R, dev = payload.shape[0], payload.device
        m2, dp = payload.new_empty((R, 2)), payload.new_empty((R, 1))
        cn, op = payload.new_empty((R, 3)), payload.new_empty((R, 1))
        rad = torch.empty((R, 2), dtype=torch.int32, device=dev)
        if radii_bits.data_ptr() == payload.data_ptr() + 28 and radii_bits.stride(0) == payload.stride(0) == 9:
            # payload and radii bits are columns 0..6 and 7..8 of ONE received message [R, 9]
            whole = torch.as_strided(payload, (R, 9), (9, 1))
            _copy_message(whole, [0, 2, 3, 6, 7], [m2, dp, cn, op, rad.view(torch.float32)], to_msg=False, row_map=row_map)
        else:
            _copy_groups([payload[:, 0:2], payload[:, 2:3], payload[:, 3:6], payload[:, 6:7], radii_bits],
                         [m2, dp, cn, op, rad.view(torch.float32)], R, row_map, map_dst=True)
        ctx.mark_non_differentiable(rad)
        ctx.like = (R, payload.dtype, dev, row_map)
        return m2, dp.reshape(R), cn, op.reshape(R), rad

    @staticmethod
    def backward(ctx, v_m2, v_dp, v_cn, v_op, _v_rad):
        R, dt, dev, row_map = ctx.like
        parts = [(v_m2, 0, 2), (v_dp, 2, 1), (v_cn, 3, 3), (v_op, 6, 1)]
        if all(p[0] is None for p in parts):
            return None, None, None
        full = all(p[0] is not None for p in parts)
        v = (torch.empty if full else torch.zeros)((R, 7), dtype=dt, device=dev)
        srcs = [p[0].reshape(R, p[2]) for p in parts if p[0] is not None]
        if full:
            _copy_message(v, [0, 2, 3, 6], srcs, to_msg=True, row_map=row_map)
        else:
            dsts = [v[:, p[1]:p[1] + p[2]] for p in parts if p[0] is not None]
            _copy_groups(srcs, dsts, R, row_map, map_dst=False)
        return v, None, None


class _UnpackRows(torch.autograd.Function):
    """Received feature rows [R, D] in exchange order -> [R, D] in [C_local][sum N] order (one kernel each way)."""

    @staticmethod
    def forward(ctx, rows, row_map: _RowMap):
        out = torch.empty((rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
        _copy_message(rows, [0], [out], to_msg=False, row_map=row_map)
        ctx.row_map = row_map
        return out

    @staticmethod
    def backward(ctx, v):
        if v is None:
            return None, None
        back = torch.empty((v.shape[0], v.shape[1]), dtype=v.dtype, device=v.device)
        _copy_message(back, [0], [v], to_msg=True, row_map=ctx.row_map)
        return back, None


def _force_exchange() -> bool:
    """GSPLAT_AMD_FORCE_EXCHANGE=1: run the exchanges even in a world of one rank (single-GPU smoke test of seam B)."""
    return os.environ.get("GSPLAT_AMD_FORCE_EXCHANGE", "") not in ("", "0")


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
        """Process-group checks as ``gsplat/rendering.py:178-198`` (ValueError); mode checks as the reference's C++
        validation (``Rendering.cpp:187-233``: RuntimeError, same messages). rasterization() has already run the full
        validation (``rendering._validate_rasterization_inputs``); the mode checks are repeated here for direct users."""
        if not dist.is_available():
            raise ValueError("distributed=True requires torch.distributed to be available.")
        if not dist.is_initialized():
            raise ValueError("distributed=True requires an initialized default torch.distributed process group.")
        # GSPLAT_AMD_ALLOW_NON_NCCL=1: test hook - two ranks on ONE GPU cannot form an RCCL group, gloo moves the same
        # messages through the host (tests/test_gpu_distributed_multirank.py)
        if device.type == "cuda" and dist.get_backend() != "nccl" and os.environ.get("GSPLAT_AMD_ALLOW_NON_NCCL") != "1":
            raise ValueError("distributed=True currently supports only the default NCCL process group "
                             f"(RCCL on ROCm); got backend '{dist.get_backend()}'.")
        if len(batch_dims) != 0:
            raise RuntimeError("distributed=True does not support batch dimensions")
        if sparse_grad:
            raise RuntimeError("distributed=True does not support sparse_grad=True")
        if absgrad:
            raise RuntimeError("distributed=True does not support absgrad=True")
        if camera_model != "pinhole":
            raise RuntimeError("distributed=True only supports camera_model='pinhole'")
        if colors is not None and sh_degree is None and colors.dim() == 3:
            raise RuntimeError("distributed=True only supports per-Gaussian colors")
        W, r = dist.get_world_size(), dist.get_rank()
        if n_local is None:
            raise ValueError("n_local is required")
        gathered = _gather_counts(W, n_local, n_cameras, device)
        cams = [g[1] for g in gathered]
        if any(c != cams[0] for c in cams):
            raise RuntimeError(f"distributed=True requires the same number of cameras on every rank, got {cams}")
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
        if W == 1 and not _force_exchange():  # a single rank owns every camera and every Gaussian: nothing moves
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


    # -- seam B, dense rows, overlapped with compute ------------------------------------------------
    def overlaps(self, packed: bool) -> bool:
        """Dense rows on more than one rank use the two-message exchange below."""
        return (not packed) and (self.world_size > 1 or _force_exchange())

    def geometry_payload(self, radii, means2d, depths, conics, opacities):
        """The geometry message of seam B, [C * N, 9] = means2d | depth | conic | opacity | radii bits, packed by one
        kernel. Call it right after the projection and BEFORE the colours are computed: its autograd nodes carry the wait
        for the reverse geometry exchange, and nodes created after them (the SH evaluation) run before them in the
        backward pass. Returns an opaque handle for scatter_dense_begin."""
        self._geo_bwd = _Pending()
        R = radii.numel() // 2
        msg, payload = _PackGeometry.apply(means2d.reshape(R, 2), depths.reshape(R), conics.reshape(R, 3),
                                           opacities.reshape(R), radii.reshape(R, 2))
        return msg, _WaitGrad.apply(payload, self._geo_bwd)

    def scatter_dense_begin(self, geometry, feats):
        """Launch seam B as TWO messages — geometry (+ radii), then feature rows — and wait only for the first:
        tile intersection needs nothing but geometry, so the feature rows travel while it runs; in the backward pass the
        feature gradients come back first and the SH backward overlaps the return of the geometry gradients.
        Returns (radii, means2d, depths, conics, opacities, features) where ``features()`` waits for and returns the
        received [C_local, sum N_i, D] rows (None without features). Same values as scatter_projection(False, ...)."""
        W, Cl, Nl = self.world_size, self.c_local, self.n_local
        in_s, out_s = [Cl * Nl] * W, [Cl * n for n in self.n_per_rank]
        geo_fwd, col_fwd = _Pending(), _Pending()
        msg, payload = geometry
        recv_g, recv_r = _AsyncExchange.apply(payload, None, in_s, out_s, geo_fwd, getattr(self, "_geo_bwd", None), msg)
        recv_c = None
        if feats is not None:
            recv_c, _ = _AsyncExchange.apply(feats.reshape(W * Cl * Nl, -1), None, in_s, out_s, col_fwd, None)

        # Source rank i contributed [C_local, N_i, *] rows; downstream wants [C_local, sum N_i, *]. One camera per rank: the
        # received rows already are in that order. Several: ONE kernel per message moves every row to its place while
        # it splits the fields (the reference - and round 1 here - concatenates per-source pieces and then makes every
        # field contiguous: at::cat + 5 copies each way, 3 ms of a 10.9 ms step at 4 M Gaussians x 4 cameras per rank).
        N = sum(self.n_per_rank)
        row_map = None if Cl == 1 else _RowMap(Cl, self.n_per_rank)

        geo_fwd.wait()
        m2, dp, cn, op, rad = _UnpackGeometry.apply(recv_g, recv_r, row_map)
        out = (rad.reshape(Cl, N, 2), m2.reshape(Cl, N, 2), dp.reshape(Cl, N), cn.reshape(Cl, N, 3), op.reshape(Cl, N))

        def features():
            if recv_c is None:
                return None
            col_fwd.wait()
            if row_map is None:
                return recv_c.reshape(1, N, recv_c.shape[-1])
            return _UnpackRows.apply(recv_c, row_map).reshape(Cl, N, recv_c.shape[-1])

        return out + (features,)


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
