"""Multi-GPU partitioning of the unprojection path (one process per GPU,
``torch.distributed``; backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` in
the CPU tests).

Every output voxel is independent, so the path shards two ways:

* **frames x proposals** (``shard_range`` / ``shard_batch``): each rank takes a
  contiguous slice of the batch items (frames with all their views' heatmaps, or
  windows with their frames).  No data-path collective: the volumes stay
  rank-local for the per-sample networks that consume them, exactly like the
  reference's DDP data split (``tools/train.py:140``).
* **z-slabs** of one dense grid (``zslab_range`` / ``gather_z_slabs``): rank r
  computes planes ``[z0, z1)`` of every (b, c, x, y) column from replicated
  heatmaps -- for the kernels that is just a shorter z axis -- and ONE
  all-gather reassembles ``[B,C,X,Y,Z]``.  z is the fastest axis of the volume,
  so the gathered ``[R,B,C,X,Y,Z/R]`` buffer is permuted once on the device.

Nothing here touches the kernels' arithmetic; results are bit-identical to the
unsharded call.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced slice ``[lo, hi)`` of ``n`` items for ``rank``."""
    if not 0 <= rank < world:
        raise ValueError(f'rank {rank} outside world of {world}')
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(feature_maps, per_item, rank, world):
    """Slice the batch dimension: ``feature_maps`` (list of ``[B,C,H,W]``) and
    every sequence in ``per_item`` (centers, scales, cameras, proposals ...)."""
    B = feature_maps[0].shape[0]
    lo, hi = shard_range(B, rank, world)
    fms = [fm[lo:hi] for fm in feature_maps]
    rest = [p[lo:hi] for p in per_item]
    return fms, rest, (lo, hi)


def zslab_range(Z, rank, world):
    """Planes ``[z0, z1)`` of rank ``rank``; Z must divide evenly so that the
    all-gather needs no padding."""
    if Z % world != 0:
        raise ValueError(f'{Z} z-planes do not split evenly over {world} ranks')
    step = Z // world
    return rank * step, (rank + 1) * step


def allgather_slabs(local, group=None, out=None):
    """The collective alone: z-slabs ``[B,C,X,Y,Zr]`` (same shape on every rank) ->
    ``[R,B,C,X,Y,Zr]`` on every rank (ONE ``all_gather_into_tensor``)."""
    world = dist.get_world_size(group)
    local = local.contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.view(-1), group=group)
    return out


def assemble_z_slabs(slabs, out=None):
    """``[R,B,C,X,Y,Zr]`` -> ``[B,C,X,Y,R*Zr]``: z is the fastest axis of the volume, so the
    rank index moves inside the row.  One HIP kernel on the device (``tempo_assemble_zslabs``);
    CPU tensors (the gloo tests) go through the equivalent torch permute."""
    if slabs.is_cuda and slabs.dtype == torch.float32:
        from . import ops
        return ops.assemble_zslabs(slabs.contiguous(), out)
    R, B, C, X, Y, Zr = slabs.shape
    res = slabs.permute(1, 2, 3, 4, 0, 5).reshape(B, C, X, Y, R * Zr)
    if out is not None:
        out.copy_(res)
        return out
    return res


def gather_z_slabs(local, group=None):
    """All-gather z-slabs ``[B,C,X,Y,Zr]`` (same shape on every rank) into the
    full ``[B,C,X,Y,Zr*R]`` volume, on every rank."""
    return assemble_z_slabs(allgather_slabs(local, group))


def unproject_root_zslab(layer, feature_maps, grid_size, grid_center, cube_size,
                         centers, scales, cameras, group=None):
    """Dense-grid stress path: every rank holds all heatmaps, computes its
    z-slab with ``layer`` (a ``ProjectLayer``-compatible module) and the slabs
    are all-gathered.  ``layer.compute_axes`` is not needed: a z-slab is the
    same grid with a shorter z axis, which the module accepts as ``zslab``."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    z0, z1 = zslab_range(cube_size[2], rank, world)
    local, _ = layer.get_voxel(feature_maps, grid_size, grid_center, cube_size, centers,
                               scales, cameras, zslab=(z0, z1))
    return gather_z_slabs(local, group)
