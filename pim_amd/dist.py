"""Multi-GPU plumbing of the path: one process per GPU, the scene replicated, work sharded
with no data-path collective except ONE reduce of the accumulated images at read-back time.

  image path   : samples sharded by index -- rank r of N renders samples r, r+N, r+2N, ...; each rank keeps its
                 own running-mean image; the mean over ranks (equal sample counts) is one SUM reduce / N
  lightmap path: texel range sharded in contiguous tiles; disjoint writes, no reduction (gather only)

torch.distributed is plumbing here (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""


def sample_indices(rank, world, steps):
    """global sample indices rank `rank` renders in `steps` steps (disjoint across ranks, union = 0..steps*world-1)"""
    return [i * world + rank for i in range(steps)]


def texel_range(rank, world, total, tile=64 * 64):
    """contiguous [begin, end) of the lmCount*lmSize^2 bake work items for this rank, aligned to whole tiles"""
    tiles = (total + tile - 1) // tile
    per = (tiles + world - 1) // world
    begin = min(total, rank * per * tile)
    end = min(total, (rank + 1) * per * tile)
    return begin, end


def reduce_mean_image(tensor, world, dist, dst=0):
    """the path's one exchange step: tensor holds this rank's mean image (equal sample counts on every rank);
    after the call rank `dst` holds the mean over all ranks' samples"""
    if world > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            tensor.mul_(1.0 / world)
    return tensor
