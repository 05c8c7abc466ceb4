"""Multi-GPU plumbing of the path: one process per GPU, the scene replicated, work sharded
with no data-path collective except ONE reduce of the accumulated images at read-back time and, once per step, the
SUM all-reduce of the light-histogram increments (cells x emissives x 4 bytes) that keeps every replica's sampling
distributions equal to a single process's.

  image path   : samples sharded by index -- rank r of N renders samples r, r+N, r+2N, ...; each rank keeps its
                 own running-mean image; the mean over ranks (equal sample counts) is one SUM reduce / N
  lightmap path: 64x64-texel tiles round-robin over the ranks (or contiguous ranges); disjoint writes, no reduction (gather only)

torch.distributed is plumbing here (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""


def sample_indices(rank, world, steps):
    """global sample indices rank `rank` renders in `steps` steps (disjoint across ranks, union = 0..steps*world-1)"""
    return [i * world + rank for i in range(steps)]


def sample_range(rank, world, total):
    """strong scaling (BASELINE config 3: "64 spp sharded"): contiguous [begin, end) of `total` samples for this rank,
    sizes differing by at most one, union = 0..total-1"""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def tile_owner(tile_index, world):
    """lightmap bake (BASELINE config 4): 64x64-texel tiles round-robin over the ranks, so that the unmapped
    texels, which cluster in the late pages, spread evenly"""
    return tile_index % world


def tiles_of(rank, world, lm_count, lm_size, tile=64):
    """tile indices (page * tilesPerPage + ty * tilesPerRow + tx) rank `rank` bakes"""
    per_row = (lm_size + tile - 1) // tile
    total = lm_count * per_row * per_row
    return list(range(rank, total, world))


def texel_range(rank, world, total, tile=64 * 64):
    """contiguous [begin, end) of the lmCount*lmSize^2 bake work items for this rank, aligned to whole tiles"""
    tiles = (total + tile - 1) // tile
    per = (tiles + world - 1) // world
    begin = min(total, rank * per * tile)
    end = min(total, (rank + 1) * per * tile)
    return begin, end


def _through_host(tensor, dist):
    """a device tensor under the gloo backend (rehearsals of the N-process flow on a box with fewer GPUs than ranks): the
    collective runs on a host copy"""
    return tensor.is_cuda and dist.get_backend() == "gloo"


def reduce_mean_image(tensor, world, dist, dst=0, weight=None, total=None):
    """the path's one exchange step: tensor holds this rank's mean image; after the call rank `dst` holds the mean over
    all ranks' samples.  Equal sample counts (weight None, or weight * world == total): SUM reduce, then / world.
    Unequal counts (strong scaling with a remainder): every rank scales by weight / total first."""
    if world > 1:
        equal = weight is None or total is None or weight * world == total
        if not equal:
            tensor.mul_(float(weight) / float(total))
        if _through_host(tensor, dist):
            host = tensor.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            tensor.copy_(host)
        else:
            dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
        if equal and dist.get_rank() == dst:
            tensor.mul_(1.0 / world)
    return tensor


def allreduce_light_histograms(live, base, world, dist):
    """LightOnHit's feedback (path_tracer.c:1709-1733) lands in the replica of the rank that traced the path.  Before
    UpdateDists every rank therefore adds the increments the OTHER ranks made since the last UpdateDists:
    live' = base + SUM over ranks of (live - base), base = the histograms as the last UpdateDists left them (identical on all
    ranks).  Integer adds, so live' is exactly what one process tracing all the ranks' paths would hold, on every rank;
    UpdateDists then gives every replica the same pdf / cdf bit for bit.  `live`, `base`: int32 tensors (u32 bits; the
    arithmetic wraps the same way) on the collective's device.  Returns live'."""
    if world <= 1:
        return live
    delta = live - base
    if _through_host(delta, dist):
        host = delta.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        delta.copy_(host)
    else:
        dist.all_reduce(delta, op=dist.ReduceOp.SUM)
    return base + delta
