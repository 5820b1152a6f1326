"""Multi-GPU sharding of the path-tracing path: one process per GPU, samples-per-pixel
partitioned across ranks, one sum-reduce of the full-frame radiance buffers to rank 0.

Every (pixel, sample) is independent in the reference; the only coupling is the per-pixel
average and the clamp that follows it (scene.cpp:112-113), so partial sums are reduced
UNCLAMPED and resolved once on rank 0.  The sampler is keyed on the global sample index, so
every rank derives the same stratum permutation and takes its slice of it.
"""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def sample_slice(rank, world, total):
    """Contiguous, balanced slice [begin, end) of the sample indices 0..total-1 for `rank`."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad rank / world / total")
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_radiance(buf, dst=0):
    """Sum the ranks' full-frame radiance buffers into rank `dst` (RCCL on GPUs, gloo on CPU).

    A direct reduce-to-root: on MI355X's point-to-point xGMI mesh the root receives its peers'
    buffers over distinct links, and there is one reduce per frame, so no bucketing."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
    return buf


def render_sharded(render_slice, total_samples, buf, rank=None, world=None, dst=0):
    """Run `render_slice(begin, end, buf)` on this rank's sample slice, then reduce into `dst`.

    `render_slice` must ADD the un-normalised radiance of samples [begin, end) into `buf`."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    b, e = sample_slice(rank, world, total_samples)
    if e > b:
        render_slice(b, e, buf)
    return reduce_radiance(buf, dst)
