"""Multi-GPU sharding of the path-tracing path, one process per GPU.  Two partitions:

* a still frame: samples-per-pixel partitioned across ranks, one sum-reduce of the full-frame
  radiance buffers to rank 0;
* an animation (the reference driver's loop over `time`, main.cpp:83-101): whole frames dealt
  round-robin, frame f to rank f mod world.  Frames are independent, so the data path has no
  collective; the finished 8-bit frames are gathered to rank 0 only to be written in order.

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


def frame_slice(rank, world, nframes):
    """Frame numbers of `rank`: f with f mod world == rank, ascending."""
    if world <= 0 or not (0 <= rank < world) or nframes < 0:
        raise ValueError("bad rank / world / nframes")
    return list(range(rank, nframes, world))


def frame_times(nframes, dt=0.015, t0=0.0):
    """The clock of the reference's animation loop (main.cpp:83: `time += 0.015` from 0), as the
    same running sum so that frame f sees the bit-identical `time` on whichever rank renders it."""
    out, t = [], float(t0)
    for _ in range(nframes):
        out.append(t)
        t += dt
    return out


def render_frames(render_frame, nframes, rank=None, world=None, dst=0, dt=0.015):
    """Render an animation frame-partitioned: this rank calls `render_frame(f, time)` for its
    frames (each call returns the frame as an array) and rank `dst` receives all of them in frame
    order; other ranks get None.  No collective runs while frames are being rendered."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if on else 0
    if world is None:
        world = dist.get_world_size() if on else 1
    times = frame_times(nframes, dt)
    mine = [(f, render_frame(f, times[f])) for f in frame_slice(rank, world, nframes)]
    if not on or world == 1:
        return [img for _, img in mine]
    parts = [None] * world if rank == dst else None
    dist.gather_object(mine, parts, dst=dst)
    if rank != dst:
        return None
    frames = [None] * nframes
    for part in parts:
        for f, img in part:
            frames[f] = img
    return frames


class FrameSink:
    """Takes finished 8-bit frames off the rendering stream so that frame f's trip to the host and
    whatever is done with it there (`consume(f, array)`, e.g. the AVI writer: main.cpp:87-98 does
    this inline, frame after frame) overlap the rendering of frame f + 1.

    The renderer resolves into `device_buffer(f)` (two device buffers alternate) and calls
    `submit(f)`; a side stream copies the buffer to pinned host memory once the rendering stream
    has reached that point, and a worker thread waits for the copy and calls `consume`.  `close()`
    drains.  torch is used for the buffers, the streams and the events only."""

    def __init__(self, height, width, consume, device):
        import queue
        import threading
        import torch
        self._torch = torch
        self._dev = [torch.empty(height, width, 3, dtype=torch.uint8, device=device) for _ in range(2)]
        self._host = [torch.empty(height, width, 3, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._free = [threading.Semaphore(1), threading.Semaphore(1)]        # a buffer pair is reused only after its frame was consumed
        self._copy = torch.cuda.Stream(device=device)
        self._q = queue.Queue()
        self._consume = consume
        self._err = None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def device_buffer(self, f):
        """The device buffer frame f is to be resolved into (blocks while its previous frame is still on its way)."""
        self._free[f & 1].acquire()
        return self._dev[f & 1]

    def submit(self, f):
        torch = self._torch
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._copy):
            self._copy.wait_event(done)
            self._host[f & 1].copy_(self._dev[f & 1], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self._copy)
        self._q.put((f, copied))

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            f, copied = item
            try:
                copied.synchronize()
                self._consume(f, self._host[f & 1].numpy())
            except Exception as e:           # reported by close()
                self._err = e
            finally:
                self._free[f & 1].release()

    def close(self):
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err
